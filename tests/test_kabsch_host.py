"""CPU: the arithmetic of csrc/kabsch.hip (csrc/kabsch_math.h: Horn rotation + closed-form gradient through the polar factor)
compiled with gcc exactly as the device compiles it, against the reference's formulation (lib/utils/solver.py:4-37: SVD +
reflection fix) and ITS autograd gradient, in binary64 -- including reflected (det < 0) and near-degenerate inputs."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "map-free-reloc_amd", "csrc")
_HOST = r'''
#include "kabsch_math.h"
void kb_fwd(const double *H, int B, double *R) { for (int b = 0; b < B; ++b) kb_horn_rotation(H + 9 * b, R + 9 * b); }
void kb_bwd(const double *H, const double *G, int B, double *gH) { for (int b = 0; b < B; ++b) kb_rotation_backward(H + 9 * b, G + 9 * b, gH + 9 * b); }
'''


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    d = tmp_path_factory.mktemp("kb")
    (d / "kb_host.c").write_text(_HOST)
    subprocess.run(["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-shared", "-fPIC", f"-I{CSRC}", "-o", str(d / "kb_host.so"), str(d / "kb_host.c"), "-lm"],
                   check=True)
    return ctypes.CDLL(str(d / "kb_host.so"))


def _svd_rotation(H):
    U, _, Vh = torch.linalg.svd(H)
    flip = torch.ones(H.shape[0], 3, dtype=H.dtype)
    flip[:, 2] = torch.sign(torch.linalg.det(U @ Vh))
    return (Vh.transpose(1, 2) * flip[:, None, :]) @ U.transpose(1, 2)


def _run(host, H, G):
    B = H.shape[0]
    Hn, Gn = np.ascontiguousarray(H.numpy().reshape(B, 9)), np.ascontiguousarray(G.numpy().reshape(B, 9))
    R, g = np.zeros((B, 9)), np.zeros((B, 9))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    host.kb_fwd(p(Hn), B, p(R)); host.kb_bwd(p(Hn), p(Gn), B, p(g))
    return R.reshape(B, 3, 3), g.reshape(B, 3, 3)


def test_rotation_and_gradient_equal_the_svd_route(host):
    torch.manual_seed(1)
    B = 300
    H = torch.randn(B, 3, 3, dtype=torch.float64)
    H[:40] = H[:40] @ torch.diag(torch.tensor([1.0, 1.0, -1.0], dtype=torch.float64))          # reflected configurations
    H[40:60] *= 1e-3; H[60:80] *= 1e3                                                          # scale invariance of R
    G = torch.randn(B, 3, 3, dtype=torch.float64)
    Hr = H.clone().requires_grad_()
    R = _svd_rotation(Hr)
    (R * G).sum().backward()
    Rk, gk = _run(host, H, G)
    assert np.abs(Rk - R.detach().numpy()).max() < 1e-11
    assert np.abs(np.linalg.det(Rk) - 1).max() < 1e-12
    ref = Hr.grad.numpy()
    assert (np.abs(gk - ref).reshape(B, -1).max(1) <= 1e-9 * np.maximum(1.0, np.abs(ref).reshape(B, -1).max(1))).all()


def test_points_route_matches_reference_procrustes(host):
    """from point sets, like the heads call it: rotation AND translation of lib/utils/solver.py's procrustes"""
    from mapfree_reloc_amd.regression.geometry import procrustes
    g = torch.Generator().manual_seed(3)
    A = torch.randn(16, 6, 3, generator=g, dtype=torch.float64)
    Rt = _svd_rotation(torch.randn(16, 3, 3, generator=g, dtype=torch.float64))
    Bp = A @ Rt.transpose(1, 2) + torch.randn(16, 1, 3, generator=g, dtype=torch.float64) + 0.01 * torch.randn(16, 6, 3, generator=g, dtype=torch.float64)
    R, t = procrustes(A, Bp)                                    # host tensors: the SVD branch (the reference's own arithmetic)
    a0, b0 = A.mean(1, keepdim=True), Bp.mean(1, keepdim=True)
    H = (A - a0).transpose(1, 2) @ (Bp - b0)
    Rk, _ = _run(host, H, torch.zeros_like(H))
    assert np.abs(Rk - R.numpy()).max() < 1e-10
    assert np.abs((b0 - a0 @ torch.from_numpy(Rk).transpose(1, 2)).numpy() - t.numpy()).max() < 1e-10


def test_degenerate_inputs_stay_finite(host):
    H = torch.zeros(4, 3, 3, dtype=torch.float64)
    H[1] = torch.diag(torch.tensor([1.0, 1.0, 0.0], dtype=torch.float64))                       # planar
    H[2] = torch.diag(torch.tensor([1.0, 0.5, -0.5], dtype=torch.float64))                      # s_2 + s_3 = 0 after the reflection fix
    H[3] = torch.eye(3, dtype=torch.float64)
    R, g = _run(host, H, torch.ones_like(H))
    assert np.isfinite(R).all() and np.isfinite(g).all()
    assert np.abs(np.linalg.det(R[1:]) - 1).max() < 1e-12
