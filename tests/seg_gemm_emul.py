"""numpy statement of what csrc/conv_gemm_bf16.hip computes from its arguments (address arithmetic included), used by the CPU tests
to check the host-side layouts of regression/conv_bf16.py (haloed images, segment tables, tap shifts, K splits) without a GPU.
TEST INFRASTRUCTURE ONLY."""
import numpy as np
import torch


def _f32(t):
    return t.detach().float().numpy() if t.dtype == torch.bfloat16 else t.detach().numpy()


def seg_gemm(A, a_off, sA, segA, Bm, b_off, sB, segB, Lk, nkc_total, nkc_z, bias, C, ldc, M, N, nz=1, zA=None, zB=None, zC=None, zk=None):
    a, b = _f32(A).reshape(-1), _f32(Bm).reshape(-1)
    segA, segB = segA.numpy(), segB.numpy()
    out = np.zeros((nz, M, N), np.float32)
    ii, jj, kk = np.arange(M)[:, None], np.arange(N)[:, None], np.arange(32)[None, :]
    for z in range(nz):
        za = int(zA[z]) if zA is not None else 0
        zb = int(zB[z]) if zB is not None else 0
        k0 = int(zk[z]) if zk is not None else 0
        for c in range(k0, min(k0 + nkc_z, nkc_total)):
            s, w = (32 * c) // Lk, (32 * c) % Lk
            ia = a_off + za + ii * sA + int(segA[s]) + w + kk
            ib = b_off + zb + jj * sB + int(segB[s]) + w + kk
            assert ia.min() >= 0 and ia.max() < a.size and ib.min() >= 0 and ib.max() < b.size, "operand read outside its buffer"
            assert (a_off + za + int(segA[s]) + w) % 8 == 0 and sA % 8 == 0 and (b_off + zb + int(segB[s]) + w) % 8 == 0 and sB % 8 == 0, "16-byte alignment"
            out[z] += a[ia] @ b[ib].T
        if bias is not None:
            out[z] += bias.numpy()[None, :]
    flat = C.view(-1)
    for z in range(nz):
        zc = int(zC[z]) if zC is not None else 0
        idx = torch.from_numpy((zc + np.arange(M)[:, None] * ldc + np.arange(N)[None, :]).reshape(-1))
        flat[idx] = torch.from_numpy(out[z].reshape(-1)).to(C.dtype)


def pack_nhwc_halo(x, guard_rows, Wp):
    """torch statement of mfr_conv_pack_nhwc_halo"""
    B, C, H, W = x.shape
    Mp = B * (H + 2) * Wp
    out = torch.zeros((2 * guard_rows + Mp) * C, dtype=torch.bfloat16, device=x.device)
    out[guard_rows * C:(guard_rows + Mp) * C].view(B, H + 2, Wp, C)[:, 1:H + 1, 1:W + 1].copy_(x.permute(0, 2, 3, 1))
    return out


def pack_cm_halo(x, Wq, L, ncopies, first_shift, slack):
    """torch statement of mfr_conv_pack_cm_halo (positions a shift moves across a row end are zero)"""
    B, C, H, W = x.shape
    Hp = H + 2
    hal = torch.zeros(B * C, Hp, Wq, dtype=torch.bfloat16, device=x.device)
    hal[:, 1:H + 1, 1:W + 1].copy_(x.reshape(B * C, H, W))
    out = torch.zeros(ncopies, 2 * slack + B * C * L, dtype=torch.bfloat16, device=x.device)
    for k in range(ncopies):
        s = first_shift + k
        sh = torch.zeros_like(hal)
        if s >= 0:
            sh[:, :, :Wq - s] = hal[:, :, s:]
        else:
            sh[:, :, -s:] = hal[:, :, :Wq + s]
        img = out[k, slack:slack + B * C * L].view(B * C, L)
        img[:, :Hp * Wq] = sh.view(B * C, Hp * Wq)
    return out


def unpack_nchw(haloed, B, N, H, W, Wp):
    """torch statement of mfr_conv_unpack_nchw"""
    return haloed.view(B, H + 2, Wp, N)[:, 1:H + 1, 1:W + 1].permute(0, 3, 1, 2).contiguous()
