"""CPU: (1) the oracle restatement of the correlation-volume aggregator (oracle/rpr_ref.py) against outputs AND gradients of
the reference's own CorrelationVolumeWarping (tests/golden/ref_rpr_aggregator.npz, written by oracle/gen_rpr_golden.py by
executing /root/reference/lib/models/regression);  (2) this package's encoder / head / losses / model wiring against the
reference's RegressionModel end to end (ref_rpr_model_*.npz): eval forward, training forward, loss and backward -- with the
fused-kernel aggregator swapped for the oracle's materialised one (the HIP kernel itself is checked in the -m gpu tests)."""
import os

import numpy as np
import pytest
import torch

from mapfree_reloc_amd.config import get_cfg_defaults
from mapfree_reloc_amd.regression.model import RegressionModel
from oracle import rpr_ref
from oracle.gen_rpr_golden import CASES

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("name,kw", [("full", {}), ("half", {"half": True}), ("nopos", {"position_encoder": False}),
                                     ("norm_im1", {"normalise": True, "im1": True}), ("nomax", {"max_score": False})])
def test_oracle_aggregator_matches_reference(name, kw):
    g = np.load(os.path.join(GOLD, "ref_rpr_aggregator.npz"))
    v0, v1 = _t(g[f"{name}_vol0"]).requires_grad_(), _t(g[f"{name}_vol1"]).requires_grad_()
    y = rpr_ref.aggregate_materialised(v0, v1, **kw)
    np.testing.assert_allclose(y.detach().numpy(), g[f"{name}_out"], rtol=1e-5, atol=1e-6)
    (y * _t(g[f"{name}_w"])).sum().backward()
    np.testing.assert_allclose(v0.grad.numpy(), g[f"{name}_dvol0"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(v1.grad.numpy(), g[f"{name}_dvol1"], rtol=1e-4, atol=2e-6)


def build_case(name, device="cpu", materialised=True):
    ov, B, H, W = CASES[name]
    cfg = get_cfg_defaults()
    cfg.merge_from_list(ov)
    model = RegressionModel(cfg)
    g = np.load(os.path.join(GOLD, f"ref_rpr_model_{name}.npz"))
    if materialised and cfg.AGGREGATOR.TYPE == "CorrelationVolumeWarping":
        model.aggregator = rpr_ref.MaterialisedAggregator(model.aggregator)
    model = model.to(device)
    data = {k: _t(g[k]).to(device) for k in ("image0", "image1", "T_0to1")}
    model.train()
    with torch.no_grad():
        model(dict(data))                         # materialise lazy layers exactly like the generator
    rpr_ref.fill_deterministic(model, seed=7)
    return model, data, g


def check_case(model, data, g, tol):
    model.eval()
    with torch.no_grad():
        d = dict(data)
        vol0 = model.encoder(d["image0"])
        np.testing.assert_allclose(vol0.float().cpu().numpy(), g["eval_vol0"], rtol=tol, atol=tol)
        R, t = model(d)
        np.testing.assert_allclose(R.cpu().numpy(), g["eval_R"], atol=10 * tol)
        np.testing.assert_allclose(t.cpu().numpy(), g["eval_t"], atol=10 * tol)
        for k, v in zip(("R_loss", "t_loss", "loss"), model.loss_fn(d)):
            np.testing.assert_allclose(v.reshape(-1).cpu().numpy(), g[f"eval_{k}"], rtol=20 * tol, atol=20 * tol)
    model.train()
    d = dict(data)
    d["image0"] = d["image0"].clone().requires_grad_()
    R, t = model(d)
    np.testing.assert_allclose(R.detach().cpu().numpy(), g["train_R"], atol=20 * tol)
    np.testing.assert_allclose(t.detach().cpu().numpy(), g["train_t"], atol=20 * tol)
    R_loss, t_loss, loss = model.loss_fn(d)
    np.testing.assert_allclose(loss.detach().reshape(-1).cpu().numpy(), g["train_loss"], rtol=20 * tol, atol=20 * tol)
    loss.sum().backward()
    gi = d["image0"].grad.cpu().numpy()
    ref = g["grad_image0"]
    assert np.abs(gi - ref).max() <= 50 * tol * max(np.abs(ref).max(), 1e-6), (np.abs(gi - ref).max(), np.abs(ref).max())
    params = dict(model.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    have = sorted(n for n, p in params.items() if p.grad is not None)
    assert have == names                                      # same parameter names receive gradients: state-dict compatible
    norms = np.array([float(params[n].grad.double().norm()) for n in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=100 * tol, atol=1e-6 * g["grad_norms"].max())  # conv biases under BatchNorm have zero true gradient: pure round-off
    for key in g.files:
        if key.startswith("grad::"):
            ref = g[key]
            got = params[key[6:]].grad.cpu().numpy()
            assert np.abs(got - ref).max() <= 100 * tol * max(np.abs(ref).max(), 1e-7), key


@pytest.mark.parametrize("name", ["3d3d", "qkv_bins", "concat_resnet"])
def test_model_matches_reference_cpu(name):
    if name == "qkv_bins":
        pytest.skip("QKV aggregator runs through the HIP kernel only; covered by the -m gpu test")
    torch.manual_seed(0)
    model, data, g = build_case(name)
    check_case(model, data, g, tol=2e-5)


def test_state_dict_keys_are_the_references():
    """every parameter of the reference model exists here under the same state-dict key (checkpoint compatibility)"""
    for name, (ov, *_ ) in CASES.items():
        cfg = get_cfg_defaults()
        cfg.merge_from_list(ov)
        g = np.load(os.path.join(GOLD, f"ref_rpr_model_{name}.npz"))
        assert {str(n) for n in g["grad_names"]} <= {n for n, _ in RegressionModel(cfg).named_parameters()}, name
