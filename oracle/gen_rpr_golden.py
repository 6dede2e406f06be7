"""Generate tests/golden/ref_rpr_*.npz by EXECUTING THE REFERENCE'S OWN lib/models/regression code (fp32, CPU).

Runs only in the build container (needs /root/reference).  pytorch_lightning, kornia and transforms3d are not installed:
  * pytorch_lightning.LightningModule is replaced by torch.nn.Module (+ a no-op `log`) -- no arithmetic involved;
  * yacs.config.CfgNode is this repo's CfgNode (attribute bag; no arithmetic);
  * kornia / transforms3d are stubbed and any call into them raises, so nothing in these fixtures depends on them: the
    quaternion head / losses are NOT pinned here;  the 6-D `Direct*` heads are not pinned either: the reference's
    rotationutils.normalize_vector (lib/utils/rotationutils.py:14) hard-codes `.cuda()` and cannot run on this CPU box.
Weights are filled by oracle/rpr_ref.fill_deterministic (a function of parameter name and shape), so the tests rebuild
the same weights instead of loading 60 MB.

Usage: python oracle/gen_rpr_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
REF = "/root/reference"
OUT = os.path.join(HERE, "..", "tests", "golden")

from oracle.rpr_ref import fill_deterministic  # noqa: E402


def _raise(*a, **k):
    raise RuntimeError("stubbed third-party call: not available offline")


def _install_stubs():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        def log(self, *a, **k):
            pass
    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl
    for name in ("kornia", "kornia.geometry", "kornia.geometry.conversions", "transforms3d", "transforms3d.quaternions"):
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules["kornia.geometry.conversions"].quaternion_to_rotation_matrix = _raise
    sys.modules["kornia.geometry.conversions"].rotation_matrix_to_quaternion = _raise
    for fn in ("qmult", "qinverse", "rotate_vector", "quat2mat"):
        setattr(sys.modules["transforms3d.quaternions"], fn, _raise)
    # yacs is not installed either: the reference's config/default.py gets this repo's CfgNode (same attribute semantics)
    import importlib
    node = importlib.import_module("mapfree_reloc_amd.config.node")
    yc = types.ModuleType("yacs.config"); yc.CfgNode = node.CfgNode
    sys.modules["yacs"] = types.ModuleType("yacs"); sys.modules["yacs.config"] = yc
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        mp = types.ModuleType("matplotlib"); mp.pyplot = types.ModuleType("matplotlib.pyplot")
        sys.modules["matplotlib"] = mp; sys.modules["matplotlib.pyplot"] = mp.pyplot


def reference_cfg(overrides):
    from config.default import cfg as ref_cfg
    c = ref_cfg.clone()
    c.merge_from_list(overrides)
    return c


CASES = {
    # name: (cfg overrides, B, H, W)
    "3d3d": (["MODEL", "Regression", "ENCODER.TYPE", "ResUNet", "ENCODER.BLOCK_TYPE", 1, "ENCODER.NUM_BLOCKS", "1-1-1",
              "ENCODER.NOT_CONCAT", False, "ENCODER.NUM_OUT_LAYERS", 32, "AGGREGATOR.TYPE", "CorrelationVolumeWarping",
              "AGGREGATOR.POSITION_ENCODER", True, "AGGREGATOR.POSITION_ENCODER_IM1", False, "AGGREGATOR.MAX_SCORE_CHANNEL", True,
              "HEAD.TYPE", "ProcrustesDeepResBlock", "HEAD.ADD_BASIS", True, "HEAD.AVG_POOL", True,
              "TRAINING.ROT_LOSS", "rot_angle_loss", "TRAINING.TRANS_LOSS", "trans_l1_loss", "TRAINING.LAMBDA", 1.0,
              "DATASET.HEIGHT", 96, "DATASET.WIDTH", 72], 2, 96, 72),
    "qkv_bins": (["MODEL", "Regression", "ENCODER.TYPE", "ResUNet", "ENCODER.BLOCK_TYPE", 1, "ENCODER.NUM_BLOCKS", "1-1-1",
                  "ENCODER.NOT_CONCAT", False, "ENCODER.NUM_OUT_LAYERS", 32, "AGGREGATOR.TYPE", "CorrelationVolumeWarpingQKV",
                  "AGGREGATOR.POSITION_ENCODER", True, "AGGREGATOR.MAX_SCORE_CHANNEL", True, "AGGREGATOR.RESIDUAL_ATT", True,
                  "HEAD.TYPE", "AngularBinsDeepResBlockMLP", "HEAD.AVG_POOL", True, "HEAD.SEPARATE_SCALE", True,
                  "TRAINING.ROT_LOSS", "rot_bin_loss", "TRAINING.TRANS_LOSS", "trans_sphbin_loss", "TRAINING.LAMBDA", 0.0,
                  "DATASET.HEIGHT", 64, "DATASET.WIDTH", 80], 2, 64, 80),
    "concat_resnet": (["MODEL", "Regression", "ENCODER.TYPE", "ResNet", "ENCODER.BLOCK_TYPE", 0, "ENCODER.NUM_BLOCKS", "1-1-1",
                       "AGGREGATOR.TYPE", "Concat", "HEAD.TYPE", "ProcrustesResBlockMLP", "HEAD.ADD_BASIS", False,
                       "HEAD.NUM_PTS", 8, "TRAINING.ROT_LOSS", "rot_frobenius_loss", "TRAINING.TRANS_LOSS", "trans_ang_loss",
                       "TRAINING.LAMBDA", 0.5, "DATASET.HEIGHT", 128, "DATASET.WIDTH", 96], 3, 128, 96),
}


def random_pose(g, B):
    A = torch.randn(B, 3, 3, generator=g)
    Q, _ = torch.linalg.qr(A)
    Q = Q * torch.sign(torch.linalg.det(Q))[:, None, None]
    T = torch.eye(4).repeat(B, 1, 1)
    # moderate rotations: blend towards identity so the angles are not all near pi
    T[:, :3, :3] = torch.matrix_exp(0.35 * (Q - Q.transpose(1, 2)) / 2)
    T[:, :3, 3] = torch.randn(B, 3, generator=g) * 0.4
    return T


def run_model_case(name, overrides, B, H, W):
    from lib.models.regression.model import RegressionModel
    cfg = reference_cfg(overrides)
    torch.manual_seed(0)
    model = RegressionModel(cfg)
    g = torch.Generator().manual_seed(1234)
    im0 = torch.rand(B, 3, H, W, generator=g, requires_grad=True)
    im1 = torch.rand(B, 3, H, W, generator=g)
    data = {"image0": im0, "image1": im1, "T_0to1": random_pose(g, B)}
    model.train()
    with torch.no_grad():
        model(dict(data))                        # materialise LazyLinear layers
    fill_deterministic(model, seed=7)
    out = {"image0": im0.detach().numpy(), "image1": im1.numpy(), "T_0to1": data["T_0to1"].numpy()}
    # eval-mode forward (running statistics), then a training-mode forward + backward (batch statistics)
    model.eval()
    with torch.no_grad():
        d = dict(data)
        vol0 = model.encoder(im0)
        agg = model.aggregator(vol0, model.encoder(im1))
        R, t = model(d)
        out.update(eval_vol0=vol0.numpy(), eval_agg=agg.numpy(), eval_R=R.numpy(), eval_t=t.numpy())
        out.update({f"eval_{k}": v.numpy() for k, v in zip(("R_loss", "t_loss", "loss"), (x.reshape(-1) for x in model.loss_fn(d)))})
    model.train()
    d = dict(data)
    R, t = model(d)
    R_loss, t_loss, loss = model.loss_fn(d)
    loss.sum().backward()
    out.update(train_R=R.detach().numpy(), train_t=t.detach().numpy(), train_R_loss=R_loss.detach().reshape(-1).numpy(),
               train_t_loss=t_loss.detach().reshape(-1).numpy(), train_loss=loss.detach().reshape(-1).numpy(),
               grad_image0=im0.grad.numpy())
    names, norms = [], []
    for n, p in sorted(model.named_parameters()):
        if p.grad is not None:
            names.append(n); norms.append(float(p.grad.double().norm()))
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms)
    small = [n for n, p in model.named_parameters() if p.grad is not None and p.numel() <= 20000][:12]
    for n in small:
        out["grad::" + n] = dict(model.named_parameters())[n].grad.numpy()
    # BatchNorm running statistics after the one training step (momentum update)
    out["bn_after::" + "encoder.firstbn.running_mean" if hasattr(model.encoder, "firstbn") else "none"] = (
        model.encoder.firstbn.running_mean.numpy() if hasattr(model.encoder, "firstbn") else np.zeros(1))
    np.savez_compressed(os.path.join(OUT, f"ref_rpr_model_{name}.npz"), **out)
    print(name, "loss", float(loss.sum()), "params with grad", len(names), "R_loss", float(R_loss), "t_loss", float(t_loss))


def run_aggregator_cases():
    from lib.models.regression.aggregator import CorrelationVolumeWarping
    from config.default import cfg as ref_cfg
    out = {}
    variants = {"full": {}, "half": {"CV_HALF_CHANNELS": True}, "nopos": {"POSITION_ENCODER": False},
                "norm_im1": {"NORMALISE_DOT": True, "POSITION_ENCODER_IM1": True}, "nomax": {"MAX_SCORE_CHANNEL": False}}
    g = torch.Generator().manual_seed(99)
    B, D, H, W = 2, 32, 20, 15
    for name, ov in variants.items():
        c = ref_cfg.AGGREGATOR.clone()
        c.POSITION_ENCODER, c.POSITION_ENCODER_IM1, c.MAX_SCORE_CHANNEL = True, False, True
        for k, v in ov.items():
            c[k] = v
        agg = CorrelationVolumeWarping(c, D)
        v0 = (torch.randn(B, D, H, W, generator=g) * 0.7).requires_grad_()
        v1 = (torch.randn(B, D, H, W, generator=g) * 0.7).requires_grad_()
        y = agg(v0, v1)
        wgt = torch.randn(y.shape, generator=g)
        (y * wgt).sum().backward()
        out.update({f"{name}_vol0": v0.detach().numpy(), f"{name}_vol1": v1.detach().numpy(), f"{name}_out": y.detach().numpy(),
                    f"{name}_w": wgt.numpy(), f"{name}_dvol0": v0.grad.numpy(), f"{name}_dvol1": v1.grad.numpy()})
        print("aggregator", name, tuple(y.shape))
    np.savez_compressed(os.path.join(OUT, "ref_rpr_aggregator.npz"), **out)


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    run_aggregator_cases()
    for name, (ov, B, H, W) in CASES.items():
        run_model_case(name, ov, B, H, W)


if __name__ == "__main__":
    main()
