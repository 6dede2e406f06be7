"""Offline (CPU) feasibility study for the next round: what happens to SuperPoint+SuperGlue matches and PnP poses if the 3x3
convolutions of the SuperPoint encoder are computed with Winograd F(4x4,3x3) in fp32 instead of an exact fp32 accumulation?

The CPU oracle nets (oracle/nets_ref.py, PyTorch fp32) are run twice on the same synthetic pairs: once as they are, once with every
3x3 stride-1 convolution of `SuperPointRef` replaced by an fp32 emulation of F(4x4,3x3) (filters transformed in f64 and stored f32,
input / output transforms and channel accumulation in f32).  Reported per pair: keypoint-set and match-set agreement, pose delta.
usage: python tools/diag_wino_f43_matcher.py [pairs] [layers: all|low]   (test infrastructure: imports oracle/)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from mapfree_reloc_amd import images as IM
from oracle import pipeline_ref as PR

BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def conv_f43(x, w, b):
    """x [B,C,H,W] f32, w [K,C,3,3], padding 1 -> [B,K,H,W] through F(4x4,3x3), f32 arithmetic"""
    B, C, H, W = x.shape
    K = w.shape[0]
    nh, nw = (H + 3) // 4, (W + 3) // 4
    xp = F.pad(x, (1, 4 * nw - W + 1, 1, 4 * nh - H + 1))
    t = xp.unfold(2, 6, 4).unfold(3, 6, 4)                                   # [B,C,nh,nw,6,6]
    bt = BT.float()
    V = torch.einsum("ia,bcxyaj->bcxyij", bt, t)
    V = torch.einsum("bcxyij,kj->bcxyik", V, bt)
    U = (G @ w.double() @ G.T).float()                                       # [K,C,6,6], packed once
    M = torch.einsum("bcxyij,kcij->bkxyij", V, U)
    at = AT.float()
    Y = torch.einsum("ia,bkxyaj->bkxyij", at, M)
    Y = torch.einsum("bkxyij,lj->bkxyil", Y, at)                             # [B,K,nh,nw,4,4]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, K, 4 * nh, 4 * nw)[:, :, :H, :W]
    return y + b.view(1, -1, 1, 1)


class Patch:
    def __init__(self, sp, names):
        self.sp, self.names, self.saved = sp, names, {}

    def __enter__(self):
        for n in self.names:
            m = getattr(self.sp, n)
            self.saved[n] = m.forward
            m.forward = (lambda mod: (lambda x: conv_f43(x, mod.weight, mod.bias)))(m)

    def __exit__(self, *a):
        for n, f in self.saved.items():
            getattr(self.sp, n).forward = f


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    layers = {"all": ["conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convDa"],
              "low": ["conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convDa"]}[which]
    torch.set_num_threads(os.cpu_count() or 8)
    sp, sg = PR._nets("sg")
    recs = []
    for s in range(n_pairs):
        p = IM.synthetic_pair(1000 + s)
        a = PR.sg_pnp_pair(p["img0"], p["img1"], p["depth0"], p["K"], p["K"], 1000 + s)
        with Patch(sp, layers):
            b = PR.sg_pnp_pair(p["img0"], p["img1"], p["depth0"], p["K"], p["K"], 1000 + s)
        sa = {tuple(np.round(r, 3)) for r in a["pts"]}
        sb = {tuple(np.round(r, 3)) for r in b["pts"]}
        rot = float(np.arccos(np.clip((np.trace(a["R"].T @ b["R"]) - 1) / 2, -1, 1))) if a["status"] == 0 and b["status"] == 0 else None
        dt = float(np.linalg.norm(a["t"] - b["t"])) if rot is not None else None
        recs.append(dict(seed=1000 + s, matches_exact=len(sa), matches_f43=len(sb), common=len(sa & sb), identical=sa == sb, drot_rad=rot, dt_m=dt,
                         inliers=(int(a["n_inliers"]), int(b["n_inliers"]))))
        print(recs[-1], flush=True)
    print(json.dumps(dict(layers=which, pairs=n_pairs, identical_match_sets=sum(r["identical"] for r in recs),
                          mean_common_fraction=float(np.mean([r["common"] / max(1, r["matches_exact"]) for r in recs])),
                          max_drot_rad=max((r["drot_rad"] or 0) for r in recs), max_dt_m=max((r["dt_m"] or 0) for r in recs))))


if __name__ == "__main__":
    main()
