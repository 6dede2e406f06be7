"""does the captured forward+loss+backward produce the eager gradients?  Same parameters, same batch, no optimiser step in between:
eager vs eager (run-to-run noise), eager vs replay, replay vs replay, for two different batches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mapfree_reloc_amd.config import get_cfg_defaults
from mapfree_reloc_amd.regression.train import SyntheticPairs, Trainer
from oracle.gen_rpr_golden import CASES

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
cfg = get_cfg_defaults()
cfg.merge_from_list(CASES["3d3d"][0])
cfg.merge_from_list(["TRAINING.LR", 1e-4, "TRAINING.GRAD_CLIP", 0.0, "TRAINING.PRECISION", prec, "TRAINING.GRAPH_STEP", True])
src = SyntheticPairs(4, 96, 72, "cuda:0", seed=21)
b0, b1 = src.batch(), src.batch()
tr = Trainer(cfg, "cuda:0", sample=b0).build()
tr.model.train()
keys = sorted(b0)
rel = lambda a, b: float((a - b).norm() / a.norm())


def eager(b):
    l = tr._fwd_bwd(b)
    return tr._flat.clone(), l[2].item()


e0, le0 = eager(b0)
e0b, _ = eager(b0)
e1, le1 = eager(b1)
print("eager vs eager (same batch)", rel(e0, e0b), "| eager batch0 vs batch1", rel(e0, e1), flush=True)
# capture (3 warm-up runs + capture on b0) without any optimiser step
from mapfree_reloc_amd.nets.graph import GraphedCall
g = GraphedCall(lambda *ts: tr._fwd_bwd(dict(zip(keys, ts))), [b0[k] for k in keys], warmup=3, clone_outputs=True)
l = g(*[b0[k] for k in keys]); r0 = tr._flat.clone()
l1 = g(*[b1[k] for k in keys]); r1 = tr._flat.clone()
l0b = g(*[b0[k] for k in keys]); r0b = tr._flat.clone()
print("replay vs eager batch0", rel(e0, r0), "loss", le0, l[2].item())
print("replay vs eager batch1", rel(e1, r1), "loss", le1, l1[2].item())
print("replay vs replay batch0", rel(r0, r0b))
e0c, _ = eager(b0)
print("eager after replays vs eager before", rel(e0, e0c))
