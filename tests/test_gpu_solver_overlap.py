"""-m gpu: pipeline.SolverOverlap (overlap_solver=True: the RANSAC stage of call i on a second HIP stream under the matcher of call i + 1).  Several
calls issued back to back WITHOUT any synchronisation in between, one join at the end (what bench.py does), must give the very bits of the one-stream
pipeline for every call -- also when the batches alternate (nothing of call i may be overwritten by call i + 1 before the solver has read it)."""
import numpy as np
import pytest
import torch

from mapfree_reloc_amd import images as IM
from mapfree_reloc_amd.pipeline import LoFTREmatPipeline, SuperGluePnPPipeline

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batches(seeds_list):
    out = []
    for seeds in seeds_list:
        sb = IM.synthetic_batch(seeds, hard=1)
        out.append({k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in sb.items() if isinstance(v, np.ndarray)})
    return out


@pytest.mark.parametrize("kind", ["sg_pnp", "loftr_emat"])
def test_overlapped_solver_stage_equals_one_stream(kind):
    bs = _batches([[5000, 5001, 5002], [5003, 5004, 5005], [5006, 5007, 5008]])
    if kind == "sg_pnp":
        mk = lambda ov: SuperGluePnPPipeline(DEV, overlap_solver=ov)
        call = lambda p, d: p(d["images"], d["depth0"], d["K0"], d["K1"], d["pair_ids"])
    else:
        mk = lambda ov: LoFTREmatPipeline(DEV, overlap_solver=ov)
        call = lambda p, d: p(d["images"], d["depth0"], d["depth1"], d["K0"], d["K1"], d["pair_ids"])
    plain, over = mk(False), mk(True)
    order = [0, 1, 2, 0, 2, 1, 1]
    want = []
    for i in order:
        o = call(plain, bs[i])
        torch.cuda.synchronize()
        want.append({k: v.clone() for k, v in o.items() if isinstance(v, torch.Tensor)})
    got = [call(over, bs[i]) for i in order]              # no synchronisation between the calls
    over.join()
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        for k in ("status", "n_inliers", "n_corr", "R", "t", "pts0", "pts1"):
            a, b = w[k], g[k]
            assert torch.equal(torch.nan_to_num(a.double(), nan=-7.0), torch.nan_to_num(b.double(), nan=-7.0)), k
    assert int((want[0]["status"] == 0).sum()) >= 2
