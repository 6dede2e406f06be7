"""-m gpu: the reference's plugin surface on top of the HIP path: config merge,
build_model, FeatureMatchingModel.forward contract, PrecomputedMatching from an npz on disk,
per-pair solver classes == batched kernels == oracle, NaN-pose convention."""
import os

import numpy as np
import pytest
import torch

import mapfree_reloc_amd as mfr
from mapfree_reloc_amd import synth, wire
from mapfree_reloc_amd.builder import build_model
from mapfree_reloc_amd.config import get_cfg_defaults
from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu


def _cfg(solver, tmp, matcher="Precomputed"):
    cfg = get_cfg_defaults()
    cfg.MODEL = "FeatureMatching"
    cfg.FEATURE_MATCHING = matcher
    cfg.POSE_SOLVER = solver
    cfg.MATCHES_FILE_PATH = "{scene_root}/correspondences_SG.npz"
    cfg.PNP.RANSAC_ITER, cfg.PNP.REPROJECTION_INLIER_THRESHOLD, cfg.PNP.CONFIDENCE = 1000, 3, 0.9999
    cfg.EMAT_RANSAC.PIX_THRESHOLD, cfg.EMAT_RANSAC.SCALE_THRESHOLD, cfg.EMAT_RANSAC.CONFIDENCE = 2.0, 0.1, 0.9999
    return cfg


def _scene(tmp_path, n_list, seeds):
    pairs = [synth.make_pair(s, n, outlier_frac=0.3) if n else synth.make_pair(s, 8) for s, n in zip(seeds, n_list)]
    rows = [np.concatenate([p["pts0"], p["pts1"]], 1)[:n] if n else np.full((1, 4), np.nan) for p, n in zip(pairs, n_list)]
    wire.save_correspondences(os.path.join(tmp_path, "correspondences_SG.npz"), rows)
    datas = []
    for i, p in enumerate(pairs):
        datas.append({
            "depth0": torch.from_numpy(p["depth0"])[None], "depth1": torch.from_numpy(p["depth1"])[None],
            "K_color0": torch.from_numpy(p["K0"])[None], "K_color1": torch.from_numpy(p["K1"])[None],
            "pair_id": torch.tensor([i]), "scene_id": ["s00000"], "scene_root": [str(tmp_path)],
            "pair_names": [["seq0/frame_00000.jpg"], [f"seq1/frame_{i:05d}.jpg"]]})
    return pairs, datas


def test_pnp_model_forward_contract(tmp_path):
    n_list = [600, 0, 3, 200]
    pairs, datas = _scene(tmp_path, n_list, [1, 2, 3, 4])
    model = build_model(_cfg("PNP", tmp_path))
    for i, (p, data) in enumerate(zip(pairs, datas)):
        R, t = model(data)
        assert R.shape == (1, 3, 3) and t.shape == (1, 1, 3) and R.dtype == torch.float32      # model.py:38-39
        n = n_list[i]
        st, Rr, tr, ninl = O.pnp_solve(p["pts0"][:n], p["pts1"][:n], p["depth0"], p["K0"], p["K1"], seed=0, pair_id=i)
        if st != 0:
            assert torch.isnan(R).all() and torch.isnan(t).all() and data["inliers"] == 0        # NaN convention
        else:
            assert data["inliers"] == ninl
            np.testing.assert_array_equal(R[0].numpy(), Rr.astype(np.float32))
            np.testing.assert_array_equal(t[0, 0].numpy(), tr.reshape(3).astype(np.float32))
            assert synth.rot_err_deg(Rr, p["R_gt"]) < 0.3


def test_emat_metric_model_forward(tmp_path):
    n_list = [900, 4, 300]
    pairs, datas = _scene(tmp_path, n_list, [11, 12, 13])
    model = build_model(_cfg("EssentialMatrixMetric", tmp_path))
    for i, (p, data) in enumerate(zip(pairs, datas)):
        R, t = model(data)
        n = n_list[i]
        ref = O.emat_solve(p["pts0"][:n], p["pts1"][:n], p["K0"], p["K1"], 2.0, 0.9999, 1000, 0, i)
        if ref["status"] != 0:
            assert torch.isnan(R).all() and data["inliers"] == 0
            continue
        sc = O.scale_lift(p["pts0"][:n], p["pts1"][:n], ref["mask"], p["depth0"], p["depth1"], p["K0"], p["K1"], ref["R"], ref["t"])
        cnt, bs, _ = O.scale_ransac(sc, 0.1)
        assert data["inliers"] == cnt                                                         # Q2: confidence = scale inliers
        np.testing.assert_array_equal(t[0, 0].numpy(), (bs * ref["t"]).astype(np.float32))
        np.testing.assert_array_equal(model.pose_solver.mask.ravel(), ref["mask"])              # Q7


def test_emat_up_to_scale_solver_shapes(tmp_path):
    pairs, datas = _scene(tmp_path, [500], [21])
    model = build_model(_cfg("EssentialMatrix", tmp_path))
    R, t = model(datas[0])
    assert abs(float(t.norm()) - 1.0) < 1e-5 and datas[0]["inliers"] > 100


def test_unknown_config_values_raise(tmp_path):
    with pytest.raises(NotImplementedError):
        build_model(_cfg("Nope", tmp_path))
    cfg = _cfg("PNP", tmp_path); cfg.MODEL = "Regression"
    with pytest.raises(NotImplementedError):
        build_model(cfg)


def test_batch1_graph_replay_equals_eager():
    """the batch-1 online SuperGlue matcher replayed from a captured HIP graph (nets/graph.py) returns exactly the eager result,
    also after the static input buffer has been overwritten with another pair"""
    from mapfree_reloc_amd.datasets import SyntheticScene, collate_batch1
    from mapfree_reloc_amd.matching.feature_matching import SuperGlueMatching
    cfg = _cfg("PNP", "", matcher="SuperGlue")
    cfg.ALLOW_SYNTHETIC_WEIGHTS = True
    sc = SyntheticScene(3, frames=3)
    samples = [collate_batch1(sc[i]) for i in range(3)]
    cfg.HIP.GRAPH_BATCH1 = False
    eager = SuperGlueMatching(cfg)
    cfg.HIP.GRAPH_BATCH1 = True
    graphed = SuperGlueMatching(cfg)
    for s in samples + samples[:1]:
        a0, a1 = eager.get_correspondences(s)
        b0, b1 = graphed.get_correspondences(s)
        assert len(a0) > 100 and np.array_equal(a0, b0) and np.array_equal(a1, b1)


def test_batch1_loftr_graph_replay_equals_eager():
    """LoFTRMatching: the stage before the match count (backbone incl. the library's stride-2 convolutions, coarse transformer,
    dual-softmax matching) replayed from a HIP graph, the fine stage eager on the graph's static buffers -- identical to the fully
    eager matcher, pair after pair with changing inputs and when a pair comes back"""
    from mapfree_reloc_amd.datasets import SyntheticScene, collate_batch1
    from mapfree_reloc_amd.matching.feature_matching import LoFTRMatching
    cfg = _cfg("EssentialMatrixMetric", "", matcher="LoFTR")
    cfg.ALLOW_SYNTHETIC_WEIGHTS = True
    sc = SyntheticScene(5, frames=3)
    samples = [collate_batch1(sc[i]) for i in range(3)]
    cfg.HIP.GRAPH_BATCH1 = False
    eager = LoFTRMatching(cfg)
    cfg.HIP.GRAPH_BATCH1 = True
    graphed = LoFTRMatching(cfg)
    for s in samples + samples[:2]:
        a0, a1 = eager.get_correspondences(s)
        b0, b1 = graphed.get_correspondences(s)
        assert len(a0) > 100 and np.array_equal(a0, b0) and np.array_equal(a1, b1)
    assert graphed.use_graph and len(graphed._graphs) == 1


@pytest.mark.parametrize("B", [1, 3])
def test_fused_pipeline_graph_replay_equals_eager(B):
    """SuperGluePnPPipeline(graph=True): the whole step replayed from one HIP graph returns exactly the eager results, batch after
    batch with changing inputs (the regression this guards: memset NODES did not re-zero the library's counters on replay)"""
    from mapfree_reloc_amd import images as IM
    from mapfree_reloc_amd.pipeline import SuperGluePnPPipeline
    dev = torch.device("cuda:0")
    eager, graphed = SuperGluePnPPipeline(dev), SuperGluePnPPipeline(dev, graph=True)
    keys = ("images", "depth0", "K0", "K1", "pair_ids")
    kept = []
    for k in range(3):
        sb = IM.synthetic_batch([50 * k + i for i in range(B)])
        d = {key: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for key, v in sb.items()}
        a = eager(*[d[key] for key in keys])
        b = graphed(*[d[key] for key in keys])
        kept.append((a, b))
    torch.cuda.synchronize()
    for a, b in kept:                                         # results handed out earlier are copies, not the static buffers
        assert int(a["n_corr"].min()) > 20
        for key in ("n_corr", "n_inliers", "status", "pts0", "pts1"):
            assert torch.equal(a[key], b[key]), key
        assert torch.equal(torch.nan_to_num(a["R"]), torch.nan_to_num(b["R"])) and torch.equal(torch.nan_to_num(a["t"]), torch.nan_to_num(b["t"]))


_CAPTURE_FAIL = r"""
import sys
sys.path.insert(0, %r)
import torch
from mapfree_reloc_amd.nets.graph import GraphCaptureError, GraphedCall
dev = torch.device("cuda:0")
x = torch.arange(1024, device=dev, dtype=torch.float32)
before = torch.cuda.current_stream()
try:
    GraphedCall(lambda t: t * float(t.sum().item()), [x])          # .item() synchronises: illegal while capturing
    print("NO ERROR")
except GraphCaptureError as e:
    print("raised", type(e).__name__)
assert torch.cuda.current_stream() == before
assert float((x * 2).sum().item()) == 2.0 * 1023 * 512             # the process still launches and synchronises
g = GraphedCall(lambda t: t * 3 + 1, [x])
assert torch.equal(g(x + 1), (x + 1) * 3 + 1)
print("usable")
"""


def test_failed_graph_capture_falls_back_and_leaves_the_process_usable(tmp_path):
    """a function that cannot be captured (host synchronisation inside) raises GraphCaptureError, the current stream is restored
    and later work -- eager and captured -- runs normally.  Runs in a child process: a capture that cannot be unwound must not
    take the test session with it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CAPTURE_FAIL % root], capture_output=True, text=True, timeout=300)
    assert "raised GraphCaptureError" in r.stdout and "usable" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]
