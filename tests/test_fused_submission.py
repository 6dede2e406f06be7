"""Host logic of the scene-sharded fused submission driver (submission.predict_fused, SURVEY.md 8e / configs[3]),
CPU only: a deterministic stand-in for the GPU pipeline is injected so that sharding, batching, per-scene atomic
files, resume, the one all_gather (gloo, world size 2) and the rank-0 zip are exercised end to end.  The real
pipeline behind the same function is covered by tests/test_gpu_fused_submission.py."""
import os
import subprocess
import sys
import zipfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_STUB = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch


class StubPipeline:
    """deterministic function of the batch contents (images, depth, K, RANSAC stream ids) standing in for
    FusedPosePipeline: rotation about z by an angle derived from the image sum, every 7th pair fails"""
    device = "cpu"

    def __init__(self):
        self.calls = []

    def __call__(self, b):
        n = b["seed_ids"].numel()
        self.calls.append((b["scene_id"], n))
        assert b["images"].shape[0] == 2 * n and b["depth0"].shape[0] == n and b["K0"].shape == (n, 3, 3)
        s = b["images"].double().reshape(n, -1).sum(1) + b["depth0"].double().reshape(n, -1).mean(1) + b["seed_ids"].double()
        a = (s %% 1.0) * 0.5
        R = torch.zeros(n, 3, 3, dtype=torch.float64)
        R[:, 0, 0] = torch.cos(a); R[:, 0, 1] = -torch.sin(a); R[:, 1, 0] = torch.sin(a); R[:, 1, 1] = torch.cos(a); R[:, 2, 2] = 1
        t = torch.stack([s %% 3.0, (s * 7) %% 2.0, b["K0"][:, 0, 0].double() / 600.0], 1)
        st = ((b["global_ids"] %% 7) == 3).to(torch.int32)
        R[st != 0] = float("nan"); t[st != 0] = float("nan")
        return dict(R=R, t=t, n_inliers=(b["global_ids"] * 3 + 11).to(torch.int32), status=st)
'''

_WORKER = _STUB + r'''
import os
from pathlib import Path
import torch.distributed as dist
from mapfree_reloc_amd import submission
from mapfree_reloc_amd.config import get_cfg_defaults
out_root = Path(sys.argv[1])
dist.init_process_group("gloo")
cfg = get_cfg_defaults()
cfg.DATASET.SYNTHETIC = [5, 3]; cfg.DATASET.HEIGHT = 48; cfg.DATASET.WIDTH = 40
pipe = StubPipeline()
z = submission.predict_fused(cfg, "test", out_root, pipeline=pipe, batch_pairs=2, prefetch=1)
assert (z is not None) == (dist.get_rank() == 0)
print("rank", dist.get_rank(), "scenes", sorted({c[0] for c in pipe.calls}), "ok")
dist.destroy_process_group()
'''


def _cfg(n_scenes=5, frames=3):
    from mapfree_reloc_amd.config import get_cfg_defaults
    cfg = get_cfg_defaults()
    cfg.DATASET.SYNTHETIC = [n_scenes, frames]; cfg.DATASET.HEIGHT = 48; cfg.DATASET.WIDTH = 40
    return cfg


def _stub():
    ns = {}
    exec(_STUB % ROOT, ns)
    return ns["StubPipeline"]()


def test_predict_fused_world1_zip_layout_and_resume(tmp_path):
    from mapfree_reloc_amd import submission
    pipe = _stub()
    z = submission.predict_fused(_cfg(), "test", tmp_path / "a", pipeline=pipe, batch_pairs=2, prefetch=0)
    # batches span scene boundaries (5 scenes x 3 pairs at B=2: seven full batches + the rank's last one), no per-scene tail batch
    assert [c[1] for c in pipe.calls] == [2] * 7 + [1]
    assert submission.LAST_RUN_STATS["pairs"] == 15 and submission.LAST_RUN_STATS["batches"] == 8
    with zipfile.ZipFile(z) as zf:
        assert zf.namelist() == [f"pose_s{i:05d}.txt" for i in range(5)]
        lines = zf.read("pose_s00001.txt").decode().split("\n")
    # global ids 3,4,5 -> pair 3 fails (id % 7 == 3) and is left out like submission.py:48-49
    assert [l.split(" ")[0] for l in lines] == ["seq1/frame_00005.jpg", "seq1/frame_00010.jpg"]
    f = lines[0].split(" ")
    assert len(f) == 9 and all(len(x.split(".")[1]) == 6 for x in f[1:8]) and f[8] == str(4 * 3 + 11)
    q = np.array(list(map(float, f[1:5])))
    assert abs(np.linalg.norm(q) - 1) < 1e-5 and q[0] >= 0
    # the per-scene files are the resume markers: a second run touches nothing and re-creates the same archive
    pipe2 = _stub()
    z2 = submission.predict_fused(_cfg(), "test", tmp_path / "a", pipeline=pipe2, batch_pairs=2, prefetch=0)
    assert pipe2.calls == [] and open(z2, "rb").read() == open(z, "rb").read()
    # losing one scene file -> only that scene is recomputed
    os.remove(tmp_path / "a" / "poses" / "pose_s00003.txt")
    pipe3 = _stub()
    submission.predict_fused(_cfg(), "test", tmp_path / "a", pipeline=pipe3, batch_pairs=4, prefetch=2)
    assert [c[0] for c in pipe3.calls] == ["s00003"]
    assert open(tmp_path / "a" / "submission.zip", "rb").read() == open(z, "rb").read()
    # every call leaves one line in the run log beside the pose files: what it computed and how long the loop took
    import json
    log = [json.loads(l) for l in open(tmp_path / "a" / "poses" / "run_log_rank0.jsonl")]
    assert [r["pairs"] for r in log] == [15, 0, 3] and [r["scenes_computed"] for r in log] == [5, 0, 1]
    assert all(r["seconds"] >= 0 and r["world"] == 1 and r["decode"] in ("process", "thread") for r in log)
    # ... but only for the SAME configuration: pose files left by another solver / matcher / threshold are stale and recomputed
    # (the manifest beside them carries a hash of the merged configuration and the split), never mixed into the new archive
    cfg2 = _cfg()
    cfg2.PNP.REPROJECTION_INLIER_THRESHOLD = 5
    pipe4 = _stub()
    submission.predict_fused(cfg2, "test", tmp_path / "a", pipeline=pipe4, batch_pairs=4, prefetch=0)
    assert sum(c[1] for c in pipe4.calls) == 15 and submission.LAST_RUN_STATS["scenes_computed"] == 5


def test_predict_fused_gloo_world2_equals_world1(tmp_path):
    from mapfree_reloc_amd import submission
    z1 = submission.predict_fused(_cfg(), "test", tmp_path / "w1", pipeline=_stub(), batch_pairs=2, prefetch=0)
    script = tmp_path / "w.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", str(script), str(tmp_path / "w2")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2
    # contiguous scene blocks balanced by pair count: 5 scenes x 3 pairs -> rank 0: s0..s2 (or s0,s1), rank 1 the rest
    assert "rank 0 scenes ['s00000', 's00001'" in r.stdout and "'s00004']" in r.stdout.split("rank 1 scenes")[1]
    assert open(tmp_path / "w2" / "submission.zip", "rb").read() == open(z1, "rb").read()


def test_fused_lines_equal_the_per_pair_loop(tmp_path):
    """predict_fused and the reference-style per-pair loop `predict` format the same pose to the same text"""
    from mapfree_reloc_amd import submission
    from mapfree_reloc_amd.datasets import make_loader, list_scenes, PairBatchLoader
    cfg = _cfg(2, 3)
    pipe = _stub()
    z = submission.predict_fused(cfg, "test", tmp_path / "f", pipeline=pipe, batch_pairs=3, prefetch=0)
    # per-pair loop: a model that evaluates the same stand-in on the single-pair batch the loader would build
    scenes = list_scenes(cfg, "test")
    batches = list(PairBatchLoader(scenes, 1, prefetch=0, pin=False))

    class M:
        i = 0

        def __call__(self, data):
            o = _stub()(batches[self.i]); self.i += 1
            data["inliers"] = int(o["n_inliers"][0])
            return o["R"].float(), o["t"].float()[:, None]
    res = submission.predict(make_loader(cfg, "test"), M())
    submission.save_submission(res, tmp_path / "p.zip", deterministic=True)
    assert open(tmp_path / "p.zip", "rb").read() == open(z, "rb").read()


def test_missing_data_root_is_an_error_not_a_silent_synthetic_run(tmp_path):
    from mapfree_reloc_amd.config import get_cfg_defaults
    from mapfree_reloc_amd.datasets import make_loader, MissingDataError
    cfg = get_cfg_defaults()
    cfg.DATASET.DATA_ROOT = str(tmp_path / "nope"); cfg.DATASET.HEIGHT = 720; cfg.DATASET.WIDTH = 540
    with pytest.raises(MissingDataError):
        make_loader(cfg, "val")
    cfg.DATASET.SYNTHETIC = [1, 2]
    assert len(list(make_loader(cfg, "val"))) == 2


def test_submission_cli_needs_explicit_synthetic_and_weights(tmp_path):
    """submission.main on a config without data: error; and the online matchers refuse to invent weights"""
    from mapfree_reloc_amd import submission
    from mapfree_reloc_amd.datasets import MissingDataError
    y = tmp_path / "c.yaml"
    y.write_text("MODEL: 'FeatureMatching'\nFEATURE_MATCHING: 'Precomputed'\nPOSE_SOLVER: 'PNP'\nMATCHES_FILE_PATH: '{scene_root}/c.npz'\n"
                 "DATASET:\n  DATA_ROOT: '%s'\n  HEIGHT: 720\n  WIDTH: 540\n"
                 "PNP:\n  RANSAC_ITER: 1000\n  REPROJECTION_INLIER_THRESHOLD: 3\n  CONFIDENCE: 0.9999\n" % (tmp_path / "none"))
    with pytest.raises(MissingDataError):
        submission.main([str(y), "-o", str(tmp_path / "o"), "--fused"])
    from mapfree_reloc_amd.config import get_cfg_defaults
    from mapfree_reloc_amd.nets import weights as WT
    cfg = get_cfg_defaults()
    with pytest.raises(FileNotFoundError):
        WT.synthetic_or_raise("SuperPoint", cfg, WT.superpoint_state_dict)
    cfg.ALLOW_SYNTHETIC_WEIGHTS = True
    with pytest.warns(UserWarning):
        assert "conv1a.weight" in WT.synthetic_or_raise("SuperPoint", cfg, WT.superpoint_state_dict)


def test_pair_batch_loader_threads_and_order(tmp_path):
    from mapfree_reloc_amd.datasets import list_scenes, PairBatchLoader, DevicePrefetcher
    scenes = list_scenes(_cfg(3, 5), "test")
    a = list(PairBatchLoader(scenes, 2, prefetch=0, pin=False))
    b = list(DevicePrefetcher(PairBatchLoader(scenes, 2, prefetch=3, pin=False), "cpu"))
    # 3 scenes x 5 pairs at B=2, batches span scenes: 7 full batches + 1; scene ends are reported per batch
    assert len(a) == len(b) == 8 and [x["global_ids"].tolist() for x in a] == [x["global_ids"].tolist() for x in b]
    assert torch.cat([x["global_ids"] for x in a]).tolist() == list(range(15))
    assert all(torch.equal(x["images"], y["images"]) and x["names"] == y["names"] for x, y in zip(a, b))
    assert a[2]["scenes_done"] == ["s00000"] and a[2]["scene_ids"] == ["s00000", "s00001"] and a[2]["seed_ids"].tolist() == [20, 0]
    assert a[1]["scenes_done"] == [] and a[7]["scenes_done"] == ["s00002"] and [len(x["names"]) for x in a] == [2] * 7 + [1]
    # per-scene batches on request (the pre-round-3 behaviour)
    c = list(PairBatchLoader(scenes, 2, prefetch=0, pin=False, span_scenes=False))
    assert len(c) == 9 and c[2]["last_of_scene"] and not c[1]["last_of_scene"] and c[2]["seed_ids"].tolist() == [20]
    assert a[0]["images"].shape == (4, 1, 48, 40) and a[0]["images"].dtype == torch.float32


def test_checkpoint_layouts_load_through_the_matcher_path(tmp_path):
    """upstream file layouts (matchers.py:16-18, 65-71): superpoint_v1.pth / superglue_*.pth are flat state dicts,
    LoFTR's *_ot.ckpt is a Lightning checkpoint {'state_dict': {'matcher.<key>': ...}} with extra keys the reference
    tolerates through strict=False.  The loader must give back exactly the tensors that were saved."""
    from mapfree_reloc_amd.nets import weights as WT
    sp, sg, lo = WT.superpoint_state_dict(), WT.superglue_state_dict(), WT.loftr_state_dict()
    torch.save(sp, tmp_path / "superpoint_v1.pth"); torch.save(sg, tmp_path / "superglue_indoor.pth")
    ck = {"epoch": 3, "global_step": 7, "state_dict": {**{"matcher." + k: v for k, v in lo.items()},
                                                       "matcher.coarse_matching.bin_score": torch.tensor(1.0)}}   # OT head of the *_ot files
    torch.save(ck, tmp_path / "indoor_ot.ckpt")
    a = WT.load_checkpoint(tmp_path / "superpoint_v1.pth")
    assert a.keys() == sp.keys() and all(torch.equal(a[k], sp[k]) for k in sp)
    b = WT.load_checkpoint(tmp_path / "superglue_indoor.pth")
    assert b.keys() == sg.keys() and torch.equal(b["gnn.layers.17.mlp.3.weight"], sg["gnn.layers.17.mlp.3.weight"])
    c = WT.strip_prefix(WT.load_checkpoint(tmp_path / "indoor_ot.ckpt"), "matcher.")
    assert set(lo) <= set(c) and "coarse_matching.bin_score" in c and all(torch.equal(c[k], lo[k]) for k in lo)
    # the folded SuperGlue operands built from the re-loaded file equal the ones built from memory
    from mapfree_reloc_amd.nets.superglue import fold_weights
    f0, f1 = fold_weights(sg, 2), fold_weights(b, 2)
    assert torch.equal(f0["layers"][1]["w1"], f1["layers"][1]["w1"]) and torch.equal(f0["wf"], f1["wf"])
    # a pickled-code payload is refused (weights_only)
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    with open(tmp_path / "evil.pth", "wb") as f:
        pickle.dump({"state_dict": Evil()}, f)
    with pytest.raises(Exception):
        WT.load_checkpoint(tmp_path / "evil.pth")
