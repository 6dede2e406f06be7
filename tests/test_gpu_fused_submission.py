"""-m gpu: submission.predict_fused with the real FusedPosePipeline (BASELINE configs[3] in small):
Precomputed correspondences + every solver -> byte-equal to the per-pair plugin loop; online SuperGlue / LoFTR
matcher stages -> same frames, same poses as the per-pair loop within the fp32 matcher's batch-shape noise."""
import os
import zipfile

import numpy as np
import pytest
import torch

from mapfree_reloc_amd import submission, synth, wire
from mapfree_reloc_amd.builder import build_model
from mapfree_reloc_amd.config import get_cfg_defaults
from mapfree_reloc_amd.datasets import collate_batch1

pytestmark = pytest.mark.gpu


class _CorrScene:
    """synthetic-correspondence scene with the dataset interface PairBatchLoader / make_loader need"""

    def __init__(self, root, sid, seeds, n_list):
        self.scene_id, self.scene_root = sid, os.path.join(str(root), sid)
        os.makedirs(self.scene_root, exist_ok=True)
        self.pairs = [synth.make_pair(s, max(n, 8), outlier_frac=0.3) for s, n in zip(seeds, n_list)]
        rows = [np.concatenate([p["pts0"], p["pts1"]], 1)[:n] if n else np.full((1, 4), np.nan) for p, n in zip(self.pairs, n_list)]
        # wire format rows are indexed by pair_id = index * 5 (quirk Q4): fill the skipped rows with NaN
        full = []
        for r in rows:
            full.append(r); full.extend([np.full((1, 4), np.nan)] * 4)
        wire.save_correspondences(os.path.join(self.scene_root, "correspondences_SG.npz"), full)

    def __len__(self):
        return len(self.pairs)

    def pair_name(self, i):
        return f"seq1/frame_{5 * i:05d}.jpg"

    def __getitem__(self, i):
        p = self.pairs[i]
        H, W = p["depth0"].shape
        return {"image0": torch.zeros(1, H, W), "image1": torch.zeros(1, H, W),
                "depth0": torch.from_numpy(p["depth0"]), "depth1": torch.from_numpy(p["depth1"]),
                "K_color0": torch.from_numpy(p["K0"]), "K_color1": torch.from_numpy(p["K1"]), "pair_id": 5 * i,
                "scene_id": self.scene_id, "scene_root": self.scene_root, "pair_names": ("seq0/frame_00000.jpg", self.pair_name(i))}


def _cfg(solver, matcher="Precomputed"):
    cfg = get_cfg_defaults()
    cfg.MODEL, cfg.FEATURE_MATCHING, cfg.POSE_SOLVER = "FeatureMatching", matcher, solver
    cfg.MATCHES_FILE_PATH = "{scene_root}/correspondences_SG.npz"
    cfg.PNP.RANSAC_ITER, cfg.PNP.REPROJECTION_INLIER_THRESHOLD, cfg.PNP.CONFIDENCE = 1000, 3, 0.9999
    cfg.EMAT_RANSAC.PIX_THRESHOLD, cfg.EMAT_RANSAC.SCALE_THRESHOLD, cfg.EMAT_RANSAC.CONFIDENCE = 2.0, 0.1, 0.9999
    cfg.PROCRUSTES.MAX_CORR_DIST = 0.05
    cfg.ALLOW_SYNTHETIC_WEIGHTS = True
    return cfg


@pytest.mark.parametrize("solver", ["PNP", "EssentialMatrixMetric", "EssentialMatrix", "Procrustes"])
def test_precomputed_fused_equals_per_pair_loop_bytewise(tmp_path, solver):
    scenes = [_CorrScene(tmp_path, "s00001", [1, 2, 3], [600, 0, 3]), _CorrScene(tmp_path, "s00002", [4, 5, 6, 7, 8], [200, 900, 450, 5, 300])]
    cfg = _cfg(solver)
    z = submission.predict_fused(cfg, "test", tmp_path / "fused", batch_pairs=4, scenes=scenes)
    model = build_model(cfg)
    res = submission.predict((collate_batch1(sc[i]) for sc in scenes for i in range(len(sc))), model)
    submission.save_submission(res, tmp_path / "loop.zip", deterministic=True)
    with zipfile.ZipFile(z) as a, zipfile.ZipFile(tmp_path / "loop.zip") as b:
        assert a.namelist() == b.namelist() and len(a.namelist()) == 2
        for n in a.namelist():
            assert a.read(n) == b.read(n), n
    assert open(z, "rb").read() == open(tmp_path / "loop.zip", "rb").read()


@pytest.mark.parametrize("matcher,solver", [("SuperGlue", "PNP"), ("LoFTR", "EssentialMatrixMetric")])
def test_online_fused_vs_per_pair_loop(tmp_path, matcher, solver):
    cfg = _cfg(solver, matcher)
    cfg.DATASET.SYNTHETIC = [2, 3]; cfg.DATASET.HEIGHT = 720; cfg.DATASET.WIDTH = 540
    z = submission.predict_fused(cfg, "test", tmp_path / "fused", batch_pairs=2)
    from mapfree_reloc_amd.datasets import make_loader
    res = submission.predict(make_loader(cfg, "test"), build_model(cfg))
    with zipfile.ZipFile(z) as a:
        assert a.namelist() == ["pose_s00000.txt", "pose_s00001.txt"]
        for sid in ("s00000", "s00001"):
            fl = [l.split(" ") for l in a.read(f"pose_{sid}.txt").decode().split("\n")]
            pl = [str(p).split(" ") for p in res[sid]]
            assert [f[0] for f in fl] == [p[0] for p in pl] and len(fl) == 3
            for f, p in zip(fl, pl):
                # batch 2 vs batch 1 through fp32 library GEMMs: the poses agree far below the benchmark's resolution
                np.testing.assert_allclose(np.array(f[1:8], float), np.array(p[1:8], float), atol=2e-3)
                assert abs(int(f[8]) - int(p[8])) <= max(3, int(0.02 * int(p[8])))


def test_submission_cli_fused_on_synthetic(tmp_path):
    """python -m mapfree_reloc_amd.submission <yaml> --synthetic 2 2 --fused end to end (single process): config merge, scene
    listing, pinned prefetching loader, fused pipeline, per-scene files, zip"""
    y = tmp_path / "sg_pnp.yaml"
    y.write_text("MODEL: 'FeatureMatching'\nFEATURE_MATCHING: 'SuperGlue'\nPOSE_SOLVER: 'PNP'\nALLOW_SYNTHETIC_WEIGHTS: True\n"
                 "DATASET:\n  HEIGHT: 720\n  WIDTH: 540\nPNP:\n  RANSAC_ITER: 1000\n  REPROJECTION_INLIER_THRESHOLD: 3\n  CONFIDENCE: 0.9999\n")
    out = tmp_path / "res"
    submission.main([str(y), "-o", str(out), "--fused", "--synthetic", "2", "2", "--batch_pairs", "2"])
    with zipfile.ZipFile(out / "submission.zip") as z:
        assert z.namelist() == ["pose_s00000.txt", "pose_s00001.txt"]
        lines = z.read("pose_s00001.txt").decode().split("\n")
    assert len(lines) == 2 and lines[0].startswith("seq1/frame_00000.jpg ") and len(lines[0].split(" ")) == 9
    assert (out / "poses" / "pose_s00000.txt").exists()


class _SharedRefScene:
    """a scene whose pairs all have the SAME reference image (what every Map-free val / test scene is, mapfree.py:148-165): three queries
    of one synthetic scene (plain, with moving objects + occluder, with corrupted depth on top)"""
    shared_reference = True

    def __init__(self, seed, sid):
        from mapfree_reloc_amd import images as IM
        self.scene_id, self.scene_root = sid, f"/synthetic/{sid}"
        self.p = [IM.synthetic_pair(seed, 720, 540, hard=h) for h in (0, 1, 2)]
        self.img0 = torch.from_numpy(self.p[0]["img0"])[None].expand(3, -1, -1).contiguous()        # ONE tensor for all pairs
        assert all(np.array_equal(q["img0"], self.p[0]["img0"]) for q in self.p)

    def __len__(self):
        return len(self.p)

    def pair_name(self, i):
        return f"seq1/frame_{5 * i:05d}.jpg"

    def __getitem__(self, i):
        q = self.p[i]
        return {"image0": self.img0, "image1": torch.from_numpy(q["img1"])[None].expand(3, -1, -1).contiguous(),
                "depth0": torch.from_numpy(q["depth0"]), "depth1": torch.from_numpy(q["depth1"]),
                "K_color0": torch.from_numpy(q["K"].copy()), "K_color1": torch.from_numpy(q["K"].copy()), "pair_id": 5 * i,
                "scene_id": self.scene_id, "scene_root": self.scene_root, "pair_names": ("seq0/frame_00000.jpg", self.pair_name(i))}


def test_reference_view_feature_cache_is_bitwise_neutral(tmp_path):
    """SuperPoint once per distinct reference view (HIP.REF_FEATURE_CACHE) vs once per pair: the same archive byte for byte, on two
    scenes whose batches (4 pairs) straddle the scene boundary, so both the in-batch grouping and the cross-batch cache are used"""
    from mapfree_reloc_amd.pipeline import FusedPosePipeline
    scenes = [_SharedRefScene(11, "s00000"), _SharedRefScene(12, "s00001")]
    zs, stats = [], None
    for cache in (True, False):
        cfg = _cfg("PNP", "SuperGlue")
        cfg.HIP.REF_FEATURE_CACHE = cache
        pipe = FusedPosePipeline(cfg)
        zs.append(submission.predict_fused(cfg, "test", tmp_path / f"c{int(cache)}", pipeline=pipe, batch_pairs=4, scenes=scenes))
        if cache:
            stats = pipe.match.stats
    assert stats == {"reference_views_run": 2, "reference_views_reused": 4}, stats
    assert open(zs[0], "rb").read() == open(zs[1], "rb").read()
    with zipfile.ZipFile(zs[0]) as a:
        lines = a.read("pose_s00000.txt").decode().split("\n")
    assert len(lines) == 3 and all("nan" not in l for l in lines)                  # real poses were compared, not failures
