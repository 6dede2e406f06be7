"""-m gpu: the reference's two-stage route and this package's fused online route are the SAME estimator on the same files
(VERDICT r4 missing-4).

Reference flow: compute.py (etc/feature_matching_baselines/compute.py:72-86: the matcher reads seq0/frame_00000.jpg and every seq1 frame
through read_image -- an 8-bit gray plane / 255, matchers.py:101-104) -> correspondences_SG.npz -> submission.py with
FEATURE_MATCHING = 'Precomputed' + a solver.  Fused flow: submission.predict_fused with FEATURE_MATCHING = 'SuperGlue': batched loaders
decode the same files, the matcher runs online, the solver consumes its device-resident matches.

On a small Map-free tree of real (lossy, coloured) JPEG files + uint16 depth PNGs the two must produce identical correspondence sets and
bit-equal poses: same gray plane on both routes (datasets.gray_plane), batch-shape-independent kernels, the same counter-based RANSAC
stream per pair_id."""
import os
import zipfile

import numpy as np
import pytest
import torch

from mapfree_reloc_amd import compute, images as IM, submission, wire
from mapfree_reloc_amd.builder import build_model
from mapfree_reloc_amd.config import get_cfg_defaults
from mapfree_reloc_amd.datasets import make_loader

pytestmark = pytest.mark.gpu


def _write_tree(root, n_scenes=2, frames=(11, 6)):
    """val/<scene>/{seq0,seq1}/frame_XXXXX.jpg (JPEG quality 92 of TINTED views: R, G, B differ, so the luma is not a byte already),
    frame_XXXXX.dptkitti.png (uint16 mm), poses.txt, intrinsics.txt; pairs = keyframe x every 5th seq1 frame (mapfree.py:148-165)"""
    from PIL import Image
    for s in range(n_scenes):
        sc = root / "val" / f"s{s:05d}"
        (sc / "seq0").mkdir(parents=True); (sc / "seq1").mkdir()
        seed = 100 * s + 7
        ref = IM.synthetic_pair(seed)

        def save(gray, depth, rel):
            rgb = np.stack([0.92 * gray + 0.03, gray, 0.8 * gray + 0.1 * gray * gray], -1)
            Image.fromarray(np.round(np.clip(rgb, 0, 1) * 255).astype(np.uint8)).save(sc / rel, format="JPEG", quality=92)
            Image.fromarray(np.round(depth * 1000).astype(np.uint16)).save(str(sc / rel).replace(".jpg", ".dptkitti.png"))
        lp, lk = ["# frame qw qx qy qz tx ty tz"], ["# frame fx fy cx cy W H"]
        K = ref["K"]
        save(ref["img0"], ref["depth0"], "seq0/frame_00000.jpg")
        lp.append("seq0/frame_00000.jpg 1 0 0 0 0 0 0"); lk.append(f"seq0/frame_00000.jpg {K[0, 0]} {K[1, 1]} {K[0, 2]} {K[1, 2]} 540 720")
        for i in range(frames[s]):
            # the sampled frames (every 5th) are second views of the keyframe's scene -- plain, with moving objects + an occluder, with
            # corrupted depth on top (images.synthetic_pair hard = 0 / 1 / 2: the same first view); the others show other scenes
            q = IM.synthetic_pair(seed, hard=(i // 5) % 3) if i % 5 == 0 else IM.synthetic_pair(seed + 31 + i)
            save(q["img1"], q["depth1"], f"seq1/frame_{i:05d}.jpg")
            lp.append(f"seq1/frame_{i:05d}.jpg 1 0 0 0 {-q['t_gt'][0]} 0 0"); lk.append(f"seq1/frame_{i:05d}.jpg {K[0, 0]} {K[1, 1]} {K[0, 2]} {K[1, 2]} 540 720")
        (sc / "poses.txt").write_text("\n".join(lp) + "\n"); (sc / "intrinsics.txt").write_text("\n".join(lk) + "\n")


def _cfg(root, matcher):
    cfg = get_cfg_defaults()
    cfg.MODEL, cfg.FEATURE_MATCHING, cfg.POSE_SOLVER = "FeatureMatching", matcher, "PNP"
    cfg.MATCHES_FILE_PATH = "{scene_root}/correspondences_SG.npz"
    cfg.PNP.RANSAC_ITER, cfg.PNP.REPROJECTION_INLIER_THRESHOLD, cfg.PNP.CONFIDENCE = 1000, 3, 0.9999
    cfg.DATASET.DATA_ROOT, cfg.DATASET.HEIGHT, cfg.DATASET.WIDTH, cfg.DATASET.ESTIMATED_DEPTH = str(root), 720, 540, "dptkitti"
    cfg.ALLOW_SYNTHETIC_WEIGHTS = True
    return cfg


def test_offline_route_equals_fused_online_route(tmp_path):
    _write_tree(tmp_path)
    # ---- route A, the reference's: offline matcher over EVERY seq1 frame -> npz -> Precomputed + PnP, one pair at a time
    compute.main(["-ds", "Mapfree", "-m", "SG", "--data_root", str(tmp_path)])
    cfg_a = _cfg(tmp_path, "Precomputed")
    res = submission.predict(make_loader(cfg_a, "val"), build_model(cfg_a))
    submission.save_submission(res, tmp_path / "offline.zip", deterministic=True)
    # ---- route B: batched loaders + online SuperPoint / SuperGlue + PnP (batches of 3 pairs: they straddle the scene boundary)
    cfg_b = _cfg(tmp_path, "SuperGlue")
    from mapfree_reloc_amd.pipeline import FusedPosePipeline
    pipe = FusedPosePipeline(cfg_b)
    zb = submission.predict_fused(cfg_b, "val", tmp_path / "fused", pipeline=pipe, batch_pairs=3)
    with zipfile.ZipFile(tmp_path / "offline.zip") as a, zipfile.ZipFile(zb) as b:
        assert a.namelist() == b.namelist() == ["pose_s00000.txt", "pose_s00001.txt"]
        for n in a.namelist():
            la, lb = a.read(n).decode().split("\n"), b.read(n).decode().split("\n")
            assert len(la) == len(lb) and len(la) >= 2
            assert la == lb, (n, [(x, y) for x, y in zip(la, lb) if x != y][:2])                 # frames, poses (every printed digit) and inlier counts
        first = a.read("pose_s00000.txt").decode().split("\n")[0].split(" ")
        assert first[0] == "seq1/frame_00000.jpg" and int(first[8]) > 100 and "nan" not in first      # a real pose was compared, not two failures
    # ---- and the correspondence sets themselves: the npz rows of the sampled frames == what the fused matcher stage produces on the
    # loader's batch (rows of the wire format are indexed by pair_id = 5 * index, quirk Q4)
    from mapfree_reloc_amd.datasets import list_scenes, PairBatchLoader, DevicePrefetcher
    scenes = list_scenes(cfg_b, "val")
    loader = PairBatchLoader(scenes, 4, prefetch=1, pin=True, global_offsets=[0, len(scenes[0])], workers=2, decode="thread")
    seen = 0
    for batch in DevicePrefetcher(loader, pipe.device):
        m = pipe.match(batch)
        torch.cuda.synchronize()
        for p in range(len(batch["seed_ids"])):
            root = batch["scene_roots"][p]
            corr = np.load(os.path.join(root, "correspondences_SG.npz"))["correspondences"]
            want0, want1 = wire.strip_nan(corr[int(batch["seed_ids"][p])].astype(np.float32))
            n = int(m["n_corr"][p])
            assert n == len(want0), (root, int(batch["seed_ids"][p]), n, len(want0))
            assert np.array_equal(m["pts0"][p, :n].cpu().numpy(), want0) and np.array_equal(m["pts1"][p, :n].cpu().numpy(), want1)
            seen += 1
    loader.close()
    assert seen == len(scenes[0]) + len(scenes[1]) == 5
