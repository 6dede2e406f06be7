"""CPU: this package's Map-free pair lists (training overlap window, val/test sub-sampling, multi-frame windows), the rescaled
float64 intrinsics and the scene-balanced sampler against outputs of the REFERENCE'S OWN lib/datasets/{mapfree,sampler}.py executed in
the build container (tests/golden/ref_data_pairs.npz, written by oracle/gen_data_golden.py)."""
import os

import numpy as np
import pytest

from mapfree_reloc_amd.datasets import MapFreeScene, MapFreeSceneMultiFrame, SceneBalancedSampler
from oracle.gen_data_golden import write_tree

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_data_pairs.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def test_training_and_val_pair_lists_equal_the_reference(g, tmp_path):
    T, n_frames = int(g["T"]), int(g["n_frames"])
    lim = tuple(float(v) for v in g["limits"])
    write_tree(tmp_path / "train", n_frames, g["idxs"], g["overlaps"], train=True)
    write_tree(tmp_path / "val", n_frames, g["idxs"], g["overlaps"], train=False)
    sc = MapFreeScene(tmp_path / "train", (270, 360), 1, None, lim)
    assert np.array_equal(np.asarray(sc.pairs, np.int64), g["train_single"])
    K = sc.K["seq0/frame_00000.jpg"]
    assert K.dtype == np.float64 and np.array_equal(K, g["K_first"])                      # bit-equal rescaled intrinsics
    mf = MapFreeSceneMultiFrame(tmp_path / "train", (270, 360), T, None, lim)
    assert np.array_equal(np.asarray([p[:3] for p in mf.pairs], np.int64).reshape(-1, 3), g["train_multi_head"])
    assert np.array_equal(np.asarray([p[3] for p in mf.pairs], np.int64).reshape(-1, T), g["train_multi_window"])
    vs = MapFreeScene(tmp_path / "val", (540, 720), 5)
    assert np.array_equal(np.asarray(vs.pairs, np.int64), g["val_single"])
    vm = MapFreeSceneMultiFrame(tmp_path / "val", (540, 720), T)
    assert np.array_equal(np.asarray([p[3] for p in vm.pairs], np.int64).reshape(-1, T), g["val_multi_window"])


@pytest.mark.parametrize("tag,repl", [("repl", True), ("norepl", False)])
def test_scene_balanced_sampler_equals_the_reference_stream(g, tag, repl):
    """same generator seed (66), same draw order: the epoch index lists are IDENTICAL to RandomConcatSampler's, epoch after epoch"""
    s = SceneBalancedSampler([int(v) for v in g["sampler_sizes"]], 6, repl)
    assert np.array_equal(np.asarray(list(s), np.int64), g[f"sampler_{tag}_e0"])
    assert np.array_equal(np.asarray(list(s), np.int64), g[f"sampler_{tag}_e1"])


def test_pose_lines_and_zip_members_equal_the_reference(tmp_path):
    """submission.Pose.__str__ / save_submission against the reference's own submission.py executed on the same seeded poses
    (tests/golden/ref_submission_format.npz, oracle/gen_submission_golden.py): every line byte for byte (float32 and float64 inputs,
    negative zero, tiny / large magnitudes, int / float / numpy confidences), member names, order and contents of the archive"""
    import zipfile
    from mapfree_reloc_amd.submission import Pose, save_submission
    from oracle.gen_submission_golden import cases
    gz = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_submission_format.npz"))
    cs = cases()
    assert [str(Pose(n, q, t, c)) for n, q, t, c in cs] == gz["lines"].tolist()
    res = {"s00460": [Pose(*c) for c in cs[:10]], "s00461": [], "s00462": [Pose(*c) for c in cs[10:]]}
    save_submission(res, tmp_path / "s.zip")
    with zipfile.ZipFile(tmp_path / "s.zip") as zf:
        assert zf.namelist() == gz["members"].tolist()
        assert [zf.read(m).decode("utf-8") for m in zf.namelist()] == gz["texts"].tolist()


def test_config_schema_holds_every_reference_key_with_its_default():
    """config/default.py of the reference, executed (oracle/gen_config_golden.py): all 69 keys exist here with the same default, so
    every yaml the reference accepts merges and behaves the same; this package's additions are extra keys only"""
    import json
    from mapfree_reloc_amd.config import get_cfg_defaults
    from oracle.gen_config_golden import flat
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_config_defaults.json")))
    ours = flat(get_cfg_defaults())
    assert len(ref) == 69 and not [k for k in ref if k not in ours]
    assert {k: (ref[k], ours[k]) for k in ref if ref[k] != ours[k]} == {}
