"""Generate tests/golden/ref_data_pairs.npz by EXECUTING THE REFERENCE'S OWN dataset / sampler code (build container only):

  * lib/datasets/mapfree.py  MapFreeScene.load_pairs (training branch: overlaps.npz + overlap window; val/test branch) and
    MapFreeSceneMultiFrame (sample_offset branches), read_intrinsics (float64 rescaled K);
  * lib/datasets/sampler.py  RandomConcatSampler (scene-balanced sampling, with / without replacement).

cv2, pytorch_lightning and transforms3d are stubbed at import (none of them is touched by the code paths executed here).
tests/test_data_golden.py rebuilds the same synthetic trees from the arrays stored in the fixture and compares this package's
datasets.MapFreeScene / MapFreeSceneMultiFrame / SceneBalancedSampler with the reference's outputs."""
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: None


def write_tree(root, n_frames, idxs, overlaps, train=True):
    """the files the reference's MapFreeScene.__init__ reads (no images needed for the pair lists)"""
    root = Path(root)
    root.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(1)
    lp, lk = ["# name qw qx qy qz tx ty tz"], ["# name fx fy cx cy W H"]
    for s in (0, 1):
        for i in (range(n_frames) if (train or s == 1) else range(1)):
            nme = f"seq{s}/frame_{i:05d}.jpg"
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            lp.append(nme + " " + " ".join(f"{v:.8f}" for v in np.r_[q, rng.normal(size=3)]))
            lk.append(nme + f" {500.0 + i} {510.0 + i} 269.3 359.7 540 720")
    (root / "poses.txt").write_text("\n".join(lp) + "\n")
    (root / "poses_device.txt").write_text("\n".join(lp) + "\n")
    (root / "intrinsics.txt").write_text("\n".join(lk) + "\n")
    if train:
        np.savez(root / "overlaps.npz", idxs=idxs, overlaps=overlaps)


def main():
    for name in ("cv2", "pytorch_lightning", "transforms3d", "transforms3d.quaternions", "tqdm"):
        sys.modules[name] = _Stub(name)
    sys.modules["pytorch_lightning"]._logger = _Stub("logger")
    sys.path.insert(0, REF)
    from lib.datasets.mapfree import MapFreeScene, MapFreeSceneMultiFrame
    from lib.datasets.sampler import RandomConcatSampler
    rng = np.random.default_rng(2024)
    n_frames, n_pairs, T = 12, 150, 3
    idxs = np.stack([rng.integers(0, 2, n_pairs), rng.integers(0, n_frames, n_pairs), rng.integers(0, 2, n_pairs),
                     rng.integers(0, n_frames, n_pairs)], 1).astype(np.uint16)
    overlaps = rng.uniform(0.05, 1.0, n_pairs).astype(np.float32)
    out = dict(idxs=idxs, overlaps=overlaps, n_frames=np.int64(n_frames), T=np.int64(T), limits=np.array([0.4, 0.8]))
    with tempfile.TemporaryDirectory() as td:
        write_tree(Path(td) / "train", n_frames, idxs, overlaps, train=True)
        write_tree(Path(td) / "val", n_frames, idxs, overlaps, train=False)
        sc = MapFreeScene(Path(td) / "train", resize=(270, 360), sample_factor=1, overlap_limits=(0.4, 0.8))
        out["train_single"] = np.asarray(sc.pairs, np.int64)
        out["K_first"] = np.asarray(sc.K["seq0/frame_00000.jpg"])                    # float64 after correct_intrinsic_scale
        assert out["K_first"].dtype == np.float64
        mf = MapFreeSceneMultiFrame(Path(td) / "train", resize=(270, 360), sample_factor=T + 1, overlap_limits=(0.4, 0.8), sample_offset=T)
        out["train_multi_head"] = np.asarray([(a, b, c) for a, b, c, _ in mf.pairs], np.int64).reshape(-1, 3)
        out["train_multi_window"] = np.asarray([w for _, _, _, w in mf.pairs], np.int64).reshape(-1, T)
        vs = MapFreeScene(Path(td) / "val", resize=(540, 720), sample_factor=5)
        out["val_single"] = np.asarray(vs.pairs, np.int64)
        vm = MapFreeSceneMultiFrame(Path(td) / "val", resize=(540, 720), sample_factor=T + 1, sample_offset=T)
        out["val_multi_window"] = np.asarray([w for _, _, _, w in vm.pairs], np.int64).reshape(-1, T)
    # scene-balanced sampler: two epochs in a row from one sampler (the generator runs on), with and without replacement
    sizes = [13, 4, 29]
    ds = torch.utils.data.ConcatDataset([torch.utils.data.TensorDataset(torch.zeros(n)) for n in sizes])
    for tag, repl in (("repl", True), ("norepl", False)):
        s = RandomConcatSampler(ds, 6, repl, shuffle=True)
        out[f"sampler_{tag}_e0"] = np.asarray(list(s), np.int64)
        out[f"sampler_{tag}_e1"] = np.asarray(list(s), np.int64)
    out["sampler_sizes"] = np.asarray(sizes, np.int64)
    np.savez_compressed(os.path.join(OUT, "ref_data_pairs.npz"), **out)
    print({k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
