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


def read_color_image(path, resize):
    """lib/datasets/utils.py:58-74: RGB, resized to (w, h), float /255, [3,h,w].  PIL bilinear stands in
    for cv2.resize (INTER_LINEAR); sub-grey-level differences (cv2 is not installed offline)."""
    from PIL import Image
    im = Image.open(path).convert("RGB").resize((int(resize[0]), int(resize[1])), Image.BILINEAR)
    return torch.from_numpy(np.asarray(im, dtype=np.float32)).permute(2, 0, 1) / 255


class MapFreeScene:
    """val/test reader of one scene directory (lib/datasets/mapfree.py:16-270, the no-overlaps branch):
    intrinsics.txt / poses.txt parsing (:36-75), pairs = keyframe seq0/frame_00000 x every
    `sample_factor`-th seq1 frame (:148-165), sample dict (:211-268) incl. pair_id = index *
    sample_factor (:265, quirk Q4).  Intrinsics are rescaled exactly like correct_intrinsic_scale
    (float64 result, as upstream)."""

    def __init__(self, scene_root, resize, sample_factor=5, estimated_depth=None):
        import re
        self.scene_root, self.resize = str(scene_root), resize
        self.sample_factor, self.estimated_depth = sample_factor, estimated_depth
        self.poses, self.K = {}, {}
        with open(os.path.join(self.scene_root, "poses.txt")) as f:
            for line in f:
                if "#" in line:
                    continue
                parts = line.strip().split(" ")
                qt = np.array(list(map(float, parts[1:])))
                self.poses[parts[0]] = (qt[:4], qt[4:])
        with open(os.path.join(self.scene_root, "intrinsics.txt")) as f:
            for line in f:
                if "#" in line:
                    continue
                parts = line.strip().split(" ")
                fx, fy, cx, cy, W, H = map(float, parts[1:])
                K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32)
                if resize is not None:
                    T = np.eye(3)
                    T[0, 0] = resize[0] / W; T[0, 2] = resize[0] / W / 2 - 0.5
                    T[1, 1] = resize[1] / H; T[1, 2] = resize[1] / H / 2 - 0.5
                    K = T @ K
                self.K[parts[0]] = K
        ids = sorted(int(re.search(r"_(\d+)\..*$", fn).group(1)) for fn in self.poses if "seq0" not in fn)
        self.pairs = [(0, 0, 1, i) for i in ids][0::sample_factor]

    def __len__(self):
        return len(self.pairs)

    def __getitem__(self, index):
        from . import evaluation as E
        sa, ia, sb, ib = self.pairs[index]
        p1, p2 = f"seq{sa}/frame_{ia:05}.jpg", f"seq{sb}/frame_{ib:05}.jpg"
        img1 = read_color_image(os.path.join(self.scene_root, p1), self.resize)
        img2 = read_color_image(os.path.join(self.scene_root, p2), self.resize)
        if self.estimated_depth is not None:
            d1 = read_depth_image(os.path.join(self.scene_root, p1).replace(".jpg", f".{self.estimated_depth}.png"))
            d2 = read_depth_image(os.path.join(self.scene_root, p2).replace(".jpg", f".{self.estimated_depth}.png"))
        else:
            d1 = d2 = torch.tensor([])
        (q1, t1), (q2, t2) = self.poses[p1], self.poses[p2]
        q12 = E.qmult(q2, E.qinverse(q1))
        t12 = t2 - E.rotate_vector(t1, q12)
        T = np.eye(4, dtype=np.float32); T[:3, :3] = E.quat2mat(q12); T[:3, -1] = t12
        return {"image0": img1, "depth0": d1, "image1": img2, "depth1": d2, "T_0to1": torch.from_numpy(T),
                "K_color0": torch.from_numpy(self.K[p1].copy()), "K_color1": torch.from_numpy(self.K[p2].copy()),
                "dataset_name": "Mapfree", "scene_id": os.path.basename(self.scene_root.rstrip("/")),
                "scene_root": self.scene_root, "pair_id": index * self.sample_factor, "pair_names": (p1, p2)}


def make_loader(cfg, split="val"):
    """batch-1 iterator over the split (submission.py:76-79); real data when DATA_ROOT/<split> exists,
    otherwise the synthetic stand-in"""
    root = cfg.DATASET.DATA_ROOT
    if root and os.path.isdir(os.path.join(str(root), split)):
        resize = (cfg.DATASET.WIDTH, cfg.DATASET.HEIGHT)
        scenes = sorted(d for d in os.listdir(os.path.join(str(root), split)) if os.path.isdir(os.path.join(str(root), split, d)))
        if cfg.DATASET.SCENES:
            scenes = [s for s in scenes if s in cfg.DATASET.SCENES]

        def gen():
            for s in scenes:
                sc = MapFreeScene(os.path.join(str(root), split, s), resize, 5, cfg.DATASET.ESTIMATED_DEPTH)
                for i in range(len(sc)):
                    yield collate_batch1(sc[i])
        return gen()
    ds = SyntheticMapFree(H=cfg.DATASET.HEIGHT or 720, W=cfg.DATASET.WIDTH or 540)
    return (collate_batch1(ds[i]) for i in range(len(ds)))
