"""Offline matcher plugins with the reference's API (etc/feature_matching_baselines/matchers.py):
class M(resize: (W, H), outdoor: bool) with match(pair_path: (str, str)) -> ndarray [N,4]
(x0,y0,x1,y1) in the resized pixel frame, or np.full((1,4), nan) when there is no correspondence
(:59, :120); registered in MATCHERS and driven by compute.py (-ds Mapfree -m SG|LoFTR).

Image reading: the reference uses SuperGlue's read_image (cv2.imread GRAYSCALE, cv2.resize of the float
image to (W,H), /255).  cv2 is not available offline: the file is decoded with PIL and the
gray plane is datasets.gray_plane: ITU-R 601-2 luma rounded to a byte, cv2-style half-pixel bilinear
resize of the FLOAT gray image when the size changes, / 255.  At the files' own size (Map-free: 540 x 720 =
config/mapfree.yaml) this is the ONE plane every route of the package feeds the matcher (offline matchers,
batched loaders, online plugin: tests/test_gpu_routes_agree.py).  With a real resize the reference itself has
two orders -- this offline reader resizes the gray image, its dataset class resizes the 8-bit RGB image
(lib/datasets/utils.py:58-74) -- and so has this package: offline = here, every loader / plugin route = the
dataset's order (datasets.MapFreeScene.resize_is_native).  (Unpinned against OpenCV: libjpeg's
own grayscale output can differ from the luma of its RGB output by a grey level; tests/external/gen_cv_golden.py
dumps cv2's planes for three JPEGs when run off-box.)

Weights: upstream checkpoints when present (same file names as matchers.py:17,70), else the seeded
synthetic weights of nets/weights.py with a warning.
"""
import os
import warnings

import numpy as np
import torch

from .nets import weights as WT


def read_image(path, resize):
    from .datasets import read_gray_plane
    return read_gray_plane(path, resize)


def _weights(path, synth, what):
    if path and os.path.exists(path):
        return WT.load_checkpoint(path)
    warnings.warn(f"{what}: checkpoint {path!r} not found -> seeded synthetic weights (results are NOT meaningful)")
    return synth()


class SuperGlue_matcher:
    def __init__(self, resize, outdoor=False, weights_dir="SuperGlue/models/weights"):
        from .nets.superpoint import SuperPointHIP
        from .nets.superglue import SuperGlueHIP
        self.resize = resize
        self.device = torch.device("cuda")
        sp = _weights(os.path.join(weights_dir, "superpoint_v1.pth"), WT.superpoint_state_dict, "SuperPoint")
        sg = _weights(os.path.join(weights_dir, f"superglue_{'outdoor' if outdoor else 'indoor'}.pth"),
                      WT.superglue_state_dict, "SuperGlue")
        self.sp = SuperPointHIP(sp, self.device, nms_radius=4, keypoint_threshold=0.005, max_keypoints=1024)   # :65-67
        self.sg = SuperGlueHIP(sg, self.device, sinkhorn_iterations=20, match_threshold=0.2)                   # :70-71

    def match_tensors(self, im0, im1):
        ims = torch.from_numpy(np.stack([im0, im1]))[:, None].to(self.device)
        out = self.sg(self.sp(ims), tuple(ims.shape[-2:]))
        n = int(out["n_corr"][0])
        if n > 0:
            return torch.cat([out["pts0"][0, :n], out["pts1"][0, :n]], 1).cpu().numpy()
        print("no correspondences")
        return np.full((1, 4), np.nan)

    def match(self, pair_path):
        '''return correspondences between images (w/ path pair_path)'''
        return self.match_tensors(read_image(pair_path[0], self.resize), read_image(pair_path[1], self.resize))


class LoFTR_matcher:
    def __init__(self, resize, outdoor=False, weights_dir="LoFTR/weights"):
        from .pipeline import LoFTREmatPipeline
        self.resize = resize
        sd = _weights(os.path.join(weights_dir, "outdoor_ot.ckpt" if outdoor else "indoor_ot.ckpt"), WT.loftr_state_dict, "LoFTR")
        sd = {k[len("matcher."):] if k.startswith("matcher.") else k: v for k, v in sd.items()}
        self._pipe = LoFTREmatPipeline("cuda", loftr_state=sd)

    def match_tensors(self, im0, im1):
        ims = torch.from_numpy(np.stack([im0, im1]))[:, None].to(self._pipe.device)
        out = self._pipe.match(ims)
        n = int(out["n_corr"][0])
        if n > 0:
            return torch.cat([out["pts0"][0, :n], out["pts1"][0, :n]], 1).cpu().numpy()
        print("no correspondences")
        return np.full((1, 4), np.nan)

    def match(self, pair_path):
        return self.match_tensors(read_image(pair_path[0], self.resize), read_image(pair_path[1], self.resize))


class SIFT_matcher:
    """matchers.py:123-188: 2048 SIFT features, rootSIFT, 2-NN, ratio 0.8.  Detection is OpenCV's (or the
    `detector` callable's); the descriptor leg runs in csrc/descriptor_match.hip."""

    def __init__(self, resize, outdoor=False, detector=None):
        from .descriptor_ops import DescriptorRatioMatcher
        from .matching.feature_matching import _cv_sift_detector
        self.resize = resize
        self.detector = detector if detector is not None else _cv_sift_detector(2048)     # :146-147
        self.matcher = DescriptorRatioMatcher(0.8)                                          # :145

    def match_arrays(self, img0, img1):
        out = self.matcher([self.detector(img0)], [self.detector(img1)])
        n = int(out["n_corr"][0])
        if n > 0:
            return torch.cat([out["pts0"][0, :n], out["pts1"][0, :n]], 1).cpu().numpy()
        print("no correspondences")
        return np.full((1, 4), np.nan)

    def match(self, pair_path):
        g = [np.round(read_image(p, self.resize) * 255.0).astype(np.uint8) for p in pair_path]
        return self.match_arrays(g[0], g[1])


MATCHERS = {'LoFTR': LoFTR_matcher, 'SG': SuperGlue_matcher, 'SIFT': SIFT_matcher}
