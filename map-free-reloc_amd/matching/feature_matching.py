"""Correspondence plugins (`cfg.FEATURE_MATCHING`): objects with
`get_correspondences(data) -> (pts1, pts2)`, each ndarray [N,2] float32 in the dataset-resized
pixel frame, N may be 0 (np.array([])).  Mirrors lib/models/matching/feature_matching.py.

  PrecomputedMatching  npz lookup by pair_id (feature_matching.py:5-50) -- host IO glue, same
                       semantics incl. lazy per-scene reload, float32 cast and NaN stripping
  SuperGlueMatching    NEW: online SuperPoint+SuperGlue on the GPU (the reference only has it
                       offline, etc/feature_matching_baselines/matchers.py:62-120)
  LoFTRMatching        NEW: online LoFTR on the GPU (reference: offline only, matchers.py:12-59)
  SIFTMatching         feature_matching.py:53-118: detectAndCompute from OpenCV (or a caller-supplied
                       detector; raises ImportError when neither exists), rootSIFT + exact 2-NN + ratio
                       test on the GPU (csrc/descriptor_match.hip) instead of FLANN
"""
import numpy as np

from .. import wire


class PrecomputedMatching:
    def __init__(self, cfg):
        self.correspondences = None
        self.debug = cfg.DEBUG
        if '{' in cfg.MATCHES_FILE_PATH:
            self.matches_file_path = cfg.MATCHES_FILE_PATH
            self.scene_id = None
            self.pairs_txt = cfg.DATASET.PAIRS_TXT.TEST
        else:
            self.load_correspondences(cfg.MATCHES_FILE_PATH)

    def load_correspondences(self, file_path):
        self.correspondences = wire.load_correspondences(file_path)

    def get_correspondences(self, data):
        if hasattr(self, 'scene_id'):
            if self.scene_id != data['scene_id'][0]:
                self.scene_id = data['scene_id'][0]
                scene_root = data['scene_root'][0]
                self.load_correspondences(self.matches_file_path.format(scene_root=scene_root, pairs_txt=self.pairs_txt))
        pair_id = int(data['pair_id'].item()) if hasattr(data['pair_id'], 'item') else int(data['pair_id'])
        return wire.strip_nan(self.correspondences[pair_id])


from ..datasets import to_gray as _to_gray, gray_plane as _gray_plane   # the matcher's gray plane: byte-rounded BT.601 luma / 255 (datasets.gray_plane)


class _GrayPairStage:
    """data['image0'], data['image1'] ([1,3,H,W] or [1,1,H,W] in [0,1]) -> ONE host array [2,1,H,W] f32 of the matcher's gray plane (datasets.to_gray), written with
    numpy on the calling thread into a persistent buffer.  No torch CPU kernel touches the images: on the GPU boxes' 256-core hosts
    a torch elementwise op / stack over a 1.5 MB image fans out over every core and costs ~20 ms in thread wake-ups per call
    (tools/diag_plugin_prof2.py), more than the whole matcher on the GPU."""

    def __init__(self):
        self.buf = self.np = None

    def __call__(self, data):
        import torch
        ims = [data['image0'][0], data['image1'][0]]
        if any(isinstance(im, torch.Tensor) and im.is_cuda for im in ims):      # device-resident inputs: torch ops on the device
            return torch.stack([_to_gray(im) for im in ims])[:, None].to(torch.float32)
        H, W = ims[0].shape[-2:]
        if self.np is None or self.np.shape[-2:] != (H, W):
            self.np = np.empty((2, 1, H, W), np.float32)
            self.buf = torch.from_numpy(self.np)
        for k, im in enumerate(ims):
            a = im.numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
            g = self.np[k, 0]
            if a.shape[0] == 1:
                g[...] = a[0]
            else:                                               # datasets.to_gray's values, written in place
                u8 = np.rint(a.astype(np.float32, copy=False) * np.float32(255)).astype(np.uint8)
                _gray_plane(np.moveaxis(u8, 0, -1), None, g)
        return self.buf


class SuperGlueMatching:
    """online matcher: data['image0'], data['image1'] ([1,3,H,W] in [0,1]) -> correspondences"""

    def __init__(self, cfg):
        import torch
        from ..nets import weights as WT
        from ..nets.superpoint import SuperPointHIP
        from ..nets.superglue import SuperGlueHIP
        sg = cfg.SUPERGLUE
        sp_sd = WT.load_checkpoint(sg.SUPERPOINT_WEIGHTS) if sg.SUPERPOINT_WEIGHTS else \
            WT.synthetic_or_raise("SuperPoint", cfg, lambda: WT.superpoint_state_dict(sg.SYNTHETIC_SEED))
        sg_sd = WT.load_checkpoint(sg.SUPERGLUE_WEIGHTS) if sg.SUPERGLUE_WEIGHTS else WT.synthetic_or_raise("SuperGlue", cfg, WT.superglue_state_dict)
        self.device = torch.device("cuda")
        self.sp = SuperPointHIP(sp_sd, self.device, sg.NMS_RADIUS, sg.KEYPOINT_THRESHOLD, sg.MAX_KEYPOINTS)
        self.sg = SuperGlueHIP(sg_sd, self.device, sg.SINKHORN_ITERATIONS, sg.MATCH_THRESHOLD)
        self.use_graph = bool(cfg.HIP.GRAPH_BATCH1)
        self._graphs = {}
        self._gray = _GrayPairStage()
        from ..pipeline import RangeGuard
        self.guard = RangeGuard(self.device, lambda: SuperGlueMatching(cfg))       # f16x2 range guard: flag travels with the result, exact twin on demand

    def _forward(self, ims):
        with self.guard:
            out = self.sg(self.sp(ims), tuple(ims.shape[-2:]))
        # one packed result so that the host needs a single D2H copy: [n | range flag | pts0 (K x 2) | pts1 (K x 2)]
        import torch
        return torch.cat([out["n_corr"].to(torch.float32), self.guard.flag.to(torch.float32), out["pts0"].reshape(-1), out["pts1"].reshape(-1)])

    def get_correspondences(self, data):
        import torch
        ims = self._gray(data)
        flat = None
        if self.use_graph:                                   # batch 1 is launch-bound: replay the whole forward from one HIP graph
            from ..nets.graph import GraphCaptureError, GraphedCall
            key = tuple(ims.shape)
            try:
                if key not in self._graphs:
                    self._graphs[key] = GraphedCall(self._forward, [ims.to(self.device)])
                flat = self._graphs[key](ims).cpu().numpy()
            except GraphCaptureError as e:
                import warnings
                warnings.warn(f"SuperGlueMatching: {e}; running eagerly from now on")
                self.use_graph = False
        if flat is None:
            flat = self._forward(ims.to(self.device).contiguous()).cpu().numpy()
        if flat[1] != 0:                                     # an activation left the f16x2 range: the same pair through the exact (bf16x3) twin
            from .. import options
            self.guard.reruns += 1
            with options.override(SPLIT="bf16x3"):
                return self.guard.twin().get_correspondences(data)
        n = int(flat[0])
        if n == 0:
            e = np.array([])
            return e, e
        K = (len(flat) - 2) // 4
        return flat[2:2 + 2 * K].reshape(K, 2)[:n].copy(), flat[2 + 2 * K:].reshape(K, 2)[:n].copy()


class LoFTRMatching:
    """online matcher: LoFTR coarse-to-fine on the GPU (the reference only has it offline,
    etc/feature_matching_baselines/matchers.py:12-59), incl. the right-padding to a multiple of 8 (quirk Q3)"""

    def __init__(self, cfg):
        import torch
        from ..nets import weights as WT
        from ..nets.loftr import LoFTRHIP
        lw = cfg.LOFTR.WEIGHTS
        sd = WT.strip_prefix(WT.load_checkpoint(lw), "matcher.") if lw else WT.synthetic_or_raise("LoFTR", cfg, WT.loftr_state_dict)
        self.device = torch.device("cuda")
        self.net = LoFTRHIP(sd, self.device)
        self._gray = _GrayPairStage()
        self.use_graph = bool(cfg.HIP.GRAPH_BATCH1)
        self._graphs = {}
        from ..pipeline import RangeGuard
        self.guard = RangeGuard(self.device, lambda: LoFTRMatching(cfg))

    def _coarse(self, ims):
        import torch
        H, W = ims.shape[-2:]
        with torch.no_grad(), self.guard:
            return self.net.coarse_stage(torch.nn.functional.pad(ims, (0, (-W) % 8, 0, (-H) % 8)).contiguous())

    def get_correspondences(self, data):
        import torch
        ims = self._gray(data)
        c = None
        if self.use_graph:          # everything before the match count is known: ~350 launches replayed as one HIP graph
            from ..nets.graph import GraphCaptureError, GraphedCall
            key = tuple(ims.shape)
            try:
                if key not in self._graphs:
                    self._graphs[key] = GraphedCall(self._coarse, [ims.to(self.device)])
                c = self._graphs[key](ims)
            except GraphCaptureError as e:
                import warnings
                warnings.warn(f"LoFTRMatching: {e}; running eagerly from now on")
                self.use_graph = False
        if c is None:
            c = self._coarse(ims.to(self.device))
        with torch.no_grad():
            with self.guard.keep():
                out = self.net.fine_stage(c)                   # reads the graph's static buffers before the next replay
            n_t = out["n_corr"][:1].to(torch.float32)
            flat = torch.cat([n_t, self.guard.flag.to(torch.float32), out["pts0"][0].reshape(-1), out["pts1"][0].reshape(-1)]).cpu().numpy()      # one D2H copy
        if flat[1] != 0:                                       # f16x2 range guard fired: the exact (bf16x3) twin
            from .. import options
            self.guard.reruns += 1
            with options.override(SPLIT="bf16x3"):
                return self.guard.twin().get_correspondences(data)
        n = int(flat[0])
        if n == 0:
            e = np.array([])
            return e, e
        L = (len(flat) - 2) // 4
        return flat[2:2 + 2 * L].reshape(L, 2)[:n].copy(), flat[2 + 2 * L:].reshape(L, 2)[:n].copy()


def _cv_sift_detector(num_features):
    """detectAndCompute of OpenCV SIFT (feature_matching.py:58,82-83) -- keypoint detection and
    description are NOT part of the accelerated path; cv2 is used when importable"""
    try:
        import cv2 as cv
    except ImportError as e:
        raise ImportError(
            "SIFT detection needs OpenCV (cv.SIFT_create; feature_matching.py:58), which is not installed; pass "
            "detector=callable(gray_u8 [H,W]) -> (kpts [n,2] f32, desc [n,128] f32) to use another "
            "SIFT implementation -- everything after detectAndCompute runs on the GPU") from e
    sift = cv.SIFT_create(num_features)

    def detect(gray):
        kp, des = sift.detectAndCompute(gray, None)
        if des is None:
            return np.zeros((0, 2), np.float32), np.zeros((0, 128), np.float32)
        return np.float32([k.pt for k in kp]).reshape(-1, 2), np.float32(des)
    return detect


class SIFTMatching:
    """feature_matching.py:53-118.  detectAndCompute stays with the caller-supplied / OpenCV detector
    (CPU); rootSIFT + 2-NN + ratio test run in csrc/descriptor_match.hip (exact 2-NN instead of FLANN's
    approximate kd-forest)."""

    def __init__(self, cfg, detector=None):
        from ..descriptor_ops import DescriptorRatioMatcher
        self.ratio_threshold = cfg.SIFT.RATIO_THRESHOLD
        self.detector = detector if detector is not None else _cv_sift_detector(cfg.SIFT.NUM_FEATURES)
        self.matcher = DescriptorRatioMatcher(self.ratio_threshold)
        self.debug = cfg.DEBUG

    @staticmethod
    def transform_grayscale(img):
        """[3,H,W] float in [0,1] -> u8 gray: (255*img).astype(uint8) then COLOR_RGB2GRAY, which OpenCV
        evaluates in 14-bit fixed point: (4899 R + 9617 G + 1868 B + 8192) >> 14 (feature_matching.py:61-65)"""
        a = np.asarray(img.permute(1, 2, 0).cpu().numpy() if hasattr(img, "permute") else img)
        a = (255 * a).astype(np.uint8).astype(np.int32)
        return ((4899 * a[..., 0] + 9617 * a[..., 1] + 1868 * a[..., 2] + 8192) >> 14).astype(np.uint8)

    def get_correspondences(self, data):
        img0 = self.transform_grayscale(data['image0'].squeeze(0))
        img1 = self.transform_grayscale(data['image1'].squeeze(0))
        out = self.matcher([self.detector(img0)], [self.detector(img1)])
        n = int(out["n_corr"][0])
        pts1 = out["pts0"][0, :n].cpu().numpy().reshape(-1, 2)
        pts2 = out["pts1"][0, :n].cpu().numpy().reshape(-1, 2)
        return pts1, pts2
