"""Input contract of the hot path: the per-pair `data` dict of MapFreeScene.__getitem__
(lib/datasets/mapfree.py:250-268) -- image0/1 f32 [1,3,H,W] in [0,1], depth0/1 f32
[1,H,W] metres (uint16 PNG / 1000, lib/datasets/utils.py:77-81), K_color0/1 [1,3,3] rescaled
like correct_intrinsic_scale (utils.py:117-130: FLOAT64 whenever the dataset resizes, i.e. always on Map-free), T_0to1, pair_id (= index * 5, mapfree.py:265),
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
    return K                                               # float64, as upstream (the float64 eye(3) promotes the product)


def read_depth_image(path):
    """utils.py:77-81: uint16 millimetres -> float32 metres"""
    from PIL import Image
    d = np.asarray(Image.open(path), dtype=np.uint16)
    return torch.from_numpy((d / 1000).astype(np.float32))


class SyntheticScene:
    """one synthetic scene: `frames` query frames against the keyframe, the reference's sample schema"""

    def __init__(self, scene_index, frames=4, H=720, W=540, sample_factor=5, seed=0):
        self.s, self.frames, self.H, self.W, self.sample_factor, self.seed = scene_index, frames, H, W, sample_factor, seed
        self.scene_id = f"s{scene_index:05d}"
        self.scene_root = f"/synthetic/{self.scene_id}"

    def __len__(self):
        return self.frames

    def pair_name(self, f):
        return f"seq1/frame_{f * self.sample_factor:05d}.jpg"

    def __getitem__(self, f):
        p = IM.synthetic_pair(self.seed + 1000 * self.s + f, self.H, self.W)
        rgb = lambda im: torch.from_numpy(im)[None].expand(3, -1, -1).contiguous()
        T = np.eye(4, dtype=np.float32); T[:3, :3] = p["R_gt"]; T[:3, 3] = p["t_gt"]
        return {
            "image0": rgb(p["img0"]), "image1": rgb(p["img1"]),
            "depth0": torch.from_numpy(p["depth0"]), "depth1": torch.from_numpy(p["depth1"]),
            # float64, like every sample of the Map-free loader it stands in for (correct_intrinsic_scale, utils.py:117-130)
            "K_color0": torch.from_numpy(p["K"].copy()), "K_color1": torch.from_numpy(p["K"].copy()),
            "T_0to1": torch.from_numpy(T), "pair_id": f * self.sample_factor,
            "scene_id": self.scene_id, "scene_root": self.scene_root,
            "pair_names": ("seq0/frame_00000.jpg", f"seq1/frame_{f * self.sample_factor:05d}.jpg"),
        }


class SyntheticMapFree(torch.utils.data.Dataset if hasattr(torch.utils, "data") else object):
    """n_scenes x frames_per_scene synthetic pairs with the reference's sample schema"""

    def __init__(self, n_scenes=2, frames_per_scene=4, H=720, W=540, sample_factor=5, seed=0):
        self.scenes = [SyntheticScene(s, frames_per_scene, H, W, sample_factor, seed) for s in range(n_scenes)]
        self.items = [(s, f) for s in range(n_scenes) for f in range(frames_per_scene)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        s, f = self.items[i]
        return self.scenes[s][f]


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


def resize_bilinear_u8(a, wh):
    """cv2.resize(img, (w, h)) with its default INTER_LINEAR on a uint8 [H,W,C] image: half-pixel-centre bilinear sampling WITHOUT
    antialiasing (source coordinate (x + 0.5) * scale - 0.5, edge-clamped), result rounded back to uint8.  At an exact 2x downscale
    (720x540 -> 360x270, the regression configs) every sample falls midway between two pixels: a 2x2 box average.  PIL's
    Image.BILINEAR is NOT this: it widens the triangle filter when shrinking ([1/8, 3/8, 3/8, 1/8] at 2x) and blurs the input.
    (OpenCV evaluates the same interpolation in 11-bit fixed point: the two can differ by one grey level on a rounding tie; cv2 is not
    installed offline, so that last bit is unpinned.)"""
    H, W = a.shape[:2]
    w, h = int(wh[0]), int(wh[1])
    if (w, h) == (W, H):
        return a

    def taps(n_out, n_in):
        s = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(s).astype(np.int64)
        f = (s - i0).astype(np.float32)
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), f
    y0, y1, fy = taps(h, H)
    x0, x1, fx = taps(w, W)
    af = a.astype(np.float32)
    top = af[y0][:, x0] * (1 - fx)[None, :, None] + af[y0][:, x1] * fx[None, :, None]
    bot = af[y1][:, x0] * (1 - fx)[None, :, None] + af[y1][:, x1] * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def resize_bilinear_f32(a, wh):
    """cv2.resize(img.astype('float32'), (w, h)) (INTER_LINEAR) of a single-channel float32 [H,W] image -- SuperGlue's read_image resizes the
    float gray image, so nothing is rounded back to uint8: the same half-pixel-centre taps as resize_bilinear_u8, float32 arithmetic"""
    H, W = a.shape[:2]
    w, h = int(wh[0]), int(wh[1])
    if (w, h) == (W, H):
        return a

    def taps(n_out, n_in):
        s = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(s).astype(np.int64)
        f = (s - i0).astype(np.float32)
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), f
    y0, y1, fy = taps(h, H)
    x0, x1, fx = taps(w, W)
    af = a.astype(np.float32, copy=False)
    top = af[y0][:, x0] * (1 - fx)[None, :] + af[y0][:, x1] * fx[None, :]
    bot = af[y1][:, x0] * (1 - fx)[None, :] + af[y1][:, x1] * fx[None, :]
    return (top * (1 - fy)[:, None] + bot * fy[:, None]).astype(np.float32)


def luma_u8(rgb):
    """[..., 3] uint8 RGB -> uint8 gray, the value an 8-bit grayscale read of the file yields (matchers.py:101-104 -> SuperGlue's read_image:
    cv2.imread(path, GRAYSCALE)): ITU-R 601-2 luma ROUNDED to a byte.  PIL's "L" conversion, restated so that every route shares it:
    (19595 R + 38470 G + 7471 B + 2^15) >> 16."""
    u = rgb.astype(np.uint32)
    return ((u[..., 0] * 19595 + u[..., 1] * 38470 + u[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def gray_plane(rgb_u8, resize=None, out=None):
    """THE matcher input of every route (offline matchers.read_image, the batched loaders, the per-pair online plugin): uint8 luma of the
    decoded RGB bytes, resized as a float image when a size is asked for (SuperGlue's read_image order: gray, then resize), / 255 in float32
    -> [h, w] float32 in [0, 1], written into `out` when given"""
    g8 = luma_u8(rgb_u8)
    if resize is not None and (int(resize[0]), int(resize[1])) != (g8.shape[1], g8.shape[0]):
        g = resize_bilinear_f32(g8.astype(np.float32), resize)
        g /= np.float32(255)
    else:
        g = _luts()[0][g8]
    if out is None:
        return g
    out[...] = g
    return out


def read_color_image(path, resize):
    """lib/datasets/utils.py:58-74: RGB, cv2.resize to (w, h) (INTER_LINEAR: resize_bilinear_u8 above), float /255, [3,h,w]."""
    from PIL import Image
    im = resize_bilinear_u8(np.asarray(Image.open(path).convert("RGB")), resize) if resize is not None else np.asarray(Image.open(path).convert("RGB"))
    # the /255 in numpy on the calling thread (same IEEE fp32 quotient as torch's): a torch CPU op over a 1.2 M-element image fans out over
    # every host core (256 on the GPU boxes) and costs ~20 ms in thread wake-ups, several times the JPEG decode
    a = np.ascontiguousarray(np.asarray(im, dtype=np.float32).transpose(2, 0, 1))
    a /= np.float32(255)
    return torch.from_numpy(a)


_GRAY_LUT = None
_DEPTH_LUT = None
_HOST_LIB = False           # False = not looked for yet, None = absent (numpy expressions), else the ctypes handle of csrc/libmfr_host.so


def _host_lib():
    """csrc/libmfr_host.so (host_decode.c: the loaders' two per-pixel loops in C), or None -- then the numpy expressions run"""
    global _HOST_LIB
    if _HOST_LIB is False:
        import ctypes
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmfr_host.so")
        try:
            lib = ctypes.CDLL(path)
            vp, sz = ctypes.c_void_p, ctypes.c_size_t
            lib.mfr_host_gray_from_rgb.argtypes = [vp, sz, vp, vp]; lib.mfr_host_gray_from_rgb.restype = None
            lib.mfr_host_depth_from_u16.argtypes = [vp, sz, vp, vp]; lib.mfr_host_depth_from_u16.restype = None
            _HOST_LIB = lib if lib.mfr_host_abi_version() == 2 else None
        except (OSError, AttributeError):
            _HOST_LIB = None
    return _HOST_LIB


def _luts():
    global _GRAY_LUT, _DEPTH_LUT
    if _GRAY_LUT is None:
        _GRAY_LUT = np.ascontiguousarray(np.arange(256, dtype=np.float32) / np.float32(255))           # gray byte / 255 (read_image's float32 quotients)
        _DEPTH_LUT = np.ascontiguousarray((np.arange(65536, dtype=np.float64) / 1000).astype(np.float32))   # read_depth_image's values
    return _GRAY_LUT, _DEPTH_LUT


def read_gray_plane(path, resize, out=None):
    """gray_plane() of a file -- what matchers.read_image(path, resize) returns, bit for bit -- WITHOUT any float RGB image, written into
    `out` ([h, w] float32, C-contiguous) when given.  At the file's own size (Map-free: 540 x 720, config/mapfree.yaml) a pixel is one
    integer luma + one 256-entry table look-up of byte / 255 (csrc/host_decode.c; the numpy expression when that library is not built)."""
    from PIL import Image
    pim = Image.open(path)
    if pim.mode != "RGB":
        pim = pim.convert("RGB")                                # (a JPEG opens as RGB: convert() would only copy it)
    im = np.asarray(pim)
    h, w = im.shape[:2]
    native = resize is None or (int(resize[0]), int(resize[1])) == (w, h)
    lib = _host_lib()
    if lib is None or not native:
        return gray_plane(im, resize, out)
    lut, _ = _luts()
    im = np.ascontiguousarray(im)
    if out is None:
        out = np.empty((h, w), dtype=np.float32)
    assert out.dtype == np.float32 and out.shape == (h, w) and out.flags["C_CONTIGUOUS"]
    lib.mfr_host_gray_from_rgb(im.ctypes.data, h * w, lut.ctypes.data, out.ctypes.data)
    return out


def read_depth_plane(path, out=None):
    """read_depth_image as a numpy array (into `out` when given): float32(v / 1000.0), tabulated once for the 65 536 possible values"""
    from PIL import Image
    d = np.ascontiguousarray(np.asarray(Image.open(path), dtype=np.uint16))
    lib = _host_lib()
    if lib is None:
        g = (d / 1000).astype(np.float32)
        if out is None:
            return g
        out[...] = g
        return out
    _, lut = _luts()
    if out is None:
        out = np.empty(d.shape, dtype=np.float32)
    assert out.dtype == np.float32 and out.shape == d.shape and out.flags["C_CONTIGUOUS"]
    lib.mfr_host_depth_from_u16(d.ctypes.data, d.size, lut.ctypes.data, out.ctypes.data)
    return out


import collections as _collections
import threading as _threading

_FRAME_CACHE = _collections.OrderedDict()        # (scene_root, frame, resize, depth kind, bw) -> (image, depth): the keyframes of the scenes in flight
_FRAME_LOCK = _threading.Lock()


def clear_frame_cache():
    """drop the cached keyframes (tools / tests that rewrite a scene directory in place)"""
    with _FRAME_LOCK:
        _FRAME_CACHE.clear()


class MapFreeScene:
    """reader of one scene directory (lib/datasets/mapfree.py:16-270, single-frame queries): intrinsics.txt / poses.txt
    parsing (:36-75); val/test scenes: pairs = keyframe seq0/frame_00000 x every `sample_factor`-th seq1 frame (:148-165);
    TRAINING scenes (an `overlaps.npz` is present, :85-112): pairs = the pre-computed (seqA, imA, seqB, imB) rows whose
    overlap score lies strictly inside `overlap_limits`; sample dict (:211-268) incl. pair_id = index * sample_factor
    (:265, quirk Q4).  Intrinsics are rescaled exactly like correct_intrinsic_scale (float64 result, as upstream).
    `black_white`: the reference's Grayscale(num_output_channels=3) training transform (datamodules.py:38-40)."""

    def __init__(self, scene_root, resize, sample_factor=5, estimated_depth=None, overlap_limits=None, black_white=False):
        import re
        self.scene_root, self.resize = str(scene_root), resize
        self.scene_id = os.path.basename(self.scene_root.rstrip("/"))
        self.sample_factor, self.estimated_depth = sample_factor, estimated_depth
        self.poses, self.K = {}, {}
        self._file_sizes = set()                                # (W, H) of the frames as intrinsics.txt lists them
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
                self._file_sizes.add((int(W), int(H)))
                K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32)
                if resize is not None:
                    T = np.eye(3)
                    T[0, 0] = resize[0] / W; T[0, 2] = resize[0] / W / 2 - 0.5
                    T[1, 1] = resize[1] / H; T[1, 2] = resize[1] / H / 2 - 0.5
                    K = T @ K
                self.K[parts[0]] = K
        self.black_white = bool(black_white)
        ov = os.path.join(self.scene_root, "overlaps.npz")
        if os.path.exists(ov):                                  # training scene
            f = np.load(ov, allow_pickle=False)
            idxs, overlaps = np.asarray(f["idxs"]), np.asarray(f["overlaps"])
            if overlap_limits is not None and overlap_limits[0] is not None:
                lo, hi = overlap_limits
                idxs = idxs[np.logical_and(lo < overlaps, overlaps < hi)]
            if sample_factor != 1:
                raise ValueError("training scenes (overlaps.npz) take sample_factor 1 (mapfree.py:112)")
            self.pairs = [tuple(int(v) for v in row) for row in idxs]
        else:
            ids = sorted(int(re.search(r"_(\d+)\..*$", fn).group(1)) for fn in self.poses if "seq0" not in fn)
            self.pairs = [(0, 0, 1, i) for i in ids][0::sample_factor]
        # every pair of a val / test scene has the SAME reference view (the keyframe): consumers may compute its features once
        self.shared_reference = len({(sa, ia) for sa, ia, _, _ in self.pairs}) == 1 and len(self.pairs) > 0

    def __len__(self):
        return len(self.pairs)

    def pair_name(self, index):
        sa, ia, sb, ib = self.pairs[index]
        return f"seq{sb}/frame_{ib:05}.jpg"

    def _frame(self, rel, keep=False):
        """decoded image [3,h,w] f32 + depth map [h,w] f32 (or an empty tensor) of one frame.  keep=True goes through a small
        process-wide cache: a val / test scene pairs its one keyframe with every query (mapfree.py:148-165), the reference re-reads and
        re-decodes it 116 times per scene; here it is decoded once per scene (read-only tensors, shared by the samples; 4 entries,
        least recently used first out, so a run over 130 scenes holds a handful of frames, not 130)."""
        path = os.path.join(self.scene_root, rel)
        try:
            st = os.stat(path)
            ident = (st.st_mtime_ns, st.st_size)                # a scene regenerated at the same path is a different frame
        except OSError:
            ident = None
        key = (self.scene_root, rel, ident, tuple(self.resize) if self.resize is not None else None, self.estimated_depth, self.black_white)
        if keep:
            with _FRAME_LOCK:
                hit = _FRAME_CACHE.get(key)
                if hit is not None:
                    _FRAME_CACHE.move_to_end(key)
                    return hit
        img = read_color_image(os.path.join(self.scene_root, rel), self.resize)
        if self.black_white:                                    # torchvision Grayscale: ITU-R 601-2 luma, replicated
            img = _luma3(img)
        if self.estimated_depth is not None:
            d = read_depth_image(os.path.join(self.scene_root, rel).replace(".jpg", f".{self.estimated_depth}.png"))
        else:
            d = torch.tensor([])
        if keep:
            with _FRAME_LOCK:
                hit = _FRAME_CACHE.get(key)                     # another thread may have decoded it meanwhile: hand out ONE object
                if hit is not None:
                    return hit
                _FRAME_CACHE[key] = (img, d)                    # shared by every sample of the scene: consumers must not write into them
                while len(_FRAME_CACHE) > 4:
                    _FRAME_CACHE.popitem(last=False)
        return img, d

    def _gray_frame(self, rel, keep=False, out_g=None, out_d=None):
        """(gray plane [h,w] f32, depth [h,w] f32 or None) of one frame as numpy arrays -- to_gray(image) / depth of _frame(rel), bit for bit at
        the file's own size (gray_pair, the only caller, refuses other sizes), without the float RGB image, decoded straight into out_g / out_d when given; keep=True: the same small least-recently-used cache as
        _frame under its own keys (the cached arrays are copied into out_g / out_d)"""
        path = os.path.join(self.scene_root, rel)
        dpath = path.replace(".jpg", f".{self.estimated_depth}.png") if self.estimated_depth is not None else None
        if not keep:
            return read_gray_plane(path, self.resize, out_g), (read_depth_plane(dpath, out_d) if dpath else None)
        try:
            st = os.stat(path)
            ident = (st.st_mtime_ns, st.st_size)
        except OSError:
            ident = None
        key = ("gray", self.scene_root, rel, ident, tuple(self.resize) if self.resize is not None else None, self.estimated_depth)
        with _FRAME_LOCK:
            hit = _FRAME_CACHE.get(key)
            if hit is not None:
                _FRAME_CACHE.move_to_end(key)
        if hit is None:
            hit = (read_gray_plane(path, self.resize), read_depth_plane(dpath) if dpath else None)
            with _FRAME_LOCK:
                _FRAME_CACHE[key] = hit
                while len(_FRAME_CACHE) > 4:
                    _FRAME_CACHE.popitem(last=False)
        g, d = hit
        if out_g is not None:
            out_g[...] = g
        if out_d is not None and d is not None:
            out_d[...] = d
        return g, d

    has_gray_pair = True        # the loaders' fast path (an explicit capability: subclasses whose pairs are not (frame, frame) switch it off)

    def resize_is_native(self):
        """is cfg's (W, H) the size the files have (Map-free: 540 x 720, config/mapfree.yaml)?  Only then is the gray plane route-independent: the
        generic per-sample route resizes the 8-bit RGB image and takes the rounded luma of THAT (lib/datasets/utils.py:58-74 -> to_gray), the fast
        route would take the luma first and resize the float gray image (SuperGlue's read_image order, matchers.py:101-104) -- different planes
        whenever a resize really happens.  With a non-native size gray_pair() therefore returns None and every loader takes the generic
        route, so the batched loaders, the reference-view cache and the per-pair plugin see ONE plane."""
        return self.resize is None or all(sz == (int(self.resize[0]), int(self.resize[1])) for sz in self._file_sizes)

    def gray_pair(self, index, want_ref=True, out=None):
        """what the batched loaders need of sample `index`: (gray0 or None, depth0, gray1, depth1, K0, K1, pair_id, (name0, name1)), numpy arrays
        with the values of to_gray(self[index]['image0' / 'image1']) and its depth maps -- bit for bit AT THE FILES' OWN SIZE, the only case this
        route serves (resize_is_native(); otherwise None).  out = (g0, d0, g1, d1) destination arrays (entries
        may be None): the planes are then decoded / copied straight into them.  Returns None when this scene's images are not plain RGB reads
        (black_white training transform): the caller takes the generic sample then."""
        if self.black_white or not self.resize_is_native():
            return None
        sa, ia, sb, ib = self.pairs[index]
        p1, p2 = f"seq{sa}/frame_{ia:05}.jpg", f"seq{sb}/frame_{ib:05}.jpg"
        og0, od0, og1, od1 = out if out is not None else (None, None, None, None)
        g0, d0 = self._gray_frame(p1, keep=True, out_g=og0 if want_ref else None, out_d=od0)
        g1, d1 = self._gray_frame(p2, out_g=og1, out_d=od1)
        return (g0 if want_ref else None, d0, g1, d1, self.K[p1].copy(), self.K[p2].copy(), index * self.sample_factor, (p1, p2))

    def __getitem__(self, index):
        from . import evaluation as E
        sa, ia, sb, ib = self.pairs[index]
        p1, p2 = f"seq{sa}/frame_{ia:05}.jpg", f"seq{sb}/frame_{ib:05}.jpg"
        img1, d1 = self._frame(p1, keep=True)                    # the map frame: the same file for every query of a val / test scene
        img2, d2 = self._frame(p2)
        (q1, t1), (q2, t2) = self.poses[p1], self.poses[p2]
        q12 = E.qmult(q2, E.qinverse(q1))
        t12 = t2 - E.rotate_vector(t1, q12)
        T = np.eye(4, dtype=np.float32); T[:3, :3] = E.quat2mat(q12); T[:3, -1] = t12
        return {"image0": img1, "depth0": d1, "image1": img2, "depth1": d2, "T_0to1": torch.from_numpy(T),
                "K_color0": torch.from_numpy(self.K[p1].copy()), "K_color1": torch.from_numpy(self.K[p2].copy()),
                "dataset_name": "Mapfree", "scene_id": os.path.basename(self.scene_root.rstrip("/")),
                "scene_root": self.scene_root, "pair_id": index * self.sample_factor, "pair_names": (p1, p2)}


class MapFreeSceneMultiFrame(MapFreeScene):
    """multi-frame queries (lib/datasets/mapfree.py:273-368 with load_pairs' sample_offset branches :89-141, :166-205; used by
    RegressionMultiFrameModel): `image1` is the stack of the `query_frames` consecutive VALID frames that end at the query frame
    [T,3,H,W] (depth1 [T,H,W]), poses / intrinsics are those of the LAST frame.
      val / test: queries = every (query_frames + 1)-th seq1 frame starting at index query_frames of the sorted frame list;
      train (overlaps.npz): every overlap row whose query frame has query_frames - 1 valid predecessors in its sequence and whose
      map frame does not fall inside that window (different sequence, or before the window, or after the query frame) -- the valid
      frame lists are taken BEFORE the overlap window is applied, like upstream.
    The device-tracking poses (poses_device.txt) only feed upstream's debug plot and are not read."""
    has_gray_pair = False       # a pair's query is a TUPLE of frames here: the batched loaders take the generic per-sample route

    def __init__(self, scene_root, resize, query_frames=9, estimated_depth=None, overlap_limits=None, black_white=False):
        import re
        T = int(query_frames)
        if T < 2:
            raise ValueError("query_frames must be >= 2 (use MapFreeScene for single-frame queries)")
        super().__init__(scene_root, resize, 1, estimated_depth, None, black_white)
        self.shared_reference = False                           # (the single-frame pair list it was computed from is replaced below)
        self.query_frames, self.sample_factor = T, T + 1
        ov = os.path.join(self.scene_root, "overlaps.npz")
        if os.path.exists(ov):
            f = np.load(ov, allow_pickle=False)
            idxs, overlaps = np.asarray(f["idxs"]).astype(np.int64), np.asarray(f["overlaps"])
            valid = {q: sorted(set(idxs[idxs[:, 0] == q, 1].tolist()) | set(idxs[idxs[:, 2] == q, 3].tolist())) for q in (0, 1)}
            where = {q: {im: k for k, im in enumerate(valid[q])} for q in (0, 1)}
            if overlap_limits is not None and overlap_limits[0] is not None:
                idxs = idxs[np.logical_and(overlap_limits[0] < overlaps, overlaps < overlap_limits[1])]
            pairs = []
            for sa, ia, sb, ib in idxs.tolist():
                k = where[sb][ib] - T + 1
                if k < 0:
                    continue
                window = valid[sb][k:k + T]
                if sa != sb or ia < window[0] or ib < ia:
                    pairs.append((sa, ia, sb, tuple(window)))
            self.pairs = pairs
        else:
            ids = sorted(int(re.search(r"_(\d+)\..*$", fn).group(1)) for fn in self.poses if "seq0" not in fn)
            self.pairs = [(0, 0, 1, tuple(ids[k - T + 1:k + 1])) for k in range(T, len(ids), T + 1)]

    def pair_name(self, index):
        sa, ia, sb, ibs = self.pairs[index]
        return f"seq{sb}/frame_{ibs[-1]:05}.jpg"

    def __getitem__(self, index):
        from . import evaluation as E
        sa, ia, sb, ibs = self.pairs[index]
        p1, p2s = f"seq{sa}/frame_{ia:05}.jpg", tuple(f"seq{sb}/frame_{ib:05}.jpg" for ib in ibs)
        rd = lambda p: read_color_image(os.path.join(self.scene_root, p), self.resize)
        img1, img2 = rd(p1), torch.stack([rd(p) for p in p2s])
        if self.black_white:
            img1, img2 = _luma3(img1), torch.stack([_luma3(im) for im in img2])
        if self.estimated_depth is not None:
            dp = lambda p: read_depth_image(os.path.join(self.scene_root, p).replace(".jpg", f".{self.estimated_depth}.png"))
            d1, d2 = dp(p1), torch.stack([dp(p) for p in p2s])
        else:
            d1 = d2 = torch.tensor([])
        (q1, t1), (q2, t2) = self.poses[p1], self.poses[p2s[-1]]
        q12 = E.qmult(q2, E.qinverse(q1))
        t12 = t2 - E.rotate_vector(t1, q12)
        T = np.eye(4, dtype=np.float32); T[:3, :3] = E.quat2mat(q12); T[:3, -1] = t12
        return {"image0": img1, "depth0": d1, "image1": img2, "depth1": d2, "T_0to1": torch.from_numpy(T),
                "K_color0": torch.from_numpy(self.K[p1].copy()), "K_color1": torch.from_numpy(self.K[p2s[-1]].copy()),
                "dataset_name": "Mapfree", "scene_id": os.path.basename(self.scene_root.rstrip("/")),
                "scene_root": self.scene_root, "pair_id": index * self.sample_factor, "pair_names": (p1, p2s)}


def _luma3(img):
    l = 0.2989 * img[0] + 0.587 * img[1] + 0.114 * img[2]
    return l[None].expand(3, -1, -1).contiguous()


class MissingDataError(FileNotFoundError):
    pass


def list_scenes(cfg, split="val"):
    """the split's scenes as indexable per-scene datasets (MapFreeScene for a real tree under
    DATASET.DATA_ROOT/<split>, SyntheticScene when DATASET.SYNTHETIC is set).  A missing data root is an ERROR
    unless the synthetic stand-in was asked for explicitly: a submission must never be built silently from fake
    scenes."""
    root = cfg.DATASET.DATA_ROOT
    syn = cfg.DATASET.SYNTHETIC if "SYNTHETIC" in cfg.DATASET else None
    if syn:
        n_scenes, frames = (int(syn[0]), int(syn[1])) if isinstance(syn, (list, tuple)) else (2, 4)
        return [SyntheticScene(s, frames, cfg.DATASET.HEIGHT or 720, cfg.DATASET.WIDTH or 540) for s in range(n_scenes)]
    if not (root and os.path.isdir(os.path.join(str(root), split))):
        raise MissingDataError(
            f"DATASET.DATA_ROOT/{split} = {os.path.join(str(root), split)!r} does not exist.  Point DATASET.DATA_ROOT at a Map-free tree, or "
            f"set DATASET.SYNTHETIC: [n_scenes, frames_per_scene] (CLI: --synthetic) to run on the synthetic stand-in on purpose.")
    resize = (cfg.DATASET.WIDTH, cfg.DATASET.HEIGHT) if cfg.DATASET.WIDTH else None
    names = sorted(d for d in os.listdir(os.path.join(str(root), split)) if os.path.isdir(os.path.join(str(root), split, d)))
    if cfg.DATASET.SCENES:
        names = [s for s in names if s in cfg.DATASET.SCENES]
    qf = int(cfg.DATASET.QUERY_FRAME_COUNT or 1)
    if qf > 1:                                                  # mapfree.py:383-396: RegressionMultiFrame data
        limits = (cfg.DATASET.MIN_OVERLAP_SCORE, cfg.DATASET.MAX_OVERLAP_SCORE) if split == "train" else None
        return [MapFreeSceneMultiFrame(os.path.join(str(root), split, s), resize, qf, cfg.DATASET.ESTIMATED_DEPTH, limits,
                                       bool(cfg.DATASET.BLACK_WHITE) and split == "train") for s in names]
    if split == "train":                                        # MapFreeDataset.__init__ (mapfree.py:371-400): sample_factor 1, overlap window
        limits = (cfg.DATASET.MIN_OVERLAP_SCORE, cfg.DATASET.MAX_OVERLAP_SCORE)
        return [MapFreeScene(os.path.join(str(root), split, s), resize, 1, cfg.DATASET.ESTIMATED_DEPTH, limits, bool(cfg.DATASET.BLACK_WHITE))
                for s in names]
    return [MapFreeScene(os.path.join(str(root), split, s), resize, 5, cfg.DATASET.ESTIMATED_DEPTH) for s in names]


def make_loader(cfg, split="val"):
    """batch-1 iterator over the split in (scene, frame) order (submission.py:76-79)"""
    scenes = list_scenes(cfg, split)
    return (collate_batch1(sc[i]) for sc in scenes for i in range(len(sc)))


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the container's CFS quota (cgroup v2 cpu.max / v1 cfs_quota): a GPU box
    shows 256 cores and grants 16"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def to_gray(img):
    """[3,H,W] or [1,H,W] float in [0,1] -> [H,W] float32: the matcher's gray plane (gray_plane above) of the image the floats were made from.
    A colour image of the loaders is byte / 255 per channel, so the bytes are recovered exactly (rint(x * 255)), the luma is ROUNDED to a byte
    as an 8-bit grayscale read of the file would deliver it (matchers.py:101-104), and divided by 255: the online / fused routes then feed
    SuperPoint the very plane the offline route (compute.py -> matchers.read_image) reads.  (Rounds 1-4 fed the unrounded float luma here --
    up to half a grey level away from the offline route's input)  A single-channel image passes through."""
    if img.shape[0] == 1:
        return img[0]
    if isinstance(img, torch.Tensor) and img.device.type != "cpu":
        u = (img.detach().float() * 255.0).round().clamp_(0, 255).to(torch.int32)          # (values a hair outside [0, 1] must not wrap modulo 256)
        g8 = (u[0] * 19595 + u[1] * 38470 + u[2] * 7471 + 0x8000) >> 16
        return g8.to(torch.float32) / 255.0
    # numpy on the calling thread: torch's CPU elementwise kernels fan a 1.5 MB image out over every host core (256 on the
    # GPU boxes), which costs more in thread wake-ups than the arithmetic (measured ~20 ms vs < 1 ms per pair)
    a = img.detach().numpy() if isinstance(img, torch.Tensor) else np.asarray(img)
    u8 = np.clip(np.rint(a.astype(np.float32, copy=False) * np.float32(255)), 0, 255).astype(np.uint8)
    return torch.from_numpy(gray_plane(np.moveaxis(u8, 0, -1)))


# ---- decode in worker PROCESSES (PairBatchLoader(decode="process")) ------------------------------------------------------------------
# The thread pool shares ONE interpreter lock between up to 32 decode threads, the loader thread and the thread that issues the GPU
# step (and 8 ranks on a host would run 8 such crowds): round 4's boxes delivered 350-400 pairs/s from it where the GPU side takes
# ~880.  Here the pairs of a batch are decoded by worker processes that write the gray planes / depth maps straight into a ring of
# SHARED-MEMORY batch slots (torch shared-memory tensors handed to the workers once, at pool start); the parent registers the slots
# as pinned host memory, so the H2D copy of a slot is an asynchronous DMA exactly as from a pinned buffer.
_PW = {}


def _slot_views(flat, n_slots, B, Hh, Ww, has_depth):
    """carve the ring's batch slots out of ONE shared tensor (one shared-memory segment = one descriptor per worker)"""
    per = 2 * B * Hh * Ww + (2 * B * Hh * Ww if has_depth else 0)
    out = []
    for k in range(n_slots):
        o = k * per
        im = flat[o:o + 2 * B * Hh * Ww].view(2 * B, 1, Hh, Ww)
        d0 = flat[o + 2 * B * Hh * Ww:o + 3 * B * Hh * Ww].view(B, Hh, Ww) if has_depth else None
        d1 = flat[o + 3 * B * Hh * Ww:o + 4 * B * Hh * Ww].view(B, Hh, Ww) if has_depth else None
        out.append(dict(images=im, depth0=d0, depth1=d1))
    return out


def _pw_init(scenes, flat, layout):
    # Everything inherited from the parent becomes permanent in this process: a garbage-collection pass of the child must never finalise the
    # PARENT's objects -- a dead multiprocessing.Pool of an earlier loader among them, whose __del__ writes to a queue under a lock that a
    # thread of the parent may have held at fork time (the child then waits for it for ever: round 5's first full CPU test run hung exactly
    # there, in the middle of an allocation inside gray_plane)
    import gc
    gc.freeze()
    torch.set_num_threads(1)
    _PW["scenes"], _PW["slots"] = scenes, _slot_views(flat, *layout)


def _pw_fill(task):
    """decode one pair into slot `k`, position `p`; `want_ref` False: the reference plane is a duplicate the parent copies"""
    import time as _t
    t0 = _t.perf_counter()
    k, p, si, i, want_ref = task
    sc = _PW["scenes"][si]
    sl = _PW["slots"][k]
    im = sl["images"].numpy()
    if getattr(sc, "has_gray_pair", False):                 # gray planes / depth straight from the files' bytes into the slot (MapFreeScene)
        has_d = sl["depth0"] is not None
        fast = sc.gray_pair(i, want_ref, out=(im[2 * p, 0], sl["depth0"].numpy()[p] if has_d else None, im[2 * p + 1, 0],
                                              sl["depth1"].numpy()[p] if has_d else None))
        if fast is not None:
            _, _, _, _, K0, K1, pid, (n0, n1) = fast
            return (np.asarray(K0), np.asarray(K1), int(pid), n1, n0, _t.perf_counter() - t0)
    smp = sc[i]
    npv = lambda t: t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    if want_ref:
        im[2 * p, 0] = npv(to_gray(smp["image0"]))
    im[2 * p + 1, 0] = npv(to_gray(smp["image1"]))
    if sl["depth0"] is not None and smp["depth0"].numel() > 0:
        sl["depth0"].numpy()[p] = npv(smp["depth0"]); sl["depth1"].numpy()[p] = npv(smp["depth1"])
    return (np.asarray(smp["K_color0"]), np.asarray(smp["K_color1"]), int(smp["pair_id"]), smp["pair_names"][1], smp["pair_names"][0],
            _t.perf_counter() - t0)


def _pw_ping(i):
    return os.getpid()


class _ProcessDecoder:
    """pool of decode processes + ring of shared-memory batch slots (see above).  A slot is reused only after the H2D copies issued from it
    have completed (DevicePrefetcher stores its event in `events[k]`)."""

    def __init__(self, scenes, B, Hh, Ww, has_depth, workers, n_slots, pin):
        import torch.multiprocessing as mp
        layout = (n_slots, B, Hh, Ww, has_depth)
        per = 2 * B * Hh * Ww + (2 * B * Hh * Ww if has_depth else 0)
        self.flat = torch.empty(n_slots * per, dtype=torch.float32).share_memory_()
        self.slots = _slot_views(self.flat, *layout)
        self.events = [None] * n_slots
        self.pinned = False
        if pin and torch.cuda.is_available():
            self.pinned = int(torch.cuda.cudart().cudaHostRegister(self.flat.data_ptr(), self.flat.numel() * 4, 0)) == 0
        # the look-up tables and the C helper are set up BEFORE the fork (inherited, not rebuilt per worker), and cyclic garbage of earlier
        # loaders (their pools) is finalised here, in the parent, not in a child (see _pw_init)
        import gc
        _luts(); _host_lib()
        gc.collect()
        # FORK, like torch's DataLoader workers: the children inherit the scene objects and the shared segment without re-importing anything
        # (spawned workers each re-imported torch: 36 s to start 32 of them under a 16-CPU container quota, tools/bench_fused_split.py) and
        # never touch the HIP runtime -- they run PIL / zlib / numpy only
        self.pool = mp.get_context("fork").Pool(workers, initializer=_pw_init, initargs=(scenes, self.flat, layout))
        self.next = 0

    def acquire(self):
        k = self.next
        self.next = (self.next + 1) % len(self.slots)
        ev = self.events[k]
        if ev is not None:
            ev.synchronize()                                   # the copies out of this slot are done
            self.events[k] = None
        return k

    def close(self):
        if self.pool is not None:
            self.pool.terminate(); self.pool.join(); self.pool = None
            import gc
            gc.collect()                                       # the pool's cycles die in THIS process, now
        if self.pinned:
            torch.cuda.cudart().cudaHostUnregister(self.flat.data_ptr())
            self.pinned = False


class PairBatchLoader:
    """Loader adjacency of the fused path (replaces the batch-1 DataLoader of
    lib/datasets/datamodules.py:42-46 + lib/datasets/utils.py:58-81 on the hot path): `items` = [(scene, index)]
    in submission order, grouped into batches of B pairs that SPAN scene boundaries (a 116-pair scene is not 3 full batches + a
    20-pair one: only the very last batch of a rank is short; `span_scenes=False` restores per-scene batches).  A background
    thread decodes the next batches into PINNED host buffers (queue depth `prefetch`), so JPEG/PNG decode and
    the H2D copy of batch i+1 overlap the kernels of batch i (DevicePrefetcher below issues the copies on a
    side stream).  Yields dict(images [2b,1,H,W] f32 gray interleaved (2p = reference view), depth0/depth1 [b,H,W],
    K0/K1 [b,3,3] in the loader's dtype (float64 on Map-free), seed_ids [b] i64 (= data['pair_id'], the RANSAC stream id the per-pair plugin uses),
    global_ids [b] i64, names [b], scene_ids [b] / scene_roots [b] (per pair), scenes_done (ids of the scenes whose LAST pair is in
    this batch), scene_id / scene_root / last_of_scene (of the batch's last pair, kept for single-scene consumers))."""

    def __init__(self, scenes, batch_pairs=32, prefetch=2, pin=None, global_offsets=None, workers=8, span_scenes=True, decode="thread"):
        """workers: decode threads per batch (PIL / zlib / numpy release the GIL): a pair is two JPEGs + one or two 16-bit PNGs,
        ~12 ms of decode on one core, so one thread feeds ~80 pairs/s where the fused pipeline consumes ~700"""
        self.scenes, self.B, self.prefetch = list(scenes), int(batch_pairs), int(prefetch)
        self.workers = max(1, int(workers))
        self._pool = None
        if decode not in ("thread", "process"):
            raise ValueError(f"PairBatchLoader: decode must be 'thread' or 'process', got {decode!r}")
        self.decode, self._proc = decode, None
        self.stats = {}                                        # process decoder: pool start, time in pool.map, summed worker task time, ...
        self.pin = torch.cuda.is_available() if pin is None else pin
        self.offsets = global_offsets
        if self.offsets is None:
            self.offsets, acc = [], 0
            for sc in self.scenes:
                self.offsets.append(acc); acc += len(sc)
        if span_scenes:
            items = [(si, i) for si, sc in enumerate(self.scenes) for i in range(len(sc))]
            self.batches = [items[lo:lo + self.B] for lo in range(0, len(items), self.B)]
        else:
            self.batches = [[(si, i) for i in range(lo, min(lo + self.B, len(sc)))] for si, sc in enumerate(self.scenes) for lo in range(0, len(sc), self.B)]

    _ref_ident = None

    def __len__(self):
        return len(self.batches)

    def _ref_key(self, si, frame):
        """(scene_root, reference frame, file identity): a scene regenerated at the same path (st_mtime_ns / st_size change) must not be served
        the features cached for its previous content; scenes without files (synthetic stand-ins) key on the names alone"""
        root = self.scenes[si].scene_root
        k = (root, frame)
        if self._ref_ident is None:
            self._ref_ident = {}
        ident = self._ref_ident.get(k)
        if ident is None:
            try:
                st = os.stat(os.path.join(root, frame))
                ident = (st.st_mtime_ns, st.st_size)
            except (OSError, TypeError):
                ident = ()
            self._ref_ident[k] = ident
        return (root, frame) + ident

    def _load(self, items):
        """decode the pairs of one batch STRAIGHT INTO the batch's (pinned) buffers: every worker thread decodes its pair, converts to
        gray and writes its slots itself; the loader thread only allocates and collects the small fields (a serial gray conversion +
        copy of 64 images per batch on this thread capped the loader at ~300 pairs/s)"""
        if self.decode == "process" and self.workers > 1:
            return self._load_process(items)
        get = lambda it: self.scenes[it[0]][it[1]]
        b = len(items)
        # shapes / dtypes of the batch buffers come from the first pair; it is decoded on this thread before the pool gets the others (an
        # attempt to overlap the two -- pool tasks that decode, then pool tasks that wait for them and fill -- halved the loader's rate:
        # workers parked in .result() while the next batches' decodes queued behind them; tools/bench_fused_split.py, round 4)
        first = get(items[0])
        Hh, Ww = first["image0"].shape[-2:]
        mk = (lambda *shape, dtype=torch.float32: torch.empty(*shape, dtype=dtype, pin_memory=True)) if self.pin else \
             (lambda *shape, dtype=torch.float32: torch.empty(*shape, dtype=dtype))
        images = mk(2 * b, 1, Hh, Ww)
        has_depth = first["depth0"].numel() > 0
        depth0 = mk(b, Hh, Ww) if has_depth else None
        depth1 = mk(b, Hh, Ww) if has_depth else None
        # packing through the buffers' numpy views: plain memcpy on the worker's own thread (a torch copy_ of a 1.5 MB plane is
        # dispatched to the intra-op thread pool)
        npv = lambda t: t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        im_np = images.numpy()
        d0_np, d1_np = (depth0.numpy(), depth1.numpy()) if has_depth else (None, None)
        gray_of = {}                                           # id(map-frame tensor) -> (the tensor, its gray plane): one conversion per scene
                                                               # keyframe; the tensor is held so that its id cannot be reused inside the batch

        def fill(p, smp=None):
            sc_ = self.scenes[items[p][0]]
            if smp is None and getattr(sc_, "has_gray_pair", False):      # gray planes / depth straight from the files' bytes into the batch buffers
                fast = sc_.gray_pair(items[p][1], True, out=(im_np[2 * p, 0], d0_np[p] if has_depth else None, im_np[2 * p + 1, 0],
                                                             d1_np[p] if has_depth else None))
                if fast is not None:
                    _, _, _, _, K0_, K1_, pid, (n0, n1) = fast
                    return (torch.as_tensor(K0_), torch.as_tensor(K1_), int(pid), n1, n0)
            smp = get(items[p]) if smp is None else smp
            k0 = id(smp["image0"])
            hit = gray_of.get(k0)
            if hit is None or hit[0] is not smp["image0"]:
                hit = gray_of[k0] = (smp["image0"], npv(to_gray(smp["image0"])))
            im_np[2 * p, 0] = hit[1]
            im_np[2 * p + 1, 0] = npv(to_gray(smp["image1"]))
            if has_depth:
                d0_np[p] = npv(smp["depth0"]); d1_np[p] = npv(smp["depth1"])
            return (torch.as_tensor(smp["K_color0"]), torch.as_tensor(smp["K_color1"]), int(smp["pair_id"]), smp["pair_names"][1], smp["pair_names"][0])
        if self.workers > 1 and b > 1:
            if self._pool is None:
                import concurrent.futures
                self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="mfr-decode")
            futs = [self._pool.submit(fill, p) for p in range(1, b)]
            meta = [fill(0, first)] + [f.result() for f in futs]             # order preserved
        else:
            meta = [fill(0, first)] + [fill(p) for p in range(1, b)]
        # intrinsics keep the loader's dtype (float64 on Map-free, float32 on resize=None datasets): the solvers evaluate
        # inv(K) / the K-normalisation in that dtype, as the reference does (include/mfr_hip.h k_dtype)
        kdt = torch.float64 if any(m[0].dtype == torch.float64 or m[1].dtype == torch.float64 for m in meta) else torch.float32
        K0 = mk(b, 3, 3, dtype=kdt); K1 = mk(b, 3, 3, dtype=kdt)
        for p, m in enumerate(meta):
            K0[p] = m[0]; K1[p] = m[1]
        sc = self.scenes[items[-1][0]]
        done = [self.scenes[si].scene_id for si, i in items if i == len(self.scenes[si]) - 1]
        return dict(images=images, depth0=depth0, depth1=depth1, K0=K0, K1=K1,
                    seed_ids=torch.tensor([m[2] for m in meta], dtype=torch.int64),
                    global_ids=torch.tensor([self.offsets[si] + i for si, i in items], dtype=torch.int64),
                    names=[m[3] for m in meta], scene_ids=[self.scenes[si].scene_id for si, _ in items],
                    scene_roots=[self.scenes[si].scene_root for si, _ in items], scenes_done=done,
                    # identity of every pair's REFERENCE view (a val / test scene pairs one keyframe with all its queries, mapfree.py:148-165):
                    # what FusedPosePipeline keys its reference-view feature cache on
                    # -- only for scenes that DECLARE the sharing (`shared_reference`: MapFreeScene val / test); None = do not cache
                    ref_keys=[self._ref_key(si, m[4]) if getattr(self.scenes[si], "shared_reference", False) else None
                              for (si, _), m in zip(items, meta)],
                    scene_id=sc.scene_id, scene_root=sc.scene_root, scene_index=items[-1][0], last_of_scene=bool(done and done[-1] == sc.scene_id))

    POOL_ANSWER_S = 600.0            # a batch is < 1 s of decode; a pool that stays silent this long is dead (never wait for it for ever)

    def _pool_map(self, fn, tasks):
        import multiprocessing as _mp
        try:
            return self._proc.pool.map_async(fn, list(tasks), chunksize=1).get(timeout=self.POOL_ANSWER_S)
        except _mp.TimeoutError:
            self.close()
            raise RuntimeError(f"PairBatchLoader: the decode workers did not answer within {self.POOL_ANSWER_S:.0f} s (pool closed)") from None

    def _load_process(self, items):
        """the same batch through the process pool: workers write into a shared-memory slot; the first pair of the loader's life is
        decoded here once to learn the shapes"""
        import time as _t
        b = len(items)
        st = self.stats
        if self._proc is None:
            t0 = _t.perf_counter()
            first = self.scenes[items[0][0]][items[0][1]]
            Hh, Ww = first["image0"].shape[-2:]
            self._proc = _ProcessDecoder(self.scenes, self.B, Hh, Ww, first["depth0"].numel() > 0, self.workers, max(self.prefetch, 0) + 4, self.pin)
            self._pool_map(_pw_ping, range(self.workers * 2))                          # every worker has started and imported its modules
            st["process_pool_start_s"] = _t.perf_counter() - t0
        pr = self._proc
        t0 = _t.perf_counter()
        k = pr.acquire()
        st["slot_wait_s"] = st.get("slot_wait_s", 0.0) + _t.perf_counter() - t0
        t0 = _t.perf_counter()
        sl = pr.slots[k]
        # reference views: decoded once per distinct (scene, shared reference) of the batch, duplicates copied below
        first_of, want = {}, []
        for p, (si, i) in enumerate(items):
            key = si if getattr(self.scenes[si], "shared_reference", False) else ("pair", p)
            want.append(key not in first_of)
            first_of.setdefault(key, p)
        meta = self._pool_map(_pw_fill, [(k, p, si, i, want[p]) for p, (si, i) in enumerate(items)])
        st["decode_map_s"] = st.get("decode_map_s", 0.0) + _t.perf_counter() - t0
        st["worker_task_s"] = st.get("worker_task_s", 0.0) + sum(m[5] for m in meta)
        st["batches"] = st.get("batches", 0) + 1
        t0 = _t.perf_counter()
        im = sl["images"].numpy()
        for p, (si, i) in enumerate(items):
            if not want[p]:
                q = first_of[si]
                np.copyto(im[2 * p, 0], im[2 * q, 0])
        st["ref_copy_s"] = st.get("ref_copy_s", 0.0) + _t.perf_counter() - t0
        K0 = torch.stack([torch.as_tensor(m[0]) for m in meta]); K1 = torch.stack([torch.as_tensor(m[1]) for m in meta])
        sc = self.scenes[items[-1][0]]
        done = [self.scenes[si].scene_id for si, i in items if i == len(self.scenes[si]) - 1]
        has_depth = sl["depth0"] is not None
        return dict(images=sl["images"][:2 * b], depth0=sl["depth0"][:b] if has_depth else None, depth1=sl["depth1"][:b] if has_depth else None,
                    K0=K0, K1=K1, seed_ids=torch.tensor([m[2] for m in meta], dtype=torch.int64),
                    global_ids=torch.tensor([self.offsets[si] + i for si, i in items], dtype=torch.int64),
                    names=[m[3] for m in meta], scene_ids=[self.scenes[si].scene_id for si, _ in items],
                    scene_roots=[self.scenes[si].scene_root for si, _ in items], scenes_done=done,
                    ref_keys=[self._ref_key(si, m[4]) if getattr(self.scenes[si], "shared_reference", False) else None
                              for (si, _), m in zip(items, meta)],
                    scene_id=sc.scene_id, scene_root=sc.scene_root, scene_index=items[-1][0], last_of_scene=bool(done and done[-1] == sc.scene_id),
                    _slot=(pr, k))

    def close(self):
        if self._proc is not None:
            self._proc.close(); self._proc = None
        if self._pool is not None:
            self._pool.shutdown(wait=False); self._pool = None

    def __iter__(self):
        if self.prefetch <= 0:
            for b in self.batches:
                yield self._load(b)
            return
        import queue
        import threading
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def work():
            try:
                for b in self.batches:
                    if stop.is_set():
                        return
                    q.put(self._load(b))
                q.put(None)
            except BaseException as e:          # surface loader errors in the consumer
                q.put(e)
        th = threading.Thread(target=work, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            while not q.empty():
                q.get_nowait()


class DevicePrefetcher:
    """host batches -> device batches, one batch ahead: the H2D copies of batch i+1 run on a side HIP stream from
    pinned memory while the compute stream works on batch i; an event makes the compute stream wait only for its own batch"""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def _put(self, hb):
        if self.stream is None:
            return dict(hb), None
        with torch.cuda.stream(self.stream):
            db = {k: (v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in hb.items() if k != "_slot"}
            ev = torch.cuda.Event(); ev.record(self.stream)
            if "_slot" in hb:                                  # shared-memory slot of the process decoder: reusable once these copies are done
                pr, k = hb["_slot"]
                pr.events[k] = ev
        return db, ev

    def __iter__(self):
        nxt = None
        for hb in self.loader:
            cur, nxt = nxt, self._put(hb)
            if cur is not None:
                yield self._ready(cur)
        if nxt is not None:
            yield self._ready(nxt)

    def _ready(self, item):
        db, ev = item
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            for v in db.values():
                if isinstance(v, torch.Tensor):
                    v.record_stream(torch.cuda.current_stream(self.device))
        return db


# ----------------------------------------------------------------------------------------------------------------
# training-side loader (SURVEY.md 8f-4 adjacency): scene-balanced sampling + worker processes + pinned batches
# ----------------------------------------------------------------------------------------------------------------
class SceneBalancedSampler:
    """TRAINING.SAMPLER 'scene_balance' (lib/datasets/sampler.py:6-77, used by datamodules.py:24-33): every epoch draws
    `n_per_scene` pair indices from EACH scene (with replacement, or a permutation padded with replacement), then shuffles
    the union.  The generator is seeded once (66 upstream) and runs on across epochs.  One difference by construction: the
    reference trains on one device; here the epoch's index list is dealt round-robin to the ranks (`rank::world`), so an
    epoch is the same set of pairs whatever the world size."""

    def __init__(self, scene_sizes, n_per_scene, replacement=True, seed=66, rank=0, world=1, shuffle=True):
        self.sizes = [int(n) for n in scene_sizes]
        self.n, self.replacement, self.shuffle = int(n_per_scene), bool(replacement), bool(shuffle)
        self.rank, self.world = int(rank), int(world)
        self.gen = torch.Generator().manual_seed(int(seed))

    def __len__(self):
        total = len(self.sizes) * self.n
        return (total - self.rank + self.world - 1) // self.world

    def epoch_indices(self):
        """the whole epoch: global (concatenated) pair indices, identical on every rank"""
        out, low = [], 0
        for size in self.sizes:
            if size <= 0:
                continue
            if self.replacement:
                r = torch.randint(low, low + size, (self.n,), generator=self.gen, dtype=torch.int64)
            else:
                r = torch.randperm(size, generator=self.gen) + low
                r = r[:self.n] if size >= self.n else torch.cat([r, torch.randint(low, low + size, (self.n - size,), generator=self.gen, dtype=torch.int64)])
            out.append(r)
            low += size
        idx = torch.cat(out) if out else torch.zeros(0, dtype=torch.int64)
        if self.shuffle:
            idx = idx[torch.randperm(len(idx), generator=self.gen)]
        return idx

    def __iter__(self):
        return iter(self.epoch_indices()[self.rank::self.world].tolist())


class _ConcatScenes(torch.utils.data.Dataset):
    """scenes back to back; only what the regression model consumes is collated (strings / empty depth tensors are dropped)"""

    KEYS = ("image0", "image1", "T_0to1", "K_color0", "K_color1")

    def __init__(self, scenes):
        self.scenes = list(scenes)
        self.cum = np.cumsum([0] + [len(s) for s in self.scenes])

    def __len__(self):
        return int(self.cum[-1])

    def __getitem__(self, i):
        si = int(np.searchsorted(self.cum, i, side="right") - 1)
        d = self.scenes[si][int(i - self.cum[si])]
        out = {k: d[k] for k in self.KEYS}
        for k in ("depth0", "depth1"):
            if isinstance(d.get(k), torch.Tensor) and d[k].numel():
                out[k] = d[k]
        return out


class TrainPairLoader:
    """batches of TRAINING.BATCH_SIZE pairs for regression/train.py: torch DataLoader worker processes decode JPEGs
    (TRAINING.NUM_WORKERS), batches arrive in pinned memory and are uploaded on a side stream one batch ahead
    (DevicePrefetcher).  Iterating yields ONE epoch of this rank's share; `forever()` chains epochs."""

    def __init__(self, scenes, batch_size, sampler=None, num_workers=0, device="cpu", drop_last=True):
        self.ds = _ConcatScenes(scenes)
        self.device = torch.device(device)
        self.dl = torch.utils.data.DataLoader(self.ds, batch_size=int(batch_size), sampler=sampler, num_workers=int(num_workers or 0),
                                              pin_memory=self.device.type == "cuda", drop_last=drop_last,
                                              persistent_workers=bool(num_workers))

    def __len__(self):
        return len(self.dl)

    def __iter__(self):
        return iter(DevicePrefetcher(self.dl, self.device))

    def forever(self):
        while True:
            yield from self


def make_train_loaders(cfg, device, rank=0, world=1):
    """(train TrainPairLoader, this rank's validation batches as a list factory) from cfg, like DataModule
    (lib/datasets/datamodules.py:35-68): scene-balanced sampling for training, drop_last sequential validation"""
    train = list_scenes(cfg, "train")
    sampler = None
    if cfg.TRAINING.SAMPLER == "scene_balance":
        sampler = SceneBalancedSampler([len(s) for s in train], cfg.TRAINING.N_SAMPLES_SCENE, bool(cfg.TRAINING.SAMPLE_WITH_REPLACEMENT),
                                       rank=rank, world=world)
    tl = TrainPairLoader(train, cfg.TRAINING.BATCH_SIZE, sampler, cfg.TRAINING.NUM_WORKERS, device)
    val = list_scenes(cfg, "val")
    vds = _ConcatScenes(val)
    share = list(range(len(vds)))[rank::world]
    vl = TrainPairLoader(val, cfg.TRAINING.BATCH_SIZE, share, cfg.TRAINING.NUM_WORKERS, device)
    return tl, vl
