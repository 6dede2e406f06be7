"""SuperPoint on MI355X: the 3x3 convolutions through the fused Winograd/MFMA kernel of
csrc/winograd_split.hip / winograd_conv.hip (conv1a: csrc/elementwise.hip; the declared option CONV = 'miopen', options.py, selects the
library convolution + epilogue kernels instead), the 1x1 heads through csrc/elementwise.hip (convPb) and csrc/gemm_split.hip (convDb, fed by
the NCHW -> rows transpose), everything after the conv heads through the hand-written HIP kernels of csrc/superpoint_post.hip.

Reference call site: SuperGlue_matcher (etc/feature_matching_baselines/matchers.py:62-120; nms 4,
threshold 0.005, max 1024 keypoints :65-67); network = un-vendored magicleap submodule, restated
per SURVEY.md Appendix A.2 with upstream parameter names (conv1a ... convDb) so that
superpoint_v1.pth loads unchanged.
"""
import torch
import torch.nn.functional as F

from .. import _lib, options
from .conv import WinoConv3x3

CAND_CAP = 32768


class SuperPointHIP:
    def __init__(self, state_dict, device="cuda", nms_radius=4, keypoint_threshold=0.005, max_keypoints=1024,
                 remove_borders=4):
        _lib.load(require_gpu=True)
        if max_keypoints <= 0 or max_keypoints > 1024:
            raise ValueError("max_keypoints must be in [1, 1024] (matchers.py:67 uses 1024)")
        self.device = torch.device(device)
        self.nms_radius, self.thr = int(nms_radius), float(keypoint_threshold)
        self.K, self.border = int(max_keypoints), int(remove_borders)
        self.fused_conv_relu = bool(options.get("FUSED_CONV_RELU"))
        self.fused_conv1 = bool(options.get("FUSED_CONV1"))
        self.use_wino = options.get("CONV") == "wino"                     # "miopen": library conv + epilogue kernels (options.py)
        self.w = {k: v.to(self.device, torch.float32).contiguous() for k, v in state_dict.items()}
        # 1x1 heads as plain matrices
        self.w["convDb.mat"] = self.w["convDb.weight"].reshape(256, 256).contiguous()
        from .linear import SplitLinear
        self.convDb = SplitLinear(self.w["convDb.mat"], self.w["convDb.bias"])
        # detector head convPb (256 -> 65, 1x1) through the implicit-GEMM kernel in the f16x2 arithmetic (round 5; a workgroup = 128 pixels of ONE
        # image, so a pixel's logits do not depend on the batch it is in -- what the reference-view feature cache needs); the bf16x3 option keeps
        # round 4's fp32 FMA-chain kernel
        self.head_pb = None
        if self.use_wino and self.convDb.split == "f16x2":
            from .conv import IgemmConv
            self.head_pb = IgemmConv(self.w["convPb.weight"], self.w["convPb.bias"], 1, 0)
        # Winograd-transformed 3x3 filters, packed in MFMA operand order (csrc/winograd_conv.hip), once per weight set
        self.upk = {}
        for name in ("conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convDa"):
            w = self.w[name + ".weight"]
            if tuple(w.shape[2:]) != (3, 3):
                continue
            self.upk[name] = WinoConv3x3(w, self.w[name + ".bias"])      # both packed filter forms; kernel chosen per shape (nets/conv.py)

    def _conv(self, x, name, relu=True, pool=False):
        """3x3 layer = ONE launch of the fused Winograd/MFMA kernel (conv + bias + ReLU [+ 2x2 max-pool],
        csrc/winograd_conv.hip).  Option CONV = 'miopen' (or an unsupported shape): library conv without bias + one fused
        HIP epilogue pass (csrc/elementwise.hip)."""
        lib = _lib.load()
        w, b = self.w[name + ".weight"], self.w[name + ".bias"]
        pad = w.shape[-1] // 2
        if self.use_wino and relu and name in self.upk:
            return self.upk[name](x, act=1, pool=pool)
        if not relu:
            if w.shape[-1] == 1 and name == "convPb" and self.head_pb is not None:
                return self.head_pb(x)
            if w.shape[-1] == 1 and self.use_wino:   # 1x1 head (convPb): own kernel, one ascending FMA chain per logit (batch-size independent bits)
                B, C, H, W = x.shape
                x = x.contiguous()
                y = torch.empty(B, w.shape[0], H, W, dtype=torch.float32, device=x.device)
                _lib.check(lib.mfr_conv1x1_nchw(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), B, C, w.shape[0], H * W, _lib.ptr(y), _lib.stream_ptr()), "mfr_conv1x1_nchw")
                return y
            if w.shape[-1] == 1:            # options CONV = "miopen": one batched library GEMM [Cout,Cin] x [Cin,HW] per image
                B, C, H, W = x.shape
                return torch.baddbmm(b.view(1, -1, 1), w.view(1, w.shape[0], C).expand(B, -1, -1), x.reshape(B, C, H * W)).view(B, -1, H, W)
            return F.conv2d(x, w, b, padding=pad)
        if self.fused_conv_relu and not pool:
            return torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], [pad, pad], [1, 1], 1)
        x = F.conv2d(x, w, None, padding=pad).contiguous()
        B, C, H, W = x.shape
        if pool:
            y = torch.empty(B, C, H // 2, W // 2, dtype=x.dtype, device=x.device)
            _lib.check(lib.mfr_bias_pool2_relu_nchw(_lib.ptr(x), _lib.ptr(b), B, C, H, W, _lib.ptr(y), _lib.stream_ptr()),
                       "mfr_bias_pool2_relu_nchw")
            return y
        _lib.check(lib.mfr_bias_relu_nchw(_lib.ptr(x), _lib.ptr(b), B, C, H * W, _lib.stream_ptr()), "mfr_bias_relu_nchw")
        return x

    def _conv1a(self, image):
        """first layer (1 -> 64 channels) as one fused HIP pass (csrc/elementwise.hip)"""
        lib = _lib.load()
        B, C, H, W = image.shape
        w, b = self.w["conv1a.weight"], self.w["conv1a.bias"]
        if C != 1 or w.shape[0] != 64 or (W & 3):
            return self._conv(image, "conv1a")
        y = torch.empty(B, 64, H, W, dtype=torch.float32, device=image.device)
        _lib.check(lib.mfr_conv3x3_c1_relu(_lib.ptr(image.contiguous()), _lib.ptr(w), _lib.ptr(b), B, H, W, 64, _lib.ptr(y),
                                           _lib.stream_ptr()), "mfr_conv3x3_c1_relu")
        return y

    def _conv1ab(self, image):
        """conv1a + ReLU + conv1b + ReLU + 2x2 max-pool in ONE kernel (csrc/winograd_split.hip mfr_sp_conv1ab_f16x2, option FUSED_CONV1): the
        64-channel full-resolution intermediate never reaches memory; bit-identical to the two launches it replaces"""
        lib = _lib.load()
        B, C, H, W = image.shape
        cv = self.upk.get("conv1b")
        if (not self.fused_conv1 or cv is None or cv.split != "f16x2" or C != 1 or H < 2 or W < 2
                or tuple(self.w["conv1a.weight"].shape) != (64, 1, 3, 3) or (cv.ci, cv.co) != (64, 64)):
            return None
        y = torch.empty(B, 64, H // 2, W // 2, dtype=torch.float32, device=image.device)
        _lib.check(lib.mfr_sp_conv1ab_f16x2(_lib.ptr(image.contiguous()), _lib.ptr(self.w["conv1a.weight"]), _lib.ptr(self.w["conv1a.bias"]),
                                            _lib.ptr(cv.u_split), _lib.ptr(cv.b), B, H, W, _lib.ptr(y), _lib.stream_ptr()), "mfr_sp_conv1ab_f16x2")
        return y

    def encode(self, image):
        x = self._conv1ab(image) if self.use_wino else None
        if x is None:
            x = self._conv1a(image); x = self._conv(x, "conv1b", pool=True)
        x = self._conv(x, "conv2a"); x = self._conv(x, "conv2b", pool=True)
        x = self._conv(x, "conv3a"); x = self._conv(x, "conv3b", pool=True)
        x = self._conv(x, "conv4a"); x = self._conv(x, "conv4b")
        return x

    # -- stage wrappers (each is one C-ABI call; exposed for stage-level parity tests) --
    def score_map(self, logits):
        lib = _lib.load()
        B, C, Hc, Wc = logits.shape
        assert C == 65
        logits = logits.contiguous()
        out = torch.empty(B, Hc * 8, Wc * 8, dtype=torch.float32, device=logits.device)
        _lib.check(lib.mfr_sp_scoremap(_lib.ptr(logits), B, Hc, Wc, _lib.ptr(out), _lib.stream_ptr()), "mfr_sp_scoremap")
        return out

    def nms_candidates(self, scores, want_dense=False):
        lib = _lib.load()
        B, H, W = scores.shape
        cand = torch.empty(B, CAND_CAP, dtype=torch.int64, device=scores.device)
        cnt = torch.empty(B, dtype=torch.int32, device=scores.device)
        dense = torch.empty_like(scores) if want_dense else None
        _lib.check(lib.mfr_sp_nms_candidates(_lib.ptr(scores), B, H, W, self.nms_radius, self.thr, self.border,
                                             _lib.ptr(dense), _lib.ptr(cand), CAND_CAP, _lib.ptr(cnt),
                                             _lib.stream_ptr()), "mfr_sp_nms_candidates")
        return cand, cnt, dense

    def select(self, cand, cnt, W):
        lib = _lib.load()
        B = cand.shape[0]
        kpts = torch.empty(B, self.K, 2, dtype=torch.float32, device=cand.device)
        sc = torch.empty(B, self.K, dtype=torch.float32, device=cand.device)
        n = torch.empty(B, dtype=torch.int32, device=cand.device)
        _lib.check(lib.mfr_sp_select_topk(_lib.ptr(cand), CAND_CAP, _lib.ptr(cnt), B, W, self.K, _lib.ptr(kpts),
                                          _lib.ptr(sc), _lib.ptr(n), _lib.stream_ptr()), "mfr_sp_select_topk")
        return kpts, sc, n

    def sample(self, dense_nhwc, kpts, n):
        lib = _lib.load()
        B, Hc, Wc, C = dense_nhwc.shape
        assert C == 256
        desc = torch.empty(B, self.K, 256, dtype=torch.float32, device=kpts.device)
        _lib.check(lib.mfr_sp_sample_descriptors(_lib.ptr(dense_nhwc.contiguous()), B, Hc, Wc, _lib.ptr(kpts),
                                                 _lib.ptr(n), self.K, _lib.ptr(desc), _lib.stream_ptr()),
                   "mfr_sp_sample_descriptors")
        return desc

    @torch.no_grad()
    def __call__(self, image):
        """image [B2,1,H,W] f32 in [0,1] on the GPU -> dict(kpts [B2,K,2] (x,y), scores [B2,K],
        desc [B2,K,256] token-major, n [B2] i32).  Rows >= n are zero."""
        x = self.encode(image)
        logits = self._conv(self._conv(x, "convPa"), "convPb", relu=False)
        scores = self.score_map(logits)
        cand, cnt, _ = self.nms_candidates(scores)
        kpts, sc, n = self.select(cand, cnt, scores.shape[2])
        # 1x1 descriptor head (convDb) through the own GEMM: every row's K loop runs in the same order whatever the number of rows.  Round 6: convDa writes its
        # output token-major itself when the direct kernel runs it (mfr_conv3x3_direct_f16x2_rows); else NCHW -> rows by csrc/elementwise.hip
        cDa_rows = self.upk["convDa"].rows(x, act=1) if (self.use_wino and "convDa" in self.upk) else None
        if cDa_rows is not None:
            B, Hc, Wc, C = cDa_rows.shape
            rows = cDa_rows.view(B * Hc * Wc, C)
        else:
            cDa = self._conv(x, "convDa")
            B, C, Hc, Wc = cDa.shape
            rows = torch.empty(B * Hc * Wc, C, dtype=torch.float32, device=cDa.device)
            _lib.check(_lib.load().mfr_nchw_to_rows(_lib.ptr(cDa.contiguous()), None, B, C, Hc * Wc, 0, _lib.ptr(rows), Hc * Wc * C, C, _lib.stream_ptr()),
                       "mfr_nchw_to_rows")
        dense = self.convDb(rows)
        desc = self.sample(dense.view(B, Hc, Wc, 256), kpts, n)
        return dict(kpts=kpts, scores=sc, desc=desc, n=n)
