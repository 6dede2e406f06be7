"""Linear layers of the matcher transformers through csrc/gemm_split.hip (include/mfr_hip.h mfr_gemm_f16x2 / mfr_gemm_bf16x3): fp32 in /
fp32 out on the 16-bit matrix cores at fp32 accuracy.  `SplitLinear` packs a weight once and applies it to row-major activations with an
arbitrary row stride (so the fused [x | message] buffers of nets/superglue.py and nets/loftr.py are used in place).  The arithmetic
(`HIP.SPLIT`: 'f16x2', the default, or 'bf16x3') is resolved ONCE, when the weight is packed, and stays with the object."""
import torch

from .. import _lib, options


class SplitLinear:
    def __init__(self, weight, bias=None, split=None):
        """weight [N, K] f32 (K % 32 == 0), bias [N] or None; split: 'f16x2' | 'bf16x3' (None: options.get('SPLIT'))"""
        lib = _lib.load(require_gpu=True)
        self.split = split or options.get("SPLIT")
        if self.split not in ("f16x2", "bf16x3"):
            raise ValueError(f"SplitLinear: unknown split {self.split!r}")
        self._pack_bytes = getattr(lib, f"mfr_gemm_{self.split}_pack_bytes")
        self._pack = getattr(lib, f"mfr_gemm_{self.split}_pack")
        self._gemm = getattr(lib, f"mfr_gemm_{self.split}")
        self._gemm_ln = getattr(lib, f"mfr_gemm_{self.split}_ln")
        self.N, self.K = int(weight.shape[0]), int(weight.shape[1])
        nb = self._pack_bytes(self.N, self.K)
        if nb == 0:
            raise ValueError(f"SplitLinear: K = {self.K} must be a multiple of 32")
        w = weight.contiguous().float()
        self.packed = torch.empty(nb, dtype=torch.uint8, device=w.device)
        _lib.check(self._pack(_lib.ptr(w), self.N, self.K, _lib.ptr(self.packed), _lib.stream_ptr()), f"mfr_gemm_{self.split}_pack")
        self.bias = None if bias is None else bias.contiguous().float()

    def windows(self, feat_nhwc, img_ids, cell_ids, wc, stride, win, out, window_bias=None):
        """out [nwin * win^2, N] (row-strided view) = tokens of the win x win windows of the NHWC map (zero padded) @ W^T (+ bias) (+ window_bias [nwin, N]
        per window): mfr_gemm_*_windows -- the fine-level gather, the merge_feat product and the broadcast add in one launch"""
        Bimg, Hf, Wf, C = feat_nhwc.shape
        nwin = img_ids.numel()
        assert C == self.K and feat_nhwc.is_contiguous() and out.shape == (nwin * win * win, self.N) and out.stride(1) == 1
        assert img_ids.dtype == torch.int32 and cell_ids.dtype == torch.int32 and img_ids.is_contiguous() and cell_ids.is_contiguous()
        assert window_bias is None or (window_bias.shape == (nwin, self.N) and window_bias.is_contiguous())
        if getattr(self, "_zero_row", None) is None:
            self._zero_row = torch.zeros(self.K, dtype=torch.float32, device=feat_nhwc.device)
        lib = _lib.load()
        _lib.check(getattr(lib, f"mfr_gemm_{self.split}_windows")(_lib.ptr(feat_nhwc), Bimg, Hf, Wf, C, _lib.ptr(img_ids), _lib.ptr(cell_ids), nwin, wc, stride, win,
                                                                   _lib.ptr(self._zero_row), _lib.ptr(self.packed), _lib.ptr(self.bias), _lib.ptr(window_bias),
                                                                   out.data_ptr(), out.stride(0), self.N, _lib.stream_ptr()), f"mfr_gemm_{self.split}_windows")
        return out

    def ln_fusable(self):
        """can the LayerNorm that follows this layer run in its epilogue (mfr_gemm_*_ln: one 128-feature block, the LDS-DMA kernel)?"""
        return self.N == 128 and self.K % 64 == 0

    def __call__(self, x, out=None, relu=False, accumulate=False, kernel_flag=0, ln=None, eps=1e-5):
        """x [M, K] (row stride x.stride(0), unit column stride) -> out [M, N] (allocated when None); accumulate: out += result.
        kernel_flag: 0 (default kernel) | 4 | 8 -- the other kernel generations, for the bitwise-agreement test.
        ln = (gamma, beta): out = [out +] LayerNorm(x W^T + b) in the same launch (N = 128 only: ln_fusable())"""
        assert x.dim() == 2 and x.shape[1] == self.K and x.stride(1) == 1 and x.dtype == torch.float32
        M = x.shape[0]
        if out is None:
            assert not accumulate
            out = torch.empty(M, self.N, dtype=torch.float32, device=x.device)
        assert out.shape == (M, self.N) and out.stride(1) == 1
        if ln is not None:
            assert self.ln_fusable() and not relu and kernel_flag == 0
            _lib.check(self._gemm_ln(x.data_ptr(), x.stride(0), _lib.ptr(self.packed), _lib.ptr(self.bias), _lib.ptr(ln[0]), _lib.ptr(ln[1]), float(eps),
                                     out.data_ptr(), out.stride(0), M, self.N, self.K, 1 if accumulate else 0, _lib.stream_ptr()), f"mfr_gemm_{self.split}_ln")
            return out
        _lib.check(self._gemm(x.data_ptr(), x.stride(0), _lib.ptr(self.packed), _lib.ptr(self.bias), out.data_ptr(), out.stride(0),
                              M, self.N, self.K, (1 if relu else 0) | (2 if accumulate else 0) | kernel_flag, _lib.stream_ptr()), f"mfr_gemm_{self.split}")
        return out


class SplitBatchedNT:
    """out[b] = x[b] @ w[b]^T * out_mul for a batch of same-shape products (mfr_gemm_f16x2_batched): the matchers' score / similarity
    matrices.  w is packed per call into scratch this object keeps (one allocation per shape); out_mul must be a power of two."""

    def __init__(self):
        self.lib = _lib.load(require_gpu=True)
        self.scratch = {}

    def __call__(self, x, w, out_mul=1.0, out=None):
        assert x.dim() == 3 and w.dim() == 3 and x.shape[0] == w.shape[0] and x.shape[2] == w.shape[2] and x.dtype == w.dtype == torch.float32
        nb, M, K = x.shape
        N = w.shape[1]
        assert x.stride(2) == 1 and w.stride(2) == 1
        per = self.lib.mfr_gemm_f16x2_pack_bytes(N, K)
        if per == 0:
            raise ValueError(f"SplitBatchedNT: K = {K} must be a multiple of 32")
        key = (nb, N, K, x.device)
        if key not in self.scratch:
            self.scratch[key] = torch.empty(nb * per, dtype=torch.uint8, device=x.device)
        pk = self.scratch[key]
        if out is None:
            out = torch.empty(nb, M, N, dtype=torch.float32, device=x.device)
        assert out.shape == (nb, M, N) and out.stride(2) == 1
        st = _lib.stream_ptr()
        _lib.check(self.lib.mfr_gemm_f16x2_pack_batched(w.data_ptr(), w.stride(1), nb, w.stride(0), N, K, float(out_mul), _lib.ptr(pk), st), "mfr_gemm_f16x2_pack_batched")
        _lib.check(self.lib.mfr_gemm_f16x2_batched(x.data_ptr(), x.stride(1), x.stride(0), _lib.ptr(pk), None, out.data_ptr(), out.stride(1), out.stride(0),
                                                   nb, M, N, K, 0, st), "mfr_gemm_f16x2_batched")
        return out


class FusedMlpLn:
    """out = [out +] LayerNorm_128(relu(x W1^T + b1) W2^T + b2) in one launch (mfr_mlp_ln_*, csrc/gemm_split.hip mlp_ln_kernel): the MLP + norm2 (+ residual) of
    a LoFTR encoder layer at d_model 128.  Built from the two SplitLinear objects of the layer (their packed weights are used as they are)."""

    def __init__(self, lin1, lin2):
        assert lin1.split == lin2.split and lin1.N == 256 and lin2.N == 128 and lin2.K == 256 and lin1.K % 64 == 0
        self.l1, self.l2 = lin1, lin2
        self._fn = getattr(_lib.load(require_gpu=True), f"mfr_mlp_ln_{lin1.split}")

    def __call__(self, x, out, ln, accumulate=False, eps=1e-5):
        assert x.dim() == 2 and x.shape[1] == self.l1.K and x.stride(1) == 1 and x.dtype == torch.float32
        M = x.shape[0]
        assert out.shape == (M, 128) and out.stride(1) == 1
        _lib.check(self._fn(x.data_ptr(), x.stride(0), self.l1.K, _lib.ptr(self.l1.packed), _lib.ptr(self.l1.bias), _lib.ptr(self.l2.packed), _lib.ptr(self.l2.bias),
                            _lib.ptr(ln[0]), _lib.ptr(ln[1]), float(eps), out.data_ptr(), out.stride(0), M, 1 if accumulate else 0, _lib.stream_ptr()),
                   f"mfr_mlp_ln_{self.l1.split}")
        return out
