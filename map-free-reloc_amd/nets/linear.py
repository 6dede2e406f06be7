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
        self.N, self.K = int(weight.shape[0]), int(weight.shape[1])
        nb = self._pack_bytes(self.N, self.K)
        if nb == 0:
            raise ValueError(f"SplitLinear: K = {self.K} must be a multiple of 32")
        w = weight.contiguous().float()
        self.packed = torch.empty(nb, dtype=torch.uint8, device=w.device)
        _lib.check(self._pack(_lib.ptr(w), self.N, self.K, _lib.ptr(self.packed), _lib.stream_ptr()), f"mfr_gemm_{self.split}_pack")
        self.bias = None if bias is None else bias.contiguous().float()

    def __call__(self, x, out=None, relu=False, accumulate=False, kernel_flag=0):
        """x [M, K] (row stride x.stride(0), unit column stride) -> out [M, N] (allocated when None); accumulate: out += result.
        kernel_flag: 0 (default kernel) | 4 | 8 -- the other kernel generations, for the bitwise-agreement test"""
        assert x.dim() == 2 and x.shape[1] == self.K and x.stride(1) == 1 and x.dtype == torch.float32
        M = x.shape[0]
        if out is None:
            assert not accumulate
            out = torch.empty(M, self.N, dtype=torch.float32, device=x.device)
        assert out.shape == (M, self.N) and out.stride(1) == 1
        _lib.check(self._gemm(x.data_ptr(), x.stride(0), _lib.ptr(self.packed), _lib.ptr(self.bias), out.data_ptr(), out.stride(0),
                              M, self.N, self.K, (1 if relu else 0) | (2 if accumulate else 0) | kernel_flag, _lib.stream_ptr()), f"mfr_gemm_{self.split}")
        return out
