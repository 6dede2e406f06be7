"""3x3 / stride 1 / pad 1 convolution layers of the matcher backbones: packed filters + kernel choice.

Three hand-written kernels compute the same layer (include/mfr_hip.h):
  * mfr_conv3x3_direct_f16x2 (csrc/conv_direct.hip, round 6)  direct implicit GEMM, halo tile split once into LDS in operand form, f16x2 arithmetic: the
                             default for SPLIT = 'f16x2' -- 1.0-1.3x the Winograd kernel on every layer shape of the two backbones at the bench batches
                             (profiles/r06_ab_direct_conv_halo.json; matrix pipe 62-74 % busy, the chip at its power limit: profiles/r06_pmc_dconv_*.json);
  * mfr_conv3x3_wino_f16x2 / _bf16x3  (csrc/winograd_split.hip)  Winograd F(2x2,3x3) on the 16-bit matrix cores at fp32 accuracy by operand
                             splitting; the arithmetic is `HIP.SPLIT` ('f16x2', the default: three partial products; 'bf16x3': six),
                             resolved when the layer's filters are packed;
  * mfr_conv3x3_wino         (csrc/winograd_conv.hip)    the same transform on the exact-fp32 matrix cores.
All meet the same parity bar (<= 2e-5 against a float64 convolution, tests/test_gpu_winograd_*.py).  The split kernel tiles the image in
blocks of 16 Winograd tiles along x; for the narrow odd-width maps of SuperPoint's 1/8 level (67 pixels = 34 tiles -> 48 computed) the
exact-fp32 kernel's linear tiling wastes nothing, so the choice is made per layer shape, once, here (`HIP.CONV_KERNEL` forces one)."""
import torch

from .. import _lib, options

# tile-block waste along x above which the exact-fp32 kernel (linear tiling) is the faster one, per arithmetic (tools/bench_conv.py:
# profiles/r03_ab_conv.json for bf16x3, profiles/r05_ab_conv.json for f16x2)
WASTE_LIMIT = {"bf16x3": 0.2, "f16x2": 0.5}


def prefer_split(H, W, split=None):
    """tile-block waste of the split kernel along x: ceil(tiles / 16) * 16 / tiles - 1; above the limit the exact kernel wins"""
    mode = options.get("CONV_KERNEL")                        # "split" / "exact" force one kernel (A/B runs, tests; options.py)
    if mode in ("split", "direct"):
        return True
    if mode == "exact":
        return False
    tiles = (W + 1) // 2
    return (-(-tiles // 16) * 16) / tiles - 1.0 <= WASTE_LIMIT[split or options.get("SPLIT")]


class WinoConv3x3:
    """one 3x3 layer: the packed filter forms (built once per weight set) and the launch"""

    def __init__(self, weight, bias, split=None):
        lib = _lib.load(require_gpu=True)
        self.split = split or options.get("SPLIT")
        if self.split not in ("f16x2", "bf16x3"):
            raise ValueError(f"WinoConv3x3: unknown split {self.split!r}")
        self.w, self.b = weight.contiguous(), bias
        self.co, self.ci = int(weight.shape[0]), int(weight.shape[1])
        dev = weight.device
        self.u_exact = self.u_split = None
        nb = lib.mfr_wino_filter_bytes(self.ci, self.co)
        if nb:
            self.u_exact = torch.empty(nb // 4, dtype=torch.float32, device=dev)
            _lib.check(lib.mfr_wino_filter_transform(_lib.ptr(self.w), self.ci, self.co, _lib.ptr(self.u_exact), _lib.stream_ptr()),
                       "mfr_wino_filter_transform")
        nb = getattr(lib, f"mfr_wino_{self.split}_filter_bytes")(self.ci, self.co)
        self.u_split = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(getattr(lib, f"mfr_wino_{self.split}_filter_transform")(_lib.ptr(self.w), self.ci, self.co, _lib.ptr(self.u_split), _lib.stream_ptr()),
                   f"mfr_wino_{self.split}_filter_transform")
        self._conv_split = getattr(lib, f"mfr_conv3x3_wino_{self.split}")
        self.direct = None                                       # built on first use (CONV_KERNEL 'auto' / 'direct', f16x2)

    def rows(self, x, act=0):
        """[B, H, W, Cout] token-major output when the direct kernel runs this layer (CONV_KERNEL 'auto' / 'direct', f16x2, Cout % 4 == 0), else None"""
        mode = options.get("CONV_KERNEL")
        if self.co % 4 or not (mode == "direct" or (mode == "auto" and self.split == "f16x2")):
            return None
        return self._direct().rows(x, act=act)

    def _direct(self):
        if self.direct is None:
            if self.split != "f16x2":
                raise ValueError("CONV_KERNEL = 'direct' exists in the f16x2 arithmetic only")
            self.direct = DirectConv3x3(self.w, self.b)
        return self.direct

    def __call__(self, x, act=0, pool=False, residual=None):
        """act: 0 none, 1 ReLU, 2 LeakyReLU(0.01); pool: fused 2x2 max-pool; residual [B,Cout,H,W] added before the activation"""
        lib = _lib.load()
        mode = options.get("CONV_KERNEL")
        if mode == "direct" or (mode == "auto" and self.split == "f16x2"):
            return self._direct()(x, act=act, pool=pool, residual=residual)
        x = x.contiguous()
        B, C, H, W = x.shape
        y = torch.empty((B, self.co, H // 2, W // 2) if pool else (B, self.co, H, W), dtype=torch.float32, device=x.device)
        res = _lib.ptr(residual.contiguous()) if residual is not None else None
        if self.u_exact is None or prefer_split(H, W, self.split):
            _lib.check(self._conv_split(_lib.ptr(x), _lib.ptr(self.u_split), _lib.ptr(self.b), res, B, C, self.co, H, W, int(act), int(pool),
                                        _lib.ptr(y), _lib.stream_ptr()), f"mfr_conv3x3_wino_{self.split}")
        else:
            _lib.check(lib.mfr_conv3x3_wino(_lib.ptr(x), _lib.ptr(self.u_exact), _lib.ptr(self.b), res, B, C, self.co, H, W, int(act), int(pool),
                                            _lib.ptr(y), _lib.stream_ptr()), "mfr_conv3x3_wino")
        return y


class DirectConv3x3:
    """the same 3x3 / stride 1 / pad 1 layer as a direct implicit GEMM with an LDS-staged halo tile (csrc/conv_direct.hip, mfr_conv3x3_direct_f16x2;
    f16x2 arithmetic only).  Same call signature as WinoConv3x3."""

    def __init__(self, weight, bias):
        lib = _lib.load(require_gpu=True)
        self.w, self.b = weight.contiguous().float(), bias
        self.co, self.ci = int(weight.shape[0]), int(weight.shape[1])
        self.split = "f16x2"
        self.packed = torch.empty(lib.mfr_conv3x3_direct_f16x2_filter_bytes(self.ci, self.co), dtype=torch.uint8, device=weight.device)
        _lib.check(lib.mfr_conv3x3_direct_f16x2_filter_pack(_lib.ptr(self.w), self.ci, self.co, _lib.ptr(self.packed), _lib.stream_ptr()),
                   "mfr_conv3x3_direct_f16x2_filter_pack")

    def rows(self, x, act=0):
        """token-major output [B, H, W, Cout] (mfr_conv3x3_direct_f16x2_rows): for the layer in front of a 1x1 / linear layer"""
        lib = _lib.load()
        x = x.contiguous()
        B, C, H, W = x.shape
        assert C == self.ci and x.dtype == torch.float32 and self.co % 4 == 0
        y = torch.empty(B, H, W, self.co, dtype=torch.float32, device=x.device)
        _lib.check(lib.mfr_conv3x3_direct_f16x2_rows(_lib.ptr(x), _lib.ptr(self.packed), _lib.ptr(self.b), B, C, self.co, H, W, int(act), _lib.ptr(y), self.co,
                                                     _lib.stream_ptr()), "mfr_conv3x3_direct_f16x2_rows")
        return y

    def strided(self, x, act=0):
        """the same filter at stride 2 (pad 1): y [B, Cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1] (mfr_conv3x3s2_direct_f16x2)"""
        lib = _lib.load()
        x = x.contiguous()
        B, C, H, W = x.shape
        assert C == self.ci and x.dtype == torch.float32
        y = torch.empty(B, self.co, (H - 1) // 2 + 1, (W - 1) // 2 + 1, dtype=torch.float32, device=x.device)
        _lib.check(lib.mfr_conv3x3s2_direct_f16x2(_lib.ptr(x), _lib.ptr(self.packed), _lib.ptr(self.b), B, C, self.co, H, W, int(act), _lib.ptr(y), _lib.stream_ptr()),
                   "mfr_conv3x3s2_direct_f16x2")
        return y

    def __call__(self, x, act=0, pool=False, residual=None):
        lib = _lib.load()
        x = x.contiguous()
        B, C, H, W = x.shape
        assert C == self.ci and x.dtype == torch.float32
        y = torch.empty((B, self.co, H // 2, W // 2) if pool else (B, self.co, H, W), dtype=torch.float32, device=x.device)
        res = _lib.ptr(residual.contiguous()) if residual is not None else None
        _lib.check(lib.mfr_conv3x3_direct_f16x2(_lib.ptr(x), _lib.ptr(self.packed), _lib.ptr(self.b), res, B, C, self.co, H, W, int(act), int(pool),
                                                _lib.ptr(y), _lib.stream_ptr()), "mfr_conv3x3_direct_f16x2")
        return y


class IgemmConv:
    """any other convolution of the backbones (strided 3x3, 1x1, 7x7) as ONE launch of the implicit-GEMM kernel (csrc/gemm_split.hip,
    mfr_conv_igemm_f16x2): the weight is laid out as the [Cout, K] matrix the kernel contracts over -- k = (dy KW + dx) * Cpad + ci with the input
    channels padded to a multiple of 32 (Cin == 1: k = dy KW + dx) -- and packed once with the linear layers' packer."""

    def __init__(self, weight, bias, stride=1, padding=None):
        lib = _lib.load(require_gpu=True)
        co, ci, kh, kw = (int(v) for v in weight.shape)
        self.co, self.ci, self.kh, self.kw = co, ci, kh, kw
        self.stride, self.pad = int(stride), (kh // 2 if padding is None else int(padding))
        K = lib.mfr_conv_igemm_k(ci, kh, kw)
        w = weight.detach().float()
        if ci == 1:
            m = torch.zeros(co, K, dtype=torch.float32, device=w.device)
            m[:, :kh * kw] = w.reshape(co, kh * kw)
        else:
            cpad = K // (kh * kw)
            m = torch.zeros(co, kh * kw, cpad, dtype=torch.float32, device=w.device)
            m[:, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
            m = m.reshape(co, K)
        nb = lib.mfr_gemm_f16x2_pack_bytes(co, K)
        self.packed = torch.empty(nb, dtype=torch.uint8, device=w.device)
        _lib.check(lib.mfr_gemm_f16x2_pack(_lib.ptr(m.contiguous()), co, K, _lib.ptr(self.packed), _lib.stream_ptr()), "mfr_gemm_f16x2_pack")
        self.b = None if bias is None else bias.contiguous().float()
        # round 6: the strided 3x3 layers through the direct halo-staged kernel (CONV_KERNEL 'auto' / 'direct'); the implicit-GEMM form stays for the A/B
        self.direct_s2 = DirectConv3x3(w, self.b) if (kh, kw, self.stride, self.pad) == (3, 3, 2, 1) and ci > 1 else None

    def __call__(self, x, relu=False, up_add=None):
        if self.direct_s2 is not None and up_add is None and options.get("CONV_KERNEL") in ("auto", "direct"):
            return self.direct_s2.strided(x, act=1 if relu else 0)
        """up_add [B, Cout, Ho / 2, Wo / 2]: y = conv(x) + bias + its 2x bilinear up-sampling (align_corners=True) in the same launch (LoFTR's FPN merge)"""
        lib = _lib.load()
        x = x.contiguous()
        B, C, H, W = x.shape
        assert C == self.ci and x.dtype == torch.float32
        Ho, Wo = (H + 2 * self.pad - self.kh) // self.stride + 1, (W + 2 * self.pad - self.kw) // self.stride + 1
        y = torch.empty(B, self.co, Ho, Wo, dtype=torch.float32, device=x.device)
        if up_add is not None:
            lo = up_add.contiguous()
            assert not relu and lo.shape == (B, self.co, Ho // 2, Wo // 2) and Ho % 2 == 0 and Wo % 2 == 0 and lo.dtype == torch.float32
            _lib.check(lib.mfr_conv_igemm_f16x2_upadd(_lib.ptr(x), _lib.ptr(self.packed), _lib.ptr(self.b), _lib.ptr(lo), Ho // 2, Wo // 2, _lib.ptr(y), B, C, H, W,
                                                      self.co, self.kh, self.kw, self.stride, self.pad, _lib.stream_ptr()), "mfr_conv_igemm_f16x2_upadd")
            return y
        _lib.check(lib.mfr_conv_igemm_f16x2(_lib.ptr(x), _lib.ptr(self.packed), _lib.ptr(self.b), _lib.ptr(y), B, C, H, W, self.co, self.kh, self.kw,
                                            self.stride, self.pad, 1 if relu else 0, _lib.stream_ptr()), "mfr_conv_igemm_f16x2")
        return y
