"""3x3 / stride-1 / padding-1 convolutions of the regression encoder's decoder (`conv` inside upconv4 / iconv4 / upconv3 / iconv3,
lib/models/regression/encoder/resunet.py:16-38, 112-128) under bf16 autocast, as implicit GEMMs on the bf16 matrix cores:
forward, gradient w.r.t. the input and gradient w.r.t. the weights all run csrc/conv_gemm_bf16.hip (`mfr_conv_gemm_bf16`), a
"segmented" NT product  C[i, j] = sum_k A[i, k] B[j, k]  with bf16 operands and fp32 accumulation.  Rounds 1-2 ran these layers
through MIOpen (7 % of the bf16 peak, 40 % of the training step).

How the three products are laid out (everything below is index arithmetic + a few memory-bound copies; the flops are in the kernel):

forward / d input   The input goes into a zero-haloed NHWC image [B, H+2, W+2, C], flattened to rows of C channels.  A filter tap
    (ky, kx) is then a CONSTANT row shift (ky-1)(W+2) + (kx-1) of that matrix, so the k axis = 9 segments of C channels, segment `tap`
    read at row offset shift * C, against the weights laid out [Cout][tap][Cin].  Results are produced for every haloed position
    (5-10 % more rows than pixels) and the interior is handed on -- in exchange no address needs a bounds test and there is no
    im2col buffer.  d input = the same product with the haloed output gradient as the image and the 180-degree rotated, transposed
    weights [Cin][tap'][Cout].
d weight   k = pixels.  Both operands are channel-major zero-haloed images [B][C][L] (row stride rounded up to 8 pixels so that the ky
    shift keeps the 16-byte alignment of the loads; the kx shift is applied when the three shifted copies of the input are written;
    what a shifted read picks up from a neighbouring row / channel / image is multiplied by the zero halo of the gradient image).
    One grid slice per (tap, K split); fp32 partial sums, added up in a fixed order: no atomics, bit-reproducible.
"""
import torch
import torch.nn.functional as F

from .. import options

# options RPR_CONV = "miopen" keeps the library convolution (A/B timing); default: the kernel of this package.  The module-level names
# are read at CALL time (tests and tools may also set them directly: None = follow the declared option)
ENABLED = None
# Backward: "hip" = the d input / d weight products of this file; "lib" = torch's convolution_backward (MIOpen).  Measured on MI355X
# (tools/bench_conv_bf16.py, profiles/r03_ab_conv_bf16*.json): the forward product beats the library 1.4-1.6x at 20 images per call and
# 2.7-3.6x at 10 (0.25-0.33 vs 0.90 ms), the backward pair does not (1.37-1.43 vs 1.09-1.23 ms at 20 images: the three shifted
# channel-major copies of the d weight product cost as much as its matrix work) -- so the default pairs the own forward with the
# library's backward; options RPR_CONV_BWD = "hip" runs everything here (what the parity tests do).
BACKWARD = None


def _enabled():
    return (options.get("RPR_CONV") != "miopen") if ENABLED is None else bool(ENABLED)


def _backward():
    return options.get("RPR_CONV_BWD") if BACKWARD is None else BACKWARD
_TABLES = {}


def _ceil(v, m):
    return (v + m - 1) // m * m


def _table(key, dev, make):
    k = (key, str(dev))
    if k not in _TABLES:
        vals, dtype = make()
        _TABLES[k] = torch.tensor(vals, dtype=dtype, device=dev)
    return _TABLES[k]


def seg_gemm(A, a_off, sA, segA, Bm, b_off, sB, segB, Lk, nkc_total, nkc_z, bias, C, ldc, M, N, nz=1, zA=None, zB=None, zC=None, zk=None):
    """C[i, j] (+ slices) = sum_k A[i, k] B[j, k]; see include/mfr_hip.h (mfr_conv_gemm_bf16).  A, Bm: flat bf16 tensors, a_off / b_off:
    element offsets of their logical origins; C: flat bf16 or f32 tensor.  Device tensors only -- there is no CPU path."""
    from .._lib import check, load, ptr, stream_ptr
    lib = load(require_gpu=True)
    p = lambda t: ptr(t) if t is not None else None
    check(lib.mfr_conv_gemm_bf16(A.data_ptr() + 2 * a_off, sA, ptr(segA), Bm.data_ptr() + 2 * b_off, sB, ptr(segB), Lk, nkc_total, nkc_z,
                                 p(bias), ptr(C), ldc, 1 if C.dtype == torch.bfloat16 else 0, M, N, nz, p(zA), p(zB), p(zC), p(zk),
                                 stream_ptr()), "mfr_conv_gemm_bf16")


def _src(x):
    x = x.contiguous()
    return x if x.dtype in (torch.float32, torch.bfloat16) else x.float()


def pack_nhwc_halo(x, guard_rows, Wp):
    """x [B, C, H, W] -> flat bf16  guard_rows * C zeros | haloed NHWC image [B, H+2, Wp, C] | guard_rows * C zeros  (one kernel,
    transposing through LDS; csrc/conv_gemm_bf16.hip).  Wp = W + 1: one zero column in front of every row, shared with the row before"""
    from .._lib import check, load, ptr, stream_ptr
    lib = load(require_gpu=True)
    x = _src(x)
    B, C, H, W = x.shape
    out = torch.empty((2 * guard_rows + B * (H + 2) * Wp) * C, dtype=torch.bfloat16, device=x.device)
    check(lib.mfr_conv_pack_nhwc_halo(ptr(x), 0 if x.dtype == torch.float32 else 1, B, C, H, W, Wp, ptr(out), guard_rows, stream_ptr()), "mfr_conv_pack_nhwc_halo")
    return out


def pack_cm_halo(x, Wq, L, ncopies, first_shift, slack):
    """x [B, C, H, W] -> [ncopies, slack + B C L + slack] bf16: channel-major haloed images (rows of Wq positions), copy k shifted by
    first_shift + k positions along its rows"""
    from .._lib import check, load, ptr, stream_ptr
    lib = load(require_gpu=True)
    x = _src(x)
    B, C, H, W = x.shape
    out = torch.empty(ncopies, 2 * slack + B * C * L, dtype=torch.bfloat16, device=x.device)
    check(lib.mfr_conv_pack_cm_halo(ptr(x), 0 if x.dtype == torch.float32 else 1, B * C, H, W, Wq, L, ncopies, first_shift, slack, ptr(out), stream_ptr()),
          "mfr_conv_pack_cm_halo")
    return out


TAP_ORDER = None                                           # None = options RPR_CONV_ORDER


def tap_tables(C, Wp, dev, order=None):
    """segment tables of the forward / d input product: k = (tap, channel).  'tap_outer': 9 segments of C channels (all channels of
    tap 0, then tap 1, ...).  'tap_inner': segments of 32 channels, the 9 taps of one channel chunk back to back -- the rows a workgroup
    reads for the 9 taps of a chunk are the same rows shifted by at most W+3, so eight of the nine reads hit the CU's L1 instead of L2."""
    order = order or TAP_ORDER or options.get("RPR_CONV_ORDER")
    shift = lambda t: ((t // 3 - 1) * Wp + (t % 3 - 1)) * C
    if order == "tap_outer":
        segA = _table(("fwdA", C, Wp), dev, lambda: ([shift(t) for t in range(9)] + [0], torch.int64))
        segB = _table(("fwdB", C), dev, lambda: ([t * C for t in range(10)], torch.int64))
        return segA, segB, C
    segA = _table(("fwdAi", C, Wp), dev, lambda: ([shift(t) + 32 * cc for cc in range(C // 32) for t in range(9)] + [0], torch.int64))
    segB = _table(("fwdBi", C), dev, lambda: ([t * C + 32 * cc for cc in range(C // 32) for t in range(9)] + [0], torch.int64))
    return segA, segB, 32


def unpack_nchw(haloed, B, N, H, W, Wp):
    """haloed NHWC result [B, H+2, Wp, N] bf16 -> contiguous [B, N, H, W] bf16 (one transposing kernel)"""
    from .._lib import check, load, ptr, stream_ptr
    lib = load(require_gpu=True)
    y = torch.empty(B, N, H, W, dtype=torch.bfloat16, device=haloed.device)
    check(lib.mfr_conv_unpack_nchw(ptr(haloed), B, N, H, W, Wp, ptr(y), stream_ptr()), "mfr_conv_unpack_nchw")
    return y


def _conv_haloed(x, wmat, bias):
    """x [B, C, H, W] (f32 / bf16) * wmat [N, 9 C] bf16 (k = tap * C + c) -> y [B, N, H, W] bf16, contiguous"""
    B, C, H, W = x.shape
    N = wmat.shape[0]
    # ONE zero column per image row (it is the left neighbour of its row's first pixel and the right neighbour of the previous row's
    # last one): 1.5-3 % fewer rows than a halo on both sides -- and 508 instead of 516 workgroups for the 92x68 layers, which is the
    # difference between one and two rounds on 256 CUs x 2 workgroups
    Hp, Wp = H + 2, W + 1
    Mp = B * Hp * Wp
    G = _ceil(Wp + 1, 8)                                    # guard rows before / after: the taps of the first / last haloed rows
    xp = pack_nhwc_halo(x, G, Wp)
    segA, segB, Lk = tap_tables(C, Wp, x.device)
    out = torch.empty(Mp, N, dtype=torch.bfloat16, device=x.device)
    seg_gemm(xp, G * C, C, segA, wmat, 0, 9 * C, segB, Lk, 9 * C // 32, 9 * C // 32, bias, out, N, Mp, N)
    return unpack_nchw(out, B, N, H, W, Wp)


def _wgrad(x, gy, splits=None):
    """sum over pixels of x [B, C, H, W] (shifted by the tap) times gy [B, N, H, W] -> dW [N, C, 3, 3] f32"""
    B, C, H, W = x.shape
    N = gy.shape[1]
    Hp, Wq = H + 2, _ceil(W + 2, 8)
    L = _ceil(Hp * Wq, 32)
    dev = x.device
    n_el = B * C * L
    slack = _ceil(Wq + 8, 8)
    stride = 2 * slack + n_el
    buf = pack_cm_halo(x, Wq, L, 3, -1, slack)              # copy kx holds x_haloed[p + kx - 1] at position p
    gq = pack_cm_halo(gy, Wq, L, 1, 0, 0)
    tiles = ((C + 255) // 256) * ((N + 127) // 128) * 9
    S = int(splits or max(1, min(int(options.get('RPR_WGRAD_SPLITS')), 512 // tiles)))          # all slices resident at once: 256 CUs x 2 workgroups
    nkc_total = B * L // 32
    nkc_z = -(-nkc_total // S)
    S = -(-nkc_total // nkc_z)
    key = (C, N, L, Wq, B, S, nkc_z, stride, slack)
    zA = _table(("wgA",) + key, dev, lambda: ([(t % 3) * stride + slack + (t // 3 - 1) * Wq for t in range(9) for s in range(S)], torch.int64))
    zC = _table(("wgC",) + key, dev, lambda: ([z * C * N for z in range(9 * S)], torch.int64))
    zk = _table(("wgK",) + key, dev, lambda: ([s * nkc_z for t in range(9) for s in range(S)], torch.int32))
    segA = _table(("wgSA",) + key, dev, lambda: ([b * C * L for b in range(B + 1)], torch.int64))
    segB = _table(("wgSB",) + key, dev, lambda: ([b * N * L for b in range(B + 1)], torch.int64))
    part = torch.empty(9 * S, C, N, dtype=torch.float32, device=dev)
    seg_gemm(buf.view(-1), 0, L, segA, gq.view(-1), 0, L, segB, L, nkc_total, nkc_z, None, part.view(-1), N, C, N, nz=9 * S, zA=zA, zC=zC, zk=zk)
    dw = part.view(9, S, C, N).sum(1) if S > 1 else part.view(9, C, N)
    return dw.view(3, 3, C, N).permute(3, 2, 0, 1)


def _require_library():
    """the HIP library must load when a device tensor arrives: a missing / unloadable libmfr_hip.so raises here (round 4 swallowed the
    error -- and with it a NameError of this very function -- and silently trained through the library convolution)"""
    from .._lib import load
    load(require_gpu=True)
    return True


def supported(x, weight, stride=1, padding=1):
    N, C, kh, kw = weight.shape
    return (_enabled() and x.is_cuda and kh == 3 and kw == 3 and stride in (1, (1, 1)) and padding in (1, (1, 1))
            and C % 32 == 0 and N % 32 == 0 and x.dim() == 4 and _require_library())


class _Conv3x3BF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        N, C = weight.shape[:2]
        wmat = weight.detach().permute(0, 2, 3, 1).reshape(N, 9 * C).to(torch.bfloat16).contiguous()
        b32 = bias.detach().float().contiguous() if bias is not None else None
        y = _conv_haloed(x.detach(), wmat, b32)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.bias_dtype = bias.dtype if bias is not None else None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        N, C = weight.shape[:2]
        gx = gw = gb = None
        if _backward() == "lib":
            mask = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], bool(ctx.has_bias and ctx.needs_input_grad[2])]
            xb, wb, gyb = x.detach().to(torch.bfloat16), weight.detach().to(torch.bfloat16), gy.to(torch.bfloat16).contiguous()
            gx, gw, gb = torch.ops.aten.convolution_backward(gyb, xb, wb, [N] if ctx.has_bias else None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, mask)
            return (gx.to(x.dtype) if gx is not None else None, gw.to(weight.dtype) if gw is not None else None,
                    gb.to(ctx.bias_dtype) if gb is not None else None)
        if ctx.needs_input_grad[0]:
            wflip = weight.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(C, 9 * N).to(torch.bfloat16).contiguous()
            gx = _conv_haloed(gy, wflip, None).to(x.dtype)
        if ctx.needs_input_grad[1]:
            gw = _wgrad(x.detach(), gy).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.float().sum((0, 2, 3)).to(ctx.bias_dtype)
        return gx, gw, gb


def conv3x3_bf16(x, weight, bias=None):
    """F.conv2d(x, weight, bias, stride=1, padding=1) as autocast(bfloat16) computes it: bf16 operands, fp32 accumulation, bf16 result;
    differentiable"""
    return _Conv3x3BF16.apply(x, weight, bias)
