"""Image encoders of the regression model: pre-activation residual units, the strided ResNet and the ResUNet
(reference: lib/models/regression/encoder/{preact.py:13-64, resnet.py:7-37, resunet.py:16-128}).

Attribute names (bn1/conv1/.../shortcut.0, firstconv/firstbn, encoder1..3, upconv4.conv1.{conv,normalize}, iconv4, ...)
are the reference's, so its checkpoints' `encoder.*` keys load unchanged.  The decoder's four 3x3 convolutions (1024->512 twice at
1/16, 512->256 twice at 1/8 of the input: 85 % of the encoder's flops) run, under bf16 autocast on the GPU, as implicit GEMMs on the bf16
matrix cores (regression/conv_bf16.py, csrc/conv_gemm_bf16.hip: forward, d input, d weight); the strided / 7x7 / 1x1 / small-channel
convolutions stay library calls (MIOpen through torch)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import conv_bf16


class ViewBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d whose TRAINING statistics are taken per VIEW when the batch holds several views back to back
    (`view_groups(2)`: images [0, B) are the reference views, [B, 2B) the queries).  The reference encodes the two images of a pair in
    two encoder calls (lib/models/regression/model.py:64-66), i.e. BatchNorm normalises each view's batch with its own statistics and
    updates the running estimates twice, view 0 first.  Running both views through the encoder in ONE pass (TRAINING.SIAMESE_BATCH:
    half the launches, convolutions at twice the batch) keeps exactly that arithmetic when every BatchNorm does the same: per-view
    statistics, two running updates in the same order.  Parameters, buffers and state-dict keys are nn.BatchNorm2d's."""
    groups = 1                                              # class-wide switch, set by `view_groups`

    def forward(self, x):
        g = ViewBatchNorm2d.groups
        if g <= 1 or not self.training or x.shape[0] % g:
            return super().forward(x)
        n = x.shape[0] // g
        return torch.cat([super(ViewBatchNorm2d, self).forward(x[k * n:(k + 1) * n]) for k in range(g)], 0)


class view_groups:
    """context manager: BatchNorm layers of this module treat the batch as `g` consecutive views (see ViewBatchNorm2d)"""

    def __init__(self, g):
        self.g = int(g)

    def __enter__(self):
        self.prev, ViewBatchNorm2d.groups = ViewBatchNorm2d.groups, self.g

    def __exit__(self, *exc):
        ViewBatchNorm2d.groups = self.prev
        return False


def _norm(planes, enabled=True):
    return ViewBatchNorm2d(planes) if enabled else nn.Identity()


class PreActUnit(nn.Module):
    """BN-ReLU-conv residual unit (He et al., identity mappings), basic (two 3x3) or bottleneck (1x1, 3x3, 1x1 with
    4x expansion).  The projection shortcut, when shapes change, is applied to the ACTIVATED input (preact.py:31-32)."""

    def __init__(self, in_planes, planes, stride=1, bn=True, bottleneck=False):
        super().__init__()
        self.expansion = 4 if bottleneck else 1
        out_planes = planes * self.expansion
        if bottleneck:
            shapes = [(in_planes, planes, 1, 1), (planes, planes, 3, stride), (planes, out_planes, 1, 1)]
        else:
            shapes = [(in_planes, planes, 3, stride), (planes, planes, 3, 1)]
        for i, (cin, cout, k, s) in enumerate(shapes, 1):
            setattr(self, f"bn{i}", _norm(cin, bn))
            setattr(self, f"conv{i}", nn.Conv2d(cin, cout, k, stride=s, padding=k // 2, bias=False))
        self.n_convs = len(shapes)
        if stride != 1 or in_planes != out_planes:
            self.shortcut = nn.Sequential(nn.Conv2d(in_planes, out_planes, 1, stride=stride, bias=False))
        else:
            self.shortcut = None

    def forward(self, x):
        a = F.relu(self.bn1(x))
        skip = x if self.shortcut is None else self.shortcut(a)
        y = self.conv1(a)
        for i in range(2, self.n_convs + 1):
            y = getattr(self, f"conv{i}")(F.relu(getattr(self, f"bn{i}")(y)))
        return y + skip


class PreActBlock(PreActUnit):
    expansion_factor = 1

    def __init__(self, in_planes, planes, stride=1, bn=True):
        super().__init__(in_planes, planes, stride, bn, bottleneck=False)


class PreActBottleneck(PreActUnit):
    expansion_factor = 4

    def __init__(self, in_planes, planes, stride=1):
        super().__init__(in_planes, planes, stride, True, bottleneck=True)


_BLOCKS = (PreActBlock, PreActBottleneck)


def _stage(block, in_planes, planes, count, stride):
    units = []
    for i in range(count):
        units.append(block(in_planes, planes, stride if i == 0 else 1))
        in_planes = planes * block.expansion_factor
    return nn.Sequential(*units), in_planes


def _block_counts(cfg):
    return [int(n) for n in str(cfg.NUM_BLOCKS).strip().split("-")]


class ResNet(nn.Module):
    """7x7/2 stem then three stages, each followed by a 2x2 average pool (resnet.py:7-37): 1/16 resolution."""

    def __init__(self, cfg):
        super().__init__()
        block, counts = _BLOCKS[cfg.BLOCK_TYPE], _block_counts(cfg)
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=1, bias=False)
        self.layer1, c = _stage(block, 64, 64, counts[0], 1)
        self.layer2, c = _stage(block, c, 128, counts[1], 2)
        self.layer3, c = _stage(block, c, 256, counts[2], 2)
        self.num_out_layers = c

    def forward(self, x):
        x = self.conv1(x)
        for stage in (self.layer1, self.layer2, self.layer3):
            x = F.avg_pool2d(stage(x), 2)
        return x


class conv(nn.Module):
    """conv + BatchNorm + ELU (resunet.py:16-27); class and attribute names are state-dict keys"""

    def __init__(self, cin, cout, kernel_size, stride):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=(kernel_size - 1) // 2)
        self.normalize = ViewBatchNorm2d(cout)

    def forward(self, x):
        c = self.conv
        if (x.is_cuda and c.kernel_size == (3, 3) and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16
                and conv_bf16.supported(x, c.weight, c.stride, c.padding)):
            # the decoder's 3x3 convolutions under bf16 autocast: implicit GEMMs on the bf16 matrix cores (csrc/conv_gemm_bf16.hip)
            y = conv_bf16.conv3x3_bf16(x, c.weight, c.bias)
        else:
            y = c(x)
        return F.elu(self.normalize(y))


class _UpsampleAC(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) for device tensors through csrc/loftr_fused.hip
    (mfr_upsample_bilinear: same arithmetic, all (plane, row) pairs in flight instead of one thread per output pixel);
    the backward is ATen's own scatter kernel."""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        from .._lib import check, load, ptr, stream_ptr
        lib = load(require_gpu=True)
        x = x.contiguous()
        B, C, H, W = x.shape
        out = torch.empty(B, C, Ho, Wo, dtype=x.dtype, device=x.device)
        check(lib.mfr_upsample_bilinear(ptr(x), ptr(out), B * C, H, W, Ho, Wo, 0 if x.dtype == torch.float32 else 1, stream_ptr()),
              "mfr_upsample_bilinear")
        ctx.in_shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.in_shape
        gi = torch.ops.aten.upsample_bilinear2d_backward(g.contiguous(), [g.shape[2], g.shape[3]], [B, C, H, W], True, None, None)
        return gi, None, None


def upsample_bilinear_ac(x, scale):
    """bilinear, align_corners=True, output size floor(in * scale) like F.interpolate(scale_factor=scale)"""
    Ho, Wo = int(x.shape[2] * scale), int(x.shape[3] * scale)
    if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16):
        return _UpsampleAC.apply(x, Ho, Wo)
    return F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=True)


class upconv(nn.Module):
    """bilinear (align_corners) upsampling by `scale`, then conv+BN+ELU (resunet.py:30-38)"""

    def __init__(self, cin, cout, kernel_size, scale):
        super().__init__()
        self.scale = scale
        self.conv1 = conv(cin, cout, kernel_size, 1)

    def forward(self, x):
        return self.conv1(upsample_bilinear_ac(x, self.scale))


def _centre_pad_to(x, ref):
    """zero-pad x (odd sizes after the strided stages) to ref's spatial size, surplus on the bottom/right (resunet.py:93-100)"""
    dy, dx = ref.shape[2] - x.shape[2], ref.shape[3] - x.shape[3]
    if dy == 0 and dx == 0:
        return x
    return F.pad(x, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))


class ResUNet(nn.Module):
    """ResNet stem + three stages down to 1/16, two upsample-and-merge steps back to 1/4, 1x1 projection to
    NUM_OUT_LAYERS channels (resunet.py:41-128).  On the Map-free 360x270 input: 92x68 = 6256 positions x 32 channels."""

    def __init__(self, cfg, num_in_layers=3):
        super().__init__()
        block, counts = _BLOCKS[cfg.BLOCK_TYPE], _block_counts(cfg)
        self.firstconv = nn.Conv2d(num_in_layers, 64, 7, stride=2, padding=3, bias=False)
        self.firstbn = ViewBatchNorm2d(64)
        self.firstmaxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.encoder1, c1 = _stage(block, 64, 64, counts[0], 1)
        self.encoder2, c2 = _stage(block, c1, 128, counts[1], 2)
        self.encoder3, c3 = _stage(block, c2, 256, counts[2], 2)
        # decoder widths are fixed by the reference to the bottleneck layout (256/512/1024)
        skip = [256, 512, 1024]
        self.not_concat = bool(getattr(cfg, "NOT_CONCAT", False))
        self.upconv4 = upconv(skip[2], 512, 3, 2)
        self.iconv4 = conv(512 if self.not_concat else skip[1] + 512, 512, 3, 1)
        self.upconv3 = upconv(512, 256, 3, 2)
        self.iconv3 = conv(256 if self.not_concat else skip[0] + 256, 256, 3, 1)
        self.num_out_layers = int(getattr(cfg, "NUM_OUT_LAYERS", None) or 128)
        self.outconv = conv(256, self.num_out_layers, 1, 1)

    def forward(self, x):
        from .. import options
        nhwc = bool(options.get("RPR_ENCODER_NHWC")) and x.is_cuda
        if nhwc:
            # the strided stages on channels-last ACTIVATIONS: MIOpen's convolutions are NHWC kernels and transposed every NCHW operand in
            # and out (236 transposes, 2.8 ms of a 26.7 ms training step).  The weights keep their layout (the fused optimizer wants
            # parameters, gradients and moments in one layout; their transposes are small).
            x = x.contiguous(memory_format=torch.channels_last)
        x1 = self.firstmaxpool(F.relu(self.firstbn(self.firstconv(x))))
        x2 = self.encoder1(x1)
        x3 = self.encoder2(x2)
        x4 = self.encoder3(x3)
        if nhwc:                                            # the decoder's own kernels (upsampling, implicit-GEMM convolutions) take NCHW
            x2, x3, x4 = x2.contiguous(), x3.contiguous(), x4.contiguous()
        y = self.upconv4(x4)
        if not self.not_concat:
            y = torch.cat([y, _centre_pad_to(x3, y)], dim=1)
        y = self.upconv3(self.iconv4(y))
        if not self.not_concat:
            y = torch.cat([y, _centre_pad_to(x2, y)], dim=1)
        return self.outconv(self.iconv3(y))


ENCODERS = {"ResNet": ResNet, "ResUNet": ResUNet}
