"""CPU ORACLE (TEST INFRASTRUCTURE ONLY) -- plain PyTorch-fp32 restatement of LoFTR as the reference
runs it: LoFTR_matcher (etc/feature_matching_baselines/matchers.py:12-59) instantiates
`LoFTR(config=default_cfg)` from the un-vendored submodule etc/feature_matching_baselines/LoFTR
(zju3dv/LoFTR; empty offline, version hint lib/datasets/sampler.py:1) and loads `*_ot.ckpt` with
strict=False (:16-18).  `default_cfg` = dual-softmax coarse matching (thr 0.2, border 2, temperature
0.1), linear attention, ResNetFPN_8_2 (128; 128/196/256), coarse d256 x 8 heads x 4 [self,cross],
fine d128 x 8 heads x 1 [self,cross], window 5, TEMP_BUG_FIX False.

Restated from the PUBLISHED architecture with upstream parameter names so the real checkpoints load;
parity vs the upstream source / weights is UNPINNED (neither is available offline).  Sub-steps that
HuggingFace's EfficientLoFTR shares with LoFTR (dual-softmax, mutual-max mask, border mask,
spatial expectation) are cross-checked in tests/test_nets_ref_vs_hf.py.

Nothing in the product package imports this module.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def conv1x1(i, o, stride=1):
    return nn.Conv2d(i, o, kernel_size=1, stride=stride, padding=0, bias=False)


def conv3x3(i, o, stride=1):
    return nn.Conv2d(i, o, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = conv3x3(in_planes, planes, stride)
        self.conv2 = conv3x3(planes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None if stride == 1 else nn.Sequential(conv1x1(in_planes, planes, stride=stride), nn.BatchNorm2d(planes))

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


class ResNetFPN_8_2(nn.Module):
    def __init__(self, initial_dim=128, block_dims=(128, 196, 256)):
        super().__init__()
        self.in_planes = initial_dim
        self.conv1 = nn.Conv2d(1, initial_dim, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(initial_dim)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(block_dims[0], 1)
        self.layer2 = self._make_layer(block_dims[1], 2)
        self.layer3 = self._make_layer(block_dims[2], 2)
        self.layer3_outconv = conv1x1(block_dims[2], block_dims[2])
        self.layer2_outconv = conv1x1(block_dims[1], block_dims[2])
        self.layer2_outconv2 = nn.Sequential(conv3x3(block_dims[2], block_dims[2]), nn.BatchNorm2d(block_dims[2]),
                                             nn.LeakyReLU(), conv3x3(block_dims[2], block_dims[1]))
        self.layer1_outconv = conv1x1(block_dims[0], block_dims[1])
        self.layer1_outconv2 = nn.Sequential(conv3x3(block_dims[1], block_dims[1]), nn.BatchNorm2d(block_dims[1]),
                                             nn.LeakyReLU(), conv3x3(block_dims[1], block_dims[0]))

    def _make_layer(self, dim, stride):
        l1 = BasicBlock(self.in_planes, dim, stride=stride)
        l2 = BasicBlock(dim, dim, stride=1)
        self.in_planes = dim
        return nn.Sequential(l1, l2)

    def forward(self, x):
        x0 = self.relu(self.bn1(self.conv1(x)))
        x1 = self.layer1(x0)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x3_out = self.layer3_outconv(x3)
        x3_out_2x = F.interpolate(x3_out, scale_factor=2., mode='bilinear', align_corners=True)
        x2_out = self.layer2_outconv(x2)
        x2_out = self.layer2_outconv2(x2_out + x3_out_2x)
        x2_out_2x = F.interpolate(x2_out, scale_factor=2., mode='bilinear', align_corners=True)
        x1_out = self.layer1_outconv(x1)
        x1_out = self.layer1_outconv2(x1_out + x2_out_2x)
        return [x3_out, x1_out]


def position_encoding_sine(d_model, H, W, temp_bug_fix=False):
    """upstream PositionEncodingSine; temp_bug_fix=False keeps the operator-precedence quirk of the
    released checkpoints: (-log(1e4) / d_model) // 2 == -1.0"""
    pe = torch.zeros((d_model, H, W))
    y_position = torch.ones((H, W)).cumsum(0).float().unsqueeze(0)
    x_position = torch.ones((H, W)).cumsum(1).float().unsqueeze(0)
    if temp_bug_fix:
        div_term = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / (d_model // 2)))
    else:
        div_term = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / d_model // 2))
    div_term = div_term[:, None, None]
    pe[0::4, :, :] = torch.sin(x_position * div_term)
    pe[1::4, :, :] = torch.cos(x_position * div_term)
    pe[2::4, :, :] = torch.sin(y_position * div_term)
    pe[3::4, :, :] = torch.cos(y_position * div_term)
    return pe


def elu_feature_map(x):
    return F.elu(x) + 1


def linear_attention(queries, keys, values, eps=1e-6):
    """upstream LinearAttention.forward ([N,L,H,D]), no masks (eval, unpadded)"""
    Q = elu_feature_map(queries)
    K = elu_feature_map(keys)
    v_length = values.size(1)
    values = values / v_length
    KV = torch.einsum("nshd,nshv->nhdv", K, values)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
    return (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * v_length).contiguous()


class LoFTREncoderLayer(nn.Module):
    def __init__(self, d_model, nhead):
        super().__init__()
        self.dim = d_model // nhead
        self.nhead = nhead
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.mlp = nn.Sequential(nn.Linear(d_model * 2, d_model * 2, bias=False), nn.ReLU(True),
                                 nn.Linear(d_model * 2, d_model, bias=False))
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, x, source):
        bs = x.size(0)
        query = self.q_proj(x).view(bs, -1, self.nhead, self.dim)
        key = self.k_proj(source).view(bs, -1, self.nhead, self.dim)
        value = self.v_proj(source).view(bs, -1, self.nhead, self.dim)
        message = linear_attention(query, key, value)
        message = self.merge(message.view(bs, -1, self.nhead * self.dim))
        message = self.norm1(message)
        message = self.mlp(torch.cat([x, message], dim=2))
        message = self.norm2(message)
        return x + message


class LocalFeatureTransformer(nn.Module):
    def __init__(self, d_model, nhead, layer_names):
        super().__init__()
        self.layer_names = layer_names
        self.layers = nn.ModuleList([LoFTREncoderLayer(d_model, nhead) for _ in layer_names])

    def forward(self, feat0, feat1):
        for layer, name in zip(self.layers, self.layer_names):
            if name == 'self':
                feat0 = layer(feat0, feat0)
                feat1 = layer(feat1, feat1)
            else:                       # upstream: feat1's cross-attention sees the UPDATED feat0
                feat0 = layer(feat0, feat1)
                feat1 = layer(feat1, feat0)
        return feat0, feat1


def mask_border(m, b, v=False):
    m[:, :b] = v; m[:, :, :b] = v; m[:, :, :, :b] = v; m[:, :, :, :, :b] = v
    m[:, -b:] = v; m[:, :, -b:] = v; m[:, :, :, -b:] = v; m[:, :, :, :, -b:] = v


def coarse_matching(feat_c0, feat_c1, hw0, hw1, thr=0.2, border_rm=2, temperature=0.1, scale=8):
    """upstream CoarseMatching (dual_softmax) + get_coarse_match, eval mode"""
    feat_c0, feat_c1 = feat_c0 / feat_c0.shape[-1] ** .5, feat_c1 / feat_c1.shape[-1] ** .5
    sim = torch.einsum("nlc,nsc->nls", feat_c0, feat_c1) / temperature
    conf = F.softmax(sim, 1) * F.softmax(sim, 2)
    mask = conf > thr
    n = conf.shape[0]
    mask = mask.view(n, hw0[0], hw0[1], hw1[0], hw1[1]).clone()
    mask_border(mask, border_rm, False)
    mask = mask.view(n, hw0[0] * hw0[1], hw1[0] * hw1[1])
    mask = mask * (conf == conf.max(dim=2, keepdim=True)[0]) * (conf == conf.max(dim=1, keepdim=True)[0])
    mask_v, all_j_ids = mask.max(dim=2)
    b_ids, i_ids = torch.where(mask_v)
    j_ids = all_j_ids[b_ids, i_ids]
    mconf = conf[b_ids, i_ids, j_ids]
    mkpts0_c = torch.stack([i_ids % hw0[1], i_ids // hw0[1]], dim=1) * scale
    mkpts1_c = torch.stack([j_ids % hw1[1], j_ids // hw1[1]], dim=1) * scale
    return dict(b_ids=b_ids, i_ids=i_ids, j_ids=j_ids, mconf=mconf, mkpts0_c=mkpts0_c.float(), mkpts1_c=mkpts1_c.float(),
                conf_matrix=conf)


def spatial_expectation2d(heatmap):
    """kornia.geometry.subpix.dsnt.spatial_expectation2d(normalized_coordinates=True): [M,W,W] -> [M,2] (x,y)"""
    M, Wh, Ww = heatmap.shape
    xs = torch.linspace(-1, 1, Ww); ys = torch.linspace(-1, 1, Wh)
    ex = (heatmap * xs[None, None, :]).sum((1, 2))
    ey = (heatmap * ys[None, :, None]).sum((1, 2))
    return torch.stack([ex, ey], 1)


class LoFTRRef(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = ResNetFPN_8_2()
        self.loftr_coarse = LocalFeatureTransformer(256, 8, ['self', 'cross'] * 4)
        self.fine_preprocess = nn.Module()
        self.fine_preprocess.down_proj = nn.Linear(256, 128, bias=True)
        self.fine_preprocess.merge_feat = nn.Linear(2 * 128, 128, bias=True)
        self.loftr_fine = LocalFeatureTransformer(128, 8, ['self', 'cross'])
        self.W = 5

    @torch.no_grad()
    def forward(self, image0, image1):
        """[1,1,H,W] x2 (H, W multiples of 8) -> dict(mkpts0_f, mkpts1_f [M,2], mconf [M], ...)"""
        N = image0.shape[0]
        if image0.shape == image1.shape:
            fc, ff = self.backbone(torch.cat([image0, image1], 0))
            (feat_c0, feat_c1), (feat_f0, feat_f1) = fc.split(N), ff.split(N)
        else:
            (feat_c0, feat_f0), (feat_c1, feat_f1) = self.backbone(image0), self.backbone(image1)
        hw0_c, hw1_c, hw0_f, hw1_f = feat_c0.shape[2:], feat_c1.shape[2:], feat_f0.shape[2:], feat_f1.shape[2:]
        pe0 = position_encoding_sine(256, *hw0_c); pe1 = position_encoding_sine(256, *hw1_c)
        feat_c0 = (feat_c0 + pe0[None]).flatten(2).transpose(1, 2)          # 'n c h w -> n (h w) c'
        feat_c1 = (feat_c1 + pe1[None]).flatten(2).transpose(1, 2)
        feat_c0, feat_c1 = self.loftr_coarse(feat_c0, feat_c1)
        cm = coarse_matching(feat_c0, feat_c1, tuple(hw0_c), tuple(hw1_c), scale=image0.shape[2] // hw0_c[0])
        b_ids, i_ids, j_ids = cm["b_ids"], cm["i_ids"], cm["j_ids"]
        M = len(b_ids)
        W = self.W
        out = dict(cm)
        if M == 0:
            out.update(mkpts0_f=cm["mkpts0_c"], mkpts1_f=cm["mkpts1_c"])
            return out
        stride = hw0_f[0] // hw0_c[0]
        f0u = F.unfold(feat_f0, kernel_size=(W, W), stride=stride, padding=W // 2)
        f1u = F.unfold(feat_f1, kernel_size=(W, W), stride=stride, padding=W // 2)
        C = feat_f0.shape[1]
        f0u = f0u.view(N, C, W * W, -1).permute(0, 3, 2, 1)                 # 'n (c ww) l -> n l ww c'
        f1u = f1u.view(N, C, W * W, -1).permute(0, 3, 2, 1)
        f0u, f1u = f0u[b_ids, i_ids], f1u[b_ids, j_ids]                     # [M, ww, C]
        fcw = self.fine_preprocess.down_proj(torch.cat([feat_c0[b_ids, i_ids], feat_c1[b_ids, j_ids]], 0))
        fcf = self.fine_preprocess.merge_feat(torch.cat([torch.cat([f0u, f1u], 0), fcw[:, None].expand(-1, W * W, -1)], -1))
        f0u, f1u = torch.chunk(fcf, 2, dim=0)
        f0u, f1u = self.loftr_fine(f0u, f1u)
        picked = f0u[:, W * W // 2, :]
        sim = torch.einsum('mc,mrc->mr', picked, f1u)
        heat = torch.softmax(sim / C ** .5, dim=1).view(-1, W, W)
        coords = spatial_expectation2d(heat)
        scale1 = image1.shape[2] // hw1_f[0]
        out.update(mkpts0_f=cm["mkpts0_c"], mkpts1_f=cm["mkpts1_c"] + coords * (W // 2) * scale1, expec_f=coords)
        return out


def loftr_match_pair(model, image0, image1):
    """LoFTR_matcher.match (matchers.py:24-59) on loaded [1,1,H,W] tensors, incl. the padding quirk Q3
    (the test reads inp.size(1) == 1 channel so it always fires; pad = size % 8)"""
    import numpy as np
    if image0.size(2) % 8 != 0 or image0.size(1) % 8 != 0:
        pad_bottom = image0.size(2) % 8
        pad_right = image0.size(3) % 8
        image0 = F.pad(image0, (0, pad_right, 0, pad_bottom)); image1 = F.pad(image1, (0, pad_right, 0, pad_bottom))
    out = model(image0, image1)
    k0, k1 = out["mkpts0_f"].numpy(), out["mkpts1_f"].numpy()
    if k0.shape[0] > 0:
        return np.concatenate([k0, k1], axis=1)
    return np.full((1, 4), np.nan)
