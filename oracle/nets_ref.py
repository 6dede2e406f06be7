"""CPU ORACLE (TEST INFRASTRUCTURE ONLY) -- plain PyTorch-fp32 restatement of the matcher
networks the reference calls through its (empty, un-vendored) git submodules:

  * SuperPoint + SuperGlue  <- etc/feature_matching_baselines/SuperGlue
    (magicleap/SuperGluePretrainedNetwork; call sites matchers.py:62-120, hyper-parameters
    matchers.py:65-71: nms_radius 4, keypoint_threshold 0.005, max_keypoints 1024,
    sinkhorn_iterations 20, match_threshold 0.2).

The upstream source is not available offline, so this file restates the
PUBLISHED architectures with upstream's parameter names, and is
cross-checked against the independent HuggingFace `transformers` implementations with identical
weights (tests/test_nets_ref_vs_hf.py).  Parity vs the real pretrained upstream networks is
UNPINNED (no weights, no source offline).

Nothing in the product package imports this module.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------ SuperPoint
def simple_nms(scores, nms_radius):
    """upstream superpoint.py simple_nms: iterated max-pool NMS, 2 suppression rounds"""
    def max_pool(x):
        return F.max_pool2d(x, kernel_size=nms_radius * 2 + 1, stride=1, padding=nms_radius)
    zeros = torch.zeros_like(scores)
    max_mask = scores == max_pool(scores)
    for _ in range(2):
        supp_mask = max_pool(max_mask.float()) > 0
        supp_scores = torch.where(supp_mask, zeros, scores)
        new_max_mask = supp_scores == max_pool(supp_scores)
        max_mask = max_mask | (new_max_mask & (~supp_mask))
    return torch.where(max_mask, scores, zeros)


def sample_descriptors(keypoints, descriptors, s=8):
    """upstream sample_descriptors: bilinear grid_sample(align_corners=True) + L2"""
    b, c, h, w = descriptors.shape
    keypoints = keypoints - s / 2 + 0.5
    keypoints = keypoints / torch.tensor([(w * s - s / 2 - 0.5), (h * s - s / 2 - 0.5)]).to(keypoints)[None]
    keypoints = keypoints * 2 - 1
    descriptors = F.grid_sample(descriptors, keypoints.view(b, 1, -1, 2), mode="bilinear", align_corners=True)
    descriptors = F.normalize(descriptors.reshape(b, c, -1), p=2, dim=1)
    return descriptors


class SuperPointRef(nn.Module):
    """SuperPoint (DeTone et al. 2018) with upstream parameter names (conv1a ... convDb)."""

    def __init__(self, nms_radius=4, keypoint_threshold=0.005, max_keypoints=1024, remove_borders=4):
        super().__init__()
        self.nms_radius, self.keypoint_threshold = nms_radius, keypoint_threshold
        self.max_keypoints, self.remove_borders = max_keypoints, remove_borders
        c1, c2, c3, c4, c5 = 64, 64, 128, 128, 256
        self.conv1a = nn.Conv2d(1, c1, 3, 1, 1); self.conv1b = nn.Conv2d(c1, c1, 3, 1, 1)
        self.conv2a = nn.Conv2d(c1, c2, 3, 1, 1); self.conv2b = nn.Conv2d(c2, c2, 3, 1, 1)
        self.conv3a = nn.Conv2d(c2, c3, 3, 1, 1); self.conv3b = nn.Conv2d(c3, c3, 3, 1, 1)
        self.conv4a = nn.Conv2d(c3, c4, 3, 1, 1); self.conv4b = nn.Conv2d(c4, c4, 3, 1, 1)
        self.convPa = nn.Conv2d(c4, c5, 3, 1, 1); self.convPb = nn.Conv2d(c5, 65, 1, 1, 0)
        self.convDa = nn.Conv2d(c4, c5, 3, 1, 1); self.convDb = nn.Conv2d(c5, 256, 1, 1, 0)

    def encode(self, image):
        x = F.relu(self.conv1a(image)); x = F.relu(self.conv1b(x)); x = F.max_pool2d(x, 2, 2)
        x = F.relu(self.conv2a(x)); x = F.relu(self.conv2b(x)); x = F.max_pool2d(x, 2, 2)
        x = F.relu(self.conv3a(x)); x = F.relu(self.conv3b(x)); x = F.max_pool2d(x, 2, 2)
        x = F.relu(self.conv4a(x)); x = F.relu(self.conv4b(x))
        return x

    def score_map(self, x):
        cPa = F.relu(self.convPa(x))
        scores = self.convPb(cPa)
        scores = F.softmax(scores, 1)[:, :-1]
        b, _, h, w = scores.shape
        scores = scores.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8)
        scores = scores.permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
        return scores

    def dense_descriptors(self, x):
        cDa = F.relu(self.convDa(x))
        return F.normalize(self.convDb(cDa), p=2, dim=1)

    @torch.no_grad()
    def forward(self, image):
        """image [B,1,H,W] in [0,1] -> list of (keypoints [N,2] (x,y), scores [N], descriptors [256,N])"""
        x = self.encode(image)
        raw = self.score_map(x)
        scores = simple_nms(raw, self.nms_radius)
        desc = self.dense_descriptors(x)
        b, h8, w8 = scores.shape
        out = []
        for i in range(b):
            s = scores[i]
            kp = torch.nonzero(s > self.keypoint_threshold)                 # (y, x) raster order
            sc = s[kp[:, 0], kp[:, 1]]
            bd = self.remove_borders
            m = (kp[:, 0] >= bd) & (kp[:, 0] < h8 - bd) & (kp[:, 1] >= bd) & (kp[:, 1] < w8 - bd)
            kp, sc = kp[m], sc[m]
            if self.max_keypoints >= 0 and len(kp) > self.max_keypoints:
                # torch.topk order among exact ties is unspecified upstream; pinned here to
                # (score desc, raster index asc) -- the order the device kernel reproduces
                order = torch.sort(sc, descending=True, stable=True).indices[: self.max_keypoints]
                kp, sc = kp[order], sc[order]
            kp = torch.flip(kp, [1]).float()                                # (x, y)
            d = sample_descriptors(kp[None], desc[i:i + 1], 8)[0]
            out.append((kp, sc, d))
        return out


# ------------------------------------------------------------------------------ SuperGlue
def MLP(channels, do_bn=True):
    n = len(channels)
    layers = []
    for i in range(1, n):
        layers.append(nn.Conv1d(channels[i - 1], channels[i], kernel_size=1, bias=True))
        if i < (n - 1):
            if do_bn:
                layers.append(nn.BatchNorm1d(channels[i]))
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


def normalize_keypoints(kpts, height, width):
    size = kpts.new_tensor([[float(width), float(height)]])
    center = size / 2
    scaling = size.max(1, keepdim=True).values * 0.7
    return (kpts - center[:, None, :]) / scaling[:, None, :]


class KeypointEncoder(nn.Module):
    def __init__(self, feature_dim, layers):
        super().__init__()
        self.encoder = MLP([3] + layers + [feature_dim])
        nn.init.constant_(self.encoder[-1].bias, 0.0)

    def forward(self, kpts, scores):
        inputs = [kpts.transpose(1, 2), scores.unsqueeze(1)]
        return self.encoder(torch.cat(inputs, dim=1))


def attention(query, key, value):
    dim = query.shape[1]
    scores = torch.einsum("bdhn,bdhm->bhnm", query, key) / dim ** 0.5
    prob = F.softmax(scores, dim=-1)
    return torch.einsum("bhnm,bdhm->bdhn", prob, value), prob


class MultiHeadedAttention(nn.Module):
    def __init__(self, num_heads, d_model):
        super().__init__()
        assert d_model % num_heads == 0
        self.dim = d_model // num_heads
        self.num_heads = num_heads
        self.merge = nn.Conv1d(d_model, d_model, kernel_size=1)
        self.proj = nn.ModuleList([nn.Conv1d(d_model, d_model, kernel_size=1) for _ in range(3)])

    def forward(self, query, key, value):
        b = query.size(0)
        query, key, value = [l(x).view(b, self.dim, self.num_heads, -1)
                             for l, x in zip(self.proj, (query, key, value))]
        x, _ = attention(query, key, value)
        return self.merge(x.contiguous().view(b, self.dim * self.num_heads, -1))


class AttentionalPropagation(nn.Module):
    def __init__(self, feature_dim, num_heads):
        super().__init__()
        self.attn = MultiHeadedAttention(num_heads, feature_dim)
        self.mlp = MLP([feature_dim * 2, feature_dim * 2, feature_dim])
        nn.init.constant_(self.mlp[-1].bias, 0.0)

    def forward(self, x, source):
        message = self.attn(x, source, source)
        return self.mlp(torch.cat([x, message], dim=1))


class AttentionalGNN(nn.Module):
    def __init__(self, feature_dim, layer_names):
        super().__init__()
        self.layers = nn.ModuleList([AttentionalPropagation(feature_dim, 4) for _ in range(len(layer_names))])
        self.names = layer_names

    def forward(self, desc0, desc1):
        for layer, name in zip(self.layers, self.names):
            if name == "cross":
                src0, src1 = desc1, desc0
            else:
                src0, src1 = desc0, desc1
            delta0, delta1 = layer(desc0, src0), layer(desc1, src1)
            desc0, desc1 = (desc0 + delta0), (desc1 + delta1)
        return desc0, desc1


def log_sinkhorn_iterations(Z, log_mu, log_nu, iters):
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1)


def log_optimal_transport(scores, alpha, iters):
    b, m, n = scores.shape
    one = scores.new_tensor(1)
    ms, ns = (m * one).to(scores), (n * one).to(scores)
    bins0 = alpha.expand(b, m, 1)
    bins1 = alpha.expand(b, 1, n)
    alpha = alpha.expand(b, 1, 1)
    couplings = torch.cat([torch.cat([scores, bins0], -1), torch.cat([bins1, alpha], -1)], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])
    log_mu, log_nu = log_mu[None].expand(b, -1), log_nu[None].expand(b, -1)
    Z = log_sinkhorn_iterations(couplings, log_mu, log_nu, iters)
    return Z - norm


def arange_like(x, dim):
    return x.new_ones(x.shape[dim]).cumsum(0) - 1


class SuperGlueRef(nn.Module):
    """SuperGlue (Sarlin et al. 2020), upstream parameter names (kenc, gnn, final_proj, bin_score)."""

    def __init__(self, sinkhorn_iterations=20, match_threshold=0.2, descriptor_dim=256,
                 keypoint_encoder=(32, 64, 128, 256), n_gnn=9):
        super().__init__()
        self.sinkhorn_iterations, self.match_threshold = sinkhorn_iterations, match_threshold
        self.kenc = KeypointEncoder(descriptor_dim, list(keypoint_encoder))
        self.gnn = AttentionalGNN(descriptor_dim, ["self", "cross"] * n_gnn)
        self.final_proj = nn.Conv1d(descriptor_dim, descriptor_dim, kernel_size=1, bias=True)
        self.bin_score = nn.Parameter(torch.tensor(1.0))

    @torch.no_grad()
    def forward(self, kpts0, scores0, desc0, kpts1, scores1, desc1, image_hw):
        """kpts [1,N,2] (x,y), scores [1,N], desc [1,256,N]; returns dict like upstream"""
        H, W = image_hw
        if kpts0.shape[1] == 0 or kpts1.shape[1] == 0:
            s0, s1 = kpts0.shape[:-1], kpts1.shape[:-1]
            return dict(matches0=kpts0.new_full(s0, -1, dtype=torch.int64), matches1=kpts1.new_full(s1, -1, dtype=torch.int64),
                        matching_scores0=kpts0.new_zeros(s0), matching_scores1=kpts1.new_zeros(s1))
        k0 = normalize_keypoints(kpts0, H, W)
        k1 = normalize_keypoints(kpts1, H, W)
        d0 = desc0 + self.kenc(k0, scores0)
        d1 = desc1 + self.kenc(k1, scores1)
        d0, d1 = self.gnn(d0, d1)
        m0, m1 = self.final_proj(d0), self.final_proj(d1)
        scores = torch.einsum("bdn,bdm->bnm", m0, m1) / 256 ** 0.5
        Z = log_optimal_transport(scores, self.bin_score, self.sinkhorn_iterations)
        max0, max1 = Z[:, :-1, :-1].max(2), Z[:, :-1, :-1].max(1)
        i0, i1 = max0.indices, max1.indices
        mutual0 = arange_like(i0, 1)[None] == i1.gather(1, i0)
        mutual1 = arange_like(i1, 1)[None] == i0.gather(1, i1)
        zero = Z.new_tensor(0)
        ms0 = torch.where(mutual0, max0.values.exp(), zero)
        ms1 = torch.where(mutual1, ms0.gather(1, i1), zero)
        valid0 = mutual0 & (ms0 > self.match_threshold)
        valid1 = mutual1 & valid0.gather(1, i1)
        matches0 = torch.where(valid0, i0, i0.new_tensor(-1))
        matches1 = torch.where(valid1, i1, i1.new_tensor(-1))
        return dict(matches0=matches0, matches1=matches1, matching_scores0=ms0, matching_scores1=ms1,
                    log_assignment=Z, mdesc0=m0, mdesc1=m1)


def superglue_match_pair(sp, sg, image0, image1):
    """SuperGlue_matcher.match (matchers.py:93-120) on already-loaded [1,1,H,W] tensors:
    returns [N,4] (x0,y0,x1,y1) float32 or a single NaN row."""
    import numpy as np
    H, W = image0.shape[-2:]
    (k0, s0, d0), = sp(image0)
    (k1, s1, d1), = sp(image1)
    pred = sg(k0[None], s0[None], d0[None], k1[None], s1[None], d1[None], (H, W))
    m = pred["matches0"][0].numpy()
    k0, k1 = k0.numpy(), k1.numpy()
    valid = m > -1                                                            # matchers.py:111
    mk0, mk1 = k0[valid], k1[m[valid]]
    if mk0.shape[0] > 0:
        return np.concatenate([mk0, mk1], axis=1)                             # :115-116
    return np.full((1, 4), np.nan)                                            # :120
