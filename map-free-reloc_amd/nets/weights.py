"""Deterministic synthetic weights with upstream parameter names.

No checkpoints exist offline (superpoint_v1.pth / superglue_{indoor,outdoor}.pth /
{indoor,outdoor}_ot.ckpt; matchers.py:17,70), so benchmarks and parity tests run on seeded random
weights of the exact upstream architectures.  The state-dict keys are upstream's, so the real
checkpoints load through the same path (`load_checkpoint`) when they are available.

Plain i.i.d. weights make SuperGlue's assignment flat (no matches -> the solver stage would be
idle), so the SuperGlue recipe is "structured random": the GNN updates and the keypoint encoder
are scaled down (the network stays a mild perturbation of its input descriptors) and the final
projection is scaled up so that the score matrix is peaky.  Accuracy of such weights is
meaningless; shapes, arithmetic and control flow are exactly those of the trained network.
"""
import torch


def _randn(g, shape, std):
    return torch.randn(shape, generator=g) * std


def superpoint_state_dict(seed=1234):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    chans = [("conv1a", 1, 64, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
             ("conv3a", 64, 128, 3), ("conv3b", 128, 128, 3), ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
             ("convPa", 128, 256, 3), ("convPb", 256, 65, 1), ("convDa", 128, 256, 3), ("convDb", 256, 256, 1)]
    for name, ci, co, k in chans:
        fan_in = ci * k * k
        w = _randn(g, (co, ci, k, k), 1.0)
        # zero-sum filters and zero biases: an untrained ReLU stack otherwise responds mostly to
        # the image's DC level and every descriptor comes out (almost) the same vector
        w = w - w.mean(dim=(1, 2, 3), keepdim=True)
        sd[f"{name}.weight"] = w / (w.std(dim=(1, 2, 3), keepdim=True) + 1e-9) * (2.0 / fan_in) ** 0.5
        sd[f"{name}.bias"] = torch.zeros(co)
    # a peaky detector head: logits of the zero-bias stack scale with image contrast (~1e-2)
    sd["convPb.weight"] = sd["convPb.weight"] * 80.0
    return sd


def superglue_state_dict(seed=4321, gnn_gain=0.1, kenc_gain=0.1, proj_gain=12.0, n_layers=18):
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv1d(name, ci, co, gain=1.0, zero_bias=False):
        sd[f"{name}.weight"] = _randn(g, (co, ci, 1), gain / ci ** 0.5)
        sd[f"{name}.bias"] = torch.zeros(co) if zero_bias else _randn(g, (co,), 0.02 * gain)

    def bn(name, c):
        sd[f"{name}.weight"] = 1.0 + _randn(g, (c,), 0.05)
        sd[f"{name}.bias"] = _randn(g, (c,), 0.05)
        sd[f"{name}.running_mean"] = _randn(g, (c,), 0.05)
        sd[f"{name}.running_var"] = 1.0 + 0.1 * torch.rand((c,), generator=g)
        sd[f"{name}.num_batches_tracked"] = torch.tensor(0)
    ch = [3, 32, 64, 128, 256, 256]
    for i in range(1, 6):
        last = i == 5
        conv1d(f"kenc.encoder.{3 * (i - 1)}", ch[i - 1], ch[i], gain=kenc_gain if last else 1.4, zero_bias=last)
        if not last:
            bn(f"kenc.encoder.{3 * (i - 1) + 1}", ch[i])
    for l in range(n_layers):
        p = f"gnn.layers.{l}"
        for j in range(3):
            conv1d(f"{p}.attn.proj.{j}", 256, 256, gain=2.0 if j < 2 else 1.0)
        conv1d(f"{p}.attn.merge", 256, 256)
        conv1d(f"{p}.mlp.0", 512, 512, gain=1.4)
        bn(f"{p}.mlp.1", 512)
        conv1d(f"{p}.mlp.3", 512, 256, gain=gnn_gain, zero_bias=True)
    conv1d("final_proj", 256, 256, gain=proj_gain)
    sd["bin_score"] = torch.tensor(1.0)
    return sd


def loftr_state_dict(seed=2468, feat_gain=20.0, msg_gain=0.1):
    """LoFTR (default_cfg architecture) with upstream parameter names (`backbone.*`,
    `loftr_coarse.layers.N.*`, `fine_preprocess.*`, `loftr_fine.layers.N.*`).  Structured random like
    the SuperGlue recipe: zero-sum conv filters, identity-like BatchNorm, coarse features scaled so
    that the dual-softmax is peaky (conf > 0.2 needs a ~9 nat margin over 6120 candidates), and
    transformer messages scaled down through norm2 so the GNN stays a perturbation."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, ci, co, k, gain=1.0):
        w = _randn(g, (co, ci, k, k), 1.0)
        w = w - w.mean(dim=(1, 2, 3), keepdim=True)
        sd[f"{name}.weight"] = w / (w.std(dim=(1, 2, 3), keepdim=True) + 1e-9) * (gain * (2.0 / (ci * k * k)) ** 0.5)

    def bn(name, c):
        sd[f"{name}.weight"] = torch.ones(c); sd[f"{name}.bias"] = torch.zeros(c)
        sd[f"{name}.running_mean"] = torch.zeros(c); sd[f"{name}.running_var"] = torch.ones(c)
        sd[f"{name}.num_batches_tracked"] = torch.tensor(0)

    def block(name, ci, co, stride):
        conv(f"{name}.conv1", ci, co, 3); conv(f"{name}.conv2", co, co, 3); bn(f"{name}.bn1", co); bn(f"{name}.bn2", co)
        if stride != 1:
            conv(f"{name}.downsample.0", ci, co, 1); bn(f"{name}.downsample.1", co)
    conv("backbone.conv1", 1, 128, 7, gain=8.0); bn("backbone.bn1", 128)
    dims = [128, 196, 256]
    block("backbone.layer1.0", 128, 128, 1); block("backbone.layer1.1", 128, 128, 1)
    block("backbone.layer2.0", 128, 196, 2); block("backbone.layer2.1", 196, 196, 1)
    block("backbone.layer3.0", 196, 256, 2); block("backbone.layer3.1", 256, 256, 1)
    conv("backbone.layer3_outconv", 256, 256, 1, gain=feat_gain)
    conv("backbone.layer2_outconv", 196, 256, 1)
    conv("backbone.layer2_outconv2.0", 256, 256, 3); bn("backbone.layer2_outconv2.1", 256); conv("backbone.layer2_outconv2.3", 256, 196, 3)
    conv("backbone.layer1_outconv", 128, 196, 1)
    conv("backbone.layer1_outconv2.0", 196, 196, 3); bn("backbone.layer1_outconv2.1", 196); conv("backbone.layer1_outconv2.3", 196, 128, 3)

    def lin(name, ci, co, gain=1.0, bias=False):
        sd[f"{name}.weight"] = _randn(g, (co, ci), gain / ci ** 0.5)
        if bias:
            sd[f"{name}.bias"] = torch.zeros(co)

    def encoder(prefix, d, n):
        for l in range(n):
            p = f"{prefix}.layers.{l}"
            for nm in ("q_proj", "k_proj", "v_proj", "merge"):
                lin(f"{p}.{nm}", d, d)
            lin(f"{p}.mlp.0", 2 * d, 2 * d, gain=1.4); lin(f"{p}.mlp.2", 2 * d, d)
            sd[f"{p}.norm1.weight"] = torch.ones(d); sd[f"{p}.norm1.bias"] = torch.zeros(d)
            sd[f"{p}.norm2.weight"] = torch.full((d,), msg_gain); sd[f"{p}.norm2.bias"] = torch.zeros(d)
    encoder("loftr_coarse", 256, 8)
    lin("fine_preprocess.down_proj", 256, 128, gain=0.05, bias=True)
    lin("fine_preprocess.merge_feat", 256, 128, bias=True)
    encoder("loftr_fine", 128, 2)
    return sd


def load_checkpoint(path):
    """upstream .pth / .ckpt -> flat state dict.  superpoint_v1.pth / superglue_*.pth are plain state dicts;
    LoFTR's *_ot.ckpt are Lightning checkpoints that keep it under 'state_dict' with a `matcher.` prefix
    (matchers.py:17-18 loads them strict=False after upstream strips nothing -- see strip_prefix).  Tensors only:
    weights_only=True refuses pickled code in user-supplied files."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    return sd


def strip_prefix(sd, prefix):
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in sd.items()}


def synthetic_or_raise(what, cfg, maker):
    """seeded synthetic weights are only handed out when the config opted into synthetic operation
    (DATASET.SYNTHETIC or ALLOW_SYNTHETIC_WEIGHTS): a real run without checkpoints must not produce
    plausible-looking poses from random networks"""
    import warnings
    allow = bool(cfg.get("ALLOW_SYNTHETIC_WEIGHTS", False)) or bool(cfg.DATASET.get("SYNTHETIC", None))
    if not allow:
        raise FileNotFoundError(f"{what}: no checkpoint configured (set the *_WEIGHTS key), and synthetic weights were not "
                                f"requested (ALLOW_SYNTHETIC_WEIGHTS: True or DATASET.SYNTHETIC)")
    warnings.warn(f"{what}: seeded synthetic weights in use (results are NOT meaningful)")
    return maker()
