"""SuperGlue on MI355X: linear layers through PyTorch-ROCm GEMMs, attention and the
optimal-transport matching head through csrc/attention.hip and csrc/superglue_match.hip.

Reference call site: SuperGlue_matcher (etc/feature_matching_baselines/matchers.py:62-120; 20
Sinkhorn iterations, match threshold 0.2 :70-71); network = un-vendored magicleap submodule,
restated per SURVEY.md Appendix A.3 with upstream parameter names so superglue_{indoor,outdoor}.pth
load unchanged.  Layout is token-major [2B, K, 256] (image 2p = reference view, 2p+1 = query view
of pair p); upstream's [dim, head] channel split (channel c = 4 d + h) is re-ordered ONCE at load
time to head-major (c' = 64 h + d) by permuting the q/k/v output rows and the merge input columns,
so every head is a contiguous 256-byte slice for the attention kernel.
"""
import torch
import torch.nn.functional as F

from .. import _lib
from .linear import SplitBatchedNT, SplitLinear


def _fold_bn(w, b, sd, bn):
    g, beta = sd[bn + ".weight"], sd[bn + ".bias"]
    mean, var = sd[bn + ".running_mean"], sd[bn + ".running_var"]
    s = g / torch.sqrt(var + 1e-5)
    return w * s[:, None], (b - mean) * s + beta


def fold_weights(state_dict, n_layers=18):
    """Upstream SuperGlue parameters -> the fused f32 operands the GPU path multiplies with (pure host code).

    Per GNN layer the upstream graph is  x <- x + W2 relu(BN(W1 [x ; Wm a + bm] + b1)) + b2  with a = attention(...).
    Exact re-associations, done once in float64:
      * BatchNorm folded into mlp.0 (and into the keypoint encoder);
      * the merge projection only feeds the MLP:  W1 [x ; Wm a + bm] = [W1x | W1m Wm] [x ; a] + W1m bm
        -> one 256x256 GEMM per layer disappears and [x ; a] is a single buffer the attention kernel writes into;
      * the constant b2 is carried as an offset c (x = x~ + c, c <- c + b2) folded into the NEXT layer's biases
        (Wqkv c, W1x c) and into final_proj -> the residual update is a pure in-place GEMM epilogue;
      * upstream's [dim, head] channel split (channel c = 4 d + h) re-ordered to head-major (64 h + d) by permuting
        the q/k/v output rows and the merge input columns.
    No elementwise kernel (cat / relu / add) is left between the GEMMs.
    Returns dict(kenc=[(w,b)...], layers=[dict(wqkv,bqkv,w1,b1,w2,cross)...], wf, bf, bin_score)."""
    sd = {k: v.double() for k, v in state_dict.items()}
    lin = lambda n: sd[n + ".weight"].squeeze(-1)
    f32 = lambda t: t.float().contiguous()
    kenc = []
    for i in range(4):
        w, b = _fold_bn(lin(f"kenc.encoder.{3 * i}"), sd[f"kenc.encoder.{3 * i}.bias"], sd, f"kenc.encoder.{3 * i + 1}")
        kenc.append((f32(w), f32(b)))
    kenc.append((f32(lin("kenc.encoder.12")), f32(sd["kenc.encoder.12.bias"])))
    perm = torch.tensor([(c % 64) * 4 + c // 64 for c in range(256)])          # head-major <- upstream channel
    layers = []
    c = torch.zeros(256, dtype=torch.float64)
    for l in range(n_layers):
        p = f"gnn.layers.{l}"
        wqkv = torch.cat([lin(f"{p}.attn.proj.{j}")[perm] for j in range(3)], 0)
        bqkv = torch.cat([sd[f"{p}.attn.proj.{j}.bias"][perm] for j in range(3)], 0)
        wm, bm = lin(f"{p}.attn.merge")[:, perm], sd[f"{p}.attn.merge.bias"]
        w1, b1 = _fold_bn(lin(f"{p}.mlp.0"), sd[f"{p}.mlp.0.bias"], sd, f"{p}.mlp.1")
        w1x, w1m = w1[:, :256], w1[:, 256:]
        layers.append(dict(wqkv=f32(wqkv), bqkv=f32(bqkv + wqkv @ c), w1=f32(torch.cat([w1x, w1m @ wm], 1)),
                           b1=f32(b1 + w1m @ bm + w1x @ c), w2=f32(lin(f"{p}.mlp.3")), cross=(l % 2 == 1)))
        c = c + sd[f"{p}.mlp.3.bias"]
    wf = lin("final_proj")
    return dict(kenc=kenc, layers=layers, wf=f32(wf), bf=f32(sd["final_proj.bias"] + wf @ c), bin_score=float(sd["bin_score"]))


class SuperGlueHIP:
    def __init__(self, state_dict, device="cuda", sinkhorn_iterations=20, match_threshold=0.2, n_layers=18):
        _lib.load(require_gpu=True)
        self.device = torch.device(device)
        self.iters, self.match_thr = int(sinkhorn_iterations), float(match_threshold)
        from .. import options
        self.att_variant = 0 if options.get("SPLIT") == "f16x2" else 2       # the arithmetic of the attention kernel, resolved once (options.py)
        fw = fold_weights(state_dict, n_layers)
        dev = lambda t: t.to(self.device).contiguous()
        self.kenc = [(dev(w), dev(b)) for w, b in fw["kenc"]]
        self.kenc_lin = [SplitLinear(w, b) for w, b in self.kenc[1:]]
        self.layers = [dict(wqkv=dev(L["wqkv"]), bqkv=dev(L["bqkv"]), w1t=dev(L["w1"]).t(), b1=dev(L["b1"]), w2t=dev(L["w2"]).t(),
                            cross=L["cross"]) for L in fw["layers"]]
        # the three GEMMs of a layer through csrc/gemm_split.hip (weights split / packed once here); the plain tensors above stay for
        # the stage-level parity tests
        for L in self.layers:
            L["lin_qkv"] = SplitLinear(L["wqkv"], L["bqkv"])
            L["lin1"] = SplitLinear(L["w1t"].t(), L["b1"])
            L["lin2"] = SplitLinear(L["w2t"].t())
        self.wf, self.bf = dev(fw["wf"]), dev(fw["bf"])
        self.lin_final = SplitLinear(self.wf, self.bf)
        self.score_gemm = SplitBatchedNT() if self.lin_final.split == "f16x2" else None      # (the bf16x3 arithmetic keeps the library's batched fp32 GEMM)
        self.bin_score = fw["bin_score"]
        self._ws = None
        self._size = {}

    def attention(self, qkv, n_tok, cross, out=None, ldo=256, variant=None):
        """qkv [B2,K,768] (q | k | v, each head-major) -> message [B2,K,256] (or into `out`, row stride ldo floats);
        variant None = the arithmetic resolved at construction (HIP.SPLIT: 0 = f16x2, 2 = bf16x3), 1 = exact-fp32 matrix-core kernel (A/B, tests)"""
        lib = _lib.load()
        if variant is None:
            variant = self.att_variant
        B2, K, _ = qkv.shape
        if out is None:
            out = torch.empty(B2, K, 256, dtype=torch.float32, device=qkv.device)
        base = qkv.data_ptr()
        _lib.check(lib.mfr_sg_attention_variant(base, base + 256 * 4, base + 512 * 4, 768, B2, K, 4, _lib.ptr(n_tok),
                                                1 if cross else 0, out.data_ptr(), ldo, int(variant), _lib.stream_ptr()), "mfr_sg_attention")
        return out

    def sinkhorn_match(self, S, n0, n1, kpts0, kpts1, maxN=None, variant=0):
        lib = _lib.load()
        B, ldS, _ = S.shape
        K = kpts0.shape[1]
        maxN = maxN or ldS
        dev = S.device
        need = lib.mfr_sg_match_workspace_bytes(B, ldS)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        m0 = torch.empty(B, ldS, dtype=torch.int32, device=dev)
        ms0 = torch.empty(B, ldS, dtype=torch.float32, device=dev)
        pts0 = torch.zeros(B, maxN, 2, dtype=torch.float32, device=dev)
        pts1 = torch.zeros(B, maxN, 2, dtype=torch.float32, device=dev)
        nc = torch.empty(B, dtype=torch.int32, device=dev)
        _lib.check(lib.mfr_sg_sinkhorn_match_variant(
            _lib.ptr(S.contiguous()), B, ldS, _lib.ptr(n0), _lib.ptr(n1), self.bin_score, self.iters, self.match_thr,
            _lib.ptr(kpts0.contiguous()), _lib.ptr(kpts1.contiguous()), K, _lib.ptr(self._ws), self._ws.numel(),
            _lib.ptr(m0), _lib.ptr(ms0), _lib.ptr(pts0), _lib.ptr(pts1), maxN, _lib.ptr(nc), int(variant), _lib.stream_ptr()),
            "mfr_sg_sinkhorn_match")
        return dict(matches0=m0, matching_scores0=ms0, pts0=pts0, pts1=pts1, n_corr=nc)

    @torch.no_grad()
    def final_descriptors(self, kpts, scores, desc, n, image_hw):
        H, W = image_hw
        B2, K, _ = kpts.shape
        key = (H, W, kpts.device)
        if key not in self._size:      # cached: an H2D copy is not allowed while a HIP graph is capturing
            self._size[key] = torch.tensor([float(W), float(H)], device=kpts.device)
        size = self._size[key]
        kn = (kpts - size / 2) / (float(max(W, H)) * 0.7)                             # normalize_keypoints
        # keypoint encoder MLP(3 -> 32 -> 64 -> 128 -> 256).  First layer (K = 3): three broadcast multiply-adds in a fixed order; the others
        # through csrc/gemm_split.hip -- no library GEMM, and every row's sums are independent of the batch size
        w0, b0 = self.kenc[0]
        h = (kn[..., 0:1] * w0[:, 0] + kn[..., 1:2] * w0[:, 1]) + (scores.unsqueeze(-1) * w0[:, 2] + b0)
        h = F.relu_(h).reshape(B2 * K, -1)
        for i, lin in enumerate(self.kenc_lin):
            h = lin(h, relu=i < len(self.kenc_lin) - 1)
        # xa = [x~ | a]: the running descriptors (minus the folded bias offset) and the attention output side by side
        xa = torch.empty(B2 * K, 512, dtype=torch.float32, device=kpts.device)
        xv, av = xa[:, :256], xa[:, 256:]
        torch.add(desc.reshape(B2 * K, 256), h.reshape(B2 * K, 256), out=xv)
        qkv = torch.empty(B2 * K, 768, dtype=torch.float32, device=kpts.device)
        hid = torch.empty(B2 * K, 512, dtype=torch.float32, device=kpts.device)
        for L in self.layers:
            L["lin_qkv"](xv, out=qkv)
            self.attention(qkv.view(B2, K, 768), n, L["cross"], out=av, ldo=512)
            L["lin1"](xa, out=hid, relu=True)                             # relu(W1' [x~ ; a] + b1') in the GEMM epilogue
            L["lin2"](hid, out=xv, accumulate=True)                       # x~ += W2 hid, in place
        return self.lin_final(xv).view(B2, K, 256)

    @torch.no_grad()
    def __call__(self, sp_out, image_hw, maxN=None):
        """sp_out: SuperPointHIP output for the interleaved [2B] images -> dict(pts0, pts1 [B,maxN,2],
        n_corr [B], matches0, matching_scores0)"""
        kpts, scores, desc, n = sp_out["kpts"], sp_out["scores"], sp_out["desc"], sp_out["n"]
        md = self.final_descriptors(kpts, scores, desc, n, image_hw)
        # scores = mdesc0 . mdesc1 / sqrt(256): one batched f16x2 launch, 1/16 folded (exactly) into the per-row scale of the packed operand
        if self.score_gemm is not None:
            S = self.score_gemm(md[0::2], md[1::2], out_mul=1.0 / 16.0)
        else:
            S = torch.bmm(md[0::2], md[1::2].transpose(1, 2)) * (1.0 / 16.0)
        return self.sinkhorn_match(S, n[0::2].contiguous(), n[1::2].contiguous(),
                                   kpts[0::2].contiguous(), kpts[1::2].contiguous(), maxN)
