"""LoFTR on MI355X: ResNet-FPN backbone, linear layers and LayerNorm through PyTorch-ROCm (dense
library work), the coarse transformer's linear attention, the dual-softmax coarse matching and the
fine-window gather through csrc/loftr.hip.

Reference call site: LoFTR_matcher (etc/feature_matching_baselines/matchers.py:12-59): LoFTR
(default_cfg) + `*_ot.ckpt` loaded strict=False; network = un-vendored zju3dv/LoFTR submodule,
restated per SURVEY.md Appendix A.4 with upstream parameter names (`backbone.*`, `loftr_coarse.*`,
`fine_preprocess.*`, `loftr_fine.*`) so the real checkpoints load through the same path.
BatchNorm (eval) is folded into the preceding bias-free convolutions at load time.

The fine-level transformer (2M sequences of 25 tokens, d 128) and the 5x5 expectation are tiny and
stay on torch ops; everything that touches the 6120-token coarse level is a HIP kernel.
"""
import math

import torch
import torch.nn.functional as F

from .. import _lib, options


def _fold(w, sd, bn):
    s = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + 1e-5)
    return w * s[:, None, None, None], sd[bn + ".bias"] - sd[bn + ".running_mean"] * s


def position_encoding_sine(d_model, H, W, device):
    """upstream PositionEncodingSine with TEMP_BUG_FIX False: (-log(1e4)/d_model)//2 == -1.0"""
    pe = torch.zeros((d_model, H, W))
    y_position = torch.ones((H, W)).cumsum(0).float().unsqueeze(0)
    x_position = torch.ones((H, W)).cumsum(1).float().unsqueeze(0)
    div_term = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / d_model // 2))[:, None, None]
    pe[0::4] = torch.sin(x_position * div_term); pe[1::4] = torch.cos(x_position * div_term)
    pe[2::4] = torch.sin(y_position * div_term); pe[3::4] = torch.cos(y_position * div_term)
    return pe.to(device)


class LoFTRHIP:
    def __init__(self, state_dict, device="cuda", thr=0.2, border_rm=2, temperature=0.1, window=5):
        _lib.load(require_gpu=True)
        self.device = torch.device(device)
        self.thr, self.border, self.temp, self.W = float(thr), int(border_rm), float(temperature), int(window)
        sd = {k: v.float() for k, v in state_dict.items()}
        dev = lambda t: t.to(self.device).contiguous()
        w = {}

        def convbn(name, conv, bn):
            cw, cb = _fold(sd[conv + ".weight"], sd, bn)
            w[name] = (dev(cw), dev(cb))

        def conv(name, key):
            w[name] = (dev(sd[key + ".weight"]), None)
        convbn("conv1", "backbone.conv1", "backbone.bn1")
        for L, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
            for b in (0, 1):
                p = f"backbone.{L}.{b}"
                convbn(f"{L}.{b}.c1", f"{p}.conv1", f"{p}.bn1"); convbn(f"{L}.{b}.c2", f"{p}.conv2", f"{p}.bn2")
                if b == 0 and stride != 1:
                    convbn(f"{L}.{b}.ds", f"{p}.downsample.0", f"{p}.downsample.1")
        conv("l3out", "backbone.layer3_outconv"); conv("l2out", "backbone.layer2_outconv"); conv("l1out", "backbone.layer1_outconv")
        convbn("l2out2.0", "backbone.layer2_outconv2.0", "backbone.layer2_outconv2.1"); conv("l2out2.3", "backbone.layer2_outconv2.3")
        convbn("l1out2.0", "backbone.layer1_outconv2.0", "backbone.layer1_outconv2.1"); conv("l1out2.3", "backbone.layer1_outconv2.3")
        self.w = w
        # stride-1 3x3 convolutions go through the fused Winograd/MFMA kernel (csrc/winograd_conv.hip): transformed
        # filters packed once here; BatchNorm is already folded into (w, b).  options CONV = "miopen" keeps the library path.
        self.upk = {}
        self.igemm = {}
        lib = _lib.load()
        if options.get("CONV") == "wino":
            from .conv import IgemmConv, WinoConv3x3
            own_rest = options.get("SPLIT") == "f16x2"                   # the implicit-GEMM kernel exists in the f16x2 arithmetic
            strides = {"conv1": 2, "layer2.0.c1": 2, "layer2.0.ds": 2, "layer3.0.c1": 2, "layer3.0.ds": 2}
            for name, (cw, cb) in w.items():
                if tuple(cw.shape[2:]) == (3, 3) and name not in strides:
                    self.upk[name] = WinoConv3x3(cw, cb)                 # both packed filter forms; kernel chosen per shape (nets/conv.py)
                elif own_rest:
                    # strided 3x3, 7x7 and every 1x1 convolution: one launch of the implicit-GEMM kernel each (rounds 1-4: MIOpen / hipBLASLt)
                    self.igemm[name] = IgemmConv(cw, cb, strides.get(name, 1))

        def encoder(prefix, n):
            layers = []
            for l in range(n):
                p = f"{prefix}.layers.{l}"
                layers.append(dict(
                    wq=dev(sd[f"{p}.q_proj.weight"]), wkv=dev(torch.cat([sd[f"{p}.k_proj.weight"], sd[f"{p}.v_proj.weight"]], 0)),
                    wm=dev(sd[f"{p}.merge.weight"]), w1=dev(sd[f"{p}.mlp.0.weight"]), w2=dev(sd[f"{p}.mlp.2.weight"]),
                    n1=(dev(sd[f"{p}.norm1.weight"]), dev(sd[f"{p}.norm1.bias"])),
                    n2=(dev(sd[f"{p}.norm2.weight"]), dev(sd[f"{p}.norm2.bias"]))))
            return layers
        self.coarse = encoder("loftr_coarse", 8)
        self.fine = encoder("loftr_fine", 2)
        self.down_proj = (dev(sd["fine_preprocess.down_proj.weight"]), dev(sd["fine_preprocess.down_proj.bias"]))
        self.merge_feat = (dev(sd["fine_preprocess.merge_feat.weight"]), dev(sd["fine_preprocess.merge_feat.bias"]))
        self._ws_la = self._ws_cm = None
        self._fine_lin = None
        # the coarse similarity matrix feat_c0 feat_c1^T / C as one batched f16x2 launch (csrc/gemm_split.hip); the bf16x3 arithmetic keeps the library's GEMM
        if options.get("SPLIT") == "f16x2":
            from .linear import SplitBatchedNT
            self.sim_gemm = SplitBatchedNT()
        else:
            self.sim_gemm = None
        self._pe = {}

    # ------------------------------------------------------------------ backbone (torch / MIOpen)
    def _c(self, x, name, stride=1, act=None, residual=None):
        """conv (+folded BN) [+ residual] [+ activation]; stride-1 3x3 layers: one fused Winograd launch"""
        cw, cb = self.w[name]
        if stride == 1 and name in self.upk:
            return self.upk[name](x, act={None: 0, "relu": 1, "leaky": 2}[act], residual=residual)
        if name in self.igemm and residual is None and act in (None, "relu"):
            assert self.igemm[name].stride == stride
            return self.igemm[name](x, relu=act == "relu")
        if cw.shape[-1] == 1:
            # 1x1 convolutions (FPN lateral / output convs, the stride-2 downsample of a BasicBlock) are plain matrix products:
            # one batched library GEMM [Cout,Cin] x [Cin,HW] per image; stride 2 = the same on the even-pixel sub-grid
            if stride != 1:
                x = x[:, :, ::stride, ::stride]
            B, C, H, W = x.shape
            w3 = cw.view(1, cw.shape[0], C).expand(B, -1, -1)
            x3 = x.reshape(B, C, H * W)
            x = (torch.baddbmm(cb.view(1, -1, 1), w3, x3) if cb is not None else torch.bmm(w3, x3)).view(B, -1, H, W)
        else:
            x = F.conv2d(x, cw, cb, stride=stride, padding=cw.shape[-1] // 2)
        if residual is not None:
            x = x + residual
        if act == "relu":
            x = F.relu_(x)
        elif act == "leaky":
            x = F.leaky_relu_(x, 0.01)
        return x

    def _block(self, x, name, stride):
        y = self._c(x, f"{name}.c1", stride, "relu")
        if stride != 1:
            x = self._c(x, f"{name}.ds", stride)
        return self._c(y, f"{name}.c2", 1, "relu", residual=x)          # relu(x + bn2(conv2(y)))

    def backbone(self, x):
        x0 = self._c(x, "conv1", 2, "relu")
        x1 = self._block(self._block(x0, "layer1.0", 1), "layer1.1", 1)
        x2 = self._block(self._block(x1, "layer2.0", 2), "layer2.1", 1)
        x3 = self._block(self._block(x2, "layer3.0", 2), "layer3.1", 1)
        x3_out = self._c(x3, "l3out")
        x2_out = self._c(self._c(self._lateral_merge(x2, "l2out", x3_out), "l2out2.0", 1, "leaky"), "l2out2.3")
        h1 = self._c(self._lateral_merge(x1, "l1out", x2_out), "l1out2.0", 1, "leaky")
        # round 6: the fine map leaves its last convolution token-major (NHWC memory, returned as an NCHW VIEW) when the direct kernel runs it: coarse_tail's
        # NCHW -> NHWC transposition pass disappears
        fine = self.upk["l1out2.3"].rows(h1) if "l1out2.3" in self.upk else None
        x1_out = fine.permute(0, 3, 1, 2) if fine is not None else self._c(h1, "l1out2.3")
        return x3_out, x1_out

    def _lateral_merge(self, x, name, lo):
        """FPN merge  layerN_outconv(x) + interpolate(lo, x2, bilinear, align_corners=True):  one launch when the 1x1 convolution runs the own
        implicit-GEMM kernel (its epilogue samples lo: round 6), else convolution + the in-place up-sample-and-add kernel"""
        if name in self.igemm and x.shape[2] == 2 * lo.shape[2] and x.shape[3] == 2 * lo.shape[3]:
            return self.igemm[name](x, up_add=lo)
        return self.upsample2x_add(lo, self._c(x, name))

    # ------------------------------------------------------------------ HIP stage wrappers
    def linear_attention(self, q, kv):
        """q [B,L,256]; kv [B,L,512] (k | v) -> message [B,L,256]  (upstream LinearAttention)"""
        lib = _lib.load()
        B, L, D = q.shape
        heads = D // 32
        need = lib.mfr_loftr_linear_attention_workspace_bytes(B, L, heads)
        if self._ws_la is None or self._ws_la.numel() < need:
            self._ws_la = torch.empty(need, dtype=torch.uint8, device=q.device)
        out = torch.empty(B, L, D, dtype=torch.float32, device=q.device)
        base = kv.data_ptr()
        # q / kv may be column slices of one [B, L, 3 D] buffer (self layers): rows stay L * ld apart, only the row stride differs
        assert q.stride(2) == 1 and kv.stride(2) == 1 and q.stride(0) == L * q.stride(1) and kv.stride(0) == L * kv.stride(1)
        _lib.check(lib.mfr_loftr_linear_attention(q.data_ptr(), q.stride(1), base, base + D * 4, kv.stride(1), B, L, heads, _lib.ptr(self._ws_la),
                                                  self._ws_la.numel(), _lib.ptr(out), D, _lib.stream_ptr()),
                   "mfr_loftr_linear_attention")
        return out

    def fine_attention(self, q, kv):
        """q [Bw,25,128]; kv [Bw,25,256] (k | v) -> message [Bw,25,128]: LinearAttention of the fine transformer
        (8 heads x 16, 5x5 windows), one wavefront per window (csrc/loftr.hip)"""
        lib = _lib.load()
        Bw, L, D = q.shape
        assert q.stride(2) == 1 and kv.stride(2) == 1 and q.stride(0) == L * q.stride(1) and kv.stride(0) == L * kv.stride(1)
        out = torch.empty(Bw, L, D, dtype=torch.float32, device=q.device)
        base = kv.data_ptr()
        _lib.check(lib.mfr_loftr_fine_attention(q.data_ptr(), q.stride(1), base, base + D * 4, kv.stride(1), Bw, L, D, 8, _lib.ptr(out), D,
                                                _lib.stream_ptr()), "mfr_loftr_fine_attention")
        return out

    def coarse_match(self, S, hw0, hw1, variant=0):
        """variant 0: two sweeps over S (default); 1: the four-sweep kernels of round 1 (A/B, cross-check)"""
        lib = _lib.load()
        B, L0, L1 = S.shape
        need = lib.mfr_loftr_coarse_match_workspace_bytes(B, L0, L1)
        if self._ws_cm is None or self._ws_cm.numel() < need:
            self._ws_cm = torch.empty(need, dtype=torch.uint8, device=S.device)
        i_ids = torch.empty(B, L0, dtype=torch.int32, device=S.device); j_ids = torch.empty_like(i_ids)
        mconf = torch.empty(B, L0, dtype=torch.float32, device=S.device)
        n = torch.empty(B, dtype=torch.int32, device=S.device)
        _lib.check(lib.mfr_loftr_coarse_match_variant(_lib.ptr(S.contiguous()), B, hw0[0], hw0[1], hw1[0], hw1[1], self.temp, self.thr,
                                                      self.border, _lib.ptr(self._ws_cm), self._ws_cm.numel(), _lib.ptr(i_ids),
                                                      _lib.ptr(j_ids), _lib.ptr(mconf), _lib.ptr(n), variant, _lib.stream_ptr()),
                   "mfr_loftr_coarse_match")
        return i_ids, j_ids, mconf, n

    def upsample2x_add(self, lo, y):
        """y += F.interpolate(lo, scale_factor=2, bilinear, align_corners=True), one pass (csrc/loftr_fused.hip)"""
        lib = _lib.load()
        B, C, H, W = lo.shape
        assert y.shape == (B, C, 2 * H, 2 * W) and y.is_contiguous()
        _lib.check(lib.mfr_upsample2x_add(_lib.ptr(lo.contiguous()), _lib.ptr(y), B * C, H, W, _lib.stream_ptr()), "mfr_upsample2x_add")
        return y

    @staticmethod
    def layernorm(x, norm, out, residual=None):
        """out = [residual +] LayerNorm(x) over the last dim (128 / 256); x, out, residual are 2-D row-strided views
        (csrc/loftr_fused.hip: strides let the result land inside the MLP's [x | message] operand / update x in place)"""
        lib = _lib.load()
        rows, C = x.shape
        assert x.stride(1) == 1 and out.stride(1) == 1 and out.shape == x.shape
        _lib.check(lib.mfr_layernorm(x.data_ptr(), x.stride(0), _lib.ptr(norm[0]), _lib.ptr(norm[1]),
                                     residual.data_ptr() if residual is not None else None, residual.stride(0) if residual is not None else 0,
                                     rows, C, 1e-5, out.data_ptr(), out.stride(0), _lib.stream_ptr()), "mfr_layernorm")
        return out

    def coarse_match_features(self, f0, f1, hw):
        """coarse features [B,L,256] x2 -> dual-softmax mutual-NN matches (upstream CoarseMatching, dual_softmax)"""
        C = f0.shape[-1]
        # (f0 / sqrt C) . (f1 / sqrt C): for C a power of two (256) scaling the 6 MB operand is EXACT and commutes with the
        # contraction bit for bit, so the 150 MB/pair similarity matrix is written once and never rescaled in place
        if C & (C - 1) == 0 and self.sim_gemm is not None:
            S = self.sim_gemm(f0, f1, out_mul=1.0 / C)               # one batched f16x2 launch, 1 / C folded into the packed operand's row scales
        elif C & (C - 1) == 0:
            S = torch.bmm(f0 * (1.0 / C), f1.transpose(1, 2))        # strided operands are fine for the batched GEMM
        else:
            S = torch.bmm(f0, f1.transpose(1, 2))
            S.mul_(1.0 / C)
        return self.coarse_match(S, hw, hw)

    def fine_match(self, xf, M, lin_idx, k1, pts1, out_scale, expec=None):
        """xf [2, M * W * W, 256] (fine features in the first 128 columns) -> pts1[lin_idx] = k1[lin_idx] + expectation * out_scale (in place)"""
        lib = _lib.load()
        _lib.check(lib.mfr_loftr_fine_match(_lib.ptr(xf[0]), _lib.ptr(xf[1]), xf.shape[-1], 128, M, self.W, out_scale,
                                            _lib.ptr(lin_idx) if lin_idx is not None else None, _lib.ptr(k1) if k1 is not None else None,
                                            _lib.ptr(pts1) if pts1 is not None else None, _lib.ptr(expec) if expec is not None else None,
                                            _lib.stream_ptr()), "mfr_loftr_fine_match")

    def gather_windows(self, feat_nhwc, img_ids, cell_ids, wc, stride, out=None):
        lib = _lib.load()
        Bimg, Hf, Wf, C = feat_nhwc.shape
        M = img_ids.numel()
        if out is None:
            out = torch.empty(M, self.W * self.W, C, dtype=torch.float32, device=feat_nhwc.device)
        assert out.is_contiguous() and out.shape == (M, self.W * self.W, C)
        _lib.check(lib.mfr_loftr_gather_windows(_lib.ptr(feat_nhwc), Bimg, Hf, Wf, C, _lib.ptr(img_ids), _lib.ptr(cell_ids), M,
                                                wc, stride, self.W, _lib.ptr(out), _lib.stream_ptr()), "mfr_loftr_gather_windows")
        return out

    # ------------------------------------------------------------------ transformer layers
    # Features live in xm [2, n, 2C]: xm[s, :, :C] = tokens of side s (0 = image0 rows, 1 = image1 rows), xm[s, :, C:] = the
    # layer's normalised message -- i.e. the MLP's cat([x, message]) operand exists in place and is never copied.
    def _layer(self, Lw, xm, src, attn, nb, L):
        """one LoFTREncoderLayer on the rows of xm [n, 2C] (updated in place) attending to src [n, C] (row-strided view)"""
        n, C2 = xm.shape
        C = C2 // 2
        x = xm[:, :C]
        if "lq" not in Lw:        # the five bias-free projections through csrc/gemm_split.hip, weights split / packed once per layer
            from .linear import SplitLinear
            Lw["lq"], Lw["lkv"], Lw["lm"] = SplitLinear(Lw["wq"]), SplitLinear(Lw["wkv"]), SplitLinear(Lw["wm"])
            Lw["l1"], Lw["l2"] = SplitLinear(Lw["w1"]), SplitLinear(Lw["w2"])
            Lw["lqkv"] = SplitLinear(torch.cat([Lw["wq"], Lw["wkv"]], 0))         # self layers: q | k | v from ONE pass over x (round 6)
        if src.data_ptr() == x.data_ptr() and src.stride() == x.stride() and src.shape == x.shape:
            qkv = Lw["lqkv"](x)                                                   # [n, 3 C]: the same rows feed q and k / v
            msg = attn(qkv.view(nb, L, 3 * C)[..., :C], qkv.view(nb, L, 3 * C)[..., C:])
        else:
            q = Lw["lq"](x)
            kv = Lw["lkv"](src)
            msg = attn(q.view(nb, L, C), kv.view(nb, L, 2 * C))
        if Lw["lm"].ln_fusable() and n > 0:
            # d_model 128 (the fine level): both LayerNorms run in the epilogue of the linear layer before them (round 6: the rows are 2.4 M x 128
            # floats per tensor, every separate pass is its full HBM read + write)
            Lw["lm"](msg.view(n, C), out=xm[:, C:], ln=Lw["n1"])
            if "mlp" not in Lw:
                from .linear import FusedMlpLn
                Lw["mlp"] = FusedMlpLn(Lw["l1"], Lw["l2"])
            Lw["mlp"](xm, out=x, ln=Lw["n2"], accumulate=True)                     # x += norm2(mlp([x | message])): hidden activations stay on chip
            return xm
        self.layernorm(Lw["lm"](msg.view(n, C)), Lw["n1"], xm[:, C:])
        hid = Lw["l1"](xm, relu=True)                                              # relu([x | message] W1^T) in the GEMM epilogue
        self.layernorm(Lw["l2"](hid), Lw["n2"], x, residual=x)                     # x += norm2(mlp)
        return xm

    @staticmethod
    def _torch_linear_attention(nhead):
        def attn(q, kv):
            B, L, D = q.shape
            k, v = kv.split(D, -1)
            Q = F.elu(q.view(B, L, nhead, -1)) + 1
            K = F.elu(k.reshape(B, L, nhead, -1)) + 1
            V = v.reshape(B, L, nhead, -1) / L
            KV = torch.einsum("nshd,nshv->nhdv", K, V)
            Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(1)) + 1e-6)
            return (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * L).reshape(B, L, D)
        return attn

    def _transformer(self, layers, xm, attn, nb, L):
        """xm [2, nb * L, 2C] in place: self layers on both sides at once, cross layers feat0 first, then feat1 against the
        UPDATED feat0 (upstream order)"""
        C = xm.shape[-1] // 2
        both = xm.view(-1, 2 * C)
        for l, Lw in enumerate(layers):
            if l % 2 == 0:
                self._layer(Lw, both, both[:, :C], attn, 2 * nb, L)
            else:
                self._layer(Lw, xm[0], xm[1][:, :C], attn, nb, L)
                self._layer(Lw, xm[1], xm[0][:, :C], attn, nb, L)
        return xm

    # ------------------------------------------------------------------ full forward
    @torch.no_grad()
    def __call__(self, images):
        """images [2B,1,H,W] (interleaved pairs; H, W multiples of 8) -> dict(pts0, pts1 [B,L0,2],
        n_corr [B], mconf [B,L0]) in the matcher's pixel frame (mkpts0_f / mkpts1_f)."""
        return self.fine_stage(self.coarse_stage(images))

    def coarse_stage(self, images):
        """backbone -> coarse transformer -> dual-softmax matching: everything BEFORE the match count is known.  Fixed shapes and
        no host synchronisation, so the batch-1 plugin path replays this stage from one captured HIP graph (nets/graph.py).
        Three steps with tensor interfaces (tools/loftr_stage_diff.py substitutes the oracle's tensors between them)."""
        fc, ff = self.backbone(images)
        xm = self.coarse_tokens(fc)
        self._transformer(self.coarse, xm, self.linear_attention, images.shape[0] // 2, fc.shape[2] * fc.shape[3])
        return self.coarse_tail(xm, ff, tuple(fc.shape[2:]), images.shape[2])

    def coarse_tokens(self, fc):
        """coarse map [2B,256,hc,wc] (+ positional encoding) -> xm [2, B * L0, 512]: NCHW -> token-major, pairs de-interleaved
        (side-major), one strided copy into the left half"""
        B2, C, hc, wc = fc.shape
        key = (hc, wc)
        if key not in self._pe:
            self._pe[key] = position_encoding_sine(C, hc, wc, fc.device)
        L0 = hc * wc
        xm = torch.empty(2, (B2 // 2) * L0, 2 * C, dtype=torch.float32, device=fc.device)
        _lib.check(_lib.load().mfr_nchw_to_rows(_lib.ptr(fc.contiguous()), _lib.ptr(self._pe[key].contiguous()), B2, C, L0, 1, _lib.ptr(xm), L0 * 2 * C, 2 * C,
                                                _lib.stream_ptr()), "mfr_nchw_to_rows")
        return xm

    def coarse_tail(self, xm, ff, hw, H):
        """transformed tokens xm + fine map ff [2B,128,Hf,Wf] -> dual-softmax matches, coarse keypoints, the fine map in NHWC"""
        hc, wc = hw
        L0 = hc * wc
        B = xm.shape[1] // L0
        B2 = 2 * B
        f0, f1 = xm[0].view(B, L0, 512)[..., :256], xm[1].view(B, L0, 512)[..., :256]     # row-strided views
        i_ids, j_ids, mconf, n = self.coarse_match_features(f0, f1, (hc, wc))
        scale = H // hc
        # coarse keypoints (padded layout)
        ii, jj = i_ids.long(), j_ids.long()
        valid = torch.arange(L0, device=xm.device)[None] < n[:, None]
        k0 = torch.stack([ii % wc, ii // wc], -1).float() * scale
        k1 = torch.stack([jj % wc, jj // wc], -1).float() * scale
        Cf, Hf, Wf = ff.shape[1:]
        if ff.permute(0, 2, 3, 1).is_contiguous() and not ff.is_contiguous():             # already NHWC memory (backbone, round 6)
            ff_nhwc = ff.permute(0, 2, 3, 1)
        else:
            ff_nhwc = torch.empty(B2, Hf, Wf, Cf, dtype=torch.float32, device=ff.device)    # [2B, Hf, Wf, 128]: LDS-tiled transpose (csrc/elementwise.hip)
            _lib.check(_lib.load().mfr_nchw_to_rows(_lib.ptr(ff.contiguous()), None, B2, Cf, Hf * Wf, 0, _lib.ptr(ff_nhwc), Hf * Wf * Cf, Cf, _lib.stream_ptr()),
                       "mfr_nchw_to_rows")
        return dict(xm=xm, ff_nhwc=ff_nhwc, i_ids=i_ids, j_ids=j_ids, ii=ii, jj=jj, mconf=mconf, n=n, valid=valid, k0=k0, k1=k1,
                    hc=hc, wc=wc, H=H)

    def fine_stage(self, c):
        """the matched cells' 5x5 windows -> fine transformer -> sub-pixel expectation (data-dependent sizes: one host
        synchronisation for the match count)"""
        xm, ff_nhwc, i_ids, j_ids, ii, jj, mconf, n, valid, k0, k1 = (c[k] for k in ("xm", "ff_nhwc", "i_ids", "j_ids", "ii", "jj", "mconf",
                                                                                     "n", "valid", "k0", "k1"))
        hc, wc, H = c["hc"], c["wc"], c["H"]
        B, L0 = valid.shape
        dev = valid.device
        f0, f1 = xm[0].view(B, L0, 512)[..., :256], xm[1].view(B, L0, 512)[..., :256]
        b_ids, slot = torch.where(valid)                                            # (host sync: match count)
        M = b_ids.numel()
        pts1 = k1.clone()
        if M > 0:
            mi, mj = ii[b_ids, slot], jj[b_ids, slot]
            Hf = ff_nhwc.shape[1]
            stride = Hf // hc
            WW = self.W * self.W
            # down_proj on the matched coarse features, merge_feat(cat[window, coarse]) = window Wa^T + (coarse Wb^T + b): the coarse half is
            # constant over the 25 taps.  All three products through csrc/gemm_split.hip (rounds 1-4: library GEMMs); the window product lands
            # in the left half of the fine transformer's [x | message] operand, the per-window constant is added in place.
            if self._fine_lin is None:
                from .linear import SplitLinear
                wmf, bmf = self.merge_feat
                self._fine_lin = (SplitLinear(self.down_proj[0], self.down_proj[1]), SplitLinear(wmf[:, 128:].contiguous(), bmf),
                                  SplitLinear(wmf[:, :128].contiguous()))
            lin_down, lin_coarse, lin_win = self._fine_lin
            cw = lin_coarse(lin_down(torch.cat([f0[b_ids, mi], f1[b_ids, mj]], 0)))
            xf = torch.empty(2, M * WW, 256, dtype=torch.float32, device=dev)
            # round 6: the windows' tokens are read from the NHWC fine map inside the product and cw is added per window in its epilogue -- no gather
            # kernel, no [2 M, 25, 128] window tensor, no broadcast add (upstream FinePreprocess: unfold -> gather -> merge_feat)
            lin_win.windows(ff_nhwc, torch.cat([2 * b_ids, 2 * b_ids + 1]).int(), torch.cat([mi, mj]).int(), wc, stride, self.W,
                            out=xf.view(2 * M * WW, 256)[:, :128], window_bias=cw)
            self._transformer(self.fine, xf, self.fine_attention if self.W == 5 else self._torch_linear_attention(8), M, WW)
            # FineMatching: centre-feature correlation, softmax, spatial expectation and the sub-pixel update in one kernel
            self.fine_match(xf, M, (b_ids * L0 + slot).int(), k1, pts1, float((self.W // 2) * (H // Hf)))
        zero = torch.zeros_like(k0)
        return dict(pts0=torch.where(valid[..., None], k0, zero).contiguous(), pts1=torch.where(valid[..., None], pts1, zero).contiguous(),
                    n_corr=n, mconf=mconf, i_ids=i_ids, j_ids=j_ids)
