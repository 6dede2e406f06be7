"""Stage-by-stage bisect of the LoFTR HIP path against the CPU oracle.

For each census pair (images.synthetic_batch seeds, hard = 0 / 1 / 2) and each ARITHMETIC variant of the HIP path
  f16x2          the shipped default (options SPLIT = f16x2, two-sweep dual softmax with v_exp_f32)
  f16x2+dsm4     the same with the four-sweep dual-softmax kernels of round 1 (precise expf, IEEE divisions)
  bf16x3         the exact three-term split (library convolutions for the strided / 1x1 layers, library similarity product)
it reports
  (1) per-stage deviation with the ORACLE's tensors fed to that stage (so one stage's error is not another's input):
        backbone (coarse / fine maps), coarse transformer (tokens), dual softmax (match set, confidences), fine stage (sub-pixel offsets);
  (2) the FINAL coordinates and the pose (oracle solver on the HIP match list -- the HIP solver is bit-equal to it, tests/test_gpu_emat_parity.py)
      when the HIP path takes over at the input / after the backbone / after the transformer / after the matching.
Usage: python tools/loftr_stage_diff.py [--seeds 5007 ...] [--hard 2] [--variants f16x2 bf16x3] [--f64] [--out gpurun_out/loftr_stage_diff.json]"""
import argparse
import functools
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapfree_reloc_amd  # noqa: E402,F401
from mapfree_reloc_amd import images as IM, options  # noqa: E402
from mapfree_reloc_amd.nets import weights as WT  # noqa: E402
from mapfree_reloc_amd.nets.loftr import LoFTRHIP  # noqa: E402
from oracle import pipeline_ref as PR, loftr_ref as LR  # noqa: E402
from tools.loftr_sensitivity import pose_delta, solve  # noqa: E402

DEV = "cuda:0"


@torch.no_grad()
def oracle_stages(m, im0, im1):
    """LoFTRRef.forward (oracle/loftr_ref.py) with every intermediate kept"""
    fc, ff = m.backbone(torch.cat([im0, im1], 0))
    hc, wc = fc.shape[2:]
    pe = LR.position_encoding_sine(256, hc, wc)
    t0 = (fc[:1] + pe[None]).flatten(2).transpose(1, 2)
    t1 = (fc[1:] + pe[None]).flatten(2).transpose(1, 2)
    t0, t1 = m.loftr_coarse(t0, t1)
    cm = LR.coarse_matching(t0, t1, (hc, wc), (hc, wc), scale=im0.shape[2] // hc)
    b, i, j = cm["b_ids"], cm["i_ids"], cm["j_ids"]
    W = m.W
    stride = ff.shape[2] // hc
    C = ff.shape[1]
    u = F.unfold(ff, kernel_size=(W, W), stride=stride, padding=W // 2).view(2, C, W * W, -1).permute(0, 3, 2, 1)
    f0u, f1u = u[0][i], u[1][j]
    fcw = m.fine_preprocess.down_proj(torch.cat([t0[b, i], t1[b, j]], 0))
    fcf = m.fine_preprocess.merge_feat(torch.cat([torch.cat([f0u, f1u], 0), fcw[:, None].expand(-1, W * W, -1)], -1))
    f0u, f1u = torch.chunk(fcf, 2, dim=0)
    f0u, f1u = m.loftr_fine(f0u, f1u)
    heat = torch.softmax(torch.einsum('mc,mrc->mr', f0u[:, W * W // 2, :], f1u) / C ** .5, dim=1).view(-1, W, W)
    coords = LR.spatial_expectation2d(heat)
    pts = torch.cat([cm["mkpts0_c"], cm["mkpts1_c"] + coords * (W // 2) * (im0.shape[2] // ff.shape[2])], 1).numpy()
    return dict(fc=fc, ff=ff, t0=t0, t1=t1, i=i, j=j, mconf=cm["mconf"], coords=coords, pts=pts, hw=(hc, wc))


def _dev(t):
    return t.to(DEV).contiguous()


def _stat(got, want):
    d = (got.double() - want.double()).abs()
    return dict(max_abs=float(d.max()), rms=float(d.pow(2).mean().sqrt()), ref_rms=float(want.double().pow(2).mean().sqrt()),
                ref_max=float(want.abs().max()))


def _pts(out):
    n = int(out["n_corr"][0])
    return torch.cat([out["pts0"][0, :n], out["pts1"][0, :n]], 1).cpu().numpy()


def _coord_delta(pts, ref):
    ka = {(int(r[0]), int(r[1]), int(r[2]) // 8, int(r[3]) // 8): r for r in ref}
    kb = {(int(r[0]), int(r[1]), int(r[2]) // 8, int(r[3]) // 8): r for r in pts}
    ca = {k[:2] for k in ka}; cb = {k[:2] for k in kb}
    common = sorted(set(ka) & set(kb))
    d = np.array([np.abs(ka[k].astype(np.float64) - kb[k]).max() for k in common]) if common else np.zeros(1)
    return dict(n_ref=len(ref), n_hip=len(pts), same_coarse_cell_pair=len(common), only_ref=len(ca - cb), only_hip=len(cb - ca),
                coords_differ=int((d > 0).sum()), max_px=float(d.max()), p99_px=float(np.quantile(d, 0.99)), rms_px=float(np.sqrt((d ** 2).mean())))


@torch.no_grad()
def hip_variant(name, o, images, sb, seed, base_pose, sd):
    options.reset()
    options.set("SPLIT", "bf16x3" if name == "bf16x3" else "f16x2")
    hip = LoFTRHIP(sd, DEV)
    if name.endswith("dsm4"):
        hip.coarse_match = functools.partial(hip.coarse_match, variant=1)
    hc, wc = o["hw"]
    L0 = hc * wc
    H = images.shape[2]
    rec = {}

    def finish(pts):
        r = solve(pts, sb, 0, seed) if len(pts) >= 5 else None
        out = _coord_delta(pts, o["pts"])
        out["pose"] = pose_delta(base_pose, r) if (r is not None and base_pose is not None) else None
        return out

    # ---- (1) stage deviations on the oracle's inputs
    fc, ff = hip.backbone(images)
    rec["backbone_coarse_map"] = _stat(fc.cpu(), o["fc"]); rec["backbone_fine_map"] = _stat(ff.cpu(), o["ff"])
    if "fc64" in o:         # both fp32 evaluations against the SAME network in float64: is the HIP path as close to the truth as the reference's fp32 is?
        rec["backbone_vs_float64"] = dict(hip_coarse=_stat(fc.cpu(), o["fc64"]), hip_fine=_stat(ff.cpu(), o["ff64"]),
                                          oracle_f32_coarse=_stat(o["fc"], o["fc64"]), oracle_f32_fine=_stat(o["ff"], o["ff64"]))
    fc_o, ff_o = _dev(o["fc"]), _dev(o["ff"])
    xm = hip.coarse_tokens(fc_o)
    hip._transformer(hip.coarse, xm, hip.linear_attention, 1, L0)
    tok = torch.stack([xm[0][:, :256], xm[1][:, :256]]).cpu()
    rec["transformer_tokens_on_oracle_maps"] = _stat(tok, torch.cat([o["t0"], o["t1"]]))
    xm_o = torch.zeros(2, L0, 512, device=DEV)
    xm_o[0, :, :256] = _dev(o["t0"][0]); xm_o[1, :, :256] = _dev(o["t1"][0])
    c = hip.coarse_tail(xm_o.clone(), ff_o, (hc, wc), H)
    n = int(c["n"][0])
    mh = {(int(a), int(b)): float(cf) for a, b, cf in zip(c["i_ids"][0, :n].tolist(), c["j_ids"][0, :n].tolist(), c["mconf"][0, :n].tolist())}
    mo = {(int(a), int(b)): float(cf) for a, b, cf in zip(o["i"].tolist(), o["j"].tolist(), o["mconf"].tolist())}
    both = sorted(set(mh) & set(mo))
    rec["dual_softmax_on_oracle_tokens"] = dict(
        n_ref=len(mo), n_hip=len(mh), differ=len(set(mh) ^ set(mo)),
        conf_of_differing=[round({**mh, **mo}[k], 6) for k in sorted(set(mh) ^ set(mo))][:8],
        max_rel_conf=float(max(abs(mh[k] - mo[k]) / mo[k] for k in both)) if both else None)
    # fine stage on the oracle's tokens, maps AND matches
    M = len(o["i"])
    ii = torch.zeros(1, L0, dtype=torch.long, device=DEV); jj = torch.zeros_like(ii)
    ii[0, :M] = _dev(o["i"]); jj[0, :M] = _dev(o["j"])
    nn = torch.tensor([M], dtype=torch.int32, device=DEV)
    valid = torch.arange(L0, device=DEV)[None] < nn[:, None]
    sc = H // hc
    c2 = dict(c, xm=xm_o.clone(), i_ids=ii.int(), j_ids=jj.int(), ii=ii, jj=jj, n=nn, valid=valid,
              k0=torch.stack([ii % wc, ii // wc], -1).float() * sc, k1=torch.stack([jj % wc, jj // wc], -1).float() * sc,
              mconf=torch.zeros(1, L0, device=DEV))
    pf = _pts(hip.fine_stage(c2))
    rec["take_over_after_matching"] = finish(pf)
    off_o = (o["coords"].double().numpy() * 4.0)                                     # same order as the oracle's list (ascending i)
    k1o = np.stack([(o["j"].numpy() % wc) * sc, (o["j"].numpy() // wc) * sc], 1)
    off_h = pf[:, 2:].astype(np.float64) - k1o
    rec["fine_offsets_on_oracle_inputs_px"] = dict(max_abs=float(np.abs(off_h - off_o).max()), rms=float(np.sqrt(((off_h - off_o) ** 2).mean())),
                                                   note="offset = pts1 - coarse cell (f32 sum with a coordinate up to 720: ulp 6e-5 px)")
    # ---- (2) take-over points
    rec["take_over_after_transformer"] = finish(_pts(hip.fine_stage(hip.coarse_tail(xm_o.clone(), ff_o, (hc, wc), H))))
    xm = hip.coarse_tokens(fc_o)
    hip._transformer(hip.coarse, xm, hip.linear_attention, 1, L0)
    rec["take_over_after_backbone"] = finish(_pts(hip.fine_stage(hip.coarse_tail(xm, ff_o, (hc, wc), H))))
    rec["whole_hip_path"] = finish(_pts(hip(images)))
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[5007])
    ap.add_argument("--hard", type=int, default=2)
    ap.add_argument("--variants", nargs="+", default=["f16x2", "f16x2+dsm4", "bf16x3"])
    ap.add_argument("--f64", action="store_true", help="also evaluate the oracle's backbone in float64 and report HIP-vs-f64 next to oracle-f32-vs-f64")
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "loftr_stage_diff.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    model = PR._nets("loftr")
    sd = WT.loftr_state_dict()
    res = dict(hard=a.hard, pairs=[])
    for s in a.seeds:
        sb = IM.synthetic_batch([s], hard=a.hard)
        im0, im1 = PR._t(sb["images"][0, 0]), PR._t(sb["images"][1, 0])
        pw = im0.shape[3] % 8                                                        # quirk Q3 (matchers.py:41-46)
        im0, im1 = F.pad(im0, (0, pw)), F.pad(im1, (0, pw))
        o = oracle_stages(model, im0, im1)
        if a.f64:
            import copy
            with torch.no_grad():
                o["fc64"], o["ff64"] = copy.deepcopy(model.backbone).double()(torch.cat([im0, im1], 0).double())
        base = solve(o["pts"], sb, 0, s)
        rec = dict(seed=s, matches=len(o["pts"]), emat_inliers=None if base is None else base["n_emat"],
                   inlier_fraction=None if base is None else round(base["n_emat"] / len(o["pts"]), 4), variants={})
        images = torch.cat([im0, im1], 0).to(DEV)
        for v in a.variants:
            rec["variants"][v] = hip_variant(v, o, images, sb, s, base, sd)
            print(s, v, json.dumps(rec["variants"][v]), flush=True)
        res["pairs"].append(rec)
    options.reset()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
