"""DESIGN.md = docs/DESIGN.template.md with its tables and numbers filled from the committed records under profiles/ (so that text and records cannot
disagree: VERDICT r5 weak 1a / 8).   python tools/fill_design.py"""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, *a)
rd = lambda name: open(P("docs", name)).read().rstrip("\n")


def stats(cfg):
    rows = list(csv.DictReader(open(P("profiles", f"r06_bench_{cfg}_kernel_stats.csv"))))
    meta = json.load(open(P("profiles", f"r06_bench_{cfg}_kernel_stats.json")))
    def ms(*subs, exclude=()):
        return sum(float(r["MsPerStep"]) for r in rows if any(s in r["Name"] for s in subs) and not any(e in r["Name"] for e in exclude))
    return rows, meta, ms


f2 = lambda v: f"{v:.2f}"
bench = json.loads(open(P("profiles", "r06_bench_default.json")).read().strip().splitlines()[-1])
sec = {("loftr" if "LoFTR" in s["metric"] else "rpr"): s for s in bench.get("secondary", [])}
_, sg_meta, sg = stats("sg_pnp")
_, lo_meta, lo = stats("loftr_emat")
roof = bench["roofline"]
cpu = bench["cpu_baseline"]
try:
    hostfed = json.load(open(P("profiles", "r06_fused_split_1gpu.json")))
    hf = hostfed.get("sg_pnp", {}).get("pairs_per_s") or next((v.get("pairs_per_s") for v in hostfed.values() if isinstance(v, dict) and "pairs_per_s" in v), None)
except Exception:
    hf = None
sub = {
    "SG_PPS": f"{bench['value']:.1f}", "SG_MS": f2(bench["ms_per_step"]), "SG_KMS": f2(sg_meta["kernel_ms_per_step"]), "SG_LAUNCHES": f"{sg_meta['launches_per_step']:.0f}",
    "LOFTR_PPS": f"{sec['loftr']['value']:.1f}", "LOFTR_MS": f2(sec["loftr"]["ms_per_step"]), "LOFTR_KMS": f2(lo_meta["kernel_ms_per_step"]),
    "LOFTR_LAUNCHES": f"{lo_meta['launches_per_step']:.0f}",
    "RPR_PPS": f"{sec['rpr']['value']:.1f}", "RPR_MS": f2(sec["rpr"]["ms_per_step"]),
    "HOSTFED": f"{hf:.0f}" if hf else "(see record)",
    "C1_MS": f2(roof["avg_launch_ms"]), "C1_TF": f"{roof['achieved']:.0f}", "C1_FRAC": f"{roof['frac']:.3f}", "C1_PIPE": f"{roof['mfma_pipe_frac']:.3f}",
    "CPU_PPS": f"{cpu['value']:.2f}", "CPU_CORES": str(cpu["cores"]),
    "SG_C1": f2(sg("wino_split_c1")), "SG_DCONV": f2(sg("conv_direct")), "SG_ATT": f2(sg("sg_attention")), "SG_GEMM": f2(sg("gemm_split")),
    "SG_SINK": f2(sg("sg_sweep", "sg_colmerge", "sg_rowmax", "sg_colmax", "sg_match")), "SG_SPPOST": f2(sg("sp_nms", "sp_select", "sp_sample", "sp_scoremap")),
    "SG_IGEMM": f2(sg("conv_igemm")), "SG_PNP": f2(sg("pnp_")),
    "LOFTR_DCONV": f2(lo("conv_direct")), "LOFTR_GEMM": f2(lo("gemm_split_d")), "LOFTR_MLPLN": f2(lo("mlp_ln")), "LOFTR_FATT": f2(lo("fine_attention")),
    "LOFTR_IGEMM": f2(lo("conv_igemm")), "LOFTR_LA": f2(lo("la_kv", "la_out", "la_fold")), "LOFTR_DSM": f2(lo("dsm_")), "LOFTR_EMAT": f2(lo("emat_", "scale_")),
}
census = subprocess.check_output([sys.executable, P("tools", "census_table.py")], text=True).rstrip("\n")
out = open(P("docs", "DESIGN.template.md")).read()
parts = {"CENSUS_TABLE_PLACEHOLDER": census, "KERNEL_TABLE_PLACEHOLDER": rd("_design_kernel_table.md"), "SECTION44_PLACEHOLDER": rd("_design_44.md"),
         "SECTION5_PLACEHOLDER": rd("_design_58.md").split("## 8. What comes next")[0].rstrip("\n"),
         "SECTION8_PLACEHOLDER": "## 8. What comes next" + rd("_design_58.md").split("## 8. What comes next")[1], "VERDICT_PLACEHOLDER": rd("_design_9.md")}
for k, v in parts.items():
    assert k in out, k
    out = out.replace(k, v)
for k in sorted(sub, key=len, reverse=True):
    out = out.replace(k, sub[k])
open(P("DESIGN.md"), "w").write(out)
left = [w for w in ("PLACEHOLDER", "SG_", "LOFTR_", "C1_", "RPR_", "CPU_", "HOSTFED") if w in out]
print("DESIGN.md written,", len(out), "bytes; unresolved markers:", left, "; longest line", max(len(l) for l in out.splitlines()))
