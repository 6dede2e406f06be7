# round 4: rocprofv3 kernel statistics of the TIMED steps of the three bench configurations (warm-up and library search excluded:
# tools/kernel_stats_timed.py) and the HBM-traffic PMC passes of LoFTR's dominant convolution launch -> gpurun_out/r04_*
O=gpurun_out
run_cfg() {   # config anchor init_steps warmup steps
  c=$1
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d /root/repo/$O/prof4_$c -o run --output-format csv -- python /root/repo/bench.py --config $c --steps $5 --warmup $4 --no-cpu-baseline --no-secondary > /root/repo/$O/prof4_$c.json 2> /root/repo/$O/prof4_$c.err )
  python tools/kernel_stats_timed.py $O/prof4_$c/run_kernel_trace.csv $O/r04_bench_${c}_kernel_stats.csv --anchor $2 --warmup $(( $3 + $4 )) --steps $5 --json $O/r04_bench_${c}_kernel_stats.json | cut -c1-400
  tail -1 $O/prof4_$c.json | cut -c1-200
  rm -f $O/prof4_$c/run_kernel_trace.csv
}
run_cfg sg_pnp pnp_select 6 3 10
run_cfg loftr_emat emat_select 3 3 6
run_cfg rpr_train cw_fwd 3 3 10
# ---- PMC: LoFTR layer1_outconv2.0 (196 -> 196 channels at 360x272, 32 images = 16 pairs), the kernel nets/conv.py picks (variant 0)
cat > /tmp/run_l1out2.py <<PY
import sys, torch
sys.path.insert(0, "/root/repo")
import mapfree_reloc_amd as m
from mapfree_reloc_amd import _lib
lib = _lib.load(require_gpu=True); dev = torch.device("cuda")
B, ci, co, H, W = 32, 196, 196, 360, 272
x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.02; b = torch.randn(co, device=dev)
u3 = torch.empty(lib.mfr_wino_bf16x3_filter_bytes(ci, co), dtype=torch.uint8, device=dev)
lib.mfr_wino_bf16x3_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u3), _lib.stream_ptr())
y = torch.empty(B, co, H, W, device=dev)
for _ in range(3):
    assert lib.mfr_conv3x3_wino_bf16x3(_lib.ptr(x), _lib.ptr(u3), _lib.ptr(b), None, B, ci, co, H, W, 2, 0, _lib.ptr(y), _lib.stream_ptr()) == 0
torch.cuda.synchronize()
PY
mkdir -p $O/pmc5
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs -d /root/repo/$O/pmc5/p$i -o run --output-format csv -- python /tmp/run_l1out2.py > /root/repo/$O/pmc5/p$i.log 2>&1
done
cd /root/repo
python - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(list); dur = []; name = None
for f in glob.glob("gpurun_out/pmc5/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wino_bf16x3" in r["Kernel_Name"] and "filter" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); name = r["Kernel_Name"]
for f in glob.glob("gpurun_out/pmc5/p1/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "wino_bf16x3" in r["Kernel_Name"] and "filter" not in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
d = {c: sum(v) / len(v) for c, v in acc.items()}
alg = 32 * 196 * 360 * 272 * 4 * 2
out = {"kernel": (name or "")[:60] + " (LoFTR layer1_outconv2.0: 196 -> 196 channels at 360x272, 32 images, LeakyReLU)", "FETCH_SIZE_KB": d.get("FETCH_SIZE"), "WRITE_SIZE_KB": d.get("WRITE_SIZE"),
       "hbm_bytes_per_launch": 2 * d.get("FETCH_SIZE", 0) * 1024 + d.get("WRITE_SIZE", 0) * 1024, "algorithmic_bytes_per_launch": alg, "pairs_per_step": 16,
       "l2_hit_rate": d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]) if "TCC_HIT_sum" in d else None,
       "launch_ms_under_pmc": sum(dur) / max(len(dur), 1),
       "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; separate --pmc passes",
       "command": "tools/gpu_r4_profiles.sh (rocprofv3 --kernel-trace --pmc <one counter group per pass> on the isolated launch)", "round": 4}
json.dump(out, open("gpurun_out/r04_pmc_loftr_l1out2.json", "w"), indent=1)
print(json.dumps(out)[:500])
PY
find gpurun_out/pmc5 gpurun_out/prof4_* -name '*.csv' -size +1M -delete 2>/dev/null
