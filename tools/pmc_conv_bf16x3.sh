# PMC passes on the conv1b launch (64->64 channels, 64 images 720x540, pooled) of the exact-fp32 and the bf16x3 Winograd kernels:
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes), L2 hit rate, SQ busy / wait breakdown -> gpurun_out/pmc3/summary.json
mkdir -p gpurun_out/pmc3
cat > /tmp/run_conv1b3.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
import mapfree_reloc_amd as m
from mapfree_reloc_amd import _lib
lib = _lib.load(require_gpu=True); dev = torch.device("cuda")
B, ci, co, H, W = 64, 64, 64, 720, 540
x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.04; b = torch.randn(co, device=dev)
u = torch.empty(lib.mfr_wino_filter_bytes(ci, co) // 4, device=dev)
lib.mfr_wino_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u), _lib.stream_ptr())
u3 = torch.empty(lib.mfr_wino_bf16x3_filter_bytes(ci, co), dtype=torch.uint8, device=dev)
lib.mfr_wino_bf16x3_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u3), _lib.stream_ptr())
y = torch.empty(B, co, H // 2, W // 2, device=dev)
for _ in range(3):
    assert lib.mfr_conv3x3_wino(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b), None, B, ci, co, H, W, 1, 1, _lib.ptr(y), _lib.stream_ptr()) == 0
    assert lib.mfr_conv3x3_wino_bf16x3(_lib.ptr(x), _lib.ptr(u3), _lib.ptr(b), None, B, ci, co, H, W, 1, 1, _lib.ptr(y), _lib.stream_ptr()) == 0
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
i=0
PASSES=${PMC_PASSES:-6}
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  if [ $i -ge $PASSES ]; then break; fi
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs -d /root/repo/gpurun_out/pmc3/p$i -o run --output-format csv -- python /tmp/run_conv1b3.py > /root/repo/gpurun_out/pmc3/p$i.log 2>&1
done
cd /root/repo
python - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc3/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = "bf16x3" if "wino_bf16x3_kernel" in r["Kernel_Name"] else "exact_fp32" if "wino_conv3x3" in r["Kernel_Name"] else None
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc3/p1/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = "bf16x3" if "wino_bf16x3_kernel" in r["Kernel_Name"] else "exact_fp32" if "wino_conv3x3" in r["Kernel_Name"] else None
        if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
out = {}
for k in acc:
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    d["launch_ms_under_pmc"] = sum(dur[k]) / max(len(dur[k]), 1)
    if "FETCH_SIZE" in d: d["hbm_read_GB_corrected(2x)"] = 2 * d["FETCH_SIZE"] * 1024 / 1e9
    if "WRITE_SIZE" in d: d["hbm_write_GB"] = d["WRITE_SIZE"] * 1024 / 1e9
    if "TCC_HIT_sum" in d: d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    out[k] = d
out["algorithmic_bytes"] = {"input": 64 * 64 * 720 * 540 * 4, "output_pooled": 64 * 64 * 360 * 270 * 4}
json.dump(out, open("gpurun_out/pmc3/summary.json", "w"), indent=1)
if "bf16x3" in out and "FETCH_SIZE" in out["bf16x3"] and "WRITE_SIZE" in out["bf16x3"]:
    d = out["bf16x3"]
    json.dump({"kernel": "wino_bf16x3_kernel<true> (conv1b: 64->64 channels, 64 images 720x540, pooled output)", "FETCH_SIZE_KB": d["FETCH_SIZE"], "WRITE_SIZE_KB": d["WRITE_SIZE"],
               "hbm_bytes_per_launch": 2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024,
               "algorithmic_bytes_per_launch": out["algorithmic_bytes"]["input"] + out["algorithmic_bytes"]["output_pooled"], "pairs_per_step": 32,
               "l2_hit_rate": d.get("l2_hit_rate"), "launch_ms_under_pmc": d["launch_ms_under_pmc"],
               "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; separate --pmc passes",
               "command": "tools/pmc_conv_bf16x3.sh (rocprofv3 --kernel-trace --pmc <one counter group per pass> on the isolated conv1b launch)", "round": 3},
              open("gpurun_out/r03_pmc_conv1b.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -3 gpurun_out/pmc3/p6.log
find gpurun_out/pmc3 -name '*.csv' -size +1M -delete
