# PMC passes on the conv1b launch (64->64 channels, 64 images 720x540, pooled) of the bf16x3 Winograd kernels: the one-wavefront-per-SIMD
# kernel (variant 32) and the round-4 kernel (variant ${W2_VARIANT:-2}).  Separate passes per counter group -> gpurun_out/pmc4/summary.json
mkdir -p gpurun_out/pmc4
cat > /tmp/run_conv1b4.py <<PY
import sys, torch
sys.path.insert(0, "/root/repo")
import mapfree_reloc_amd as m
from mapfree_reloc_amd import _lib
lib = _lib.load(require_gpu=True); dev = torch.device("cuda")
B, ci, co, H, W = 64, 64, 64, 720, 540
x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.04; b = torch.randn(co, device=dev)
u3 = torch.empty(lib.mfr_wino_bf16x3_filter_bytes(ci, co), dtype=torch.uint8, device=dev)
lib.mfr_wino_bf16x3_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u3), _lib.stream_ptr())
y = torch.empty(B, co, H // 2, W // 2, device=dev)
for _ in range(3):
    for v in (32, ${W2_VARIANT:-2}):
        assert lib.mfr_conv3x3_wino_bf16x3_variant(_lib.ptr(x), _lib.ptr(u3), _lib.ptr(b), None, B, ci, co, H, W, 1, 1, v, _lib.ptr(y), _lib.stream_ptr()) == 0
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
i=0
PASSES=${PMC_PASSES:-7}
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"; do
  if [ $i -ge $PASSES ]; then break; fi
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs -d /root/repo/gpurun_out/pmc4/p$i -o run --output-format csv -- python /tmp/run_conv1b4.py > /root/repo/gpurun_out/pmc4/p$i.log 2>&1
done
cd /root/repo
python - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
def key(n):
    return "w2" if "w2_kernel" in n or "p8_kernel" in n else "one_wave" if "wino_bf16x3_kernel" in n else None
for f in glob.glob("gpurun_out/pmc4/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc4/p1/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
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
json.dump(out, open("gpurun_out/pmc4/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -3 gpurun_out/pmc4/p7.log
find gpurun_out/pmc4 -name '*.csv' -size +1M -delete
