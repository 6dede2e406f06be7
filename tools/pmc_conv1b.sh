# SQ counters of the conv1b launch of wino_conv3x3_kernel (64->64 channels, 64 images 720x540, pooled): two rocprofv3
# --pmc passes (8 SQ slots per pass), summarised into gpurun_out/sq/pmc_wino_sq.json.
mkdir -p gpurun_out/sq
cat > /tmp/run_conv1b.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
import mapfree_reloc_amd as m
from mapfree_reloc_amd import _lib
lib = _lib.load(require_gpu=True); dev = torch.device("cuda")
B, ci, co, H, W = 64, 64, 64, 720, 540
x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.04; b = torch.randn(co, device=dev)
u = torch.empty(lib.mfr_wino_filter_bytes(ci, co) // 4, device=dev)
lib.mfr_wino_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u), _lib.stream_ptr())
y = torch.empty(B, co, H // 2, W // 2, device=dev)
for _ in range(4):
    assert lib.mfr_conv3x3_wino(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b), None, B, ci, co, H, W, 1, 1, _lib.ptr(y), _lib.stream_ptr()) == 0
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES -d /root/repo/gpurun_out/sq/p1 -o run --output-format csv -- python /tmp/run_conv1b.py > /root/repo/gpurun_out/sq/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d /root/repo/gpurun_out/sq/p2 -o run --output-format csv -- python /tmp/run_conv1b.py > /root/repo/gpurun_out/sq/p2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d /root/repo/gpurun_out/sq/p3 -o run --output-format csv -- python /tmp/run_conv1b.py > /root/repo/gpurun_out/sq/p3.log 2>&1
cd /root/repo
python - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(list); dur = []
for f in glob.glob("gpurun_out/sq/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wino_conv3x3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/sq/p1/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "wino_conv3x3" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
out = {k: sum(v) / len(v) for k, v in acc.items()}
out["launch_ms_under_pmc"] = sum(dur) / max(len(dur), 1)
out["kernel"] = "wino_conv3x3_shared_kernel<pool, pair loads> (default variant), conv1b: 64->64 ch, 64 images 720x540"
out["note"] = "SQ_* cycle counters are summed over all SIMDs; WAVE_CYCLES / WAIT_* / ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)"
json.dump(out, open("gpurun_out/sq/pmc_wino_sq.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -2 gpurun_out/sq/p3.log
find gpurun_out/sq -name '*.csv' -size +1M -delete
