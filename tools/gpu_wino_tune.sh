set -x
mkdir -p gpurun_out/wino
timeout 200 python tools/diag_wino.py --no-time > gpurun_out/wino/correct.txt 2>&1
timeout 200 python tools/diag_wino_abl.py > gpurun_out/wino/abl.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS -d /root/repo/gpurun_out/wino/pmc1 -o run --output-format csv -- python /root/repo/tools/diag_wino_abl.py 0 > /root/repo/gpurun_out/wino/pmc1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT -d /root/repo/gpurun_out/wino/pmc2 -o run --output-format csv -- python /root/repo/tools/diag_wino_abl.py 0 > /root/repo/gpurun_out/wino/pmc2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d /root/repo/gpurun_out/wino/pmc3 -o run --output-format csv -- python /root/repo/tools/diag_wino_abl.py 0 > /root/repo/gpurun_out/wino/pmc3.log 2>&1
cd /root/repo
cat gpurun_out/wino/correct.txt | tail -9; cat gpurun_out/wino/abl.txt
for d in pmc1 pmc2 pmc3; do f=$(find gpurun_out/wino/$d -name '*counter_collection.csv' | head -1); echo $d $f; python - "$f" <<'PY'
import sys, csv, collections
f = sys.argv[1]
if not f: sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r.get("Kernel_Name", "")[:40]
    if "wino_conv" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    n[(k, r["Counter_Name"])] += 1
for k in acc:
    for c, v in acc[k].items(): print(k, c, v / n[(k, c)], n[(k, c)])
PY
done
find gpurun_out/wino -name '*.csv' -size +2M -delete
