# round-2 session-3, second GPU pass: flat-mapped bilinear upsampling (regression decoder + LoFTR FPN), training-step options A/B,
# refreshed LoFTR numbers + kernel trace, and the HIP-graph experiment with every GEMM off the hipBLASLt "UserArgs" path
set -x
O=gpurun_out/r2g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rpr_parity.py tests/test_gpu_loftr_parity.py -m gpu -q > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
timeout 300 python bench.py --config rpr_train --steps 20 --warmup 3 > $O/bench_rpr.json 2> $O/bench_rpr.err
timeout 200 python bench.py --config rpr_train --steps 20 --warmup 3 --no-cpu-baseline --rpr-opts siamese > $O/bench_rpr_siamese.json 2> $O/bench_rpr_siamese.err
PYTORCH_MIOPEN_SUGGEST_NHWC=1 PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM=1 timeout 200 python bench.py --config rpr_train --steps 20 --warmup 3 --no-cpu-baseline --rpr-opts channels_last > $O/bench_rpr_cl.json 2> $O/bench_rpr_cl.err
timeout 200 python bench.py --config rpr_train --steps 20 --warmup 3 --no-cpu-baseline --batch 32 > $O/bench_rpr_b32.json 2> $O/bench_rpr_b32.err
timeout 300 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_loftr.json 2> $O/bench_loftr.err
cd /tmp && export TMPDIR=/tmp
for C in rpr_train loftr_emat; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_$C -o run -- python /root/repo/bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline > /root/repo/$O/prof_$C.json 2> /root/repo/$O/prof_$C.err
  DB=$(find /root/repo/$O/prof_$C -name '*.db' | head -1)
  python /root/repo/tools/rocprof_summary.py "$DB" /root/repo/$O/kernel_stats_$C.csv
  rm -rf /root/repo/$O/prof_$C
done
cd /root/repo
for Bp in 8 32; do
  DISABLE_ADDMM_CUDA_LT=1 TORCH_BLAS_PREFER_HIPBLASLT=0 timeout 150 python tools/diag_graph.py $Bp > $O/graph_nolt_b$Bp.log 2>&1; echo "rc=$?" >> $O/graph_nolt_b$Bp.log
done
for f in $O/pytest_new.log $O/graph_nolt_b8.log $O/graph_nolt_b32.log; do tail -n 4 $f; done
python - <<PY
import json
for f in ("bench_rpr","bench_rpr_siamese","bench_rpr_cl","bench_rpr_b32","bench_loftr"):
    try:
        d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
    except Exception as e: print(f, "ERR", e, open("$O/"+f+".err").read()[-600:])
PY
