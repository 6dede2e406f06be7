# round-2 session-3 GPU pass: new tests first (f-4 kernel + trainer, two-sweep dual softmax), then the rest of the suite, the three
# bench configs, kernel traces of the training step, and the rocBLAS / HIP-graph experiment last (it may fault)
set -x
O=gpurun_out/r2f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rpr_parity.py tests/test_gpu_loftr_parity.py -m gpu -q > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_rpr_parity.py --deselect tests/test_gpu_loftr_parity.py > $O/pytest_rest.log 2>&1; echo "pytest rc=$?" >> $O/pytest_rest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
timeout 400 python bench.py --config rpr_train --steps 20 --warmup 3 > $O/bench_rpr.json 2> $O/bench_rpr.err
timeout 120 python tools/ab_coarse_match.py 16 > $O/ab_coarse_match.json 2> $O/ab_coarse_match.err
timeout 300 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_loftr.json 2> $O/bench_loftr.err
timeout 300 python bench.py --no-cpu-baseline > $O/bench_sg.json 2> $O/bench_sg.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_rpr -o run -- python /root/repo/bench.py --config rpr_train --steps 6 --warmup 2 --no-cpu-baseline > /root/repo/$O/prof_rpr.json 2> /root/repo/$O/prof_rpr.err
DB=$(find /root/repo/$O/prof_rpr -name '*.db' | head -1)
python /root/repo/tools/rocprof_summary.py "$DB" /root/repo/$O/kernel_stats_rpr_train.csv
rm -rf /root/repo/$O/prof_rpr
cd /root/repo
# experiment: does the whole step capture + replay as one HIP graph when the GEMMs go through rocBLAS instead of hipBLASLt?
for Bp in 1 8 32; do
  TORCH_BLAS_PREFER_HIPBLASLT=0 timeout 150 python tools/diag_graph.py $Bp > $O/graph_rocblas_b$Bp.log 2>&1; echo "rc=$?" >> $O/graph_rocblas_b$Bp.log
done
TORCH_BLAS_PREFER_HIPBLASLT=0 timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_sg_rocblas.json 2> $O/bench_sg_rocblas.err
tail -3 $O/pytest_new.log $O/pytest_rest.log; tail -1 $O/smoke.log; cat $O/ab_coarse_match.json; tail -2 $O/graph_rocblas_b*.log
python - <<PY
import json
for f in ("bench_rpr","bench_loftr","bench_sg","bench_sg_rocblas"):
    try:
        d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d.get("cpu_baseline"))
    except Exception as e: print(f, "ERR", e, open("$O/"+f+".err").read()[-600:])
PY
