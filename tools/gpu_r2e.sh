set -x
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_loftr_parity.py tests/test_gpu_nets_parity.py tests/test_gpu_offline_stage.py tests/test_gpu_parity_census.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 400 python bench.py --no-cpu-baseline > $O/bench_sg.json 2> $O/bench_sg.err
timeout 400 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_loftr.json 2> $O/bench_loftr.err
cd /tmp && export TMPDIR=/tmp
for C in sg_pnp loftr_emat; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_$C -o run -- python /root/repo/bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline > /root/repo/$O/prof_$C.json 2> /root/repo/$O/prof_$C.err
  DB=$(find /root/repo/$O/prof_$C -name '*.db' | head -1)
  python /root/repo/tools/rocprof_summary.py "$DB" /root/repo/$O/kernel_stats_$C.csv
  rm -rf /root/repo/$O/prof_$C
done
cd /root/repo
tail -4 $O/pytest.log; python - <<PY
import json
for f in ("bench_sg","bench_loftr"):
    d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
