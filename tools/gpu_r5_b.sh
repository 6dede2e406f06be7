# round 5, call B: f16x2 attention parity + timing, bench
set -x
O=gpurun_out/r5b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_nets_parity.py -q -x > $O/pytest_nets.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_nets.log
timeout 200 python tools/bench_attention.py $O/ab_attention.json > $O/ab_attention.log 2>&1; tail -20 $O/ab_attention.log
timeout 300 python bench.py --no-secondary --no-cpu-baseline > $O/bench_f16x2.json 2> $O/bench_f16x2.err; tail -c 300 $O/bench_f16x2.json
