set -x
O=gpurun_out/r5c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_routes_agree.py -q -x > $O/pytest_routes.log 2>&1; echo "routes rc=$?"; tail -25 $O/pytest_routes.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "all rc=$?"; tail -30 $O/pytest_all.log
timeout 300 python bench.py --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json
timeout 300 python bench.py --config loftr_emat --no-secondary --no-cpu-baseline --steps 8 --warmup 2 > $O/bench_loftr.json 2> $O/bench_loftr.err; tail -c 200 $O/bench_loftr.json
