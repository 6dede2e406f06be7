set -x
O=gpurun_out/r6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_solver_overlap.py tests/test_gpu_loftr_parity.py tests/test_gpu_range_guard.py -m gpu -q -x > $O/pytest_e.log 2>&1; tail -15 $O/pytest_e.log
timeout 600 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_loftr_e.json 2> $O/bench_loftr_e.err; cut -c1-300 $O/bench_loftr_e.json; tail -3 $O/bench_loftr_e.err
timeout 600 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --overlap 0 > $O/bench_loftr_e0.json 2> $O/bench_loftr_e0.err; cut -c1-300 $O/bench_loftr_e0.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_sg_e.json 2> $O/bench_sg_e.err; cut -c1-300 $O/bench_sg_e.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --overlap 0 > $O/bench_sg_e0.json 2> $O/bench_sg_e0.err; cut -c1-300 $O/bench_sg_e0.json
