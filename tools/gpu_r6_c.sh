set -x
O=gpurun_out/r6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_conv_igemm.py tests/test_gpu_loftr_parity.py -m gpu -q -x > $O/pytest_c.log 2>&1; tail -5 $O/pytest_c.log
timeout 600 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_loftr_c.json 2> $O/bench_loftr_c.err; cut -c1-400 $O/bench_loftr_c.json
bash tools/gpu_r6_profiles.sh loftr_emat
