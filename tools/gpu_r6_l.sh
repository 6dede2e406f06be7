O=gpurun_out/r6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_direct.py tests/test_gpu_conv_igemm.py tests/test_gpu_loftr_parity.py tests/test_gpu_range_guard.py -m gpu -q -x > $O/pytest_l.log 2>&1; tail -3 $O/pytest_l.log
timeout 600 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_loftr_l.json 2> $O/bench_loftr_l.err; cut -c1-200 $O/bench_loftr_l.json; tail -3 $O/bench_loftr_l.err
bash tools/gpu_r6_profiles.sh loftr_emat > $O/profiles_l.log 2>&1; grep "conv_igemm\|conv_direct" gpurun_out/r06_bench_loftr_emat_kernel_stats.csv | cut -c1-60,200-290
