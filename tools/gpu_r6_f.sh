set -x
O=gpurun_out/r6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_loftr_parity.py tests/test_gpu_parity_census.py -m gpu -q -x > $O/pytest_f.log 2>&1; tail -5 $O/pytest_f.log
timeout 300 python tools/bench_fine_mlp.py $O/fine_mlp.json > $O/fine_mlp.log 2>&1; cat $O/fine_mlp.log
timeout 600 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_loftr_f.json 2> $O/bench_loftr_f.err; cut -c1-300 $O/bench_loftr_f.json; tail -3 $O/bench_loftr_f.err
bash tools/gpu_r6_profiles.sh sg_pnp loftr_emat > $O/profiles_f.log 2>&1; tail -3 $O/profiles_f.log | cut -c1-300
