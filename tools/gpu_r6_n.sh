O=gpurun_out/r6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nets_parity.py tests/test_gpu_parity_census.py -m gpu -q -x > $O/pytest_n.log 2>&1; tail -4 $O/pytest_n.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_sg_n.json 2> $O/bench_sg_n.err; cut -c1-200 $O/bench_sg_n.json
bash tools/gpu_r6_profiles.sh sg_pnp > $O/profiles_n.log 2>&1; grep -E "sp_nms|sp_select" gpurun_out/r06_bench_sg_pnp_kernel_stats.csv | cut -c1-40,120-200
