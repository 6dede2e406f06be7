# round-3 GPU pass behind profiles/r03_*: full GPU test suite, PMC passes on conv1b (bf16x3), rocprofv3 kernel stats of the bench step,
# the harder parity census, the host-fed configs[3] leg with and without the graph replay.  bash tools/gpu_r3_final.sh
O=gpurun_out
timeout 700 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/r03_gputest_full.log; tail -3 $O/r03_gputest_full.log
PMC_PASSES=4 timeout 300 bash tools/pmc_conv_bf16x3.sh > $O/r03_pmc.log 2>&1; tail -2 $O/r03_pmc.log | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_r3 -o run --output-format csv -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > /root/repo/$O/prof_r3_bench.json 2> /root/repo/$O/prof_r3.err )
[ -f $O/prof_r3/run_kernel_stats.csv ] && cp $O/prof_r3/run_kernel_stats.csv $O/r03_bench_sg_pnp_kernel_stats.csv && head -12 $O/r03_bench_sg_pnp_kernel_stats.csv | cut -c1-160
rm -f $O/prof_r3/run_kernel_trace.csv
timeout 400 python tools/parity_census.py --sg 32 --loftr 8 --procrustes 16 --sift 16 --hard 2 --out $O/r03_parity_census_hard2.json 2>&1 | grep -v Warn | cut -c1-600
timeout 150 python tools/bench_fused_split.py --scenes 40 --configs sg_pnp --no-resume-legs --graph 0 --out $O/r03_fused_split_40_eager.json 2>&1 | tail -1 | cut -c1-500
timeout 150 python tools/bench_fused_split.py --scenes 40 --configs sg_pnp --no-resume-legs --graph 1 --out $O/r03_fused_split_40_graph.json 2>&1 | tail -1 | cut -c1-500
