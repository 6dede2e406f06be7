mkdir -p gpurun_out/loftr
timeout 300 python tools/bench_loftr.py > gpurun_out/loftr/bench.json 2> gpurun_out/loftr/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/loftr/prof -o run -- python /root/repo/tools/bench_loftr.py --steps 3 --warmup 1 > /root/repo/gpurun_out/loftr/prof_bench.json 2> /root/repo/gpurun_out/loftr/prof.err
cd /root/repo
DB=$(find gpurun_out/loftr/prof -name '*.db' | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/loftr/kernel_stats.csv; rm -f "$DB"
cat gpurun_out/loftr/bench.json; tail -3 gpurun_out/loftr/bench.err; head -25 gpurun_out/loftr/kernel_stats.csv | cut -c1-150
timeout 200 python -m pytest tests/test_gpu_winograd_conv.py -q 2>&1 | tail -3
