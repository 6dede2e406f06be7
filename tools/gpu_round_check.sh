set -x
mkdir -p gpurun_out/final
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/final/smoke.log 2>&1
timeout 300 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/final/prof -o run -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/final/prof_bench.json 2> /root/repo/gpurun_out/final/prof.err
cd /root/repo
DB=$(find gpurun_out/final/prof -name '*.db' | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/final/kernel_stats.csv
rm -f "$DB"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /root/repo/gpurun_out/final/pmc_$C -o run --output-format csv -- python /root/repo/bench.py --steps 2 --warmup 0 --no-cpu-baseline > /root/repo/gpurun_out/final/pmc_$C.log 2>&1
done
cd /root/repo
F=$(find gpurun_out/final/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); Wr=$(find gpurun_out/final/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py "$F" "$Wr" gpurun_out/final/pmc_hbm.csv gpurun_out/final/pmc_wino.json wino_conv3x3 7962624000 32
find gpurun_out/final -name '*counter_collection.csv' -size +8M -delete; find gpurun_out/final -name '*kernel_trace.csv' -size +8M -delete
head -8 gpurun_out/final/pmc_hbm.csv; cat gpurun_out/final/pmc_wino.json
tail -3 gpurun_out/final/pytest.log; cat gpurun_out/final/smoke.log | tail -2; cat gpurun_out/final/bench.json
