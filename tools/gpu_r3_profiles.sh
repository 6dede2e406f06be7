# round-3: full GPU suite + smoke + rocprofv3 kernel stats of the two secondary configurations -> gpurun_out/r03_*
O=gpurun_out
timeout 700 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/r03_gputest_full.log; tail -2 $O/r03_gputest_full.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -1
for c in loftr_emat rpr_train; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_$c -o run --output-format csv -- python /root/repo/bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > /root/repo/$O/prof_$c.json 2> /root/repo/$O/prof_$c.err )
  [ -f $O/prof_$c/run_kernel_stats.csv ] && cp $O/prof_$c/run_kernel_stats.csv $O/r03_bench_${c}_kernel_stats.csv && head -8 $O/r03_bench_${c}_kernel_stats.csv | cut -c1-150
  rm -f $O/prof_$c/run_kernel_trace.csv
done
