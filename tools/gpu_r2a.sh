# round-2 first GPU pass: tests, smoke, both bench configs, parity census, kernel-trace of both configs
set -x
O=gpurun_out/r2a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
timeout 400 python bench.py > $O/bench_sg.json 2> $O/bench_sg.err
timeout 400 python bench.py --config loftr_emat --steps 8 --warmup 2 > $O/bench_loftr.json 2> $O/bench_loftr.err
timeout 100 python bench.py --gpus 2 > $O/bench_g2.out 2>&1; echo "rc=$?" >> $O/bench_g2.out
timeout 600 python tools/parity_census.py --out $O/parity_census.json > $O/census.log 2>&1
cd /tmp && export TMPDIR=/tmp
for C in sg_pnp loftr_emat; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_$C -o run -- python /root/repo/bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline > /root/repo/$O/prof_$C.json 2> /root/repo/$O/prof_$C.err
  DB=$(find /root/repo/$O/prof_$C -name '*.db' | head -1)
  python /root/repo/tools/rocprof_summary.py "$DB" /root/repo/$O/kernel_stats_$C.csv
  rm -rf /root/repo/$O/prof_$C
done
cd /root/repo
tail -5 $O/pytest.log; tail -2 $O/smoke.log; cat $O/bench_sg.json $O/bench_loftr.json; tail -3 $O/bench_g2.out; cat $O/census.log | tail -4
head -30 $O/kernel_stats_loftr_emat.csv
