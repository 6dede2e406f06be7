# round-2 closing pass: whole GPU suite + smoke on the final code, plugin-path timing with the LoFTR coarse stage graphed,
# PMC HBM traffic of the correlation-volume backward kernels (separate passes)
set -x
O=gpurun_out/r2m; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
timeout 200 python tools/bench_plugin.py --pairs 24 --out $O/bench_plugin.json > $O/bench_plugin.log 2>&1
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 110 rocprofv3 --kernel-trace --pmc $C -d /root/repo/$O/pmc_$C -o run --output-format csv -- python /root/repo/bench.py --config rpr_train --steps 2 --warmup 0 --no-cpu-baseline > /root/repo/$O/pmc_$C.log 2>&1
done
cd /root/repo
F=$(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); Wr=$(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py "$F" "$Wr" $O/pmc_hbm_rpr_train.csv $O/pmc_cw_bwd_kv.json cw_bwd_kv 49547520 10
find $O -name '*counter_collection.csv' -size +8M -delete; find $O -name '*kernel_trace.csv' -size +8M -delete
tail -n 4 $O/pytest.log; tail -n 1 $O/smoke.log; grep "^[A-Z]" $O/bench_plugin.log | tail -n 6; cat $O/pmc_cw_bwd_kv.json; grep "cw_" $O/pmc_hbm_rpr_train.csv
