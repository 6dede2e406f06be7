# round-2 profile pass: kernel traces of both bench configs, HBM-traffic PMC passes, SQ counters of the conv1b launch
set -x
O=gpurun_out/r2p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in sg_pnp loftr_emat; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_$C -o run -- python /root/repo/bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline > /root/repo/$O/prof_$C.json 2> /root/repo/$O/prof_$C.err
  DB=$(find /root/repo/$O/prof_$C -name '*.db' | head -1)
  python /root/repo/tools/rocprof_summary.py "$DB" /root/repo/$O/kernel_stats_$C.csv
  rm -rf /root/repo/$O/prof_$C
  for CN in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $CN -d /root/repo/$O/pmc_${C}_$CN -o run --output-format csv -- python /root/repo/bench.py --config $C --steps 2 --warmup 0 --no-cpu-baseline > /root/repo/$O/pmc_${C}_$CN.log 2>&1
  done
done
cd /root/repo
F=$(find $O/pmc_sg_pnp_FETCH_SIZE -name '*counter_collection.csv' | head -1); Wr=$(find $O/pmc_sg_pnp_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py "$F" "$Wr" $O/pmc_hbm_sg_pnp.csv $O/pmc_conv1b.json wino_conv3x3 7962624000 32
F=$(find $O/pmc_loftr_emat_FETCH_SIZE -name '*counter_collection.csv' | head -1); Wr=$(find $O/pmc_loftr_emat_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py "$F" "$Wr" $O/pmc_hbm_loftr_emat.csv $O/pmc_loftr_l1out2.json wino_conv3x3_pipe 4913356800 16
find $O -name '*counter_collection.csv' -size +4M -delete; find $O -name '*kernel_trace.csv' -size +4M -delete; find $O -name '*.db' -delete
bash tools/pmc_conv1b.sh > $O/sq.log 2>&1; cp gpurun_out/sq/pmc_wino_sq.json $O/pmc_conv1b_sq.json
head -12 $O/pmc_hbm_sg_pnp.csv | cut -c1-200; cat $O/pmc_conv1b.json $O/pmc_loftr_l1out2.json $O/pmc_conv1b_sq.json
