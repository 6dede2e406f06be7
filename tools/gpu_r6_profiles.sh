# round 6: rocprofv3 kernel statistics of the TIMED steps of the bench configurations (warm-up and library search excluded:
# tools/kernel_stats_timed.py) -> gpurun_out/r05_*   usage: bash tools/gpu_r5_profiles.sh [configs...]
O=gpurun_out
run_cfg() {   # config anchor init_steps warmup steps
  c=$1
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d /root/repo/$O/prof6_$c -o run --output-format csv -- python /root/repo/bench.py --config $c --steps $5 --warmup $4 --no-cpu-baseline --no-secondary > /root/repo/$O/prof6_$c.json 2> /root/repo/$O/prof6_$c.err )
  python tools/kernel_stats_timed.py $O/prof6_$c/run_kernel_trace.csv $O/r06_bench_${c}_kernel_stats.csv --anchor $2 --warmup $(( $3 + $4 )) --steps $5 --json $O/r06_bench_${c}_kernel_stats.json | cut -c1-400
  tail -1 $O/prof6_$c.json | cut -c1-200
  rm -f $O/prof6_$c/run_kernel_trace.csv
  head -40 $O/r06_bench_${c}_kernel_stats.csv | cut -c1-200
}
for c in ${@:-sg_pnp loftr_emat}; do
  case $c in
    sg_pnp) run_cfg sg_pnp pnp_select 6 3 10;;
    loftr_emat) run_cfg loftr_emat emat_select 3 3 6;;
    rpr_train) run_cfg rpr_train cw_fwd 3 3 10;;
  esac
done
