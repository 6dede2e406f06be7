# round 6, final records: GPU suite + smoke + default bench (the driver's commands), timed-step kernel statistics of the three bench configurations,
# PMC passes of the dominant kernels, the three-level parity census, the LoFTR stage bisect with the float64 leg, the host-fed run.  -> gpurun_out/
set -x
O=gpurun_out/final6
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
bash tools/gpu_r6_profiles.sh sg_pnp loftr_emat rpr_train > $O/profiles.log 2>&1; grep -E "^\{\"metric|kernel_ms_per_step" $O/profiles.log | cut -c1-200
for w in conv1ab dconv_l1out2 dconv_l1 dconv_conv2a gemm; do
  case $w in conv1ab) k=wino_split_c1;; dconv*) k=conv_direct;; gemm) k=gemm_split_d;; esac
  bash tools/pmc_kernel.sh $w $k gpurun_out/r06_pmc_$w.json > $O/pmc_$w.log 2>&1; tail -1 $O/pmc_$w.log | cut -c1-300
done
cp gpurun_out/r06_pmc_dconv_l1out2.json gpurun_out/r06_pmc_loftr_l1out2.json
timeout 600 python tools/ab_direct_halo.py gpurun_out/r06_ab_direct_conv_halo.json > $O/ab_direct_halo.log 2>&1; cut -c1-130 $O/ab_direct_halo.log
for l in conv2a l1 l1out2; do timeout 120 python tools/dconv_timeline.py --layer $l gpurun_out/r06_dconv_timeline_$l.json > $O/dconv_timeline_$l.log 2>&1; done
for h in 0 1 2; do
  case $h in 0) t=easy;; 1) t=hard;; 2) t=hard2;; esac
  timeout 1200 python tools/parity_census.py --sg 64 --loftr 16 --procrustes 16 --sift 32 --hard $h --out gpurun_out/r06_parity_census_$t.json > $O/census_$t.log 2>&1; tail -4 $O/census_$t.log | cut -c1-400
done
timeout 900 python tools/loftr_stage_diff.py --seeds 5000 5005 5007 5015 --hard 2 --f64 --variants f16x2 bf16x3 --out gpurun_out/r06_loftr_stage_diff_f64.json > $O/stage_diff.log 2>&1; tail -2 $O/stage_diff.log | cut -c1-300
timeout 600 python tools/bench_fused_split.py --root /tmp/mapfree_syn --configs sg_pnp,loftr_emat --workers 15 --no-resume-legs --out gpurun_out/r06_fused_split_1gpu.json > $O/fused.log 2>&1; tail -3 $O/fused.log | cut -c1-600
