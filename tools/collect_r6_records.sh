# copies the records of tools/gpu_r6_final.sh (gpurun_out/, scratch) into profiles/ (tracked)
set -e
cd "$(dirname "$0")/.."
for f in r06_ab_direct_conv_halo.json r06_bench_loftr_emat_kernel_stats.json r06_bench_rpr_train_kernel_stats.json r06_bench_sg_pnp_kernel_stats.json \
         r06_bench_loftr_emat_kernel_stats.csv r06_bench_rpr_train_kernel_stats.csv r06_bench_sg_pnp_kernel_stats.csv \
         r06_dconv_timeline_conv2a.json r06_dconv_timeline_l1.json r06_dconv_timeline_l1out2.json r06_fused_split_1gpu.json r06_loftr_stage_diff_f64.json \
         r06_parity_census_easy.json r06_parity_census_hard.json r06_parity_census_hard2.json r06_pmc_conv1ab.json r06_pmc_dconv_conv2a.json r06_pmc_dconv_l1.json \
         r06_pmc_dconv_l1out2.json r06_pmc_gemm.json r06_pmc_loftr_l1out2.json; do
  [ -f gpurun_out/$f ] && cp gpurun_out/$f profiles/$f
done
cp gpurun_out/final6/bench.json profiles/r06_bench_default.json
tail -3 gpurun_out/final6/pytest.log > profiles/r06_gpu_pytest_tail.txt
cp gpurun_out/final6/smoke.log profiles/r06_smoke.txt
ls profiles | grep -c r06_
