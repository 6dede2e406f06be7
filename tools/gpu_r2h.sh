set -x
O=gpurun_out/r2h; mkdir -p $O
for t in 1 2; do
  timeout 120 python -u tools/diag_graph_phase.py 8 > $O/phase_b8_lt_$t.log 2>&1; echo "rc=$?" >> $O/phase_b8_lt_$t.log
  DISABLE_ADDMM_CUDA_LT=1 TORCH_BLAS_PREFER_HIPBLASLT=0 timeout 120 python -u tools/diag_graph_phase.py 8 > $O/phase_b8_nolt_$t.log 2>&1; echo "rc=$?" >> $O/phase_b8_nolt_$t.log
done
for f in $O/phase_*.log; do echo "== $f"; grep -v amdgpu.ids $f | tail -n 8; done
