O=gpurun_out/r2o; mkdir -p $O
timeout 70 python tools/diag_graph_step.py fp32 > $O/diag_graph_step_fp32.log 2>&1
grep -v "Warning\|warn\|amdgpu" $O/diag_graph_step_fp32.log | tail -n 8
