set -x
O=gpurun_out/r2i; mkdir -p $O
for t in 1 2; do
  timeout 120 python -u tools/diag_graph_phase.py 8 > $O/phase_b8_$t.log 2>&1; echo "rc=$?" >> $O/phase_b8_$t.log
done
timeout 150 python -u tools/diag_graph_phase.py 32 > $O/phase_b32.log 2>&1; echo "rc=$?" >> $O/phase_b32.log
for f in $O/phase_*.log; do echo "== $f"; grep -v amdgpu.ids $f | tail -n 8; done
