set -x
O=gpurun_out/r2l; mkdir -p $O
timeout 120 python tools/diag_plugin_steps.py > $O/plugin_steps.log 2>&1
timeout 200 python tools/bench_plugin.py --pairs 24 --out $O/bench_plugin.json > $O/bench_plugin.log 2>&1
timeout 200 python -m pytest tests/test_gpu_plugin_surface.py -m gpu -q > $O/pytest_plugin.log 2>&1; echo "rc=$?" >> $O/pytest_plugin.log
grep -v "Warning\|warn" $O/plugin_steps.log | tail -n 12; grep "^[A-Z]" $O/bench_plugin.log | tail -n 6; tail -n 3 $O/pytest_plugin.log
