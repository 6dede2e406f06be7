set -x
O=gpurun_out/r2j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_plugin_surface.py tests/test_gpu_fused_submission.py tests/test_gpu_offline_stage.py tests/test_gpu_nets_parity.py -m gpu -q > $O/pytest_graph.log 2>&1; echo "pytest rc=$?" >> $O/pytest_graph.log
for g in 1 0; do
  timeout 200 python bench.py --no-cpu-baseline --graph $g > $O/bench_sg_graph$g.json 2> $O/bench_sg_graph$g.err
  timeout 200 python bench.py --no-cpu-baseline --graph $g --batch 8 --steps 40 > $O/bench_sg_b8_graph$g.json 2> $O/bench_sg_b8_graph$g.err
done
timeout 300 python tools/bench_plugin.py --pairs 24 --out $O/bench_plugin.json > $O/bench_plugin.log 2>&1
tail -n 5 $O/pytest_graph.log; tail -n 6 $O/bench_plugin.log
python - <<PY
import json
for f in ("bench_sg_graph1","bench_sg_graph0","bench_sg_b8_graph1","bench_sg_b8_graph0"):
    try:
        d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["config"].get("launch"))
    except Exception as e: print(f, "ERR", e, open("$O/"+f+".err").read()[-600:])
PY
