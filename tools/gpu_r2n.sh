# last GPU pass of the round (3 minutes of budget): the sync-free Kabsch kernel + graph-step trainer tests, training bench A/B
set -x
O=gpurun_out/r2n; mkdir -p $O
timeout 150 python -m pytest tests/test_gpu_rpr_parity.py -m gpu -q -x > $O/pytest_rpr.log 2>&1; echo "pytest rc=$?" >> $O/pytest_rpr.log
timeout 60 python bench.py --config rpr_train --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_rpr.json 2> $O/bench_rpr.err
timeout 60 python bench.py --config rpr_train --steps 20 --warmup 3 --no-cpu-baseline --rpr-opts graph > $O/bench_rpr_graph.json 2> $O/bench_rpr_graph.err
tail -n 12 $O/pytest_rpr.log | cut -c1-200
python - <<PY
import json
for f in ("bench_rpr","bench_rpr_graph"):
    try:
        d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"].get("graph_step"), d["config"]["last_losses"], d["roofline"]["avg_launch_ms"])
    except Exception as e: print(f, "ERR", e, open("$O/"+f+".err").read()[-800:])
PY
