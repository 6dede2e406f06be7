# full GPU pass: all gpu tests, smoke, both bench configs -> gpurun_out/$1
O=gpurun_out/${1:-full}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
timeout 400 python bench.py > $O/bench_sg.json 2> $O/bench_sg.err
timeout 400 python bench.py --config loftr_emat --steps 8 --warmup 2 > $O/bench_loftr.json 2> $O/bench_loftr.err
tail -4 $O/pytest.log; tail -1 $O/smoke.log; python - <<PY
import json
for f in ("bench_sg","bench_loftr"):
    try:
        d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["config"].get("parity"), d.get("cpu_baseline"))
    except Exception as e: print(f, "ERR", e, open("$O/"+f+".err").read()[-500:])
PY
