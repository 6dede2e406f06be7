# round-2 final pass: whole GPU suite, smoke, the three bench lines (with CPU baselines + parity legs), plugin-path profile
set -x
O=gpurun_out/r2k; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
timeout 300 python bench.py > $O/bench_sg_pnp.json 2> $O/bench_sg_pnp.err
timeout 300 python bench.py --config loftr_emat --steps 8 --warmup 2 > $O/bench_loftr_emat.json 2> $O/bench_loftr_emat.err
timeout 300 python bench.py --config rpr_train > $O/bench_rpr_train.json 2> $O/bench_rpr_train.err
timeout 120 python tools/diag_plugin_prof2.py > $O/plugin_prof.log 2>&1
timeout 200 python tools/bench_plugin.py --pairs 24 --out $O/bench_plugin.json > $O/bench_plugin.log 2>&1
tail -n 6 $O/pytest.log; tail -n 1 $O/smoke.log; head -n 40 $O/plugin_prof.log | cut -c1-160
python - <<PY
import json
for f in ("bench_sg_pnp","bench_loftr_emat","bench_rpr_train"):
    try:
        d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["config"].get("parity"), d.get("cpu_baseline"))
    except Exception as e: print(f, "ERR", e, open("$O/"+f+".err").read()[-600:])
PY
