# round 6: GPU suite + smoke + default bench (what the driver runs at round end) -> gpurun_out/r6/
set -x
O=gpurun_out/r6
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS:-} > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err
tail -5 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-1500 $O/bench.json
