set -x
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 400 python bench.py > $O/bench_sg.json 2> $O/bench_sg.err
timeout 400 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_loftr.json 2> $O/bench_loftr.err
tail -5 $O/pytest.log; cat $O/bench_sg.json $O/bench_loftr.json
