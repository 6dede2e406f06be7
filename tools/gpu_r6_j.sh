O=gpurun_out/r6
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_j.log 2>&1; tail -12 $O/pytest_j.log
timeout 600 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_loftr_j.json 2> $O/bench_loftr_j.err; cut -c1-200 $O/bench_loftr_j.json; tail -3 $O/bench_loftr_j.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_sg_j.json 2> $O/bench_sg_j.err; cut -c1-200 $O/bench_sg_j.json; tail -3 $O/bench_sg_j.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --hip-opt CONV_KERNEL=split > $O/bench_sg_j_wino.json 2> $O/bench_sg_j_wino.err; cut -c1-200 $O/bench_sg_j_wino.json
timeout 600 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --hip-opt CONV_KERNEL=split > $O/bench_loftr_j_wino.json 2> $O/bench_loftr_j_wino.err; cut -c1-200 $O/bench_loftr_j_wino.json
