set -x
O=gpurun_out/r6
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest_d.log 2>&1; echo "pytest rc=$?" >> $O/pytest_d.log; tail -8 $O/pytest_d.log
timeout 600 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_loftr_d.json 2> $O/bench_loftr_d.err; cut -c1-300 $O/bench_loftr_d.json
bash tools/gpu_r6_profiles.sh loftr_emat | cut -c1-200 | head -30
