O=gpurun_out/r6
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv_direct.py -m gpu -q -x > $O/pytest_k.log 2>&1; tail -3 $O/pytest_k.log
timeout 600 python tools/ab_direct_halo.py $O/ab_direct_halo_tail.json loftr > $O/ab_direct_halo_tail.log 2>&1; cat $O/ab_direct_halo_tail.log | cut -c1-130
timeout 600 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_loftr_k.json 2> $O/bench_loftr_k.err; cut -c1-200 $O/bench_loftr_k.json
