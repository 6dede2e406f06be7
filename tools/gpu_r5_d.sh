cd /root/repo
python -m pytest tests/test_gpu_gemm_split.py -x -q -m gpu -k batched 2>&1 | tail -15
python -m pytest tests/test_gpu_nets_parity.py tests/test_gpu_routes_agree.py -x -q -m gpu 2>&1 | tail -8
python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_d_sg.json; cat gpurun_out/bench_d_sg.json | cut -c1-400
python bench.py --config loftr_emat --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_d_loftr.json; cat gpurun_out/bench_d_loftr.json | cut -c1-400
