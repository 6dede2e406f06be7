cd /root/repo
python -m pytest tests/test_gpu_emat_parity.py tests/test_gpu_parity_census.py tests/test_gpu_solver_parity.py -x -q -m gpu 2>&1 | tail -4
python bench.py --config loftr_emat --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/lk -o run --output-format csv -- python /root/repo/bench.py --config loftr_emat --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; grep "emat_\|scale_" /tmp/lk/run_kernel_stats.csv | cut -c1-60,150-260
