cd /root/repo
python -m pytest tests/test_gpu_solver_parity.py tests/test_gpu_parity_census.py -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/lk -o run --output-format csv -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1; grep "pnp_\|sg_sweep\|sp_nms" /tmp/lk/run_kernel_stats.csv | awk -F'","' '{print substr($1,1,40), $2, $4}'
