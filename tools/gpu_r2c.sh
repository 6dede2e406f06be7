set -x
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_loftr_parity.py tests/test_gpu_offline_stage.py tests/test_gpu_parity_census.py::test_census_loftr_emat_8_pairs tests/test_gpu_fused_submission.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 400 python bench.py --config loftr_emat --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_loftr.json 2> $O/bench_loftr.err
cd /tmp && export TMPDIR=/tmp
C=loftr_emat
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_$C -o run -- python /root/repo/bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline > /root/repo/$O/prof_$C.json 2> /root/repo/$O/prof_$C.err
DB=$(find /root/repo/$O/prof_$C -name '*.db' | head -1)
python /root/repo/tools/rocprof_summary.py "$DB" /root/repo/$O/kernel_stats_$C.csv
rm -rf /root/repo/$O/prof_$C
cd /root/repo
tail -15 $O/pytest.log; cat $O/bench_loftr.json; tail -3 $O/bench_loftr.err
