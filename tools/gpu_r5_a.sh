# round 5, call A: f16x2 gate (probe), parity of the split kernels, A/B timings, bench A/B
set -x
O=gpurun_out/r5a; mkdir -p $O
timeout 120 tools/ubench/f16x2_probe > $O/f16x2_probe.jsonl 2> $O/probe.err; echo "probe rc=$?"
head -12 $O/f16x2_probe.jsonl
timeout 900 python -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_winograd_split.py -q -x > $O/pytest_split.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_split.log
timeout 200 python tools/bench_gemm.py $O/ab_gemm.json > $O/ab_gemm.log 2>&1; tail -4 $O/ab_gemm.log
timeout 300 python tools/bench_conv.py $O/ab_conv.json > $O/ab_conv.log 2>&1; tail -11 $O/ab_conv.log
timeout 300 python bench.py --no-secondary --no-cpu-baseline > $O/bench_f16x2.json 2> $O/bench_f16x2.err; tail -c 600 $O/bench_f16x2.json
timeout 300 python bench.py --no-secondary --no-cpu-baseline --hip-opt SPLIT=bf16x3 > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err; tail -c 300 $O/bench_bf16x3.json
timeout 300 python bench.py --config loftr_emat --no-secondary --no-cpu-baseline --steps 8 --warmup 2 > $O/bench_loftr_f16x2.json 2> $O/bench_loftr.err; tail -c 300 $O/bench_loftr_f16x2.json
