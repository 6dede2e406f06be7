cd /root/repo
python -m pytest tests/test_gpu_rpr_parity.py tests/test_gpu_conv_bf16.py -x -q -m gpu 2>&1 | tail -4
python bench.py --config rpr_train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_rpr.json; cut -c1-300 gpurun_out/bench_rpr.json
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/bench_rpr.json').read())
for k in d["roofline"].get("other_kernels",[]):
    print(k.get("kernel","")[:80], k.get("launches_timed"), k.get("avg_launch_ms"), k.get("frac"))
PY
