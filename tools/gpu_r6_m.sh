O=gpurun_out/r6
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_m.log 2>&1; tail -4 $O/pytest_m.log
