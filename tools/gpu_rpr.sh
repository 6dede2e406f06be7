# f-4 row: GPU parity tests of the regression path -> gpurun_out/rpr
O=gpurun_out/rpr; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rpr_parity.py -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
