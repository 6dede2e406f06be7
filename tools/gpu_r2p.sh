O=gpurun_out/r2p; mkdir -p $O
timeout 60 python -m pytest tests/test_gpu_rpr_parity.py -m gpu -q -k "graph_step or kabsch" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log | cut -c1-220
