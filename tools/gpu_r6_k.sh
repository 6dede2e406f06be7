O=gpurun_out/r6
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv_direct.py -m gpu -q -x > $O/pytest_k.log 2>&1; tail -3 $O/pytest_k.log
timeout 600 python tools/ab_direct_halo.py $O/ab_direct_halo.json loftr > $O/ab_direct_halo.log 2>&1; cat $O/ab_direct_halo.log | cut -c1-130
timeout 900 python tools/parity_census.py --sg 32 --loftr 0 --procrustes 0 --sift 0 --hard 2 --out $O/census_sg_hard2_direct.json > $O/census_sg_hard2_direct.log 2>&1; tail -3 $O/census_sg_hard2_direct.log | cut -c1-1500
