set -x
O=gpurun_out/r6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_range_guard.py -m gpu -q -x > $O/pytest_guard.log 2>&1; tail -3 $O/pytest_guard.log
timeout 300 python tools/ab_direct_conv.py $O/ab_direct_conv.json > $O/ab_direct_conv.log 2>&1; tail -4 $O/ab_direct_conv.log
for h in 0 1 2; do
  timeout 900 python tools/parity_census.py --sg 32 --loftr 16 --procrustes 8 --sift 8 --hard $h --out $O/census_hard$h.json > $O/census_hard$h.log 2>&1; tail -4 $O/census_hard$h.log | cut -c1-900
done
timeout 600 python tools/bench_fused_split.py --root /tmp/mapfree_syn --configs sg_pnp --graph 1 --workers 15 --no-resume-legs --out $O/fused_split_graph_w15.json > $O/fused_graph.log 2>&1; tail -3 $O/fused_graph.log | cut -c1-1200
timeout 600 python tools/bench_fused_split.py --root /tmp/mapfree_syn --configs sg_pnp --graph 0 --workers 15 --no-resume-legs --out $O/fused_split_eager_w15.json > $O/fused_eager.log 2>&1; tail -3 $O/fused_eager.log | cut -c1-1200
