set -x
O=gpurun_out/r6
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv_direct.py -m gpu -q -x > $O/pytest_g.log 2>&1; tail -5 $O/pytest_g.log
for l in l1nores l1 conv2a conv1b l2out2; do timeout 120 python tools/dconv_timeline.py --layer $l $O/dconv_timeline_$l.json > $O/dconv_timeline_$l.log 2>&1; cat $O/dconv_timeline_$l.log; done
timeout 600 python tools/ab_direct_halo.py $O/ab_direct_halo.json > $O/ab_direct_halo.log 2>&1; cat $O/ab_direct_halo.log | cut -c1-250
