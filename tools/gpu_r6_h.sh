O=gpurun_out/r6
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv_direct.py -m gpu -q -x > $O/pytest_g.log 2>&1; tail -3 $O/pytest_g.log
for l in conv2a l1; do timeout 120 python tools/dconv_timeline.py --layer $l $O/dconv_timeline_$l.json 2>&1 | grep -E "layer|  0 start|  1 stage|  2 step|step 1: at|loop exit|scale|first channel|end \(" ; done
timeout 600 python tools/ab_direct_halo.py $O/ab_direct_halo.json > $O/ab_direct_halo.log 2>&1; cat $O/ab_direct_halo.log | cut -c1-130
