for w in dconv_l1 dconv_conv2a; do bash tools/pmc_kernel.sh $w conv_direct gpurun_out/r06_pmc_$w.json 2>&1 | tail -1 | cut -c1-1600; done
