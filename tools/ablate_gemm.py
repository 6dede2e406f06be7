"""Where the time of the transformer linear layers goes (csrc/gemm_bf16x3.hip), on the three SuperGlue shapes at M = 65536:
one tile per workgroup (round 3) vs persistent 128x128 workgroups (with / without deferred tile stores) vs the eight-wavefront 256x128 kernel
with two LDS stages, and the ablations: output stores removed (wrong results by construction), all wavefronts in the same phase -> gpurun_out/r04_ablate_gemm.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mapfree_reloc_amd import _lib
from mapfree_reloc_amd.nets.linear import SplitLinear

lib = _lib.load(require_gpu=True)
dev = "cuda:0"
VARIANTS = (("one_tile_per_workgroup", 4), ("persistent_128x128", 8), ("persistent_128x128_deferred_stores", 16), ("eight_wavefronts_256x128_opposite_phase", 32),
            ("persistent_w_by_lds_dma_x_two_steps_ahead", 0),
            ("persistent_128x128_no_stores", 8 | 256), ("persistent_128x128_no_global_loads", 8 | 512), ("persistent_128x128_no_x_loads", 8 | 768),
            ("eight_wavefronts_same_phase", 32 | 512), ("lds_dma_no_stores", 256), ("lds_dma_no_loads", 512))
res = {}
for name, M, K, N, relu, acc in (("qkv 256->768", 65536, 256, 768, 0, 0), ("mlp1 512->512 relu", 65536, 512, 512, 1, 0), ("mlp2 512->256 +=", 65536, 512, 256, 0, 1),
                                 ("loftr 256->256", 195840, 256, 256, 0, 0)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    y = torch.randn(M, N, device=dev)
    lin = SplitLinear(w, b)
    rec = {}
    for rep in range(2):
        for tag, fl in VARIANTS:
            flags = (1 if relu else 0) | (2 if acc else 0) | fl
            def go():
                _lib.check(lib.mfr_gemm_bf16x3(x.data_ptr(), x.stride(0), _lib.ptr(lin.packed), _lib.ptr(lin.bias), y.data_ptr(), y.stride(0), M, N, K, flags, _lib.stream_ptr()), "gemm")
            for _ in range(3): go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): go()
            e1.record(); torch.cuda.synchronize()
            rec[tag] = round(min(rec.get(tag, 1e9), e0.elapsed_time(e1) / 20), 4)
    fl6 = 6 * 2.0 * M * K * N
    rec["bf16_tflops"] = {t: round(fl6 / rec[t] / 1e9, 1) for t, _ in VARIANTS[:5]}
    rec["hbm_bytes_algorithmic"] = 4 * M * (K + N * (2 if acc else 1))
    res[name] = rec
    print(name, rec, flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
