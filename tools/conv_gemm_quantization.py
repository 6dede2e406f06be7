"""Offline model (no GPU) of what workgroup-count quantisation costs the decoder-convolution GEMMs of csrc/conv_gemm_bf16.hip on 256 CUs,
and what a stream-K tail would recover.  For every layer x images-per-call x tile shape: workgroups, rounds at the resident-slot count,
efficiency = useful tile work / (rounds x slots), and the same with the LAST partial round split along K over all slots (stream-K: the
tail's tiles are cut into equal K ranges so that every slot works; cost: one fp32 partial tile per extra split, added by a fix-up pass).
Measured anchor points (profiles/r03_ab_conv_bf16_b10.json / _b20.json): 92x68 layers 508 workgroups on 512 slots -> 1087 TFLOP/s, 516 ->
694; 46x34 layers 264 (one per CU) -> 625-647, 528 (16-workgroup second round) -> 743.
python tools/conv_gemm_quantization.py [--out profiles/r03_conv_gemm_quantization_study.json]"""
import argparse
import json
import math

LAYERS = [("upconv4/iconv4 1024->512 @46x34", 1024, 512, 46, 34), ("upconv3/iconv3 512->256 @92x68", 512, 256, 92, 68)]
TILES = [(256, 128, 2), (128, 128, 3), (256, 256, 1), (128, 256, 2)]      # (rows, channels, resident workgroups per CU)
CUS = 256


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="profiles/r03_conv_gemm_quantization_study.json")
    a = ap.parse_args()
    res = []
    for name, C, N, H, W in LAYERS:
        for B in (10, 20):
            Mp = B * (H + 2) * (W + 1)
            for bm, bn, occ in TILES:
                wg = math.ceil(Mp / bm) * math.ceil(N / bn)
                slots = CUS * occ
                rounds = math.ceil(wg / slots)
                useful = Mp * N / (bm * bn)                         # tiles' worth of useful work (edge tiles are partly empty)
                eff = useful / (rounds * slots)
                tail = wg - (rounds - 1) * slots                    # workgroups of the last round
                # stream-K: full rounds as they are; the tail's `tail` tiles are cut along K into `slots` pieces of equal size
                t_streamk = (rounds - 1) + tail / slots
                eff_sk = useful / (t_streamk * slots)
                res.append(dict(layer=name, images=B, rows=Mp, tile=f"{bm}x{bn}", resident_per_cu=occ, workgroups=wg, slots=slots, rounds=rounds,
                                last_round_fill=round(tail / slots, 3), efficiency=round(eff, 3), efficiency_stream_k=round(eff_sk, 3),
                                extra_partial_tiles_stream_k=max(0, slots - tail) if tail < slots else 0))
    for r in res:
        print(json.dumps(r))
    json.dump({"what": __doc__.split("\n\n")[0], "rows": res}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
