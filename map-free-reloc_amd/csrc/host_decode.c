/* host_decode.c -- libmfr_host.so: the two per-pixel loops of the batched loaders (datasets.py), in C.
 *
 * Reference: lib/datasets/utils.py:58-81 (read_color_image: RGB / 255 as float32; read_depth_image: uint16 mm / 1000 as float32) and the
 * matchers' gray conversion.  The loaders of the fused path want, per frame, the float32 luma plane
 *     (0.299f * (R / 255f) + 0.587f * (G / 255f)) + 0.114f * (B / 255f)          (float32 products and sums, in this order)
 * and the metric depth float32(v / 1000.0).  R, G, B are bytes and v is a uint16, so both are table look-ups of the SAME rounded values the
 * numpy expressions produce (the tables are built by numpy in datasets.py); what this file saves is the float RGB image in between
 * (transpose + divide + three multiplies over 1.2 M elements: 2.7 ms of an 8 ms frame on one core) and, for the caller, one copy: the
 * results are written straight into the batch slot.  Plain C, no dependencies; not a compute path of the GPU product (that is
 * libmfr_hip.so) -- without this library datasets.py runs the numpy expressions.
 */
#include <stddef.h>
#include <stdint.h>

/* rgb: n pixels, 3 bytes each (PIL "RGB" raw order); t0 / t1 / t2: 256 floats each; out: n floats */
void mfr_host_gray_from_rgb(const uint8_t *rgb, size_t n, const float *t0, const float *t1, const float *t2, float *out)
{
    for (size_t i = 0; i < n; ++i) {
        const float a = t0[rgb[3 * i]] + t1[rgb[3 * i + 1]];
        out[i] = a + t2[rgb[3 * i + 2]];
    }
}

/* d: n uint16 values; lut: 65536 floats; out: n floats */
void mfr_host_depth_from_u16(const uint16_t *d, size_t n, const float *lut, float *out)
{
    for (size_t i = 0; i < n; ++i) out[i] = lut[d[i]];
}

int mfr_host_abi_version(void) { return 1; }
