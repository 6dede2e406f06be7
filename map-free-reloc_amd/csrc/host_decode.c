/* host_decode.c -- libmfr_host.so: the two per-pixel loops of the batched loaders (datasets.py), in C.
 *
 * Reference: the matchers' gray input (etc/feature_matching_baselines/matchers.py:101-104 -> SuperGlue's read_image: an 8-bit grayscale
 * read of the file, / 255 as float32) and lib/datasets/utils.py:77-81 (read_depth_image: uint16 mm / 1000 as float32).  The loaders of
 * the fused path want, per frame, the float32 gray plane  float32(luma_u8(R, G, B)) / 255f  with luma_u8 = (19595 R + 38470 G + 7471 B +
 * 2^15) >> 16 (ITU-R 601-2 rounded to a byte: datasets.luma_u8) and the metric depth float32(v / 1000.0).  Both are table look-ups of
 * the SAME rounded values the numpy expressions produce (the tables are built by numpy in datasets.py); what this file saves is the
 * intermediate arrays and, for the caller, one copy: the results are written straight into the batch slot.  Plain C, no dependencies;
 * not a compute path of the GPU product (that is libmfr_hip.so) -- without this library datasets.py runs the numpy expressions.
 */
#include <stddef.h>
#include <stdint.h>

/* rgb: n pixels, 3 bytes each (PIL "RGB" raw order); lut: 256 floats (byte / 255); out: n floats */
void mfr_host_gray_from_rgb(const uint8_t *rgb, size_t n, const float *lut, float *out)
{
    for (size_t i = 0; i < n; ++i)
        out[i] = lut[((uint32_t)rgb[3 * i] * 19595u + (uint32_t)rgb[3 * i + 1] * 38470u + (uint32_t)rgb[3 * i + 2] * 7471u + 0x8000u) >> 16];
}

/* d: n uint16 values; lut: 65536 floats; out: n floats */
void mfr_host_depth_from_u16(const uint16_t *d, size_t n, const float *lut, float *out)
{
    for (size_t i = 0; i < n; ++i) out[i] = lut[d[i]];
}

int mfr_host_abi_version(void) { return 2; }
