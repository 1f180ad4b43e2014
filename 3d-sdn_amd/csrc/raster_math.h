// Device arithmetic of the rasterizer.  Every function reproduces, operation for operation, what the
// reference's kernels compute (file:line under /root/reference/geometric/neural_renderer/), including
// the sub-expressions CUDA evaluates in double because of un-suffixed literals.  The translation unit
// is compiled with -ffp-contract=off and correctly rounded fp32 divide/sqrt, so `a * b + c` below is
// two roundings, like in the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdn {

// CUDA float->int conversion: NaN -> 0, saturating (rasterize.py uses implicit conversions).
__device__ __forceinline__ int cvt_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (int)0x80000000;
    return (int)v;
}

// rasterize.py:120,252,307,537
__device__ __forceinline__ bool is_backface(const float f[9])
{
    const float a = (f[7] - f[1]) * (f[3] - f[0]);
    const float b = (f[4] - f[1]) * (f[6] - f[0]);
    return a < b;
}

// rasterize.py:138,258,546:  0.5 * (v * is + is - 1)   (0.5 * x is exact, so float == double path)
__device__ __forceinline__ float ndc_to_pixel(float v, float is_f)
{
    float t = v * is_f;
    t = t + is_f;
    t = t - 1.0f;
    return 0.5f * t;
}

// rasterize.py:292-293: (2. * i + 1 - is) / is in double, rounded to float
__device__ __forceinline__ float pixel_to_ndc(int i, int is)
{
    return (float)((2. * (double)i + 1. - (double)is) / (double)is);
}

// rasterize.py:147-155,261-269.  p[k] = (x,y) of vertex k in pixel coordinates; returns denominator.
__device__ __forceinline__ float face_inverse(const float px[3], const float py[3], float inv[9])
{
    inv[0] = py[1] - py[2];
    inv[1] = px[2] - px[1];
    inv[2] = px[1] * py[2] - px[2] * py[1];
    inv[3] = py[2] - py[0];
    inv[4] = px[0] - px[2];
    inv[5] = px[2] * py[0] - px[0] * py[2];
    inv[6] = py[0] - py[1];
    inv[7] = px[1] - px[0];
    inv[8] = px[0] * py[1] - px[1] * py[0];
    float den = px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2]);
    den = den + px[1] * (py[2] - py[0]);
#pragma unroll
    for (int k = 0; k < 9; k++) inv[k] = inv[k] / den;
    return den;
}

// rasterize.py:311-313 -- true when pixel (xp,yp) in NDC passes all three edge tests
__device__ __forceinline__ bool inside_ndc(const float f[9], float xp, float yp)
{
    if ((yp - f[1]) * (f[3] - f[0]) < (xp - f[0]) * (f[4] - f[1])) return false;
    if ((yp - f[4]) * (f[6] - f[3]) < (xp - f[3]) * (f[7] - f[4])) return false;
    if ((yp - f[7]) * (f[0] - f[6]) < (xp - f[6]) * (f[1] - f[7])) return false;
    return true;
}

// rasterize.py:316-328 -- clamp through fmax/fmin (NaN loses), renormalise
__device__ __forceinline__ void bary_weights(const float inv[9], int xi, int yi, float w[3])
{
    const float fx = (float)xi, fy = (float)yi;
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float t = inv[3 * k + 0] * fx + inv[3 * k + 1] * fy;
        t = t + inv[3 * k + 2];
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        w[k] = t;
        sum = sum + t;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] = w[k] / sum;
}

// rasterize.py:331 -- 1. / (w0/z0 + w1/z1 + w2/z2).  A double reciprocal rounded to float equals the
// correctly rounded float reciprocal (53 >= 2*24+2), so float division is exact parity.
__device__ __forceinline__ float persp_depth(const float w[3], float z0, float z1, float z2)
{
    float s = w[0] / z0 + w[1] / z1;
    s = s + w[2] / z2;
    return 1.0f / s;
}

// order-preserving map float -> uint32 (all finite floats, negative included)
__device__ __forceinline__ uint32_t ord_bits(float v)
{
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_unbits(uint32_t o)
{
    const uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(b);
}

// rasterize.py:398-423 -- trilinear sampling of a per-face ts^3 texture.  `texel(isc, k)` fetches channel
// k of texel isc; returns the 8 (index, weight) pairs through out arrays when requested.
struct TexCoord {
    int base[3];
    float frac[3];
};

__device__ __forceinline__ TexCoord texture_coord(const float w[3], float depth, float z0, float z1, float z2,
                                                  int ts, double eps)
{
    TexCoord tc;
    const float z[3] = {z0, z1, z2};
    const double c = (double)(ts - 1) - eps;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float q = depth / z[k];
        const float tif = (float)(((double)w[k] * c) * (double)q);
        const int b = cvt_i32(tif);
        tc.base[k] = b;
        tc.frac[k] = tif - (float)b;
    }
    return tc;
}

__device__ __forceinline__ void texture_corner(const TexCoord& tc, int pn, int ts, int& isc, float& wgt)
{
    float w = 1.0f;
    int idx[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (((pn >> k) & 1) == 0) {
            w = w * (1.0f - tc.frac[k]);
            idx[k] = tc.base[k];
        } else {
            w = w * tc.frac[k];
            idx[k] = tc.base[k] + 1;
        }
    }
    isc = idx[0] * ts * ts + idx[1] * ts + idx[2];
    wgt = w;
}

}  // namespace sdn
