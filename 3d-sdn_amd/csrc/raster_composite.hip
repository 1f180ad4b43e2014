// Per-frame compositing of the rendered objects: the reference pastes every object's PIL-resized
// silhouette / normal / depth render into frame-sized canvases on the host, far to near
// (/root/reference/geometric/scripts/main.py:541-602).  One launch here: a thread owns a frame pixel, walks the objects
// NEAR to far, and for the first one whose resized, rounded mask covers the pixel evaluates Pillow's two-pass bilinear
// resampling for that pixel only (ImagingResample: horizontal pass rounded to the pixel type, then vertical; 22-bit
// fixed point for the 8-bit mask / normal images, double accumulation for the float depth).  The source windows and
// weights (precompute_coeffs / normalize_coeffs_8bpc) are prepared by the host, derender3d/compositing.py.
// Compiled without FMA contraction: results are bit-identical to the PIL path (tests/test_gpu_composite.py).
#include <hip/hip_runtime.h>

#include "sdn_common.h"

namespace sdn {

constexpr int COMP_BITS = 22;  // Pillow Resample.c: PRECISION_BITS = 32 - 8 - 2

struct CompObj {  // one row of the host's object table
    int idx, size, left, top, boff, koff, ksize;
};

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// torchvision to_pil_image on a float tensor: pic.mul(255).byte()
__device__ __forceinline__ int u8_mask(float v) { return (int)(unsigned char)(int)(v * 255.f); }
__device__ __forceinline__ int u8_normal(float v) { return (int)(unsigned char)(int)((v / 2.f + 0.5f) * 255.f); }

template <int (*CONV)(float)>
__device__ int resample8(const float* plane, int R, const CompObj& o, const int* bounds, const int* kk8, int py, int px)
{
    if (o.size == R) return CONV(plane[py * R + px]);
    const int* b = bounds + 2 * o.boff;
    const int* k = kk8 + o.koff;
    const int y0 = b[2 * py], yc = b[2 * py + 1], x0 = b[2 * px], xc = b[2 * px + 1];
    int accv = 1 << (COMP_BITS - 1);
    for (int ky = 0; ky < yc; ky++) {
        const float* row = plane + (size_t)(y0 + ky) * R + x0;
        int acch = 1 << (COMP_BITS - 1);
        for (int kx = 0; kx < xc; kx++) acch += CONV(row[kx]) * k[px * o.ksize + kx];
        accv += clip8(acch >> COMP_BITS) * k[py * o.ksize + ky];
    }
    return clip8(accv >> COMP_BITS);
}

__device__ float resample_depth(const float* plane, int R, float zoom, const CompObj& o, const int* bounds,
                                const double* kkf, int py, int px)
{
    // main.py:586: min(depth * zoom / 100, 1) before the resize
    if (o.size == R) return fminf(plane[py * R + px] * zoom / 100.0f, 1.0f);
    const int* b = bounds + 2 * o.boff;
    const double* k = kkf + o.koff;
    const int y0 = b[2 * py], yc = b[2 * py + 1], x0 = b[2 * px], xc = b[2 * px + 1];
    double accv = 0.0;
    for (int ky = 0; ky < yc; ky++) {
        const float* row = plane + (size_t)(y0 + ky) * R + x0;
        double acch = 0.0;
        for (int kx = 0; kx < xc; kx++) acch += (double)fminf(row[kx] * zoom / 100.0f, 1.0f) * k[px * o.ksize + kx];
        accv += (double)(float)acch * k[py * o.ksize + ky];
    }
    return (float)accv;
}

__global__ __launch_bounds__(256) void k_composite(const float* __restrict__ masks, const float* __restrict__ normals,
                                                   const float* __restrict__ depth_maps, const float* __restrict__ zooms,
                                                   int R, const CompObj* __restrict__ objs, int m,
                                                   const int* __restrict__ bounds, const int* __restrict__ kk8,
                                                   const double* __restrict__ kkf, int H, int W, float* inst, float* nrm,
                                                   float* dep)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t RR = (size_t)R * R;
    for (int j = m - 1; j >= 0; j--) {  // the table is in painter's order (far first): the last covering object wins
        const CompObj o = objs[j];
        const int px = x - o.left, py = y - o.top;
        if ((unsigned)px >= (unsigned)o.size || (unsigned)py >= (unsigned)o.size) continue;
        const int m8 = resample8<u8_mask>(masks + o.idx * RR, R, o, bounds, kk8, py, px);
        if (m8 < 128) continue;  // torch.round(m8 / 255) == 0
        const size_t p = (size_t)y * W + x;
        inst[p] = (float)(1 + o.idx);
        for (int c = 0; c < 3; c++)
            nrm[(size_t)c * H * W + p] =
                (float)resample8<u8_normal>(normals + (o.idx * 3 + c) * RR, R, o, bounds, kk8, py, px) / 255.f;
        dep[p] = resample_depth(depth_maps + o.idx * RR, R, zooms[o.idx], o, bounds, kkf, py, px);
        return;
    }
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_composite_frame(const float* masks, const float* normals, const float* depth_maps, const float* zooms, int n,
                                int R, const int32_t* objs, int m, const int32_t* bounds, const int32_t* kk8,
                                const double* kkf, int H, int W, float* inst, float* nrm, float* dep, sdnStream stream)
{
    if (!masks || !normals || !depth_maps || !zooms || !objs || !bounds || !kk8 || !kkf || !inst || !nrm || !dep)
        return fail(SDN_EINVAL, "sdn_composite_frame: null pointer");
    if (n < 1 || R < 1 || m < 0 || H < 1 || W < 1) return fail(SDN_EINVAL, "sdn_composite_frame: bad sizes");
    if (m == 0) return SDN_OK;
    static_assert(sizeof(CompObj) == 7 * sizeof(int32_t), "object table row");
    hipLaunchKernelGGL(k_composite, dim3((unsigned)((W + 63) / 64), (unsigned)((H + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, masks, normals, depth_maps, zooms, R, reinterpret_cast<const CompObj*>(objs), m,
                       bounds, kk8, kkf, H, W, inst, nrm, dep);
    return check_launch("k_composite");
}
