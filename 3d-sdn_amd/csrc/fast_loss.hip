// Mean absolute error between two fp32 tensors of the same memory layout: torch.nn.L1Loss() as the textural model uses
// it -- `criterionFeat` (/root/reference/textural/models/pix2pixHD_model.py:86), the discriminator feature-matching
// loss over 4 feature maps x 3 scales (:213-221) and the image reconstruction term.  torch evaluates it as sub, abs,
// mean (and sgn, mul, div on the way back): six passes over feature maps of up to 123 MB each.  Here the forward is one
// read of both operands (per-thread fp32 partial sums, one fp64 atomic per workgroup) and the backward one read of both
// and one write:  d/da mean|a - b| = sgn(a - b) / n  (sgn(0) = 0, as torch).  HBM-bound.
#include <hip/hip_runtime.h>

#include "sdn_common.h"

namespace sdn {

typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void k_l1_sum(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                double* __restrict__ sum)
{
    __shared__ float red[4];
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * 256;
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 x = reinterpret_cast<const f32x4*>(a)[i], y = reinterpret_cast<const f32x4*>(b)[i];
        s += (fabsf(x[0] - y[0]) + fabsf(x[1] - y[1])) + (fabsf(x[2] - y[2]) + fabsf(x[3] - y[3]));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) s += fabsf(a[4 * n4 + threadIdx.x] - b[4 * n4 + threadIdx.x]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(sum, (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]);
}

__global__ void k_l1_mean(const double* __restrict__ sum, long n, float* __restrict__ out) { out[0] = (float)(sum[0] / (double)n); }

// ga = sgn(a - b) * gout / n,  gb = -ga  (either may be null)
__global__ __launch_bounds__(256) void k_l1_grad(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                 const float* __restrict__ gout, float* __restrict__ ga,
                                                 float* __restrict__ gb)
{
    const float scale = gout[0] / (float)n;
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 x = reinterpret_cast<const f32x4*>(a)[i], y = reinterpret_cast<const f32x4*>(b)[i];
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float d = x[e] - y[e];
            g[e] = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
        }
        if (ga) reinterpret_cast<f32x4*>(ga)[i] = g;
        if (gb) reinterpret_cast<f32x4*>(gb)[i] = -g;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = 4 * n4 + threadIdx.x;
        const float d = a[i] - b[i];
        const float g = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
        if (ga) ga[i] = g;
        if (gb) gb[i] = -g;
    }
}

static unsigned blocks_for(long n)
{
    const long want = (n / 4 + 255) / 256;
    return (unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_l1_loss_fwd(const float* a, const float* b, long n, double* sum, float* out, sdnStream stream)
{
    if (!a || !b || !sum || !out || n < 1) return fail(SDN_EINVAL, "sdn_l1_loss_fwd: bad arguments");
    if (((uintptr_t)a | (uintptr_t)b) & 15) return fail(SDN_EINVAL, "sdn_l1_loss_fwd: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sum, 0, sizeof(double), st) != hipSuccess) return fail(SDN_ELAUNCH, "sdn_l1_loss_fwd: memset");
    hipLaunchKernelGGL(k_l1_sum, dim3(blocks_for(n)), dim3(256), 0, st, a, b, n, sum);
    hipLaunchKernelGGL(k_l1_mean, dim3(1), dim3(1), 0, st, sum, n, out);
    return check_launch("k_l1_sum");
}

SDN_API int sdn_l1_loss_bwd(const float* a, const float* b, long n, const float* grad_out, float* grad_a, float* grad_b,
                            sdnStream stream)
{
    if (!a || !b || !grad_out || n < 1 || (!grad_a && !grad_b)) return fail(SDN_EINVAL, "sdn_l1_loss_bwd: bad arguments");
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)grad_a | (uintptr_t)grad_b) & 15)
        return fail(SDN_EINVAL, "sdn_l1_loss_bwd: operands must be 16-byte aligned");
    hipLaunchKernelGGL(k_l1_grad, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, grad_out, grad_a, grad_b);
    return check_launch("k_l1_grad");
}
