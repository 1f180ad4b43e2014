// Mean absolute error between two fp32 tensors of the same memory layout: torch.nn.L1Loss() as the textural model uses
// it -- `criterionFeat` (/root/reference/textural/models/pix2pixHD_model.py:86), the discriminator feature-matching
// loss over 4 feature maps x 3 scales (:213-221) and the image reconstruction term.  torch evaluates it as sub, abs,
// mean (and sgn, mul, div on the way back): six passes over feature maps of up to 123 MB each.  Here the forward is one
// read of both operands (per-thread fp32 partial sums, one fp64 atomic per workgroup) and the backward one read of both
// and one write:  d/da mean|a - b| = sgn(a - b) / n  (sgn(0) = 0, as torch).  HBM-bound.
// Also here (r04): the silhouette + FFD-penalty loss of the geometric branch's optimisation loop (sdn_silhouette_loss_*).
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

// ---- the loss of the test-time optimisation loop (/root/reference/geometric/scripts/main.py:445-451):
//     loss = mean( mse_loss(masks, target, reduce=False) [* (1 - ignore)] + 100 * mean(ffd ** 2) )
// torch runs it as mse, pow, mean, mul, add, (mul,) mean forward and as many kernels again backward -- a dozen 3-9 us launches
// around a 0.9 ms frame step.  Here: two partial-sum launches (fp64 atomics into sums[0..1] = sum of the weighted squared
// error, sum of ffd^2) + one finishing thread, and ONE backward launch:
//     d loss / d masks = 2 (masks - target) (1 - ignore) g / N,   d loss / d ffd = 200 ffd g mean(1 - ignore) / n_ffd
// (the scalar 100 mean(ffd^2) is added to every element before the mean, so the ignore weights reach it through their mean).
// sums[2] receives sum(1 - ignore) (= N without an ignore map).
// (r04: at most 512 blocks -- with one block per 1024 elements the 3 x 2304 fp64 atomics on the same three addresses took
// 50 us of a 1 ms frame step; 16-byte loads when the operands allow)
__global__ __launch_bounds__(256) void k_sil_loss_sum(const float* __restrict__ m, const float* __restrict__ t,
                                                      const float* __restrict__ ign, long n, const float* __restrict__ ffd,
                                                      long nffd, int vec4, double* __restrict__ partial)
{
    __shared__ float red[3][4];
    const long stride = (long)gridDim.x * 256;
    const long first = (long)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f, q = 0.f, w = 0.f;
    if (vec4) {
        const float4* m4 = reinterpret_cast<const float4*>(m);
        const float4* t4 = reinterpret_cast<const float4*>(t);
        const float4* i4 = reinterpret_cast<const float4*>(ign);
        for (long i = first; i < (n >> 2); i += stride) {
            const float4 a = m4[i], b = t4[i];
            float4 k = make_float4(1.f, 1.f, 1.f, 1.f);
            if (ign) {
                const float4 g = i4[i];
                k = make_float4(1.f - g.x, 1.f - g.y, 1.f - g.z, 1.f - g.w);
            }
            // the element order of the scalar loop (sums of products, no contraction across elements)
            s += (a.x - b.x) * (a.x - b.x) * k.x;
            s += (a.y - b.y) * (a.y - b.y) * k.y;
            s += (a.z - b.z) * (a.z - b.z) * k.z;
            s += (a.w - b.w) * (a.w - b.w) * k.w;
            w += k.x + k.y + k.z + k.w;
        }
    } else {
        for (long i = first; i < n; i += stride) {
            const float d = m[i] - t[i];
            const float k = ign ? 1.f - ign[i] : 1.f;
            s += d * d * k;
            w += k;
        }
    }
    for (long i = first; i < nffd; i += stride) q += ffd[i] * ffd[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
        w += __shfl_xor(w, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s;
        red[1][threadIdx.x >> 6] = q;
        red[2][threadIdx.x >> 6] = w;
    }
    __syncthreads();
    // one partial per block, summed in block order by k_sil_loss_finish: no atomics, nothing to zero, the same bits every run
    if (threadIdx.x < 3)
        partial[3 * blockIdx.x + threadIdx.x] = (double)red[threadIdx.x][0] + (double)red[threadIdx.x][1] +
                                                (double)red[threadIdx.x][2] + (double)red[threadIdx.x][3];
}

__global__ __launch_bounds__(64) void k_sil_loss_finish(double* __restrict__ sums, int nblocks, long n, long nffd,
                                                         float* __restrict__ out)
{
    // sums[0..2] <- the block partials at sums[3 + 3 * block + k] in block order (lane l takes blocks l, l + 64, ...; then a
    // fixed butterfly);  mean over N of (e_i k_i + c k_i) with c = 100 mean(ffd^2):  (sum e k + c sum k) / N
    double a[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += 64)
#pragma unroll
        for (int k = 0; k < 3; k++) a[k] += sums[3 + 3 * b + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 3; k++) a[k] += __shfl_xor(a[k], o, 64);
    if (threadIdx.x == 0) {
        sums[0] = a[0];
        sums[1] = a[1];
        sums[2] = a[2];
        const double c = nffd > 0 ? 100.0 * a[1] / (double)nffd : 0.0;
        out[0] = (float)((a[0] + c * a[2]) / (double)n);
    }
}

__global__ __launch_bounds__(256) void k_sil_loss_grad(const float* __restrict__ m, const float* __restrict__ t,
                                                       const float* __restrict__ ign, long n, const float* __restrict__ ffd,
                                                       long nffd, const double* __restrict__ sums, const float* __restrict__ gout,
                                                       float* __restrict__ gm, float* __restrict__ gffd)
{
    const float g = gout[0];
    const float sm = 2.f * g / (float)n;
    const long stride = (long)gridDim.x * 256;
    if (gm)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
            gm[i] = sm * (m[i] - t[i]) * (ign ? 1.f - ign[i] : 1.f);
    if (gffd) {
        const float sf = (float)(200.0 * (double)g * (sums[2] / (double)n) / (double)nffd);
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nffd; i += stride) gffd[i] = sf * ffd[i];
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

SDN_API int sdn_silhouette_loss_fwd(const float* masks, const float* target, const float* ignore, long n, const float* ffd,
                                    long nffd, double* sums, float* out, sdnStream stream)
{
    if (!masks || !target || !sums || !out || n < 1 || nffd < 0 || (nffd && !ffd))
        return fail(SDN_EINVAL, "sdn_silhouette_loss_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int vec4 = (n & 3) == 0 && ((((uintptr_t)masks | (uintptr_t)target | (uintptr_t)ignore) & 15) == 0);
    long want = ((vec4 ? n / 4 : n) + 1023) / 1024;   // four loads per thread before another block pays
    if (want < 1) want = 1;
    const int nblocks = (int)(want > SDN_SIL_LOSS_BLOCKS ? SDN_SIL_LOSS_BLOCKS : want);
    hipLaunchKernelGGL(k_sil_loss_sum, dim3((unsigned)nblocks), dim3(256), 0, st, masks, target, ignore, n, ffd, nffd, vec4,
                       sums + 3);
    hipLaunchKernelGGL(k_sil_loss_finish, dim3(1), dim3(64), 0, st, sums, nblocks, n, nffd, out);
    return check_launch("k_sil_loss_sum");
}

SDN_API int sdn_silhouette_loss_bwd(const float* masks, const float* target, const float* ignore, long n, const float* ffd,
                                    long nffd, const double* sums, const float* grad_out, float* grad_masks, float* grad_ffd,
                                    sdnStream stream)
{
    if (!masks || !target || !sums || !grad_out || n < 1 || (!grad_masks && !grad_ffd) || (grad_ffd && (!ffd || nffd < 1)))
        return fail(SDN_EINVAL, "sdn_silhouette_loss_bwd: bad arguments");
    hipLaunchKernelGGL(k_sil_loss_grad, dim3(blocks_for(4 * n)), dim3(256), 0, (hipStream_t)stream, masks, target, ignore, n, ffd,
                       nffd, sums, grad_out, grad_masks, grad_ffd);
    return check_launch("k_sil_loss_grad");
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
