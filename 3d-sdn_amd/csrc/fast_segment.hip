// Instance-wise average pooling of the textural Encoder (textural/models/networks.py:310-325): every pixel of an
// instance is replaced by the instance's mean feature.  The reference walks np.unique(inst) on the host and masks per
// (instance, channel); torch's index_add_ does it on the device but serialises on the ~10 instance rows (0.7 ms per
// call at 384 x 1248).  Here: a workgroup reduces its pixels into an LDS table keyed by the dense segment id (LDS
// atomics, the segments are spatially coherent) and adds the non-empty rows to the global table once; a second kernel
// divides and broadcasts.  The backward pass is the same operator applied to the incoming gradient
// (d/dx of mean-broadcast = mean-broadcast of the gradient).  HBM-bound: 2 reads + 1 write of the [N, C, H, W] map.
#include <hip/hip_runtime.h>

#include "sdn_common.h"

namespace sdn {

constexpr int SEG_LDS = 4096;  // table rows kept in LDS; larger id counts go straight to global atomics

// x [N, C, HW] (NCHW), seg [N, HW] dense ids in [0, K): sums [C, K] += x, counts [K] += 1 (from channel 0's blocks)
__global__ __launch_bounds__(256) void k_segment_sum(const float* __restrict__ x, const int* __restrict__ seg,
                                                     float* __restrict__ sums, float* __restrict__ counts, int C, int HW,
                                                     int K, int chunk)
{
    __shared__ float tab[SEG_LDS];
    __shared__ float cnt[SEG_LDS];
    const int n = blockIdx.z, c = blockIdx.y;
    const int p0 = blockIdx.x * chunk, p1 = min(p0 + chunk, HW);
    const bool use_lds = K <= SEG_LDS;
    const bool count = c == 0;
    if (use_lds) {
        for (int k = threadIdx.x; k < K; k += 256) {
            tab[k] = 0.f;
            cnt[k] = 0.f;
        }
        __syncthreads();
    }
    const float* xp = x + ((size_t)n * C + c) * HW;
    const int* sp = seg + (size_t)n * HW;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        const int k = sp[p];
        const float v = xp[p];
        if (use_lds) {
            atomicAdd(&tab[k], v);
            if (count) atomicAdd(&cnt[k], 1.f);
        } else {
            unsafeAtomicAdd(sums + (size_t)c * K + k, v);
            if (count) unsafeAtomicAdd(counts + k, 1.f);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += 256) {
            if (count && cnt[k] != 0.f) unsafeAtomicAdd(counts + k, cnt[k]);
            if (tab[k] != 0.f) unsafeAtomicAdd(sums + (size_t)c * K + k, tab[k]);
        }
    }
}

// out[n, c, p] = sums[c, seg[n, p]] / counts[seg[n, p]]
__global__ __launch_bounds__(256) void k_segment_bcast(const float* __restrict__ sums, const float* __restrict__ counts,
                                                       const int* __restrict__ seg, float* __restrict__ out, int C,
                                                       int HW, int K)
{
    const int n = blockIdx.z, c = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int k = seg[(size_t)n * HW + p];
    out[((size_t)n * C + c) * HW + p] = sums[(size_t)c * K + k] / counts[k];
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_segment_mean(const float* x, const int32_t* seg, int N, int C, int HW, int K, float* sums, float* counts,
                             float* out, sdnStream stream)
{
    if (!x || !seg || !sums || !counts || !out) return fail(SDN_EINVAL, "sdn_segment_mean: null pointer");
    if (N < 1 || C < 1 || HW < 1 || K < 1) return fail(SDN_EINVAL, "sdn_segment_mean: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(sums, 0, (size_t)C * K * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(counts, 0, (size_t)K * sizeof(float), st);
    if (e != hipSuccess) return fail(SDN_ELAUNCH, "sdn_segment_mean: memset: %s", hipGetErrorString(e));
    // ~2048 blocks in total, at least 4096 pixels each (one LDS table flush per block)
    int chunks = 2048 / (N * C);
    if (chunks < 1) chunks = 1;
    int chunk = (HW + chunks - 1) / chunks;
    if (chunk < 4096) chunk = 4096;
    chunks = (HW + chunk - 1) / chunk;
    hipLaunchKernelGGL(k_segment_sum, dim3((unsigned)chunks, (unsigned)C, (unsigned)N), dim3(256), 0, st, x, seg, sums, counts,
                       C, HW, K, chunk);
    hipLaunchKernelGGL(k_segment_bcast, dim3((unsigned)((HW + 255) / 256), (unsigned)C, (unsigned)N), dim3(256), 0, st, sums,
                       counts, seg, out, C, HW, K);
    return check_launch("k_segment_mean");
}
