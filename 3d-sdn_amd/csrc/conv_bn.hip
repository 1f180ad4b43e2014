// BatchNorm2d (+ residual add + ReLU), 3x3 stride-2 max pooling and global average pooling for the derender3d encoder
// (ResNet-18: geometric/derender3d/models/derenderer.py:25-27 builds torchvision.models.resnet18; its convolutions run on
// conv_gemm.hip / conv_wgrad.hip).  Tensors are channels-last fp32 [N, H, W, C] with C a multiple of 4; a thread owns
// four consecutive channels (16-B accesses), a block owns CH = min(C, 64) channels x a slice of the N*H*W positions, so
// per-channel reductions end in ONE fp64 atomic per (block, channel).
//
// BatchNorm semantics follow torch.nn.BatchNorm2d (the module torchvision's ResNet uses):
//   train: mean / biased variance of the batch over (N, H, W), running statistics updated with the UNBIASED variance;
//   eval : running statistics.  Either way  y = (x - mean) * rstd * gamma + beta  =  x * scale + shift.
// The BasicBlock tail  relu(bn2(conv2(.)) + identity)  is one pass: out = relu(x * scale + shift + res).
// All HBM-bound: algorithmic bytes = one read (+ residual read) + one write per element.
#include "conv_common.h"
#include "sdn_common.h"

namespace sdn {

struct BnLay {
    int c0, prow, pstep;
    __device__ BnLay(int C)
    {
        const int CH = C < 64 ? C : 64;
        const int c4n = CH >> 2;
        c0 = blockIdx.y * CH + (threadIdx.x % c4n) * 4;
        prow = threadIdx.x / c4n;
        pstep = 256 / c4n;
    }
};

// block-level sum over the threads that share a channel quad, then one atomic per channel: vals[4] of each of K kinds
template <int K>
__device__ __forceinline__ void block_reduce_atomic(double (&v)[K][4], int C, double* __restrict__ sums /*[C][K]*/)
{
    __shared__ double red[256][K * 4 + 1];
    const int CH = C < 64 ? C : 64;
    const int c4n = CH >> 2;
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
        for (int j = 0; j < 4; j++) red[threadIdx.x][k * 4 + j] = v[k][j];
    __syncthreads();
    if ((int)threadIdx.x < c4n) {
        double t[K][4];
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
            for (int j = 0; j < 4; j++) t[k][j] = 0;
        for (int r = threadIdx.x; r < 256; r += c4n)
#pragma unroll
            for (int k = 0; k < K; k++)
#pragma unroll
                for (int j = 0; j < 4; j++) t[k][j] += red[r][k * 4 + j];
        const int c = blockIdx.y * CH + threadIdx.x * 4;
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
            for (int j = 0; j < 4; j++) atomicAdd(&sums[(size_t)(c + j) * K + k], t[k][j]);
    }
}

// sums[c] = (sum x, sum x^2) over all rows
__global__ __launch_bounds__(256) void k_bn_stats(const float* __restrict__ x, long rows, int C, long rows_per_block,
                                                  double* __restrict__ sums)
{
    const BnLay L(C);
    const long lo = (long)blockIdx.x * rows_per_block, hi = min(lo + rows_per_block, rows);
    double v[2][4] = {};
    for (long p = lo + L.prow; p < hi; p += L.pstep) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)p * C + L.c0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            v[0][j] += (double)a[j];
            v[1][j] += (double)a[j] * (double)a[j];
        }
    }
    block_reduce_atomic<2>(v, C, sums);
}

// per channel: (mean, rstd) -> mr, (scale, shift) -> ss; training also updates the running statistics
__global__ void k_bn_finalize(const double* __restrict__ sums, double count, int C, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float* __restrict__ running_mean,
                              float* __restrict__ running_var, float momentum, float eps, int training,
                              float2* __restrict__ mr, float2* __restrict__ ss)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double mean, var;
    if (training) {
        mean = sums[2 * c] / count;
        var = sums[2 * c + 1] / count - mean * mean;
        if (var < 0) var = 0;
        if (running_mean && running_var) {
            const double unbiased = count > 1 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    } else {
        mean = running_mean[c];
        var = running_var[c];
    }
    // torch computes invstd = 1 / sqrt(var + eps) in the accumulate type (float for float tensors)
    const float rstd = 1.f / sqrtf((float)var + eps);
    const float m = (float)mean;
    mr[c] = make_float2(m, rstd);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * rstd;
    ss[c] = make_float2(sc, b - m * sc);
}

// out = f(x * scale + shift + res), f = ReLU when relu
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, const float2* __restrict__ ss,
                                                  const float* __restrict__ res, float* __restrict__ out, long rows,
                                                  int C, long rows_per_block, int relu)
{
    const BnLay L(C);
    float sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float2 v = ss[L.c0 + j];
        sc[j] = v.x;
        sh[j] = v.y;
    }
    const long lo = (long)blockIdx.x * rows_per_block, hi = min(lo + rows_per_block, rows);
    for (long p = lo + L.prow; p < hi; p += L.pstep) {
        const size_t off = (size_t)p * C + L.c0;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + off);
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (res) r = *reinterpret_cast<const f32x4*>(res + off);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float y = v[j] * sc[j] + sh[j] + r[j];
            v[j] = relu ? fmaxf(y, 0.f) : y;
        }
        *reinterpret_cast<f32x4*>(out + off) = v;
    }
}

// gm = g masked by the ReLU (out > 0) -- also the gradient of the residual branch -- and
// sums[c] = (sum gm, sum gm * xhat), xhat = (x - mean) * rstd
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const float* __restrict__ g, const float* __restrict__ out,
                                                       const float* __restrict__ x, const float2* __restrict__ mr,
                                                       float* __restrict__ gm, long rows, int C, long rows_per_block,
                                                       int relu, double* __restrict__ sums)
{
    const BnLay L(C);
    float mean[4], rstd[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float2 v = mr[L.c0 + j];
        mean[j] = v.x;
        rstd[j] = v.y;
    }
    const long lo = (long)blockIdx.x * rows_per_block, hi = min(lo + rows_per_block, rows);
    double v[2][4] = {};
    for (long p = lo + L.prow; p < hi; p += L.pstep) {
        const size_t off = (size_t)p * C + L.c0;
        f32x4 gg = *reinterpret_cast<const f32x4*>(g + off);
        const f32x4 xx = *reinterpret_cast<const f32x4*>(x + off);
        if (relu) {
            const f32x4 oo = *reinterpret_cast<const f32x4*>(out + off);
#pragma unroll
            for (int j = 0; j < 4; j++) gg[j] = oo[j] > 0.f ? gg[j] : 0.f;
        }
        *reinterpret_cast<f32x4*>(gm + off) = gg;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float xh = (xx[j] - mean[j]) * rstd[j];
            v[0][j] += (double)gg[j];
            v[1][j] += (double)gg[j] * (double)xh;
        }
    }
    block_reduce_atomic<2>(v, C, sums);
}

// dx = gamma * rstd * (gm - mean(gm) - xhat * mean(gm * xhat))   (training)
//    = gamma * rstd * gm                                          (eval: the statistics are constants)
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ gm, const float* __restrict__ x,
                                                      const float2* __restrict__ mr, const float* __restrict__ gamma,
                                                      const double* __restrict__ sums, double count, int training,
                                                      float* __restrict__ dx, long rows, int C, long rows_per_block)
{
    const BnLay L(C);
    float mean[4], rstd[4], k1[4], mg[4], mgx[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float2 v = mr[L.c0 + j];
        mean[j] = v.x;
        rstd[j] = v.y;
        k1[j] = (gamma ? gamma[L.c0 + j] : 1.f) * v.y;
        mg[j] = training ? (float)(sums[2 * (L.c0 + j)] / count) : 0.f;
        mgx[j] = training ? (float)(sums[2 * (L.c0 + j) + 1] / count) : 0.f;
    }
    const long lo = (long)blockIdx.x * rows_per_block, hi = min(lo + rows_per_block, rows);
    for (long p = lo + L.prow; p < hi; p += L.pstep) {
        const size_t off = (size_t)p * C + L.c0;
        f32x4 gg = *reinterpret_cast<const f32x4*>(gm + off);
        const f32x4 xx = *reinterpret_cast<const f32x4*>(x + off);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float xh = (xx[j] - mean[j]) * rstd[j];
            gg[j] = k1[j] * (gg[j] - mg[j] - xh * mgx[j]);
        }
        *reinterpret_cast<f32x4*>(dx + off) = gg;
    }
}

// ---- MaxPool2d(kernel 3, stride 2, padding 1): window (2 oy - 1 + ky, 2 ox - 1 + kx); idx = ky * 3 + kx of the FIRST
// maximum in scan order (ATen's `val > maxval` rule), kept for the backward pass
__global__ __launch_bounds__(256) void k_maxpool_fwd(const float* __restrict__ x, int N, int H, int W, int C, int OH,
                                                     int OW, float* __restrict__ out, int8_t* __restrict__ idx)
{
    const int c4n = C >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)N * OH * OW * c4n;
    if (i >= total) return;
    const int c = (int)(i % c4n) * 4;
    long q = i / c4n;
    const int ox = (int)(q % OW);
    q /= OW;
    const int oy = (int)(q % OH), n = (int)(q / OH);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {-1, -1, -1, -1};
    for (int ky = 0; ky < 3; ky++) {
        const int iy = 2 * oy - 1 + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < 3; kx++) {
            const int ix = 2 * ox - 1 + kx;
            if (ix < 0 || ix >= W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + iy) * W + ix) * C + c);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (v[j] > best[j] || bi[j] < 0 || v[j] != v[j]) {
                    best[j] = v[j];
                    bi[j] = ky * 3 + kx;
                }
        }
    }
    const size_t o = (((size_t)n * OH + oy) * OW + ox) * C + c;
    *reinterpret_cast<f32x4*>(out + o) = best;
    idx[o] = (int8_t)bi[0];
    idx[o + 1] = (int8_t)bi[1];
    idx[o + 2] = (int8_t)bi[2];
    idx[o + 3] = (int8_t)bi[3];
}

// gather form (no atomics, deterministic): input (iy, ix) is tap (ky, kx) of output ((iy + 1 - ky) / 2, (ix + 1 - kx) / 2)
__global__ __launch_bounds__(256) void k_maxpool_bwd(const float* __restrict__ g, const int8_t* __restrict__ idx, int N,
                                                     int H, int W, int C, int OH, int OW, float* __restrict__ gin)
{
    const int c4n = C >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)N * H * W * c4n;
    if (i >= total) return;
    const int c = (int)(i % c4n) * 4;
    long q = i / c4n;
    const int ix = (int)(q % W);
    q /= W;
    const int iy = (int)(q % H), n = (int)(q / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < 3; ky++) {
        const int t = iy + 1 - ky;
        if (t < 0 || (t & 1)) continue;
        const int oy = t >> 1;
        if (oy >= OH) continue;
        for (int kx = 0; kx < 3; kx++) {
            const int u = ix + 1 - kx;
            if (u < 0 || (u & 1)) continue;
            const int ox = u >> 1;
            if (ox >= OW) continue;
            const size_t o = (((size_t)n * OH + oy) * OW + ox) * C + c;
            const f32x4 gg = *reinterpret_cast<const f32x4*>(g + o);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (idx[o + j] == ky * 3 + kx) acc[j] += gg[j];
        }
    }
    *reinterpret_cast<f32x4*>(gin + (((size_t)n * H + iy) * W + ix) * C + c) = acc;
}

// ---- AdaptiveAvgPool2d(1) (derenderer.py:26): out[n, c] = mean over the HW positions; backward broadcasts g / HW
__global__ __launch_bounds__(256) void k_avgpool_fwd(const float* __restrict__ x, int N, int HW, int C,
                                                     float* __restrict__ out)
{
    const int c4n = C >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * c4n) return;
    const int n = i / c4n, c = (i % c4n) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < HW; p++) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)n * HW + p) * C + c);
        s += v;
    }
    const float inv = 1.f / (float)HW;
    s[0] *= inv;
    s[1] *= inv;
    s[2] *= inv;
    s[3] *= inv;
    *reinterpret_cast<f32x4*>(out + (size_t)n * C + c) = s;
}

__global__ __launch_bounds__(256) void k_avgpool_bwd(const float* __restrict__ g, int N, int HW, int C,
                                                     float* __restrict__ gin)
{
    const int c4n = C >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * HW * c4n) return;
    const int c = (int)(i % c4n) * 4;
    const long q = i / c4n;
    const int n = (int)(q / HW);
    f32x4 v = *reinterpret_cast<const f32x4*>(g + (size_t)n * C + c);
    const float inv = 1.f / (float)HW;
    v[0] *= inv;
    v[1] *= inv;
    v[2] *= inv;
    v[3] *= inv;
    *reinterpret_cast<f32x4*>(gin + (size_t)q * C + c) = v;
}

static bool bn_shape_ok(long rows, int C) { return rows > 0 && C >= 4 && (C % 4) == 0 && (C <= 64 ? 64 % C == 0 || C == 64 : C % 64 == 0); }

static void bn_grid(long rows, int C, dim3& grid, long& rpb)
{
    const int CH = C < 64 ? C : 64;
    const int chunks = C / CH;
    // ~2048 blocks in total, at least 256 rows each
    long want = 2048 / chunks;
    if (want < 1) want = 1;
    rpb = (rows + want - 1) / want;
    if (rpb < 256) rpb = 256;
    grid = dim3(cdiv(rows, rpb), chunks, 1);
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_bn_forward(const float* x, long rows, int C, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float momentum, float eps, int training, const float* res, int relu,
                           float* out, float* mr, float* ss, double* sums, sdnStream stream)
{
    if (!x || !out || !mr || !ss || !bn_shape_ok(rows, C)) return fail(SDN_EINVAL, "sdn_bn_forward: bad arguments (rows %ld, C %d)", rows, C);
    if (training && !sums) return fail(SDN_EINVAL, "sdn_bn_forward: training needs the sums scratch");
    if (!training && (!running_mean || !running_var)) return fail(SDN_EINVAL, "sdn_bn_forward: eval mode needs the running statistics");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid;
    long rpb;
    bn_grid(rows, C, grid, rpb);
    if (training) {
        if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, st) != hipSuccess) return fail(SDN_ELAUNCH, "sdn_bn_forward: memset");
        k_bn_stats<<<grid, 256, 0, st>>>(x, rows, C, rpb, sums);
    }
    k_bn_finalize<<<cdiv(C, 256), 256, 0, st>>>(sums, (double)rows, C, gamma, beta, running_mean, running_var, momentum, eps,
                                               training, (float2*)mr, (float2*)ss);
    k_bn_apply<<<grid, 256, 0, st>>>(x, (const float2*)ss, res, out, rows, C, rpb, relu);
    return check_launch("sdn_bn_forward");
}

SDN_API int sdn_bn_backward(const float* g, const float* out, const float* x, const float* mr, const float* gamma, long rows,
                            int C, int training, int relu, float* gm, float* dx, double* sums, sdnStream stream)
{
    if (!g || !x || !mr || !gm || !dx || !sums || (relu && !out) || !bn_shape_ok(rows, C))
        return fail(SDN_EINVAL, "sdn_bn_backward: bad arguments (rows %ld, C %d)", rows, C);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid;
    long rpb;
    bn_grid(rows, C, grid, rpb);
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, st) != hipSuccess) return fail(SDN_ELAUNCH, "sdn_bn_backward: memset");
    k_bn_bwd_reduce<<<grid, 256, 0, st>>>(g, out, x, (const float2*)mr, gm, rows, C, rpb, relu, sums);
    k_bn_bwd_apply<<<grid, 256, 0, st>>>(gm, x, (const float2*)mr, gamma, sums, (double)rows, training, dx, rows, C, rpb);
    return check_launch("sdn_bn_backward");
}

SDN_API int sdn_maxpool3x3s2_fwd(const float* x, int N, int H, int W, int C, float* out, int8_t* idx, sdnStream stream)
{
    if (!x || !out || !idx || N <= 0 || H <= 0 || W <= 0 || C < 4 || C % 4) return fail(SDN_EINVAL, "sdn_maxpool3x3s2_fwd: bad arguments");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * OH * OW * (C / 4);
    k_maxpool_fwd<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(x, N, H, W, C, OH, OW, out, idx);
    return check_launch("sdn_maxpool3x3s2_fwd");
}

SDN_API int sdn_maxpool3x3s2_bwd(const float* g, const int8_t* idx, int N, int H, int W, int C, float* gin, sdnStream stream)
{
    if (!g || !gin || !idx || N <= 0 || H <= 0 || W <= 0 || C < 4 || C % 4) return fail(SDN_EINVAL, "sdn_maxpool3x3s2_bwd: bad arguments");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * H * W * (C / 4);
    k_maxpool_bwd<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(g, idx, N, H, W, C, OH, OW, gin);
    return check_launch("sdn_maxpool3x3s2_bwd");
}

SDN_API int sdn_avgpool_global(const float* x, int N, int HW, int C, float* out, int backward, sdnStream stream)
{
    if (!x || !out || N <= 0 || HW <= 0 || C < 4 || C % 4) return fail(SDN_EINVAL, "sdn_avgpool_global: bad arguments");
    if (!backward)
        k_avgpool_fwd<<<cdiv((long)N * (C / 4), 256), 256, 0, (hipStream_t)stream>>>(x, N, HW, C, out);
    else
        k_avgpool_bwd<<<cdiv((long)N * HW * (C / 4), 256), 256, 0, (hipStream_t)stream>>>(x, N, HW, C, out);
    return check_launch("sdn_avgpool_global");
}
