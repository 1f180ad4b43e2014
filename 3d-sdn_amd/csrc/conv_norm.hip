// HBM-bound companions of the textural conv kernels: InstanceNorm apply / backward, activation backward, reflection
// fold, weight packing.  All tensors are channels-last fp32 [N, H, W, Cp] with Cp a multiple of 16; a thread owns four
// consecutive channels (16-B accesses, a wave covers 1 KiB of one or more positions).
//
// Reference: nn.InstanceNorm2d(affine=False, track_running_stats=True) + ReLU / LeakyReLU(0.2) / Tanh / ReflectionPad2d
// as separate cuDNN / ATen kernels, textural/models/networks.py:24-30, 211-283, 412-449.  InstanceNorm needs the whole
// plane's statistics, so it cannot live inside the conv kernel: the conv epilogue accumulates sum / sum of squares
// (fp64 atomics), and ONE pass here normalises in place, adds the residual of a ResnetBlock (networks.py:281-283),
// optionally materialises LeakyReLU (discriminator features are returned to the caller) and updates the running
// statistics.  ReLU after a norm is never materialised: consumers apply it on load.
#include "conv_common.h"
#include "sdn_common.h"

namespace sdn {

// stats: [N, STAT_SLOTS, Cp, 2] (sum, sum of squares), see conv_common.h
__device__ __forceinline__ void load_stats(const double* stats, int n, int c, int Cp, double& s1, double& s2)
{
    s1 = s2 = 0;
#pragma unroll
    for (int k = 0; k < STAT_SLOTS; k++) {
        const double* p = stats + (((size_t)n * STAT_SLOTS + k) * Cp + c) * 2;
        s1 += p[0];
        s2 += p[1];
    }
}

__device__ __forceinline__ void mean_rstd(const double* stats, int n, int c, int Cp, double inv_cnt, float eps,
                                          float& mean, float& rstd)
{
    double a, b;
    load_stats(stats, n, c, Cp, a, b);
    const double m = a * inv_cnt;
    double var = b * inv_cnt - m * m;
    if (var < 0) var = 0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// xhat = (z - mean) * rstd in place;  y = xhat (act 0) or LeakyReLU(xhat) (act 1) in place;
// out2 (optional) = y + f(res), f = ReLU when res_relu.
__global__ __launch_bounds__(256) void k_in_apply(float* __restrict__ z, const double* __restrict__ stats,
                                                  const float* __restrict__ res, float* __restrict__ out2, int HW,
                                                  int Cp, float eps, int act, int res_relu, int pix_per_block)
{
    const int n = blockIdx.y;
    const int c4n = Cp >> 2;
    const int c4 = threadIdx.x % c4n, prow = threadIdx.x / c4n, pstep = 256 / c4n;
    float mean[4], rstd[4];
    const double inv_cnt = 1.0 / (double)HW;
#pragma unroll
    for (int j = 0; j < 4; j++) mean_rstd(stats, n, c4 * 4 + j, Cp, inv_cnt, eps, mean[j], rstd[j]);
    const int p_lo = blockIdx.x * pix_per_block, p_hi = min(p_lo + pix_per_block, HW);
    for (int p = p_lo + prow; p < p_hi; p += pstep) {
        const size_t off = ((size_t)n * HW + p) * Cp + c4 * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(z + off);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float x = (v[j] - mean[j]) * rstd[j];
            if (act == 1) x = x > 0.f ? x : 0.2f * x;
            v[j] = x;
        }
        *reinterpret_cast<f32x4*>(z + off) = v;
        if (out2) {
            f32x4 r = *reinterpret_cast<const f32x4*>(res + off);
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] += res_relu ? fmaxf(r[j], 0.f) : r[j];
            *reinterpret_cast<f32x4*>(out2 + off) = v;
        }
    }
}

// running_mean / running_var of nn.InstanceNorm2d(track_running_stats=True) in training mode: batch mean of the
// per-instance mean and of the UNBIASED per-instance variance, momentum 0.1; num_batches_tracked += 1.
__global__ void k_in_running(const double* __restrict__ stats, int N, int C, int Cp, int HW, float momentum,
                             float* __restrict__ running_mean, float* __restrict__ running_var,
                             long long* __restrict__ num_batches)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && num_batches) *num_batches += 1;
    if (c >= C) return;
    double ms = 0, vs = 0;
    for (int n = 0; n < N; n++) {
        double a, b;
        load_stats(stats, n, c, Cp, a, b);
        const double m = a / HW;
        double var = b / HW - m * m;
        if (var < 0) var = 0;
        ms += m;
        vs += HW > 1 ? var * HW / (HW - 1.0) : var;
    }
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)(ms / N);
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(vs / N);
}

// effective gradient wrt xhat and xhat itself from the stored tensor
//   mode 0: stored xhat, no activation            g_eff = g
//   mode 1: stored xhat, consumers apply ReLU     g_eff = g * (xhat > 0)
//   mode 2: stored y = LeakyReLU(xhat)            xhat = y > 0 ? y : 5 y,  g_eff = g * (y > 0 ? 1 : 0.2)
__device__ __forceinline__ void eff(float g, float s, int mode, float& ge, float& xh)
{
    if (mode == 1) {
        ge = s > 0.f ? g : 0.f;
        xh = s;
    } else if (mode == 2) {
        ge = s > 0.f ? g : 0.2f * g;
        xh = s > 0.f ? s : 5.0f * s;
    } else {
        ge = g;
        xh = s;
    }
}

// sums[n, c] = { sum_p g_eff, sum_p g_eff * xhat }   (fp64 atomics into a zeroed buffer)
__global__ __launch_bounds__(256) void k_in_bwd_reduce(const float* __restrict__ g, const float* __restrict__ stored,
                                                       double* __restrict__ sums, int HW, int Cp, int mode,
                                                       int pix_per_block)
{
    __shared__ float red[256][8];
    const int n = blockIdx.y;
    const int c4n = Cp >> 2;
    const int c4 = threadIdx.x % c4n, prow = threadIdx.x / c4n, pstep = 256 / c4n;
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const int p_lo = blockIdx.x * pix_per_block, p_hi = min(p_lo + pix_per_block, HW);
    for (int p = p_lo + prow; p < p_hi; p += pstep) {
        const size_t off = ((size_t)n * HW + p) * Cp + c4 * 4;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + off);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(stored + off);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float ge, xh;
            eff(gv[j], sv[j], mode, ge, xh);
            s1[j] += ge;
            s2[j] += ge * xh;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        red[threadIdx.x][j] = s1[j];
        red[threadIdx.x][4 + j] = s2[j];
    }
    __syncthreads();
    if (prow == 0) {
        for (int r = 1; r < pstep; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                s1[j] += red[r * c4n + c4][j];
                s2[j] += red[r * c4n + c4][4 + j];
            }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double* d = sums + ((size_t)n * Cp + c4 * 4 + j) * 2;
            unsafeAtomicAdd(d, (double)s1[j]);
            unsafeAtomicAdd(d + 1, (double)s2[j]);
        }
    }
}

// dz = rstd * (g_eff - mean(g_eff) - xhat * mean(g_eff * xhat)), written over g
__global__ __launch_bounds__(256) void k_in_bwd_apply(float* __restrict__ g, const float* __restrict__ stored,
                                                      const double* __restrict__ sums,
                                                      const double* __restrict__ fwd_stats, int HW, int Cp, float eps,
                                                      int mode, int pix_per_block)
{
    const int n = blockIdx.y;
    const int c4n = Cp >> 2;
    const int c4 = threadIdx.x % c4n, prow = threadIdx.x / c4n, pstep = 256 / c4n;
    float m1[4], m2[4], rstd[4];
    const double inv_cnt = 1.0 / (double)HW;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const size_t k = ((size_t)n * Cp + c4 * 4 + j) * 2;
        float mean;
        mean_rstd(fwd_stats, n, c4 * 4 + j, Cp, inv_cnt, eps, mean, rstd[j]);
        m1[j] = (float)(sums[k] * inv_cnt);
        m2[j] = (float)(sums[k + 1] * inv_cnt);
    }
    const int p_lo = blockIdx.x * pix_per_block, p_hi = min(p_lo + pix_per_block, HW);
    for (int p = p_lo + prow; p < p_hi; p += pstep) {
        const size_t off = ((size_t)n * HW + p) * Cp + c4 * 4;
        f32x4 gv = *reinterpret_cast<const f32x4*>(g + off);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(stored + off);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float ge, xh;
            eff(gv[j], sv[j], mode, ge, xh);
            gv[j] = rstd[j] * (ge - m1[j] - xh * m2[j]);
        }
        *reinterpret_cast<f32x4*>(g + off) = gv;
    }
}

// layers without a norm: dz = g * act'(y) in place (act 1 LeakyReLU from y's sign, act 2 tanh: 1 - y^2, act 3 ReLU
// deferred: stored pre-activation, mask by its sign);  bias_grad[c] += sum dz  (fp32 atomics, optional)
__global__ __launch_bounds__(256) void k_act_bwd(float* __restrict__ g, const float* __restrict__ y,
                                                 float* __restrict__ bias_grad, long npos, int Cp, int act,
                                                 int pix_per_block)
{
    __shared__ float red[256][4];
    const int c4n = Cp >> 2;
    const int c4 = threadIdx.x % c4n, prow = threadIdx.x / c4n, pstep = 256 / c4n;
    float s[4] = {0, 0, 0, 0};
    const long p_lo = (long)blockIdx.x * pix_per_block, p_hi = min(p_lo + (long)pix_per_block, npos);
    for (long p = p_lo + prow; p < p_hi; p += pstep) {
        const size_t off = (size_t)p * Cp + c4 * 4;
        f32x4 gv = *reinterpret_cast<const f32x4*>(g + off);
        if (act) {
            const f32x4 yv = *reinterpret_cast<const f32x4*>(y + off);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (act == 1)
                    gv[j] = yv[j] > 0.f ? gv[j] : 0.2f * gv[j];
                else if (act == 2)
                    gv[j] = gv[j] * (1.f - yv[j] * yv[j]);
                else
                    gv[j] = yv[j] > 0.f ? gv[j] : 0.f;
            }
            *reinterpret_cast<f32x4*>(g + off) = gv;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) s[j] += gv[j];
    }
    if (!bias_grad) return;
#pragma unroll
    for (int j = 0; j < 4; j++) red[threadIdx.x][j] = s[j];
    __syncthreads();
    if (prow == 0) {
        for (int r = 1; r < pstep; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) s[j] += red[r * c4n + c4][j];
#pragma unroll
        for (int j = 0; j < 4; j++) unsafeAtomicAdd(bias_grad + c4 * 4 + j, s[j]);
    }
}

// Adjoint of ReflectionPad2d(P): gp [N, H+2P, W+2P, Cp] -> out [N, H, W, Cp] (=, or += with accumulate)
__global__ __launch_bounds__(256) void k_reflect_fold(const float* __restrict__ gp, float* __restrict__ out, int N, int H,
                                                      int W, int Cp, int Pd, int accumulate)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int c4n = Cp >> 2;
    const long total = (long)N * H * W * c4n;
    if (i >= total) return;
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const int Hp = H + 2 * Pd, Wp = W + 2 * Pd;
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y + Pd;
    if (y >= 1 && y <= Pd) ys[ny++] = Pd - y;
    if (y <= H - 2 && y >= H - 1 - Pd) ys[ny++] = 2 * (H - 1) - y + Pd;
    xs[nx++] = x + Pd;
    if (x >= 1 && x <= Pd) xs[nx++] = Pd - x;
    if (x <= W - 2 && x >= W - 1 - Pd) xs[nx++] = 2 * (W - 1) - x + Pd;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < ny; a++)
        for (int b = 0; b < nx; b++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(gp + (((size_t)n * Hp + ys[a]) * Wp + xs[b]) * Cp + c4 * 4);
            s += v;
        }
    float* o = out + (((size_t)n * H + y) * W + x) * Cp + c4 * 4;
    if (accumulate) s += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = s;
}

// packed[r, t * Ccp + c] = W[r * sr + c * sc + tapidx[t]] split to bf16 hi / lo; zero where r >= R, c >= C or in the
// K padding.  (sr, sc) select Conv2d [O,I,kh,kw] vs ConvTranspose2d [I,O,kh,kw] and forward vs data-gradient use.
__global__ __launch_bounds__(256) void k_pack_weights(const float* __restrict__ w, int R, int C, long sr, long sc,
                                                      const int* __restrict__ tapidx, int ntaps, int Ccp, int Kp,
                                                      int rows, __bf16* __restrict__ hi, __bf16* __restrict__ lo)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * Kp) return;
    const int r = (int)(i / Kp), k = (int)(i % Kp);
    const int t = k / Ccp, c = k % Ccp;
    float v = 0.f;
    if (r < R && t < ntaps && c < C) v = w[(size_t)r * sr + (size_t)c * sc + tapidx[t]];
    const __bf16 h = (__bf16)v;
    hi[i] = h;
    if (lo) lo[i] = (__bf16)(v - (float)h);
}

// grad_w[r * sr + c * sc + tapidx[t]] += dw[r, t * Ccp + c]  (the inverse map; every parameter element is hit by at most
// one (r, c, t) per call, so a plain read-modify-write is race-free)
__global__ __launch_bounds__(256) void k_unpack_grad(const float* __restrict__ dw, int R, int C, long sr, long sc,
                                                     const int* __restrict__ tapidx, int ntaps, int Ccp,
                                                     float* __restrict__ grad_w)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long ncols = (long)ntaps * Ccp;
    if (i >= (long)R * ncols) return;
    const int r = (int)(i / ncols);
    const int k = (int)(i % ncols);
    const int t = k / Ccp, c = k % Ccp;
    if (c >= C) return;
    grad_w[(size_t)r * sr + (size_t)c * sc + tapidx[t]] += dw[i];
}

// reductions end in one atomic per (block, channel): keep the block count near `target` in total
static int ppb_for_reduce(long npos, int Cp, int batch, int target)
{
    const int per_iter = 1024 / Cp > 0 ? 1024 / Cp : 1;
    long per_image = target / (batch > 0 ? batch : 1);
    if (per_image < 16) per_image = 16;
    long ppb = (npos + per_image - 1) / per_image;
    ppb = ((ppb + per_iter - 1) / per_iter) * per_iter;
    if (ppb < per_iter) ppb = per_iter;
    return (int)ppb;
}

static int ppb_for(long npos, int Cp)
{
    // aim at ~2048 blocks; each block iteration covers 1024 / Cp positions
    const int per_iter = 1024 / Cp > 0 ? 1024 / Cp : 1;
    long ppb = (npos + 2047) / 2048;
    ppb = ((ppb + per_iter - 1) / per_iter) * per_iter;
    if (ppb < per_iter) ppb = per_iter;
    return (int)ppb;
}

}  // namespace sdn

using namespace sdn;

static int check_cp(const char* who, int Cp)
{
    // a block's 256 threads cover 1024 / Cp whole positions per iteration: Cp must be a power of two in [16, 1024]
    if (Cp < 16 || Cp > 1024 || (Cp & (Cp - 1)))
        return fail(SDN_EINVAL, "%s: padded channel count %d must be a power of two in [16, 1024]", who, Cp);
    return SDN_OK;
}

SDN_API int sdn_in_apply(float* z, const double* stats, const float* res, float* out2, int N, int HW, int C, int Cp,
                         float eps, int act, int res_relu, float momentum, float* running_mean, float* running_var,
                         long long* num_batches, sdnStream stream)
{
    int rc = check_cp("sdn_in_apply", Cp);
    if (rc) return rc;
    if (!z || !stats || (out2 && !res)) return fail(SDN_EINVAL, "sdn_in_apply: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int ppb = ppb_for(HW, Cp);
    hipLaunchKernelGGL(k_in_apply, dim3(cdiv(HW, ppb), N), dim3(256), 0, st, z, stats, res, out2, HW, Cp, eps, act,
                       res_relu, ppb);
    if ((rc = check_launch("k_in_apply"))) return rc;
    if (running_mean && running_var) {
        hipLaunchKernelGGL(k_in_running, dim3(cdiv(C, 256)), dim3(256), 0, st, stats, N, C, Cp, HW, momentum,
                           running_mean, running_var, num_batches);
        rc = check_launch("k_in_running");
    }
    return rc;
}

SDN_API int sdn_in_bwd(float* g, const float* stored, const double* fwd_stats, double* sums, int N, int HW, int Cp,
                       float eps, int mode, sdnStream stream)
{
    int rc = check_cp("sdn_in_bwd", Cp);
    if (rc) return rc;
    if (!g || !stored || !fwd_stats || !sums) return fail(SDN_EINVAL, "sdn_in_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)N * Cp, st) != hipSuccess) return fail(SDN_ELAUNCH, "sdn_in_bwd: memset");
    const int rppb = ppb_for_reduce(HW, Cp, N, 1024);
    hipLaunchKernelGGL(k_in_bwd_reduce, dim3(cdiv(HW, rppb), N), dim3(256), 0, st, g, stored, sums, HW, Cp, mode, rppb);
    if ((rc = check_launch("k_in_bwd_reduce"))) return rc;
    const int ppb = ppb_for(HW, Cp);
    hipLaunchKernelGGL(k_in_bwd_apply, dim3(cdiv(HW, ppb), N), dim3(256), 0, st, g, stored, sums, fwd_stats, HW, Cp, eps,
                       mode, ppb);
    return check_launch("k_in_bwd_apply");
}

SDN_API int sdn_act_bwd(float* g, const float* y, float* bias_grad, long npos, int Cp, int act, sdnStream stream)
{
    int rc = check_cp("sdn_act_bwd", Cp);
    if (rc) return rc;
    if (!g || (act && !y)) return fail(SDN_EINVAL, "sdn_act_bwd: null pointer");
    if (!act && !bias_grad) return SDN_OK;
    const int ppb = bias_grad ? ppb_for_reduce(npos, Cp, 1, 1024) : ppb_for(npos, Cp);
    hipLaunchKernelGGL(k_act_bwd, dim3(cdiv(npos, ppb)), dim3(256), 0, (hipStream_t)stream, g, y, bias_grad, npos, Cp,
                       act, ppb);
    return check_launch("k_act_bwd");
}

SDN_API int sdn_reflect_fold(const float* gp, float* out, int N, int H, int W, int Cp, int pad, int accumulate,
                             sdnStream stream)
{
    if (!gp || !out || (Cp & 3) || pad < 1 || pad >= H || pad >= W) return fail(SDN_EINVAL, "sdn_reflect_fold: bad argument");
    const long total = (long)N * H * W * (Cp / 4);
    hipLaunchKernelGGL(k_reflect_fold, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, gp, out, N, H, W, Cp,
                       pad, accumulate);
    return check_launch("k_reflect_fold");
}

SDN_API int sdn_conv_pack_weights(const float* w, int R, int C, long sr, long sc, const int32_t* tapidx, int ntaps,
                                  int Ccp, int Kp, int rows, void* hi, void* lo, sdnStream stream)
{
    if (!w || !tapidx || !hi || Kp < ntaps * Ccp || rows < R || Ccp < C) return fail(SDN_EINVAL, "sdn_conv_pack_weights: bad argument");
    hipLaunchKernelGGL(k_pack_weights, dim3(cdiv((long)rows * Kp, 256)), dim3(256), 0, (hipStream_t)stream, w, R, C, sr,
                       sc, tapidx, ntaps, Ccp, Kp, rows, (__bf16*)hi, (__bf16*)lo);
    return check_launch("k_pack_weights");
}

SDN_API int sdn_conv_unpack_grad(const float* dw, int R, int C, long sr, long sc, const int32_t* tapidx, int ntaps,
                                 int Ccp, float* grad_w, sdnStream stream)
{
    if (!dw || !tapidx || !grad_w || Ccp < C) return fail(SDN_EINVAL, "sdn_conv_unpack_grad: bad argument");
    hipLaunchKernelGGL(k_unpack_grad, dim3(cdiv((long)R * ntaps * Ccp, 256)), dim3(256), 0, (hipStream_t)stream, dw, R,
                       C, sr, sc, tapidx, ntaps, Ccp, grad_w);
    return check_launch("k_unpack_grad");
}
