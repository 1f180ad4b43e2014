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
#include "conv_pack.h"
#include "sdn_common.h"

namespace sdn {

// Thread layout shared by the kernels below: a block owns CH = min(Cp, 64) consecutive channels (blockIdx.z selects the
// chunk) and a slice of positions (blockIdx.x) of image blockIdx.y; its 256 threads are (CH/4 channel quads) x
// (1024/CH positions per iteration), so every wave access is 4 x 256 B contiguous.  Reductions therefore end in ONE
// atomic per (block, channel) with few blocks per channel.
struct Lay {
    int c0, prow, pstep;
    __device__ Lay(int Cp)
    {
        const int CH = Cp < 64 ? Cp : 64;
        const int c4n = CH >> 2;
        c0 = blockIdx.z * CH + (threadIdx.x % c4n) * 4;
        prow = threadIdx.x / c4n;
        pstep = 256 / c4n;
    }
};

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

// bf16 operand planes of four consecutive channels (conv_planes.hip): hi = bf16(x), lo = bf16(x - hi), 8-byte stores
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
__device__ __forceinline__ void store_planes(__bf16* __restrict__ hi, long stride, size_t off, const f32x4& v, int relu)
{
    float a = v[0], b = v[1], c = v[2], d = v[3];
    if (relu) {
        a = fmaxf(a, 0.f);
        b = fmaxf(b, 0.f);
        c = fmaxf(c, 0.f);
        d = fmaxf(d, 0.f);
    }
    const SplitBf16 p = split2(a, b), q = split2(c, d);
    *reinterpret_cast<bf16x4_t*>(hi + off) = bf16x4_t{p.hi[0], p.hi[1], q.hi[0], q.hi[1]};
    *reinterpret_cast<bf16x4_t*>(hi + stride + off) = bf16x4_t{p.lo[0], p.lo[1], q.lo[0], q.lo[1]};
}

// mr[n, c] = (mean, rstd) of nn.InstanceNorm2d: biased variance, eps inside the square root.  One thread per (n, c).
// The threads of n == 0 also update running_mean / running_var the way torch does in training mode: batch mean of the
// per-instance mean and of the UNBIASED per-instance variance, momentum 0.1 (num_batches_tracked is left alone, as
// torch's InstanceNorm leaves it).
__global__ __launch_bounds__(256) void k_in_finalize(const double* __restrict__ stats, int N, int C, int Cp, int HW,
                                                     float eps, float momentum, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var, float2* __restrict__ mr)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * Cp) return;
    const int n = i / Cp, c = i - n * Cp;
    double a, b;
    load_stats(stats, n, c, Cp, a, b);
    const double m = a / HW;
    double var = b / HW - m * m;
    if (var < 0) var = 0;
    mr[i] = make_float2((float)m, (float)(1.0 / sqrt(var + (double)eps)));
    if (n == 0 && c < C && running_mean && running_var) {
        double ms = 0, vs = 0;
        for (int k = 0; k < N; k++) {
            load_stats(stats, k, c, Cp, a, b);
            const double mk = a / HW;
            double vk = b / HW - mk * mk;
            if (vk < 0) vk = 0;
            ms += mk;
            vs += HW > 1 ? vk * HW / (HW - 1.0) : vk;
        }
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)(ms / N);
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(vs / N);
    }
}

// xhat = (z - mean) * rstd in place;  y = xhat (act 0) or LeakyReLU(xhat) (act 1) in place;
// out2 (optional) = y + f(res), f = ReLU when res_relu.  planes (optional): the bf16 (hi, lo) split of what the tensor's
// consumers multiply -- out2 when there is one, else y, through ReLU when planes_relu (the deferred ReLU).
__global__ __launch_bounds__(256) void k_in_apply(float* __restrict__ z, const float2* __restrict__ mr,
                                                  const float* __restrict__ res, float* __restrict__ out2, int HW,
                                                  int Cp, int act, int res_relu, int pix_per_block,
                                                  __bf16* __restrict__ planes, long plane_stride, int planes_relu)
{
    const Lay L(Cp);
    const int n = blockIdx.y;
    float mean[4], rstd[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float2 v = mr[(size_t)n * Cp + L.c0 + j];
        mean[j] = v.x;
        rstd[j] = v.y;
    }
    const int p_lo = blockIdx.x * pix_per_block, p_hi = min(p_lo + pix_per_block, HW);
    for (int p = p_lo + L.prow; p < p_hi; p += L.pstep) {
        const size_t off = ((size_t)n * HW + p) * Cp + L.c0;
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
        if (planes) store_planes(planes, plane_stride, off, v, planes_relu);
    }
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

// block-level sum of per-thread float[NV] partials over the position rows, then one fp64 atomic per value by row 0
template <int NV>
__device__ __forceinline__ void block_reduce_rows(float (&v)[NV], const Lay& L, float (*red)[NV])
{
#pragma unroll
    for (int j = 0; j < NV; j++) red[threadIdx.x][j] = v[j];
    __syncthreads();
    if (L.prow == 0) {
        const int c4n = 256 / L.pstep;
        for (int r = 1; r < L.pstep; r++)
#pragma unroll
            for (int j = 0; j < NV; j++) v[j] += red[r * c4n + threadIdx.x][j];
    }
}

// sums[n, c] = { sum_p g_eff, sum_p g_eff * xhat }   (fp64 atomics into a zeroed buffer)
__global__ __launch_bounds__(256) void k_in_bwd_reduce(const float* __restrict__ g, const float* __restrict__ stored,
                                                       double* __restrict__ sums, int HW, int Cp, int mode,
                                                       int pix_per_block)
{
    __shared__ float red[256][8];
    const Lay L(Cp);
    const int n = blockIdx.y;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int p_lo = blockIdx.x * pix_per_block, p_hi = min(p_lo + pix_per_block, HW);
    for (int p = p_lo + L.prow; p < p_hi; p += L.pstep) {
        const size_t off = ((size_t)n * HW + p) * Cp + L.c0;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + off);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(stored + off);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float ge, xh;
            eff(gv[j], sv[j], mode, ge, xh);
            s[j] += ge;
            s[4 + j] += ge * xh;
        }
    }
    block_reduce_rows<8>(s, L, red);
    if (L.prow == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double* d = sums + ((size_t)n * Cp + L.c0 + j) * 2;
            unsafeAtomicAdd(d, (double)s[j]);
            unsafeAtomicAdd(d + 1, (double)s[4 + j]);
        }
    }
}

// dz = rstd * (g_eff - mean(g_eff) - xhat * mean(g_eff * xhat)), written over g (and, optionally, as bf16 planes)
__global__ __launch_bounds__(256) void k_in_bwd_apply(float* __restrict__ g, const float* __restrict__ stored,
                                                      const double* __restrict__ sums, const float2* __restrict__ mr,
                                                      int HW, int Cp, int mode, int pix_per_block,
                                                      __bf16* __restrict__ planes, long plane_stride)
{
    const Lay L(Cp);
    const int n = blockIdx.y;
    float m1[4], m2[4], rstd[4];
    const double inv_cnt = 1.0 / (double)HW;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const size_t k = (size_t)n * Cp + L.c0 + j;
        rstd[j] = mr[k].y;
        m1[j] = (float)(sums[2 * k] * inv_cnt);
        m2[j] = (float)(sums[2 * k + 1] * inv_cnt);
    }
    const int p_lo = blockIdx.x * pix_per_block, p_hi = min(p_lo + pix_per_block, HW);
    for (int p = p_lo + L.prow; p < p_hi; p += L.pstep) {
        const size_t off = ((size_t)n * HW + p) * Cp + L.c0;
        f32x4 gv = *reinterpret_cast<const f32x4*>(g + off);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(stored + off);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float ge, xh;
            eff(gv[j], sv[j], mode, ge, xh);
            gv[j] = rstd[j] * (ge - m1[j] - xh * m2[j]);
        }
        *reinterpret_cast<f32x4*>(g + off) = gv;
        if (planes) store_planes(planes, plane_stride, off, gv, 0);
    }
}

// layers without a norm: dz = g * act'(y) in place (act 1 LeakyReLU from y's sign, act 2 tanh: 1 - y^2, act 3 ReLU
// deferred: stored pre-activation, mask by its sign);  bias_grad[c] += sum dz  (fp32 atomics, optional).
// Positions of all images are one flat range (blockIdx.y unused).
__global__ __launch_bounds__(256) void k_act_bwd(float* __restrict__ g, const float* __restrict__ y,
                                                 float* __restrict__ bias_grad, long npos, int Cp, int act,
                                                 int pix_per_block, __bf16* __restrict__ planes, long plane_stride)
{
    __shared__ float red[256][4];
    const Lay L(Cp);
    float s[4] = {0, 0, 0, 0};
    const long p_lo = (long)blockIdx.x * pix_per_block, p_hi = min(p_lo + (long)pix_per_block, npos);
    for (long p = p_lo + L.prow; p < p_hi; p += L.pstep) {
        const size_t off = (size_t)p * Cp + L.c0;
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
        if (planes) store_planes(planes, plane_stride, off, gv, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) s[j] += gv[j];
    }
    if (!bias_grad) return;
    block_reduce_rows<4>(s, L, red);
    if (L.prow == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) unsafeAtomicAdd(bias_grad + L.c0 + j, s[j]);
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

// The logical weight matrix  Wm[r, k(t, c)] = W[r * sr + c * sc + tapidx[t]],  k = t * Ccp + c -- or, when Ccp % 32 == 0 and
// Kp == ntaps * Ccp, channel-block-major: k = ((c / 32) * ntaps + t) * 32 + c % 32  (zero where r >= R, c >= C or in the K
// padding; (sr, sc) select Conv2d [O,I,kh,kw] vs ConvTranspose2d [I,O,kh,kw] and forward vs data-gradient use), split
// to bf16 hi / lo and stored in MFMA fragment order for k_conv_gemm:
//     packed[((r / 32) * (Kp / 16) + k / 16) * 2 + part][lane = r % 32 + 32 * ((k % 16) / 8)][k % 8]
__global__ __launch_bounds__(256) void k_pack_weights(const float* __restrict__ w, int R, int C, long sr, long sc,
                                                      const int* __restrict__ tapidx, int ntaps, int Ccp, int Kp,
                                                      int rows, __bf16* __restrict__ packed)
{
    pack_weights_group8((long)blockIdx.x * 256 + threadIdx.x, w, R, C, sr, sc, tapidx, ntaps, Ccp, Kp, rows, packed);
}

// grad_w[r * sr + c * sc + tapidx[t]] (+)= dw[r, t * Ccp + c]  (the inverse map; every parameter element is hit by at
// most one (r, c, t) per call, so a plain read-modify-write is race-free; with accumulate == 0 it is a plain store --
// a tap list that covers the whole kernel window then defines every element of grad_w, no zero fill needed)
__global__ __launch_bounds__(256) void k_unpack_grad(const float* __restrict__ dw, int R, int C, long sr, long sc,
                                                     const int* __restrict__ tapidx, int ntaps, int Ccp,
                                                     float* __restrict__ grad_w, int accumulate)
{
    unpack_grad_group4((long)blockIdx.x * 256 + threadIdx.x, dw, R, C, sr, sc, tapidx, ntaps, Ccp, grad_w, accumulate);
}

__global__ __launch_bounds__(256) void k_unpack_grad_rows(const float* __restrict__ dw, int R, int C, long sr, long sc,
                                                          const int* __restrict__ tapidx, int ntaps, int Ccp,
                                                          float* __restrict__ grad_w, int accumulate)
{
    __shared__ float tile[UNPACK_CB * (UNPACK_MAX_SC + 1)];
    __shared__ int inv[UNPACK_MAX_SC];
    unpack_grad_rows(blockIdx.x, dw, R, C, sr, sc, tapidx, ntaps, Ccp, grad_w, accumulate, tile, inv);
}

// reductions end in one atomic per (block, channel): keep the block count near `target` in total
// positions per block so that the launch has about `target` blocks in total; a multiple of the positions one block
// iteration covers, and at least 4 iterations per block
static int ppb_for(long npos, int Cp, int images, int target)
{
    const int CH = Cp < 64 ? Cp : 64;
    const int per_iter = 1024 / CH;
    const int zchunks = Cp / CH;
    long slices = target / ((long)images * zchunks);
    if (slices < 1) slices = 1;
    long ppb = (npos + slices - 1) / slices;
    if (ppb < 4 * per_iter) ppb = 4 * per_iter;
    ppb = ((ppb + per_iter - 1) / per_iter) * per_iter;
    return (int)ppb;
}

static inline int zchunks(int Cp) { return Cp < 64 ? 1 : Cp / 64; }

}  // namespace sdn

using namespace sdn;

static int check_cp(const char* who, int Cp)
{
    // a block's 256 threads cover whole positions of a 64-channel chunk: Cp must be a power of two in [16, 1024]
    if (Cp < 16 || Cp > 1024 || (Cp & (Cp - 1)))
        return fail(SDN_EINVAL, "%s: padded channel count %d must be a power of two in [16, 1024]", who, Cp);
    return SDN_OK;
}

static int check_planes(const char* who, const void* planes, long plane_stride, long elems)
{
    if (planes && (plane_stride < elems || (plane_stride & 7)))
        return fail(SDN_EINVAL, "%s: plane stride %ld must be a multiple of 8 and >= %ld elements", who, plane_stride, elems);
    return SDN_OK;
}

SDN_API int sdn_in_apply(float* z, const double* stats, float* mr, const float* res, float* out2, int N, int HW, int C,
                         int Cp, float eps, int act, int res_relu, float momentum, float* running_mean,
                         float* running_var, void* planes, long plane_stride, int planes_relu, sdnStream stream)
{
    int rc = check_cp("sdn_in_apply", Cp);
    if (rc) return rc;
    if (!z || !stats || !mr || (out2 && !res)) return fail(SDN_EINVAL, "sdn_in_apply: null pointer");
    if ((rc = check_planes("sdn_in_apply", planes, plane_stride, (long)N * HW * Cp))) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_in_finalize, dim3(cdiv((long)N * Cp, 256)), dim3(256), 0, st, stats, N, C, Cp, HW, eps, momentum,
                       running_mean, running_var, (float2*)mr);
    if ((rc = check_launch("k_in_finalize"))) return rc;
    const int ppb = ppb_for(HW, Cp, N, 4096);
    hipLaunchKernelGGL(k_in_apply, dim3(cdiv(HW, ppb), N, zchunks(Cp)), dim3(256), 0, st, z, (const float2*)mr, res, out2,
                       HW, Cp, act, res_relu, ppb, (__bf16*)planes, plane_stride, planes_relu);
    return check_launch("k_in_apply");
}

SDN_API int sdn_in_bwd(float* g, const float* stored, const float* mr, double* sums, int N, int HW, int Cp, int mode,
                       void* planes, long plane_stride, sdnStream stream)
{
    int rc = check_cp("sdn_in_bwd", Cp);
    if (rc) return rc;
    if (!g || !stored || !mr || !sums) return fail(SDN_EINVAL, "sdn_in_bwd: null pointer");
    if ((rc = check_planes("sdn_in_bwd", planes, plane_stride, (long)N * HW * Cp))) return rc;
    hipStream_t st = (hipStream_t)stream;
    // SDN_IN_BWD_SUMS_ZEROED: the caller hands in zeroed sums (a launch list keeps them in its zero-initialised arena region: one
    // memset per pass instead of one 6 us fill launch per InstanceNorm layer -- 63 per GAN step)
    const bool zeroed = (mode & SDN_IN_BWD_SUMS_ZEROED) != 0;
    mode &= ~SDN_IN_BWD_SUMS_ZEROED;
    if (!zeroed && hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)N * Cp, st) != hipSuccess)
        return fail(SDN_ELAUNCH, "sdn_in_bwd: memset");
    const int rppb = ppb_for(HW, Cp, N, 1024);
    hipLaunchKernelGGL(k_in_bwd_reduce, dim3(cdiv(HW, rppb), N, zchunks(Cp)), dim3(256), 0, st, g, stored, sums, HW, Cp,
                       mode, rppb);
    if ((rc = check_launch("k_in_bwd_reduce"))) return rc;
    const int ppb = ppb_for(HW, Cp, N, 4096);
    hipLaunchKernelGGL(k_in_bwd_apply, dim3(cdiv(HW, ppb), N, zchunks(Cp)), dim3(256), 0, st, g, stored, sums,
                       (const float2*)mr, HW, Cp, mode, ppb, (__bf16*)planes, plane_stride);
    return check_launch("k_in_bwd_apply");
}

SDN_API int sdn_act_bwd(float* g, const float* y, float* bias_grad, long npos, int Cp, int act, void* planes,
                        long plane_stride, sdnStream stream)
{
    int rc = check_cp("sdn_act_bwd", Cp);
    if (rc) return rc;
    if (!g || (act && !y)) return fail(SDN_EINVAL, "sdn_act_bwd: null pointer");
    if ((rc = check_planes("sdn_act_bwd", planes, plane_stride, npos * Cp))) return rc;
    if (!act && !bias_grad && !planes) return SDN_OK;
    const int ppb = ppb_for(npos, Cp, 1, bias_grad ? 1024 : 4096);
    hipLaunchKernelGGL(k_act_bwd, dim3(cdiv(npos, ppb), 1, zchunks(Cp)), dim3(256), 0, (hipStream_t)stream, g, y,
                       bias_grad, npos, Cp, act, ppb, (__bf16*)planes, plane_stride);
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
                                  int Ccp, int Kp, int rows, void* packed, sdnStream stream)
{
    if (!w || !tapidx || !packed || Kp < ntaps * Ccp || (Kp & 31) || rows < R || (rows & 31) || Ccp < C)
        return fail(SDN_EINVAL, "sdn_conv_pack_weights: bad argument");
    // a thread packs eight consecutive columns that must share one row, tap and channel block, and stores them as 16 bytes
    // (conv_pack.h: pack_weights_group8): the padded channel count is a multiple of 8
    if (Ccp & 7) return fail(SDN_EINVAL, "sdn_conv_pack_weights: padded channel count %d must be a multiple of 8", Ccp);
    hipLaunchKernelGGL(k_pack_weights, dim3(cdiv((long)rows * Kp / 8, 256)), dim3(256), 0, (hipStream_t)stream, w, R, C, sr,
                       sc, tapidx, ntaps, Ccp, Kp, rows, (__bf16*)packed);
    return check_launch("k_pack_weights");
}

SDN_API int sdn_conv_unpack_grad(const float* dw, int R, int C, long sr, long sc, const int32_t* tapidx, int ntaps,
                                 int Ccp, float* grad_w, int accumulate, sdnStream stream)
{
    if (!dw || !tapidx || !grad_w || Ccp < C) return fail(SDN_EINVAL, "sdn_conv_unpack_grad: bad argument");
    // unpack_grad_group4 reads four consecutive columns of one (row, tap) with a 16-byte load
    if ((Ccp & 3) || ((uintptr_t)dw & 15))
        return fail(SDN_EINVAL, "sdn_conv_unpack_grad: padded channel count %d must be a multiple of 4 and dw 16-byte aligned", Ccp);
    if (unpack_rows_ok(sc, ntaps) && sc < sr) {   // taps innermost: the LDS transpose (conv_pack.h)
        const long nb = unpack_rows_blocks(R, C);
        if (nb > 0x7fffffffL) return fail(SDN_EINVAL, "sdn_conv_unpack_grad: too many rows");
        hipLaunchKernelGGL(k_unpack_grad_rows, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dw, R, C, sr, sc, tapidx, ntaps,
                           Ccp, grad_w, accumulate);
        return check_launch("k_unpack_grad_rows");
    }
    hipLaunchKernelGGL(k_unpack_grad, dim3(cdiv((long)R * ntaps * Ccp / 4, 256)), dim3(256), 0, (hipStream_t)stream, dw, R,
                       C, sr, sc, tapidx, ntaps, Ccp, grad_w, accumulate);
    return check_launch("k_unpack_grad");
}
