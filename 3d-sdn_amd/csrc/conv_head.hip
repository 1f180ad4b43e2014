// The 7 x 7 head convolutions on the matrix cores (r05; VERDICT r04 #2c).
//
// Reference: the last layers of GlobalGenerator and Encoder, nn.ReflectionPad2d(3) + nn.Conv2d(ngf, output_nc, 7) + nn.Tanh
// (/root/reference/textural/models/networks.py:236, 306), 64 -> 3 and 16 -> 5 channels at the full 384 x 1248 resolution, and the
// stem's data gradient towards the encoder features (5 of its 48 input channels: the same shape, 64 -> 5).  cuDNN runs them as
// ordinary convolutions; r01-r04 ran them as exact-fp32 vector kernels (conv_narrow.hip: 18-28 % of the fp32 vector peak, 2.2 ms of
// a 65 ms GAN step) because a 3-channel output wastes 29 of the 32 columns of the implicit-GEMM tiles.
//
// Here: v_mfma_f32_16x16x32_bf16 with the OUTPUT CHANNELS as the 16 matrix rows (3 or 5 of 16 used) and 16 output positions as
// the columns, bf16 x 3 split products like every other conv of the library (lo*hi, hi*lo, hi*hi: fp32-class results).
//   * a workgroup (4 waves) owns 8 x 32 output positions and stages their (8 + K - 1) x (32 + K - 1) input patch in LDS (32 channels
//     at a time: two passes for 64 input channels, so that two workgroups fit a CU and overlap load and MFMA phases),
//     already split into bf16 hi / lo and ReLU'd (k_conv_narrow_fwd's idea: 49 x less gather than one tile per tap), laid out
//     [part][8-channel group][pixel][8]: the 16 lanes of an MFMA k-group read 16 consecutive pixels = 16 consecutive 16-byte
//     slots, and a channel-group plane is a multiple of 256 bytes, so every ds_read_b128 lane group hits 16 different slots;
//   * K runs over 8-channel slots u = tap * (Cin / 8) + channel group, four slots (32 k) per MFMA step -- two steps per tap for 64
//     input channels, two taps per step for 16 -- so one kernel body serves both widths; slots behind the last tap carry zero
//     weights;
//   * the weights arrive pre-split in fragment order (lane = k-group * 16 + output channel: one coalesced 1 KiB load per part and
//     step, the next step's pair in flight while this step's MFMAs issue); the host builds that buffer from the dense tap window
//     the narrow kernels use (sdn_hip/conv.py: Stage.head_mfma);
//   * per step a wave issues 12 MFMAs (4 position tiles x 3 products) for 8 LDS fragment reads and 2 weight loads.
// Epilogue: bias, tanh, 16-byte stores of four output channels per lane (channels behind the real ones come out as zeros:
// their weights and bias are zero).
#include "conv_common.h"
#include "sdn_common.h"

namespace sdn {

constexpr int HD_TH = 8, HD_TW = 32;
typedef __attribute__((ext_vector_type(4))) __bf16 head_bf16x4;

struct HeadParams {
    const float* in;    // [N, IH, IW, Cip] fp32, Cip = 8 * CG
    float* out;         // [N, QH, QW, 16 RG]
    const __bf16* w;    // [RG][S][2 (hi, lo)][64 lanes][8]
    const float* bias;  // [16 RG] or null
    double* stats;      // optional [N, STAT_SLOTS, 16 RG, 2]: InstanceNorm statistics of (acc + bias), as the implicit-GEMM kernels
    int N, IH, IW, QH, QW, dy_min, dx_min, pad_mode, in_relu, act, tiles_x, tiles_y;
};

// RG (r06): row groups of 16 output channels -- 1: the head layers (<= 16 output channels); 4: a 64-channel output, the data
// gradient of the generator head towards its 64 input channels (networks.py:236 backwards: dz has 3 of 16 padded channels).  The
// LDS fragments of a step are read once and multiplied with every group's weights.
template <int CG, int K, int RG>
__global__ __launch_bounds__(256) void k_conv_head_mfma(const HeadParams P)
{
    // 64 input channels are taken in two PASSES of 32 (4 channel groups): the patch of a pass is 68 KB, so TWO workgroups share a
    // CU and one's patch load hides behind the other's MFMA loop (first cut, r05i: one 136 KB patch, one workgroup per CU, 34
    // dependent-latency loads per thread in front of every K loop: 1.65 ms against the vector kernel's 0.93).  Slot order makes
    // this free: step s = 2 tap + pass already holds exactly the four channel groups of that pass.
    constexpr int CGP = CG < 4 ? CG : 4, NPASS = CG / CGP;
    constexpr int PH = HD_TH + K - 1, PW = HD_TW + K - 1, NPX = PH * PW, PLANE = (NPX + 15) / 16 * 16;
    constexpr int NT = K * K, U = NT * CG, S = (U + 3) / 4, CIP = 8 * CG, Q4 = 2 * CGP;
    constexpr int ITEMS = NPX * Q4, PER = (ITEMS + 255) / 256, BATCH = 9;
    __shared__ __attribute__((aligned(16))) __bf16 patch[2][CGP][PLANE][8];
    static_assert(sizeof(__bf16) * 2 * CGP * PLANE * 8 <= 80 * 1024, "two workgroups' patches must fit in a CU's LDS");
    static_assert(CG % CGP == 0 && (NPASS == 1 || S % NPASS == 0), "slot steps must split evenly over the passes");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int bx = bid % P.tiles_x;
    bid /= P.tiles_x;
    const int by = bid % P.tiles_y, n = bid / P.tiles_y;
    const int qy0 = by * HD_TH, qx0 = bx * HD_TW;
    const float* img = P.in + (size_t)n * P.IH * P.IW * CIP;

    // Lane: k-group g = lane >> 4 (8 channels of slot 4 s + g), column pl = lane & 15 (a position for the activation operand, an
    // output channel for the weight operand).  Wave w: output rows 2 w, 2 w + 1, two 16-column tiles each.
    const int g = lane >> 4, pl = lane & 15;
    f32x4 acc[RG][4];
#pragma unroll
    for (int rg = 0; rg < RG; rg++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[rg][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8* wf = reinterpret_cast<const bf16x8*>(P.w) + lane;

    for (int pass = 0; pass < NPASS; pass++) {
        // ---- this pass's channels of the input patch, split and (optionally) ReLU'd on their way into LDS; BATCH (9) loads in flight per thread
        if (pass) __syncthreads();   // every wave is done with the previous pass's patch
        for (int i0 = 0; i0 < PER; i0 += BATCH) {
            f32x4 v[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; j++) {
                const int it = tid + (i0 + j) * 256;
                v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (i0 + j < PER && it < ITEMS) {
                    const int p = it / Q4, q = it - p * Q4;
                    const int py = p / PW, px = p - py * PW;
                    int iy = qy0 + py + P.dy_min, ix = qx0 + px + P.dx_min;
                    if (P.pad_mode) {   // ReflectionPad2d folded into the gather
                        iy = iy < 0 ? -iy : (iy >= P.IH ? 2 * P.IH - 2 - iy : iy);
                        ix = ix < 0 ? -ix : (ix >= P.IW ? 2 * P.IW - 2 - ix : ix);
                    }
                    if (iy >= 0 && iy < P.IH && ix >= 0 && ix < P.IW)
                        v[j] = *reinterpret_cast<const f32x4*>(img + ((size_t)iy * P.IW + ix) * CIP + 32 * pass + 4 * q);
                }
            }
#pragma unroll
            for (int j = 0; j < BATCH; j++) {
                const int it = tid + (i0 + j) * 256;
                if (i0 + j < PER && it < ITEMS) {
                    const int p = it / Q4, q = it - p * Q4;
                    f32x4 x = v[j];
                    if (P.in_relu) {
#pragma unroll
                        for (int e = 0; e < 4; e++) x[e] = fmaxf(x[e], 0.f);
                    }
                    const SplitBf16 a = split2(x[0], x[1]), b = split2(x[2], x[3]);
                    *reinterpret_cast<head_bf16x4*>(&patch[0][q >> 1][p][(q & 1) * 4]) = head_bf16x4{a.hi[0], a.hi[1], b.hi[0], b.hi[1]};
                    *reinterpret_cast<head_bf16x4*>(&patch[1][q >> 1][p][(q & 1) * 4]) = head_bf16x4{a.lo[0], a.lo[1], b.lo[0], b.lo[1]};
                }
            }
        }
        __syncthreads();

        // ---- K loop over this pass's slot steps s = pass, pass + NPASS, ...
        // the LDS fragments of slot step `s` for this wave's four position tiles
        auto fragments = [&](const int s, bf16x8 (&ah)[4], bf16x8 (&al)[4]) {
            const int u = 4 * s + g;
            int tap = u / CG;
            const int cgl = u - tap * CG - pass * CGP;   // channel group inside this pass's patch
            if (tap >= NT) tap = 0;           // (a slot behind the last tap: its weights are zero, any valid pixel will do)
            const int ky = tap / K, kx = tap - ky * K;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int p = (2 * wave + (t >> 1) + ky) * PW + (t & 1) * 16 + kx + pl;
                ah[t] = *reinterpret_cast<const bf16x8*>(&patch[0][cgl][p][0]);
                al[t] = *reinterpret_cast<const bf16x8*>(&patch[1][cgl][p][0]);
            }
        };
        // Software pipeline.  The weight fragments come from L2 (every workgroup reads the same 100-200 KB): they are requested TWO
        // steps (~800 cycles of MFMA issue) ahead -- one step ahead (second cut, r05j) every step still waited ~200 cycles for
        // them; the LDS fragments of the next step are in flight while this step's twelve MFMAs issue.
        auto mma = [&](const int rg, const bf16x8& wh, const bf16x8& wl, const bf16x8 (&ah)[4], const bf16x8 (&al)[4]) {
            // small terms first; the four tiles' accumulators alternate so that no MFMA waits for its predecessor
#pragma unroll
            for (int t = 0; t < 4; t++) acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[t], acc[rg][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; t++) acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[t], acc[rg][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; t++) acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[t], acc[rg][t], 0, 0, 0);
        };
        const int nst = (S - pass + NPASS - 1) / NPASS;                               // slot steps of this pass
        auto sidx = [&](const int k) { return pass + NPASS * (k < nst ? k : nst - 1); };   // (clamped: loads past the end re-read the last)
        bf16x8 fah[4], fal[4], fbh[4], fbl[4];
        if constexpr (RG == 1) {
            bf16x8 wAh = wf[(size_t)sidx(0) * 128], wAl = wf[(size_t)sidx(0) * 128 + 64];
            bf16x8 wBh = wf[(size_t)sidx(1) * 128], wBl = wf[(size_t)sidx(1) * 128 + 64];
            fragments(sidx(0), fah, fal);
            for (int k = 0; k < nst; k += 2) {
                const bf16x8 wCh = wf[(size_t)sidx(k + 2) * 128], wCl = wf[(size_t)sidx(k + 2) * 128 + 64];
                const bf16x8 wDh = wf[(size_t)sidx(k + 3) * 128], wDl = wf[(size_t)sidx(k + 3) * 128 + 64];
                fragments(sidx(k + 1), fbh, fbl);
                mma(0, wAh, wAl, fah, fal);
                fragments(sidx(k + 2), fah, fal);
                if (k + 1 < nst) mma(0, wBh, wBl, fbh, fbl);
                wAh = wCh; wAl = wCl;
                wBh = wDh; wBl = wDl;
            }
        } else {
            // several row groups: a step is RG x 12 MFMAs, so the weights of the NEXT step (one step ahead) are early enough
            bf16x8 wA[RG][2], wB[RG][2];
            auto loadw = [&](bf16x8 (&wv)[RG][2], const int st) {
#pragma unroll
                for (int rg = 0; rg < RG; rg++) {
                    wv[rg][0] = wf[((size_t)rg * S + st) * 128];
                    wv[rg][1] = wf[((size_t)rg * S + st) * 128 + 64];
                }
            };
            loadw(wA, sidx(0));
            fragments(sidx(0), fah, fal);
            for (int k = 0; k < nst; k += 2) {
                loadw(wB, sidx(k + 1));
                fragments(sidx(k + 1), fbh, fbl);
#pragma unroll
                for (int rg = 0; rg < RG; rg++) mma(rg, wA[rg][0], wA[rg][1], fah, fal);
                loadw(wA, sidx(k + 2));
                fragments(sidx(k + 2), fah, fal);
                if (k + 1 < nst) {
#pragma unroll
                    for (int rg = 0; rg < RG; rg++) mma(rg, wB[rg][0], wB[rg][1], fbh, fbl);
                }
            }
        }
    }

    // ---- epilogue.  D: column = lane & 15 = position, row = 4 (lane >> 4) + reg = output channel (of row group rg)
    constexpr int COP = 16 * RG;
    __shared__ float red[4][COP][2];
#pragma unroll
    for (int rg = 0; rg < RG; rg++) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (P.bias) bv = *reinterpret_cast<const f32x4*>(P.bias + 16 * rg + 4 * g);
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int qy = qy0 + 2 * wave + (t >> 1), qx = qx0 + (t & 1) * 16 + pl;
            if (qy >= P.QH || qx >= P.QW) continue;
            f32x4 o = acc[rg][t] + bv;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                s1[e] += o[e];
                s2[e] += o[e] * o[e];
            }
            if (P.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = o[e] > 0.f ? o[e] : 0.2f * o[e];
            } else if (P.act == 2) {
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = tanhf(o[e]);
            }
            *reinterpret_cast<f32x4*>(P.out + (((size_t)n * P.QH + qy) * P.QW + qx) * COP + 16 * rg + 4 * g) = o;
        }
        if (P.stats) {   // (workgroup-uniform)
            // the 16 positions of a k-group's lanes meet by cross-lane adds, the four waves in LDS, one fp64 atomic per
            // (workgroup, channel, moment) into the slot of this block -- the statistics contract of k_conv_gemm / k_conv_tile
#pragma unroll
            for (int e = 0; e < 4; e++) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s1[e] += __shfl_xor(s1[e], o, 64);
                    s2[e] += __shfl_xor(s2[e], o, 64);
                }
                if (pl == 0) {
                    red[wave][16 * rg + 4 * g + e][0] = s1[e];
                    red[wave][16 * rg + 4 * g + e][1] = s2[e];
                }
            }
        }
    }
    if (P.stats) {
        __syncthreads();
        if (tid < 2 * COP) {
            const int c = tid >> 1, k = tid & 1;
            const float v = (red[0][c][k] + red[1][c][k]) + (red[2][c][k] + red[3][c][k]);
            const int slot = (by * P.tiles_x + bx) & (STAT_SLOTS - 1);
            unsafeAtomicAdd(P.stats + (((size_t)n * STAT_SLOTS + slot) * COP + c) * 2 + k, (double)v);
        }
    }
}

template <int CG, int K, int RG>
static int launch_head(const HeadParams& P, hipStream_t st)
{
    hipLaunchKernelGGL((k_conv_head_mfma<CG, K, RG>), dim3((unsigned)(P.tiles_x * P.tiles_y * P.N)), dim3(256), 0, st, P);
    return check_launch("k_conv_head_mfma");
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_conv_head_steps(int Cip, int KH, int KW, int* steps)
{
    if (!steps || Cip < 8 || (Cip & 7) || KH < 1 || KW < 1) return fail(SDN_EINVAL, "sdn_conv_head_steps: bad argument");
    *steps = (KH * KW * (Cip / 8) + 3) / 4;
    return SDN_OK;
}

SDN_API int sdn_conv_head_mfma(const float* in, int N, int IH, int IW, int Cip, float* out, int QH, int QW, int Cop,
                               int rows_used, const void* w_frag, int KH, int KW, int dy_min, int dx_min, int pad_mode,
                               int in_relu, const float* bias, int act, double* stats, sdnStream stream)
{
    if (!in || !out || !w_frag) return fail(SDN_EINVAL, "sdn_conv_head_mfma: null pointer");
    if (Cop != 16 && !(Cop == 64 && Cip == 16))
        return fail(SDN_EINVAL, "sdn_conv_head_mfma: the output tensor must have 16 (padded) channels, or 64 over a 16-channel input; got %d over %d", Cop, Cip);
    if (rows_used < 1 || rows_used > Cop) return fail(SDN_EINVAL, "sdn_conv_head_mfma: rows_used %d not in 1..%d", rows_used, Cop);
    if (KH != 7 || KW != 7 || (Cip != 16 && Cip != 64))
        return fail(SDN_EINVAL, "sdn_conv_head_mfma: built for 7 x 7 windows over 16 or 64 input channels (got %d x %d over %d)", KH, KW, Cip);
    if (N < 1 || QH < 1 || QW < 1 || IH < 1 || IW < 1) return fail(SDN_EINVAL, "sdn_conv_head_mfma: bad geometry");
    if (pad_mode && (IH < KH || IW < KW)) return fail(SDN_EINVAL, "sdn_conv_head_mfma: image smaller than the reflected border");
    HeadParams P;
    P.in = in; P.out = out; P.w = (const __bf16*)w_frag; P.bias = bias; P.stats = stats;
    P.N = N; P.IH = IH; P.IW = IW; P.QH = QH; P.QW = QW; P.dy_min = dy_min; P.dx_min = dx_min;
    P.pad_mode = pad_mode; P.in_relu = in_relu; P.act = act;
    P.tiles_x = (QW + HD_TW - 1) / HD_TW;
    P.tiles_y = (QH + HD_TH - 1) / HD_TH;
    if ((long)P.tiles_x * P.tiles_y * N > 0x7fffffffL) return fail(SDN_EINVAL, "sdn_conv_head_mfma: grid too large");
    hipStream_t st = (hipStream_t)stream;
    // algorithmic work (the real output channels; 16 rows and 3 products per algorithmic one are issued): the head kernels' slot
    TimedLaunch timed(TIME_CONV_NARROW, st, 2.0 * (double)N * QH * QW * KH * KW * Cip * rows_used);
    if (Cip == 64) return launch_head<8, 7, 1>(P, st);
    if (Cop == 64) return launch_head<2, 7, 4>(P, st);
    return launch_head<2, 7, 1>(P, st);
}
