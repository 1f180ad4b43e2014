// k_conv_halo: stride-1 convolutions with the INPUT PATCH staged in LDS (r04) -- the 3x3 residual-block layers and the 4x4
// stride-1 discriminator layers, forward and data gradient.
//
// Reference: Conv2d(k = 3 | 4, stride 1) of ResnetBlock / NLayerDiscriminator, /root/reference/textural/models/networks.py:244-283,
// 431-442 (cuDNN in the reference), and their data gradients (stride-1 convolutions with the flipped window).
//
// k_conv_tile copies a fresh 256 x 32-channel activation tile per tap: for a 3x3 window every input row travels L2 -> LDS
// nine times per channel block, 32 of the 48 KB a K step moves, and the counters say the kernel is bound by exactly that --
// what one CU can keep in flight towards its L2 (profiles/r04a_*: MFMA pipe 62 % busy, LDS 21 %, L2 hit rate 95 %).  Here a
// workgroup owns a TH x TW block of output positions (TH * TW <= 256) and keeps, per 32-channel block, the (TH + kh - 1) x
// (TW + kw - 1) input PATCH in LDS: copied once (<= 416 rows x 64 B x (hi, lo): 53 KB instead of 9 x 32 KB), double
// buffered over the channel blocks.  The taps of a channel block then only move the fragment-read addresses: patch row of
// output (ty, tx) under tap (ky, kx) = (ty + ky) * PW + (tx + kx) -- one scalar delta per tap.  Weights stream through three
// 16 KB stages as in k_conv_tile.  Per K step a workgroup now copies 16 KB of weights + 5.9 KB of patch (averaged over 9
// taps) instead of 48 KB.  Everything else -- bf16 (hi, lo) operand planes, LDS-DMA with the XOR swizzle on the source
// address, three MFMA products in fp32, mid-step counted wait + barrier, copies between MFMA groups, epilogue with bias /
// activation / InstanceNorm statistics -- is k_conv_tile's.
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"
#include "conv_dma.h"
#include "sdn_common.h"

namespace sdn {

static __device__ __attribute__((aligned(256))) unsigned g_zero_page_h[64];

struct HaloTaps {
    int n;
    signed char ky[CONV_MAX_TAPS];   // tap row / column inside the window, 0 .. kh-1 / kw-1, in the launch's tap order
    signed char kx[CONV_MAX_TAPS];
};

struct ConvHaloParams {
    const __bf16* in;        // planes [2][N, IH, IW, Cip]
    long plane_stride;
    float* out;              // [N, OH, OW, Cop]
    const __bf16* w;         // [w_rows][nsteps][2][32], step = cb * ntaps + t  (sdn_conv_pack_weights_kmajor)
    const float* bias;
    double* stats;
    int N, IH, IW, Cip, OH, OW, Cop;
    int TH, TW, tiles_y, tiles_x;   // output block of a workgroup, blocks per image
    int PW, PR;                      // patch width, patch rows (PH * PW)
    int dy_min, dx_min;              // input offset of the window's first tap
    int nsteps, w_rows, pad_mode, act, accumulate, ntiles;
    HaloTaps taps;
};

constexpr int HALO_PRMAX = 416;   // patch rows a buffer holds: 18 x 18 (3x3 window, 16 x 16 block), 19 x 19 (4x4), 14 x 22, ...

template <int V>
__global__ __launch_bounds__(512, 2) void k_conv_halo(const ConvHaloParams P)
{
    constexpr int TM = 2, TN = 2;
    constexpr int BM = 256, BN = 128;
    constexpr int A_PLANE = HALO_PRMAX * 64, PATCH = 2 * A_PLANE;   // bytes
    constexpr int B_PLANE = BN * 64, BSTAGE = 2 * B_PLANE;
    constexpr int B_BASE = 2 * PATCH;
    __shared__ __attribute__((aligned(1024))) char smem[2 * PATCH + 3 * BSTAGE];   // 155 648 B

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: an XCD's blocks are neighbouring position blocks with all channel tiles of each
    const int ntiles = P.ntiles;
    const unsigned nblk = gridDim.x;
    const unsigned xcd = blockIdx.x & 7u, jb = blockIdx.x >> 3;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7u;
    const unsigned v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + jb;
    const int mt_global = (int)(v / (unsigned)ntiles);
    const int per_img = P.tiles_y * P.tiles_x;
    const int n = mt_global / per_img;
    const int tyx = mt_global - n * per_img;
    const int ty0 = (tyx / P.tiles_x) * P.TH, tx0 = (tyx % P.tiles_x) * P.TW;
    const int n0 = (int)(v % (unsigned)ntiles) * BN;
    const int py0 = ty0 + P.dy_min, px0 = tx0 + P.dx_min;   // input coordinates of patch row 0

    // ---- patch copies: the patch is NG = ceil(PR / 16) groups of 16 rows x 64 B per plane; wave w copies groups w, w + 8,
    // w + 16, w + 24 (hi and lo), one group per K step during the first four taps of the PREVIOUS channel block.  Lane l ->
    // row 16 g + (l >> 2), physical 16-B chunk (l & 3); logical chunk = physical ^ ((row >> 2) & 3) = (l & 3) ^ ((l >> 4) & 3).
    const int lchunk = (lane & 3) ^ ((lane >> 4) & 3);
    const int NG = (P.PR + 15) >> 4;
    const int ih2 = 2 * P.IH - 2, iw2 = 2 * P.IW - 2;
    const bool reflect = P.pad_mode != 0;
    unsigned pixoff[4];
    bool pixok[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = 16 * (wave + 8 * k) + (lane >> 2);
        const int pyj = j / P.PW, pxj = j - pyj * P.PW;
        int iy = py0 + pyj, ix = px0 + pxj;
        int ry = max(iy, -iy), rx = max(ix, -ix);
        ry = min(ry, ih2 - ry);
        rx = min(rx, iw2 - rx);
        iy = reflect ? ry : iy;
        ix = reflect ? rx : ix;
        pixok[k] = ((int)(j < P.PR) & (int)((unsigned)iy < (unsigned)P.IH) & (int)((unsigned)ix < (unsigned)P.IW)) != 0;
        pixoff[k] = (unsigned)((iy * P.IW + ix) * P.Cip + lchunk * 8) * 2u;
    }
    const char* in_n = (const char*)(P.in + (size_t)n * P.IH * P.IW * P.Cip);
    const long lo_bytes = P.plane_stride * 2;
    const char* zero = (const char*)g_zero_page_h;
    // weights: row n0 + (tid >> 2), 128 B per (row, step)
    const int srow = tid >> 2;
    const int wchunk = (lane & 3) ^ ((srow >> 2) & 3);
    const char* wsrc = (const char*)P.w + ((size_t)(n0 + srow) * P.nsteps) * 128 + wchunk * 16;

    // tap table in the lanes of one VGPR: (ky, kx) of tap t
    int tapv = 0;
    if (lane < P.taps.n) tapv = ((int)P.taps.ky[lane] << 8) | (int)P.taps.kx[lane];
    const int ntaps = P.taps.n;
    const int ncb = P.Cip >> 5;
    const int nsteps = P.nsteps;

    // copy group k (0 .. 3) of the patch of channel block `cb` into patch buffer (cb & 1); groups behind the patch are skipped
    // by the CALLER (wave-uniform)
    auto patch_copy = [&](int k, int cb) __attribute__((always_inline)) {
        const unsigned po = k == 0 ? pixoff[0] : (k == 1 ? pixoff[1] : (k == 2 ? pixoff[2] : pixoff[3]));
        const bool ok = k == 0 ? pixok[0] : (k == 1 ? pixok[1] : (k == 2 ? pixok[2] : pixok[3]));
        const char* a = in_n + po + (unsigned)cb * 64u;
        const char* hi = select_ptr(ok, a, zero);
        const char* lo = select_ptr(ok, a + lo_bytes, zero);
        char* d = smem + (cb & 1) * PATCH + (wave + 8 * k) * 1024;
        glds16(hi, lds_addr(d));
        glds16(lo, lds_addr(d + A_PLANE));
    };
    auto weight_copy = [&](auto plane_c, int stage_b, int bs) __attribute__((always_inline)) {
        constexpr int pl = decltype(plane_c)::value;
        const char* wp = wsrc + (size_t)bs * 128 + pl * 64;
        char* d = smem + B_BASE + stage_b + pl * B_PLANE + 16 * wave * 64;
        glds16(wp, lds_addr(d));
    };

    // ---- fragment geometry: MFMA row m of this lane -> output (ty, tx) of the block -> patch row of tap (0, 0)
    const int wm0 = (wave >> 1) * TM * 32, wn0 = (wave & 1) * TN * 32;
    const int fr = lane & 31, fkh = lane >> 5;
    int j0[TM];
#pragma unroll
    for (int mt = 0; mt < TM; mt++) {
        const int m = wm0 + mt * 32 + fr;
        const bool ok = m < P.TH * P.TW;
        const int tyy = ok ? m / P.TW : 0, txx = ok ? m - tyy * P.TW : 0;
        j0[mt] = tyy * P.PW + txx;
    }
    int boffb[TN][2];
#pragma unroll
    for (int nt = 0; nt < TN; nt++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int row = wn0 + nt * 32 + fr;
            boffb[nt][ks] = row * 64 + (((2 * ks + fkh) ^ ((row >> 2) & 3)) << 4);
        }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mt = 0; mt < TM; mt++)
#pragma unroll
        for (int nt = 0; nt < TN; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;

    bf16x8 af[2][2][TM], bf[2][2][TN];   // [k half][hi, lo][tile]
    // fragments of k half `ks` of the step with (patch buffer byte offset pb, tap delta in patch rows, weight stage bstage)
#define HALO_READ(ks, pb, delta, bstage)                                                                               \
    {                                                                                                                  \
        _Pragma("unroll") for (int mt = 0; mt < TM; mt++)                                                              \
        {                                                                                                              \
            const int j = j0[mt] + (delta);                                                                            \
            const int off = (j << 6) + (((2 * (ks) + fkh) ^ ((j >> 2) & 3)) << 4);                                     \
            af[ks][0][mt] = *reinterpret_cast<const bf16x8*>(smem + (pb) + off);                                       \
            af[ks][1][mt] = *reinterpret_cast<const bf16x8*>(smem + (pb) + A_PLANE + off);                             \
        }                                                                                                              \
        _Pragma("unroll") for (int nt = 0; nt < TN; nt++)                                                              \
        {                                                                                                              \
            bf[ks][0][nt] = *reinterpret_cast<const bf16x8*>(smem + B_BASE + (bstage) + boffb[nt][ks]);                \
            bf[ks][1][nt] = *reinterpret_cast<const bf16x8*>(smem + B_BASE + (bstage) + B_PLANE + boffb[nt][ks]);      \
        }                                                                                                              \
    }
#define HALO_MFMA(ks, pp, mt, nt)                                                                                      \
    if constexpr (V != 2)                                                                                              \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][(pp) == 0 ? 1 : 0][mt], bf[ks][(pp) == 1 ? 1 : 0][nt], \
                                                              acc[mt][nt], 0, 0, 0);                                   \
    else                                                                                                               \
        asm volatile("" : "+v"(acc[mt][nt]) : "v"(af[ks][(pp) == 0 ? 1 : 0][mt]), "v"(bf[ks][(pp) == 1 ? 1 : 0][nt]));
#define HALO_GROUP(ks, pp)                                                                                             \
    _Pragma("unroll") for (int mt = 0; mt < TM; mt++) _Pragma("unroll") for (int nt = 0; nt < TN; nt++)               \
        HALO_MFMA(ks, pp, mt, nt)

    // ---- prologue: the patch of channel block 0, weight stages 0 .. 2
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (wave + 8 * k < NG) patch_copy(k, 0);
    weight_copy(std::integral_constant<int, 0>{}, 0, 0);
    weight_copy(std::integral_constant<int, 1>{}, 0, 0);
    if (nsteps > 1) {
        weight_copy(std::integral_constant<int, 0>{}, BSTAGE, 1);
        weight_copy(std::integral_constant<int, 1>{}, BSTAGE, 1);
    }
    if (nsteps > 2) {
        weight_copy(std::integral_constant<int, 0>{}, 2 * BSTAGE, 2);
        weight_copy(std::integral_constant<int, 1>{}, 2 * BSTAGE, 2);
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // K loop: step s = (cb, t).  Scalars: t, cb of the current step; t1, cb1 of the next.
    int t = 0, cb = 0;
    int bcur = 0, bnx1 = BSTAGE;     // weight stage of step s / s + 1 (byte offsets)
    bool prevA = false;              // the previous step issued patch copies behind its MFMAs
    {
        const int tp = __builtin_amdgcn_readlane(tapv, 0);
        HALO_READ(0, 0, (tp >> 8) * P.PW + (tp & 0xff), 0);
    }
    // MAIN: a step with at least three successors (copies are issued, counted waits); else a tail step (no copies, full waits)
#define HALO_STEP(MAIN)                                                                                                \
    {                                                                                                                  \
        int t1 = t + 1, cb1 = cb;                                                                                      \
        if (t1 >= ntaps) {                                                                                             \
            t1 = 0;                                                                                                    \
            cb1 = cb + 1;                                                                                              \
        }                                                                                                              \
        const int tp = __builtin_amdgcn_readlane(tapv, t);                                                             \
        const int delta = (tp >> 8) * P.PW + (tp & 0xff);                                                              \
        const int pb = (cb & 1) * PATCH;                                                                               \
        HALO_READ(1, pb, delta, bcur);                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        HALO_GROUP(0, 0);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        HALO_GROUP(0, 1);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        HALO_GROUP(0, 2);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if ((MAIN) || s + 1 < nsteps) {                                                                                \
            /* the weights of step s + 1 (and, at a channel-block boundary, the next patch) have landed for everybody, and */ \
            /* everybody has issued its last reads of stage s: counted wait -- the copies issued behind the previous step's */ \
            /* MFMAs (2 weight copies, + 2 patch copies) may still fly */                                              \
            if (!(MAIN))                                                                                               \
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                                          \
            else if (prevA)                                                                                            \
                asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");                                          \
            else                                                                                                       \
                asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");                                          \
            const int tp1 = __builtin_amdgcn_readlane(tapv, t1);                                                       \
            HALO_READ(0, (cb1 & 1) * PATCH, (tp1 >> 8) * P.PW + (tp1 & 0xff), bnx1);                                   \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        HALO_GROUP(1, 0);                                                                                              \
        if ((MAIN) && V != 1) weight_copy(std::integral_constant<int, 0>{}, bcur, s + 3);                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        HALO_GROUP(1, 1);                                                                                              \
        if ((MAIN) && V != 1) weight_copy(std::integral_constant<int, 1>{}, bcur, s + 3);                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        HALO_GROUP(1, 2);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        /* the next channel block's patch: one 16-row group (hi, lo) per step during the first four taps */            \
        prevA = (MAIN) && V != 1 && t < 4 && cb + 1 < ncb && wave + 8 * t < NG;                                        \
        if (prevA) patch_copy(t, cb + 1);                                                                              \
        t = t1;                                                                                                        \
        cb = cb1;                                                                                                      \
        bcur = bnx1;                                                                                                   \
        bnx1 = bnx1 + BSTAGE == 3 * BSTAGE ? 0 : bnx1 + BSTAGE;                                                        \
    }
    int s = 0;
    for (; s + 3 < nsteps; s++) HALO_STEP(true);
    for (; s < nsteps; s++) HALO_STEP(false);
#undef HALO_STEP
#undef HALO_READ
#undef HALO_MFMA
#undef HALO_GROUP
    __syncthreads();   // every wave is done with the LDS: the epilogue reuses it

    // ---- epilogue (k_conv_tile's, with the block's 2-D row -> pixel map)
    int* s_outpix = reinterpret_cast<int*>(smem);
    float* red = reinterpret_cast<float*>(smem + 4096);
    if (tid < BM) {
        int o = -1;
        if (tid < P.TH * P.TW) {
            const int tyy = tid / P.TW, txx = tid - tyy * P.TW;
            const int oy = ty0 + tyy, ox = tx0 + txx;
            if (oy < P.OH && ox < P.OW) o = (n * P.OH + oy) * P.OW + ox;
        }
        s_outpix[tid] = o;
    }
    __syncthreads();
    const int col = lane & 31;
#pragma unroll
    for (int nt = 0; nt < TN; nt++) {
        const int co = n0 + wn0 + nt * 32 + col;
        const bool co_ok = co < P.Cop;
        const float bias = (co_ok && P.bias) ? P.bias[co] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < TM; mt++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = wm0 + mt * 32 + mfma_row(r, lane);
                const int o = s_outpix[row];
                if (o < 0 || !co_ok) continue;
                float x = acc[mt][nt][r] + bias;
                s1 += x;
                s2 += x * x;
                if (P.act == 1)
                    x = x > 0.f ? x : 0.2f * x;
                else if (P.act == 2)
                    x = tanhf(x);
                float* dst = P.out + (size_t)o * P.Cop + co;
                if (P.accumulate) x += *dst;
                *dst = x;
            }
        }
        if (P.stats) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                const int slot = ((wave >> 1) * BN + wn0 + nt * 32 + col) * 2;
                red[slot] = s1;
                red[slot + 1] = s2;
            }
        }
    }
    if (P.stats) {
        __syncthreads();
        if (tid < BN) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                s1 += red[(w * BN + tid) * 2];
                s2 += red[(w * BN + tid) * 2 + 1];
            }
            const int co = n0 + tid;
            if (co < P.Cop) {
                const int slot = tyx & (STAT_SLOTS - 1);
                double* st = P.stats + (((size_t)n * STAT_SLOTS + slot) * P.Cop + co) * 2;
                unsafeAtomicAdd(st, (double)s1);
                unsafeAtomicAdd(st + 1, (double)s2);
            }
        }
    }
}

// output block (TH x TW <= 256) for an OH x OW image and a kh x kw window: the candidate that wastes the fewest MFMA rows and
// whose patch fits a buffer; 0 when none does
static bool halo_block(int OH, int OW, int kh, int kw, int* TH, int* TW)
{
    double best = 0.0;
    *TH = *TW = 0;
    for (int th = 4; th <= 64; th++) {
        for (int tw = 4; tw <= 64; tw++) {
            if (th * tw > 256 || (th + kh - 1) * (tw + kw - 1) > HALO_PRMAX) continue;
            const long blocks = (long)((OH + th - 1) / th) * ((OW + tw - 1) / tw);
            const double eff = (double)OH * OW / (double)(blocks * 256);
            if (eff > best + 1e-9) {
                best = eff;
                *TH = th;
                *TW = tw;
            }
        }
    }
    return *TH > 0;
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_conv_halo_blocks(int N, int OH, int OW, int Cop, int kh, int kw, int* TH, int* TW, long* blocks)
{
    if (!TH || !TW || !blocks || N < 1 || OH < 1 || OW < 1 || kh < 1 || kw < 1 || Cop < 1)
        return fail(SDN_EINVAL, "sdn_conv_halo_blocks: bad arguments");
    if (!halo_block(OH, OW, kh, kw, TH, TW)) {
        *blocks = 0;
        return SDN_OK;
    }
    *blocks = (long)((OH + *TH - 1) / *TH) * ((OW + *TW - 1) / *TW) * N * ((Cop + 127) / 128);
    return SDN_OK;
}

SDN_API int sdn_conv_halo(const void* in_planes, long plane_stride, int N, int IH, int IW, int Cip, float* out, int OH, int OW,
                          int Cop, int ntaps, const int8_t* dy, const int8_t* dx, int pad_mode, const void* w_kmajor,
                          int w_rows, const float* bias, int act, double* stats, int accumulate, sdnStream stream)
{
    if (!in_planes || !out || !w_kmajor || !dy || !dx) return fail(SDN_EINVAL, "sdn_conv_halo: null pointer");
    if (ntaps < 9 || ntaps > CONV_MAX_TAPS) return fail(SDN_EINVAL, "sdn_conv_halo: ntaps %d not in 9..%d", ntaps, CONV_MAX_TAPS);
    if ((Cip & 31) || (Cop & 15)) return fail(SDN_EINVAL, "sdn_conv_halo: Cip %% 32, Cop %% 16 (%d, %d)", Cip, Cop);
    if (N < 1 || OH < 1 || OW < 1 || IH < 1 || IW < 1) return fail(SDN_EINVAL, "sdn_conv_halo: bad geometry");
    if ((size_t)IH * IW * Cip * 2 >= 0x7fffff00u) return fail(SDN_EINVAL, "sdn_conv_halo: one input image plane must stay below 2 GiB");
    int dy0 = 127, dy1 = -128, dx0 = 127, dx1 = -128;
    for (int t = 0; t < ntaps; t++) {
        dy0 = dy[t] < dy0 ? dy[t] : dy0;
        dy1 = dy[t] > dy1 ? dy[t] : dy1;
        dx0 = dx[t] < dx0 ? dx[t] : dx0;
        dx1 = dx[t] > dx1 ? dx[t] : dx1;
    }
    const int kh = dy1 - dy0 + 1, kw = dx1 - dx0 + 1;
    if (kh * kw != ntaps) return fail(SDN_EINVAL, "sdn_conv_halo: the taps must fill their %d x %d window", kh, kw);
    ConvHaloParams P;
    if (!halo_block(OH, OW, kh, kw, &P.TH, &P.TW)) return fail(SDN_EINVAL, "sdn_conv_halo: no output block fits a %d x %d window", kh, kw);
    P.in = (const __bf16*)in_planes; P.plane_stride = plane_stride; P.out = out;
    P.w = (const __bf16*)w_kmajor; P.bias = bias; P.stats = stats;
    P.N = N; P.IH = IH; P.IW = IW; P.Cip = Cip; P.OH = OH; P.OW = OW; P.Cop = Cop;
    P.tiles_y = (OH + P.TH - 1) / P.TH; P.tiles_x = (OW + P.TW - 1) / P.TW;
    P.PW = P.TW + kw - 1; P.PR = (P.TH + kh - 1) * P.PW;
    P.dy_min = dy0; P.dx_min = dx0;
    P.nsteps = ntaps * (Cip >> 5); P.w_rows = w_rows;
    P.pad_mode = pad_mode; P.act = act; P.accumulate = accumulate;
    P.ntiles = (Cop + 127) / 128;
    if (w_rows < P.ntiles * 128) return fail(SDN_EINVAL, "sdn_conv_halo: weight rows %d < %d", w_rows, P.ntiles * 128);
    P.taps.n = ntaps;
    for (int t = 0; t < ntaps; t++) {
        P.taps.ky[t] = (signed char)(dy[t] - dy0);
        P.taps.kx[t] = (signed char)(dx[t] - dx0);
    }
    const long blocks = (long)P.tiles_y * P.tiles_x * N * P.ntiles;
    hipStream_t st = (hipStream_t)stream;
    TimedLaunch timed(TIME_CONV_GEMM, st, 2.0 * N * OH * OW * (double)ntaps * Cip * Cop);
#ifdef SDN_TILE_PROBES
    {
        const char* e = getenv("SDN_TILE_VARIANT");
        const int var = e ? atoi(e) : 0;
        if (var == 1) { hipLaunchKernelGGL((k_conv_halo<1>), dim3((unsigned)blocks), dim3(512), 0, st, P); return check_launch("k_conv_halo"); }
        if (var == 2) { hipLaunchKernelGGL((k_conv_halo<2>), dim3((unsigned)blocks), dim3(512), 0, st, P); return check_launch("k_conv_halo"); }
    }
#endif
    hipLaunchKernelGGL((k_conv_halo<0>), dim3((unsigned)blocks), dim3(512), 0, st, P);
    return check_launch("k_conv_halo");
}
