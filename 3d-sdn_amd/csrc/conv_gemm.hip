// Implicit-GEMM convolution on the CDNA4 matrix cores: the forward / data-gradient workhorse of the textural networks.
//
// Reference: the Conv2d / ConvTranspose2d layers of GlobalGenerator, Encoder, NLayerDiscriminator
// (/root/reference/textural/models/networks.py:211-239, 286-308, 412-449), which the reference runs through cuDNN with
// ReflectionPad2d / InstanceNorm2d / ReLU as separate kernels.
//
// One kernel covers every case as a "gather GEMM" over channels-last activations:
//     out[n, qy*os + py, qx*os + px, co] (+)= act( bias[co] + sum_t sum_ci  f(in[n, qy*is + dy_t, qx*is + dx_t, ci]) * W_t[co, ci] )
//   - Conv2d (stride s, pad p):            os = 1, is = s, (dy, dx) = (ky - p, kx - p);
//   - ConvTranspose2d, and the data gradient of a strided Conv2d, as s*s phase launches: os = s, is = 1, each phase
//     owning the taps whose parity matches (no multiplications by inserted zeros);
//   - data gradient of a stride-1 Conv2d:  (dy, dx) = (p - ky, p - kx), transposed weights;
//   - coordinates outside the input are zeros or reflected (ReflectionPad2d folded into the gather);
//   - f = ReLU when the producer stored its pre-activation (the activation is applied on load, never materialised).
// GEMM view: M = output positions of one image (tile 128), N = output channels (tile 128 / 64 / 32), K = taps x padded
// input channels, walked in 16-channel groups.  A (activations, fp32 in HBM) is split to bf16 hi/lo while staged to LDS
// (double-buffered: one barrier per 32-deep step).  B (weights) never touches LDS: sdn_conv_pack_weights stores it
// pre-split in MFMA FRAGMENT order -- for every (32 output channels, 16 k) block the 64 lanes' 8-element fragments are
// consecutive, hi block then lo block -- so a wave fetches each operand block with one coalesced 1 KiB load straight
// into the registers the MFMA reads.  Epilogue: bias, LeakyReLU / tanh, InstanceNorm statistics (per (n, c)
// sum and sum of squares, fp64 atomics) and coalesced 128-B channel-contiguous stores.
//
// Roofline: MFMA-bound for the 1024-channel residual blocks (K = 9216), HBM/gather-bound for the 7x7 stem/head layers.
#include "conv_common.h"
#include "sdn_common.h"

namespace sdn {

struct ConvTaps {
    int n;
    signed char dy[CONV_MAX_TAPS];
    signed char dx[CONV_MAX_TAPS];
};

struct ConvGemmParams {
    const float* in;   // [N, IH, IW, Cip]
    float* out;        // [N, OH, OW, Cop]
    const __bf16* w;     // layout 0: [Corows / 32][Kp / 16][2 (hi, lo)][64 lanes][8] fragment-major (k_conv_gemm);
                         // layout 1: [Corows][Kp / 32][2 (hi, lo)][32] K-major rows (k_conv_gemm_ws)
    const float* bias;   // [>= Cop] or null
    double* stats;       // [N, STAT_SLOTS, Cop, 2] or null
    int N, IH, IW, Cip;
    int OH, OW, Cop;
    int QH, QW, istride, ostride, py, px;
    int Kp;
    int pad_mode, in_relu, act, accumulate;
    int ntiles;  // output-channel tiles (set by the launcher)
    ConvTaps taps;
};

template <int WM, int WN, int TM, int TN, int NPART>
// 3 workgroups per CU (146 VGPRs, 44 KB LDS each): more latency hiding, and 544-block grids still fit in one round
__global__ __launch_bounds__(256, 3) void k_conv_gemm(const ConvGemmParams P)
{
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(WM * WN == 4 && BM == 128, "four waves, 128 output positions per block");
    constexpr int A_ELEMS = lds_tile_elems(BM);
    constexpr int A_BUF = NPART * A_ELEMS;  // one stage: hi tile (+ lo tile)
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * A_BUF];
    __shared__ int s_outpix[BM];
    __shared__ int s_dy[CONV_MAX_TAPS], s_dx[CONV_MAX_TAPS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Q = P.QH * P.QW;
    const int mtiles = (Q + BM - 1) / BM;
    // XCD-aware tile order.  Hardware block b runs on XCD b % 8, and each XCD has its own L2: give every XCD a
    // contiguous range of output-position tiles with ALL channel tiles of each (channel tile fastest), so that the
    // blocks resident on one XCD at a time share their activation tiles (x ntiles) and the same few weight tiles.
    const int ntiles = P.ntiles;
    const unsigned nblk = gridDim.x;
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7u;
    const unsigned v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;  // bijective, see DESIGN.md
    const int mt_global = (int)(v / (unsigned)ntiles);
    const int n = mt_global / mtiles;
    const int mtile = mt_global - n * mtiles;
    const int m0 = mtile * BM;
    const int n0 = (int)(v % (unsigned)ntiles) * BN;

    if (tid < P.taps.n) {
        s_dy[tid] = P.taps.dy[tid];
        s_dx[tid] = P.taps.dx[tid];
    }
    if (tid < BM) {
        const int q = m0 + tid;
        int o = -1;
        if (q < Q) {
            const int qy = q / P.QW, qx = q - qy * P.QW;
            o = (n * P.OH + qy * P.ostride + P.py) * P.OW + qx * P.ostride + P.px;
        }
        s_outpix[tid] = o;
    }

    // ---- A loader: thread -> (row, 16-channel half) of the 128 x 32 step tile
    const int arow = tid >> 1, ahalf = tid & 1;
    const int aq = m0 + arow;
    const bool arow_ok = aq < Q;
    const int aqy = arow_ok ? aq / P.QW : 0, aqx = arow_ok ? aq - aqy * P.QW : 0;
    const int iy0 = aqy * P.istride, ix0 = aqx * P.istride;
    const int gpt = P.Cip >> 4;  // 16-channel groups per tap
    const int G = P.taps.n * gpt;
    int a_tap = 0, a_cg = ahalf;  // group index g = 2 * step + ahalf, kept as (tap, group in tap)
    while (a_cg >= gpt) {
        a_cg -= gpt;
        a_tap++;
    }
    const float* in_n = P.in + (size_t)n * P.IH * P.IW * P.Cip;

    // ---- B operand: this wave's TN column tiles, fragment-major in HBM
    const int wm0 = (wave / WN) * TM * 32, wn0 = (wave % WN) * TN * 32;
    const int ks16_total = P.Kp >> 4;
    const __bf16* wbase = P.w + ((size_t)((n0 + wn0) >> 5) * ks16_total) * 1024 + lane * 8;

    const int nsteps = P.Kp / CONV_BK;

    __syncthreads();  // tap table visible

    // Software pipeline (one barrier per 32-deep step s):
    //   LDS tile (s & 1) holds step s (bf16 hi / lo);  register set "cur" holds the RAW fp32 data of step s + 1 (loaded
    //   one whole step earlier);  at the top of step s the loads of step s + 2 are issued into the other register set.
    //   The split / ReLU / LDS store of step s + 1 is written BETWEEN the MFMAs of step s, so its VALU and DS-write
    //   instructions issue in the shadow of the matrix pipe (4-5 issue slots per 32-cycle MFMA), and the B fragments
    //   of step s + 1 are re-loaded half a step ahead into the registers the finished k16 half just released.
    struct ARegs {
        f32x4 v0, v1, v2, v3;
    };
    ARegs ra, rb;
    bf16x8 bfr[TN][2][NPART];  // [column tile][k16 half][hi, lo]; constant indices only (stays in registers)

#define CONV_LOAD_A(R)                                                                                                 \
    {                                                                                                                  \
        bool ok = arow_ok && a_tap < P.taps.n;                                                                         \
        int iy = 0, ix = 0;                                                                                            \
        if (ok) {                                                                                                      \
            iy = iy0 + s_dy[a_tap];                                                                                    \
            ix = ix0 + s_dx[a_tap];                                                                                    \
            ok = resolve_coord(iy, P.IH, P.pad_mode) && resolve_coord(ix, P.IW, P.pad_mode);                           \
        }                                                                                                              \
        R.v0 = R.v1 = R.v2 = R.v3 = f32x4{0.f, 0.f, 0.f, 0.f};                                                         \
        if (ok) {                                                                                                      \
            const f32x4* src = reinterpret_cast<const f32x4*>(in_n + ((size_t)iy * P.IW + ix) * P.Cip + a_cg * 16);    \
            R.v0 = src[0];                                                                                             \
            R.v1 = src[1];                                                                                             \
            R.v2 = src[2];                                                                                             \
            R.v3 = src[3];                                                                                             \
        }                                                                                                              \
        a_cg += 2;                                                                                                     \
        while (a_cg >= gpt) {                                                                                          \
            a_cg -= gpt;                                                                                               \
            a_tap++;                                                                                                   \
        }                                                                                                              \
    }

#define CONV_LOAD_B(step, ks)                                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < TN; nt++) _Pragma("unroll") for (int pp = 0; pp < NPART; pp++)             \
        bfr[nt][ks][pp] = *reinterpret_cast<const bf16x8*>(                                                            \
            wbase + (((size_t)nt * ks16_total + 2 * (step) + (ks)) * 2 + pp) * 512);

    // split pair q (0..3) of two f32x4 (x0 = channels 0-3, x1 = channels 4-7 of the half) into packed bf16 words
    auto split_pair = [&](const f32x4& x0, const f32x4& x1, int q, uint32_t& hw, uint32_t& lw) {
        float f0 = q < 2 ? x0[2 * q] : x1[2 * (q - 2)];
        float f1 = q < 2 ? x0[2 * q + 1] : x1[2 * (q - 2) + 1];
        if (P.in_relu) {
            f0 = fmaxf(f0, 0.f);
            f1 = fmaxf(f1, 0.f);
        }
        const SplitBf16 sp = split2(f0, f1);
        hw = __builtin_bit_cast(uint32_t, sp.hi);
        lw = __builtin_bit_cast(uint32_t, sp.lo);
    };

    constexpr int TILES = TM * TN;        // 32x32 MFMA tiles per wave
    constexpr int PPS = 4 / TILES;        // bf16 pairs converted in the shadow of each tile's three MFMAs
    static_assert(TILES == 1 || TILES == 2 || TILES == 4, "pair schedule");

    // k16 half `ks` of the tile at As.  With FILL, the split / store of half `ks` of the NEXT step's raw data (x0, x1)
    // is placed between the MFMA groups and pinned there (sched_barrier), so that it issues in the matrix pipe's shadow.
#define CONV_HALF(As, An, ks, x0, x1, FILL)                                                                            \
    {                                                                                                                  \
        bf16x8 af[NPART][TM];                                                                                          \
        uint32_t hw[4], lw[4];                                                                                         \
        _Pragma("unroll") for (int mt = 0; mt < TM; mt++)                                                              \
        {                                                                                                              \
            const int off = lds_row(wm0 + mt * 32 + fr) + (ks)*16 + fkq;                                               \
            af[0][mt] = *reinterpret_cast<const bf16x8*>((As) + off);                                                  \
            if constexpr (NPART == 2) af[NPART - 1][mt] = *reinterpret_cast<const bf16x8*>((As) + A_ELEMS + off);      \
        }                                                                                                              \
        _Pragma("unroll") for (int mt = 0; mt < TM; mt++) _Pragma("unroll") for (int nt = 0; nt < TN; nt++)            \
        {                                                                                                              \
            if constexpr (NPART == 2) {                                                                                \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[NPART - 1][mt], bfr[nt][ks][0], acc[mt][nt], \
                                                                      0, 0, 0);                                        \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][mt], bfr[nt][ks][NPART - 1], acc[mt][nt], \
                                                                      0, 0, 0);                                        \
            }                                                                                                          \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][mt], bfr[nt][ks][0], acc[mt][nt], 0, 0, 0);    \
            if (FILL) {                                                                                                \
                _Pragma("unroll") for (int q = (mt * TN + nt) * PPS; q < (mt * TN + nt + 1) * PPS; q++)                \
                    split_pair(x0, x1, q, hw[q], lw[q]);                                                               \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
        }                                                                                                              \
        if (FILL) {                                                                                                    \
            *reinterpret_cast<uint4*>((An) + lds_row(arow) + ahalf * 16 + (ks)*8) = uint4{hw[0], hw[1], hw[2], hw[3]}; \
            if constexpr (NPART == 2)                                                                                  \
                *reinterpret_cast<uint4*>((An) + A_ELEMS + lds_row(arow) + ahalf * 16 + (ks)*8) =                      \
                    uint4{lw[0], lw[1], lw[2], lw[3]};                                                                 \
        }                                                                                                              \
    }

    // one pipeline step that has a successor: CUR holds raw step s + 1, NXT receives step s + 2
#define CONV_STEP(s, CUR, NXT)                                                                                         \
    {                                                                                                                  \
        __bf16* As = smem + ((s)&1) * A_BUF;                                                                           \
        __bf16* An = smem + (((s) + 1) & 1) * A_BUF;                                                                   \
        __syncthreads();                                                                                               \
        if ((s) + 2 < nsteps) CONV_LOAD_A(NXT);                                                                        \
        CONV_HALF(As, An, 0, CUR.v0, CUR.v1, true);                                                                    \
        CONV_LOAD_B((s) + 1, 0);                                                                                       \
        CONV_HALF(As, An, 1, CUR.v2, CUR.v3, true);                                                                    \
        CONV_LOAD_B((s) + 1, 1);                                                                                       \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mt = 0; mt < TM; mt++)
#pragma unroll
        for (int nt = 0; nt < TN; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;

    (void)G;
    const int fr = lane & 31, fkq = (lane >> 5) * 8;
    // prologue: tile 0 -> LDS, raw tile 1 -> rb, B fragments of step 0
    CONV_LOAD_A(ra);
    if (nsteps > 1) CONV_LOAD_A(rb);
    CONV_LOAD_B(0, 0);
    CONV_LOAD_B(0, 1);
    {
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int q = 0; q < 4; q++) split_pair(h ? ra.v2 : ra.v0, h ? ra.v3 : ra.v1, q, hw[q], lw[q]);
            *reinterpret_cast<uint4*>(smem + lds_row(arow) + ahalf * 16 + h * 8) = uint4{hw[0], hw[1], hw[2], hw[3]};
            if constexpr (NPART == 2)
                *reinterpret_cast<uint4*>(smem + A_ELEMS + lds_row(arow) + ahalf * 16 + h * 8) =
                    uint4{lw[0], lw[1], lw[2], lw[3]};
        }
    }
    int step = 0;
    for (; step + 2 < nsteps; step += 2) {  // both steps have successors
        CONV_STEP(step, rb, ra);
        CONV_STEP(step + 1, ra, rb);
    }
    if (step + 1 < nsteps) {  // two steps left: the first still stages its successor
        CONV_STEP(step, rb, ra);
        step++;
    }
    {  // last step: MFMAs only
        __bf16* As = smem + (step & 1) * A_BUF;
        __syncthreads();
        CONV_HALF(As, As, 0, ra.v0, ra.v1, false);
        CONV_HALF(As, As, 1, ra.v2, ra.v3, false);
    }

    // ---- epilogue
    if (P.stats) __syncthreads();  // every wave is done with the A tiles before they are reused for the statistics
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
                float v = acc[mt][nt][r] + bias;
                s1 += v;
                s2 += v * v;
                if (P.act == 1)
                    v = v > 0.f ? v : 0.2f * v;
                else if (P.act == 2)
                    v = tanhf(v);
                float* dst = P.out + (size_t)o * P.Cop + co;
                if (P.accumulate)
                    *dst += v;
                else
                    *dst = v;
            }
        }
        if (P.stats) {
            // per-column partial sums of this wave -> LDS (the tiles are dead after the main loop's last barrier)
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                float* red = reinterpret_cast<float*>(smem);
                const int slot = ((wave / WN) * BN + wn0 + nt * 32 + col) * 2;
                red[slot] = s1;
                red[slot + 1] = s2;
            }
        }
    }
    if (P.stats) {
        __syncthreads();
        if (tid < BN) {
            const float* red = reinterpret_cast<const float*>(smem);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; w++) {
                s1 += red[(w * BN + tid) * 2];
                s2 += red[(w * BN + tid) * 2 + 1];
            }
            const int co = n0 + tid;
            if (co < P.Cop) {
                const int slot = mtile & (STAT_SLOTS - 1);
                double* st = P.stats + (((size_t)n * STAT_SLOTS + slot) * P.Cop + co) * 2;
                unsafeAtomicAdd(st, (double)s1);
                unsafeAtomicAdd(st + 1, (double)s2);
            }
        }
    }
}

// Wave-specialised variant for layers with >= 128 output channels and many positions: a 256 x 128 output tile per
// workgroup of 12 waves.  Waves 0-7 are CONSUMERS (4 x 2 grid, 64 x 64 each: two MFMA waves per SIMD, which hide each
// other's LDS latency), waves 8-11 are PRODUCERS (one per SIMD: the hardware deals a workgroup's waves to SIMDs
// cyclically).  At step s the consumers multiply tile s from LDS buffer (s & 1) while the producers gather, split
// (fp32 -> bf16 hi / lo, ReLU on load) and store tile s + 1 into buffer ((s + 1) & 1) and already have the global loads of
// tile s + 2 in flight; one barrier per step hands the buffers over.  Staging (vector memory, VALU, LDS stores) thus
// runs on other waves than the MFMAs, and the 256 x 128 tile needs 48 KB of operands per 2 x 768 MFMA cycles instead of
// the 32 KB per 768 of the 128 x 128 kernel (the L2 / Infinity-cache path was what bounded that one).
template <int NPART>
__global__ __launch_bounds__(768, 3) void k_conv_gemm_ws(const ConvGemmParams P)
{
    constexpr int WM = 4, WN = 2, TM = 2, TN = 2;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(BM == 256 && BN == 128, "256 x 128 tile");
    constexpr int A_ELEMS = lds_tile_elems(BM), B_ELEMS = lds_tile_elems(BN);
    constexpr int STAGE = NPART * (A_ELEMS + B_ELEMS);  // one buffer: A hi (+ lo), B hi (+ lo)
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * STAGE];
    __shared__ int s_outpix[BM];
    __shared__ int s_dy[CONV_MAX_TAPS], s_dx[CONV_MAX_TAPS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 8;  // wave-uniform
    const int Q = P.QH * P.QW;
    const int mtiles = (Q + BM - 1) / BM;
    // XCD-aware tile order.  Hardware block b runs on XCD b % 8, and each XCD has its own L2: give every XCD a
    // contiguous range of output-position tiles with ALL channel tiles of each (channel tile fastest), so that the
    // blocks resident on one XCD at a time share their activation tiles (x ntiles) and the same few weight tiles.
    const int ntiles = P.ntiles;
    const unsigned nblk = gridDim.x;
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7u;
    const unsigned v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;  // bijective
    const int mt_global = (int)(v / (unsigned)ntiles);
    const int n = mt_global / mtiles;
    const int mtile = mt_global - n * mtiles;
    const int m0 = mtile * BM;
    const int n0 = (int)(v % (unsigned)ntiles) * BN;

    if (tid < P.taps.n) {  // (taps.n <= 64 < blockDim)
        s_dy[tid] = P.taps.dy[tid];
        s_dx[tid] = P.taps.dx[tid];
    }
    if (tid < BM) {
        const int q = m0 + tid;
        int o = -1;
        if (q < Q) {
            const int qy = q / P.QW, qx = q - qy * P.QW;
            o = (n * P.OH + qy * P.ostride + P.py) * P.OW + qx * P.ostride + P.px;
        }
        s_outpix[tid] = o;
    }
    const int nsteps = P.Kp / CONV_BK;
    const int wm0 = ((wave & 7) / WN) * TM * 32, wn0 = ((wave & 7) % WN) * TN * 32;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mt = 0; mt < TM; mt++)
#pragma unroll
        for (int nt = 0; nt < TN; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;

    __syncthreads();  // tap table visible

    if (producer) {
        // ---------------------------------------------------------------- producers: 256 threads stage A and B
        const int ptid = tid - 512;
        const int arow = ptid >> 1, ahalf = ptid & 1;  // A rows arow and arow + 128, B row arow: (row, 16-channel half)
        const int aq = m0 + arow, aq2 = aq + 128;
        const bool arow_ok = aq < Q, arow2_ok = aq2 < Q;
        const int aqy = arow_ok ? aq / P.QW : 0, aqx = arow_ok ? aq - aqy * P.QW : 0;
        const int aqy2 = arow2_ok ? aq2 / P.QW : 0, aqx2 = arow2_ok ? aq2 - aqy2 * P.QW : 0;
        const int iy0 = aqy * P.istride, ix0 = aqx * P.istride;
        const int iy02 = aqy2 * P.istride, ix02 = aqx2 * P.istride;
        const int gpt = P.Cip >> 4;  // 16-channel groups per tap
        int a_tap = 0, a_cg = ahalf;  // group index g = 2 * step + ahalf, kept as (tap, group in tap)
        while (a_cg >= gpt) {
            a_cg -= gpt;
            a_tap++;
        }
        const float* in_n = P.in + (size_t)n * P.IH * P.IW * P.Cip;
        const bool b_active = arow < BN;  // B tile: (row, 16-k half), rows < BN
        // weights: [row][step][part][32 k]
        const __bf16* bsrc = P.w + ((size_t)(n0 + (b_active ? arow : 0)) * nsteps) * (2 * CONV_BK) + ahalf * 16;

        // Straight-line code from here on: every PROD_LOAD issues the same number of loads, unconditionally (rows /
        // taps that do not exist read a dummy address and are zeroed at store time), so the compiler can wait for the
        // OLDER register set with a counted s_waitcnt vmcnt(N) while the newer set's loads stay in flight.  (With the
        // loads inside `if (ok)` it fell back to vmcnt(0) and every step paid a full memory round trip.)
        struct Raw {
            f32x4 v0, v1, v2, v3;      // 16 activations of row arow
            f32x4 u0, u1, u2, u3;      // 16 activations of row arow + 128
            uint4 bh0, bh1, bl0, bl1;  // 16 weights hi, 16 weights lo
            bool ok, ok2;
        };
        Raw ra, rb;
        const int last_tap = P.taps.n - 1, last_step = nsteps - 1;

#define PROD_LOAD(R, step)                                                                                             \
    {                                                                                                                  \
        const int tap = min(a_tap, last_tap);                                                                          \
        int iy = iy0 + s_dy[tap], ix = ix0 + s_dx[tap];                                                                \
        bool ok = arow_ok && a_tap <= last_tap;                                                                        \
        const bool oky = resolve_coord(iy, P.IH, P.pad_mode), okx = resolve_coord(ix, P.IW, P.pad_mode);               \
        ok = ok && oky && okx;                                                                                         \
        const size_t aoff = ok ? ((size_t)iy * P.IW + ix) * P.Cip + a_cg * 16 : (size_t)0;                             \
        const f32x4* src = reinterpret_cast<const f32x4*>(in_n + aoff);                                                \
        R.v0 = src[0];                                                                                                 \
        R.v1 = src[1];                                                                                                 \
        R.v2 = src[2];                                                                                                 \
        R.v3 = src[3];                                                                                                 \
        R.ok = ok;                                                                                                     \
        {                                                                                                              \
            int jy = iy02 + s_dy[tap], jx = ix02 + s_dx[tap];                                                          \
            bool k2 = arow2_ok && a_tap <= last_tap;                                                                   \
            const bool k2y = resolve_coord(jy, P.IH, P.pad_mode), k2x = resolve_coord(jx, P.IW, P.pad_mode);           \
            k2 = k2 && k2y && k2x;                                                                                     \
            const size_t boff = k2 ? ((size_t)jy * P.IW + jx) * P.Cip + a_cg * 16 : (size_t)0;                         \
            const f32x4* src2 = reinterpret_cast<const f32x4*>(in_n + boff);                                           \
            R.u0 = src2[0];                                                                                            \
            R.u1 = src2[1];                                                                                            \
            R.u2 = src2[2];                                                                                            \
            R.u3 = src2[3];                                                                                            \
            R.ok2 = k2;                                                                                                \
        }                                                                                                              \
        const uint4* sh = reinterpret_cast<const uint4*>(bsrc + (size_t)min((step), last_step) * (2 * CONV_BK));       \
        R.bh0 = sh[0];                                                                                                 \
        R.bh1 = sh[1];                                                                                                 \
        if constexpr (NPART == 2) {                                                                                    \
            R.bl0 = sh[4];                                                                                             \
            R.bl1 = sh[5];                                                                                             \
        }                                                                                                              \
        a_cg += 2;                                                                                                     \
        {                                                                                                              \
            const int w1 = a_cg >= gpt ? 1 : 0;                                                                        \
            a_cg -= w1 ? gpt : 0;                                                                                      \
            a_tap += w1;                                                                                               \
            const int w2 = a_cg >= gpt ? 1 : 0;                                                                        \
            a_cg -= w2 ? gpt : 0;                                                                                      \
            a_tap += w2;                                                                                               \
        }                                                                                                              \
    }

        auto split4 = [&](f32x4 x, bool ok, uint32_t& h01, uint32_t& h23, uint32_t& l01, uint32_t& l23) {
            if (P.in_relu) {
                x[0] = fmaxf(x[0], 0.f);
                x[1] = fmaxf(x[1], 0.f);
                x[2] = fmaxf(x[2], 0.f);
                x[3] = fmaxf(x[3], 0.f);
            }
            x[0] = ok ? x[0] : 0.f;
            x[1] = ok ? x[1] : 0.f;
            x[2] = ok ? x[2] : 0.f;
            x[3] = ok ? x[3] : 0.f;
            const SplitBf16 s0 = split2(x[0], x[1]), s1 = split2(x[2], x[3]);
            h01 = __builtin_bit_cast(uint32_t, s0.hi);
            h23 = __builtin_bit_cast(uint32_t, s1.hi);
            l01 = __builtin_bit_cast(uint32_t, s0.lo);
            l23 = __builtin_bit_cast(uint32_t, s1.lo);
        };

#define PROD_STORE(R, buf)                                                                                             \
    {                                                                                                                  \
        __bf16* As = smem + (buf)*STAGE;                                                                               \
        __bf16* Bs = As + NPART * A_ELEMS;                                                                             \
        uint4 h0, h1, l0, l1;                                                                                          \
        split4(R.v0, R.ok, h0.x, h0.y, l0.x, l0.y);                                                                    \
        split4(R.v1, R.ok, h0.z, h0.w, l0.z, l0.w);                                                                    \
        split4(R.v2, R.ok, h1.x, h1.y, l1.x, l1.y);                                                                    \
        split4(R.v3, R.ok, h1.z, h1.w, l1.z, l1.w);                                                                    \
        uint4* da = reinterpret_cast<uint4*>(As + lds_row(arow) + ahalf * 16);                                         \
        da[0] = h0;                                                                                                    \
        da[1] = h1;                                                                                                    \
        if constexpr (NPART == 2) {                                                                                    \
            uint4* dl = reinterpret_cast<uint4*>(As + A_ELEMS + lds_row(arow) + ahalf * 16);                           \
            dl[0] = l0;                                                                                                \
            dl[1] = l1;                                                                                                \
        }                                                                                                              \
        split4(R.u0, R.ok2, h0.x, h0.y, l0.x, l0.y);                                                                   \
        split4(R.u1, R.ok2, h0.z, h0.w, l0.z, l0.w);                                                                   \
        split4(R.u2, R.ok2, h1.x, h1.y, l1.x, l1.y);                                                                   \
        split4(R.u3, R.ok2, h1.z, h1.w, l1.z, l1.w);                                                                   \
        da = reinterpret_cast<uint4*>(As + lds_row(arow + 128) + ahalf * 16);                                          \
        da[0] = h0;                                                                                                    \
        da[1] = h1;                                                                                                    \
        if constexpr (NPART == 2) {                                                                                    \
            uint4* dl = reinterpret_cast<uint4*>(As + A_ELEMS + lds_row(arow + 128) + ahalf * 16);                     \
            dl[0] = l0;                                                                                                \
            dl[1] = l1;                                                                                                \
        }                                                                                                              \
        if (b_active) {                                                                                                \
            uint4* db = reinterpret_cast<uint4*>(Bs + lds_row(arow) + ahalf * 16);                                     \
            db[0] = R.bh0;                                                                                             \
            db[1] = R.bh1;                                                                                             \
            if constexpr (NPART == 2) {                                                                                \
                uint4* dbl = reinterpret_cast<uint4*>(Bs + B_ELEMS + lds_row(arow) + ahalf * 16);                      \
                dbl[0] = R.bl0;                                                                                        \
                dbl[1] = R.bl1;                                                                                        \
            }                                                                                                          \
        }                                                                                                              \
    }

        // prologue: tile 0 staged, raw tile 1 in rb.  nsteps is even (Kp % 64 == 0, checked by the launcher).
        PROD_LOAD(ra, 0);
        PROD_LOAD(rb, 1);
        PROD_STORE(ra, 0);
        // step s: issue the loads of tile s + 2, stage tile s + 1 (its raw data was loaded one step earlier).  Loads and
        // stores past the last tile are harmless: they read valid dummy addresses and fill a buffer nobody reads.
        for (int step = 0; step < nsteps; step += 2) {
            __syncthreads();
            PROD_LOAD(ra, step + 2);
            PROD_STORE(rb, 1);
            __syncthreads();
            PROD_LOAD(rb, step + 3);
            PROD_STORE(ra, 0);
        }
#undef PROD_LOAD
#undef PROD_STORE
    } else {
        // ---------------------------------------------------------------- consumers: MFMAs on the staged tile
        for (int step = 0; step < nsteps; step++) {
            __syncthreads();
            const __bf16* As = smem + (step & 1) * STAGE;
            const __bf16* Bs = As + NPART * A_ELEMS;
            mfma_step<TM, TN, NPART>(As, Bs, wm0, wn0, A_ELEMS, B_ELEMS, lane, acc);
        }
    }

    // ---- epilogue (consumer waves hold the accumulators; producers only take part in the barriers)
    if (P.stats) __syncthreads();  // every wave is done with the tiles before they are reused for the statistics
    const int col = lane & 31;
    if (!producer) {
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
                    float val = acc[mt][nt][r] + bias;
                    s1 += val;
                    s2 += val * val;
                    if (P.act == 1)
                        val = val > 0.f ? val : 0.2f * val;
                    else if (P.act == 2)
                        val = tanhf(val);
                    float* dst = P.out + (size_t)o * P.Cop + co;
                    if (P.accumulate)
                        *dst += val;
                    else
                        *dst = val;
                }
            }
            if (P.stats) {
                // per-column partial sums of this wave -> LDS
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lane < 32) {
                    float* red = reinterpret_cast<float*>(smem);
                    const int slot = (((wave & 7) / WN) * BN + wn0 + nt * 32 + col) * 2;
                    red[slot] = s1;
                    red[slot + 1] = s2;
                }
            }
        }
    }
    if (P.stats) {
        __syncthreads();
        if (tid < BN) {
            const float* red = reinterpret_cast<const float*>(smem);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; w++) {
                s1 += red[(w * BN + tid) * 2];
                s2 += red[(w * BN + tid) * 2 + 1];
            }
            const int co = n0 + tid;
            if (co < P.Cop) {
                const int slot = mtile & (STAT_SLOTS - 1);
                double* st = P.stats + (((size_t)n * STAT_SLOTS + slot) * P.Cop + co) * 2;
                unsafeAtomicAdd(st, (double)s1);
                unsafeAtomicAdd(st + 1, (double)s2);
            }
        }
    }
}

template <int WM, int WN, int TM, int TN>
static int launch_conv(ConvGemmParams P, int npart, hipStream_t st)
{
    constexpr int BN = WN * TN * 32;
    const int Q = P.QH * P.QW;
    P.ntiles = (P.Cop + BN - 1) / BN;
    const dim3 grid((unsigned)(((Q + 127) / 128) * P.N * P.ntiles));
    // algorithmic work of this launch: 2 * positions * taps * Cin(padded) * Cout(padded) flops
    TimedLaunch timed(TIME_CONV_GEMM, st, 2.0 * P.N * Q * (double)P.taps.n * P.Cip * P.Cop);
    if (npart == 2)
        hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 2>), grid, dim3(256), 0, st, P);
    else
        hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 1>), grid, dim3(256), 0, st, P);
    return check_launch("k_conv_gemm");
}

static int launch_conv_ws(ConvGemmParams P, int npart, hipStream_t st)
{
    const int Q = P.QH * P.QW;
    P.ntiles = (P.Cop + 127) / 128;
    const dim3 grid((unsigned)(((Q + 255) / 256) * P.N * P.ntiles));
    TimedLaunch timed(TIME_CONV_GEMM, st, 2.0 * P.N * Q * (double)P.taps.n * P.Cip * P.Cop);
    if (npart == 2)
        hipLaunchKernelGGL((k_conv_gemm_ws<2>), grid, dim3(768), 0, st, P);
    else
        hipLaunchKernelGGL((k_conv_gemm_ws<1>), grid, dim3(768), 0, st, P);
    return check_launch("k_conv_gemm_ws");
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_conv_gemm(const float* in, int N, int IH, int IW, int Cip, float* out, int OH, int OW, int Cop, int QH,
                          int QW, int istride, int ostride, int py, int px, int ntaps, const int8_t* dy,
                          const int8_t* dx, int pad_mode, int in_relu, const void* w_packed, int w_layout, int Kp,
                          int w_rows, const float* bias, int act, double* stats, int accumulate, int precision,
                          sdnStream stream)
{
    if (!in || !out || !w_packed || !dy || !dx) return fail(SDN_EINVAL, "sdn_conv_gemm: null pointer");
    if (ntaps < 1 || ntaps > CONV_MAX_TAPS) return fail(SDN_EINVAL, "sdn_conv_gemm: ntaps %d not in 1..%d", ntaps, CONV_MAX_TAPS);
    if ((Cip & 15) || (Cop & 15)) return fail(SDN_EINVAL, "sdn_conv_gemm: channel counts must be padded to 16 (%d, %d)", Cip, Cop);
    if (Kp % CONV_BK || Kp < ntaps * Cip) return fail(SDN_EINVAL, "sdn_conv_gemm: Kp %d does not cover %d taps x %d", Kp, ntaps, Cip);
    if (precision != 1 && precision != 3) return fail(SDN_EINVAL, "sdn_conv_gemm: precision must be 1 (bf16) or 3 (bf16x3)");
    if (N < 1 || QH < 1 || QW < 1 || istride < 1 || ostride < 1) return fail(SDN_EINVAL, "sdn_conv_gemm: bad geometry");
    if ((QH - 1) * ostride + py >= OH || (QW - 1) * ostride + px >= OW) return fail(SDN_EINVAL, "sdn_conv_gemm: output grid exceeds the output tensor");
    ConvGemmParams P;
    P.in = in; P.out = out; P.w = (const __bf16*)w_packed; P.bias = bias; P.stats = stats;
    P.N = N; P.IH = IH; P.IW = IW; P.Cip = Cip; P.OH = OH; P.OW = OW; P.Cop = Cop;
    P.QH = QH; P.QW = QW; P.istride = istride; P.ostride = ostride; P.py = py; P.px = px; P.Kp = Kp;
    P.pad_mode = pad_mode; P.in_relu = in_relu; P.act = act; P.accumulate = accumulate;
    P.taps.n = ntaps;
    for (int t = 0; t < ntaps; t++) {
        P.taps.dy[t] = dy[t];
        P.taps.dx[t] = dx[t];
    }
    const int npart = precision == 3 ? 2 : 1;
    hipStream_t st = (hipStream_t)stream;
    if (w_layout != 0 && w_layout != 1) return fail(SDN_EINVAL, "sdn_conv_gemm: weight layout %d", w_layout);
    if (w_layout == 1) {  // wave-specialised 256 x 128 kernel: K-major weight rows, K in pairs of steps
        if (Cop < 128 || w_rows < ((Cop + 127) / 128) * 128 || (Kp & 63))
            return fail(SDN_EINVAL, "sdn_conv_gemm: layout 1 needs Cout >= 128, padded rows and Kp %% 64 == 0");
        return launch_conv_ws(P, npart, st);
    }
    // the weight matrix must hold a whole number of N tiles
    if (Cop > 64) {
        if (w_rows < ((Cop + 127) / 128) * 128) return fail(SDN_EINVAL, "sdn_conv_gemm: weight rows %d < padded Cout", w_rows);
        return launch_conv<2, 2, 2, 2>(P, npart, st);
    }
    if (Cop > 32) {
        if (w_rows < 64) return fail(SDN_EINVAL, "sdn_conv_gemm: weight rows %d < 64", w_rows);
        return launch_conv<2, 2, 2, 1>(P, npart, st);
    }
    if (w_rows < 32) return fail(SDN_EINVAL, "sdn_conv_gemm: weight rows %d < 32", w_rows);
    return launch_conv<4, 1, 1, 1>(P, npart, st);
}
