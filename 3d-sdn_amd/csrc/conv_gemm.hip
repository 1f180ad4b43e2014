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

// r05: the s*s phase launches of a ConvTranspose2d forward / strided-conv data gradient as ONE launch.  Every phase keeps
// its own output sub-grid (QH x QW at offset (py, px)), taps, packed weights and K length; the launcher gives each phase an
// 8-aligned block range (so that block b of a phase still runs on XCD b % 8), longest K first.  Per phase the grids are small
// (24 x 78 positions x 4 images = 59 position tiles x 4 channel tiles for 768 workgroup slots) and the 4-tap phase of a 3x3
// kernel runs four times as long as its 1-tap phase: as four launches the chip idles behind each of them.
constexpr int CONV_MAX_PHASES = 4, CONV_PHASE_TAPS = 16;
struct ConvPhase {
    const __bf16* w;
    int QH, QW, py, px, Kp, ntaps, first, nblocks;
    signed char dy[CONV_PHASE_TAPS];
    signed char dx[CONV_PHASE_TAPS];
};

struct ConvGemmParams {
    const float* in;   // [N, IH, IW, Cip]
    float* out;        // [N, OH, OW, Cop]
    const __bf16* w;     // [Corows / 32][Kp / 16][2 (hi, lo)][64 lanes][8]  fragment-major, see sdn_conv_pack_weights
    const float* bias;   // [>= Cop] or null
    double* stats;       // [N, STAT_SLOTS, Cop, 2] or null
    int N, IH, IW, Cip;
    int OH, OW, Cop;
    int QH, QW, istride, ostride, py, px;
    int Kp, w_rows;
    int pad_mode, in_relu, act, accumulate;
    int ntiles;  // output-channel tiles (set by the launcher)
    // split K (set by the launcher): ksplit > 1 slices the K steps over ksplit blocks per output tile, which add their
    // raw partial sums into a zeroed (or accumulated-into) output; bias / activation / statistics then run as a
    // separate pass (k_bias_act_stats).  For layers whose M x N grid alone cannot fill 256 CUs (batch-1 inference, the
    // 192 x 624 default configuration).
    int ksplit, steps_per_split;
    // deterministic split K: slice z stores its partial tile to partials[z * out_elems + ...] (plain stores) and
    // k_split_reduce adds the slices in slice order; null: the slices meet in `out` through float atomics
    float* partials;
    size_t out_elems;
    ConvTaps taps;
    int nphase;   // 0: one launch = one phase (the fields above); else ph[0..nphase) replace w, QH, QW, py, px, Kp, taps
    ConvPhase ph[CONV_MAX_PHASES];
};

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// A-tile rows at the plain 80-B pitch: 5 r mod 16 sends the rows of every ds_read_b128 lane group ({0-3, 12-15, 20-27},
// ...) to 16 different 16-B slots.  (conv_common.h's lds_row adds 64 B every 16 rows for the wgrad kernel's transposing
// 4-byte stores; with it the fragment reads of this kernel conflicted two ways.  PMC, relu variant of the 128 x 128
// kernel: SQ_LDS_BANK_CONFLICT 20.8 M -> 10.4 M cycles per dispatch, profiles/r01_y_pmc_conv_sq.json ->
// r01_y2_pmc_conv_sq.json; the remainder are the 16-B stores.  No change in run time: LDS is not what bounds the kernel.)
__device__ __forceinline__ constexpr int lds_row_g(int r) { return r * LDS_PITCH; }
constexpr int TAP_OUTSIDE = -(1 << 20);  // dy of the tap slots behind the last tap: every coordinate test fails

template <int WM, int WN, int TM, int TN, int NPART, bool RELU>
// 3 workgroups per CU (168 VGPRs, 44 KB LDS each): more latency hiding, and 544-block grids still fit in one round
__global__ __launch_bounds__(256, 3) void k_conv_gemm(const ConvGemmParams P)
{
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(WM * WN == 4 && BM == 128, "four waves, 128 output positions per block");
    constexpr int A_ELEMS = lds_tile_elems(BM);
    constexpr int A_BUF = NPART * A_ELEMS;  // one stage: hi tile (+ lo tile)
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * A_BUF];
    __shared__ int s_outpix[BM];
    __shared__ int s_dy[CONV_MAX_TAPS + 2], s_dx[CONV_MAX_TAPS + 2];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the phase of this block (wave-uniform scalars; nphase == 0: the launch is its own single phase)
    int pQH = P.QH, pQW = P.QW, ppy = P.py, ppx = P.px, pKp = P.Kp, pntaps = P.taps.n;
    const __bf16* pw = P.w;
    unsigned bid = blockIdx.x, nblk = gridDim.x;
    int phase = -1;
    if (P.nphase > 0) {
        phase = 0;
        while (phase + 1 < P.nphase && blockIdx.x >= (unsigned)P.ph[phase + 1].first) phase++;
        pQH = P.ph[phase].QH; pQW = P.ph[phase].QW; ppy = P.ph[phase].py; ppx = P.ph[phase].px;
        pKp = P.ph[phase].Kp; pntaps = P.ph[phase].ntaps; pw = P.ph[phase].w;
        bid -= (unsigned)P.ph[phase].first;
        nblk = (unsigned)P.ph[phase].nblocks;
    }
    const int Q = pQH * pQW;
    const int mtiles = (Q + BM - 1) / BM;
    // XCD-aware tile order.  Hardware block b runs on XCD b % 8, and each XCD has its own L2: give every XCD a
    // contiguous range of output-position tiles with ALL channel tiles of each (channel tile fastest), so that the
    // blocks resident on one XCD at a time share their activation tiles (x ntiles) and the same few weight tiles.
    const int ntiles = P.ntiles;
    const unsigned xcd = bid & 7u, j = bid >> 3;   // (a phase's first block is a multiple of 8: bid % 8 is still the XCD)
    const unsigned q8 = nblk >> 3, r8 = nblk & 7u;
    const unsigned vz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;  // bijective, see DESIGN.md
    const unsigned tiles_all = nblk / (unsigned)P.ksplit;
    const int zs = (int)(vz / tiles_all);  // K slice
    const unsigned v = vz - (unsigned)zs * tiles_all;
    if (phase >= 0 && v >= (unsigned)(mtiles * P.N * ntiles)) return;   // padding block of a phase's 8-aligned range
    const int mt_global = (int)(v / (unsigned)ntiles);
    const int n = mt_global / mtiles;
    const int mtile = mt_global - n * mtiles;
    const int m0 = mtile * BM;
    const int n0 = (int)(v % (unsigned)ntiles) * BN;

    if (tid < CONV_MAX_TAPS + 2) {
        int tdy = TAP_OUTSIDE, tdx = 0;
        if (tid < pntaps) {
            if (phase >= 0) {
                tdy = P.ph[phase].dy[tid & (CONV_PHASE_TAPS - 1)];
                tdx = P.ph[phase].dx[tid & (CONV_PHASE_TAPS - 1)];
            } else {
                tdy = P.taps.dy[tid];
                tdx = P.taps.dx[tid];
            }
        }
        s_dy[tid] = tdy;
        s_dx[tid] = tdx;
    }
    if (tid < BM) {
        const int q = m0 + tid;
        int o = -1;
        if (q < Q) {
            const int qy = q / pQW, qx = q - qy * pQW;
            o = (n * P.OH + qy * P.ostride + ppy) * P.OW + qx * P.ostride + ppx;
        }
        s_outpix[tid] = o;
    }

    // ---- A loader: thread -> (row, 16-channel half) of the 128 x 32 step tile.  Raw buffer loads over one image: a
    // coordinate outside the image (zero padding, rows behind the last output position, the K padding behind the last
    // tap) becomes an out-of-range offset and the hardware returns zeros -- no branches, no zero fill.
    const int arow = tid >> 1, ahalf = tid & 1;
    const int aq = m0 + arow;
    const bool arow_ok = aq < Q;
    const int aqy = arow_ok ? aq / pQW : 0, aqx = arow_ok ? aq - aqy * pQW : 0;
    const int iy0 = arow_ok ? aqy * P.istride : TAP_OUTSIDE, ix0 = aqx * P.istride;
    const int gpt = P.Cip >> 4;  // 16-channel groups per tap
    const int step_lo = zs * P.steps_per_split;
    const int nsteps = min(pKp / CONV_BK - step_lo, P.steps_per_split);
    // K order.  Tap-major (k = t * Cip + c) re-reads an activation line once per tap, 32 steps apart -- by then other
    // workgroups' traffic has pushed it out of the 4 MB L2 and it comes over the fabric again (counters: 1.34 GB per launch
    // of the 1024-channel data gradient for 99 MB of operands).  When a step never straddles two taps (Cip % 32 == 0, no K
    // padding) the order is CHANNEL-BLOCK-major instead: step = cb * ntaps + t, k = step * 32 + c % 32 with c = cb * 32 + ...,
    // so the nine taps of a 32-channel block follow each other and the re-reads hit in L2 (sdn_conv_pack_weights lays the
    // weight columns out by the same rule).
    const int ntaps_s = pntaps;
    const bool cmajor = (P.Cip & 31) == 0 && pKp == ntaps_s * P.Cip;
    // (tap, 16-channel group in tap) of the even half of the next step: wave-uniform, advanced with scalar ops
    int st0, sc0;
    if (cmajor) {
        const int cb = step_lo / ntaps_s;
        st0 = step_lo - cb * ntaps_s;
        sc0 = 2 * cb;
    } else {
        st0 = (2 * step_lo) / gpt;
        sc0 = (2 * step_lo) - st0 * gpt;
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(P.in + (size_t)n * P.IH * P.IW * P.Cip), 0, (int)((size_t)P.IH * P.IW * P.Cip * 4), 0x00020000);
    const int ih2 = 2 * P.IH - 2, iw2 = 2 * P.IW - 2;

    // ---- B operand: this wave's TN column tiles, fragment-major in HBM; scalar offsets, one 1 KiB load per fragment
    const int wm0 = (wave / WN) * TM * 32, wn0 = (wave % WN) * TN * 32;
    const int ks16_total = pKp >> 4;
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)pw, 0, (int)((size_t)P.w_rows * pKp * 4), 0x00020000);
    const int wb0 = (((n0 + wn0) >> 5) * ks16_total + 2 * step_lo) * 2048;  // bytes
    const int wbn = ks16_total * 2048;                                       // bytes between column tiles
    const int wlane = lane * 16;

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

    // (tap, channel group) of this thread's half for the NEXT CONV_LOAD_A, with the tap offsets already fetched from
    // LDS: the table read is issued one step ahead so that its latency never sits between a barrier and the loads
    int tdy, tdx, tcg;
#define CONV_NEXT_TAP()                                                                                                \
    {                                                                                                                  \
        int c1 = sc0 + 1, t1 = st0;                                                                                    \
        if (c1 >= gpt) {                                                                                               \
            c1 -= gpt;                                                                                                 \
            t1++;                                                                                                      \
        }                                                                                                              \
        int my_tap = ahalf ? t1 : st0;                                                                                 \
        if (cmajor && sc0 >= gpt) my_tap = ntaps_s; /* behind the last step: an always-outside tap slot */             \
        tcg = ahalf ? c1 : sc0;                                                                                        \
        tdy = s_dy[my_tap];                                                                                            \
        tdx = s_dx[my_tap];                                                                                            \
    }

    unsigned aoff;  // byte offset of this thread's 64 B of the next A tile (or the out-of-range marker)
#define CONV_ADDR_A()                                                                                                  \
    {                                                                                                                  \
        int iy = iy0 + tdy, ix = ix0 + tdx;                                                                            \
        if (P.pad_mode) { /* ReflectionPad2d: |v|, then mirrored at the far edge */                                    \
            iy = max(iy, -iy);                                                                                         \
            ix = max(ix, -ix);                                                                                         \
            iy = min(iy, ih2 - iy);                                                                                    \
            ix = min(ix, iw2 - ix);                                                                                    \
        }                                                                                                              \
        const bool ok = (unsigned)iy < (unsigned)P.IH && (unsigned)ix < (unsigned)P.IW;                                \
        aoff = ok ? (unsigned)(((iy * P.IW + ix) * P.Cip + tcg * 16) * 4) : 0x80000000u;                               \
        if (cmajor) {                                                                                                  \
            st0++;                                                                                                     \
            if (st0 >= ntaps_s) {                                                                                      \
                st0 = 0;                                                                                               \
                sc0 += 2;                                                                                              \
            }                                                                                                          \
        } else {                                                                                                       \
            sc0 += 2;                                                                                                  \
            if (sc0 >= gpt) {                                                                                          \
                sc0 -= gpt;                                                                                            \
                st0++;                                                                                                 \
            }                                                                                                          \
            if (sc0 >= gpt) {                                                                                          \
                sc0 -= gpt;                                                                                            \
                st0++;                                                                                                 \
            }                                                                                                          \
        }                                                                                                              \
        CONV_NEXT_TAP();                                                                                               \
    }
#define CONV_ISSUE_A(R)                                                                                                \
    {                                                                                                                  \
        R.v0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, aoff, 0, 0));                   \
        R.v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, aoff + 16, 0, 0));              \
        R.v2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, aoff + 32, 0, 0));              \
        R.v3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, aoff + 48, 0, 0));              \
    }
#define CONV_LOAD_A(R)                                                                                                 \
    {                                                                                                                  \
        CONV_ADDR_A();                                                                                                 \
        CONV_ISSUE_A(R);                                                                                               \
    }

#define CONV_LOAD_B(step, ks)                                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < TN; nt++) _Pragma("unroll") for (int pp = 0; pp < NPART; pp++)             \
        bfr[nt][ks][pp] = __builtin_bit_cast(                                                                          \
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wlane,                                               \
                                                          wb0 + nt * wbn + ((2 * (step) + (ks)) * 2 + pp) * 1024, 0));

    // Split of one bf16 pair in three stages of 2-3 VALU each, so that the MFMA gaps of a k16 half carry one stage each:
    //   0: pick the two floats (+ ReLU);  1: high parts (round to nearest even) and their fp32 images;
    //   2: low parts = bf16(x - high).
    float cf0[4], cf1[4], cb0[4], cb1[4];
    uint32_t hw[4], lw[4];
#define CONV_STAGE(sidx, x0, x1)                                                                                       \
    {                                                                                                                  \
        const int q_ = (sidx) / 3, sub_ = (sidx) % 3;                                                                  \
        if (sub_ == 0) {                                                                                               \
            cf0[q_] = q_ < 2 ? x0[2 * q_] : x1[2 * (q_ - 2)];                                                          \
            cf1[q_] = q_ < 2 ? x0[2 * q_ + 1] : x1[2 * (q_ - 2) + 1];                                                  \
            if constexpr (RELU) {                                                                                      \
                asm("v_max_f32 %0, 0, %1" : "=v"(cf0[q_]) : "v"(cf0[q_]));                                             \
                asm("v_max_f32 %0, 0, %1" : "=v"(cf1[q_]) : "v"(cf1[q_]));                                             \
            }                                                                                                          \
        } else if (sub_ == 1) {                                                                                        \
            const bf16x2 h_ = __builtin_convertvector(f32x2{cf0[q_], cf1[q_]}, bf16x2);                                \
            hw[q_] = __builtin_bit_cast(uint32_t, h_);                                                                 \
            cb0[q_] = __builtin_bit_cast(float, hw[q_] << 16);                                                         \
            cb1[q_] = __builtin_bit_cast(float, hw[q_] & 0xffff0000u);                                                 \
        } else if constexpr (NPART == 2) {                                                                             \
            const bf16x2 l_ = __builtin_convertvector(f32x2{cf0[q_], cf1[q_]} - f32x2{cb0[q_], cb1[q_]}, bf16x2);      \
            lw[q_] = __builtin_bit_cast(uint32_t, l_);                                                                 \
        }                                                                                                              \
    }

    constexpr int TILES = TM * TN;        // 32x32 MFMA tiles per wave
    constexpr int NPROD = NPART == 2 ? 3 : 1;  // MFMAs per tile and k16 half
    // A wave with a single 32x32 tile would issue its three products back to back into one accumulator (dependent MFMAs
    // stall the pipe); it accumulates each product in its own registers and adds them up before the epilogue.
    constexpr bool ONE_TILE = TILES == 1 && NPART == 2;
    static_assert(TILES == 1 || TILES == 2 || TILES == 4, "pair schedule");

    // k16 half `ks` of the tile at As.  With FILL, the split / store of half `ks` of the NEXT step's raw data (x0, x1)
    // is placed between the MFMA groups and pinned there (sched_barrier), so that it issues in the matrix pipe's shadow.
#define CONV_HALF(As, An, ks, x0, x1, FILL, HOOK0, HOOK1)                                                                            \
    {                                                                                                                  \
        bf16x8 af[NPART][TM];                                                                                          \
        _Pragma("unroll") for (int mt = 0; mt < TM; mt++)                                                              \
        {                                                                                                              \
            const int off = lds_row_g(wm0 + mt * 32 + fr) + (ks)*16 + fkq;                                               \
            af[0][mt] = *reinterpret_cast<const bf16x8*>((As) + off);                                                  \
            if constexpr (NPART == 2) af[NPART - 1][mt] = *reinterpret_cast<const bf16x8*>((As) + A_ELEMS + off);      \
        }                                                                                                              \
        _Pragma("unroll") for (int pp = 0; pp < NPROD; pp++) _Pragma("unroll") for (int mt = 0; mt < TM; mt++)         \
            _Pragma("unroll") for (int nt = 0; nt < TN; nt++)                                                          \
        {                                                                                                              \
            /* product order lo*hi, hi*lo, hi*hi; consecutive MFMAs never share an accumulator */                      \
            const int ia = (NPART == 2 && pp == 0) ? NPART - 1 : 0, ib = (NPART == 2 && pp == 1) ? NPART - 1 : 0;      \
            f32x16& dst = ONE_TILE ? accp[pp] : acc[mt][nt];                                                           \
            dst = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ia][mt], bfr[nt][ks][ib], dst, 0, 0, 0);                  \
            if (FILL) {                                                                                                \
                constexpr int NM = NPROD * TILES;                                                                      \
                const int i = (pp * TM + mt) * TN + nt;                                                                \
                _Pragma("unroll") for (int sg = 12 * i / NM; sg < 12 * (i + 1) / NM; sg++) CONV_STAGE(sg, x0, x1);     \
                if (i == 0) {                                                                                          \
                    HOOK0;                                                                                             \
                }                                                                                                      \
                if (i == (NM > 1 ? 1 : 0)) {                                                                           \
                    HOOK1;                                                                                             \
                }                                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
        }                                                                                                              \
        if (FILL) {                                                                                                    \
            *reinterpret_cast<uint4*>((An) + lds_row_g(arow) + ahalf * 16 + (ks)*8) = uint4{hw[0], hw[1], hw[2], hw[3]}; \
            if constexpr (NPART == 2)                                                                                  \
                *reinterpret_cast<uint4*>((An) + A_ELEMS + lds_row_g(arow) + ahalf * 16 + (ks)*8) =                      \
                    uint4{lw[0], lw[1], lw[2], lw[3]};                                                                 \
        }                                                                                                              \
    }

    // one pipeline step that has a successor: CUR holds raw step s + 1, NXT receives step s + 2
#define CONV_STEP(s, CUR, NXT)                                                                                         \
    {                                                                                                                  \
        __bf16* As = smem + ((s)&1) * A_BUF;                                                                           \
        __bf16* An = smem + (((s) + 1) & 1) * A_BUF;                                                                   \
        __syncthreads();                                                                                               \
        /* the gather of step s + 2 rides in the first two MFMA gaps of the step */                                    \
        CONV_HALF(As, An, 0, CUR.v0, CUR.v1, true, if ((s) + 2 < nsteps) CONV_ADDR_A(),                                \
                  if ((s) + 2 < nsteps) CONV_ISSUE_A(NXT));                                                            \
        CONV_HALF(As, An, 1, CUR.v2, CUR.v3, true, CONV_LOAD_B((s) + 1, 0), );                                         \
        CONV_LOAD_B((s) + 1, 1);                                                                                       \
    }

    f32x16 acc[TM][TN];
    f32x16 accp[3];  // ONE_TILE only
#pragma unroll
    for (int mt = 0; mt < TM; mt++)
#pragma unroll
        for (int nt = 0; nt < TN; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;
#pragma unroll
    for (int pp = 0; pp < 3; pp++)
#pragma unroll
        for (int r = 0; r < 16; r++) accp[pp][r] = 0.f;

    const int fr = lane & 31, fkq = (lane >> 5) * 8;
    // prologue: tile 0 -> LDS, raw tile 1 -> rb, B fragments of step 0
    CONV_NEXT_TAP();
    CONV_LOAD_A(ra);
    if (nsteps > 1) CONV_LOAD_A(rb);
    CONV_LOAD_B(0, 0);
    CONV_LOAD_B(0, 1);
    {
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int sg = 0; sg < 12; sg++) {
                if (h) {
                    CONV_STAGE(sg, ra.v2, ra.v3);
                } else {
                    CONV_STAGE(sg, ra.v0, ra.v1);
                }
            }
            *reinterpret_cast<uint4*>(smem + lds_row_g(arow) + ahalf * 16 + h * 8) = uint4{hw[0], hw[1], hw[2], hw[3]};
            if constexpr (NPART == 2)
                *reinterpret_cast<uint4*>(smem + A_ELEMS + lds_row_g(arow) + ahalf * 16 + h * 8) =
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
        CONV_HALF(As, As, 0, ra.v0, ra.v1, false, , );
        CONV_HALF(As, As, 1, ra.v2, ra.v3, false, , );
    }

    if constexpr (ONE_TILE) acc[0][0] = (accp[0] + accp[1]) + accp[2];

    // ---- epilogue
    if (P.ksplit > 1) {  // partial sums of this K slice
        const int col = lane & 31;
#pragma unroll
        for (int nt = 0; nt < TN; nt++) {
            const int co = n0 + wn0 + nt * 32 + col;
            if (co >= P.Cop) continue;
#pragma unroll
            for (int mt = 0; mt < TM; mt++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int o = s_outpix[wm0 + mt * 32 + mfma_row(r, lane)];
                    if (o < 0) continue;
                    if (P.partials)
                        P.partials[(size_t)zs * P.out_elems + (size_t)o * P.Cop + co] = acc[mt][nt][r];
                    else
                        unsafeAtomicAdd(P.out + (size_t)o * P.Cop + co, acc[mt][nt][r]);
                }
        }
        return;
    }
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

// Deterministic split K: out = (accumulate ? out : 0) + partials[0] + partials[1] + ... in slice order.
__global__ __launch_bounds__(256) void k_split_reduce(const float* __restrict__ partials, int ksplit, size_t elems4,
                                                      float* __restrict__ out, int accumulate)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems4) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (accumulate) v = reinterpret_cast<const f32x4*>(out)[i];
    for (int z = 0; z < ksplit; z++) v += reinterpret_cast<const f32x4*>(partials)[(size_t)z * elems4 + i];
    reinterpret_cast<f32x4*>(out)[i] = v;
}

// Second pass of a split-K launch, in place on out [N, HW, Cop]: v = out + bias; statistics of v; activation.
// grid (position chunks, N); a thread owns 4 channels and walks the chunk's positions.
__global__ __launch_bounds__(256) void k_bias_act_stats(float* __restrict__ out, const float* __restrict__ bias, int act,
                                                        double* __restrict__ stats, int HW, int Cop, int chunk)
{
    __shared__ float red[256 * 8];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int C4 = Cop >> 2;
    const int p0 = blockIdx.x * chunk, p1 = min(p0 + chunk, HW);
    for (int cq0 = 0; cq0 < C4; cq0 += 256) {
        const int TR = min(C4 - cq0, 256);  // threads per position
        const int RP = 256 / TR;            // positions in flight
        const int cq = cq0 + tid % TR, r = tid / TR;
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < RP) {
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (bias) b = *reinterpret_cast<const f32x4*>(bias + 4 * cq);
            for (int p = p0 + r; p < p1; p += RP) {
                f32x4* ptr = reinterpret_cast<f32x4*>(out + ((size_t)n * HW + p) * Cop) + cq;
                f32x4 v = *ptr + b;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    s1[e] += v[e];
                    s2[e] += v[e] * v[e];
                    if (act == 1)
                        v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                    else if (act == 2)
                        v[e] = tanhf(v[e]);
                }
                if (bias || act) *ptr = v;
            }
        }
        if (stats) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                red[tid * 8 + e] = s1[e];
                red[tid * 8 + 4 + e] = s2[e];
            }
            __syncthreads();
            if (tid < TR) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float a = 0.f, q = 0.f;
                    for (int rr = 0; rr < RP; rr++) {
                        a += red[(rr * TR + tid) * 8 + e];
                        q += red[(rr * TR + tid) * 8 + 4 + e];
                    }
                    const int co = 4 * (cq0 + tid) + e;
                    double* stp = stats + (((size_t)n * STAT_SLOTS + (blockIdx.x & (STAT_SLOTS - 1))) * Cop + co) * 2;
                    unsafeAtomicAdd(stp, (double)a);
                    unsafeAtomicAdd(stp + 1, (double)q);
                }
            }
            __syncthreads();
        }
    }
}

template <int WM, int WN, int TM, int TN>
static int launch_conv(ConvGemmParams P, int npart, hipStream_t st, void* workspace, size_t workspace_bytes)
{
    constexpr int BN = WN * TN * 32;
    const int Q = P.QH * P.QW;
    P.ntiles = (P.Cop + BN - 1) / BN;
    const int tiles = ((Q + 127) / 128) * P.N * P.ntiles;
    const int nsteps = P.Kp / CONV_BK;
    // split K when the output grid alone leaves most of the 256 CUs idle and the K loop is long enough to share out;
    // only for launches that own the whole output tensor (the zero fill must not touch other phases' results)
    const bool dense = P.ostride == 1 && P.py == 0 && P.px == 0 && P.QH == P.OH && P.QW == P.OW;
    int ksplit = 1;
    const bool fused_tail = P.bias || P.act || P.stats;
    // thresholds from a sweep on MI355X (160 / 512, 500 / 944 ... 2500 / 3072 tiles / target blocks): splitting grids
    // that already give every CU a tile only adds atomics
    constexpr int max_tiles = 160, target = 512;
    if (dense && tiles <= max_tiles && nsteps >= 16 && !(P.accumulate && fused_tail)) {
        ksplit = min(min((target + tiles - 1) / tiles, nsteps / 8), 32);
        if (ksplit < 2) ksplit = 1;
    }
    P.steps_per_split = (nsteps + ksplit - 1) / ksplit;
    P.ksplit = (nsteps + P.steps_per_split - 1) / P.steps_per_split;
    const bool split = P.ksplit > 1;
    const size_t out_elems = (size_t)P.N * P.OH * P.OW * P.Cop;
    P.out_elems = out_elems;
    P.partials = nullptr;
    if (split && workspace) {  // deterministic mode
        if (workspace_bytes < (size_t)P.ksplit * out_elems * sizeof(float))
            return fail(SDN_ENOMEM, "sdn_conv_gemm: workspace %zu < %zu bytes", workspace_bytes,
                        (size_t)P.ksplit * out_elems * sizeof(float));
        P.partials = (float*)workspace;
    }
    if (split && !P.accumulate && !P.partials) {
        const hipError_t e = hipMemsetAsync(P.out, 0, out_elems * sizeof(float), st);
        if (e != hipSuccess) return fail(SDN_ELAUNCH, "sdn_conv_gemm: memset: %s", hipGetErrorString(e));
    }
    const dim3 grid((unsigned)(tiles * P.ksplit));
    // algorithmic work of this launch: 2 * positions * taps * Cin(padded) * Cout(padded) flops
    TimedLaunch timed(TIME_CONV_GEMM, st, 2.0 * P.N * Q * (double)P.taps.n * P.Cip * P.Cop);
    if (npart == 2) {
        if (P.in_relu)
            hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 2, true>), grid, dim3(256), 0, st, P);
        else
            hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 2, false>), grid, dim3(256), 0, st, P);
    } else {
        if (P.in_relu)
            hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 1, true>), grid, dim3(256), 0, st, P);
        else
            hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 1, false>), grid, dim3(256), 0, st, P);
    }
    if (split && P.partials)
        hipLaunchKernelGGL(k_split_reduce, dim3(cdiv((long)(out_elems / 4), 256)), dim3(256), 0, st, P.partials, P.ksplit,
                           out_elems / 4, P.out, P.accumulate);
    if (split && fused_tail) {
        const int HW = P.OH * P.OW;
        const int chunks = max(1, min((HW + 63) / 64, (1024 + P.N - 1) / P.N));
        const int chunk = (HW + chunks - 1) / chunks;
        hipLaunchKernelGGL(k_bias_act_stats, dim3((unsigned)((HW + chunk - 1) / chunk), (unsigned)P.N), dim3(256), 0, st,
                           P.out, P.bias, P.act, P.stats, HW, P.Cop, chunk);
    }
    return check_launch("k_conv_gemm");
}

// every phase of P.ph in one launch: 8-aligned block ranges, longest K loop first (the phases arrive sorted)
template <int WM, int WN, int TM, int TN>
static int launch_conv_phases(ConvGemmParams P, int npart, hipStream_t st)
{
    constexpr int BN = WN * TN * 32;
    P.ntiles = (P.Cop + BN - 1) / BN;
    P.ksplit = 1;
    P.steps_per_split = 1 << 30;
    P.partials = nullptr;
    P.out_elems = (size_t)P.N * P.OH * P.OW * P.Cop;
    long first = 0;
    double work = 0.0;
    for (int k = 0; k < P.nphase; k++) {
        ConvPhase& F = P.ph[k];
        const long tiles = (long)(((long)F.QH * F.QW + 127) / 128) * P.N * P.ntiles;
        F.first = (int)first;
        F.nblocks = (int)((tiles + 7) & ~7L);
        first += F.nblocks;
        work += 2.0 * P.N * (double)F.QH * F.QW * F.ntaps * P.Cip * P.Cop;
    }
    if (first < 1 || first > 0x7fffffffL) return fail(SDN_EINVAL, "sdn_conv_gemm_phases: grid of %ld blocks", first);
    const dim3 grid((unsigned)first);
    TimedLaunch timed(TIME_CONV_GEMM, st, work);
    if (npart == 2) {
        if (P.in_relu)
            hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 2, true>), grid, dim3(256), 0, st, P);
        else
            hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 2, false>), grid, dim3(256), 0, st, P);
    } else {
        if (P.in_relu)
            hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 1, true>), grid, dim3(256), 0, st, P);
        else
            hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 1, false>), grid, dim3(256), 0, st, P);
    }
    return check_launch("k_conv_gemm (phases)");
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_conv_gemm_phases(const float* in, int N, int IH, int IW, int Cip, float* out, int OH, int OW, int Cop,
                                 int istride, int ostride, int nphase, const int32_t* QH, const int32_t* QW, const int32_t* py,
                                 const int32_t* px, const int32_t* ntaps, const int8_t* taps, int pad_mode, int in_relu,
                                 const void* const* w_packed, const int32_t* Kp, int w_rows, const float* bias, int act,
                                 double* stats, int accumulate, int precision, sdnStream stream)
{
    if (!in || !out || !QH || !QW || !py || !px || !ntaps || !taps || !w_packed || !Kp)
        return fail(SDN_EINVAL, "sdn_conv_gemm_phases: null pointer");
    if (nphase < 1 || nphase > CONV_MAX_PHASES) return fail(SDN_EINVAL, "sdn_conv_gemm_phases: %d phases not in 1..%d", nphase, CONV_MAX_PHASES);
    if ((Cip & 15) || (Cop & 15)) return fail(SDN_EINVAL, "sdn_conv_gemm_phases: channel counts must be padded to 16 (%d, %d)", Cip, Cop);
    if (precision != 1 && precision != 3) return fail(SDN_EINVAL, "sdn_conv_gemm_phases: precision must be 1 (bf16) or 3 (bf16x3)");
    if (N < 1 || istride < 1 || ostride < 1) return fail(SDN_EINVAL, "sdn_conv_gemm_phases: bad geometry");
    if ((size_t)IH * IW * Cip * 4 >= 0x7fffff00u) return fail(SDN_EINVAL, "sdn_conv_gemm_phases: one input image must stay below 2 GiB");
    ConvGemmParams P;
    P.in = in; P.out = out; P.w = nullptr; P.bias = bias; P.stats = stats;
    P.N = N; P.IH = IH; P.IW = IW; P.Cip = Cip; P.OH = OH; P.OW = OW; P.Cop = Cop;
    P.QH = P.QW = 1; P.istride = istride; P.ostride = ostride; P.py = P.px = 0; P.Kp = CONV_BK;
    P.pad_mode = pad_mode; P.in_relu = in_relu; P.act = act; P.accumulate = accumulate; P.w_rows = w_rows;
    P.taps.n = 0;
    // phases in descending K order (the longest workgroups are dispatched first); input order breaks ties
    int order[CONV_MAX_PHASES];
    for (int k = 0; k < nphase; k++) order[k] = k;
    for (int a = 1; a < nphase; a++)
        for (int b = a; b > 0 && Kp[order[b]] > Kp[order[b - 1]]; b--) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
    int toff[CONV_MAX_PHASES], acc = 0;
    for (int k = 0; k < nphase; k++) {
        toff[k] = acc;
        if (ntaps[k] < 1 || ntaps[k] > CONV_PHASE_TAPS) return fail(SDN_EINVAL, "sdn_conv_gemm_phases: phase %d has %d taps (1..%d)", k, ntaps[k], CONV_PHASE_TAPS);
        acc += 2 * ntaps[k];
    }
    P.nphase = nphase;
    for (int s = 0; s < nphase; s++) {
        const int k = order[s];
        ConvPhase& F = P.ph[s];
        if (!w_packed[k]) return fail(SDN_EINVAL, "sdn_conv_gemm_phases: phase %d has no weights", k);
        if (QH[k] < 1 || QW[k] < 1 || py[k] < 0 || px[k] < 0 || (QH[k] - 1) * ostride + py[k] >= OH || (QW[k] - 1) * ostride + px[k] >= OW)
            return fail(SDN_EINVAL, "sdn_conv_gemm_phases: phase %d: output grid exceeds the output tensor", k);
        if (Kp[k] % CONV_BK || Kp[k] < ntaps[k] * Cip) return fail(SDN_EINVAL, "sdn_conv_gemm_phases: phase %d: Kp %d does not cover %d taps x %d", k, Kp[k], ntaps[k], Cip);
        if ((size_t)w_rows * Kp[k] * 4 >= 0x7fffff00u) return fail(SDN_EINVAL, "sdn_conv_gemm_phases: packed weights must stay below 2 GiB");
        F.w = (const __bf16*)w_packed[k];
        F.QH = QH[k]; F.QW = QW[k]; F.py = py[k]; F.px = px[k]; F.Kp = Kp[k]; F.ntaps = ntaps[k];
        for (int t = 0; t < CONV_PHASE_TAPS; t++) {
            F.dy[t] = t < ntaps[k] ? taps[toff[k] + t] : 0;
            F.dx[t] = t < ntaps[k] ? taps[toff[k] + ntaps[k] + t] : 0;
        }
    }
    for (int s = nphase; s < CONV_MAX_PHASES; s++) P.ph[s] = P.ph[0];
    const int npart = precision == 3 ? 2 : 1;
    hipStream_t st = (hipStream_t)stream;
    if (Cop > 64) {
        if (w_rows < ((Cop + 127) / 128) * 128) return fail(SDN_EINVAL, "sdn_conv_gemm_phases: weight rows %d < padded Cout", w_rows);
        return launch_conv_phases<2, 2, 2, 2>(P, npart, st);
    }
    if (Cop > 32) {
        if (w_rows < 64) return fail(SDN_EINVAL, "sdn_conv_gemm_phases: weight rows %d < 64", w_rows);
        return launch_conv_phases<2, 2, 2, 1>(P, npart, st);
    }
    if (w_rows < 32) return fail(SDN_EINVAL, "sdn_conv_gemm_phases: weight rows %d < 32", w_rows);
    return launch_conv_phases<4, 1, 1, 1>(P, npart, st);
}

SDN_API int sdn_conv_gemm_workspace_bytes(int N, int OH, int OW, int Cop, size_t* out)
{
    if (N < 1 || OH < 1 || OW < 1 || Cop < 1 || !out) return fail(SDN_EINVAL, "sdn_conv_gemm_workspace_bytes: bad arguments");
    // at most 32 K slices, and only launches of <= 160 tiles of 128 positions x <= 128 channels are split
    const size_t elems = (size_t)N * OH * OW * Cop;
    const size_t cap = (size_t)160 * 128 * 128 * 2;
    *out = 32 * (elems < cap ? elems : cap) * sizeof(float);
    return SDN_OK;
}

SDN_API int sdn_conv_gemm(const float* in, int N, int IH, int IW, int Cip, float* out, int OH, int OW, int Cop, int QH,
                          int QW, int istride, int ostride, int py, int px, int ntaps, const int8_t* dy,
                          const int8_t* dx, int pad_mode, int in_relu, const void* w_packed, int Kp,
                          int w_rows, const float* bias, int act, double* stats, int accumulate, int precision,
                          void* workspace, size_t workspace_bytes, sdnStream stream)
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
    P.pad_mode = pad_mode; P.in_relu = in_relu; P.act = act; P.accumulate = accumulate; P.w_rows = w_rows;
    // the gather addresses one image, and the weight fetch the packed matrix, through 32-bit buffer offsets
    if ((size_t)IH * IW * Cip * 4 >= 0x7fffff00u) return fail(SDN_EINVAL, "sdn_conv_gemm: one input image must stay below 2 GiB");
    if ((size_t)w_rows * Kp * 4 >= 0x7fffff00u) return fail(SDN_EINVAL, "sdn_conv_gemm: packed weights must stay below 2 GiB");
    P.nphase = 0;
    P.taps.n = ntaps;
    for (int t = 0; t < ntaps; t++) {
        P.taps.dy[t] = dy[t];
        P.taps.dx[t] = dx[t];
    }
    const int npart = precision == 3 ? 2 : 1;
    hipStream_t st = (hipStream_t)stream;
    // the weight matrix must hold a whole number of N tiles
    if (Cop > 64) {
        if (w_rows < ((Cop + 127) / 128) * 128) return fail(SDN_EINVAL, "sdn_conv_gemm: weight rows %d < padded Cout", w_rows);
        return launch_conv<2, 2, 2, 2>(P, npart, st, workspace, workspace_bytes);
    }
    if (Cop > 32) {
        if (w_rows < 64) return fail(SDN_EINVAL, "sdn_conv_gemm: weight rows %d < 64", w_rows);
        return launch_conv<2, 2, 2, 1>(P, npart, st, workspace, workspace_bytes);
    }
    if (w_rows < 32) return fail(SDN_EINVAL, "sdn_conv_gemm: weight rows %d < 32", w_rows);
    return launch_conv<4, 1, 1, 1>(P, npart, st, workspace, workspace_bytes);
}
