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
// input channels, walked in 32-channel steps.  Neither operand is touched by a VALU instruction on its way to the MFMA:
//   A (activations) arrives PRE-SPLIT: sdn_split_planes stored every tensor once as two bf16 planes (hi, lo; ReLU
//     already applied where the consumer would have applied it), and the kernel gathers 16-byte pieces of them straight
//     into LDS with global_load_lds (LDS-DMA: no staging registers, no ds_write); the LDS image is 64-B rows with the
//     16-B chunks XOR-swizzled by (row >> 2) & 3 -- the swizzle is applied on the per-lane SOURCE address, the DMA
//     destination is lane-linear -- so that the ds_read_b128 fragment reads are bank-conflict free; out-of-range
//     positions (zero padding, tile tails) read a zero page;
//   B (weights) is pre-split by sdn_conv_pack_weights into K-major rows with hi / lo interleaved per 32-deep step and
//     takes the same LDS-DMA path (same swizzle, rows = output channels).
// Because every load of the main loop is an LDS-DMA, nothing forces the compiler to drain the memory pipe: the loop waits
// with ONE hand-placed `s_waitcnt vmcnt(0)` per step for DMAs that were issued a whole step earlier, followed by a raw
// s_barrier.  (hipcc turns every wait into vmcnt(0) as soon as ordinary register loads and LDS-DMAs are mixed in a loop,
// and drains before __syncthreads(): the register-fragment version of B measured no faster than the staged one.)
// Double-buffered stages of 32 KB (A hi/lo + B hi/lo), 2 workgroups per CU, XCD-aware tile order.
// Epilogue: bias, LeakyReLU / tanh, InstanceNorm statistics (per (n, c) sum and sum of squares, block-reduced, fp64
// atomics into 8 slots) and coalesced 128-B channel-contiguous stores.
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
    const __bf16* in;  // planes [2 (hi, lo)][N, IH, IW, Cip] bf16, written by sdn_split_planes
    const __bf16* zero_page;  // >= 64 B of zeros (device)
    long plane_stride;  // elements between the hi and the lo plane
    float* out;        // [N, OH, OW, Cop]
    const __bf16* w;     // [Corows][Kp / 32][2 (hi, lo)][32]  K-major rows, hi / lo interleaved per 32-deep step
    const float* bias;   // [>= Cop] or null
    double* stats;       // [N, STAT_SLOTS, Cop, 2] or null
    int N, IH, IW, Cip;
    int OH, OW, Cop;
    int QH, QW, istride, ostride, py, px;
    int Kp;
    int pad_mode, act, accumulate;
    int ntiles;  // output-channel tiles (set by the launcher)
    ConvTaps taps;
};

// NSTAGE LDS stages: DMAs run NSTAGE - 1 steps ahead.  What bounds this kernel is Little's law on the L2 -> LDS path
// (PMC: 47 % of the wave cycles are spent in s_waitcnt / barrier, latency under load ~3500 cycles): throughput = bytes in
// flight per CU / latency.  The 256 x 128 tile (8 waves) needs 24 KB of operands per 128x128x32 unit of MFMA work instead
// of 32 KB, and with 3 stages keeps 96 KB in flight per CU instead of 64 KB.
template <int WM, int WN, int TM, int TN, int NPART, int NSTAGE>
__global__ __launch_bounds__(WM * WN * 64) void k_conv_gemm(const ConvGemmParams P)
{
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NWAVES = WM * WN;
    static_assert(BM == NWAVES * 32, "every wave stages 32 rows of the A tile");
    static_assert(NSTAGE == 2 || (BN / 16) % NWAVES == 0, "counted waits need the same DMA count in every wave");
    constexpr int DMA_PER_STEP = NPART * (2 + (BN / 16 + NWAVES - 1) / NWAVES);  // wave-instructions per wave and step
    constexpr int A_ELEMS = BM * CONV_BK;   // one plane of the A tile: 128 rows x 64 B, chunks XOR-swizzled
    constexpr int B_ELEMS = BN * CONV_BK;   // one plane of the B tile
    constexpr int A_BUF = NPART * (A_ELEMS + B_ELEMS);  // one stage: A hi (+ lo), B hi (+ lo)
    __shared__ __attribute__((aligned(16))) __bf16 smem[NSTAGE * A_BUF];
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

    // ---- A staging by LDS-DMA.  One wave-instruction moves 64 lanes x 16 B = 16 rows x 64 B (one plane, one 32-channel
    // step).  Wave w owns rows [32 w, 32 w + 32): lane l serves rows 32 w + (l >> 2) and + 16, physical chunk l & 3.
    // The LOGICAL chunk it must fetch for that slot is c = (l & 3) ^ ((row >> 2) & 3) -- the same for both of its rows
    // and for every step, so each lane walks a fixed 8-channel column of the step: group half c >> 1, offset (c & 1) * 8.
    const int rloc = lane >> 2;
    const int row_a = wave * 32 + rloc, row_b = row_a + 16;
    const int chunk = (lane & 3) ^ ((row_a >> 2) & 3);
    const int khalf = chunk >> 1, koff = (chunk & 1) * 8;
    const int qa = m0 + row_a, qb = m0 + row_b;
    const bool oka = qa < Q, okb = qb < Q;
    const int qya = oka ? qa / P.QW : 0, qxa = oka ? qa - qya * P.QW : 0;
    const int qyb = okb ? qb / P.QW : 0, qxb = okb ? qb - qyb * P.QW : 0;
    const int iya0 = qya * P.istride, ixa0 = qxa * P.istride;
    const int iyb0 = qyb * P.istride, ixb0 = qxb * P.istride;
    const int gpt = P.Cip >> 4;  // 16-channel groups per tap
    int a_tap = 0, a_cg = khalf;  // group index g = 2 * step + khalf, kept as (tap, group in tap)
    while (a_cg >= gpt) {
        a_cg -= gpt;
        a_tap++;
    }
    const __bf16* in_n = P.in + (size_t)n * P.IH * P.IW * P.Cip;
    const int last_tap = P.taps.n - 1;

    // ---- B staging: 16-row blocks of the weight tile, block j by wave j % 4; same lane -> (row, chunk) rule as A
    const int wm0 = (wave / WN) * TM * 32, wn0 = (wave % WN) * TN * 32;
    const int nsteps = P.Kp / CONV_BK;
    const __bf16* wlane = P.w + ((size_t)(n0 + rloc) * nsteps) * (2 * CONV_BK) + chunk * 8;  // row n0 + rloc, step 0, hi

    __syncthreads();  // tap table visible

    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void glb_void;

    // issue the DMAs of one step into stage `buf` (rows a / b, planes hi / lo: 2 or 4 wave-instructions)
#define CONV_STAGE(buf, bstep)                                                                                         \
    {                                                                                                                  \
        const int tap = min(a_tap, last_tap);                                                                          \
        const bool tap_ok = a_tap <= last_tap;                                                                         \
        const int dyv = s_dy[tap], dxv = s_dx[tap];                                                                    \
        int iy = iya0 + dyv, ix = ixa0 + dxv;                                                                          \
        bool va = oka && tap_ok;                                                                                       \
        const bool vay = resolve_coord(iy, P.IH, P.pad_mode), vax = resolve_coord(ix, P.IW, P.pad_mode);               \
        va = va && vay && vax;                                                                                         \
        const __bf16* ga = va ? in_n + ((size_t)iy * P.IW + ix) * P.Cip + a_cg * 16 + koff : P.zero_page;              \
        int jy = iyb0 + dyv, jx = ixb0 + dxv;                                                                          \
        bool vb = okb && tap_ok;                                                                                       \
        const bool vby = resolve_coord(jy, P.IH, P.pad_mode), vbx = resolve_coord(jx, P.IW, P.pad_mode);               \
        vb = vb && vby && vbx;                                                                                         \
        const __bf16* gb = vb ? in_n + ((size_t)jy * P.IW + jx) * P.Cip + a_cg * 16 + koff : P.zero_page;              \
        __bf16* dst = smem + (buf)*A_BUF + wave * 32 * 32;                                                             \
        __builtin_amdgcn_global_load_lds((glb_void*)ga, (lds_void*)dst, 16, 0, 0);                                     \
        __builtin_amdgcn_global_load_lds((glb_void*)gb, (lds_void*)(dst + 16 * 32), 16, 0, 0);                         \
        if constexpr (NPART == 2) {                                                                                    \
            const __bf16* la = va ? ga + P.plane_stride : P.zero_page;                                                 \
            const __bf16* lb = vb ? gb + P.plane_stride : P.zero_page;                                                 \
            __builtin_amdgcn_global_load_lds((glb_void*)la, (lds_void*)(dst + A_ELEMS), 16, 0, 0);                     \
            __builtin_amdgcn_global_load_lds((glb_void*)lb, (lds_void*)(dst + A_ELEMS + 16 * 32), 16, 0, 0);           \
        }                                                                                                              \
        _Pragma("unroll") for (int jb = 0; jb < BN / 16; jb += NWAVES)                                                 \
        {                                                                                                              \
            const int j = jb + wave;                                                                                   \
            if (j < BN / 16) {                                                                                         \
                const __bf16* gw = wlane + ((size_t)(16 * j) * nsteps + (bstep)) * (2 * CONV_BK);                      \
                __bf16* db = smem + (buf)*A_BUF + NPART * A_ELEMS + j * 16 * 32;                                       \
                __builtin_amdgcn_global_load_lds((glb_void*)gw, (lds_void*)db, 16, 0, 0);                              \
                if constexpr (NPART == 2)                                                                              \
                    __builtin_amdgcn_global_load_lds((glb_void*)(gw + CONV_BK), (lds_void*)(db + B_ELEMS), 16, 0, 0);  \
            }                                                                                                          \
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

    // MFMAs of k16 half `ks` of the tile at As (swizzled 64-B rows)
#define CONV_MFMA_HALF(As, ks)                                                                                         \
    {                                                                                                                  \
        bf16x8 af[NPART][TM];                                                                                          \
        _Pragma("unroll") for (int mt = 0; mt < TM; mt++)                                                              \
        {                                                                                                              \
            const int row = wm0 + mt * 32 + fr;                                                                        \
            const int off = row * 32 + (((2 * (ks) + fkh) ^ ((row >> 2) & 3)) << 3);                                   \
            af[0][mt] = *reinterpret_cast<const bf16x8*>((As) + off);                                                  \
            if constexpr (NPART == 2) af[NPART - 1][mt] = *reinterpret_cast<const bf16x8*>((As) + A_ELEMS + off);      \
        }                                                                                                              \
        bf16x8 bf[NPART][TN];                                                                                          \
        _Pragma("unroll") for (int nt = 0; nt < TN; nt++)                                                              \
        {                                                                                                              \
            const int row = wn0 + nt * 32 + fr;                                                                        \
            const int off = row * 32 + (((2 * (ks) + fkh) ^ ((row >> 2) & 3)) << 3);                                   \
            bf[0][nt] = *reinterpret_cast<const bf16x8*>((As) + NPART * A_ELEMS + off);                                \
            if constexpr (NPART == 2)                                                                                  \
                bf[NPART - 1][nt] = *reinterpret_cast<const bf16x8*>((As) + NPART * A_ELEMS + B_ELEMS + off);          \
        }                                                                                                              \
        _Pragma("unroll") for (int mt = 0; mt < TM; mt++) _Pragma("unroll") for (int nt = 0; nt < TN; nt++)            \
        {                                                                                                              \
            if constexpr (NPART == 2) {                                                                                \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[NPART - 1][mt], bf[0][nt], acc[mt][nt], 0, 0, \
                                                                      0);                                              \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][mt], bf[NPART - 1][nt], acc[mt][nt], 0, 0, \
                                                                      0);                                              \
            }                                                                                                          \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][mt], bf[0][nt], acc[mt][nt], 0, 0, 0);         \
        }                                                                                                              \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mt = 0; mt < TM; mt++)
#pragma unroll
        for (int nt = 0; nt < TN; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;

    const int fr = lane & 31, fkh = lane >> 5;
    // prologue: NSTAGE - 1 steps in flight
#pragma unroll
    for (int pre = 0; pre < NSTAGE - 1; pre++)
        if (pre < nsteps) CONV_STAGE(pre, pre);
    int stage = 0;  // step % NSTAGE
    for (int step = 0; step < nsteps; step++) {
        const __bf16* As = smem + stage * A_BUF;
        // This wave's DMAs of step `step` have landed (those of the following NSTAGE - 2 steps may still fly); past the
        // barrier everybody's have, and every wave has finished reading the stage that is refilled next (its MFMAs of
        // step - 1 precede this point in program order).  Inline asm on purpose: __syncthreads() would make hipcc
        // drain the DMAs just issued below as well.
        if (NSTAGE == 3 && step + 1 < nsteps) {
            static_assert(DMA_PER_STEP == 6 || DMA_PER_STEP == 3 || NSTAGE == 2, "vmcnt immediate");
            if constexpr (DMA_PER_STEP == 6)
                asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const int nxt = step + NSTAGE - 1;
        if (nxt < nsteps) {
            const int sn = stage == 0 ? NSTAGE - 1 : stage - 1;  // (step + NSTAGE - 1) % NSTAGE
            CONV_STAGE(sn, nxt);
        }
        CONV_MFMA_HALF(As, 0);
        CONV_MFMA_HALF(As, 1);
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }
    __syncthreads();  // all tiles consumed (the statistics reuse the LDS)
#undef CONV_STAGE
#undef CONV_MFMA_HALF

    // ---- epilogue
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

template <int WM, int WN, int TM, int TN, int NSTAGE>
static int launch_conv(ConvGemmParams P, int npart, hipStream_t st)
{
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, THREADS = WM * WN * 64;
    const int Q = P.QH * P.QW;
    P.ntiles = (P.Cop + BN - 1) / BN;
    const dim3 grid((unsigned)(((Q + BM - 1) / BM) * P.N * P.ntiles));
    // algorithmic work of this launch: 2 * positions * taps * Cin(padded) * Cout(padded) flops
    TimedLaunch timed(TIME_CONV_GEMM, st, 2.0 * P.N * Q * (double)P.taps.n * P.Cip * P.Cop);
    if (npart == 2)
        hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 2, NSTAGE>), grid, dim3(THREADS), 0, st, P);
    else
        hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 1, NSTAGE>), grid, dim3(THREADS), 0, st, P);
    return check_launch("k_conv_gemm");
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_conv_gemm(const void* in_planes, long plane_stride, const void* zero_page, int N, int IH, int IW, int Cip, float* out, int OH, int OW, int Cop, int QH,
                          int QW, int istride, int ostride, int py, int px, int ntaps, const int8_t* dy,
                          const int8_t* dx, int pad_mode, const void* w_packed, int Kp,
                          int w_rows, const float* bias, int act, double* stats, int accumulate, int precision,
                          sdnStream stream)
{
    if (!in_planes || !zero_page || !out || !w_packed || !dy || !dx) return fail(SDN_EINVAL, "sdn_conv_gemm: null pointer");
    if (plane_stride < (long)N * IH * IW * Cip) return fail(SDN_EINVAL, "sdn_conv_gemm: plane stride %ld too small", plane_stride);
    if (ntaps < 1 || ntaps > CONV_MAX_TAPS) return fail(SDN_EINVAL, "sdn_conv_gemm: ntaps %d not in 1..%d", ntaps, CONV_MAX_TAPS);
    if ((Cip & 15) || (Cop & 15)) return fail(SDN_EINVAL, "sdn_conv_gemm: channel counts must be padded to 16 (%d, %d)", Cip, Cop);
    if (Kp % CONV_BK || Kp < ntaps * Cip) return fail(SDN_EINVAL, "sdn_conv_gemm: Kp %d does not cover %d taps x %d", Kp, ntaps, Cip);
    if (precision != 1 && precision != 3) return fail(SDN_EINVAL, "sdn_conv_gemm: precision must be 1 (bf16) or 3 (bf16x3)");
    if (N < 1 || QH < 1 || QW < 1 || istride < 1 || ostride < 1) return fail(SDN_EINVAL, "sdn_conv_gemm: bad geometry");
    if ((QH - 1) * ostride + py >= OH || (QW - 1) * ostride + px >= OW) return fail(SDN_EINVAL, "sdn_conv_gemm: output grid exceeds the output tensor");
    ConvGemmParams P;
    P.in = (const __bf16*)in_planes; P.zero_page = (const __bf16*)zero_page; P.plane_stride = plane_stride; P.out = out; P.w = (const __bf16*)w_packed; P.bias = bias; P.stats = stats;
    P.N = N; P.IH = IH; P.IW = IW; P.Cip = Cip; P.OH = OH; P.OW = OW; P.Cop = Cop;
    P.QH = QH; P.QW = QW; P.istride = istride; P.ostride = ostride; P.py = py; P.px = px; P.Kp = Kp;
    P.pad_mode = pad_mode; P.act = act; P.accumulate = accumulate;
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
        // many positions: 256 x 128 tiles, 8 waves, 3 stages; few: 128 x 128 tiles, 4 waves, 2 stages
        if ((long)QH * QW >= 512) return launch_conv<4, 2, 2, 2, 3>(P, npart, st);
        return launch_conv<2, 2, 2, 2, 2>(P, npart, st);
    }
    if (Cop > 32) {
        if (w_rows < 64) return fail(SDN_EINVAL, "sdn_conv_gemm: weight rows %d < 64", w_rows);
        return launch_conv<2, 2, 2, 1, 2>(P, npart, st);
    }
    if (w_rows < 32) return fail(SDN_EINVAL, "sdn_conv_gemm: weight rows %d < 32", w_rows);
    return launch_conv<4, 1, 1, 1, 2>(P, npart, st);
}
