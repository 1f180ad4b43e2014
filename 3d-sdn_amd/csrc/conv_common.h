// Shared device pieces of the textural conv kernels (conv_gemm.hip, conv_wgrad.hip): bf16 split, LDS tile geometry and
// the MFMA inner product of two K-major LDS tiles.
//
// Numerics.  The textural networks (reference: textural/models/networks.py) are fp32 modules and the parity gate is 1e-3
// relative on activations through ~37 conv + InstanceNorm layers, plus a backward pass whose gradients are far below
// fp16's range.  gfx950 has no TF32-like MFMA, and the exact f32 MFMA runs at 1/16 of the bf16 rate, so the contraction
// runs on v_mfma_f32_32x32x16_bf16 with every fp32 operand split into two bf16 terms, x = hi + lo (|lo| <= 2^-9 |x|),
// and three products accumulated in fp32:  hi*hi + hi*lo + lo*hi.  The dropped lo*lo term and lo's own rounding are
// ~2^-17 relative, i.e. fp32-class results at 1/3 of the bf16 MFMA peak (~830 TFLOP/s, 5x the f32 MFMA/VALU peak).
// NPART = 1 keeps only hi*hi (plain bf16, 3x faster, ~2^-9 per operand): selectable, never the default.
#pragma once
#include <hip/hip_runtime.h>

namespace sdn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int CONV_BK = 32;       // K elements per main-loop step (two k16 MFMA sub-steps)
constexpr int CONV_MAX_TAPS = 64;
// InstanceNorm statistics are accumulated with fp64 atomics into STAT_SLOTS interleaved copies ([N][slot][C][2], slot =
// output tile index mod STAT_SLOTS) so that the thousands of tiles of a full-resolution layer do not serialise on one
// address per channel; readers add the slots up.
constexpr int STAT_SLOTS = 8;

// LDS tiles are [row][k] bf16 with 32 k per row.  Row pitch 40 bf16 = 80 B: 16-B aligned for ds_read_b128 /
// ds_write_b128 and, because 5 (the pitch in 16-B slots) is odd, the 16 rows a b128 lane group touches fall on 16
// different slots of the 256-B bank row.  Every 16 rows another 64 B is skipped so that the transposing 4-byte stores
// of the wgrad kernel (lanes = 16 k-pairs x 4 channel groups) spread over both halves of the banks.
constexpr int LDS_PITCH = 40;                                                            // bf16 per row
__device__ __host__ constexpr int lds_row(int r) { return r * LDS_PITCH + (r >> 4) * 32; }  // in bf16 elements
__device__ __host__ constexpr int lds_tile_elems(int rows) { return rows * LDS_PITCH + (rows >> 4) * 32; }

struct SplitBf16 {
    bf16x2 hi, lo;
};

// two fp32 -> packed (hi, hi) and (lo, lo).  Five VALU per pair: one v_cvt_pk_bf16_f32 (round to nearest even) for the
// high parts, a shift and a mask to read them back as fp32, one packed subtract, one v_cvt_pk_bf16_f32 for the low parts.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ SplitBf16 split2(float a, float b)
{
    SplitBf16 s;
    s.hi = __builtin_convertvector(f32x2{a, b}, bf16x2);
    const unsigned hp = __builtin_bit_cast(unsigned, s.hi);
    const f32x2 back = {__builtin_bit_cast(float, hp << 16), __builtin_bit_cast(float, hp & 0xffff0000u)};
    s.lo = __builtin_convertvector(f32x2{a, b} - back, bf16x2);
    return s;
}

// acc[TM][TN] += A(32*TM rows) x B(32*TN rows)^T over one 32-deep K step.  A*/B* point at the wave's first row of the
// hi tiles; the lo tiles follow at a_lo_off / b_lo_off elements.  Fragment layout of v_mfma_f32_32x32x16_bf16: lane l
// holds row (l & 31), k = 8 * (l >> 5) .. +7 of its 32 x 16 operand block -- the same rule for A and B, so any
// consistent k order inside the LDS rows is valid.
// NACC = 3: one accumulator set per product (for waves with a single tile), summed by the caller; else NACC = 1.
template <int TM, int TN, int NPART, int NACC>
__device__ __forceinline__ void mfma_step(const __bf16* __restrict__ As, const __bf16* __restrict__ Bs, int a_row0,
                                          int b_row0, int a_lo_off, int b_lo_off, int lane, f32x16 (&acc)[NACC][TM][TN])
{
    const int r = lane & 31, kq = (lane >> 5) * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        bf16x8 a[NPART][TM], b[NPART][TN];
#pragma unroll
        for (int mt = 0; mt < TM; mt++) {
            const int off = lds_row(a_row0 + mt * 32 + r) + ks * 16 + kq;
            a[0][mt] = *reinterpret_cast<const bf16x8*>(As + off);
            if constexpr (NPART == 2) a[NPART - 1][mt] = *reinterpret_cast<const bf16x8*>(As + a_lo_off + off);
        }
#pragma unroll
        for (int nt = 0; nt < TN; nt++) {
            const int off = lds_row(b_row0 + nt * 32 + r) + ks * 16 + kq;
            b[0][nt] = *reinterpret_cast<const bf16x8*>(Bs + off);
            if constexpr (NPART == 2) b[NPART - 1][nt] = *reinterpret_cast<const bf16x8*>(Bs + b_lo_off + off);
        }
        // product order lo*hi, hi*lo, hi*hi, each over all tiles: consecutive MFMAs never share an accumulator (a
        // dependent MFMA right behind its producer stalls the matrix pipe)
#pragma unroll
        for (int pp = 0; pp < (NPART == 2 ? 3 : 1); pp++)
#pragma unroll
            for (int mt = 0; mt < TM; mt++)
#pragma unroll
                for (int nt = 0; nt < TN; nt++) {
                    const int ia = (NPART == 2 && pp == 0) ? NPART - 1 : 0, ib = (NPART == 2 && pp == 1) ? NPART - 1 : 0;
                    f32x16& dst = NACC == 3 ? acc[pp][mt][nt] : acc[0][mt][nt];
                    dst = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia][mt], b[ib][nt], dst, 0, 0, 0);
                }
    }
}

// row of accumulator register `reg` of a 32x32 MFMA tile for this lane (column = lane & 31)
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// reflect (no edge repeat, torch ReflectionPad2d) or report out of range
__device__ __forceinline__ bool resolve_coord(int& v, int n, int pad_mode)
{
    if (pad_mode == 1) {
        if (v < 0) v = -v;
        if (v >= n) v = 2 * n - 2 - v;
        return v >= 0 && v < n;
    }
    return v >= 0 && v < n;
}

}  // namespace sdn
