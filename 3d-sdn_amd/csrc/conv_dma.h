// LDS-DMA pieces shared by the tiled MFMA kernels (conv_tile.hip, conv_wtile.hip).
//
// `global_load_lds_dwordx4` copies 16 B per lane from a per-lane global address to LDS at M0 + lane * 16 (a wave
// instruction fills 1 KiB, lane-linear).  It is issued from inline asm on purpose: hipcc then neither counts it in its
// own s_waitcnt bookkeeping nor treats it as a pending LDS store -- with the builtin, ROCm 7.2 put an `s_waitcnt vmcnt(0)`
// in front of the `ds_read_b64_tr_b16` fragment reads of every K step (the whole pipeline drained once per step).  The
// kernels wait for their copies themselves: counted `s_waitcnt vmcnt(N)` + raw `s_barrier`, then the fragment reads
// (guide: cdna_hip_programming.md section 5 "Pipelining across barriers", section 5.7 "LDS-DMA recipe").
#pragma once
#include <hip/hip_runtime.h>

namespace sdn {

typedef __attribute__((address_space(3))) char lds_char;

// LDS byte address of a pointer into a __shared__ array (wave-uniform when the pointer is)
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long)(lds_char*)p; }

// one LDS-DMA wave instruction: lane l copies 16 B from gsrc to LDS byte (lds_dst + 16 l); M0 is saved and restored
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// ok ? a : b per lane as two v_cndmask, from inline asm on purpose: written as a C++ select, hipcc turned the expensive arm (a
// 64-bit address computation) into a divergent branch around it, which cut the K-step loop into several basic blocks and
// with it the pinned MFMA / copy interleave
__device__ __forceinline__ const char* select_ptr(bool ok, const char* a, const char* b)
{
    const unsigned long m = __builtin_amdgcn_ballot_w64(ok);
    const unsigned long ua = (unsigned long)a, ub = (unsigned long)b;
    unsigned lo, hi;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"((unsigned)ub), "v"((unsigned)ua), "s"(m));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"((unsigned)(ub >> 32)), "v"((unsigned)(ua >> 32)), "s"(m));
    return (const char*)(((unsigned long)hi << 32) | lo);
}

}  // namespace sdn
