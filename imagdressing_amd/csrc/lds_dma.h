// LDS-DMA (buffer_load ... lds) helpers shared by the kernels that stream operand tiles straight from global memory into
// LDS without a VGPR round trip (attention_d40.hip, row_linear.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

// One LDS-DMA piece: 64 lanes x 16 bytes from `rsrc` at per-lane byte offset `voff` (out of range reads 0) to LDS bytes
// [lds_addr, lds_addr + 1024) -- lane-linear, so tiles cannot be padded; bank conflicts are avoided by choosing WHICH
// 16-byte piece of global memory a lane fetches (source-side swizzle).  Inline asm on purpose: through the builtin hipcc
// assumes the DMA may alias every later ds_read of the kernel's one LDS array and drains it (s_waitcnt vmcnt(0)) in front of
// the first fragment read -- the whole point is to keep it in flight.  Completion is waited for by hand (dma_wait*) before
// the barrier that publishes the tile.  M0 (the DMA's LDS base) is saved and restored inside the statement; the leading
// s_nop covers a descriptor / offset register written by a VALU just before (hipcc does not see hazards inside an asm string).
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const v4i_t& rsrc, uint32_t lds_addr, uint32_t voff) {
    uint32_t keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}
// The same piece for steady-state loops whose descriptor and LDS address were produced by SALU instructions well ahead: without the
// five leading wait states (they cover a descriptor SGPR written by a VALU -- v_readfirstlane -- immediately before)
__device__ __forceinline__ void dma16_nonop(const v4i_t& rsrc, uint32_t lds_addr, uint32_t voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// all but the N youngest pieces of this wave (N a compile-time constant; the named forms below are older call sites)
template <int N> __device__ __forceinline__ void dma_wait_keep_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma_wait_keep2() { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }   // all but the 2 youngest pieces
__device__ __forceinline__ void dma_wait_keep3() { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }   // all but the 3 youngest pieces
__device__ __forceinline__ void dma_wait_keep4() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }   // all but the 4 youngest pieces
__device__ __forceinline__ void dma_wait_keep5() { asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }   // all but the 5 youngest pieces
__device__ __forceinline__ v4i_t raw_rsrc(const void* base, uint32_t bytes) {      // stride 0, raw addressing, wave-uniform by construction
    const uint64_t a = (uint64_t)base;
    v4i_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

}  // namespace
