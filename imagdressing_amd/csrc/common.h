// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of the IMAGDressing hot path.
// Wave = 64 lanes everywhere; MFMA shape used throughout: v_mfma_f32_32x32x16_bf16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef uint16_t bf16_t;  // raw storage type of a bf16 element in global/LDS memory

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// unpack 8 bf16 (a 16-byte vector) into 8 floats
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]);
    v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
    return v;
}

// ---- MFMA ------------------------------------------------------------------------------
// D[32x32] += A[32x16] * B[16x32].  Operand fragments (8 bf16 per lane):
//   A: lane l holds row  i = l & 31, contraction slots 8*(l>>5) .. +7
//   B: lane l holds col  j = l & 31, contraction slots 8*(l>>5) .. +7
//   D: lane l, reg r holds  D[(r&3) + 8*(r>>2) + 4*(l>>5)][l & 31]
// Only the A/B *pairing* of contraction slots matters for the result (a sum), so callers are
// free to permute the contraction index as long as A and B use the same permutation.
__device__ __forceinline__ f32x16 mfma32(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// row of D held by (reg r, half hi)
__device__ __forceinline__ int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// swap bits 2 and 3 of a 5-bit row index (an involution).  Loading A-operand row pi(i) into
// MFMA row i makes lane (col, hi) hold, in regs 8g..8g+7, the 8 *consecutive* source rows
// 16g + 8hi .. +7 -- i.e. exactly a B/A operand fragment for a following MFMA whose
// contraction runs over those rows (used to chain QK^T -> PV without any cross-lane traffic).
__device__ __forceinline__ int swap23(int i) { return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1); }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

#define IMD_DEVINL __device__ __forceinline__
