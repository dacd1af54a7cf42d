// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of the IMAGDressing hot path.
// Wave = 64 lanes everywhere; MFMA shape used throughout: v_mfma_f32_32x32x16_bf16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef uint16_t bf16_t;  // raw 16-bit storage of one activation / weight element (bf16 OR fp16)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// ---- element traits: F16 = false -> bfloat16, F16 = true -> IEEE half ---------------------------
// Both are 16-bit, both feed v_mfma_f32_32x32x16_* at the same rate; fp16 carries 3 more mantissa
// bits (what the reference runs, and what its "fp16 atol 1e-2" parity bar assumes), bf16 the range.
// Conversions compile to single gfx950 instructions (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, RNE).
template <bool F16> struct El;

template <> struct El<false> {
    static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
    static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
    static __device__ __forceinline__ float tof(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        f32x2_t v = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
    }
    static __device__ __forceinline__ bf16_t fromf(float f) { return (bf16_t)(pack2(f, 0.f) & 0xffffu); }
    static __device__ __forceinline__ f32x16 mfma(const uint4& a, const uint4& b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {       // D[16x16] += A[16x32] B[32x16]
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

template <> struct El<true> {
    static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
    static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
    static __device__ __forceinline__ float tof(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        f32x2_t v = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
    }
    static __device__ __forceinline__ bf16_t fromf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
    static __device__ __forceinline__ f32x16 mfma(const uint4& a, const uint4& b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// unpack / pack a 16-byte vector of 8 elements
template <bool F16> __device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    using E = El<F16>;
    f[0] = E::lo(v.x); f[1] = E::hi(v.x); f[2] = E::lo(v.y); f[3] = E::hi(v.y);
    f[4] = E::lo(v.z); f[5] = E::hi(v.z); f[6] = E::lo(v.w); f[7] = E::hi(v.w);
}
template <bool F16> __device__ __forceinline__ uint4 pack8(const float* f) {
    using E = El<F16>;
    uint4 v;
    v.x = E::pack2(f[0], f[1]); v.y = E::pack2(f[2], f[3]);
    v.z = E::pack2(f[4], f[5]); v.w = E::pack2(f[6], f[7]);
    return v;
}

// ---- MFMA ------------------------------------------------------------------------------
// D[32x32] += A[32x16] * B[16x32]  (El<F16>::mfma).  Operand fragments (8 elements per lane):
//   A: lane l holds row  i = l & 31, contraction slots 8*(l>>5) .. +7
//   B: lane l holds col  j = l & 31, contraction slots 8*(l>>5) .. +7
//   D: lane l, reg r holds  D[(r&3) + 8*(r>>2) + 4*(l>>5)][l & 31]
// Only the A/B *pairing* of contraction slots matters for the result (a sum), so callers are
// free to permute the contraction index as long as A and B use the same permutation.
// D[16x16] += A[16x32] * B[32x16]  (El<F16>::mfma16, v_mfma_f32_16x16x32_*; half the work in about half the matrix-pipe time):
//   A: lane l holds row  i = l & 15, contraction slots 8*(l>>4) .. +7      B: lane l holds col j = l & 15, same slots
//   D: lane l, reg r (0..3) holds  D[4*(l>>4) + r][l & 15]

// row of D held by (reg r, half hi)
__device__ __forceinline__ int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// swap bits 2 and 3 of a 5-bit row index (an involution).  Loading A-operand row pi(i) into
// MFMA row i makes lane (col, hi) hold, in regs 8g..8g+7, the 8 *consecutive* source rows
// 16g + 8hi .. +7 -- i.e. exactly a B/A operand fragment for a following MFMA whose
// contraction runs over those rows (used to chain QK^T -> PV without any cross-lane traffic).
__device__ __forceinline__ int swap23(int i) { return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1); }

// x * sigmoid(x) with the hardware exp2 / rcp (1 ulp-class approximations; inputs are 16-bit activations)
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// exact-erf GELU (torch F.gelu default, used by GEGLU and the resampler FeedForward) with erf from
// Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below 16-bit output resolution): 1 rcp + 1 exp2 + 6 fma
// instead of libm erff's ~30 instructions -- the GEGLU epilogue evaluates it 8192 times per 128x128 tile.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = fmaf(-poly, e, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }

#define IMD_DEVINL __device__ __forceinline__
