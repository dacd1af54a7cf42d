// What does the softmax's exponential COST beside the matrix work of the level-0 attention step, and is there a cheaper way to get P = 2^(s - m)
// into a 16-bit MFMA operand?  (Round-5 review item 6b: "the step is bounded by 64 v_exp_f32 + 32 converts per 12 MFMA-equivalents: measure a packed
// 16-bit exponential path".)  Register-only loops -- no memory, no LDS, no barrier -- at two waves per SIMD (512 workgroups of four waves), one
// iteration = the 64-key double step of attention_d40.hip for two 32-query blocks of a wave:
//     matrix work   20 v_mfma_f32_32x32x16 + 8 v_mfma_f32_16x16x32        (bf16 or f16 operands)
//     softmax work  64 scores -> 64 P values packed into 32 words + 16 v_permlane16_swap (the 16x16x32 tail's operand order)
// Softmax variants (all produce a packed 16-bit P from an fp32 score that already has the running maximum subtracted):
//     exp32     v_exp_f32 per element, v_cvt_pk_{bf16,f16}_f32 per pair                                  (what the kernel does)
//     exp16     v_cvt_pkrtz_f16_f32 per pair, v_exp_f16 per element (lo / hi half), v_pack_b32_f16        (fp16 only)
//     poly16    v_cvt_pkrtz_f16_f32 per pair, then PACKED fp16 arithmetic on the pair: round-to-integer by a magic add, cubic on the fraction,
//               exponent inserted with packed integer ops                                              (fp16 only; ~2^-10 relative error)
//     schr      bf16 by integer construction: I = (s + 127) * 128 with a quadratic correction of the mantissa (error 0.3 %: below one bf16 ulp),
//               v_cvt_u32_f32, halves packed with v_perm                                                 (bf16 only)
// Besides the mixes, the issue cost of the single instructions involved (two waves per SIMD, 256 independent instances per iteration).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/exp_path_probe.hip -o gpurun_out/exp_path_probe && gpurun_out/exp_path_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t cvt_pk_f16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
}

enum { SM_NONE = 0, SM_EXP32 = 1, SM_EXP16 = 2, SM_POLY16 = 3, SM_SCHR = 4, SM_EXP_ONLY = 5, SM_CVT_SWAP_ONLY = 6 };

// one pair of scores -> one packed word
template <bool F16, int SM>
__device__ __forceinline__ uint32_t softmax_pair(float s0, float s1) {
    if constexpr (SM == SM_EXP32 || SM == SM_EXP_ONLY) {
        const float e0 = __builtin_amdgcn_exp2f(s0), e1 = __builtin_amdgcn_exp2f(s1);
        if constexpr (SM == SM_EXP_ONLY) return __float_as_uint(e0) ^ __float_as_uint(e1);
        return F16 ? cvt_pk_f16(e0, e1) : cvt_pk_bf16(e0, e1);
    } else if constexpr (SM == SM_CVT_SWAP_ONLY) {
        return F16 ? cvt_pk_f16(s0, s1) : cvt_pk_bf16(s0, s1);
    } else if constexpr (SM == SM_EXP16) {
        const h2 x = __builtin_bit_cast(h2, cvt_pk_f16(s0, s1));
        h2 y;
        y[0] = __builtin_exp2f16(x[0]);
        y[1] = __builtin_exp2f16(x[1]);
        return __builtin_bit_cast(uint32_t, y);
    } else if constexpr (SM == SM_POLY16) {
        // x in [-24, 0] as fp16 pair.  r = rint(x) through the magic constant 1536 (ulp 1 in [1024, 2048)); f = x - r in [-0.5, 0.5];
        // 2^f ~ 1 + f (c1 + f (c2 + f c3)); the integer r sits in the low bits of (x + 1536)'s pattern: shifted into the exponent field and added
        const h2 x = __builtin_bit_cast(h2, cvt_pk_f16(s0, s1));
        const h2 magic = {(_Float16)1536.0f, (_Float16)1536.0f};
        const h2 t = x + magic;
        const h2 r = t - magic;
        const h2 f = x - r;
        const h2 c3 = {(_Float16)0.0555f, (_Float16)0.0555f}, c2 = {(_Float16)0.2402f, (_Float16)0.2402f}, c1 = {(_Float16)0.6931f, (_Float16)0.6931f};
        const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
        h2 pl = c3 * f + c2;
        pl = pl * f + c1;
        pl = pl * f + one;
        const us2 ti = __builtin_bit_cast(us2, t);
        const us2 sh = {10, 10};
        const us2 ex = ti << sh;                       // (low bits of t = 512 + r; the 512 falls out of the 16-bit lane with the shift)
        const us2 res = __builtin_bit_cast(us2, pl) + ex;
        return __builtin_bit_cast(uint32_t, res);
    } else if constexpr (SM == SM_SCHR) {
        // bf16 pattern of 2^s: exponent field = floor(s) + 127, mantissa = 128 (2^f - 1) ~ 128 (f + c f (f - 1)), c = 0.3371 (max error 0.3 %)
        auto one = [](float s) -> uint32_t {
            const float f = s - __builtin_floorf(s);                              // v_fract_f32
            const float g = f * (0.6629f + 0.3371f * f);                          // f + c f (f - 1)
            const float y = ((s - f) + g + 127.0f) * 128.0f;
            return (uint32_t)(int)y;                                              // v_cvt_i32_f32 (s >= -126: non-negative)
        };
        const uint32_t a = one(s0), b = one(s1);
        return a | (b << 16);                                                    // v_lshl_or_b32
    }
    return 0;
}

// MM: 0 no matrix work, 1 the double step's MFMAs.  SM: softmax variant
template <bool F16, int MM, int SM>
__global__ __launch_bounds__(256, 2) void step(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    f32x4 acct[2];
    float s[64];
    uint32_t pw[32];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    acct[0] = acct[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 64; ++i) s[i] = -0.01f * (float)(lane + i);
#pragma unroll
    for (int i = 0; i < 32; ++i) pw[i] = 0x3c003c00u + lane + i;
    bf16x8 a, b;
    f16x8 ah, bh;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i));
        ah[i] = (_Float16)(0.001f * (lane + i)); bh[i] = (_Float16)(0.002f * (lane - i));
    }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        // (opaque touch: the scores are "new" every iteration -- without it the softmax of the unchanged registers is hoisted out of the loop)
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(s[i]));
#pragma unroll
        for (int i = 0; i < 28; ++i) {
            if (MM) {
                if (F16) {
                    if (i < 20) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i & 3], 0, 0, 0);
                    else acct[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acct[i & 1], 0, 0, 0);
                } else {
                    if (i < 20) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
                    else acct[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acct[i & 1], 0, 0, 0);
                }
            }
            if (SM != SM_NONE) {
                // 32 pairs over 28 slots: slot i takes pair i, slots 0..3 also pairs 28..31; the packed word feeds the NEXT iteration's scores
                // (a dependency chain per element through the loop, none inside an iteration -- like the kernel's three interleaved streams)
                pw[i] = softmax_pair<F16, SM>(s[2 * i], s[2 * i + 1]);
                if (i < 4) pw[28 + i] = softmax_pair<F16, SM>(s[56 + 2 * i], s[57 + 2 * i]);
                if (SM != SM_EXP_ONLY && i >= 8 && i < 24) {
                    const auto r = __builtin_amdgcn_permlane16_swap(pw[i - 8], pw[i + 8 > 31 ? i - 4 : i + 8], false, false);
                    pw[i - 8] = r[0]; pw[i + 8 > 31 ? i - 4 : i + 8] = r[1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (SM != SM_NONE) {
#pragma unroll
            for (int i = 0; i < 64; i += 8) s[i] = s[i] * 0.999f - 1e-3f * __uint_as_float((pw[i >> 1] & 0x007f0000u) | 0x3f800000u);   // (keeps every value live and in range)
            if (MM) {
                if (F16) ah[0] = (_Float16)__uint_as_float((pw[3] & 0x3ffu) << 13 | 0x38000000u);
                else a[0] = (__bf16)__uint_as_float(pw[3] << 16);
            }
        }
    }
    float sink = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) sink += acc[j][0] + acc[j][9];
    sink += acct[0][0] + acct[1][1];
#pragma unroll
    for (int i = 0; i < 64; ++i) sink += s[i];
#pragma unroll
    for (int i = 0; i < 32; ++i) sink += (float)pw[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

// issue cost of single instructions: 256 independent instances per iteration, two waves per SIMD
enum { I_EXP32 = 0, I_FMA32 = 1, I_CVTBF = 2, I_CVTH = 3, I_SWAP = 4, I_EXP16 = 5, I_PKFMA16 = 6, I_FRACT = 7, I_CVTI = 8, I_PKADDU16 = 9 };
template <int WHAT>
__global__ __launch_bounds__(256, 2) void instr(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = -0.01f * (float)(lane + i) - 0.5f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                float x = v[i];
                if (WHAT == I_EXP32) asm volatile("v_exp_f32 %0, %1" : "=v"(x) : "v"(v[i]));
                if (WHAT == I_FMA32) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(x) : "v"(v[i]));
                if (WHAT == I_CVTBF) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(x) : "v"(v[i]));
                if (WHAT == I_CVTH) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %1" : "=v"(x) : "v"(v[i]));
                if (WHAT == I_EXP16) asm volatile("v_exp_f16 %0, %1" : "=v"(x) : "v"(v[i]));
                if (WHAT == I_PKFMA16) asm volatile("v_pk_fma_f16 %0, %1, %1, %1" : "=v"(x) : "v"(v[i]));
                if (WHAT == I_FRACT) asm volatile("v_fract_f32 %0, %1" : "=v"(x) : "v"(v[i]));
                if (WHAT == I_CVTI) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(x) : "v"(v[i]));
                if (WHAT == I_PKADDU16) asm volatile("v_pk_add_u16 %0, %1, %1" : "=v"(x) : "v"(v[i]));
                if (WHAT == I_SWAP) {
                    if (i & 1) continue;
                    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 1]), false, false);
                    v[i] = __uint_as_float(r[0]); v[i + 1] = __uint_as_float(r[1]);
                    continue;
                }
                v[i] = x;
            }
        }
    }
    float sink = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) sink += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

template <typename K>
static double time_kernel(K kern, int iters) {
    const int grid = 512;                 // two 4-wave workgroups per CU: two waves per SIMD
    float* out;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipFree(out);
    return best * 1e6 / iters;            // ns per iteration
}

template <bool F16, int MM, int SM>
static double run_mix(const char* what) {
    const double ns = time_kernel(step<F16, MM, SM>, 2000);
    printf("{\"probe\": \"exp_path\", \"dtype\": \"%s\", \"matrix_work\": %d, \"softmax\": \"%s\", \"waves_per_simd\": 2, \"ns_per_64_key_double_step\": %.1f}\n",
           F16 ? "f16" : "bf16", MM, what, ns);
    fflush(stdout);
    return ns;
}

template <int WHAT>
static void run_instr(const char* what, int per_iter) {
    const double ns = time_kernel(instr<WHAT>, 500);
    // two waves per SIMD issue `per_iter` instances each per iteration: ns per wave-instruction on one SIMD = ns / (2 * per_iter)
    printf("{\"probe\": \"exp_path\", \"instruction\": \"%s\", \"ns_per_wave_instruction_on_a_simd\": %.3f}\n", what, ns / (2.0 * per_iter));
    fflush(stdout);
}

int main() {
    run_instr<I_FMA32>("v_fma_f32", 256);
    run_instr<I_EXP32>("v_exp_f32", 256);
    run_instr<I_EXP16>("v_exp_f16", 256);
    run_instr<I_CVTBF>("v_cvt_pk_bf16_f32", 256);
    run_instr<I_CVTH>("v_cvt_pkrtz_f16_f32", 256);
    run_instr<I_PKFMA16>("v_pk_fma_f16", 256);
    run_instr<I_PKADDU16>("v_pk_add_u16", 256);
    run_instr<I_FRACT>("v_fract_f32", 256);
    run_instr<I_CVTI>("v_cvt_i32_f32", 256);
    run_instr<I_SWAP>("v_permlane16_swap_b32", 128);
    // bf16
    const double b_m = run_mix<false, 1, SM_NONE>("none");
    const double b_0 = run_mix<false, 1, SM_EXP32>("exp32 (shipped)");
    const double b_v = run_mix<false, 0, SM_EXP32>("exp32 (shipped)");
    run_mix<false, 0, SM_EXP_ONLY>("64 v_exp_f32 only");
    run_mix<false, 0, SM_CVT_SWAP_ONLY>("32 converts + 16 swaps only");
    const double b_s = run_mix<false, 1, SM_SCHR>("schr (integer-built bf16, quadratic mantissa)");
    run_mix<false, 0, SM_SCHR>("schr (integer-built bf16, quadratic mantissa)");
    // f16
    const double h_m = run_mix<true, 1, SM_NONE>("none");
    const double h_0 = run_mix<true, 1, SM_EXP32>("exp32 (shipped)");
    run_mix<true, 0, SM_EXP32>("exp32 (shipped)");
    const double h_e = run_mix<true, 1, SM_EXP16>("exp16 (v_exp_f16 on converted pairs)");
    run_mix<true, 0, SM_EXP16>("exp16 (v_exp_f16 on converted pairs)");
    const double h_p = run_mix<true, 1, SM_POLY16>("poly16 (packed fp16 cubic)");
    run_mix<true, 0, SM_POLY16>("poly16 (packed fp16 cubic)");
    printf("{\"probe\": \"exp_path\", \"summary\": \"64-key double step, register only, two waves per SIMD\", \"bf16\": {\"matrix_only\": %.1f, \"shipped\": %.1f, "
           "\"softmax_alone\": %.1f, \"schr\": %.1f}, \"f16\": {\"matrix_only\": %.1f, \"shipped\": %.1f, \"exp16\": %.1f, \"poly16\": %.1f}}\n",
           b_m, b_0, b_v, b_s, h_m, h_0, h_e, h_p);
    return 0;
}
