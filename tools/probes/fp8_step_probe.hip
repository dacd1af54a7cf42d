// What could an fp8 (MX e4m3) version of the level-0 attention step cost, next to the 16-bit step the kernel runs today?  Register-only loops
// (no memory, no LDS, no barrier -- the parts an fp8 kernel would share with attention_d40.hip) of the instruction mix of 64 keys x two
// 32-query blocks per wave, at two waves per SIMD:
//   bf16 (today, variant 13):  20 v_mfma_f32_32x32x16_bf16 + 8 v_mfma_f32_16x16x32_bf16, 64 v_exp_f32, 32 v_cvt_pk_bf16_f32, 16 v_permlane16_swap
//   fp8  (hypothetical):        6 v_mfma_scale_f32_32x32x64_f8f6f4 (4 QK^T with d = 40 padded to K = 64, 2 P.V) + 2 v_mfma_scale_f32_16x16x128_f8f6f4
//                               (head-dim rows 32..47), 64 v_exp_f32, 32 v_cvt_pk_fp8_f32 (P -> e4m3, one shared block exponent taken from the
//                               deferred maximum: no per-element scale search), 16 v_permlane16_swap
// Reports wall time per 64-key double step and wave slot for the MFMAs alone and for the whole mix.  The difference between the two columns is
// the most an fp8 rebuild of the kernel on today's skeleton could gain per step (staging, barrier and fragment reads come on top of both).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fp8_step_probe.hip -o gpurun_out/fp8_step_probe && gpurun_out/fp8_step_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return __builtin_bit_cast(uint32_t, r);
}

// MODE: bit 0 = fp8 mix (else bf16), bit 1 = with the VALU work (else MFMAs only)
template <int MODE>
__global__ __launch_bounds__(256, 2) void step(float* out, int iters) {
    constexpr bool FP8 = (MODE & 1) != 0, VALU = (MODE & 2) != 0;
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
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    i32x8 a8, b8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a8[i] = 0x38343c30 + lane * 0x01010101 + i; b8[i] = 0x3c383430 + lane + i * 0x01000100; }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (FP8) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i & 3], 0, 0, 0, 127, 0, 127);
                if (VALU) {                  // a sixth of the step's VALU work behind every large MFMA
#pragma unroll
                    for (int e = 0; e < 11 && 11 * i + e < 64; ++e) s[11 * i + e] = __builtin_amdgcn_exp2f(s[11 * i + e]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acct[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acct[i], 0, 0, 0, 127, 0, 127);
                if (VALU) {
#pragma unroll
                    for (int w = 0; w < 8; ++w) {      // 8 words of four e4m3 per tail MFMA = 16 converts of two values (32 per double step)
                        const int e = 4 * (8 * i + w);
                        int t = __builtin_amdgcn_cvt_pk_fp8_f32(s[e], s[e + 1], 0, false);
                        t = __builtin_amdgcn_cvt_pk_fp8_f32(s[e + 2], s[e + 3], t, true);
                        pw[8 * i + w] = (uint32_t)t;
                    }
#pragma unroll
                    for (int w = 0; w < 8; ++w) {      // (the lane swaps of the 16-row tail operand: as many as in the 16-bit kernel)
                        const auto r = __builtin_amdgcn_permlane16_swap(pw[8 * i + w], pw[16 + 8 * i + w], false, false);
                        pw[8 * i + w] = r[0]; pw[16 + 8 * i + w] = r[1];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (VALU) { a8[0] ^= (int)pw[3]; b8[1] ^= (int)pw[19]; }       // the converted words feed the next step's operands
        } else {
#pragma unroll
            for (int i = 0; i < 28; ++i) {
                if (i < 20) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
                else acct[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acct[i & 1], 0, 0, 0);
                if (VALU) {
                    if (i < 22) {
#pragma unroll
                        for (int e = 0; e < 3 && 3 * i + e < 64; ++e) s[3 * i + e] = __builtin_amdgcn_exp2f(s[3 * i + e]);
                    }
                    pw[i] = cvt_pk_bf16(s[(2 * i) & 63], s[(2 * i + 1) & 63]);
                    if (i < 4) pw[28 + i] = cvt_pk_bf16(s[(56 + 2 * i) & 63], s[(57 + 2 * i) & 63]);
                    if (i >= 8 && i < 24) {
                        const auto r = __builtin_amdgcn_permlane16_swap(pw[i - 8], pw[i + 8 > 31 ? i - 4 : i + 8], false, false);
                        pw[i - 8] = r[0]; pw[i + 8 > 31 ? i - 4 : i + 8] = r[1];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (VALU) { a[0] = (__bf16)__uint_as_float(pw[3] << 16); b[1] = (__bf16)__uint_as_float(pw[19] << 16); }
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

template <int MODE>
static double run(const char* what) {
    const int grid = 512, iters = 2000;                 // two 4-wave workgroups per CU: two waves per SIMD
    float* out;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(step<MODE>, dim3(grid), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(step<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ns = best * 1e6 / iters;
    printf("{\"probe\": \"fp8_step\", \"mix\": \"%s\", \"waves_per_simd\": 2, \"ns_per_64_key_double_step\": %.1f}\n", what, ns);
    hipFree(out);
    return ns;
}

int main() {
    const double b0 = run<0>("bf16: 20 x 32x32x16 + 8 x 16x16x32 MFMA only");
    const double f0 = run<1>("fp8: 6 x 32x32x64 f8f6f4 + 2 x 16x16x128 f8f6f4 MFMA only");
    const double b1 = run<2>("bf16: MFMA + 64 exp + 32 cvt_pk_bf16 + 16 permlane16_swap");
    const double f1 = run<3>("fp8: MFMA + 64 exp + 32 cvt_pk_fp8 + 16 permlane16_swap");
    printf("{\"probe\": \"fp8_step\", \"summary\": \"register-only double step, two waves per SIMD\", \"mfma_only_ratio_fp8_over_bf16\": %.3f, "
           "\"whole_mix_ratio_fp8_over_bf16\": %.3f, \"ns_saved_per_double_step\": %.1f}\n", f0 / b0, f1 / b1, b1 - f1);
    return 0;
}
