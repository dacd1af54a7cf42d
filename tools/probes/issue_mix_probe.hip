// What does the level-0 attention kernel's instruction mix cost on one SIMD?  Register-only loops (no memory, no LDS, no barrier)
// of the per-step mix of attention_d40.hip -- 10 v_mfma_f32_32x32x16 + 4 v_mfma_f32_16x16x32, 32 v_exp_f32, 16 v_cvt_pk_bf16_f32,
// 8 v_pk_maximum3_f16, 8 v_permlane16_swap -- alone and interleaved, at one and two waves per SIMD; cycles per iteration from
// s_memtime.  The floor of the mix with perfect overlap is what ANY schedule of that kernel can reach.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/issue_mix_probe.hip -o tools/probes/issue_mix_probe && tools/probes/issue_mix_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c) {
    const h2_t x = __builtin_bit_cast(h2_t, a), y = __builtin_bit_cast(h2_t, b), z = __builtin_bit_cast(h2_t, c);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(x, y), z));
}
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return __builtin_bit_cast(uint32_t, r);
}

// MODE bits: 1 MFMA  2 exp  4 cvt_pk  8 max3  16 permlane swap  32 interleave pinned slot by slot (else stream after stream)
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const v4i_t& rsrc, uint32_t lds_addr, uint32_t voff) {          // lds_dma.h::dma16
    uint32_t keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void dma16_slim(const v4i_t& rsrc, uint32_t lds_addr, uint32_t voff) {     // no wait states, m0 not preserved
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}

// MODE bits: 1 MFMA  2 exp  4 cvt_pk  8 max3  16 permlane swap  32 interleave pinned slot by slot (else stream after stream)
//   64 the MFMA A operands are LDS fragments (6 ds_read_b128 per step, issued two fragments ahead, like attention_d40.hip)
//   128 LDS-DMA staging: 3 one-KB pieces per wave every second step (lds_dma.h::dma16)   256 ... with the slim issue sequence
//   512 counted vmcnt wait + workgroup barrier every second step
template <int MODE>
__global__ __launch_bounds__(256, 2) void mix(float* out, unsigned long long* cyc, int iters, const char* src) {
    __shared__ __attribute__((aligned(16))) char lds[3 * 12288];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v4i_t rs;
    {
        const uint64_t ad = (uint64_t)src;
        rs[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)ad); rs[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(ad >> 32) & 0xffffu));
        rs[2] = 1 << 20; rs[3] = 0x00020000;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    for (int i = threadIdx.x; i < 3 * 12288 / 4; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = 0x3c003c00u;
    int ring = 0;
    f32x16 acc[4];
    f32x4 acct[2];
    float s[32];
    uint32_t pw[16], mq = 0;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    acct[0] = acct[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32; ++i) s[i] = -0.01f * (float)(lane + i);
#pragma unroll
    for (int i = 0; i < 16; ++i) pw[i] = 0x3c003c00u + lane + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if ((MODE & 128) && (it & 1) == 0) {             // a unit of K / V^T every second step: 3 pieces per wave
            const uint32_t dst = lds0 + (uint32_t)(ring * 12288 + wave * 3072);
            const uint32_t so = (uint32_t)(((it >> 1) & 15) * 12288 + wave * 3072 + lane * 16);
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) {
                if (MODE & 256) dma16_slim(rs, dst + pz * 1024, so + pz * 1024); else dma16(rs, dst + pz * 1024, so + pz * 1024);
            }
            ring = ring == 2 ? 0 : ring + 1;
        }
        if (MODE & 32) {
            bf16x8 fr[3];
            const char* fb = lds + ((it & 1) ? 6144 : 0) + ((MODE & 128) ? (ring == 0 ? 2 : ring - 1) * 12288 : 0) + lane * 16;
            if (MODE & 64) { fr[0] = *reinterpret_cast<const bf16x8*>(fb); fr[1] = *reinterpret_cast<const bf16x8*>(fb + 1024); }
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                if ((MODE & 64) && (i & 1) == 0 && i <= 6) fr[((i >> 1) + 2) % 3] = *reinterpret_cast<const bf16x8*>(fb + ((i >> 1) + 2) * 1024);
                if (MODE & 1) {
                    const bf16x8 av = (MODE & 64) ? fr[(i < 10 ? (i >> 1) : 5) % 3] : a;
                    if (i < 10) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b, acc[i & 3], 0, 0, 0);
                    else acct[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b, acct[i & 1], 0, 0, 0);
                }
                if (MODE & 2) {
                    const int e0 = (2 * i) & 31, e1 = (2 * i + 1) & 31;
                    s[e0] = __builtin_amdgcn_exp2f(s[e0]); s[e1] = __builtin_amdgcn_exp2f(s[e1]);
                    if (i >= 12) { const int e2 = (2 * i + 4) & 31, e3 = (2 * i + 5) & 31; s[e2] = __builtin_amdgcn_exp2f(s[e2]); s[e3] = __builtin_amdgcn_exp2f(s[e3]); }
                }
                if (MODE & 4) { pw[i] = cvt_pk(s[(2 * i) & 31], s[(2 * i + 1) & 31]); if (i >= 12) pw[i + 2] = cvt_pk(s[(2 * i + 4) & 31], s[(2 * i + 5) & 31]); }
                if ((MODE & 8) && (i & 1)) mq = pk_max3(mq, pw[i - 1], pw[i]);
                if ((MODE & 16) && i >= 4 && i < 12) {
                    const auto r = __builtin_amdgcn_permlane16_swap(pw[i - 4], pw[i + 4], false, false);
                    pw[i - 4] = r[0]; pw[i + 4] = r[1];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if (MODE & 1) {
#pragma unroll
                for (int i = 0; i < 10; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acct[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acct[i & 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE & 2) {
#pragma unroll
                for (int i = 0; i < 32; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE & 4) {
#pragma unroll
                for (int i = 0; i < 16; ++i) pw[i] = cvt_pk(s[2 * i], s[2 * i + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE & 8) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) mq = pk_max3(mq, pw[i], pw[i + 1]);
            }
            if (MODE & 16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const auto r = __builtin_amdgcn_permlane16_swap(pw[i], pw[i + 8], false, false);
                    pw[i] = r[0]; pw[i + 8] = r[1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((MODE & 512) && (it & 1)) {
            if (MODE & 128) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) sink += acc[j][0] + acc[j][7];
    sink += acct[0][0] + acct[1][1];
#pragma unroll
    for (int i = 0; i < 32; ++i) sink += s[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) sink += (float)pw[i];
    sink += (float)mq;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
static void run(const char* what, int wgs_per_cu) {
    const int grid = 256 * wgs_per_cu, iters = 2000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipMalloc(&cyc, (size_t)grid * 4 * 8);
    char* src; hipMalloc(&src, 1 << 20); hipMemset(src, 0x3c, 1 << 20);
    // one workgroup per CU: 100 KB of dynamic LDS keeps a second one off the CU
    const int lds = wgs_per_cu == 1 ? 64 * 1024 : 0;      // + 36 KB static
    hipFuncSetAttribute(reinterpret_cast<const void*>(mix<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mix<MODE>, dim3(grid), dim3(256), lds, 0, out, cyc, 10, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mix<MODE>, dim3(grid), dim3(256), lds, 0, out, cyc, iters, src);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)grid * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    // s_memtime ticks at a constant 100 MHz on this part; report both ticks and wall-derived shader cycles at the measured rate
    printf("{\"mix\": \"%s\", \"waves_per_simd\": %d, \"memtime_ticks_per_iter\": %.2f, \"wall_us_per_iter_per_wave_slot\": %.4f}\n", what, wgs_per_cu, mean / iters,
           ms * 1e3 / iters);
    hipFree(out); hipFree(cyc); hipFree(src);
}

#define BOTH(MODE, WHAT) run<MODE>(WHAT, 1); run<MODE>(WHAT, 2);
int main() {
    BOTH(1, "14 MFMA (10 x 32x32x16 + 4 x 16x16x32)")
    BOTH(2, "32 v_exp_f32")
    BOTH(4, "16 v_cvt_pk_bf16_f32")
    BOTH(8, "8 v_pk_maximum3_f16")
    BOTH(16, "8 v_permlane16_swap")
    BOTH(2 | 4 | 8 | 16, "all VALU of a step, stream after stream")
    BOTH(1 | 2, "MFMA then exp (stream after stream)")
    BOTH(1 | 2 | 32, "MFMA + 2 exp per slot, pinned")
    BOTH(1 | 2 | 4 | 8 | 16, "the whole step, stream after stream (compiler order inside streams)")
    BOTH(1 | 2 | 4 | 8 | 16 | 32, "the whole step, pinned slot by slot like attention_d40.hip")
    BOTH(1 | 4 | 8 | 16 | 32, "the step without exp, pinned")
    BOTH(1 | 2 | 4 | 32, "MFMA + exp + cvt, pinned")
    BOTH(1 | 32 | 64, "MFMA with LDS fragments, pinned")
    BOTH(1 | 2 | 4 | 8 | 16 | 32 | 64, "whole step + LDS fragment reads")
    BOTH(1 | 2 | 4 | 8 | 16 | 32 | 64 | 512, "whole step + LDS fragment reads + barrier every second step")
    BOTH(1 | 2 | 4 | 8 | 16 | 32 | 64 | 128 | 512, "whole step + LDS fragments + LDS-DMA staging (dma16) + counted wait + barrier")
    BOTH(1 | 2 | 4 | 8 | 16 | 32 | 64 | 128 | 256 | 512, "whole step + LDS fragments + LDS-DMA staging (slim issue) + counted wait + barrier")
    BOTH(1 | 2 | 4 | 8 | 16 | 32 | 128 | 256, "whole step + LDS-DMA staging (slim issue), no fragment reads, no barrier")
    BOTH(1 | 32 | 128 | 256, "MFMA + LDS-DMA staging (slim issue)")
    return 0;
}
