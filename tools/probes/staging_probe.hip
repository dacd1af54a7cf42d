// How many bytes per clock can a CU move from the L2 into LDS -- by LDS-DMA (buffer_load ... lds), through registers (buffer_load ->
// ds_write_b128), or by both paths at once?  The LDS-DMA tile kernels of this library sit on ~21 B/clk per CU of operand staging
// (DESIGN.md section 6); if the register path has a limit of its own, staging part of a tile each way would lift the sum.
// One workgroup = 256 threads and 48 KB of LDS (three per CU, like conv_patch.hip); every iteration stages PIECES one-KB pieces per wave
// from an L2-resident source (64-byte row segments, four lanes per row), `dma` of them by LDS-DMA and the rest through registers, then
// crosses a barrier.  Prints bytes per clock per CU from the wall time and the shader clock the run reports.
// Two source footprints: 2 MB (L2 hits on every XCD) and 24 MB (Infinity Cache).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/staging_probe.hip -o tools/probes/staging_probe && tools/probes/staging_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int v4i_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4i_t raw_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    v4i_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

template <int PIECES, int DMA, int LPR = 4>       // LPR lanes per row: 64-byte (4), 128-byte (8) or 256-byte (16) contiguous segments
__global__ __launch_bounds__(256, 3) void stage(const char* src, uint32_t src_bytes, int iters, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const v4i_t d = raw_rsrc(src, src_bytes);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, src_bytes, 0x00020000);
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    // piece i of this wave: 16 rows of 64 B at a 2560-byte row stride (an NHWC activation / a weight row), advancing 64 B per iteration
    uint32_t cur[PIECES], dst[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int row = ((blockIdx.x * 4 + wave) * PIECES + i) * (64 / LPR) + lane / LPR;
        cur[i] = (uint32_t)(((size_t)row * 2560 + (lane % LPR) * 16) % (src_bytes - 4096));
        dst[i] = smem_base + ((wave * PIECES + i) % 48) * 1024;
    }
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 r[PIECES > DMA ? PIECES - DMA : 1];
#pragma unroll
        for (int i = DMA; i < PIECES; ++i)
        {
            typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4u;
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)cur[i], 0, 0);
            r[i - DMA] = make_uint4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int i = 0; i < DMA; ++i)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(cur[i]), "s"(dst[i]), "s"(d) : "memory");
#pragma unroll
        for (int i = DMA; i < PIECES; ++i) *reinterpret_cast<uint4*>(smem + (dst[i] - smem_base) + lane * 16) = r[i - DMA];
#pragma unroll
        for (int i = 0; i < PIECES; ++i) cur[i] = (cur[i] + LPR * 16) % (src_bytes - 4096);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *reinterpret_cast<const uint32_t*>(smem + ((tid * 16 + it * 64) & 49151));
        __syncthreads();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int PIECES, int DMA, int LPR = 4>
static void run(const char* src, uint32_t bytes, uint32_t* sink, int clock_mhz) {
    const int iters = 400, grid = 256 * 3;
    auto k = stage<PIECES, DMA, LPR>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 48 * 1024, 0, src, bytes, iters, sink);
    hipEventRecord(e0, 0);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 48 * 1024, 0, src, bytes, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = 3.0 * iters * 3 /*WG per CU*/ * 4 /*waves*/ * PIECES * 1024.0;
    const double us = ms * 1000.0;
    printf("{\"segment_bytes\": %d, \"pieces_per_wave\": %d, \"by_dma\": %d, \"by_registers\": %d, \"us\": %.1f, \"GB_per_s_per_cu\": %.1f, \"B_per_clk_per_cu_at_%dMHz\": %.1f}\n",
           LPR * 16, PIECES, DMA, PIECES - DMA, us, bytes_per_cu / us / 1e3, clock_mhz, bytes_per_cu / us / clock_mhz);
}

int main() {
    char* src;
    uint32_t* sink;
    hipMalloc(&src, 24u << 20);
    hipMemset(src, 1, 24u << 20);
    hipMalloc(&sink, 64);
    int clk = 2400;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    clk /= 1000;
    for (uint32_t bytes : {2u << 20, 24u << 20}) {          // every XCD's L2 holds it | Infinity-Cache resident
    printf("# source footprint %u MB\n", bytes >> 20);
    run<4, 4>(src, bytes, sink, clk); run<4, 0>(src, bytes, sink, clk); run<4, 2>(src, bytes, sink, clk); run<4, 3>(src, bytes, sink, clk); run<4, 1>(src, bytes, sink, clk);
    run<8, 8>(src, bytes, sink, clk); run<8, 0>(src, bytes, sink, clk); run<8, 4>(src, bytes, sink, clk); run<8, 6>(src, bytes, sink, clk); run<8, 5>(src, bytes, sink, clk);
    run<12, 12>(src, bytes, sink, clk); run<12, 8>(src, bytes, sink, clk); run<12, 6>(src, bytes, sink, clk);
    run<8, 8, 8>(src, bytes, sink, clk); run<8, 0, 8>(src, bytes, sink, clk); run<8, 8, 16>(src, bytes, sink, clk); run<8, 0, 16>(src, bytes, sink, clk); run<8, 8, 2>(src, bytes, sink, clk); run<8, 8, 1>(src, bytes, sink, clk);
    }
    return 0;
}
