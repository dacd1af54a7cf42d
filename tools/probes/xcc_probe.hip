// Where do the workgroups of a launch run?  Records, per workgroup, the XCC (XCD) id, the hardware id word and the start time, for the
// launch geometries of the LDS-DMA tile kernels (256 threads, 48 KB of dynamic LDS, 1-D and 2-D grids), and reports how the tiles that
// share an operand under gemm_common.h's xcd_tile_order() are spread over the eight L2s.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/xcc_probe.hip -o gpurun_out/xcc_probe && gpurun_out/xcc_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ __launch_bounds__(256, 3) void probe(uint32_t* rec, int spin) {
    extern __shared__ char smem[];
    uint32_t xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const uint64_t t0 = __builtin_readcyclecounter();
    // keep the workgroup resident for a while so that the whole grid is co-resident like the real kernel's
    volatile char* s = smem;
    for (int i = 0; i < spin; ++i) s[(threadIdx.x * 16 + i) & 49151] = (char)i;
    __syncthreads();
    if (threadIdx.x == 0) {
        const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        rec[b * 4 + 0] = xcc & 15u;
        rec[b * 4 + 1] = hwid;
        rec[b * 4 + 2] = (uint32_t)t0;
        rec[b * 4 + 3] = (uint32_t)(t0 >> 32);
    }
}

static void tile_of(int flags, unsigned w, unsigned gx, int m_tiles, int n_tiles, int& tm, int& tn) {      // gemm_common.h::xcd_tile_order
    if (flags & 12) {
        const unsigned k = w & 7u, slot = w >> 3, q8 = gx >> 3, r8 = gx & 7u;
        w = (k < r8 ? k * (q8 + 1) : r8 * (q8 + 1) + (k - r8) * q8) + slot;
    }
    if (flags & 8) { tn = (int)(w / (unsigned)m_tiles); tm = (int)(w - (unsigned)tn * m_tiles); }
    else { tm = (int)(w / (unsigned)n_tiles); tn = (int)(w - (unsigned)tm * n_tiles); }
}

static void run(const char* what, int m_tiles, int n_tiles, int gy, int flags) {
    const unsigned gx = (unsigned)(m_tiles * n_tiles);
    uint32_t* d;
    hipMalloc(&d, (size_t)gx * gy * 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(gx, gy), dim3(256), 49152, 0, d, 2000);
        hipDeviceSynchronize();
    }
    std::vector<uint32_t> h((size_t)gx * gy * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // (1) is XCC == (linear block id) % 8 ?   (2) how many distinct XCCs serve the sharers of one row tile / one channel tile (y = 0)
    long lin_ok = 0, x_ok = 0;
    int off = -1;
    for (unsigned y = 0; y < (unsigned)gy; ++y)
        for (unsigned x = 0; x < gx; ++x) {
            const size_t b = (size_t)y * gx + x;
            if (off < 0) off = (int)h[0];
            lin_ok += ((h[b * 4] + 8 - off) & 7u) == (b & 7u);
            x_ok += ((h[b * 4] + 8 - off) & 7u) == (x & 7u);
        }
    std::vector<unsigned> row_mask(m_tiles, 0), col_mask(n_tiles, 0);
    for (unsigned x = 0; x < gx; ++x) {
        int tm, tn;
        tile_of(flags, x, gx, m_tiles, n_tiles, tm, tn);
        row_mask[tm] |= 1u << h[(size_t)x * 4];
        col_mask[tn] |= 1u << h[(size_t)x * 4];
    }
    double rs = 0, cs = 0;
    for (int i = 0; i < m_tiles; ++i) rs += __builtin_popcount(row_mask[i]);
    for (int i = 0; i < n_tiles; ++i) cs += __builtin_popcount(col_mask[i]);
    printf("{\"probe\": \"%s\", \"grid\": [%u, %d], \"flags\": %d, \"first_block_xcc\": %d, \"frac_xcc_eq_linear_id_mod8\": %.4f, \"frac_xcc_eq_x_mod8\": %.4f, "
           "\"mean_xccs_per_row_tile\": %.3f, \"mean_xccs_per_channel_tile\": %.3f, \"first_16_xcc\": [",
           what, gx, gy, flags, off, (double)lin_ok / (gx * (double)gy), (double)x_ok / (gx * (double)gy), rs / m_tiles, cs / n_tiles);
    for (int i = 0; i < 16 && i < (int)gx; ++i) printf("%u%s", h[(size_t)i * 4], i == 15 || i + 1 == (int)gx ? "" : ", ");
    printf("]");
    if (getenv("XCC_PROBE_CU")) {       // which CU hosts the j-th block of XCC 0?  HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
        printf(", \"cu_key_of_xcc0_blocks\": [");
        int n = 0;
        for (unsigned x = 0; x < gx && n < 128; ++x)
            if (h[(size_t)x * 4] == (unsigned)off) {
                const uint32_t w = h[(size_t)x * 4 + 1];
                printf("%s%u", n ? ", " : "", ((w >> 13) & 7u) * 32 + ((w >> 12) & 1u) * 16 + ((w >> 8) & 15u));
                ++n;
            }
        printf("]");
    }
    printf("}\n");
    hipFree(d);
}

int main() {
    run("linear 8192x640 (64 x 5 tiles of 128^2), row-tile major", 64, 5, 1, 4);
    run("linear 8192x640, channel-tile major", 64, 5, 1, 8);
    run("linear 8192x640, plain order", 64, 5, 1, 0);
    run("linear 8192x5120 (64 x 40 tiles), row-tile major", 64, 40, 1, 4);
    run("linear 2048x1280 (16 x 10 tiles) with 3 K slices", 16, 10, 3, 4);
    run("conv 512x1280 (4 x 10 tiles) with 12 K slices", 4, 10, 12, 8);
    run("conv 32768x320 (256 x 3 tiles)", 256, 3, 1, 4);
    return 0;
}
