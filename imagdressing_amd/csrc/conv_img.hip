// 3x3 / stride-1 convolution of the SMALL feature maps (8 pixels wide: the 8 x 8 level of the UNet at 512 x 512, 10 x 8 at 512 x 640) --
// the weight-streaming convolutions (gfx950, bf16 / fp16 MFMA).  Tile config 24 of imd_conv_gemm.
//
// At this level a convolution is M = 512 rows (8 images of the CFG batch x 64 pixels) against 29.5 / 59 MB of weights (Cin = 1280 /
// 2560, N = 1280): every weight byte is needed exactly once, the activations (1.3 MB) live in the L2.  The tiled kernels cut M into 64- /
// 128-row tiles, so each weight slice is fetched by 4..8 workgroups, and walk K as a chain of global -> LDS round trips (0.8 TB/s of
// weights, 0.16 of the matrix pipe: profiles/r3aa_*).  Here
//   * a workgroup owns ALL pixels of up to 4 images (one wave per image) x 64 output channels x one K slice: a weight byte is fetched by
//     one workgroup per 4 images (two for the CFG batch of the bench, neighbours on one XCD);
//   * K runs in 32-channel chunks.  The images' zero-haloed patches of a chunk ((H + 2) x 12 positions x 64 B per image) are double
//     buffered; the weights stream per TAP ROW (3 taps x 64 rows x 64 B = 12 KB) through a six-slot ring.  One "unit" = (chunk, tap
//     row) multiplies 24 MFMAs per wave while the weights of units + 1 .. + 5 and the next chunk's patches are in flight (61 KB per CU:
//     what the ~2 us memory latency needs at ~20 B/ns); everything reaches LDS by LDS-DMA (counted s_waitcnt, one barrier per unit),
//     halo = out-of-range source offset (the DMA writes zeros);
//   * four lanes fetch the four 16-byte pieces of a 64-byte row segment (the first form of this kernel moved 16-channel chunks as one
//     piece per lane: every piece a different 128-byte line, 31 us for 15 GFLOP -- the L1 refetched each line four times); the rows sit
//     unpadded in LDS with piece c of row r at position c ^ ((r >> 2) & 3), and the MFMA row -> pixel map plus the 12-position patch row
//     stride make each 16-lane group of a ds_read_b128 cover 16 distinct bank slots for every tap (kRowPix);
//   * MFMA A operand = pixels, B operand = output channels, so an accumulator register holds 32 consecutive channels of one pixel: the
//     fp32 K-slice slab is written in 128-byte segments; the slices are summed and finished (bias / time embedding / residual /
//     GroupNorm statistics) by the shared second launch of conv_gemm.hip;
//   * block index -> (K slice, image group, channel tile) so that the workgroups of one K slice -- they share the activation slice --
//     run on one XCD.
// Where the time goes (8 images, Cin = N = 1280, 6 K slices = 240 workgroups, 24.6 us; profiles/r4z_*): launch ramp + drain of a grid of
// 136-KB-LDS workgroups ~6 us, first operands from HBM ~3 us, 15.7 MB of fp32 slabs ~6 us, the 20 units ~10 us (MFMA pipe 75 % busy inside
// them) -- and the finish launch behind it.  The K-sliced two-launch form itself is what bounds these convolutions now, not the weight
// stream: a contiguous re-layout of the weights changed nothing (-DCI_PROBE=1/2/3 builds time prologue / staging / launch alone).
// Reference arithmetic: torch.nn.Conv2d(k = 3, padding = 1) inside diffusers' ResnetBlock2D at the UNet's lowest level
// (IMAGDressing_v1_pipeline.py:466,499,511 call the UNets).
#include <type_traits>

#include "gemm_common.h"
#include "lds_dma.h"

namespace {

constexpr int CI_CK = 32;                       // input channels per chunk: 64-byte row segments, two MFMA K steps
constexpr int CI_BN = 64;                       // output channels per workgroup
constexpr int CI_W = 8;                         // map width
constexpr int CI_PW = 12;                       // patch row stride in positions: halo, 8 pixels, halo, 2 unused (12: see kRowPix)
constexpr int CI_WUNIT = 3 * CI_BN * 64;        // weights of one tap row: 3 taps x 64 rows x 64 B = 12 pieces
constexpr int CI_WRING = 6;                     // weight ring: the tap rows of two chunks, i.e. unit u + 5 is staged while unit u is multiplied

template <int IMG, int NPB>
struct CI {                                     // IMG images (= waves) per workgroup, NPB 32-pixel blocks per image (H <= 4 NPB rows)
    static constexpr int NW = IMG, NT = NW * 64;
    static constexpr int MAXPOS = (4 * NPB + 2) * CI_PW;
    static constexpr int APIECES = (IMG * MAXPOS + 15) / 16;              // 16 rows of 64 B per piece
    static constexpr int APC = (APIECES + NW - 1) / NW;                   // patch pieces a wave stages per chunk (dummies included)
    static constexpr int WPU = (12 + NW - 1) / NW;                        // weight pieces a wave stages per unit
    static constexpr int ABUF = APIECES * 1024;
    static constexpr int WOFF = 2 * ABUF;
    static constexpr int DUMP = WOFF + CI_WRING * CI_WUNIT;               // where dummy pieces land
    static constexpr int LDS = DUMP + 1024;
    static_assert(LDS <= 160 * 1024, "buffers do not fit the LDS");
    static_assert(ABUF < 65536 && 3 * CI_WUNIT < 65536, "buffer / slot offsets ride in the ds_read offset field");
};

// MFMA row (A-operand lane & 31) -> pixel of the 4-row x 8-column block.  ds_read_b128 is serviced in the 16-lane groups {0-3, 12-15,
// 20-27} / {4-11, 16-19, 28-31}; the first group gets image rows 0 and 2 of the block, the second rows 1 and 3.  At 12 positions per
// patch row a group then reads patch rows P .. P + 7 and P + 24 .. P + 31 for some P (a tap shifts all alike): (row & 3, (row >> 2) & 3)
// is distinct over those sixteen, i.e. with piece c of a 64-byte row stored at c ^ ((row >> 2) & 3) they fall on 16 distinct bank slots.
__device__ constexpr unsigned char kRowPix[32] = {0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 15, 4, 5, 6, 7,
                                                  24, 25, 26, 27, 16, 17, 18, 19, 20, 21, 22, 23, 28, 29, 30, 31};

template <bool F16, int IMG, int NPB>
__global__ __launch_bounds__(IMG * 64, 1) void conv3x3_img_kernel(const ConvGemmParams p) {
    using E = El<F16>;
    using G = CI<IMG, NPB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.Hout, HW = H * CI_W, NPOS = (H + 2) * CI_PW;
    const int B = p.M / HW;
    const int n_tiles = p.N / CI_BN, groups = (B + IMG - 1) / IMG, per_split = n_tiles * groups;

    // hardware block L runs on XCD L % 8: every XCD gets a contiguous range of (K slice, image group, channel tile)
    int split, grp, tile_n;
    {
        const unsigned total = gridDim.x, Lb = blockIdx.x;
        const unsigned xcd = Lb & 7u, slot = Lb >> 3, q8 = total >> 3, r8 = total & 7u;
        const unsigned v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
        split = (int)(v / per_split);
        const int rem = (int)(v % per_split);
        tile_n = rem / groups;                       // (the image groups of one weight tile are neighbours: one fetch from HBM, one from the L2)
        grp = rem % groups;
    }
    const int n0 = tile_n * CI_BN, b0 = grp * IMG, nimg = min(IMG, B - b0);
    const int nch = p.Cin / CI_CK, cbase = nch / p.split_k, cextra = nch % p.split_k;
    const int c_begin = split * cbase + min(split, cextra), c_cnt = cbase + (split < cextra ? 1 : 0);

    // ---- staging: per-lane running source offsets, wave-uniform LDS destinations ----
    constexpr uint32_t FAR = 0x80000000u;           // stays out of range under the running adds (operands < 2 GiB: checked by the launcher)
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    const v4i_t dx = raw_rsrc(p.x, p.x_bytes), dw = raw_rsrc(p.w, p.w_bytes);
    uint32_t acur[G::APC];                          // patch piece i of the NEXT chunk to stage
    uint32_t adst[2][G::APC];                       // ... its place in patch buffer 0 / 1
    uint32_t wcur[3][G::WPU];                       // weight piece i of tap row dy, next chunk to stage
    uint32_t wdst[CI_WRING][G::WPU];                // ... its place in ring slot s
#pragma unroll
    for (int k = 0; k < G::APC; ++k) {
        const int pj = k * G::NW + wave;             // piece id: patch rows 16 pj .. + 15, four lanes per row
        const int row = pj * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        const int img = row / NPOS, pos = row - img * NPOS;
        const int yy = pos / CI_PW, xx = pos - yy * CI_PW;
        const bool ok = pj < G::APIECES && img < nimg && yy >= 1 && yy <= H && xx >= 1 && xx <= CI_W;
        acur[k] = ok ? (uint32_t)((((size_t)(b0 + img) * H + (yy - 1)) * CI_W + (xx - 1)) * p.x_pix_stride + c_begin * CI_CK + 8 * c) * 2u : FAR;
        adst[0][k] = smem_base + (pj < G::APIECES ? pj * 1024 : G::DUMP);
        adst[1][k] = smem_base + (pj < G::APIECES ? G::ABUF + pj * 1024 : G::DUMP);
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int i = 0; i < G::WPU; ++i) {
            const int pj = i * G::NW + wave;         // piece id inside the tap row's tile: rows 16 pj .. + 15 of (tap, channel)
            const int R = pj * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((R >> 2) & 3);
            const int tl = R / CI_BN, n = R % CI_BN;
            const bool ok = pj < 12 && n0 + n < p.N;
            wcur[dy][i] = ok ? (uint32_t)(((size_t)(n0 + n) * p.K + (size_t)(3 * dy + tl) * p.Cin + c_begin * CI_CK + 8 * c) * 2) : FAR;
            wdst[dy][i] = smem_base + (pj < 12 ? G::WOFF + dy * CI_WUNIT + pj * 1024 : G::DUMP);
            wdst[dy + 3][i] = smem_base + (pj < 12 ? G::WOFF + (dy + 3) * CI_WUNIT + pj * 1024 : G::DUMP);
        }
    auto dma = [&](uint32_t& cur, uint32_t dst, const v4i_t& desc) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(cur), "s"(dst), "s"(desc) : "memory");
        cur += CI_CK * 2;
    };

    // ---- fragment addresses ----
    const int q = kRowPix[col];
    int xaddr[NPB][9][2];                           // (pixel block, tap, K step) inside a patch buffer
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int P = wave * NPOS + (4 * pb + (q >> 3) + t / 3) * CI_PW + (q & 7) + t % 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xaddr[pb][t][ks] = P * 64 + (((2 * ks + hi) ^ ((P >> 2) & 3)) << 4);
        }
    int waddr[2][2];                                // (ring half, K step) of weight row `col`; (slot in the half, tap, channel block) are constants
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) waddr[hf][ks] = G::WOFF + hf * 3 * CI_WUNIT + col * 64 + (((2 * ks + hi) ^ ((col >> 2) & 3)) << 4);
#if defined(CI_PROBE) && CI_PROBE == 2
    const bool live = false;                        // timing probe: stage only
#else
    const bool live = wave < nimg;                  // (scalar: waves without an image only stage)
#endif

    f32x16 acc[NPB][2];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pb][nb][r] = 0.f;

    // unit (chunk on patch buffer buf_c, tap row dy_c) = ring slot 3 buf_c + dy_c: the weights of unit + 5 staged behind it and, at dy = 0,
    // the next chunk's patches (HBM latency is ~2 us under load and a unit multiplies for ~0.4 us: with unit + 2 in flight -- the first
    // form of this loop -- every unit waited on memory, 26 us for the 8 x 8 level's conv)
    auto unit = [&](auto buf_c, auto dy_c) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value, DY = decltype(dy_c)::value, NS = (3 * BUF + DY + 5) % CI_WRING, SRC = (DY + 2) % 3;
        if (DY == 0) {
#pragma unroll
            for (int i = 0; i < G::APC; ++i) dma(acur[i], adst[1 - BUF][i], dx);       // (that buffer was last read a chunk ago)
        }
#pragma unroll
        for (int i = 0; i < G::WPU; ++i) dma(wcur[SRC][i], wdst[NS][i], dw);          // (that slot was last read one unit ago)
        if (live) {
            // six (tap, K step) groups of 2 + NPB fragment reads and 2 NPB MFMAs; the reads of group g + 1 are issued before the MFMAs of
            // group g (one wave per SIMD: nobody else covers the LDS latency -- left to the compiler every group waited for its own reads)
            uint4 w0[2], w1[2], xf[2][NPB];
            auto fetch = [&](int sel, int g) __attribute__((always_inline)) {
                const int d = g >> 1, ks = g & 1;
                w0[sel] = *reinterpret_cast<const uint4*>(smem + waddr[BUF][ks] + DY * CI_WUNIT + (d * CI_BN) * 64);
                w1[sel] = *reinterpret_cast<const uint4*>(smem + waddr[BUF][ks] + DY * CI_WUNIT + (d * CI_BN + 32) * 64);
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) xf[sel][pb] = *reinterpret_cast<const uint4*>(smem + BUF * G::ABUF + xaddr[pb][3 * DY + d][ks]);
            };
            fetch(0, 0);
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                if (g + 1 < 6) fetch((g + 1) & 1, g + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) {
                    acc[pb][0] = E::mfma(xf[g & 1][pb], w0[g & 1], acc[pb][0]);
                    acc[pb][1] = E::mfma(xf[g & 1][pb], w1[g & 1], acc[pb][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // in-order counter: the youngest pieces are W(u+5) .. W(u+2) and the patches staged at this chunk's dy = 0; the next unit's weights
        // and (at dy = 2) the next chunk's patches are older than what may stay in flight
        dma_wait_keep_n<(DY == 2 ? 3 * G::WPU : 4 * G::WPU + G::APC)>();
        __syncthreads();
    };
    const std::integral_constant<int, 0> c0{};
    const std::integral_constant<int, 1> c1{};
    const std::integral_constant<int, 2> c2{};

    // prologue: the first chunk's patches, the weights of units 0 .. 4
#if !defined(CI_PROBE) || CI_PROBE != 3
#pragma unroll
    for (int k = 0; k < G::APC; ++k) dma(acur[k], adst[0][k], dx);
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int i = 0; i < G::WPU; ++i) dma(wcur[u % 3][i], wdst[u][i], dw);
    dma_wait_keep_n<4 * G::WPU>();
    __syncthreads();
#endif
    int ci = 0;
#if defined(CI_PROBE) && (CI_PROBE == 1 || CI_PROBE == 3)
    ci = c_cnt;                                     // timing probe: prologue and epilogue only
#endif
    for (; ci + 2 <= c_cnt; ci += 2) {
        unit(c0, c0); unit(c0, c1); unit(c0, c2);
        unit(c1, c0); unit(c1, c1); unit(c1, c2);
    }
    if (ci < c_cnt) { unit(c0, c0); unit(c0, c1); unit(c0, c2); }
    dma_wait();                         // pieces staged past the end are still landing: nobody may leave with DMA in flight into LDS

    // ---- fp32 K-slice slab: register r of accumulator (pb, nb) = pixel row mfma_row(r, hi), 32 consecutive channels across the lanes ----
    if (live) {
        float* slab = p.splitk_ws + (size_t)split * p.M * p.N;
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = pb * 32 + kRowPix[mfma_row(r, hi)];
                if (pix < HW) {
                    float* dst = slab + ((size_t)(b0 + wave) * HW + pix) * p.N + n0 + col;
                    dst[0] = acc[pb][0][r];
                    dst[32] = acc[pb][1][r];
                }
            }
    }
}

template <int IMG, int NPB>
int launch_img(const ConvGemmParams& p, hipStream_t s) {
    using G = CI<IMG, NPB>;
    const bool h = p.dtype == IMD_DTYPE_F16;
    typedef void (*kern_t)(const ConvGemmParams);
    const kern_t kern = h ? conv3x3_img_kernel<true, IMG, NPB> : conv3x3_img_kernel<false, IMG, NPB>;
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), G::LDS, "conv_img")) return rc_attr;
    const int B = p.M / (p.Hout * p.Wout);
    const long blocks = (long)p.split_k * (p.N / CI_BN) * ((B + IMG - 1) / IMG);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(IMG * 64), G::LDS, s, p);
    return imd_check_launch("conv_img");
}

}  // namespace

// tile config 24: 3x3 stride-1 convolutions of maps 8 pixels wide and at most 12 rows high, K-sliced (the slabs are the only output form)
bool imd_conv_img_supported(const ConvGemmParams& p) {
    return p.taps == 9 && p.stride == 1 && !p.ups && !p.pad_br_only && p.Hin == p.Hout && p.Win == p.Wout && p.Wout == CI_W && p.Hout >= 1 &&
           p.Hout <= 12 && p.K == 9 * p.Cin && (p.Cin % CI_CK) == 0 && (p.N % CI_BN) == 0 && p.mode == OUT_ROWMAJOR && p.act != ACT_GEGLU &&
           !p.out_f32 && p.gn_a == nullptr && (p.x_pix_stride % 8) == 0 && p.split_k >= 2 && p.x_bytes < 0x80000000u && p.w_bytes < 0x80000000u &&
           (p.M % (p.Hout * p.Wout)) == 0;
}

int imd_launch_conv_img(const ConvGemmParams& p_in, hipStream_t s) {
    ConvGemmParams p = p_in;
    p.splitk_counters = nullptr;            // (the K slices always finish with the shared second launch)
    if (!imd_conv_img_supported(p))
        return imd_set_error("conv_img: unsupported problem (needs 3x3 stride 1 on a map 8 wide and <= 12 high, Cin %% 32 == 0, N %% 64 == 0, "
                             "row-major 16-bit output, split_k >= 2 with a workspace, operands < 2 GiB)");
    const int B = p.M / (p.Hout * p.Wout);
    if (p.Hout <= 8) {
        if (B > 2) return launch_img<4, 2>(p, s);
        return launch_img<2, 2>(p, s);
    }
    if (B > 2) return launch_img<4, 3>(p, s);
    return launch_img<2, 3>(p, s);
}
