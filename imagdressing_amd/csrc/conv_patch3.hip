// 3x3 / stride-1 convolution, halo patch in LDS, 8 x 16 output pixels x 160 channels per workgroup (gfx950; tile config 22, round 4).
//
// Why another tile: the UNet's level-0 convolutions write N = 320 channels.  With conv_patch.hip's 128-channel tiles that is 2.5 tiles:
// every third workgroup owns 64 valid channels, two of its four waves idle, and a sixth of the launch's wave slots does nothing
// ("narrow last channel tile").  N = 320 = 2 x 160, and the bench batch's 64 x 64 maps give 256 pixel tiles of
// 8 x 16: 512 workgroups of 128 pixels x 160 channels fill the chip's 512 slots (two 55 KB workgroups per CU) exactly, with no idle
// wave and no padded MFMA.
//
//   workgroup: 4 waves, wave w = pixels [32 w, 32 w + 32) (two image rows of the tile) x ALL 160 channels = acc[5] f32x16 (80 VGPRs):
//              per 16-deep slice one activation fragment feeds five MFMAs (6 ds_read_b128 per 5 MFMAs), ten MFMAs per tap and
//              barrier instead of eight;
//   LDS: halo patch (8+2) x (16+2) pixels x 32 channels = 180 rows x 64 B, double-buffered over channel chunks (2 x 12 KB), weight
//        tile of a tap 160 rows x 64 B in a ring of three (3 x 10 KB): 54 KB -> two workgroups per CU;
//   both operands by LDS-DMA with conv_patch.hip's source-side swizzle (piece c of row r at c ^ ((r >> 2) & 3)) and its round-4 loop
//   form: every LDS address a compile-time offset from a loop-invariant register, three-instruction staging pieces on running source
//   offsets (2^31 = out of range for halo pixels / channel rows past N), counted waits; fused nearest-2x upsample; K slices over
//   channel chunks (fp32 slabs + the shared finish launch); epilogue shared with conv_gemm.hip, 64 pixels at a time through LDS.
// Same K order and the same MFMA sequence per accumulator as conv_patch.hip: bit-identical results.
//
// NW = 8 (tile config 23): the same wave tile with EIGHT waves = 16 x 16 pixels x 160 channels per workgroup, one workgroup per CU.  The
// level-0 convolutions are bound by the operand stream into LDS -- 28 KB per tap round of a CU at the ~21 B/clk the LDS-DMA path
// sustains IS their 66-69 us -- and that stream is mostly weights, whose bytes per MFMA fall with the
// PIXELS a staged weight tile serves.  Here one 10 KB weight tile serves 256 pixels (154 B of staging per MFMA instead of 290), and
// N = 320 at the bench batch is 128 pixel tiles x 2 channel tiles = 256 workgroups: one per CU, no tail.  78 KB of LDS.
#include <type_traits>

#include "gemm_common.h"
#include "lds_dma.h"

namespace {

constexpr int T3W = 16, P3W = T3W + 2;
constexpr int CK3 = 32;
constexpr int BN3 = 160;
constexpr int WB3 = BN3 * 64, NWR3 = 3;        // 10 pieces per tap
constexpr int CLD3 = BN3 + 4;
constexpr int EROWS3 = 64;
template <int NW> struct T3 {
    static constexpr int TH = 2 * NW;                          // pixel rows of the tile: two per wave
    static constexpr int NPIX = (TH + 2) * P3W;                // 180 | 324 patch pixels
    static constexpr int APIECES = 3 * NW;                     // one-KB patch pieces staged per chunk (three per wave): 12 | 24 >= NPIX / 16
    static constexpr int AB = APIECES * 1024;
    static constexpr int WPW = (10 + NW - 1) / NW;             // weight pieces per wave and tap: 3 | 2 (waves that run out re-fetch piece 9)
    static constexpr int LDS = 2 * AB + NWR3 * WB3;            // 55,296 | 79,872
    static_assert(APIECES * 16 >= NPIX, "the patch must fit its pieces");
    static_assert(LDS >= EROWS3 * CLD3 * 4, "the epilogue tile must fit the main-loop LDS");
};
// MFMA column (lane & 31) -> pixel of a 2 x 16 pixel block: conv_patch.hip's permutation (conflict-free ds_read_b128 groups)
__device__ constexpr unsigned char kColPix3[32] = {0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 15, 4, 5, 6, 7,
                                                   30, 31, 16, 17, 22, 23, 24, 25, 26, 27, 28, 29, 18, 19, 20, 21};

// (launch bound 2 also for NW = 8, which runs one workgroup per CU: under the resulting 128-register cap hipcc fits the eight-wave body in 100 VGPRs
// with NO scratch (-Rpass-analysis=kernel-resource-usage); with the cap lifted it takes 198 for the same loop -- the shipped, measured code is kept)
template <bool F16, int NW>
__global__ __launch_bounds__(NW * 64, 2) void conv3x3_patch3_kernel(const ConvGemmParams p) {
    using E = El<F16>;
    using T = T3<NW>;
    constexpr int T3H = T::TH, NPIX3 = T::NPIX, AB3 = T::AB, WPW = T::WPW, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hi = lane >> 5, col = lane & 31;
    const int cpix = kColPix3[col];

    const int H = p.Hout, W = p.Wout;          // output map = logical input map (fused nearest-2x upsample: twice the stored input)
    const int tiles_x = (W + T3W - 1) / T3W, tiles_y = (H + T3H - 1) / T3H;      // ragged maps: tiles hang over the edge (zeros in, no stores out)
    const int n_tiles = (p.N + BN3 - 1) / BN3;
    int bid, tile_n;
    xcd_tile_order(p.flags, (int)(gridDim.x / n_tiles), n_tiles, bid, tile_n);  // bid = pixel-tile index
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int y0 = ty * T3H, x0 = tx * T3W, n0 = tile_n * BN3;

    const int nchunks = p.Cin / CK3;
    const int split = blockIdx.y;
    const int per = (nchunks + p.split_k - 1) / p.split_k;
    const int c_begin = split * per;
    const int c_end = min(nchunks, c_begin + per);
    const int total = (c_end - c_begin) * 9;

    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    const v4i_t dx = raw_rsrc(p.x, p.x_bytes), dw = raw_rsrc(p.w, p.w_bytes);
    constexpr uint32_t FAR = 0x80000000u;      // out of range, and still out of range after the loop's running adds (operands < 2 GiB)
    uint32_t acur[3], wcur[WPW];               // running source offsets: patch pieces of the NEXT chunk to stage, weight pieces of the next tap
    uint32_t adst[3], wdst[WPW];               // LDS byte addresses of the pieces inside patch buffer 0 / ring slot 0 (wave-uniform)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int slot = (wv * 3 + i) * 64 + lane, pp = slot >> 2, piece = (slot & 3) ^ ((pp >> 2) & 3);
        acur[i] = FAR;
        if (pp < NPIX3) {
            const int iy = y0 - 1 + pp / P3W, ix = x0 - 1 + pp % P3W;      // logical pixel; the zero halo is applied AFTER the upsample
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
                acur[i] = (uint32_t)(((b * p.Hin + sy) * p.Win + sx) * p.x_pix_stride + piece * 8 + c_begin * CK3) * 2u;
            }
        }
        adst[i] = smem_base + (uint32_t)((wv * 3 + i) * 1024);
    }
    // ten weight pieces per tap over NW waves: wave w issues pieces w, w + NW ... clamped to 9 -- a wave that runs out re-fetches piece 9 (the
    // same bytes to the same place, a benign duplicate), so that every wave issues exactly WPW pieces per tap and the counted waits are uniform
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int pc = min(wv + NW * i, 9);
        const int slot = pc * 64 + lane, row = slot >> 2, piece = (slot & 3) ^ ((row >> 2) & 3);
        wcur[i] = (n0 + row < p.N) ? (uint32_t)(((size_t)(n0 + row) * p.K + piece * 8 + c_begin * CK3) * 2) : FAR;
        wdst[i] = smem_base + (uint32_t)(2 * AB3 + pc * 1024);
    }
    const uint32_t w_tap = (uint32_t)(p.Cin * 2), w_chunk = (uint32_t)(CK3 * 2) - 8u * w_tap;      // next tap / tap 8 -> tap 0 of the next chunk
    auto dma_patch = [&](auto buf_c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            asm volatile("s_add_u32 m0, %1, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                         : : "v"(acur[i]), "s"(adst[i]), "s"(dx), "n"(decltype(buf_c)::value * AB3) : "memory", "scc");
            acur[i] += (uint32_t)(CK3 * 2);
        }
    };
    auto dma_w = [&](auto ring_c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            asm volatile("s_add_u32 m0, %1, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                         : : "v"(wcur[i]), "s"(wdst[i]), "s"(dw), "n"(decltype(ring_c)::value * WB3) : "memory", "scc");
            wcur[i] += w_tap;
        }
    };

    f32x16 acc[5];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // fragment addresses (bytes inside a ring slot / a patch buffer), loop-invariant; the 16-deep slice kk = 1 is the same address ^ 32.
    // Weight row a * 32 + col has the swizzle term of row col (a * 32 / 4 is a multiple of 4), so block a is a compile-time offset.
    const int w_fr = col * 64 + ((hi ^ ((col >> 2) & 3)) << 4);
    int xa[9];
    {
        const int q = 32 * wave + cpix, r0 = (q / T3W) * P3W + (q % T3W);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int rw = r0 + (t / 3) * P3W + (t % 3);
            xa[t] = rw * 64 + ((hi ^ ((rw >> 2) & 3)) << 4);
        }
    }
    const std::integral_constant<int, 0> i0{}; const std::integral_constant<int, 1> i1{}; const std::integral_constant<int, 2> i2{};
    if (total > 0) {
        dma_patch(i0);
        dma_w(i0);
        dma_w(i1);
    }
    dma_wait();
    __syncthreads();
    auto chunk = [&](auto ab_c) __attribute__((always_inline)) {       // one 32-channel chunk out of patch buffer ab_c
        constexpr int AB = decltype(ab_c)::value;
        const char* As = smem + AB * AB3;
#pragma unroll
        for (int t = 0; t < 9; ++t) {                  // 9 taps = 3 turns of the weight ring: ring slots are compile-time
            if (t % 3 == 0) dma_w(i2); else if (t % 3 == 1) dma_w(i0); else dma_w(i1);
            if (t == 6) {                              // (the tap just staged was tap 8: the next one is tap 0 of the next chunk)
#pragma unroll
                for (int i = 0; i < WPW; ++i) wcur[i] += w_chunk - w_tap;
            }
            if (t == 5) dma_patch(std::integral_constant<int, AB ^ 1>{});          // (always 3 pieces: the counted waits rely on it)
            const char* Ws = smem + 2 * AB3 + (t % 3) * WB3;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const uint4 xf = *reinterpret_cast<const uint4*>(As + (xa[t] ^ (kk * 32)));
                uint4 wf[5];
#pragma unroll
                for (int a = 0; a < 5; ++a) wf[a] = *reinterpret_cast<const uint4*>(Ws + a * 2048 + (w_fr ^ (kk * 32)));
#pragma unroll
                for (int a = 0; a < 5; ++a) acc[a] = E::mfma(wf[a], xf, acc[a]);
            }
            // the next tap's weight pieces have landed: everything but this tap's WPW pieces (and, at taps 5 and 6, the three patch
            // pieces issued behind them at tap 5) may stay in flight
            if (t == 5 || t == 6) dma_wait_keep_n<WPW + 3>(); else dma_wait_keep_n<WPW>();
            __syncthreads();
        }
    };
    const int nch = c_end - c_begin;
    int cc = 0;
#pragma unroll 1
    for (; cc + 2 <= nch; cc += 2) { chunk(i0); chunk(i1); }
    if (cc < nch) chunk(i0);
    dma_wait();                  // pieces staged past the end are still landing: the epilogue reuses this LDS
    __syncthreads();

    // ---- epilogue (conv_gemm.hip's scheme): 64 pixels (two waves) at a time through LDS, 8 consecutive channels per thread.  A thread keeps
    // ONE 8-channel column (tid % 20) for every pixel row it emits (row lanes tid / 20; the last NT % 20 threads idle), so that the
    // GroupNorm statistics of the output (gn_stats_out, as in conv_patch.hip) accumulate in registers: per-tile, per-group fp32
    // (sum, sum of squares) of the ROUNDED values, folded in a fixed order ----
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CPR = BN3 / 8;                   // 20 chunks per pixel row
    constexpr int RL = NT / CPR;                   // 12 | 25 row lanes
    const int HW = H * W;
    float* slab = (p.split_k > 1) ? p.splitk_ws + (size_t)split * p.M * p.N : nullptr;
    const int e_col = tid % CPR, e_rl = tid / CPR;
    const int n = n0 + e_col * 8;
    const bool want_stats = p.gn_stats_out != nullptr && slab == nullptr;
    const int cpg = want_stats ? p.N / p.gn_stats_groups : 1;
    const int st_split = min(8, (n / cpg + 1) * cpg - n);       // channels [0, split) of the chunk -> its first group
    float st[4] = {0.f, 0.f, 0.f, 0.f};
    // everything the epilogue reads from memory is requested NOW, in one burst (one workgroup per CU at NW = 8: nothing else is resident to
    // hide a load issued inside the loop): bias + per-batch vector of the thread's column once, the residual of every row it will emit
    constexpr int RPT = (EROWS3 + RL - 1) / RL;    // rows per thread and pass: 6 | 3
    const bool mine = e_rl < RL && n < p.N;
    const int nv = (n + 8 <= p.N) ? 8 : 4;
    float4 pre0 = make_float4(0, 0, 0, 0), pre1 = pre0;
    const bool use_pre = mine && slab == nullptr && (p.bias != nullptr || p.rowvec != nullptr);
    if (use_pre) load_col_addends(p, p.rowvec ? b : -1, n, nv, pre0, pre1);
    uint4 rres[NW / 2][RPT];
    const bool use_rpre = mine && slab == nullptr && p.res != nullptr;
    if (use_rpre) {
#pragma unroll
        for (int wr = 0; wr < NW / 2; ++wr)
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int row = e_rl + k * RL, q = wr * EROWS3 + row;
                const int oy = y0 + q / T3W, ox = x0 + q % T3W;
                rres[wr][k] = make_uint4(0, 0, 0, 0);
                if (row < EROWS3 && oy < H && ox < W) {
                    const bf16_t* rp = p.res + (size_t)((b * H + oy) * W + ox) * p.res_ld + n;
                    if (nv == 8) rres[wr][k] = *reinterpret_cast<const uint4*>(rp);
                    else { const uint2 t2 = *reinterpret_cast<const uint2*>(rp); rres[wr][k].x = t2.x; rres[wr][k].y = t2.y; }
                }
            }
    }
#pragma unroll
    for (int wr = 0; wr < NW / 2; ++wr) {
        if ((wave >> 1) == wr) {
            const int row_l = (wave & 1) * 32 + cpix;
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float4*>(Cs + row_l * CLD3 + a * 32 + 8 * j + 4 * hi) =
                        make_float4(acc[a][4 * j], acc[a][4 * j + 1], acc[a][4 * j + 2], acc[a][4 * j + 3]);
        }
        __syncthreads();
        if (mine) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int row = e_rl + k * RL;
                if (row >= EROWS3) continue;
                const int q = wr * EROWS3 + row;
                const int oy = y0 + q / T3W, ox = x0 + q % T3W;
                if (oy >= H || ox >= W) continue;
                const int m = (b * H + oy) * W + ox;
                const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * CLD3 + e_col * 8);
                const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * CLD3 + e_col * 8 + 4);
                if (slab) {
                    slab_store8(slab, (size_t)m * p.N + n, v0, v1, n + 8 <= p.N, false);
                } else {
                    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    epilogue8<F16>(p, v, m, n, nv, HW, use_pre, pre0, pre1, use_rpre, rres[wr][k]);
                    if (want_stats) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (e < nv) {
                                const float r = E::tof(E::fromf(v[e]));           // statistics of the STORED tensor
                                if (e < st_split) { st[0] += r; st[1] += r * r; } else { st[2] += r; st[3] += r * r; }
                            }
                        }
                    }
                }
            }
        }
        if (wr + 1 < NW / 2) __syncthreads();
    }
    if (want_stats) {
        __syncthreads();                        // the fp32 tile in LDS is dead: reuse its head for the NT x 4 partials
        float* red = reinterpret_cast<float*>(smem);
        *reinterpret_cast<float4*>(red + tid * 4) = make_float4(st[0], st[1], st[2], st[3]);
        __syncthreads();
        const int G = p.gn_stats_groups;
        if (tid < G) {                          // fixed summation order: column chunk, then row lane (deterministic)
            const int g = tid;
            float S = 0.f, Q = 0.f;
            for (int j = 0; j < CPR; ++j) {
                const int nj = n0 + 8 * j;
                if (nj >= p.N) break;
                const int gj = nj / cpg;
                if (gj == g || gj + 1 == g) {
                    const int o = (gj == g) ? 0 : 2;
                    for (int rl = 0; rl < RL; ++rl) { S += red[(j + CPR * rl) * 4 + o]; Q += red[(j + CPR * rl) * 4 + o + 1]; }
                }
            }
            const int nparts = tiles_y * tiles_x * n_tiles;
            const int part = (ty * tiles_x + tx) * n_tiles + tile_n;
            float* dst = p.gn_stats_out + (((size_t)b * nparts + part) * G + g) * 2;
            dst[0] = S; dst[1] = Q;
        }
    }
}

}  // namespace

static bool patch3_geometry(const ConvGemmParams& p, int th) {
    const bool geom = p.ups ? (p.Hout == 2 * p.Hin && p.Wout == 2 * p.Win) : (p.Hin == p.Hout && p.Win == p.Wout);
    return p.taps == 9 && p.stride == 1 && !p.pad_br_only && geom && p.Hout >= th && p.Wout >= T3W && (p.Cin % CK3) == 0 &&
           p.mode == OUT_ROWMAJOR && p.act != ACT_GEGLU && p.gn_a == nullptr && (p.x_pix_stride % 8) == 0 &&
           p.x_bytes < 0x80000000u && p.w_bytes < 0x80000000u;
}
bool imd_conv_patch3_supported(const ConvGemmParams& p) { return patch3_geometry(p, T3<4>::TH); }       // tile config 22
bool imd_conv_patch4_supported(const ConvGemmParams& p) { return patch3_geometry(p, T3<8>::TH); }       // tile config 23 (16 x 16 pixel tiles)

template <int NW>
static int launch_patch3(const ConvGemmParams& p_in, hipStream_t s, const char* what) {
    ConvGemmParams p = p_in;
    p.splitk_counters = nullptr;            // (K slices always finish with the shared second launch)
    if (p.split_k > 1) p.gn_stats_out = nullptr;      // (K slices: the statistics come from the finish launch; the dispatcher validated the request)
    if (!patch3_geometry(p, T3<NW>::TH))
        return imd_set_error("%s: unsupported geometry (needs 3x3 stride 1, H >= %d, W >= 16, Cin %% 32 == 0, row-major output, operands < 2 GiB)", what, T3<NW>::TH);
    const bool h = p.dtype == IMD_DTYPE_F16;
    typedef void (*kern_t)(const ConvGemmParams);
    const kern_t kern = h ? conv3x3_patch3_kernel<true, NW> : conv3x3_patch3_kernel<false, NW>;
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), T3<NW>::LDS, "%s")) return rc_attr;
    const int B = p.M / (p.Hout * p.Wout);
    const long blocks = (long)B * ((p.Hout + T3<NW>::TH - 1) / T3<NW>::TH) * ((p.Wout + T3W - 1) / T3W) * ((p.N + BN3 - 1) / BN3);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)p.split_k), dim3(NW * 64), T3<NW>::LDS, s, p);
    return imd_check_launch(what);
}

// statistic partials per image written through gn_stats_out by the un-split launch of tile config 22 (nw = 4) / 23 (nw = 8); 0: cannot
int imd_conv_patch3_stats_parts_of(const ConvGemmParams& p, int nw) {
    const int th = nw == 8 ? T3<8>::TH : T3<4>::TH;
    if (!patch3_geometry(p, th) || p.split_k > 1 || p.out_f32 || p.gn_stats_groups <= 0 || p.gn_stats_groups > 64 || p.N % p.gn_stats_groups ||
        (p.N / p.gn_stats_groups) < 8 || (p.N % 8))
        return 0;
    return ((p.Hout + th - 1) / th) * ((p.Wout + T3W - 1) / T3W) * ((p.N + BN3 - 1) / BN3);
}

int imd_launch_conv_patch3(const ConvGemmParams& p, hipStream_t s) { return launch_patch3<4>(p, s, "conv_patch3"); }
int imd_launch_conv_patch4(const ConvGemmParams& p, hipStream_t s) { return launch_patch3<8>(p, s, "conv_patch3 (16 x 16 pixel tiles)"); }
