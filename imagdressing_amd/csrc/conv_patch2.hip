// 3x3 / stride-1 convolution, halo patch in LDS, 16 x 16 output pixels x 128 channels per workgroup (gfx950; tile config 21, round 3).
//
// conv_patch.hip's LDS-DMA kernel with a pixel tile twice as tall.  Why: the timing probes of that kernel ("What
// bounds the LDS-DMA tile kernels") put the operand stream into LDS at ~21-27 B/clk per CU whatever it carries, and on the 64 x 64
// maps that stream is mostly WEIGHTS: every 128-pixel workgroup pulls the whole [128 channels x 9 Cin] slice (737 KB at Cin = 320)
// for 128 x 128 x 9 Cin MACs.  A 256-pixel tile halves the number of workgroups per channel tile and with it the weight bytes per
// MAC; the wave tile grows from 64 x 64 to 128 pixels x 64 channels, which also takes the LDS reads from 8 to 6 ds_read_b128 per 8
// MFMAs and the barriers from one per 8 to one per 16 MFMAs of a wave.
//
//   workgroup: 4 waves, 2 (pixel halves of 8 image rows) x 2 (channel halves); wave tile 128 x 64 = acc[2][4] f32x16 (128 VGPRs)
//   LDS: halo patch (16+2) x (16+2) pixels x 32 channels = 324 rows x 64 B, double-buffered over channel chunks (2 x 24 KB),
//        weight tile of a tap 128 rows x 64 B in a ring of three (3 x 8 KB); 72 KB -> two workgroups per CU
//   both operands by LDS-DMA, unpadded 64-byte rows with the source-side swizzle of conv_patch.hip (piece c of row r at
//   c ^ ((r >> 2) & 3)); counted waits; fused nearest-2x upsample by source-pixel map; K slices over channel chunks (fp32 slabs +
//   the shared finish launch); epilogue shared with conv_gemm.hip.  GroupNorm statistics from the epilogue are NOT produced here.
#include "gemm_common.h"
#include "lds_dma.h"

namespace {

constexpr int T2H = 16, T2W = 16;
constexpr int P2W = T2W + 2, P2H = T2H + 2;
constexpr int NPIX2 = P2H * P2W;               // 324 patch pixels
constexpr int CK2 = 32;
constexpr int BN2 = 128;
constexpr int AB2 = 24 * 1024;                 // 324 rows x 64 B = 20.25 pieces -> 24 (six per wave; rows past 323 are zero-fill)
constexpr int WB2 = 8 * 1024, NWR2 = 3;
constexpr int PATCH2_LDS = 2 * AB2 + NWR2 * WB2;      // 73,728
constexpr int CLD2 = BN2 + 4;
constexpr int EROWS2 = 64;
static_assert(PATCH2_LDS >= EROWS2 * CLD2 * 4, "the epilogue tile must fit the main-loop LDS");
// MFMA column (lane & 31) -> pixel of a 2 x 16 pixel block: conv_patch.hip's permutation (conflict-free ds_read_b128 groups)
__device__ constexpr unsigned char kColPix2[32] = {0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 15, 4, 5, 6, 7,
                                                   30, 31, 16, 17, 22, 23, 24, 25, 26, 27, 28, 29, 18, 19, 20, 21};

template <bool F16>
__global__ __launch_bounds__(256, 2) void conv3x3_patch2_kernel(const ConvGemmParams p) {
    using E = El<F16>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave >> 1) * 128;         // 2 x 2 waves, each 128 pixels (8 image rows of the tile) x 64 channels
    const int wn0 = (wave & 1) * 64;
    const int hi = lane >> 5, col = lane & 31;
    const int cpix = kColPix2[col];

    const int H = p.Hout, W = p.Wout;          // output map = logical input map (fused nearest-2x upsample: twice the stored input)
    const int tiles_x = (W + T2W - 1) / T2W, tiles_y = (H + T2H - 1) / T2H;      // ragged maps: tiles hang over the edge (zeros in, no stores out)
    const int n_tiles = (p.N + BN2 - 1) / BN2;
    int bid, tile_n;
    xcd_tile_order(p.flags, (int)(gridDim.x / n_tiles), n_tiles, bid, tile_n);  // bid = pixel-tile index
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int y0 = ty * T2H, x0 = tx * T2W, n0 = tile_n * BN2;

    const int nchunks = p.Cin / CK2;
    const int split = blockIdx.y;
    const int per = (nchunks + p.split_k - 1) / p.split_k;
    const int c_begin = split * per;
    const int c_end = min(nchunks, c_begin + per);
    const int total = max(0, c_end - c_begin) * 9;

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][bb][r] = 0.f;
    const bool wave_live = n0 + wn0 < p.N;     // (the last channel tile of N = 320 is half empty)

    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    const v4i_t dx = raw_rsrc(p.x, p.x_bytes), dw = raw_rsrc(p.w, p.w_bytes);
    uint32_t a_src[6], w_src[2];               // byte offsets of this lane's 16-byte pieces at chunk 0 / tap 0 (OOB: zeros)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int slot = (wv * 6 + i) * 64 + lane, pp = slot >> 2, piece = (slot & 3) ^ ((pp >> 2) & 3);
        a_src[i] = OOB;
        if (pp < NPIX2) {
            const int iy = y0 - 1 + pp / P2W, ix = x0 - 1 + pp % P2W;            // logical pixel; the zero halo is applied AFTER the upsample
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
                a_src[i] = (uint32_t)(((b * p.Hin + sy) * p.Win + sx) * p.x_pix_stride + piece * 8) * 2u;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int slot = (wv * 2 + i) * 64 + lane, row = slot >> 2, piece = (slot & 3) ^ ((row >> 2) & 3);
        w_src[i] = (n0 + row < p.N) ? (uint32_t)(((size_t)(n0 + row) * p.K + piece * 8) * 2) : OOB;
    }
    auto dma_patch = [&](int c, int buf) {     // (always six pieces per wave, zeros past the last chunk: the counted waits rely on it)
#pragma unroll
        for (int i = 0; i < 6; ++i)
            dma16(dx, smem_base + buf * AB2 + (wv * 6 + i) * 1024, (a_src[i] != OOB && c < c_end) ? a_src[i] + (uint32_t)(c * CK2 * 2) : OOB);
    };
    auto dma_w = [&](int it, int ring) {
        const int cq = it / 9;
        const uint32_t koff = (uint32_t)(((it - cq * 9) * p.Cin + (c_begin + cq) * CK2) * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            dma16(dw, smem_base + 2 * AB2 + ring * WB2 + (wv * 2 + i) * 1024, (w_src[i] != OOB && it < total) ? w_src[i] + koff : OOB);
    };
    int w_fr[2], a_row[4];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int row = wn0 + a * 32 + col;
        w_fr[a] = row * 64 + ((hi ^ ((row >> 2) & 3)) << 4);                     // 16-deep slice kk = 0; kk = 1 is the same address ^ 32
    }
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
        const int q = wm0 + bb * 32 + cpix;
        a_row[bb] = (q / T2W) * P2W + (q % T2W);
    }
    if (total > 0) {
        dma_patch(c_begin, 0);
        dma_w(0, 0);
        dma_w(1, 1);
    }
    dma_wait();
    __syncthreads();
#pragma unroll 1
    for (int cc = 0; cc < c_end - c_begin; ++cc) {
        const int ab = cc & 1;
#pragma unroll
        for (int t = 0; t < 9; ++t) {          // 9 taps = 3 turns of the weight ring: ring slots are compile-time
            dma_w(cc * 9 + t + 2, (t + 2) % 3);
            if (t == 5) dma_patch(c_begin + cc + 1, ab ^ 1);
            if (wave_live) {
                const char* As = smem + ab * AB2;
                const char* Ws = smem + 2 * AB2 + (t % 3) * WB2;
                int xa[4];
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int rw = a_row[bb] + (t / 3) * P2W + (t % 3);
                    xa[bb] = rw * 64 + ((hi ^ ((rw >> 2) & 3)) << 4);
                }
#pragma unroll
                for (int kk = 0; kk < CK2 / 16; ++kk) {
                    uint4 wf[2], xf[4];
#pragma unroll
                    for (int a = 0; a < 2; ++a) wf[a] = *reinterpret_cast<const uint4*>(Ws + (w_fr[a] ^ (kk * 32)));
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) xf[bb] = *reinterpret_cast<const uint4*>(As + (xa[bb] ^ (kk * 32)));
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                        for (int a = 0; a < 2; ++a) acc[a][bb] = E::mfma(wf[a], xf[bb], acc[a][bb]);
                }
            }
            // the next tap's weight pieces have landed: everything but this tap's two pieces (and, at taps 5 and 6, the six patch
            // pieces issued behind them at tap 5) may stay in flight
            if (t == 5 || t == 6) dma_wait_keep_n<8>(); else dma_wait_keep_n<2>();
            __syncthreads();
        }
    }
    dma_wait();                  // zero-fill pieces past the end are still landing: the epilogue reuses this LDS
    __syncthreads();

    // ---- epilogue (conv_gemm.hip's scheme): 64 pixels at a time through LDS ----
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CPR = BN2 / 8;
    constexpr int CHUNKS = EROWS2 * CPR;
    const int HW = H * W;
    float* slab = (p.split_k > 1) ? p.splitk_ws + (size_t)split * p.M * p.N : nullptr;
    float4 col_pre0 = make_float4(0, 0, 0, 0), col_pre1 = col_pre0;
    bool use_col_pre = false;
    if (slab == nullptr && (p.bias || p.rowvec)) {          // bias / time-embedding vector of this thread's 8 columns, fetched once: the tile lies in ONE image
        const int n = n0 + (tid % CPR) * 8;
        if (n < p.N) {
            load_col_addends(p, p.rowvec ? b : -1, n, (n + 8 <= p.N) ? 8 : 4, col_pre0, col_pre1);
            use_col_pre = true;
        }
    }
#pragma unroll
    for (int wr = 0; wr < 4; ++wr) {           // pixels [64 wr, 64 wr + 64): wave row wr >> 1, its fragments bb = 2 (wr & 1), + 1
        if ((wave >> 1) == (wr >> 1)) {
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int bb = 2 * (wr & 1) + b2;
                        float* dst = Cs + (b2 * 32 + cpix) * CLD2 + wn0 + a * 32 + 8 * j + 4 * hi;
                        *reinterpret_cast<float4*>(dst) = make_float4(acc[a][bb][4 * j], acc[a][bb][4 * j + 1], acc[a][bb][4 * j + 2], acc[a][bb][4 * j + 3]);
                    }
        }
        __syncthreads();
        for (int ch = tid; ch < CHUNKS; ch += 256) {
            const int row = ch / CPR, cc = (ch - row * CPR) * 8;
            const int q = wr * EROWS2 + row;
            const int oy = y0 + q / T2W, ox = x0 + q % T2W;
            const int m = (b * H + oy) * W + ox;
            const int n = n0 + cc;
            if (n >= p.N || oy >= H || ox >= W) continue;
            const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * CLD2 + cc);
            const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * CLD2 + cc + 4);
            if (slab) {
                slab_store8(slab, (size_t)m * p.N + n, v0, v1, n + 8 <= p.N, false);
            } else {
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                epilogue8<F16>(p, v, m, n, (n + 8 <= p.N) ? 8 : 4, HW, use_col_pre, col_pre0, col_pre1);
            }
        }
        if (wr < 3) __syncthreads();
    }
}

}  // namespace

bool imd_conv_patch2_supported(const ConvGemmParams& p) {
    const bool geom = p.ups ? (p.Hout == 2 * p.Hin && p.Wout == 2 * p.Win) : (p.Hin == p.Hout && p.Win == p.Wout);
    return p.taps == 9 && p.stride == 1 && !p.pad_br_only && geom && p.Hout >= T2H && p.Wout >= T2W && (p.Cin % CK2) == 0 && p.K == 9 * p.Cin &&
           p.mode == OUT_ROWMAJOR && p.act != ACT_GEGLU && p.gn_a == nullptr && (p.x_pix_stride % 8) == 0;
}

int imd_launch_conv_patch2(const ConvGemmParams& p_in, hipStream_t s) {
    if (!imd_conv_patch2_supported(p_in))
        return imd_set_error("conv_patch2: unsupported geometry (needs 3x3 stride 1, H >= 16, W >= 16, Cin %% 32 == 0, no fused GroupNorm prologue)");
    ConvGemmParams p = p_in;
    p.splitk_counters = nullptr;               // K slices are summed by the shared finish launch
    p.gn_stats_out = nullptr;
    const bool h = p.dtype == IMD_DTYPE_F16;
    typedef void (*kern_t)(const ConvGemmParams);
    const kern_t kern = h ? conv3x3_patch2_kernel<true> : conv3x3_patch2_kernel<false>;
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), PATCH2_LDS, "conv_patch2")) return rc_attr;
    const int B = p.M / (p.Hout * p.Wout);
    const long blocks = (long)B * ((p.Hout + T2H - 1) / T2H) * ((p.Wout + T2W - 1) / T2W) * ((p.N + BN2 - 1) / BN2);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)p.split_k), dim3(256), PATCH2_LDS, s, p);
    return imd_check_launch("conv_patch2");
}
