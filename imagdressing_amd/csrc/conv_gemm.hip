// Implicit-GEMM convolution / linear layer for gfx950 (bf16 / fp16 MFMA, fp32 accumulate).
//
//   out[m, n] = epilogue( sum_k A(m, k) * W[n, k] )        m = output pixel / token, n = channel
//
// * A is gathered on the fly from an NHWC activation tensor (3x3 / 1x1 taps, stride 1|2, zero
//   padding, optional fused nearest-2x upsample of the input) -- for a plain linear layer it is
//   simply the [M, K] row-major matrix (taps = 1, H = W = 1).
// * W is the pre-packed weight [N][K] with k = tap * Cin + ci  (torch Linear layout; conv weights
//   are repacked once at load time from [Cout][Cin][3][3] to [Cout][ky][kx][Cin]).
// * Operand tiles are fetched with raw BUFFER loads: the padding halo, the M / N / K tails and
//   everything else that must read as zero simply gets an out-of-range offset and the hardware
//   returns 0 -- no branches, no exec masking, 32-bit address arithmetic only.
// * Tiles are staged global -> registers -> LDS (double buffered, one barrier per K tile), rows
//   padded by 16 B so that the 16-lane groups of a ds_read_b128 hit 16 distinct 16-B bank slots.
// * The MFMA is issued "swapped" (A-operand = weight rows, B-operand = activation rows): a lane's
//   accumulator registers hold 4 consecutive channels of ONE output row.
// * Epilogue: the fp32 accumulator tile goes through LDS once, after which every thread owns
//   8 consecutive channels of a row: bias / per-batch time-embedding vector / scale / residual /
//   SiLU / GELU / GEGLU are applied on 8-wide vectors with all their loads in flight together, and
//   the result leaves in coalesced 16-byte stores (or, for the head-split mode, straight into the
//   per-head Q / K / V^T layouts the attention kernel consumes).
// * Split-K (small-M layers, e.g. the 8x8 and 16x16 ResNet convs at K = 11520..23040): each K
//   slice writes its fp32 partial tile to a slab; a second tiny kernel sums the slabs in a fixed
//   order (deterministic) and runs the same epilogue.
//
// Reference arithmetic this replaces (diffusers==0.24.0, un-vendored; call sites
// /root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline.py:466,499,511):
// ResnetBlock2D conv1/conv2/conv_shortcut, Down/Upsample2D conv, Transformer2DModel proj_in/
// proj_out, Attention to_q/to_k/to_v/to_out, GEGLU FeedForward, TimestepEmbedding, and the
// to_k_ref/to_v_ref garment projections of adapter/attention_processor.py:600-601.
#include "gemm_common.h"

namespace {

template <int BM, int BN, int BK, int WAVES_M> struct TileCfg {
    static constexpr int STRIDE = BK * 2 + 16;                 // bytes per LDS operand row
    static constexpr int MAIN = 2 * (BM + BN) * STRIDE;        // double-buffered operand tiles
    static constexpr int CLD = BN + 4;                         // fp32 epilogue tile leading dim (floats)
    static constexpr int EROWS = BM / WAVES_M;                 // the epilogue goes through LDS one wave-row group at a time
    static constexpr int EPI = EROWS * CLD * 4;
    static constexpr int LDS = MAIN > EPI ? MAIN : EPI;
};

template <bool F16, int BM, int BN, int BK, int WAVES_M, int WAVES_N, int DEPTH>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, (WAVES_M * WAVES_N == 4) ? 2 : 1) void conv_gemm_kernel(const ConvGemmParams p) {
    constexpr int NT = WAVES_M * WAVES_N * 64;      // threads per workgroup: 256 (4 waves) or 512 (8 waves: the 256-row tiles)
    using E = El<F16>;
    using T = TileCfg<BM, BN, BK, WAVES_M>;
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = BN / WAVES_N / 32;
    constexpr int STRIDE = T::STRIDE;
    constexpr int VPR = BK / 8;                // 16-B vectors per row
    constexpr int RSTEP = NT / VPR;            // rows covered by one pass of the workgroup's threads
    constexpr int A_VECS = BM / RSTEP;
    constexpr int W_VECS = BN / RSTEP;
    constexpr int BUF = (BM + BN) * STRIDE;
    static_assert(TM >= 1 && TN >= 1 && A_VECS >= 1 && W_VECS >= 1, "tile config");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * (TM * 32);
    const int wn0 = (wave % WAVES_N) * (TN * 32);

    const int n_tiles = (p.N + BN - 1) / BN;
    int tile_m, tile_n;
    xcd_tile_order(p.flags, (p.M + BM - 1) / BM, n_tiles, tile_m, tile_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // K range of this split (whole BK tiles)
    const int nk_total = (p.K + BK - 1) / BK;
    const int split = blockIdx.y;
    const int per = (nk_total + p.split_k - 1) / p.split_k;
    const int kt_begin = split * per;
    const int kt_end = min(nk_total, kt_begin + per);

    // staging map.  BK = 32 (80-byte LDS rows, 4 vectors per row): 16 consecutive lanes that write rows R .. R+3 put three of
    // their sixteen 16-byte pieces on an occupied bank slot (PMC: 30 % of the LDS cycles of the halo-patch kernel were bank
    // conflicts of exactly this pattern); rows R, R+4, R+8, R+12 do not (5 r mod 16 = 0, 4, 8, 12).
    const int kc = tid % VPR;          // this thread's 16-B column inside a K tile
    const int r0 = (VPR == 4) ? ((tid >> 6) * 16 + ((tid & 63) >> 4) + 4 * ((tid >> 2) & 3)) : tid / VPR;     // first row it stages

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- per-row gather setup (rows are fixed for the whole K loop) ----
    const int HWo = p.Hout * p.Wout;
    const int pad = (p.taps == 9 && !p.pad_br_only) ? 1 : 0;
    const int Hl = p.ups ? p.Hin * 2 : p.Hin;   // logical (post-upsample) input size
    const int Wl = p.ups ? p.Win * 2 : p.Win;
    int a_base[A_VECS];    // pixel index of (b, 0, 0), or -1 when the row is out of range
    int a_yx[A_VECS];      // (iy0 << 16) | (ix0 & 0xffff), top-left tap position
    int a_lin[A_VECS];     // element offset of the top-left tap (non-upsampled path: tap (ky,kx) adds a uniform delta)
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) {
        const int m = m0 + r0 + i * RSTEP;
        if (m < p.M) {
            const int b = m / HWo;
            const int rem = m - b * HWo;
            const int oy = rem / p.Wout;
            const int ox = rem - oy * p.Wout;
            const int iy0 = oy * p.stride - pad, ix0 = ox * p.stride - pad;
            a_base[i] = b * p.Hin * p.Win;
            a_yx[i] = (int)(((unsigned)iy0) << 16) | (ix0 & 0xffff);
            a_lin[i] = (a_base[i] + iy0 * p.Win + ix0) * p.x_pix_stride;
        } else {
            a_base[i] = -1;
            a_yx[i] = 0;
            a_lin[i] = 0;
        }
    }
    uint32_t w_off[W_VECS];            // byte offset of (row n, column kc*8) of W, or OOB
#pragma unroll
    for (int i = 0; i < W_VECS; ++i) {
        const int n = n0 + r0 + i * RSTEP;
        w_off[i] = (n < p.N) ? (uint32_t)(((size_t)n * p.K + kc * 8) * 2) : OOB;
    }

    // DEPTH register staging sets: tiles t+1 .. t+DEPTH are in flight while tile t is multiplied (global / L2
    // latency under load is ~2 K-tile times of the big tiles and many more of the small ones; a single set stalls
    // every iteration on vmcnt)
    uint4 a_r[DEPTH][A_VECS], w_r[DEPTH][W_VECS];

    // K-tile order.  Default: k = tile * BK (tap-major, channels inner).  "tap-inner" (3x3 convs with Cin % BK == 0):
    // tile j -> channel chunk j / 9, tap j % 9, so CONSECUTIVE tiles are neighbouring taps of the same channels and
    // re-read (shifted by one pixel) the activation lines the previous tile just pulled into the L1.
    const bool tap_inner = (p.flags & 1) != 0;
    const bool w_bypass_l1 = (p.flags & 2) != 0;
    // tap of a k index without an integer division by the runtime Cin (~35 VALU instructions per K tile and wave, 40 % of the
    // kernel's VALU work on the 8x8 / 16x16 maps where a wave has only 4 MFMAs per tile): (k + 0.5) / Cin is at least 0.5 / Cin
    // away from an integer, far more than fp32 rounding for k < 2^24
    const float inv_cin = 1.0f / (float)p.Cin;
    auto load_tile = [&](int kt, uint4 (&a_reg)[A_VECS], uint4 (&w_reg)[W_VECS]) {
        int kbase = kt * BK;
        int tap = 0, ci = 0;
        if (tap_inner) {                  // (Cin % BK == 0: a tile never straddles two taps)
            const int c = kt / 9;
            tap = kt - 9 * c; ci = c * BK + kc * 8;
            kbase = tap * p.Cin + c * BK;
        }
        const int k = kbase + kc * 8;
        const bool kv = (k < p.K) && (kt < nk_total);
        if (!tap_inner) {
            ci = k;
            if (p.taps == 9) { tap = (int)(((float)k + 0.5f) * inv_cin); ci = k - tap * p.Cin; }
        }
        const int ky = tap / 3, kx = tap - ky * 3;
        const int tap_delta = (ky * p.Win + kx) * p.x_pix_stride + ci;     // same for every row of the tile
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
            int iy = (a_yx[i] >> 16) + ky;
            int ix = (int)(short)(a_yx[i] & 0xffff) + kx;
            const bool ok = kv && a_base[i] >= 0 && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
            uint32_t off;
            if (p.ups) {          // fused nearest-2x upsample: source pixel = logical pixel >> 1 (not linear in the tap)
                off = (uint32_t)((a_base[i] + (iy >> 1) * p.Win + (ix >> 1)) * p.x_pix_stride + ci) * 2u;
            } else {
                off = (uint32_t)(a_lin[i] + tap_delta) * 2u;
            }
            a_reg[i] = buf_load16(rs_x, ok ? off : OOB);
        }
#pragma unroll
        for (int i = 0; i < W_VECS; ++i) {
            const uint32_t off = (kv && w_off[i] != OOB) ? w_off[i] + (uint32_t)(kbase * 2) : OOB;
            w_reg[i] = w_bypass_l1 ? buf_load16_nl1(rs_w, off) : buf_load16(rs_w, off);
        }
    };
    auto store_a = [&](int buf, const uint4 (&a_reg)[A_VECS]) {
        char* As = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < A_VECS; ++i)
            *reinterpret_cast<uint4*>(As + (r0 + i * RSTEP) * STRIDE + kc * 16) = a_reg[i];
    };
    auto store_w = [&](int buf, const uint4 (&w_reg)[W_VECS]) {
        char* Ws = smem + buf * BUF + BM * STRIDE;
#pragma unroll
        for (int i = 0; i < W_VECS; ++i)
            *reinterpret_cast<uint4*>(Ws + (r0 + i * RSTEP) * STRIDE + kc * 16) = w_reg[i];
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int frag_off = (lane & 31) * STRIDE + (lane >> 5) * 16;
    auto mma = [&](int buf, int kk) {           // one 16-deep k step of the LDS tile `buf`
        const char* As = smem + buf * BUF;
        const char* Ws = As + BM * STRIDE;
        uint4 wf[TN], xf[TM];
#pragma unroll
        for (int a = 0; a < TN; ++a)
            wf[a] = *reinterpret_cast<const uint4*>(Ws + (wn0 + a * 32) * STRIDE + frag_off + kk * 32);
#pragma unroll
        for (int b = 0; b < TM; ++b)
            xf[b] = *reinterpret_cast<const uint4*>(As + (wm0 + b * 32) * STRIDE + frag_off + kk * 32);
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) acc[a][b] = E::mfma(wf[a], xf[b], acc[a][b]);
    };
    auto mma_tile = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) mma(buf, kk);
    };

    // Software pipeline (tiles past the K range read as zero, so the steady state needs no guards):
    // LDS buffer t&1 holds tile t (being multiplied); register set (t+i) % DEPTH holds tile t+i, i = 1..DEPTH-1;
    // iteration t refills set t % DEPTH with tile t+DEPTH, multiplies, then parks tile t+1 in the other LDS buffer.
    // Unrolled by DEPTH (even) so that every set / buffer index is a compile-time constant.
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) load_tile(kt_begin + i, a_r[i], w_r[i]);
    store_a(0, a_r[0]);
    store_w(0, w_r[0]);
    __syncthreads();

    for (int kt = kt_begin; kt < kt_end; kt += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            if (i == 0 || kt + i < kt_end) {
                load_tile(kt + i + DEPTH, a_r[i], w_r[i]);
                mma_tile(i & 1);
                store_a((i + 1) & 1, a_r[(i + 1) % DEPTH]);
                store_w((i + 1) & 1, w_r[(i + 1) % DEPTH]);
            }
            __syncthreads();
        }
    }

    // ---- epilogue, one wave-row group (EROWS rows of the tile) at a time: accumulators -> LDS (fp32; lane owns
    // row m = col, register quads own 4 consecutive channels), then every thread emits 8 consecutive channels
    // of a row.  Staging only EROWS rows keeps the LDS footprint of the small-BK configurations low.
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CPR = BN / 8;                 // chunks per row
    constexpr int CHUNKS = T::EROWS * CPR;
    const bool colmajor = p.mode == OUT_HEADS;  // consecutive threads -> consecutive tokens (coalesces V^T / Q / K rows)
    float* slab = (p.split_k > 1) ? p.splitk_ws + (size_t)split * p.M * p.N : nullptr;
    // bias / per-batch vector of this thread's 8 columns, fetched once (row-major epilogue: the column block of a thread
    // does not change from row to row when the thread count is a multiple of the chunks per row)
    constexpr bool COLS_FIXED = (NT % CPR) == 0;
    float4 col_pre0 = make_float4(0, 0, 0, 0), col_pre1 = col_pre0;
    bool use_col_pre = false;
    if (COLS_FIXED && !colmajor && slab == nullptr && (p.bias || p.rowvec)) {
        const int bi_lo = m0 / HWo, bi_hi = (min(m0 + BM, p.M) - 1) / HWo;
        const int n = n0 + (tid % CPR) * 8;
        if ((p.rowvec == nullptr || bi_lo == bi_hi) && n < p.N) {
            load_col_addends(p, p.rowvec ? bi_lo : -1, n, (n + 8 <= p.N) ? 8 : 4, col_pre0, col_pre1);
            use_col_pre = true;
        }
    }
    // GroupNorm statistics of the output tile (gn_stats_out; round 6: also from this kernel -- conv_in, the stride-2 downsampler of the 64x64 level, the
    // projections that land here): conv_patch.hip's scheme.  A thread keeps the same 8 channels for every row it emits (COLS_FIXED), so it accumulates
    // (sum, sum of squares) of the values it STORES (rounded to the element type) for the at most two groups those channels belong to; the launcher only
    // asks when the tile's rows lie in one image (H W % BM == 0) and a group has >= 8 channels.
    const bool want_stats = COLS_FIXED && p.gn_stats_out != nullptr && slab == nullptr && !colmajor;
    const int st_cpg = want_stats ? p.N / p.gn_stats_groups : 1;
    const int st_n = n0 + (tid % CPR) * 8;
    const int st_split = min(8, (st_n / st_cpg + 1) * st_cpg - st_n);       // channels [0, split) of the chunk -> its first group
    float st[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int wr = 0; wr < WAVES_M; ++wr) {
        if (wave / WAVES_N == wr) {
            const int hi = lane >> 5;
            const int col = lane & 31;
#pragma unroll
            for (int b = 0; b < TM; ++b)
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* dst = Cs + (b * 32 + col) * T::CLD + wn0 + a * 32 + 8 * j + 4 * hi;
                        *reinterpret_cast<float4*>(dst) = make_float4(acc[a][b][4 * j], acc[a][b][4 * j + 1], acc[a][b][4 * j + 2], acc[a][b][4 * j + 3]);
                    }
        }
        __syncthreads();
        for (int c = tid; c < CHUNKS; c += NT) {
            int row, cc;
            if (colmajor) { cc = (c / T::EROWS) * 8; row = c - (c / T::EROWS) * T::EROWS; }
            else { row = c / CPR; cc = (c - row * CPR) * 8; }
            const int m = m0 + wr * T::EROWS + row, n = n0 + cc;
            if (m >= p.M || n >= p.N) continue;
            const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * T::CLD + cc);
            const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * T::CLD + cc + 4);
            if (slab) {                         // raw fp32 partial tile -> slab [split][M][N]
                slab_store8(slab, (size_t)m * p.N + n, v0, v1, n + 8 <= p.N, p.splitk_counters != nullptr);
            } else {
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                const int nv = (n + 8 <= p.N) ? 8 : 4;
                epilogue8<F16>(p, v, m, n, nv, HWo, use_col_pre, col_pre0, col_pre1);
                if (want_stats) {               // v now holds the final values (bias / vector / residual / activation applied)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (e < nv) {
                            const float r = E::tof(E::fromf(v[e]));
                            if (e < st_split) { st[0] += r; st[1] += r * r; }
                            else { st[2] += r; st[3] += r * r; }
                        }
                    }
                }
            }
        }
        if (wr + 1 < WAVES_M) __syncthreads();
    }
    if (want_stats) {
        __syncthreads();                        // the fp32 tile in LDS is dead: its head takes the NT x 4 partials
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
                const int gj = nj / st_cpg;
                if (gj == g || gj + 1 == g) {
                    const int o = (gj == g) ? 0 : 2;
                    for (int rl = 0; rl < NT / CPR; ++rl) { S += red[(j + CPR * rl) * 4 + o]; Q += red[(j + CPR * rl) * 4 + o + 1]; }
                }
            }
            const int parts_m = HWo / BM;
            const int nparts = parts_m * n_tiles;
            const int part = (tile_m % parts_m) * n_tiles + tile_n;
            float* dst = p.gn_stats_out + (((size_t)(m0 / HWo) * nparts + part) * G + g) * 2;
            dst[0] = S; dst[1] = Q;
        }
    }
    // K slices summed in-kernel by the tile's last-arriving workgroup (see splitk_last_arrival) instead of by a second launch
    if (slab != nullptr && p.splitk_counters != nullptr) {
        if (splitk_last_arrival(p.splitk_counters, tile_m * n_tiles + tile_n, p.split_k, tid)) {
            for (int c = tid; c < BM * CPR; c += NT) {
                const int row = c / CPR, cc = (c - row * CPR) * 8;
                const int m = m0 + row, n = n0 + cc;
                if (m >= p.M || n >= p.N) continue;
                const int nv = (n + 8 <= p.N) ? 8 : 4;
                float v[8];
                splitk_sum8(p, m, n, nv, v);
                epilogue8<F16>(p, v, m, n, nv, HWo);
            }
        }
    }
}

// sum the split-K slabs in a fixed order and run the epilogue (row-major outputs only)
template <bool F16>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const ConvGemmParams p) {
    const int HWo = p.Hout * p.Wout;
    const int cpr = (p.N + 7) / 8;
    const long total = (long)p.M * cpr;
    const size_t slab = (size_t)p.M * p.N;
    for (long c = blockIdx.x * 256L + threadIdx.x; c < total; c += (long)gridDim.x * 256L) {
        const int m = (int)(c / cpr);
        const int n = (int)(c - (long)m * cpr) * 8;
        const int nv = (n + 8 <= p.N) ? 8 : 4;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < p.split_k; ++s) {
            const float* src = p.splitk_ws + s * slab + (size_t)m * p.N + n;
            const float4 a = *reinterpret_cast<const float4*>(src);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
            if (nv == 8) {
                const float4 b = *reinterpret_cast<const float4*>(src + 4);
                v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
            }
        }
        epilogue8<F16>(p, v, m, n, nv, HWo);
    }
}

// The same finish launch when the output feeds a GroupNorm (gn_stats_out, round 4): besides summing the slices and running the epilogue
// it writes the GroupNorm statistics of the tensor it stores -- per (image, pixel part, group) fp32 (sum, sum of squares) of the
// ROUNDED values, the layout imd_groupnorm consumes through `nparts` -- so the next group_norm skips its statistics launch (the
// 16 x 16 / 8 x 8 levels: 19 of the 46 statistics launches of a denoising step).  norm.hip's scheme: a thread owns a FIXED 8-channel
// column and walks pixel rows, so its two (group) accumulator pairs stay in registers; deterministic (no atomics).
constexpr int FS_THREADS = 320;          // 5 waves: 40 / 80 / 160 / 320 columns divide evenly

__host__ __device__ inline int fs_rows_per_part(int B, int HW, int N) {
    const int cpr = N / 8, cols = cpr < FS_THREADS ? cpr : FS_THREADS, plan = FS_THREADS / cols;
    int rpp = (int)(((long)B * HW + 511) / 512);          // ~512 blocks; never below one pass of the block's pixel lanes
    rpp = (rpp + plan - 1) / plan * plan;
    if (rpp < plan) rpp = plan;
    return rpp;
}

template <bool F16>
__global__ __launch_bounds__(FS_THREADS) void splitk_finish_stats_kernel(const ConvGemmParams p, int rpp, int nparts) {
    using E = El<F16>;
    __shared__ float red[FS_THREADS][4];
    const int HWo = p.Hout * p.Wout;
    const int cpr = p.N / 8, cols = min(cpr, FS_THREADS), plan = FS_THREADS / cols;
    const int tid = threadIdx.x, part = blockIdx.x, b = blockIdx.y;
    const int r0 = part * rpp, r1 = min(HWo, r0 + rpp);
    const int G = p.gn_stats_groups, cpg = p.N / G;
    const int my_col = tid % cols, my_pl = tid / cols;
    const size_t slab = (size_t)p.M * p.N;
    for (int cbase = 0; cbase < cpr; cbase += cols) {
        const int vec = cbase + my_col, n = vec * 8;
        const int g0 = n / cpg;
        const int split = min(8, (g0 + 1) * cpg - n);      // channels [0, split) of the chunk -> group g0, the rest -> g0 + 1
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
        if (my_pl < plan && vec < cpr) {
            for (int row = r0 + my_pl; row < r1; row += plan) {
                const int m = b * HWo + row;
                float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int sl = 0; sl < p.split_k; ++sl) {
                    const float* src = p.splitk_ws + sl * slab + (size_t)m * p.N + n;
                    const float4 a = *reinterpret_cast<const float4*>(src);
                    const float4 c = *reinterpret_cast<const float4*>(src + 4);
                    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += c.x; v[5] += c.y; v[6] += c.z; v[7] += c.w;
                }
                epilogue8<F16>(p, v, m, n, 8, HWo);            // v: the final fp32 values; the tensor holds them rounded to 16 bits
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float r = E::tof(E::fromf(v[e]));
                    if (e < split) { s0 += r; q0 += r * r; } else { s1 += r; q1 += r * r; }
                }
            }
        }
        red[tid][0] = s0; red[tid][1] = q0; red[tid][2] = s1; red[tid][3] = q1;
        __syncthreads();
        if (tid < G) {                                          // fixed fold order: column, then pixel lane
            const int g = tid;
            int vlo = (g * cpg) / 8, vhi = ((g + 1) * cpg - 1) / 8;
            vlo = max(vlo, cbase); vhi = min(vhi, min(cpr, cbase + cols) - 1);
            float S = 0.f, Q = 0.f;
            for (int v = vlo; v <= vhi; ++v) {
                const int vg0 = (v * 8) / cpg;
                for (int pl = 0; pl < plan; ++pl) {
                    const float* e = red[pl * cols + (v - cbase)];
                    if (vg0 == g) { S += e[0]; Q += e[1]; }
                    else if (vg0 + 1 == g) { S += e[2]; Q += e[3]; }
                }
            }
            float* dst = p.gn_stats_out + (((size_t)b * nparts + part) * G + g) * 2;
            if (cbase == 0) { dst[0] = S; dst[1] = Q; } else { dst[0] += S; dst[1] += Q; }
        }
        __syncthreads();
    }
}

// ---- finish launch with GroupNorm (+ SiLU) of the output (round 6; imd_conv_gemm_params.gn_out_*) ----
// One workgroup = all pixels of one (image, group): cpg channels in units of 4 (N / G = 40 or 20 at the 16x16 / 8x8 levels), HW * cpg / 4 units
// spread over 256 threads, at most FG_MAXU per thread, kept in registers between the two phases:
//   phase 1  unit value = sum of the K slices (ascending, like splitk_finish_kernel) + bias + per-batch vector (the epilogue's first addition;
//            residual / scale / activation are refused by the launcher: ResnetBlock2D.conv1 has none), rounded to the element type -- the
//            value the separate GroupNorm launch would read back -- and its (sum, sum of squares) in fp32;
//   fold     lane-wise partials -> LDS -> thread 0 sums the 256 entries in ascending order (deterministic) -> mean, rstd;
//   phase 2  y = act(r * gamma rstd + (beta - mean gamma rstd)) exactly as gn_apply_kernel forms it, packed, stored (8 bytes per unit).
constexpr int FG_THREADS = 256, FG_MAXU = 12;

template <bool F16>
__global__ __launch_bounds__(FG_THREADS) void splitk_finish_gn_kernel(const ConvGemmParams p) {
    using E = El<F16>;
    __shared__ float red_s[FG_THREADS], red_q[FG_THREADS];
    __shared__ float s_stat[2];
    const int HWo = p.Hout * p.Wout;
    const int G = p.gn_out_groups, cpg = p.N / G, upr = cpg / 4;          // units (4 channels) per pixel row of this group
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int units = HWo * upr;
    const size_t slab = (size_t)p.M * p.N;
    float v[FG_MAXU][4];
    float S = 0.f, Q = 0.f;
#pragma unroll
    for (int i = 0; i < FG_MAXU; ++i) {
        const int u = tid + i * FG_THREADS;
        v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
        if (u < units) {
            const int row = u / upr, n = g * cpg + (u - row * upr) * 4;
            const int m = b * HWo + row;
            for (int sl = 0; sl < p.split_k; ++sl) {
                const float4 a = *reinterpret_cast<const float4*>(p.splitk_ws + sl * slab + (size_t)m * p.N + n);
                v[i][0] += a.x; v[i][1] += a.y; v[i][2] += a.z; v[i][3] += a.w;
            }
            float4 add = make_float4(0, 0, 0, 0);
            if (p.bias) add = *reinterpret_cast<const float4*>(p.bias + n);
            if (p.rowvec) {
                const float4 r = *reinterpret_cast<const float4*>(p.rowvec + (size_t)b * p.rowvec_stride + n);
                add.x += r.x; add.y += r.y; add.z += r.z; add.w += r.w;
            }
            v[i][0] += add.x; v[i][1] += add.y; v[i][2] += add.z; v[i][3] += add.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float r = E::tof(E::fromf(v[i][e]));
                v[i][e] = r;
                S += r; Q += r * r;
            }
        }
    }
    red_s[tid] = S; red_q[tid] = Q;
    __syncthreads();
    if (tid == 0) {
        float St = 0.f, Qt = 0.f;
        for (int k = 0; k < FG_THREADS; ++k) { St += red_s[k]; Qt += red_q[k]; }
        const float cnt = (float)HWo * (float)cpg;
        const float mean = St / cnt;
        const float var = fmaxf(Qt / cnt - mean * mean, 0.f);
        s_stat[0] = mean;
        s_stat[1] = rsqrtf(var + p.gn_out_eps);
    }
    __syncthreads();
    const float mean = s_stat[0], rstd = s_stat[1];
    bf16_t* out = reinterpret_cast<bf16_t*>(p.out);
#pragma unroll
    for (int i = 0; i < FG_MAXU; ++i) {
        const int u = tid + i * FG_THREADS;
        if (u >= units) continue;
        const int row = u / upr, n = g * cpg + (u - row * upr) * 4;
        const int m = b * HWo + row;
        const float4 ga4 = *reinterpret_cast<const float4*>(p.gn_out_gamma + n), be4 = *reinterpret_cast<const float4*>(p.gn_out_beta + n);
        const float gam[4] = {ga4.x, ga4.y, ga4.z, ga4.w}, bet[4] = {be4.x, be4.y, be4.z, be4.w};
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ga = gam[e] * rstd;
            const float sh = bet[e] - mean * ga;
            y[e] = v[i][e] * ga + sh;
            if (p.gn_out_silu) y[e] = silu_f(y[e]);
        }
        *reinterpret_cast<uint2*>(out + (size_t)m * p.out_ld + n) = make_uint2(E::pack2(y[0], y[1]), E::pack2(y[2], y[3]));
    }
}

bool splitk_gn_out_supported(const ConvGemmParams& p) {
    if (p.gn_out_gamma == nullptr || p.gn_out_beta == nullptr || p.split_k <= 1 || p.splitk_counters != nullptr || p.mode != OUT_ROWMAJOR || p.out_f32 ||
        p.act != ACT_NONE || p.res != nullptr || p.out_scale != 1.0f || p.gn_stats_out != nullptr) return false;
    const int G = p.gn_out_groups, HW = p.Hout * p.Wout;
    if (G <= 0 || G > 65535 || p.N % G || HW <= 0 || p.M % HW) return false;
    const int cpg = p.N / G;
    if ((cpg % 4) || (p.out_ld % 4) || (p.rowvec && (p.rowvec_stride % 4))) return false;
    return (long)HW * (cpg / 4) <= (long)FG_THREADS * FG_MAXU;
}

// statistic partials per image the finish launch writes (0: this problem cannot take the statistics form)
int splitk_stats_parts_of(const ConvGemmParams& p) {
    if (p.split_k <= 1 || p.splitk_counters != nullptr || p.mode != OUT_ROWMAJOR || p.out_f32 || p.act == ACT_GEGLU || (p.N % 8) ||
        p.gn_stats_groups <= 0 || p.gn_stats_groups > 64 || p.N % p.gn_stats_groups || (p.N / p.gn_stats_groups) < 8 || p.Hout * p.Wout <= 0)
        return 0;
    const int HW = p.Hout * p.Wout, rpp = fs_rows_per_part(p.M / HW, HW, p.N);
    return (HW + rpp - 1) / rpp;
}

// the second launch of a K-sliced problem: sum the slabs, run the epilogue (+ GroupNorm statistics of the output when asked for)
int launch_splitk_finish(const ConvGemmParams& p, hipStream_t s, const char* what) {
    const bool h = p.dtype == IMD_DTYPE_F16;
    if (p.gn_out_gamma != nullptr) {
        if (!splitk_gn_out_supported(p))
            return imd_set_error("%s: gn_out_* on a problem whose finish launch cannot normalise (ask imd_conv_gemm_gn_out_supported(): K slices with a separate "
                                 "finish, row-major 16-bit output, no residual / activation / scale, 4 | N / groups, H W N / groups <= %d)", what, 4 * FG_THREADS * FG_MAXU);
        const int B = p.M / (p.Hout * p.Wout);
        if (h) hipLaunchKernelGGL(splitk_finish_gn_kernel<true>, dim3((unsigned)p.gn_out_groups, (unsigned)B), dim3(FG_THREADS), 0, s, p);
        else hipLaunchKernelGGL(splitk_finish_gn_kernel<false>, dim3((unsigned)p.gn_out_groups, (unsigned)B), dim3(FG_THREADS), 0, s, p);
        return imd_check_launch(what);
    }
    if (p.gn_stats_out != nullptr) {
        const int nparts = splitk_stats_parts_of(p);
        if (nparts == 0) return imd_set_error("%s: gn_stats_out on a K-sliced problem that cannot produce the statistics", what);
        const int HW = p.Hout * p.Wout, B = p.M / HW, rpp = fs_rows_per_part(B, HW, p.N);
        if (h) hipLaunchKernelGGL(splitk_finish_stats_kernel<true>, dim3((unsigned)nparts, (unsigned)B), dim3(FS_THREADS), 0, s, p, rpp, nparts);
        else hipLaunchKernelGGL(splitk_finish_stats_kernel<false>, dim3((unsigned)nparts, (unsigned)B), dim3(FS_THREADS), 0, s, p, rpp, nparts);
        return imd_check_launch(what);
    }
    const long chunks = (long)p.M * ((p.N + 7) / 8);
    long blocks = (chunks + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (h) hipLaunchKernelGGL(splitk_finish_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(splitk_finish_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return imd_check_launch(what);
}

template <bool F16, int BM, int BN, int BK, int WM, int WN, int DEPTH = 2>
int launch_cfg(const ConvGemmParams& p, hipStream_t s) {
    static_assert(DEPTH == 2 || DEPTH == 4, "pipeline depth");
    constexpr int lds = TileCfg<BM, BN, BK, WM>::LDS;
    auto kern = conv_gemm_kernel<F16, BM, BN, BK, WM, WN, DEPTH>;
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), lds, "conv_gemm")) return rc_attr;
    const long mt = (p.M + BM - 1) / BM, nt = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3((unsigned)(mt * nt), (unsigned)p.split_k), dim3(WM * WN * 64), lds, s, p);
    int rc = imd_check_launch("conv_gemm");
    if (rc || p.split_k <= 1 || p.splitk_counters != nullptr) return rc;
    return launch_splitk_finish(p, s, "conv_gemm split-K finish");
}

int tile_dims(int cfg, int* bm, int* bn) {
    switch (cfg) {
        case 0: *bm = 128; *bn = 128; return 0;
        case 1: *bm = 128; *bn = 64; return 0;
        case 2: *bm = 64; *bn = 64; return 0;
        case 4: case 5: case 8: case 29: *bm = 128; *bn = 128; return 0;
        case 3: case 7: *bm = 64; *bn = 64; return 0;
        case 6: *bm = 64; *bn = 320; return 0;
        case 9: case 21: case 30: case 31: *bm = 256; *bn = 128; return 0;
        case 32: *bm = 192; *bn = 128; return 0;
        case 22: *bm = 128; *bn = 160; return 0;
        case 23: *bm = 256; *bn = 160; return 0;
        case 24: *bm = 512; *bn = 64; return 0;
        case 10: case 16: *bm = 256; *bn = 256; return 0;
        case 17: case 18: case 19: case 20: case 25: case 26: case 27: case 28: *bm = 128; *bn = 128; return 0;
        case 11: *bm = 128; *bn = 320; return 0;
        default: return 1;
    }
}

}  // namespace

int g_gemm_flags = 23;   // tuning knob 2: bit0 tap-inner K order for 3x3 convs, bit1 weight loads bypass L1, bit2 XCD-aware tile order,
                         // bit4 grouped (8 row tiles) order inside an XCD's range

// tile-order flag (see xcd_tile_order): fabric bytes if every XCD owns whole row tiles (activations once, weights x8) vs
// whole channel tiles (weights once, activations x min(8, channel tiles))
int imd_gemm_pick_order(const ConvGemmParams& p, int n_tiles) {
    const double a_bytes = (double)p.x_bytes, w_bytes = (double)p.w_bytes / (p.split_k > 1 ? p.split_k : 1);
    const double a_split = a_bytes / (p.split_k > 1 && p.taps == 1 ? p.split_k : 1);
    const double cost_m = a_split + 8.0 * w_bytes;
    const double cost_n = (n_tiles < 8 ? n_tiles : 8) * a_split + w_bytes;
    return cost_m <= cost_n ? 4 : 8;
}

// statistic partials per image a launch of `p` with tile config `cfg` writes through gn_stats_out (0: none): the halo-patch kernel's own
// epilogue when it runs un-split, the finish launch of any K-sliced problem otherwise
int imd_conv_gemm_stats_parts_of(const ConvGemmParams& p_in, int cfg) {
    ConvGemmParams p = p_in;
    if (p.split_k < 1) p.split_k = 1;
    if (cfg < 0) cfg = imd_conv_gemm_choose_cfg(p.M, p.N);       // (as the launcher resolves it)
    if (p.split_k > 1) {
        if (cfg == 21 || cfg == 22 || cfg == 23 || cfg == 24 || cfg == 29 || (cfg >= 17 && cfg <= 20) || (cfg >= 25 && cfg <= 28) || (cfg >= 30 && cfg <= 32)) p.splitk_counters = nullptr;      // (these always finish with the second launch)
        return splitk_stats_parts_of(p);
    }
    if (cfg == 22 || cfg == 23) return imd_conv_patch3_stats_parts_of(p, cfg == 23 ? 8 : 4);
    if (cfg == 5 || cfg == 29) return imd_conv_patch_stats_parts_of(p);
    // the register-staged tiles whose epilogue threads keep their column block (conv_gemm_kernel's statistics epilogue): whole tiles of one image only
    if (cfg == 0 || cfg == 1 || cfg == 2 || cfg == 3 || cfg == 4 || cfg == 7) {
        int bm = 0, bn = 0;
        tile_dims(cfg, &bm, &bn);
        const int HW = p.Hout * p.Wout, G = p.gn_stats_groups;
        if (p.out_f32 || p.mode != OUT_ROWMAJOR || p.act == ACT_GEGLU || G <= 0 || G > 64 || p.N % G || (p.N / G) < 8 || (p.N % 4) || HW <= 0 || HW % bm || p.M % HW) return 0;
        return (HW / bm) * ((p.N + bn - 1) / bn);
    }
    return 0;
}

// can tile config `cfg` normalise its input rows as p.gn_in_* asks (the row-resident projections only)?
bool imd_row_linear_gn_in_supported_of(const ConvGemmParams& p, int cfg) {
    if (p.gn_in_partial == nullptr) return true;
    ConvGemmParams q = p;
    q.split_k = 1;
    switch (cfg) {
        case 12: return imd_row_linear_supported(q) && gn_in_ok(q, 320, 128);
        case 13: return imd_row_linear_k640_supported(q) && gn_in_ok(q, 640, 128);
        case 14: return imd_row_linear_k1280_supported(q) && gn_in_ok(q, 1280, 64);
        default: return false;
    }
}

// can the finish launch of `p` (tile config `cfg`) normalise its output (gn_out_*)?  The in-kernel K-slice sum (splitk_counters) is simply not used then.
bool imd_conv_gemm_gn_out_supported_of(const ConvGemmParams& p_in, int cfg) {
    ConvGemmParams p = p_in;
    (void)cfg;
    p.splitk_counters = nullptr;
    return splitk_gn_out_supported(p);
}

int imd_conv_gemm_choose_cfg(int M, int N) {
    const long b128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (b128 < 96) return 2;          // tiny problems: smaller tiles fill more CUs
    if (N <= 64) return 1;
    // many tiles (>= ~3 per CU): the 41 KB / 140-VGPR BK=32 variant keeps 3 workgroups per CU resident and wins
    // (measured on the UNet's level-0 and GEGLU shapes); with fewer tiles the BK=64 variant's longer MFMA runs win.
    if (b128 >= 700) return 4;
    return 0;
}

// number of K slices: only for row-major epilogues on problems whose tile grid cannot fill the chip
int imd_conv_gemm_choose_split(int M, int N, int K, int cfg) {
    int bm = 128, bn = 128;
    tile_dims(cfg, &bm, &bn);
    const long blocks = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    const int ktiles = (K + 63) / 64;
    if (blocks >= 200 || ktiles < 32) return 1;
    long s = (448 + blocks - 1) / blocks;           // aim at ~1.75 blocks per CU
    if (s > ktiles / 16) s = ktiles / 16;
    if (s > 8) s = 8;
    return s < 1 ? 1 : (int)s;
}

// buffer-descriptor extents of the two operands (x_bytes / w_bytes: "filled in by the library"); false when one exceeds 4 GiB.  Shared by the
// launcher and by the *_supported queries of the C ABI, whose callers hand in blocks with these fields still zero.
bool imd_conv_gemm_fill_extents(ConvGemmParams& p) {
    if (p.Hout <= 0 || p.Wout <= 0 || p.M <= 0) return false;
    const size_t npix = (size_t)(p.M / (p.Hout * p.Wout)) * p.Hin * p.Win;
    if (npix == 0) return false;
    const size_t xb = ((npix - 1) * (size_t)p.x_pix_stride + p.Cin) * 2, wb = (size_t)p.N * p.K * 2;
    if (xb >= 0xffffffffull || wb >= 0xffffffffull) return false;
    p.x_bytes = (uint32_t)xb;
    p.w_bytes = (uint32_t)wb;
    return true;
}

int imd_launch_conv_gemm(const ConvGemmParams& p_in, int cfg, hipStream_t s) {
    ConvGemmParams p = p_in;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return imd_set_error("conv_gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    if ((p.K % 8) || (p.Cin % 8) || (p.x_pix_stride % 8))
        return imd_set_error("conv_gemm: K (%d), Cin (%d) and pixel stride (%d) must be multiples of 8", p.K, p.Cin, p.x_pix_stride);
    if (p.N % 4) return imd_set_error("conv_gemm: N (%d) must be a multiple of 4", p.N);
    if (p.taps != 1 && p.taps != 9) return imd_set_error("conv_gemm: taps must be 1 or 9 (got %d)", p.taps);
    if (p.K != p.taps * p.Cin) return imd_set_error("conv_gemm: K (%d) != taps*Cin (%d)", p.K, p.taps * p.Cin);
    if (p.mode == OUT_HEADS && (p.hD % 8)) return imd_set_error("conv_gemm: head dim (%d) must be a multiple of 8", p.hD);
    if (p.act == ACT_GEGLU && (p.N % 8)) return imd_set_error("conv_gemm: GEGLU needs N %% 8 == 0");
    if (!imd_conv_gemm_fill_extents(p)) return imd_set_error("conv_gemm: operand larger than 4 GiB");
    if (cfg < 0) cfg = imd_conv_gemm_choose_cfg(p.M, p.N);
    if (p.split_k < 1) p.split_k = 1;
    // tuning bits of this call: the caller's (IMD_TUNING_PER_CALL in flags on entry) or the process-wide knob 2
    const unsigned tag = (unsigned)p_in.flags & IMD_TUNING_TAG_MASK;
    if (tag != 0 && tag != (unsigned)IMD_TUNING_PER_CALL)
        return imd_set_error("conv_gemm: flags = 0x%x on entry is neither 0 nor IMD_TUNING_PER_CALL | bits (an uninitialised parameter block?)", (unsigned)p_in.flags);
    const int gf = (tag == (unsigned)IMD_TUNING_PER_CALL) ? ((p_in.flags & 31) | (g_gemm_flags & ~31)) : g_gemm_flags;
    {
        const int bk = (cfg == 4) ? 32 : 64;
        p.flags = 0;
        // measured (profiles/r1g_gemm_flags_ab.jsonl): +8..17 % on the 64x64 / 32x32 feature maps, -2..4 % on 16x16 / 8x8
        const bool big_map = p.taps == 9 && p.Wout >= 32;
        if ((gf & 1) && big_map && (p.Cin % bk) == 0) p.flags |= 1;
        if ((gf & 2) && big_map) p.flags |= 2;
    }
    if (gf & 4) {
        int bm = 128, bn = 128;
        tile_dims(cfg, &bm, &bn);
        p.flags |= imd_gemm_pick_order(p, (p.N + bn - 1) / bn);
        p.flags |= gf & 16;
    }
#ifdef IMD_ABLATIONS
    p.flags |= gf & 224;       // bits 5..7: timing ablations of gemm_dma256.hip (A/B only; WRONG results when set; imd_set_tuning refuses them otherwise)
#endif
    if (p.split_k <= 1) p.splitk_counters = nullptr;
    // GroupNorm statistics of the output (ABI v6+): the halo-patch kernels' un-split epilogues (tile configs 5 / 22 / 23 / 29) and the finish
    // launch of any K-sliced problem produce them (imd_conv_gemm_stats_parts_of); every other request is an
    // ERROR -- a launch that silently skipped the write would leave the next imd_groupnorm(nparts > 0) reading uninitialised memory, and
    // gn_stats_groups = 0 / fewer than 8 channels per group would divide by zero / straddle more than two groups in the kernel
    if (p.gn_stats_out != nullptr && imd_conv_gemm_stats_parts_of(p, cfg) == 0)
        return imd_set_error("conv_gemm: gn_stats_out needs tile config 5 / 22 / 23 / 29 or a register-staged tile (0..4, 7) with H W %% its rows == 0, without K slices, or any K-sliced launch with a separate finish; a row-major 16-bit "
                             "output and 1 <= groups <= 64 with N %% groups == 0 and N / groups >= 8 (got cfg=%d split_k=%d groups=%d N=%d); ask "
                             "imd_conv_gemm_stats_parts() first", cfg, p.split_k, p.gn_stats_groups, p.N);
    // GroupNorm of the output inside the finish launch (ABI v9): only a K-sliced problem with a separate finish has one -- anything else is an
    // ERROR (a launch that silently skipped the normalisation would hand the caller a tensor it believes normalised)
    if (p.gn_out_gamma != nullptr && !imd_conv_gemm_gn_out_supported_of(p, cfg))
        return imd_set_error("conv_gemm: gn_out_* needs K slices with a separate finish launch, a row-major 16-bit output without residual / activation / scale / "
                             "gn_stats_out, 4 | N / groups and H W N / groups <= %d (got cfg=%d split_k=%d groups=%d N=%d); ask imd_conv_gemm_gn_out_supported() first",
                             4 * FG_THREADS * FG_MAXU, cfg, p.split_k, p.gn_out_groups, p.N);
    if (p.gn_out_gamma != nullptr) p.splitk_counters = nullptr;
    if (p.splitk_counters != nullptr) {          // one counter per output tile; larger grids keep the two-launch path
        int bm = 128, bn = 128;
        if (cfg == 5) { bm = 128; bn = 128; } else tile_dims(cfg, &bm, &bn);
        const long tiles = (cfg == 5) ? (long)(p.M / (p.Hout * p.Wout)) * ((p.Hout + 7) / 8) * ((p.Wout + 15) / 16) * ((p.N + 127) / 128)
                                      : (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
        if (tiles > IMD_SPLITK_COUNTERS || (size_t)p.M * p.N * 4 >= 0x80000000ull) p.splitk_counters = nullptr;
    }
    if (p.split_k > 1) {
        if (p.mode == OUT_HEADS || p.act == ACT_GEGLU) return imd_set_error("conv_gemm: split-K supports row-major epilogues only");
        if (!p.splitk_ws) return imd_set_error("conv_gemm: split_k = %d needs a workspace", p.split_k);
    }
    if (p.dtype != IMD_DTYPE_BF16 && p.dtype != IMD_DTYPE_F16) return imd_set_error("conv_gemm: unknown dtype %d", p.dtype);
    if ((p.gn_a != nullptr) != (p.gn_b != nullptr)) return imd_set_error("conv_gemm: gn_a and gn_b must be given together");
    if (p.gn_a != nullptr && cfg != 5) return imd_set_error("conv_gemm: the fused GroupNorm prologue needs tile config 5 (got %d)", cfg);
    if (p.res_rows != 0 && cfg != 12)
        return imd_set_error("conv_gemm: a periodic residual (res_rows = %d) exists in the K = 320 row-resident projection only (tile config 12, got %d)", p.res_rows, cfg);
    if (p.gn_in_partial != nullptr && !imd_row_linear_gn_in_supported_of(p, cfg))
        return imd_set_error("conv_gemm: gn_in_* (GroupNorm of the input rows) exists in the row-resident projections only (tile configs 12 / 13 / 14 with H W a multiple of "
                             "their row block; got cfg=%d K=%d HW=%d): ask imd_row_linear_gn_in_supported() first", cfg, p.K, p.Hout * p.Wout);
    const bool h = p.dtype == IMD_DTYPE_F16;
    switch (cfg) {
        case 0: return h ? launch_cfg<true, 128, 128, 64, 2, 2>(p, s) : launch_cfg<false, 128, 128, 64, 2, 2>(p, s);
        case 1: return h ? launch_cfg<true, 128, 64, 64, 2, 2>(p, s) : launch_cfg<false, 128, 64, 64, 2, 2>(p, s);
        case 2: return h ? launch_cfg<true, 64, 64, 64, 2, 2>(p, s) : launch_cfg<false, 64, 64, 64, 2, 2>(p, s);
        case 3: return h ? launch_cfg<true, 64, 64, 64, 2, 2, 4>(p, s) : launch_cfg<false, 64, 64, 64, 2, 2, 4>(p, s);       // 4 tiles in flight
        case 4: return h ? launch_cfg<true, 128, 128, 32, 2, 2>(p, s) : launch_cfg<false, 128, 128, 32, 2, 2>(p, s);         // 41 KB LDS: 3 workgroups / CU
        case 6: return h ? launch_cfg<true, 64, 320, 32, 2, 2>(p, s) : launch_cfg<false, 64, 320, 32, 2, 2>(p, s);               // N % 320 == 0: no idle columns, A read once
        case 7: return h ? launch_cfg<true, 64, 64, 32, 2, 2, 4>(p, s) : launch_cfg<false, 64, 64, 32, 2, 2, 4>(p, s);       // 20 KB LDS: 8 workgroups / CU
        case 8: return h ? launch_cfg<true, 128, 128, 32, 2, 2, 4>(p, s) : launch_cfg<false, 128, 128, 32, 2, 2, 4>(p, s);
        case 9: return h ? launch_cfg<true, 256, 128, 32, 4, 2>(p, s) : launch_cfg<false, 256, 128, 32, 4, 2>(p, s);             // 8 waves: operand bytes per MFMA -25 %
        case 10: return h ? launch_cfg<true, 256, 256, 32, 4, 2>(p, s) : launch_cfg<false, 256, 256, 32, 4, 2>(p, s);            // 8 waves, 64x128 per wave: -50 %
        case 11: return h ? launch_cfg<true, 128, 320, 64, 4, 2>(p, s) : launch_cfg<false, 128, 320, 64, 4, 2>(p, s);            // N = 320 k: one full-width row block per CU
        case 5: {   // LDS-resident halo patch (conv_patch.hip): 3x3 stride-1 only, optional fused GroupNorm prologue
            int rc = imd_launch_conv_patch(p, s);
            if (rc || p.split_k <= 1 || p.splitk_counters != nullptr) return rc;
            return launch_splitk_finish(p, s, "conv_patch split-K finish");
        }
        case 29: {  // the halo-patch kernel with 64-channel chunks: 128-byte rows, i.e. whole L2 lines (conv_patch.hip)
            p.splitk_counters = nullptr;
            int rc = imd_launch_conv_patch64(p, s);
            if (rc || p.split_k <= 1) return rc;
            return launch_splitk_finish(p, s, "conv_patch split-K finish");
        }
        case 21: {  // halo patch, 16 x 16 pixel tiles (conv_patch2.hip)
            p.splitk_counters = nullptr;
            int rc = imd_launch_conv_patch2(p, s);
            if (rc || p.split_k <= 1) return rc;
            return launch_splitk_finish(p, s, "conv_patch2 split-K finish");
        }
        case 22: case 23: {  // halo patch x 160 channels (conv_patch3.hip): 22 = 8 x 16 pixels, four waves; 23 = 16 x 16 pixels, eight waves (one weight tile per 256 pixels)
            int rc = cfg == 22 ? imd_launch_conv_patch3(p, s) : imd_launch_conv_patch4(p, s);
            if (rc || p.split_k <= 1) return rc;
            p.splitk_counters = nullptr;
            return launch_splitk_finish(p, s, "conv_patch3 split-K finish");
        }
        case 25: case 26: case 27: case 28: {   // the 128 x 128 LDS-DMA tiles with 128-BYTE rows (BK = 64): 25 / 27 plain linears with two / three stages, 26 / 28 3x3 convs
            if ((cfg == 26 || cfg == 28) != (p.taps == 9)) return imd_set_error("conv_gemm: tile config %d does not take taps = %d", cfg, p.taps);
            p.splitk_counters = nullptr;
            int rc = imd_launch_gemm_dma128(p, cfg >= 27 ? 13 : 12, s);
            if (rc || p.split_k <= 1) return rc;
            return launch_splitk_finish(p, s, "gemm_dma128 split-K finish");
        }
        case 24: {  // whole small maps (8 pixels wide) x 64 channels x one K slice per workgroup (conv_img.hip): every weight byte fetched once
            p.splitk_counters = nullptr;
            if (p.split_k <= 1) return imd_set_error("conv_gemm: tile config 24 writes K-slice slabs only (split_k >= 2)");
            int rc = imd_launch_conv_img(p, s);
            if (rc) return rc;
            return launch_splitk_finish(p, s, "conv_img split-K finish");
        }
        case 30: case 31: case 32: {   // LDS-DMA tiles with producer / consumer waves and a register epilogue (gemm_dma256.hip): 30 = 256 x 128 x 64, three stages,
                              // persistent over (tile, K slice) items; 31 = the same with one item per workgroup; 32 (round 6) = 192 x 128 x 64, persistent
            p.splitk_counters = nullptr;
            int rc = imd_launch_gemm_dma256(p, cfg - 30, s);
            if (rc || p.split_k <= 1) return rc;
            return launch_splitk_finish(p, s, "gemm_dma256 split-K finish");
        }
        case 12: return imd_launch_row_linear(p, 0, 0.f, s);      // row-resident kernel (row_linear.hip): K = 320, N <= 320
        case 13: return imd_launch_row_linear_k640(p, 0, 0.f, s); // row-resident split-K kernel (row_linear_k640.hip): K = 640, N % 160 == 0
        case 14: return imd_launch_row_linear_k1280(p, 0, 0.f, s); // row-resident 4-way split-K kernel (row_linear_k1280.hip): K = 1280, N % 160 == 0
        case 16: return imd_launch_gemm_dma(p, s);                 // 256 x 256 x 64 LDS-DMA tile kernel for the large linears (gemm_dma.hip)
        case 15: return imd_launch_row_qkv(p, 0, 0.f, s);          // row-resident q/k/v projection of a 320-channel block (row_qkv.hip)
        case 17: case 18: case 19: case 20: {   // 128 x 128 x 32 LDS-DMA tiles (gemm_dma.hip): 17 / 18 three-stage ring, 3 workgroups / CU; 19 / 20 four stages, 2 / CU;
                                                // 17, 19: plain linears, 18, 20: 3x3 convs gathered per tile; K slices finish like the tiled kernels'
            if ((cfg == 18 || cfg == 20) != (p.taps == 9)) return imd_set_error("conv_gemm: tile config %d does not take taps = %d", cfg, p.taps);
            p.splitk_counters = nullptr;
            int rc = imd_launch_gemm_dma128(p, cfg >= 19 ? 4 : 3, s);
            if (rc || p.split_k <= 1) return rc;
            return launch_splitk_finish(p, s, "gemm_dma128 split-K finish");
        }
        default: return imd_set_error("conv_gemm: unknown tile config %d", cfg);
    }
}
