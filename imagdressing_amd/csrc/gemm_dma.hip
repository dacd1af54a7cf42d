// 256 x 256 x 64 tiled GEMM for the large plain linear layers (taps = 1, K % 64 == 0): the GEGLU / feed-forward-out / QKV
// projections of the 32x32 and 16x16 levels (M = 8192 / 2048 rows, K = 640 ... 5120), which the gather kernel of
// conv_gemm.hip runs at 450-630 TFLOP/s (matrix pipe ~22 % busy: its operands go global -> VGPR -> ds_write -> barrier ->
// ds_read every 32- or 64-deep step with two or three small workgroups per CU to hide it).
//
// Structure (cdna_hip_programming.md, GEMM staging table: "256^2 tile, ~1 block / CU, LDS-DMA, 2 LDS buffers, BK = 64,
// vmcnt(0) + plain barrier"):
//   * one workgroup = 8 waves (4 along M x 2 along N), wave tile 64 x 128 = 8 accumulator blocks: every activation fragment
//     feeds 4 MFMAs and every weight fragment 2 (0.75 KB of LDS reads per MFMA instead of 1 KB);
//   * both operand tiles (256 rows x 64 k x 2 B = 32 KB each) go global -> LDS by DMA (buffer_load ... lds) in whole 128-byte
//     row pieces, double buffered: tile t + 1 is in flight while tile t is multiplied, ONE barrier per 64-deep step; rows
//     unpadded, piece p of row r stored at p ^ ((r >> 1) & 7) (source-side swizzle) -> conflict-free ds_read_b128;
//   * out-of-range rows (M / N tails) are out-of-range DMA offsets and arrive as zeros;
//   * the epilogue is the tiled kernel's: accumulators -> LDS (fp32, 64 rows at a time) -> 8 consecutive channels per thread
//     through epilogue8 (bias / residual / activations / GEGLU / head-split layouts), XCD-aware tile order.
// Same arithmetic as conv_gemm.hip (fp32 accumulation over K in 16-element MFMA steps, ascending), same reference layers.
#include "gemm_common.h"
#include "lds_dma.h"

namespace {

constexpr int GD_BM = 256, GD_BN = 256, GD_BK = 64;
constexpr int GD_ROWB = GD_BK * 2;                       // 128 bytes = 8 pieces per operand row and stage
constexpr int GD_A = GD_BM * GD_ROWB, GD_W = GD_BN * GD_ROWB, GD_STAGE = GD_A + GD_W;     // 32 KB + 32 KB
constexpr int GD_LDS = 2 * GD_STAGE;                     // 131072
constexpr int GD_CLD = GD_BN + 4, GD_EROWS = 64;         // fp32 epilogue staging: 64 x 260 x 4 = 66560 bytes

template <bool F16>
__global__ __launch_bounds__(512, 1) void gemm_dma_kernel(const ConvGemmParams p) {
    using E = El<F16>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, col = lane & 31;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 128;

    const int n_tiles = (p.N + GD_BN - 1) / GD_BN;
    int tile_m, tile_n;
    xcd_tile_order(p.flags, (p.M + GD_BM - 1) / GD_BM, n_tiles, tile_m, tile_n);
    const int m0 = tile_m * GD_BM, n0 = tile_n * GD_BN;
    const int nk = p.K / GD_BK;

    // ---- DMA assignments: a stage is 64 pieces of 1 KB (8 rows x 128 B each): pieces 0..31 activations, 32..63 weights ----
    const v4i_t ds_x = raw_rsrc(p.x, p.x_bytes), ds_w = raw_rsrc(p.w, p.w_bytes);
    uint32_t soff[8];        // source byte offset of this lane's 16 bytes of piece (j * 8 + wave) at k tile 0, or OOB
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int id = j * 8 + wave;
        const bool isw = j >= 4;                              // (id >= 32)
        const int q = (id & 31) * 64 + lane;                 // 16-byte slot inside the operand tile
        const int row = q >> 3, pos = q & 7;
        const int pc = pos ^ ((row >> 1) & 7);                // source piece stored at this slot
        if (isw) soff[j] = (n0 + row < p.N) ? (uint32_t)(((size_t)(n0 + row) * p.K + pc * 8) * 2) : OOB;
        else soff[j] = (m0 + row < p.M) ? (uint32_t)(((size_t)(m0 + row) * p.x_pix_stride + pc * 8) * 2) : OOB;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    auto stage = [&](int kt) {
        const uint32_t base = lds0 + (uint32_t)((kt & 1) * GD_STAGE);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int id = j * 8 + wave;
            dma16(j >= 4 ? ds_w : ds_x, base + (uint32_t)id * 1024u, soff[j] == OOB ? OOB : soff[j] + (uint32_t)(kt * GD_ROWB));
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment addresses: row (wm0 + b * 32 + col) of the activation tile, row (wn0 + a * 32 + col) of the weight tile;
    // k16 step kk reads piece 2 kk + hi, stored at (2 kk + hi) ^ ((row >> 1) & 7) = (2 kk) ^ (hi ^ f)
    const uint32_t f16 = (uint32_t)((hi ^ ((col >> 1) & 7)) << 4);       // (row >> 1) & 7 == (col >> 1) & 7: row offsets are multiples of 32
    const char* xlane = smem + (wm0 + col) * GD_ROWB;
    const char* wlane = smem + GD_A + (wn0 + col) * GD_ROWB;

    stage(0);
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        dma_wait();                    // this wave's pieces of tile kt have landed
        __syncthreads();               // ... everybody's; all waves are done with tile kt - 1 (the buffer tile kt + 1 goes to)
        if (kt + 1 < nk) stage(kt + 1);
        const char* Xs = xlane + (kt & 1) * GD_STAGE;
        const char* Ws = wlane + (kt & 1) * GD_STAGE;
#pragma unroll
        for (int kk = 0; kk < GD_BK / 16; ++kk) {
            const uint32_t po = (uint32_t)(kk * 32) ^ f16;
            uint4 wf[4], xf[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) xf[b] = *reinterpret_cast<const uint4*>(Xs + b * 32 * GD_ROWB + po);
#pragma unroll
            for (int a = 0; a < 4; ++a) wf[a] = *reinterpret_cast<const uint4*>(Ws + a * 32 * GD_ROWB + po);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = E::mfma(wf[a], xf[b], acc[a][b]);
        }
    }
    __syncthreads();                   // the last tile has been read by everybody: LDS becomes the epilogue staging area

    // ---- epilogue (conv_gemm.hip's): one 64-row wave group at a time through LDS (fp32), 8 consecutive channels per thread ----
    float* Cs = reinterpret_cast<float*>(smem);
    const int HWo = p.Hout * p.Wout;
    constexpr int CPR = GD_BN / 8;                 // 32 chunks per row
    constexpr int CHUNKS = GD_EROWS * CPR;         // 2048
    const bool colmajor = p.mode == OUT_HEADS;
    float4 col_pre0 = make_float4(0, 0, 0, 0), col_pre1 = col_pre0;
    bool use_col_pre = false;
    if (!colmajor && (p.bias || p.rowvec)) {       // 512 % 32 == 0: a thread keeps its 8 columns from row to row
        const int bi_lo = m0 / HWo, bi_hi = (min(m0 + GD_BM, p.M) - 1) / HWo;
        const int n = n0 + (tid % CPR) * 8;
        if ((p.rowvec == nullptr || bi_lo == bi_hi) && n < p.N) {
            load_col_addends(p, p.rowvec ? bi_lo : -1, n, (n + 8 <= p.N) ? 8 : 4, col_pre0, col_pre1);
            use_col_pre = true;
        }
    }
#pragma unroll 1
    for (int wr = 0; wr < 4; ++wr) {
        if ((wave >> 1) == wr) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* dst = Cs + (b * 32 + col) * GD_CLD + wn0 + a * 32 + 8 * j + 4 * hi;
                        *reinterpret_cast<float4*>(dst) = make_float4(acc[a][b][4 * j], acc[a][b][4 * j + 1], acc[a][b][4 * j + 2], acc[a][b][4 * j + 3]);
                    }
        }
        __syncthreads();
        for (int c = tid; c < CHUNKS; c += 512) {
            int row, cc;
            if (colmajor) { cc = (c / GD_EROWS) * 8; row = c - (c / GD_EROWS) * GD_EROWS; }
            else { row = c / CPR; cc = (c - row * CPR) * 8; }
            const int m = m0 + wr * GD_EROWS + row, n = n0 + cc;
            if (m >= p.M || n >= p.N) continue;
            const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * GD_CLD + cc);
            const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * GD_CLD + cc + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            epilogue8<F16>(p, v, m, n, (n + 8 <= p.N) ? 8 : 4, HWo, use_col_pre, col_pre0, col_pre1);
        }
        if (wr + 1 < 4) __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 128 x 128 x 32 tiles, 4 waves, THREE-stage LDS-DMA ring (round 3; tile config 17).  The 256^2 kernel above keeps one
// workgroup per CU and drains its DMA (vmcnt(0)) in front of every barrier with a single tile of lead; on the K = 640 ... 2560
// linears of the UNet it only ties the register-staged tiles.  This shape is the halo-patch conv's recipe (conv_patch.hip)
// applied to a plain [M, K] x [N, K]^T product: 16 KB stages (128 + 128 rows of 64 bytes), three of them = 48 KB -> three
// workgroups per CU, tile t + 2 in flight while tile t is multiplied, counted `s_waitcnt vmcnt(4)` (each wave issues exactly
// four 1-KB pieces per tile), unpadded 64-byte rows with piece c of row r at c ^ ((r >> 2) & 3) (conflict-free ds_read_b128),
// no staging registers, no VGPR -> LDS stores.  Epilogue = conv_gemm.hip's.
// ---------------------------------------------------------------------------------------------------------------------------
// NST = 4 (tile configs 19 / 20; round 3): a FOUR-stage ring, 64 KB -> two workgroups per CU with three tiles of lead instead of three
// workgroups with two.  Same bytes in flight per CU, but each workgroup's wait is one tile further behind its issue: -8...-22 % on the
// large linears on one box (profiles/r3ab_ring_depth_big_linears.txt; five stages = one workgroup per CU is slower than either).
constexpr int G1_BM = 128, G1_BN = 128, G1_BK = 32;
constexpr int G1_STAGE = (G1_BM + G1_BN) * G1_BK * 2;      // 16384
constexpr int G1_CLD = G1_BN + 4, G1_EROWS = 64;
static_assert(3 * G1_STAGE >= G1_EROWS * G1_CLD * 4, "epilogue tile must fit");

// GATHER (tile config 18): the A operand is the implicit im2col matrix of a 3x3 convolution (stride 1 | 2, zero halo, optional fused
// nearest-2x upsample, Cin % 32 == 0) -- a K tile is one tap x 32 channels (tap-inner order: consecutive tiles are neighbouring
// taps of the same channels), an A row's 64 bytes are the 32 channels of ONE input pixel, and the per-lane DMA offset is
// recomputed per tile from the row's top-left tap position; halo pixels are out-of-range offsets.  Serves the maps the halo-patch
// kernel cannot tile (8 x 8) and the stride-2 convs.
// BK = 64 (tile configs 25 / 26 / 27; round 4): 128-BYTE rows.  The L2 hands a CU one 128-byte line per request whatever part of it was asked for
// (tools/probes/staging_probe.hip: 62 GB/s per CU in 64-byte segments, 113 GB/s in 128-byte segments) -- the 64-byte rows of BK = 32 use half of
// every line they pull.  Stage = 32 KB (two stages = 64 KB = two workgroups per CU, three = 96 KB = one), eight pieces per wave and tile, piece
// c of row r at c ^ ((r >> 1) & 7).
template <bool F16, bool GATHER, int G1_NST, int BK = 32>
__global__ __launch_bounds__(256, (G1_NST * (G1_BM + G1_BN) * BK * 2 <= 49152) ? 3 : (G1_NST * (G1_BM + G1_BN) * BK * 2 <= 65536) ? 2 : 1)
void gemm_dma128_kernel(const ConvGemmParams p) {
    using E = El<F16>;
    constexpr int ROWB = BK * 2, LPR = ROWB / 16;              // bytes / 16-byte pieces per row
    constexpr int STAGE = (G1_BM + G1_BN) * ROWB;
    constexpr int PPW = G1_BM * ROWB / 1024 / 4;               // pieces per wave and operand tile: 2 | 4
    constexpr int NP = 2 * PPW;
    constexpr int G1_KEEP = (G1_NST - 2) * NP;                 // pieces of the younger tiles that may stay in flight at the per-tile wait
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, col = lane & 31;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;

    const int n_tiles = (p.N + G1_BN - 1) / G1_BN;
    int tile_m, tile_n;
    xcd_tile_order(p.flags, (p.M + G1_BM - 1) / G1_BM, n_tiles, tile_m, tile_n);
    const int m0 = tile_m * G1_BM, n0 = tile_n * G1_BN;
    // K range of this slice (split-K: blockIdx.y; fp32 slabs + the fixed-order finish launch of conv_gemm.hip)
    const int nk_total = p.K / BK;                 // (GATHER: 9 taps x Cin / BK channel chunks)
    const int per = (nk_total + p.split_k - 1) / p.split_k;
    const int kt0 = blockIdx.y * per;
    const int nk = max(0, min(nk_total, kt0 + per) - kt0);

    // a stage = 16 pieces of 1 KB (16 rows x 64 B each): pieces 0..7 activation rows, 8..15 weight rows; wave w issues pieces
    // w, w + 4 (activations) and 8 + w, 12 + w (weights)
    const v4i_t ds_x = raw_rsrc(p.x, p.x_bytes), ds_w = raw_rsrc(p.w, p.w_bytes);
    uint32_t soff[NP];
    int g_base[PPW], g_y0[PPW], g_x0[PPW], g_pc[PPW];        // GATHER: per A piece (row) of this lane
#pragma unroll
    for (int j = 0; j < PPW; ++j) { g_base[j] = -1; g_y0[j] = 0; g_x0[j] = 0; g_pc[j] = 0; }
    const int Hl = p.ups ? p.Hin * 2 : p.Hin, Wl = p.ups ? p.Win * 2 : p.Win;              // logical (post-upsample) input map
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int id = (j % PPW) * 4 + wave;                  // piece inside its operand tile
        const int q = id * 64 + lane, row = q / LPR, pc = (q % LPR) ^ (BK == 32 ? (row >> 2) & 3 : (row >> 1) & 7);
        if (j >= PPW) soff[j] = (n0 + row < p.N) ? (uint32_t)(((size_t)(n0 + row) * p.K + pc * 8) * 2) : OOB;
        else if (!GATHER) soff[j] = (m0 + row < p.M) ? (uint32_t)(((size_t)(m0 + row) * p.x_pix_stride + pc * 8) * 2) : OOB;
        else {
            soff[j] = OOB;
            const int m = m0 + row;
            if (m < p.M) {
                const int HWo = p.Hout * p.Wout, pad = p.pad_br_only ? 0 : 1;
                const int bi = m / HWo, rem = m - bi * HWo, oy = rem / p.Wout, ox = rem - oy * p.Wout;
                g_base[j] = bi * p.Hin * p.Win; g_y0[j] = oy * p.stride - pad; g_x0[j] = ox * p.stride - pad; g_pc[j] = pc * 8;
            }
        }
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int nchunk = GATHER ? p.Cin / BK : 1;
    auto stage = [&](int kt, int slot) {                      // (tiles past the end: zero-fill pieces keep the counted waits uniform)
        const uint32_t base = lds0 + (uint32_t)(slot * STAGE);
        const int kg = kt0 + kt;
        int ky = 0, kx = 0, ci = 0;
        uint32_t wk = (uint32_t)(kg * BK * 2);                // byte offset of the tile inside a weight row
        if (GATHER) {                                         // tap-inner tile order: tile kg -> (channel chunk kg / 9, tap kg % 9)
            const int c = kg / 9, tap = kg - 9 * c;
            ky = tap / 3; kx = tap - 3 * ky; ci = c * BK;
            wk = (uint32_t)((tap * p.Cin + ci) * 2);
            (void)nchunk;
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int id = (j % PPW) * 4 + wave + (j >= PPW ? 4 * PPW : 0);
            uint32_t off;
            if (j >= PPW) off = (soff[j] == OOB || kt >= nk) ? OOB : soff[j] + wk;
            else if (!GATHER) off = (soff[j] == OOB || kt >= nk) ? OOB : soff[j] + wk;
            else {
                const int iy = g_y0[j] + ky, ix = g_x0[j] + kx;
                const bool ok = kt < nk && g_base[j] >= 0 && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
                const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
                off = ok ? (uint32_t)((g_base[j] + sy * p.Win + sx) * p.x_pix_stride + ci + g_pc[j]) * 2u : OOB;
            }
            dma16(j >= PPW ? ds_w : ds_x, base + (uint32_t)id * 1024u, off);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment addresses (16-deep slice kk: piece 2 kk + hi): row offsets are multiples of 32, so the swizzle term of row (base + col) is col's
    int fo[BK / 16];
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) fo[kk] = col * ROWB + (((2 * kk + hi) ^ (BK == 32 ? (col >> 2) & 3 : (col >> 1) & 7)) << 4);
    const bool wave_live = n0 + wn0 < p.N && m0 + wm0 < p.M;

#pragma unroll
    for (int t = 0; t < G1_NST - 1; ++t) stage(t, t);
    dma_wait_keep_n<G1_KEEP>();
    __syncthreads();
    int slot = 0;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        stage(kt + G1_NST - 1, slot == 0 ? G1_NST - 1 : slot - 1);   // slot (kt - 1) % NST: last read at tile kt - 1, everybody is past that barrier
        if (wave_live) {
            const char* Xs = smem + slot * STAGE + wm0 * ROWB;
            const char* Ws = smem + slot * STAGE + G1_BM * ROWB + wn0 * ROWB;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                uint4 wf[2], xf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) wf[a] = *reinterpret_cast<const uint4*>(Ws + a * 32 * ROWB + fo[kk]);
#pragma unroll
                for (int b = 0; b < 2; ++b) xf[b] = *reinterpret_cast<const uint4*>(Xs + b * 32 * ROWB + fo[kk]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = E::mfma(wf[a], xf[b], acc[a][b]);
            }
        }
        slot = slot == G1_NST - 1 ? 0 : slot + 1;
        dma_wait_keep_n<G1_KEEP>();    // tile kt + 1 has landed (this wave's pieces; the younger tiles' stay in flight) ...
        __syncthreads();               // ... and everybody's
    }
    dma_wait();                        // zero-fill pieces past the end: the epilogue reuses this LDS
    __syncthreads();

    float* Cs = reinterpret_cast<float*>(smem);
    const int HWo = p.Hout * p.Wout;
    constexpr int CPR = G1_BN / 8;
    constexpr int CHUNKS = G1_EROWS * CPR;
    const bool colmajor = p.mode == OUT_HEADS;
    float* slab = (p.split_k > 1) ? p.splitk_ws + (size_t)blockIdx.y * p.M * p.N : nullptr;
    float4 col_pre0 = make_float4(0, 0, 0, 0), col_pre1 = col_pre0;
    bool use_col_pre = false;
    if (!colmajor && slab == nullptr && (p.bias || p.rowvec)) {
        const int bi_lo = m0 / HWo, bi_hi = (min(m0 + G1_BM, p.M) - 1) / HWo;
        const int n = n0 + (tid % CPR) * 8;
        if ((p.rowvec == nullptr || bi_lo == bi_hi) && n < p.N) {
            load_col_addends(p, p.rowvec ? bi_lo : -1, n, (n + 8 <= p.N) ? 8 : 4, col_pre0, col_pre1);
            use_col_pre = true;
        }
    }
#pragma unroll 1
    for (int wr = 0; wr < 2; ++wr) {
        if ((wave >> 1) == wr) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* dst = Cs + (b * 32 + col) * G1_CLD + wn0 + a * 32 + 8 * j + 4 * hi;
                        *reinterpret_cast<float4*>(dst) = make_float4(acc[a][b][4 * j], acc[a][b][4 * j + 1], acc[a][b][4 * j + 2], acc[a][b][4 * j + 3]);
                    }
        }
        __syncthreads();
        for (int c = tid; c < CHUNKS; c += 256) {
            int row, cc;
            if (colmajor) { cc = (c / G1_EROWS) * 8; row = c - (c / G1_EROWS) * G1_EROWS; }
            else { row = c / CPR; cc = (c - row * CPR) * 8; }
            const int m = m0 + wr * G1_EROWS + row, n = n0 + cc;
            if (m >= p.M || n >= p.N) continue;
            const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * G1_CLD + cc);
            const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * G1_CLD + cc + 4);
            if (slab) {                         // raw fp32 partial tile -> slab [split][M][N]
                slab_store8(slab, (size_t)m * p.N + n, v0, v1, n + 8 <= p.N, false);
            } else {
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                epilogue8<F16>(p, v, m, n, (n + 8 <= p.N) ? 8 : 4, HWo, use_col_pre, col_pre0, col_pre1);
            }
        }
        if (wr == 0) __syncthreads();
    }
}

}  // namespace

bool imd_conv_dma_supported(const ConvGemmParams& p) {       // tile config 18: 3x3 convs through the gathering form of the 128 x 128 x 32 DMA kernel
    return p.taps == 9 && (p.stride == 1 || p.stride == 2) && (p.Cin % G1_BK) == 0 && p.K == 9 * p.Cin && p.gn_a == nullptr &&
           (p.x_pix_stride % 8) == 0 && (!p.ups || p.stride == 1);
}

template <bool GATHER, int NST, int BK = 32>
static int launch_dma128(const ConvGemmParams& p, hipStream_t s, const char* what) {
    constexpr int LDS = NST * (G1_BM + G1_BN) * BK * 2;        // BK = 32: 49152 (three stages) | 65536 (four); BK = 64: 65536 (two) | 98304 (three)
    static_assert(LDS >= G1_EROWS * G1_CLD * 4, "epilogue tile must fit");
    const bool h = p.dtype == IMD_DTYPE_F16;
    typedef void (*kern_t)(const ConvGemmParams);
    const kern_t kern = h ? gemm_dma128_kernel<true, GATHER, NST, BK> : gemm_dma128_kernel<false, GATHER, NST, BK>;
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), LDS, "%s")) return rc_attr;
    const long mt = (p.M + G1_BM - 1) / G1_BM, nt = (p.N + G1_BN - 1) / G1_BN;
    hipLaunchKernelGGL(kern, dim3((unsigned)(mt * nt), (unsigned)p.split_k), dim3(256), LDS, s, p);
    return imd_check_launch(what);
}

// tile configs 17 / 19 (plain linears, three / four ring stages) and 18 / 20 (3x3 convs, likewise); K slices allowed
int imd_launch_gemm_dma128(const ConvGemmParams& p_in, int stages, hipStream_t s) {
    ConvGemmParams p = p_in;
    p.splitk_counters = nullptr;               // (the in-kernel reduction lives in the register-staged kernels only)
    if (stages == 12 || stages == 13) {       // 128-byte rows (BK = 64): two | three ring stages
        if (p.taps == 9) {
            if (!imd_conv_dma_supported(p) || (p.Cin % 64)) return imd_set_error("conv_dma (128-byte rows): needs a 3x3 convolution with Cin %% 64 == 0 (got Cin=%d stride=%d)", p.Cin, p.stride);
            return stages == 12 ? launch_dma128<true, 2, 64>(p, s, "conv_dma128 (BK 64)") : launch_dma128<true, 3, 64>(p, s, "conv_dma128 (BK 64, 3 stages)");
        }
        ConvGemmParams p1 = p;
        p1.split_k = 1;
        if (!imd_gemm_dma_supported(p1)) return imd_set_error("gemm_dma128: needs a plain linear layer with K %% 64 == 0 (got K=%d taps=%d)", p.K, p.taps);
        return stages == 12 ? launch_dma128<false, 2, 64>(p, s, "gemm_dma128 (BK 64)") : launch_dma128<false, 3, 64>(p, s, "gemm_dma128 (BK 64, 3 stages)");
    }
    if (stages != 3 && stages != 4) return imd_set_error("gemm_dma128: %d ring stages (3 | 4)", stages);
    if (p.taps == 9) {
        if (!imd_conv_dma_supported(p)) return imd_set_error("conv_dma: needs a 3x3 convolution with Cin %% 32 == 0 (got Cin=%d stride=%d)", p.Cin, p.stride);
        return stages == 3 ? launch_dma128<true, 3>(p, s, "conv_dma128") : launch_dma128<true, 4>(p, s, "conv_dma128 (4 stages)");
    }
    ConvGemmParams p1 = p;
    p1.split_k = 1;
    if (!imd_gemm_dma_supported(p1)) return imd_set_error("gemm_dma128: needs a plain linear layer with K %% 64 == 0 (got K=%d taps=%d)", p.K, p.taps);
    return stages == 3 ? launch_dma128<false, 3>(p, s, "gemm_dma128") : launch_dma128<false, 4>(p, s, "gemm_dma128 (4 stages)");
}

bool imd_gemm_dma_supported(const ConvGemmParams& p) {
    return p.taps == 1 && p.stride == 1 && !p.ups && p.Hin == p.Hout && p.Win == p.Wout && p.Cin == p.K && (p.K % GD_BK) == 0 &&
           p.split_k <= 1 && p.gn_a == nullptr && (p.x_pix_stride % 8) == 0;
}

int imd_launch_gemm_dma(const ConvGemmParams& p, hipStream_t s) {      // p: validated and completed (x_bytes, w_bytes, flags) by imd_launch_conv_gemm
    if (!imd_gemm_dma_supported(p)) return imd_set_error("gemm_dma: needs a plain linear layer with K %% 64 == 0 and no K split (got K=%d taps=%d split=%d)", p.K, p.taps, p.split_k);
    const bool h = p.dtype == IMD_DTYPE_F16;
    const void* kern = h ? reinterpret_cast<const void*>(gemm_dma_kernel<true>) : reinterpret_cast<const void*>(gemm_dma_kernel<false>);
    if (int rc_attr = imd_lds_attr(kern, GD_LDS, "gemm_dma")) return rc_attr;
    const long mt = (p.M + GD_BM - 1) / GD_BM, nt = (p.N + GD_BN - 1) / GD_BN;
    if (h) hipLaunchKernelGGL(gemm_dma_kernel<true>, dim3((unsigned)(mt * nt)), dim3(512), GD_LDS, s, p);
    else hipLaunchKernelGGL(gemm_dma_kernel<false>, dim3((unsigned)(mt * nt)), dim3(512), GD_LDS, s, p);
    return imd_check_launch("gemm_dma");
}
