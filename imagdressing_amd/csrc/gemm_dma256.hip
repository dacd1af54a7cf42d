// 256-row LDS-DMA GEMM for the large plain linear layers -- the GEGLU / feed-forward-out projections of the 32x32 and 16x16 levels
// (8192 x 5120 x 640, 2048 x 10240 x 1280, 8192 x 640 x 2560: ~10 % of the denoising step at 0.23-0.27 of the MFMA peak on the
// 128 x 128 tiles of gemm_dma.hip).  Round 5.  What those shapes pay for on 128 x 128 tiles is not the multiplication (K = 640 is TEN
// 64-deep steps) but what surrounds it once per tile: the launch ramp, the first operands' latency, and an epilogue through LDS that
// nothing overlaps because every workgroup of a residency round reaches it at the same time.  This kernel removes the per-tile costs
// instead of tuning the steps:
//   * 256 x BN x 64 tiles (BN = 128: 48 KB per stage, THREE stages = 144 KB; BN = 256: 64 KB per stage, two stages), 8 waves
//     (4 along M x 2 along N, wave tile 64 x BN/2), both operands global -> LDS by DMA in whole 128-byte rows (piece c of row r at
//     c ^ ((r >> 1) & 7): conflict-free ds_read_b128, whole L2 lines), counted vmcnt;
//   * PERSISTENT: one workgroup per CU walks a list of (tile, K slice) items, and the operand ring is ONE software pipeline across
//     items -- while the last steps of item i are multiplied the first stages of item i + 1 are already landing, so an item's first
//     operands cost no wait and the ramp is paid once per launch;
//   * the epilogue runs FROM REGISTERS: one v_permlane32_swap per accumulator register pair turns the 32x32 MFMA layout (4 channels per
//     lane and register group) into 8 consecutive channels per lane, which go straight through epilogue8 (bias / residual / GEGLU /
//     head-split layouts / fp32 slabs of K slices): no LDS staging, no barrier, and the ring keeps filling underneath it.
// Same arithmetic as the tiled kernels (fp32 accumulation over K in 16-element MFMA steps, ascending), same reference layers
// (diffusers-0.24 BasicTransformerBlock feed-forward: GEGLU proj + out linear).
#include "gemm_common.h"
#include "lds_dma.h"

namespace {

constexpr int G2_BM = 256, G2_BK = 64, G2_ROWB = G2_BK * 2;      // 128-byte rows
constexpr int G2_A = G2_BM * G2_ROWB;                             // 32 KB of activations per stage

struct G2Item { int m0, n0, kt0, nk, slice; };

// item w of `items` = tiles x K slices, XCD-aware: consecutive items of one XCD (w mod 8: the dispatcher hands block b to XCD b mod 8,
// and a persistent block's items stay congruent to its id mod 8 because the grid is a multiple of 8) are neighbours in tile order
__device__ __forceinline__ void g2_decode(const ConvGemmParams& p, unsigned w, unsigned items, int m_tiles, int n_tiles, int bn, int per, int nk_total,
                                          G2Item& o) {
    if (p.flags & 12) {
        const unsigned k = w & 7u, slot = w >> 3, q8 = items >> 3, r8 = items & 7u;
        w = (k < r8 ? k * (q8 + 1) : r8 * (q8 + 1) + (k - r8) * q8) + slot;
    }
    const unsigned tiles = (unsigned)(m_tiles * n_tiles);
    const unsigned slice = w / tiles;
    const unsigned t = w - slice * tiles;
    int tm, tn;
    if (p.flags & 8) { tn = (int)(t / (unsigned)m_tiles); tm = (int)(t - (unsigned)tn * m_tiles); }
    else if ((p.flags & 4) && (p.flags & 16)) {
        constexpr unsigned GM = 8;
        const unsigned per_group = GM * (unsigned)n_tiles;
        const unsigned g = t / per_group, r = t - g * per_group;
        const unsigned rows = min(GM, (unsigned)m_tiles - g * GM);
        tn = (int)(r / rows);
        tm = (int)(g * GM + (r - (unsigned)tn * rows));
    } else { tm = (int)(t / (unsigned)n_tiles); tn = (int)(t - (unsigned)tm * n_tiles); }
    o.m0 = tm * G2_BM; o.n0 = tn * bn; o.slice = (int)slice;
    o.kt0 = (int)slice * per;
    o.nk = min(nk_total, o.kt0 + per) - o.kt0;
}

template <bool F16, int BN, int NST>
__global__ __launch_bounds__(512, 1) void gemm_dma256_kernel(const ConvGemmParams p) {
    using E = El<F16>;
    constexpr int W_BYTES = BN * G2_ROWB;
    constexpr int STAGE = G2_A + W_BYTES;
    constexpr int NPA = 4, NPW = BN / 64, NP = NPA + NPW;          // 1-KB pieces (8 rows x 128 B) per wave and stage: 32 / 8 activation, BN / 64 weight
    constexpr int KEEP = (NST - 2) * NP;                            // pieces of the younger stages that may stay in flight at the per-step wait
    constexpr int WN = BN / 2, NA = WN / 32;                        // wave tile 64 x WN: 2 x NA accumulator blocks
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, col = lane & 31;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * WN;

    const int m_tiles = (p.M + G2_BM - 1) / G2_BM, n_tiles = (p.N + BN - 1) / BN;
    const unsigned items = (unsigned)(m_tiles * n_tiles) * (unsigned)p.split_k;
    const int nk_total = p.K / G2_BK;
    const int per = (nk_total + p.split_k - 1) / p.split_k;
    const int HWo = p.Hout * p.Wout;

    const v4i_t ds_x = raw_rsrc(p.x, p.x_bytes), ds_w = raw_rsrc(p.w, p.w_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    // ---- producer state: the item whose operand stages are being issued ----
    unsigned p_w = blockIdx.x;                    // its index in the item list
    bool p_live = p_w < items;
    G2Item pi = {0, 0, 0, 1, 0};
    int p_kt = 0;
    uint32_t soffA[NPA], soffW[NPW];              // source byte offset of this lane's 16 bytes of each of its pieces at K tile 0 of the item, or OOB
    auto producer_offsets = [&]() {
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            const int q = (j * 8 + wave) * 64 + lane, row = q >> 3, pc = (q & 7) ^ ((row >> 1) & 7);
            soffA[j] = (pi.m0 + row < p.M) ? (uint32_t)(((size_t)(pi.m0 + row) * p.x_pix_stride + pc * 8) * 2) : OOB;
        }
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int q = (j * 8 + wave) * 64 + lane, row = q >> 3, pc = (q & 7) ^ ((row >> 1) & 7);
            soffW[j] = (pi.n0 + row < p.N) ? (uint32_t)(((size_t)(pi.n0 + row) * p.K + pc * 8) * 2) : OOB;
        }
    };
    if (p_live) { g2_decode(p, p_w, items, m_tiles, n_tiles, BN, per, nk_total, pi); producer_offsets(); }
    auto produce = [&](int slot) {                // one stage into ring slot `slot` (past the last item: zero-fill pieces keep the counted waits uniform)
        const uint32_t base = lds0 + (uint32_t)(slot * STAGE);
        const uint32_t wk = (uint32_t)((pi.kt0 + p_kt) * G2_ROWB);
#pragma unroll
        for (int j = 0; j < NPA; ++j)
            dma16(ds_x, base + (uint32_t)(j * 8 + wave) * 1024u, (!p_live || soffA[j] == OOB) ? OOB : soffA[j] + wk);
#pragma unroll
        for (int j = 0; j < NPW; ++j)
            dma16(ds_w, base + (uint32_t)G2_A + (uint32_t)(j * 8 + wave) * 1024u, (!p_live || soffW[j] == OOB) ? OOB : soffW[j] + wk);
        if (p_live && ++p_kt == pi.nk) {          // next item of this workgroup
            p_kt = 0;
            p_w += gridDim.x;
            p_live = p_w < items;
            if (p_live) { g2_decode(p, p_w, items, m_tiles, n_tiles, BN, per, nk_total, pi); producer_offsets(); }
        }
    };

    // ---- consumer state ----
    unsigned c_w = blockIdx.x;
    if (c_w >= items) return;                     // (uniform per workgroup; nothing was issued: p_live was false)
    G2Item ci;
    g2_decode(p, c_w, items, m_tiles, n_tiles, BN, per, nk_total, ci);
    int c_kt = 0;

    f32x16 acc[NA][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    zero_acc();

    // fragment addresses (16-deep slice kk reads piece 2 kk + hi, stored at (2 kk + hi) ^ ((row >> 1) & 7); row offsets are multiples of 32)
    int fo[G2_BK / 16];
#pragma unroll
    for (int kk = 0; kk < G2_BK / 16; ++kk) fo[kk] = col * G2_ROWB + (((2 * kk + hi) ^ ((col >> 1) & 7)) << 4);

    // ---- epilogue of the consumer's item, from registers ----
    auto epilogue = [&]() {
        float* slab = (p.split_k > 1) ? p.splitk_ws + (size_t)ci.slice * p.M * p.N : nullptr;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = ci.m0 + wm0 + b * 32 + col;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    // registers 8 t .. 8 t + 3 hold channels 16 t + 4 hi + 0..3 of the block, 8 t + 4 .. 8 t + 7 channels 16 t + 8 + 4 hi + 0..3:
                    // swapping the upper lanes' copy of the first group with the lower lanes' copy of the second gives every lane the 8
                    // consecutive channels 16 t + 8 hi + 0..7 of its row
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][b][8 * t + e]), __float_as_uint(acc[a][b][8 * t + 4 + e]), false, false);
                        v[e] = __uint_as_float(r[0]); v[4 + e] = __uint_as_float(r[1]);
                    }
                    const int n = ci.n0 + wn0 + a * 32 + 16 * t + 8 * hi;
                    if (m < p.M && n < p.N) {
                        const int nv = (n + 8 <= p.N) ? 8 : 4;
                        if (slab) slab_store8(slab, (size_t)m * p.N + n, make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), nv == 8, false);
                        else epilogue8<F16>(p, v, m, n, nv, HWo);
                    }
                }
            }
        }
    };

    // ---- the pipeline ----
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) produce(t);
    dma_wait_keep_n<KEEP>();
    __syncthreads();
    int slot = 0;
#pragma unroll 1
    for (;;) {
        produce(slot == 0 ? NST - 1 : slot - 1);          // the slot read in the previous step: everybody is past that step's barrier
        {
            const char* Xs = smem + slot * STAGE + wm0 * G2_ROWB;
            const char* Ws = smem + slot * STAGE + G2_A + wn0 * G2_ROWB;
#pragma unroll
            for (int kk = 0; kk < G2_BK / 16; ++kk) {
                uint4 wf[NA], xf[2];
#pragma unroll
                for (int a = 0; a < NA; ++a) wf[a] = *reinterpret_cast<const uint4*>(Ws + a * 32 * G2_ROWB + fo[kk]);
#pragma unroll
                for (int b = 0; b < 2; ++b) xf[b] = *reinterpret_cast<const uint4*>(Xs + b * 32 * G2_ROWB + fo[kk]);
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = E::mfma(wf[a], xf[b], acc[a][b]);
            }
        }
        slot = slot == NST - 1 ? 0 : slot + 1;
        bool done = false;
        if (++c_kt == ci.nk) {                            // item finished: results leave from registers while the next item's stages land
            epilogue();
            zero_acc();
            c_kt = 0;
            c_w += gridDim.x;
            done = c_w >= items;
            if (!done) g2_decode(p, c_w, items, m_tiles, n_tiles, BN, per, nk_total, ci);
            dma_wait();                                   // the epilogue's stores and loads share vmcnt with the DMA pieces: drain, do not count
        } else {
            dma_wait_keep_n<KEEP>();                      // the next stage has landed (this wave's pieces; the younger stages' stay in flight) ...
        }
        if (done) break;
        __syncthreads();                                  // ... and everybody's
    }
}

template <int BN, int NST>
int launch_dma256(const ConvGemmParams& p, bool persistent, hipStream_t s, const char* what) {
    static bool attr_set[2] = {false, false};
    static int n_cu = 0;
    constexpr int LDS = NST * (G2_A + BN * G2_ROWB);
    static_assert(LDS <= 160 * 1024, "ring must fit the CU's LDS");
    const bool h = p.dtype == IMD_DTYPE_F16;
    typedef void (*kern_t)(const ConvGemmParams);
    const kern_t kern = h ? gemm_dma256_kernel<true, BN, NST> : gemm_dma256_kernel<false, BN, NST>;
    if (!attr_set[h]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return imd_set_error("%s: hipFuncSetAttribute failed: %s", what, hipGetErrorString(e));
        attr_set[h] = true;
    }
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return imd_set_error("%s: cannot query the device", what);
        n_cu = prop.multiProcessorCount / 8 * 8;          // a multiple of 8: a block's items stay on its XCD's share of the tile order
        if (n_cu < 8) n_cu = 8;
    }
    const long items = (long)((p.M + G2_BM - 1) / G2_BM) * ((p.N + BN - 1) / BN) * p.split_k;
    const long grid = persistent ? (items < n_cu ? items : n_cu) : items;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, s, p);
    return imd_check_launch(what);
}

}  // namespace

// tile configs 30 (256 x 128 x 64, three stages, persistent), 31 (the same, one item per workgroup), 32 (256 x 256 x 64, two stages, persistent);
// K slices go to fp32 slabs and finish with the tiled kernels' second launch
int imd_launch_gemm_dma256(const ConvGemmParams& p_in, int form, hipStream_t s) {
    ConvGemmParams p = p_in;
    p.splitk_counters = nullptr;
    ConvGemmParams p1 = p;
    p1.split_k = 1;
    if (!imd_gemm_dma_supported(p1)) return imd_set_error("gemm_dma256: needs a plain linear layer with K %% 64 == 0 (got K=%d taps=%d)", p.K, p.taps);
    const int nk_total = p.K / G2_BK, per = (nk_total + p.split_k - 1) / p.split_k;
    if (p.split_k > 1 && (long)(p.split_k - 1) * per >= nk_total) return imd_set_error("gemm_dma256: %d K slices over %d K tiles leave a slice empty", p.split_k, nk_total);
    switch (form) {
        case 0: return launch_dma256<128, 3>(p, true, s, "gemm_dma256 (256x128, persistent)");
        case 1: return launch_dma256<128, 3>(p, false, s, "gemm_dma256 (256x128)");
        case 2: return launch_dma256<256, 2>(p, true, s, "gemm_dma256 (256x256, persistent)");
        default: return imd_set_error("gemm_dma256: unknown form %d", form);
    }
}
