// 256-row LDS-DMA GEMM for the large plain linear layers -- the GEGLU / feed-forward-out / q-k-v projections of the 32x32 and 16x16 levels
// (8192 x 5120 x 640, 2048 x 10240 x 1280, 8192 x 640 x 2560: ~10 % of the denoising step at 0.23-0.27 of the MFMA peak on the
// 128 x 128 tiles of gemm_dma.hip).  Round 5; tile configs 30 (persistent) / 31 (one item per workgroup).  DESIGN.md section 2.2e has the
// ablation history that shaped it.  What those shapes pay for on 128 x 128 tiles is not the multiplication (K = 640 is TEN 64-deep steps) but
// what surrounds it once per tile -- the launch ramp, the first operands' latency, the output stream -- so this kernel is built around those:
//   * 256 x 128 x 64 tiles, THREE 48 KB stages (144 KB), both operands global -> LDS by DMA in whole 128-byte rows (piece c of row r at
//     c ^ ((r >> 1) & 7): conflict-free ds_read_b128, whole L2 lines);
//   * WAVE SPECIALISATION: four producer waves issue every LDS-DMA piece and are the only waves that wait on them; eight consumer waves
//     (4 along M x 2 along N, 64 x 64 each) read fragments, multiply and store.  One workgroup barrier per step is the whole handshake;
//   * PERSISTENT: one 12-wave workgroup per CU walks a list of (tile, K slice) items, and the producers' ring is ONE software pipeline
//     across items -- while the last steps of item i are multiplied the first stages of item i + 1 are already landing;
//   * the epilogue runs FROM REGISTERS: one v_permlane32_swap per accumulator register pair turns the 32x32 MFMA layout (4 channels per
//     lane and register group) into 8 consecutive channels per lane -> epilogue8 (bias / residual / GEGLU / head-split layouts / fp32 slabs
//     of K slices), every load of an item ahead of its first store; row-major 16-bit outputs leave through a wave-private 2 KB LDS
//     transpose so that 4 lanes store 64 contiguous bytes of a row (16-byte fragments drain at the L2's request rate: 2.5 vs 6.2 TB/s).
// Same arithmetic as the tiled kernels (fp32 accumulation over K in 16-element MFMA steps, ascending; bit-identical to tile config 25 without
// K slices), same reference layers (diffusers-0.24 BasicTransformerBlock feed-forward: GEGLU proj + out linear).
#include "gemm_common.h"
#include "lds_dma.h"

namespace {

constexpr int G2_BK = 64, G2_ROWB = G2_BK * 2;                   // 128-byte rows

struct G2Item { int m0, n0, kt0, nk, slice; };

// item w of `items` = tiles x K slices, XCD-aware: consecutive items of one XCD (w mod 8: the dispatcher hands block b to XCD b mod 8,
// and a persistent block's items stay congruent to its id mod 8 because the grid is a multiple of 8) are neighbours in tile order
__device__ __forceinline__ void g2_decode(const ConvGemmParams& p, unsigned w, unsigned items, int m_tiles, int n_tiles, int bm, int bn, int per, int nk_total,
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
    o.m0 = tm * bm; o.n0 = tn * bn; o.slice = (int)slice;
    o.kt0 = (int)slice * per;
    o.nk = min(nk_total, o.kt0 + per) - o.kt0;
}

// Why wave specialisation: vmcnt is PER WAVE and counts loads and stores alike -- a consumer that issued the DMA itself would have to drain its own
// epilogue stores (84 MB per launch at the GEGLU shape: 27-34 us, the HBM write rate) before it could trust a counted wait on the next stage --
// measured with the epilogue removed (profiles/r5b_gemm256_bench.jsonl: 97 -> 63 us).  With the roles split the stores of item i drain while
// item i + 1 is multiplied, and a step's handshake is one workgroup barrier: the producers reach it after their counted wait (stage g + 1 has
// landed), the consumers after their last fragment read of stage g (its slot may be overwritten).

// PRE: no per-batch vector and no K slices -> bias / residual are fetched ahead of the item's last step (see the consumer loop); RES: a residual is given
// BM = 256: eight consumer + four producer waves, one workgroup per CU (tile configs 30 / 31).  (BM = 128 -- four + two waves, a two-stage ring, TWO
// workgroups per CU so that each one's epilogue falls into the other's multiplication -- was measured and is not dispatched: 108-124 us where the
// 256-row form takes 65-72, profiles/r5f_gemm256_bench.jsonl: with two stages a producer can only issue stage g + 2 after the barrier that frees
// stage g, one step of lead is less than the operands' latency.)
// BM = 192 (round 6, tile config 32): six consumer + four producer waves.  For item counts that leave a third of the chip idle at 256 rows --
// feed-forward out, 8192 x 640 x 2560: 32 x 5 = 160 tiles on 256 CUs -- 43 x 5 = 215 tiles of three quarters the work run in ONE round as well.
template <int BM> struct G2Waves { static constexpr int NCW = BM / 32, NPROD = BM == 192 ? 4 : BM / 64; };
template <bool F16, int BM, int BN, int NST, bool PRE, bool RES>
__global__ __launch_bounds__((G2Waves<BM>::NCW + G2Waves<BM>::NPROD) * 64, 3) void gemm_dma256_kernel       // (HIP: the second bound is WAVES PER SIMD -- twelve (ten) waves per CU either way)
(const ConvGemmParams p) {
    using E = El<F16>;
    constexpr int NCW = G2Waves<BM>::NCW, G2_NPROD = G2Waves<BM>::NPROD;   // consumer waves (BM / 64 along M x 2 along N), producer waves
    constexpr int G2_A = BM * G2_ROWB;                               // activation bytes per stage
    constexpr int APIECES = BM / 8;                                  // 1-KB pieces (8 rows x 128 B) of the activation tile
    constexpr int W_BYTES = BN * G2_ROWB;
    constexpr int STAGE = G2_A + W_BYTES;
    constexpr int PIECES = APIECES + BN / 8;                         // pieces per stage: activation, then weight
    constexpr int PP = PIECES / G2_NPROD;                            // ... per producer wave
    static_assert(PIECES % G2_NPROD == 0 && APIECES % G2_NPROD == 0 && (NST - 1) * PP < 64, "pieces in flight per producer wave must fit the 6-bit vmcnt");
    constexpr int KEEP = (NST - 2) * PP;                             // pieces of the younger stages that may stay in flight at the per-step wait
    constexpr int WN = BN / 2, NA = WN / 32;                         // consumer wave tile 64 x WN: 2 x NA accumulator blocks
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= NCW;

    const int m_tiles = (p.M + BM - 1) / BM, n_tiles = (p.N + BN - 1) / BN;
    const unsigned items = (unsigned)(m_tiles * n_tiles) * (unsigned)p.split_k;
    const int nk_total = p.K / G2_BK;
    const int per = (nk_total + p.split_k - 1) / p.split_k;
    const int HWo = p.Hout * p.Wout;
    // timing ablations with WRONG results (tuning knob 2 bits 5..7, tools/gemm256_bench.py): what is a launch made of.  They exist in
    // -DIMD_ABLATIONS builds only (like the attention kernel's): the product library neither compiles them nor lets knob 2 carry the bits
#ifdef IMD_ABLATIONS
    const bool abl_noepi = (p.flags & 32) != 0;      // no epilogue
    const bool abl_nowait = (p.flags & 64) != 0;     // no DMA wait, no barrier in the steps: issue + fragment reads + MFMAs only
    const bool abl_nomfma = (p.flags & 128) != 0;    // no fragment reads / MFMAs: staging, synchronisation and the epilogue
#else
    constexpr bool abl_noepi = false, abl_nowait = false, abl_nomfma = false;
#endif
    if (blockIdx.x >= items) return;                 // (uniform per workgroup)

    if (producer) {
        // ================= producer waves: the operand ring, ONE software pipeline across the workgroup's items =================
        const int pw = wave - NCW;
        const v4i_t ds_x = raw_rsrc(p.x, p.x_bytes), ds_w = raw_rsrc(p.w, p.w_bytes);
        const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
        unsigned p_w = blockIdx.x;                    // the item whose stages are being issued
        bool p_live = true;
        G2Item pi;
        int p_kt = 0;
        uint32_t soff[PP];                            // source byte offset of this lane's 16 bytes of piece (j * NPROD + pw) at K tile 0 of the item, or OOB
        auto offsets = [&]() {
#pragma unroll
            for (int j = 0; j < PP; ++j) {
                const int id = j * G2_NPROD + pw;     // activation pieces first, then weight pieces
                constexpr int JW = APIECES / G2_NPROD; // (APIECES % NPROD == 0: a piece index j is an activation piece in every producer wave or in none)
                const bool isw = j >= JW;
                const int q = (isw ? id - APIECES : id) * 64 + lane, row = q >> 3, pc = (q & 7) ^ ((row >> 1) & 7);
                if (isw) soff[j] = (pi.n0 + row < p.N) ? (uint32_t)(((size_t)(pi.n0 + row) * p.K + pc * 8) * 2) : OOB;
                else soff[j] = (pi.m0 + row < p.M) ? (uint32_t)(((size_t)(pi.m0 + row) * p.x_pix_stride + pc * 8) * 2) : OOB;
            }
        };
        g2_decode(p, p_w, items, m_tiles, n_tiles, BM, BN, per, nk_total, pi);
        offsets();
        auto produce = [&](int slot) {                // one stage into ring slot `slot` (past the last item: zero-fill pieces keep the counted waits uniform)
            const uint32_t base = lds0 + (uint32_t)(slot * STAGE);
            const uint32_t wk = (uint32_t)((pi.kt0 + p_kt) * G2_ROWB);
#pragma unroll
            for (int j = 0; j < PP; ++j) {
                const uint32_t dst = base + (uint32_t)(j * G2_NPROD + pw) * 1024u;      // (weight pieces follow the activation pieces)
                const uint32_t off = (!p_live || soff[j] == OOB) ? OOB : soff[j] + wk;
                if (j >= APIECES / G2_NPROD) dma16(ds_w, dst, off); else dma16(ds_x, dst, off);       // (the leading wait states cover the VALU-written offset register)
            }
            if (p_live && ++p_kt == pi.nk) {          // next item of this workgroup
                p_kt = 0;
                p_w += gridDim.x;
                p_live = p_w < items;
                if (p_live) { g2_decode(p, p_w, items, m_tiles, n_tiles, BM, BN, per, nk_total, pi); offsets(); }
            }
        };
#pragma unroll
        for (int t = 0; t < NST - 1; ++t) produce(t);
        dma_wait_keep_n<KEEP>();
        __syncthreads();                              // barrier 0: stage 0 is in LDS
        int slot = 0;
        for (unsigned w = blockIdx.x; w < items; w += gridDim.x) {      // the same (item, K step) walk as the consumers': one barrier per step
            G2Item it;
            g2_decode(p, w, items, m_tiles, n_tiles, BM, BN, per, nk_total, it);
#pragma unroll 1
            for (int kt = 0; kt < it.nk; ++kt) {
                produce(slot == 0 ? NST - 1 : slot - 1);          // the slot the consumers read in the previous step: they are past that step's barrier
                slot = slot == NST - 1 ? 0 : slot + 1;
                if (!abl_nowait) {
                    dma_wait_keep_n<KEEP>();                      // the next stage has landed (this wave's pieces; the younger stages' stay in flight)
                    __syncthreads();
                }
            }
        }
        dma_wait();                                   // zero-fill pieces past the end must not outlive the workgroup's LDS
        return;
    }

    // ================= consumer waves: fragment reads, MFMAs, epilogue from registers =================
    const int hi = lane >> 5, col = lane & 31;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * WN;
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

    __syncthreads();                                  // barrier 0
    int slot = 0;
    // the epilogue's operands (bias of the lane's 4 column chunks, residual of its 8 output chunks) are fetched at the START of an item's last
    // step and used after it: every load of an item is then older than every store of that item, the compiler's counted waits leave the stores in
    // flight, and nothing in the epilogue waits for memory that it has just written (a load issued between two stores would: vmcnt counts in order)
    constexpr bool pre_ok = PRE;
    for (unsigned w = blockIdx.x; w < items; w += gridDim.x) {
        G2Item ci;
        g2_decode(p, w, items, m_tiles, n_tiles, BM, BN, per, nk_total, ci);
        auto step = [&]() __attribute__((always_inline)) {
            if (!abl_nomfma) {
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
            if (!abl_nowait) __syncthreads();         // the next stage is in LDS (the producers waited for it); everybody is done reading this one
        };
#pragma unroll 1
        for (int kt = 0; kt < ci.nk - 1; ++kt) step();
        // ---- the item's LAST step, with the epilogue's operands fetched in front of it ----
        float4 bn[NA][4];                             // bias in the ACCUMULATOR layout (4 channels per lane and register group): added to the finished sums, then dead
        uint4 rp[NA][2], rq[NA][2];                   // residual chunks of the lane's rows b = 0 (fetched here) and b = 1 (fetched when the epilogue starts, still ahead of every store)
        auto load_bias = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {             // register group j: channels a * 32 + 8 j + 4 hi + 0..3
                    const int nb = ci.n0 + wn0 + a * 32 + 8 * j + 4 * hi;
                    bn[a][j] = (p.bias != nullptr && nb < p.N) ? *reinterpret_cast<const float4*>(p.bias + nb) : make_float4(0, 0, 0, 0);
                }
        };
        float4 bias_l = make_float4(0, 0, 0, 0);
        if (pre_ok && !abl_noepi) {
            // (round 6) with a residual the bias does NOT wait in 32 registers across the last step: beside the 16 + 16 residual registers and the
            // 64 accumulators they pushed the kernel past its 168-register cap (14 VGPRs in scratch, two reloads inside the MFMA region).  Lanes
            // 0..15 fetch the wave's 64 bias values here (one float4 each), park them in the wave's -- otherwise unused: residual tiles store
            // directly -- 2 KB transpose area after the step, and every lane reads its four float4 per accumulator block back from LDS: four
            // live registers instead of 32, no load behind a store.
            if (!RES) load_bias();
            else if (lane < 16) {
                const int nb = ci.n0 + wn0 + 4 * lane;
                if (p.bias != nullptr && nb < p.N) bias_l = *reinterpret_cast<const float4*>(p.bias + nb);
            }
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int n = ci.n0 + wn0 + a * 32 + 16 * t + 8 * hi;
                    const int m = ci.m0 + wm0 + col;
                    rp[a][t] = make_uint4(0, 0, 0, 0);
                    if (RES && m < p.M && n + 8 <= p.N) rp[a][t] = *reinterpret_cast<const uint4*>(p.res + (size_t)m * p.res_ld + n);
                }
        }
        step();
        // ---- item finished: its results leave from registers; nothing here waits for the stores, which drain under the next item's steps ----
        if (!abl_noepi) {
            float* slab = (p.split_k > 1) ? p.splitk_ws + (size_t)ci.slice * p.M * p.N : nullptr;
            if (pre_ok) {
                float* bias_s = reinterpret_cast<float*>(smem + NST * STAGE + wave * 2048);      // (RES: the wave's transpose area is free)
                if (RES && lane < 16) *reinterpret_cast<float4*>(bias_s + 4 * lane) = bias_l;
                // sums + bias: the same fp32 addition epilogue8 performs first (v += bias), in the accumulator layout -- epilogue8 is then handed a zero addend
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 bj = RES ? *reinterpret_cast<const float4*>(bias_s + a * 32 + 8 * j + 4 * hi) : bn[a][j];
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            acc[a][b][4 * j] += bj.x; acc[a][b][4 * j + 1] += bj.y; acc[a][b][4 * j + 2] += bj.z; acc[a][b][4 * j + 3] += bj.w;
                        }
                    }
            }
            // Row-major 16-bit outputs (GEGLU included) of a tile without a channel tail leave through a WAVE-PRIVATE 2 KB transpose in LDS: the
            // register layout gives a lane 8 channels of ONE row, so a direct store instruction touches 32 rows with 32 (GEGLU: 16) bytes each --
            // 16-byte fragments that the L2 takes at its request rate, not its byte rate (84 MB drained at 2.5 TB/s, a plain fill writes at 6.2:
            // profiles/r5d_write_bw_probe.jsonl).  Through the transpose 4 consecutive lanes store 64 contiguous bytes of a row.
            if constexpr (RES) {
                // (round 6) residual tiles -- the launcher sends only row-major 16-bit outputs with whole 128-channel tiles here: the two 32-row halves
                // of the wave tile are software-pipelined so that at most 16 residual registers are alive beside the accumulators: the results of rows
                // b = 0 are finished into 16 packed registers (their accumulators and residual chunks die), THEN the residual of rows b = 1 is requested,
                // then rows b = 0 are stored -- every load still precedes every store -- and rows b = 1 follow.  Holding both halves' residual across the
                // whole epilogue (round 5) sat at the 168-register cap with 14 VGPRs in scratch.
                uint4 pk0[NA][2];
                auto finish = [&](int a, int b, int t, const uint4& rr) __attribute__((always_inline)) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][b][8 * t + e]), __float_as_uint(acc[a][b][8 * t + 4 + e]), false, false);
                        v[e] = __uint_as_float(r[0]); v[4 + e] = __uint_as_float(r[1]);
                    }
                    uint4 pk = make_uint4(0, 0, 0, 0);
                    epilogue8<F16>(p, v, ci.m0 + wm0 + b * 32 + col, ci.n0 + wn0 + a * 32 + 16 * t + 8 * hi, 8, HWo, true, make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), true, rr, &pk);
                    return pk;
                };
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int t = 0; t < 2; ++t) pk0[a][t] = finish(a, 0, t, rp[a][t]);
                const int m1 = ci.m0 + wm0 + 32 + col;
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        rq[a][t] = make_uint4(0, 0, 0, 0);
                        if (m1 < p.M) rq[a][t] = *reinterpret_cast<const uint4*>(p.res + (size_t)m1 * p.res_ld + ci.n0 + wn0 + a * 32 + 16 * t + 8 * hi);
                    }
                if (m1 - 32 < p.M) {
                    bf16_t* orow = reinterpret_cast<bf16_t*>(p.out) + (size_t)(m1 - 32) * p.out_ld + ci.n0 + wn0 + 8 * hi;
#pragma unroll
                    for (int a = 0; a < NA; ++a)
#pragma unroll
                        for (int t = 0; t < 2; ++t) *reinterpret_cast<uint4*>(orow + a * 32 + 16 * t) = pk0[a][t];
                }
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int t = 0; t < 2; ++t) pk0[a][t] = finish(a, 1, t, rq[a][t]);
                if (m1 < p.M) {
                    bf16_t* orow = reinterpret_cast<bf16_t*>(p.out) + (size_t)m1 * p.out_ld + ci.n0 + wn0 + 8 * hi;
#pragma unroll
                    for (int a = 0; a < NA; ++a)
#pragma unroll
                        for (int t = 0; t < 2; ++t) *reinterpret_cast<uint4*>(orow + a * 32 + 16 * t) = pk0[a][t];
                }
            } else {
            const bool stage_ok = slab == nullptr && p.mode == OUT_ROWMAJOR && !p.out_f32 && ci.n0 + BN <= p.N;     // (with a residual: direct stores -- the outputs are
                                                                                                                              // small there and the prefetched residual fills the registers)
            if (stage_ok) {
                const bool geglu = p.act == ACT_GEGLU;
                char* st = smem + NST * STAGE + wave * 2048;
                const int sw = (col >> 2) & 3;        // slot swizzle of row `col` (rows are 64 bytes: 16 banks)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int m = ci.m0 + wm0 + b * 32 + col;
                    // chunk (a, t) of this lane: 8 channels -> 16 bytes, or 4 GEGLU outputs -> 8 bytes
                    auto chunk = [&](int a, int t) __attribute__((always_inline)) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][b][8 * t + e]), __float_as_uint(acc[a][b][8 * t + 4 + e]), false, false);
                            v[e] = __uint_as_float(r[0]); v[4 + e] = __uint_as_float(r[1]);
                        }
                        const int n = ci.n0 + wn0 + a * 32 + 16 * t + 8 * hi;
                        uint4 pk = make_uint4(0, 0, 0, 0);
                        if (m < p.M) {
                            if (pre_ok) epilogue8<F16>(p, v, m, n, 8, HWo, true, make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), false, make_uint4(0, 0, 0, 0), &pk);
                            else epilogue8<F16>(p, v, m, n, 8, HWo, false, make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), false, make_uint4(0, 0, 0, 0), &pk);
                        }
                        return pk;
                    };
                    if (geglu) {                      // one pass: 32 rows x 64 bytes (the wave's 32 output channels); 8-byte chunk c of row r at slot c ^ (sw << 1)
#pragma unroll
                        for (int a = 0; a < NA; ++a)
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                const uint4 pk = chunk(a, t);
                                *reinterpret_cast<uint2*>(st + col * 64 + (((4 * a + 2 * t + hi) ^ (sw << 1)) << 3)) = make_uint2(pk.x, pk.y);
                            }
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int idx = lane + 64 * i, row = idx >> 2, q = (idx & 3) ^ ((row >> 2) & 3);
                            const uint4 val = *reinterpret_cast<const uint4*>(st + idx * 16);
                            const int mm = ci.m0 + wm0 + b * 32 + row;
                            if (mm < p.M) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)mm * p.out_ld + ((ci.n0 + wn0) >> 1) + q * 8) = val;
                        }
                    } else {                          // one pass per 32-channel block a: 32 rows x 64 bytes; 16-byte chunk c = 2 t + hi of row r at slot c ^ sw
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
#pragma unroll
                            for (int t = 0; t < 2; ++t) *reinterpret_cast<uint4*>(st + col * 64 + (((2 * t + hi) ^ sw) << 4)) = chunk(a, t);
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int idx = lane + 64 * i, row = idx >> 2, c = (idx & 3) ^ ((row >> 2) & 3);
                                const uint4 val = *reinterpret_cast<const uint4*>(st + idx * 16);
                                const int mm = ci.m0 + wm0 + b * 32 + row;
                                if (mm < p.M) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)mm * p.out_ld + ci.n0 + wn0 + a * 32 + c * 8) = val;
                            }
                        }
                    }
                }
            } else {
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
                            else if (pre_ok) epilogue8<F16>(p, v, m, n, nv, HWo, true, make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0));
                            else epilogue8<F16>(p, v, m, n, nv, HWo);
                        }
                    }
                }
            }
            }
            }       // !RES
        }
        zero_acc();
    }
}

template <int BM, int BN, int NST, bool PRE, bool RES>
int launch_dma256_v(const ConvGemmParams& p, bool persistent, hipStream_t s, const char* what) {
    constexpr int LDS = NST * (BM + BN) * G2_ROWB + (BM / 32) * 2048;      // ring + the consumer waves' 2 KB store-transpose areas
    constexpr int WG_PER_CU = BM >= 192 ? 1 : 2;
    static_assert(LDS * WG_PER_CU <= 160 * 1024, "ring + transpose areas must fit the CU's LDS");
    const bool h = p.dtype == IMD_DTYPE_F16;
    typedef void (*kern_t)(const ConvGemmParams);
    const kern_t kern = h ? gemm_dma256_kernel<true, BM, BN, NST, PRE, RES> : gemm_dma256_kernel<false, BM, BN, NST, PRE, RES>;
    // per DEVICE (a process may drive several GPUs): the 160 KB dynamic-LDS attribute and the CU count that sizes the persistent grid
    // (a multiple of 8: a block's items stay on its XCD's share of the tile order)
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), LDS, what)) return rc_attr;
    const int n_cu = imd_cu_count8();
    if (n_cu < 0) return imd_set_error("%s: cannot query the device", what);
    const long items = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.split_k;
    const long slots = (long)n_cu * WG_PER_CU;
    const long grid = persistent ? (items < slots ? items : slots) : items;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((G2Waves<BM>::NCW + G2Waves<BM>::NPROD) * 64), LDS, s, p);
    return imd_check_launch(what);
}

template <int BM, int BN, int NST>
int launch_dma256(const ConvGemmParams& p, bool persistent, hipStream_t s, const char* what) {
    const bool pre = p.rowvec == nullptr && p.split_k <= 1, res = p.res != nullptr;
    // (the generic epilogue8 path fetches what it needs itself; the residual-prefetching form takes row-major 16-bit outputs with whole channel tiles only)
    if (!pre || (res && (p.mode != OUT_ROWMAJOR || p.out_f32 || p.act == ACT_GEGLU || (p.N % BN) != 0 || (p.res_ld % 8) != 0 || (p.out_ld % 8) != 0)))
        return launch_dma256_v<BM, BN, NST, false, false>(p, persistent, s, what);
    return res ? launch_dma256_v<BM, BN, NST, true, true>(p, persistent, s, what) : launch_dma256_v<BM, BN, NST, true, false>(p, persistent, s, what);
}

}  // namespace

// tile configs 30 (256 x 128 x 64, three stages, persistent), 31 (the same, one item per workgroup) and 32 (192 x 128 x 64, persistent: round 6, for
// tile counts that strand CUs at 256 rows); K slices go to fp32 slabs and finish with the tiled kernels' second launch.  (A 256 x 256 x 64 two-stage form behind the same template -- tile config 32 of the first draft -- was slower on every
// feed-forward shape, 88-107 us against 55-97, profiles/r5b_gemm256_bench.jsonl: one stage of lead is not enough; it is not built.)
int imd_launch_gemm_dma256(const ConvGemmParams& p_in, int form, hipStream_t s) {
    ConvGemmParams p = p_in;
    p.splitk_counters = nullptr;
    ConvGemmParams p1 = p;
    p1.split_k = 1;
    if (!imd_gemm_dma_supported(p1)) return imd_set_error("gemm_dma256: needs a plain linear layer with K %% 64 == 0 (got K=%d taps=%d)", p.K, p.taps);
    const int nk_total = p.K / G2_BK, per = (nk_total + p.split_k - 1) / p.split_k;
    if (p.split_k > 1 && (long)(p.split_k - 1) * per >= nk_total) return imd_set_error("gemm_dma256: %d K slices over %d K tiles leave a slice empty", p.split_k, nk_total);
    switch (form) {
        case 0: return launch_dma256<256, 128, 3>(p, true, s, "gemm_dma256 (256x128, persistent)");
        case 1: return launch_dma256<256, 128, 3>(p, false, s, "gemm_dma256 (256x128)");
        case 2: return launch_dma256<192, 128, 3>(p, true, s, "gemm_dma256 (192x128, persistent)");
        default: return imd_set_error("gemm_dma256: unknown form %d", form);
    }
}
