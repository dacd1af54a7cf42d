// Fused dual-softmax ("hybrid") attention for head dim 40 -- the UNet level-0 kernel (N = M = 4096 at 512x512: 88 % of the
// hybrid-attention FLOPs of a denoising step, the kernel bench.py's roofline times).
//
//   O[b, q, h*40:(h+1)*40] = softmax(Q K1^T) V1  +  s2[b] * softmax(Q K2^T) V2
//
// = RefSAttnProcessor2_0.__call__ between the projections (/root/reference/adapter/attention_processor.py:589-612): frozen
// self attention over the image tokens plus an independently normalised cross attention over the garment tokens, the two
// half-precision results added (:612).  Same operand layouts, same C ABI entry (imd_attention) and the same arithmetic
// ideas as the generic kernel in attention.hip (swapped S^T = K Q^T so that exp2'd, packed scores ARE the B operand of
// O^T += V^T P^T; deferred row maximum folded into the QK^T MFMA through the pad column; softmax denominator produced by
// the P.V MFMA through an all-ones V^T row); what differs is the SCHEDULE:
//
//   * software pipeline over 32-key steps, written out in the source: in step j a wave issues the QK^T MFMAs of block
//     j+1, the P.V MFMAs of block j-1 and the exp2 / pack VALU work of block j -- three independent instruction streams in
//     ONE basic block (14 MFMAs against ~60 VALU), which hipcc interleaves (sched_group_barrier pattern below) so that the
//     matrix pipe runs underneath the transcendental work of the same wave instead of waiting for the other wave;
//   * the only branch of a step is the (rare) "a score ran past the deferred maximum" test, taken once per step for both
//     query blocks; the exact path behind it recomputes the block's scores from K in L2 (three 16-byte loads per lane),
//     so the hot path keeps no copy of them.  First / ragged / past-the-end blocks use the same path (forced);
//   * K and V^T are staged skewed by one block (unit u = K rows [64u+32, 64u+96) + V^T columns [64u-32, 64u+32)), so a
//     unit serves exactly the two steps of one loop iteration and the double buffer still needs ONE barrier per 64 keys;
//   * the next unit's global loads are issued unconditionally at the top of an iteration and consumed by the LDS writes
//     at its end (a conditional prefetch makes the staging registers loop-carried and hipcc then waits for them at once);
//   * the phase-0 (self) result of a two-phase row is parked in LDS (16-bit, the rounding the reference's first SDPA
//     output has), not in the output buffer: no HBM round trip, no store -> load dependency at the phase boundary.
#include <type_traits>
#include "lds_dma.h"

#include "common.h"
#include "imd_kernels.h"

namespace {

constexpr int D = 40, DPK = 48, DPV = 64, NKT = 3;
constexpr int KT = 64;                      // keys per staged unit (two 32-key steps)
constexpr int KSTR = DPK * 2 + 16;          // 112 B per K row in LDS  (7 x 16 B: conflict-free b128 fragment reads)
constexpr int VSTR = KT * 2 + 16;           // 144 B per V^T row       (9 x 16 B)
constexpr int VROWS = D + 1;                // 40 head-dim rows + the all-ones row that yields the softmax denominator
constexpr int VBYTES = VROWS * VSTR;        // rows 41..63 of the second 32-row MFMA block are NOT stored: their fragment
constexpr int BUF = VBYTES + KT * KSTR;     // reads run into the K rows behind (finite garbage into accumulator rows nobody reads)
constexpr int PARK = 4 * 5 * 1024;          // phase-0 result: 4 waves x 20 packed dwords per lane
constexpr int PARK_T = 4 * 6 * 1024;        // ... x 24 packed dwords per lane with the 16x16x32 tail (VAR & 8192)
static_assert((DPV - VROWS) * VSTR <= KT * KSTR, "phantom V^T rows must stay inside the K rows");
// LDS-DMA staging (VAR & 128): `buffer_load ... lds` writes lane-linear (wave-uniform base + lane * 16 B), so rows cannot be
// padded; bank conflicts are avoided by choosing WHICH 16-byte piece of global memory a lane fetches instead:
//   K  : 96-byte rows back to back; the six pieces of rows 8..15 (mod 16) are stored rotated by three
//   V^T: 128-byte rows back to back; piece c of row d is stored at position c ^ ((d >> 1) & 7)
// (with either map the 16 rows a ds_read_b128 lane group touches fall on 16 distinct 16-byte bank slots)
constexpr int KSTR_D = DPK * 2, VSTR_D = KT * 2, VBYTES_D = VROWS * VSTR_D, BUF_D = VBYTES_D + KT * KSTR_D;
constexpr int NRING = 3;                    // LDS-DMA ring: unit u + 2 is in flight while unit u is consumed (an L2 miss served by the
                                            // Infinity Cache takes about as long as one iteration: one unit of lead was not enough)
// LDS per workgroup (LDS-DMA variants): NRING * BUF_D + park + 1 KB dump for the fourth wave's third (empty) DMA piece
static_assert((DPV - VROWS) * VSTR_D <= KT * KSTR_D, "phantom V^T rows must stay inside the K rows");

typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4u;
__device__ __forceinline__ uint4 buf_load16(const __amdgpu_buffer_rsrc_t& rs, uint32_t byte_off) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
constexpr uint32_t OOB = 0xffffffffu;

// packed 3-input maximum: on non-negative 16-bit float patterns (fp16 OR bf16) it is the maximum of the patterns as
// integers, with inf / NaN patterns propagating
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c) {      // -> one v_pk_maximum3_f16
    const h2_t x = __builtin_bit_cast(h2_t, a), y = __builtin_bit_cast(h2_t, b), z = __builtin_bit_cast(h2_t, c);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(x, y), z));
}

// ------------------------------------------------------------------------------------------------------------------------------
// PROJ (VAR & 1048576, round 3): the block's out-projection (to_out[0] + bias + residual, attention_processor.py:614-617) inside the
// attention launch -- the level-0 hybrid block becomes two launches (norm1 + q/k/v projection, this one).  The projection needs
// all 8 heads of a row, a workgroup owns one head of 256 rows, and the heads of a row block run on different XCDs (private,
// non-coherent L2s), so the heads are joined the way gemm_common.h joins K slices: every workgroup stores its O tile WRITE-THROUGH
// (sc0 sc1), waits for the acknowledgements, then takes a ticket from a device-scope counter of its (batch, row block); the
// workgroup that draws the last ticket reads the 256 x 320 O tile back from the memory side (sc0 sc1 loads, never a stale L1 / L2
// line), multiplies it with W_o out of LDS and writes  residual + bias + O W_o^T.  Nobody waits for anybody (no deadlock, no spin);
// the counter is left at zero.  Result: identical up to fp32 summation order to imd_conv_gemm on the same O.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int AUX_WT = 17;                  // buffer cache policy bits sc0 (1) | sc1 (16): write-through stores / memory-side loads
typedef __attribute__((__vector_size__(2 * sizeof(uint32_t)))) uint32_t v2u;
constexpr int PJ_C = 320, PJ_KC = 32, PJ_ROWS = 128, PJ_NH = 160;
constexpr int PJ_LDS = 64 * (PJ_NH + 4) * 4;               // the epilogue tile (64 rows x 164 fp32 = 41 KB) > 8 KB of O rows + 10 KB of W_o rows per chunk

template <bool F16>
__device__ __forceinline__ void attn_out_proj(const AttnParams& p, int b, int row0, int nrows, char* smem) {
    using E = El<F16>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, col = lane & 31;
    char* As = smem;                                            // [128 O rows][64 B]
    char* Ws = smem + PJ_ROWS * 64;                             // [160 W_o rows][64 B]
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.proj_w), 0, PJ_C * PJ_C * 2, 0x00020000);
    // unpadded 64-byte LDS rows, piece c of row r at c ^ ((r >> 2) & 3) (gemm_dma.hip's map: conflict-free ds_read_b128).
    // A pass = 128 rows x 160 channels (wave w: rows 32 w .. 32 w + 31, five 32-channel blocks = 80 accumulator registers); a chunk =
    // 32 of the 320 input channels: 128 x 4 O pieces (two per thread) + 160 x 4 W_o pieces (two or three per thread)
    int a_dst[2], w_dst[3];
    uint32_t w_src0[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = tid + 256 * i, row = s >> 2, pc = s & 3;
        a_dst[i] = row * 64 + ((pc ^ ((row >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int s = tid + 256 * i, row = s >> 2, pc = s & 3;
        w_dst[i] = row * 64 + ((pc ^ ((row >> 2) & 3)) << 4);
        w_src0[i] = s < PJ_NH * 4 ? (uint32_t)((row * PJ_C + pc * 8) * 2) : OOB;
    }
    const int frow = 32 * wave + col;                           // this lane's O row inside the 128-row pass
    const int a_fr = frow * 64, a_sw = (frow >> 2) & 3;
    constexpr int NCH = PJ_C / PJ_KC, PF = 3;                   // 10 chunks; operands fetched THREE chunks ahead (O comes from the memory side)
    for (int ps = 0; ps * PJ_ROWS < 2 * nrows; ++ps) {          // passes: (row half, channel half)
        const int r_base = row0 + (ps >> 1) * PJ_ROWS, nh = (ps & 1) * PJ_NH;
        if (r_base - row0 >= nrows) break;
        uint32_t a_src[2], w_src[3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = tid + 256 * i, row = s >> 2, pc = s & 3;
            a_src[i] = (r_base + row < p.N) ? (uint32_t)((((size_t)b * p.N + r_base + row) * p.out_ld + pc * 8) * 2) : OOB;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) w_src[i] = w_src0[i] == OOB ? OOB : w_src0[i] + (uint32_t)(nh * PJ_C * 2);
        f32x16 acc[5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        v4u a_r[PF][2], w_r[PF][3];
        auto fetch = [&](int ch, v4u (&ad)[2], v4u (&wd)[3]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) ad[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_o, (int)(a_src[i] == OOB ? OOB : a_src[i] + ch * PJ_KC * 2), 0, AUX_WT);
#pragma unroll
            for (int i = 0; i < 3; ++i) wd[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(w_src[i] == OOB ? OOB : w_src[i] + ch * PJ_KC * 2), 0, 0);
        };
#pragma unroll
        for (int c = 0; c < PF; ++c) fetch(c, a_r[c], w_r[c]);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            __syncthreads();                                    // the previous chunk's fragment reads are done
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<v4u*>(As + a_dst[i]) = a_r[ch % PF][i];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < 2 || tid < PJ_NH * 4 - 512) *reinterpret_cast<v4u*>(Ws + w_dst[i]) = w_r[ch % PF][i];
            if (ch + PF < NCH) fetch(ch + PF, a_r[ch % PF], w_r[ch % PF]);     // issued BEFORE the matrix work of this chunk
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < PJ_KC / 16; ++kk) {
                const uint4 xf = *reinterpret_cast<const uint4*>(As + a_fr + (((2 * kk + hi) ^ a_sw) << 4));
                uint4 wf[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int wr = 32 * j + col;
                    wf[j] = *reinterpret_cast<const uint4*>(Ws + wr * 64 + (((2 * kk + hi) ^ ((wr >> 2) & 3)) << 4));
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[j] = E::mfma(wf[j], xf, acc[j]);
            }
        }
        // epilogue through LDS, 64 rows at a time (the accumulators hold one row per lane: stored directly that is 8-byte pieces at a
        // 640-byte stride, store-issue-bound; from LDS every thread moves 16 contiguous bytes of a row): residual + bias + O W_o^T
        float* Cs = reinterpret_cast<float*>(smem);
        constexpr int CLDP = PJ_NH + 4, CPRP = PJ_NH / 8;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            __syncthreads();                                    // fragment reads / the previous half's reads are done
            if ((wave >> 1) == half) {
                const int row_l = 32 * (wave & 1) + col;
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(Cs + row_l * CLDP + 32 * j + 8 * q + 4 * hi) =
                            make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
            }
            __syncthreads();
            for (int c = tid; c < 64 * CPRP; c += 256) {
                const int row_l = c / CPRP, cc = (c - row_l * CPRP) * 8;
                const int r = r_base + half * 64 + row_l;
                if (r >= p.N) continue;
                const float4 v0 = *reinterpret_cast<const float4*>(Cs + row_l * CLDP + cc);
                const float4 v1 = *reinterpret_cast<const float4*>(Cs + row_l * CLDP + cc + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                const int c0 = nh + cc;
                if (p.proj_b) {
                    const float4 b0 = *reinterpret_cast<const float4*>(p.proj_b + c0), b1 = *reinterpret_cast<const float4*>(p.proj_b + c0 + 4);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                }
                if (p.proj_res) {
                    const uint4 rr = *reinterpret_cast<const uint4*>(p.proj_res + ((size_t)b * p.N + r) * p.proj_res_ld + c0);
                    v[0] += E::lo(rr.x); v[1] += E::hi(rr.x); v[2] += E::lo(rr.y); v[3] += E::hi(rr.y);
                    v[4] += E::lo(rr.z); v[5] += E::hi(rr.z); v[6] += E::lo(rr.w); v[7] += E::hi(rr.w);
                }
                *reinterpret_cast<uint4*>(p.proj_out + ((size_t)b * p.N + r) * p.proj_out_ld + c0) =
                    make_uint4(E::pack2(v[0], v[1]), E::pack2(v[2], v[3]), E::pack2(v[4], v[5]), E::pack2(v[6], v[7]));
            }
        }
    }
}

// VAR bit 0: MFMA / VALU interleave written out and pinned   bit 1: plain (not XCD-aware) work order
//   8192 (TAIL): head-dim rows 32..40 of O^T += V^T P^T (+ the all-ones row 40 that yields the softmax denominator) on
//   v_mfma_f32_16x16x32 instead of a second 32x32x16 row block: 16 padded rows instead of 32, i.e. 6 instead of 7
//   MFMA-equivalents per query block and 32-key step (issued / algorithmic matrix work 1.41 -> 1.21).  The packed P registers
//   of a 32x32 S^T tile hold one query per LANE COLUMN (l & 31); the 16x16x32 B operand wants one query per l & 15 with the
//   four 16-lane rows carrying four 8-key slot groups: ONE v_permlane16_swap_b32 per register pair (g = 0 word, g = 1 word)
//   turns them into the operands of the two 16-query halves (lane row k then carries keys {0..7, 16..23, 8..15, 24..31}[k]
//   of the step; the V^T fragment is read in the same slot order).  Accumulators: 2 x (16 + 2 x 4) instead of 2 x 32.
// ABLATION bits (timing experiments only, results are WRONG): 4: exp2 replaced by a move   8: no barrier in the loop
//   16: no global loads / LDS stores in the loop   32: no P.V MFMAs   64: no QK^T MFMAs   256: one LDS fragment read per step
//   512: 100 KB of LDS per workgroup (one workgroup = one wave per SIMD)   2048: no packed-max / overflow test
//   4096: s_memtime instrumentation (per-phase cycle sums of every wave's loop, added into the first words of `out`)
//   32768 (WG512): 8 waves = 512 queries per workgroup (one workgroup per CU instead of two): a staged K / V^T unit serves twice
//   the queries, i.e. every wave issues 2 instead of 3 LDS-DMA pieces per 64 keys (the DMA issue sequence costs ~15 % of the
//   4-wave kernel: profiles/r3b_attn_ablations.jsonl) and there is one barrier domain per CU.  LDS-DMA staging only.
template <bool F16, int THR, int VAR>
__device__ __attribute__((always_inline)) bool attn40_body(const AttnParams& p) {
    using E = El<F16>;
    constexpr bool DMA = (VAR & 128) != 0;
    constexpr bool TAIL = (VAR & 8192) != 0;
    constexpr bool PROJ = (VAR & 1048576) != 0;            // out-projection fused: O tiles stored write-through, last head of a row block projects
    // STAT (VAR & 2097152, round 4): the main loop unrolled over the three ring slots, so that every LDS address of an iteration -- fragment
    // reads AND the LDS-DMA destinations -- is a compile-time offset; the staging sequence of a piece shrinks from ~15 instructions (ring
    // arithmetic, out-of-range selects compiled to branches, M0 save / restore, five wait states) to three (s_add m0 / s_nop / buffer_load)
    // plus one running-offset add.  The issue-mix probe (tools/probes/issue_mix_probe.hip) puts this kernel's step within 15 % of what its
    // MFMA + VALU + LDS + DMA mix costs in isolation; what is left above that is scalar / address bookkeeping like this.
    constexpr bool STAT = (VAR & 2097152) != 0;
    constexpr bool UNCHK = (VAR & 4194304) != 0;          // (with STAT) interior steps without the overflow test: experiment, see variant 13
    // DUP (VAR & 8388608, round 6): the phase-0 (self-attention) result of every row is ALSO stored to p.out_dup -- the first hybrid block of
    // the CFG batch: cond and uncond rows of an image have bit-identical Q / K / V there (same latent, same timestep, nothing text- or
    // garment-dependent upstream), so the uncond row's whole output IS the cond row's first phase.  The launch runs the cond rows only
    // (8 instead of 12 phase units at batch 4) and writes both: out = O1/l1 + s2 O2/l2 (cond), out_dup = O1/l1 (uncond) -- the same
    // 16-bit rounding of O1/l1 the one-phase row stores and the two-phase row parks, hence bit-identical to the 2B-row launch.
    constexpr bool DUP = (VAR & 8388608) != 0;
    static_assert(!DUP || (STAT && !PROJ), "the duplicated phase-0 store lives in the static-ring kernel without the fused out-projection");
    // fp16 under UNCHK (round 5): the phase's reference maximum is the first block's maximum PLUS this bias, i.e. P = 2^(s - m_first - 4):
    // the first block's largest P is 2^-4 and fp16's 65504 is reached only by a score 20 base-2 units (a factor 10^6 in weight) above the
    // first 32 keys' maximum -- the end-of-phase denominator test (inf / NaN) then re-runs the workgroup checked.  The price is at the
    // other end: P below 2^-14 is an fp16 subnormal (the MFMA keeps subnormal inputs, tools/probes/mfma_subnormal_probe.hip), below
    // 2^-25 it is 0 -- keys more than 21 units below the first maximum (weight < 5e-7 each) instead of 25 in the checked kernel.
    constexpr float FIRST_BIAS = (F16 && UNCHK) ? 4.0f : 0.0f;
    static_assert(!STAT || ((VAR & 128) && (VAR & 8192) && !(VAR & 32768)), "the static-ring loop exists for the 4-wave LDS-DMA kernel with the 16x16x32 tail");
    static_assert(!PROJ || (TAIL && !(VAR & 32768)), "the fused out-projection lives in the 4-wave kernel with the 16x16x32 tail");
    constexpr int NW = (VAR & 32768) ? 8 : 4;              // waves per workgroup
    constexpr int NT = NW * 64;
    constexpr int NPIECE = (11 + NW - 1) / NW;             // LDS-DMA pieces every wave issues per unit (dummies included)
    static_assert(NW == 4 || DMA, "the 8-wave workgroup exists for the LDS-DMA staging only");
    constexpr int PARKW = (TAIL ? PARK_T : PARK) / 4;      // phase-0 park bytes per wave
    constexpr int PARKB = NW * PARKW;
    constexpr int BUFB = DMA ? BUF_D : BUF, VBY = DMA ? VBYTES_D : VBYTES;
    constexpr float OFFS_THR = (float)THR;                 // deferred maximum: P <= 2^THR (base-2 units)
    constexpr int QPAD_T = D / 16, QPAD_HI = (D % 16) / 8;  // Q fragment / half-wave holding pad slot 40 (element 0 of the fragment)
    constexpr int L_DT = D / 32, L_REG = ((D % 32) & 3) + 4 * ((D % 32) >> 3), L_HI = ((D % 32) >> 2) & 1;   // accumulator row 40
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hi = lane >> 5;
    const int col = lane & 31;
    // XCD-aware work list (hardware block L runs on XCD L % 8): q-tile fastest, then batch, then head; every XCD gets a
    // contiguous slice, i.e. all q-tiles that re-read one (batch, head)'s K / V^T share one L2, and every XCD sees the same mix
    // of two-phase (garment) and one-phase rows
    int wx, h, b;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, total = gx * gy * gz;
        const unsigned Lb = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const unsigned xcd = Lb & 7u, slot = Lb >> 3;
        const unsigned q8 = total >> 3, r8 = total & 7u;          // bijective also when total % 8 != 0
        unsigned w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
        if ((VAR & 2) || (p.flags & 1)) w = Lb;
        wx = (int)(w % gx);
        b = (int)((w / gx) % gz);
        h = (int)(w / (gx * gz));
    }
    const int q0 = (wx * NW + wave) * 64;

    // V^T row 40 (all ones) of both buffers: written once, never restaged
    {
        constexpr int PADV = (DMA ? VSTR_D : VSTR) / 16;
        const uint32_t one2 = E::pack2(1.0f, 1.0f);
        for (int v = tid; v < (DMA ? NRING : 2) * PADV; v += NT)
            *reinterpret_cast<uint4*>(smem + (v / PADV) * BUFB + D * (DMA ? VSTR_D : VSTR) + (v % PADV) * 16) = make_uint4(one2, one2, one2, one2);
    }

    // Q fragments (B operand of S^T = K Q^T): lane = query column, 8 head-dim values per fragment
    uint4 qf[2][NKT];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + qb * 32 + col;
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q < p.N) v = *reinterpret_cast<const uint4*>(p.q + ((size_t)(b * p.H + h) * p.N + q) * DPK + t * 16 + hi * 8);
            qf[qb][t] = v;
        }
    }

    float w2 = 0.f;
    if (p.k2 != nullptr && p.scale2 != nullptr) w2 = p.scale2[b];
    const int nph = (w2 != 0.f) ? 2 : 1;

    const int kfrag = swap23(col) * KSTR + hi * 16;      // permuted key row of this lane (see common.h::swap23)
    const int vfrag = col * VSTR + hi * 16;
    // DMA layout: fragment addresses of this lane (K chunk 2f + hi of row swap23(col); V^T piece 2g + hi of row col)
    int kfd[3], vfd[4];
    {
        const int kr = swap23(col), rot = 3 * ((kr >> 3) & 1), sw = (col >> 1) & 7;
#pragma unroll
        for (int f = 0; f < 3; ++f) kfd[f] = kr * KSTR_D + ((2 * f + hi + rot) % 6) * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g) vfd[g] = col * VSTR_D + (((2 * g + hi) ^ sw) * 16);
    }
    // TAIL: V^T rows 32..47 x the step's 32 keys as the A operand of v_mfma_f32_16x16x32: lane l = row 32 + (l & 15), slot
    // group l >> 4 = keys {0, 16, 8, 24}[l >> 4] + 0..7 (the order the permlane16-swapped P registers carry).  DMA layout:
    // piece c of row 32 + r sits at position c ^ tail_sw(r) (source-side swizzle chosen for THIS read: the 16 lanes of a
    // ds_read_b128 group cover all 16 rows, eight of them at piece p and eight at piece p ^ 2 -> 16 distinct bank slots)
    int tfd[2];
    {
        const int r = lane & 15, kq = lane >> 4, pc0 = ((kq & 1) << 1) | (kq >> 1);
        const int tsw = ((r >> 1) & 7) ^ ((r >= 4 && r < 12) ? 2 : 0);
#pragma unroll
        for (int sec = 0; sec < 2; ++sec)
            tfd[sec] = DMA ? (32 + r) * VSTR_D + (((4 * sec + pc0) ^ tsw) * 16) : (32 + r) * VSTR + (4 * sec + pc0) * 16;
    }
    char* park = smem + (DMA ? NRING : 2) * BUFB + wave * PARKW + lane * 16;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    unsigned long long tm_slots = 0, tm_check = 0, tm_sync = 0, tm_loop = 0, tm_steps = 0;      // VAR & 4096 only
    bool bad = false;               // UNCHK: a softmax denominator of this lane is not finite
    for (int ph = 0; ph < nph; ++ph) {
        f32x16 o[2][2];
        f32x4 ot[2][2];                 // TAIL: rows 32..47 of O^T, [query block][16-query half]; o[.][1] is unused then
        float m_ref[2] = {0.f, 0.f};
        auto reset_state = [&]() __attribute__((always_inline)) {      // (once per phase; twice when an UNCHK phase has to be re-run checked)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                m_ref[qb] = 0.f;
                if (hi == QPAD_HI) qf[qb][QPAD_T].x &= 0xffff0000u;       // Q pad slot = -m_ref = 0
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[qb][dt] = zero16;
                ot[qb][0] = zero4; ot[qb][1] = zero4;
            }
        };
        const int L = ph ? p.L2 : p.L1;
        const int LP = ph ? p.L2P : p.L1P;
        const int kvb = ph ? (b / p.kv2_bdiv) : (b / p.kv1_bdiv);
        const bf16_t* kbase = (ph ? p.k2 : p.k1) + (size_t)(kvb * p.H + h) * L * DPK;
        const bf16_t* vbase = (ph ? p.v2t : p.v1t) + (size_t)(kvb * p.H + h) * DPV * LP;
        const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kbase), 0, (uint32_t)L * DPK * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vbase), 0, (uint32_t)D * LP * 2, 0x00020000);
        const int J = (L + 31) >> 5;                 // 32-key blocks
        const int NU = (J >> 1) + 1;                 // units: steps 0 .. 2 NU - 1 cover P.V of block J - 1

        // staging slots of this thread: K vectors v0 = tid, v1 = tid + 256 (< 384), V^T vectors likewise (< 320)
        const int kr0 = tid / 6, kc0 = tid % 6, kr1 = (tid + 256) / 6, kc1 = (tid + 256) % 6;
        const int vr0 = tid >> 3, vc0 = tid & 7, vr1 = (tid + 256) >> 3;
        const bool k1ok = tid < 128, v1ok = tid < 64;
        uint4 kreg0, kreg1, vreg0, vreg1;
        auto load_unit = [&](int u) {        // K rows [64u+32, 64u+96), V^T columns [64u-32, 64u+32); out of range reads 0
            const int krow = u * KT + 32, vcol = u * KT - 32;
            kreg0 = buf_load16(rs_k, (krow + kr0) >= 0 ? (uint32_t)((krow + kr0) * (DPK * 2) + kc0 * 16) : OOB);
            kreg1 = buf_load16(rs_k, (k1ok && (krow + kr1) >= 0) ? (uint32_t)((krow + kr1) * (DPK * 2) + kc1 * 16) : OOB);
            const int c = vcol + vc0 * 8;
            const bool cok = c >= 0 && c < LP;
            vreg0 = buf_load16(rs_v, cok ? (uint32_t)((vr0 * LP + c) * 2) : OOB);
            vreg1 = buf_load16(rs_v, (v1ok && cok) ? (uint32_t)((vr1 * LP + c) * 2) : OOB);
        };
        auto store_unit = [&](int bufi) {
            char* Vs = smem + bufi * BUF;
            char* Ks = Vs + VBYTES;
            const uint32_t one = (uint32_t)E::fromf(1.0f);
            if (kc0 == D / 8) kreg0.x = (kreg0.x & 0xffff0000u) | one;      // K[:, 40] = 1: the pad column that carries -m_ref
            if (kc1 == D / 8) kreg1.x = (kreg1.x & 0xffff0000u) | one;
            *reinterpret_cast<uint4*>(Ks + kr0 * KSTR + kc0 * 16) = kreg0;
            if (k1ok) *reinterpret_cast<uint4*>(Ks + kr1 * KSTR + kc1 * 16) = kreg1;
            *reinterpret_cast<uint4*>(Vs + vr0 * VSTR + vc0 * 16) = vreg0;
            if (v1ok) *reinterpret_cast<uint4*>(Vs + vr1 * VSTR + vc0 * 16) = vreg1;
        };

        // ---- LDS-DMA staging: 11 wave-instructions per unit (6 KB of K = 6, 5 KB of V^T = 5), instruction q = wave + 4 i ----
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const v4i_t ds_k = raw_rsrc(kbase, (uint32_t)L * DPK * 2), ds_v = raw_rsrc(vbase, (uint32_t)D * LP * 2);
        const uint32_t smem_base = (uint32_t)(uintptr_t)smem;          // LDS byte address of the dynamic region
        int dsrc[NPIECE];            // byte offset of this lane's piece inside K / V^T for unit 0
        bool dneg[NPIECE];           // V^T pieces that lie before column 0 in unit 0 (K rows before row 0 in unit -1)
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const int qi = wv + NW * i;
            if (qi >= 11) {
                dsrc[i] = 0; dneg[i] = true;
            } else if (qi < 6) {
                const int sl = 64 * qi + lane, r = sl / 6, cs = sl % 6, c = (cs + 6 - 3 * ((r >> 3) & 1)) % 6;
                dsrc[i] = (32 + r) * (DPK * 2) + c * 16;
                dneg[i] = r < 32;                                // unit -1: rows 64 u + 32 + r = r - 32
            } else {
                const int sl = 64 * (qi - 6) + lane, dd = sl >> 3, rt = dd - 32;
                const int sw = (TAIL && dd >= 32) ? (((rt >> 1) & 7) ^ ((rt >= 4 && rt < 12) ? 2 : 0)) : ((dd >> 1) & 7);
                const int ch = (sl & 7) ^ sw;
                dsrc[i] = (dd * LP + ch * 8 - 32) * 2;
                dneg[i] = ch < 4;                                // unit 0: columns ch * 8 - 32 .. < 0
            }
        }
        // STAT: per-piece loop state.  Wave 3's third piece (index 11: there are only 11 pieces) re-fetches piece 10 -- the same bytes to the
        // same place, a benign duplicate -- so that every wave issues three REAL pieces and nothing in the loop depends on the wave.
        v4i_t sdsc[NPIECE];          // descriptor of the piece (K or V^T)
        uint32_t sdst[NPIECE];       // LDS byte address of the piece inside ring slot 0
        uint32_t scur[NPIECE];       // source byte offset of the piece for the NEXT unit to stage (running; += sstr per unit)
        uint32_t sstr[NPIECE];
        auto init_running = [&]() __attribute__((always_inline)) {
        if constexpr (STAT) {
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) {
                const int qi = min(wv + NW * i, 10);
                const bool isk = qi < 6;
                int src;
                if (isk) {
                    const int sl = 64 * qi + lane, r = sl / 6, cs = sl % 6, c = (cs + 6 - 3 * ((r >> 3) & 1)) % 6;
                    src = (32 + r) * (DPK * 2) + c * 16;
                } else {
                    const int sl = 64 * (qi - 6) + lane, dd = sl >> 3, rt = dd - 32;
                    const int sw = (dd >= 32) ? (((rt >> 1) & 7) ^ ((rt >= 4 && rt < 12) ? 2 : 0)) : ((dd >> 1) & 7);
                    src = (dd * LP + ((sl & 7) ^ sw) * 8 - 32) * 2;
                }
                sstr[i] = isk ? (uint32_t)(KT * DPK * 2) : (uint32_t)(KT * 2);
                scur[i] = (uint32_t)src + 2u * sstr[i];                      // the loop's first staged unit is unit 2
                sdst[i] = smem_base + (uint32_t)(isk ? VBYTES_D + 1024 * qi : 1024 * (qi - 6));
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) sdsc[i][c4] = isk ? ds_k[c4] : ds_v[c4];
            }
        }
        };
        auto dma_static = [&](auto slot_c) __attribute__((always_inline)) {      // stage the next unit into ring slot `slot_c` (compile-time)
            constexpr int SLOT = decltype(slot_c)::value;
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) {
                asm volatile("s_add_u32 m0, %1, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                             : : "v"(scur[i]), "s"(sdst[i]), "s"(sdsc[i]), "n"(SLOT * BUF_D) : "memory", "scc");
                scur[i] += sstr[i];
            }
        };
        auto dma_unit = [&](int u, int bufi) {     // u >= 1 inside the loop: no range logic at all (rows / columns past the end read 0)
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) {      // EVERY wave issues exactly NPIECE pieces (the counted vmcnt relies on it)
                const int qi = wv + NW * i;
                const bool isk = qi < 6, none = qi >= 11;
                const bool bad = none || (isk ? (u < 0 && dneg[i]) : (u <= 0 && (u < 0 || dneg[i])));
                const uint32_t off = bad ? OOB : (uint32_t)(dsrc[i] + u * (isk ? KT * DPK * 2 : KT * 2));
                const uint32_t dst = none ? smem_base + NRING * BUF_D + PARKB
                                          : smem_base + bufi * BUF_D + (isk ? VBYTES_D + 1024 * qi : 1024 * (qi - 6));
                if ((VAR & 65536) && u >= 2) dma16_nonop(isk ? ds_k : ds_v, dst, off);      // (loop pieces: SGPR operands are loop-carried SALU values)
                else dma16(isk ? ds_k : ds_v, dst, off);
            }
        };

        f32x16 sa[2], sb[2];
        uint4 pa[2][2], pb[2][2];
        auto reset_p = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int g = 0; g < 2; ++g) pb[qb][g] = make_uint4(0, 0, 0, 0);      // P of "block -1"
        };

        // ---- one 32-key step: S_n = QK^T(block j+1), O += V^T P^T(block j-1), P_c = exp2(S_c) (block j) ----
        // 14 MFMAs (fragment f = 0..2: K chunks -> S_n, f = 3..6: V^T (dt, g) -> O; two query blocks each), 16 "pairs" of
        // scores (2 exp2 + 1 pack each) and 8 packed max3.  VAR & 1: the interleave is written out slot by slot -- MFMA i,
        // then the exp2 of pair i+1, the pack of pair i and every other slot a max3 -- and pinned with sched_barrier, with
        // the LDS fragment reads issued two fragments (four MFMAs) ahead.  Otherwise the three streams are emitted one
        // after the other and hipcc's scheduler decides.
        // TAIL schedule: fragment reads run PD fragments (2 PD MFMA slots) ahead of their use, ring of PD + 1 registers
        // (VAR & 131072: PD = 3, VAR & 262144: PD = 4; default 2)
        constexpr int PD = (VAR & 262144) ? 4 : (VAR & 131072) ? 3 : 2;
        constexpr int NFR = TAIL ? PD + 1 : 3;
        uint4 fr[NFR];               // LDS fragment ring: fragment f of a step lives in fr[(R0 + f) % 3]  (TAIL: (R0T + f) % (PD + 1))
        auto step = [&](const char* Vs, auto second_c, int j, f32x16 (&sc)[2], f32x16 (&sn)[2], uint4 (&pc)[2][2],
                        uint4 (&pp)[2][2], auto chk_c) __attribute__((always_inline)) {
            constexpr bool SECOND = decltype(second_c)::value;        // second step of a unit: K block kb = 1, V^T groups 2, 3
            // CHK = false (UNCHK kernels, interior steps only): no packed-max tracking, no overflow / first / ragged-block test, no exact path --
            // the caller guarantees a complete block j > 0 and checks the softmax denominators for overflow at the end of the phase
            constexpr bool CHK = decltype(chk_c)::value;
            constexpr int KB = SECOND ? 1 : 0, G0 = SECOND ? 2 : 0, R0 = SECOND ? 1 : 0;
            const char* Ks = Vs + VBY;
            auto frag = [&](int f, int kb, int g0) -> uint4 {
                if ((VAR & 256) && f > 0) f = 0;
                if (DMA)
                    return f < 3 ? *reinterpret_cast<const uint4*>(Ks + kb * 32 * KSTR_D + kfd[f])
                                 : *reinterpret_cast<const uint4*>(Vs + ((f - 3) >> 1) * 32 * VSTR_D + vfd[g0 + ((f - 3) & 1)]);
                return f < 3 ? *reinterpret_cast<const uint4*>(Ks + kb * 32 * KSTR + kfrag + f * 32)
                             : *reinterpret_cast<const uint4*>(Vs + ((f - 3) >> 1) * 32 * VSTR + vfrag + (g0 + ((f - 3) & 1)) * 32);
            };
            auto mma = [&](int i, const uint4& fa) {
                const int f = i >> 1, qb = i & 1;
                if (f < 3) { if (!(VAR & 64) || f == 0) sn[qb] = E::mfma(fa, qf[qb][f], f == 0 ? zero16 : sn[qb]); }
                else if (!(VAR & 32)) o[qb][(f - 3) >> 1] = E::mfma(fa, pp[qb][(f - 3) & 1], o[qb][(f - 3) >> 1]);
            };
            auto exp_pair = [&](int pr) {
                const int qb = pr >> 3, r0 = 2 * (pr & 7);
                if (VAR & 4) return;
                sc[qb][r0] = __builtin_amdgcn_exp2f(sc[qb][r0]);
                sc[qb][r0 + 1] = __builtin_amdgcn_exp2f(sc[qb][r0 + 1]);
            };
            auto word = [&](int w) -> uint32_t& {                    // packed word w = 0..15: (qb, g, component)
                uint4& v = pc[w >> 3][(w >> 2) & 1];
                return (w & 3) == 0 ? v.x : (w & 3) == 1 ? v.y : (w & 3) == 2 ? v.z : v.w;
            };
            auto cvt_pair = [&](int pr) {
                const int qb = pr >> 3, r0 = 2 * (pr & 7);
                word(pr) = E::pack2(sc[qb][r0], sc[qb][r0 + 1]);
            };
            uint32_t mq[2] = {0u, 0u};
            if constexpr (TAIL) {
                // fragment sequence of a step: 0, 1 = V^T rows 0..31 x key groups G0, G0 + 1;  2..4 = K chunks 0..2;  5 = V^T tail.
                // MFMA slots: 0..3 O^T(rows 0..31) += V^T P^T (the previous block's P in its 32x32 layout), 4..9 S_n = K Q^T,
                // 10..13 the four 16x16x32 tail MFMAs (query block, 16-query half) on P swapped in place during slots 4..11.
                auto fragT = [&](int sq, int sec) -> uint4 {
                    const int g0 = sec ? 2 : 0;
                    if ((VAR & 256) && sq > 0) sq = 0;
                    if (sq < 2) return DMA ? *reinterpret_cast<const uint4*>(Vs + vfd[g0 + sq])
                                           : *reinterpret_cast<const uint4*>(Vs + vfrag + (g0 + sq) * 32);
                    if (sq < 5) return DMA ? *reinterpret_cast<const uint4*>(Ks + sec * 32 * KSTR_D + kfd[sq - 2])
                                           : *reinterpret_cast<const uint4*>(Ks + sec * 32 * KSTR + kfrag + (sq - 2) * 32);
                    return *reinterpret_cast<const uint4*>(Vs + tfd[sec]);
                };
                constexpr int R0T = SECOND ? (6 % NFR) : 0;       // ring slot of this step's fragment 0 (6 fragments per step)
                auto mmaT = [&](int i) {
                    const uint4& fa = fr[(R0T + (i < 10 ? (i >> 1) : 5)) % NFR];
                    if (i < 4) o[i & 1][0] = E::mfma(fa, pp[i & 1][i >> 1], o[i & 1][0]);
                    else if (i < 10) sn[i & 1] = E::mfma(fa, qf[i & 1][(i >> 1) - 2], i < 6 ? zero16 : sn[i & 1]);
                    else ot[(i - 10) >> 1][i & 1] = E::mfma16(fa, pp[(i - 10) >> 1][i & 1], ot[(i - 10) >> 1][i & 1]);
                };
                auto swapT = [&](int k) {        // word k & 3 of query block k >> 2: (g = 0, g = 1) -> (queries 0..15, queries 16..31)
                    uint4& a = pp[k >> 2][0];
                    uint4& b = pp[k >> 2][1];
                    uint32_t& x = (k & 3) == 0 ? a.x : (k & 3) == 1 ? a.y : (k & 3) == 2 ? a.z : a.w;
                    uint32_t& y = (k & 3) == 0 ? b.x : (k & 3) == 1 ? b.y : (k & 3) == 2 ? b.z : b.w;
                    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
                    x = r[0]; y = r[1];
                };
                if (!SECOND) {       // (the second step's first PD fragments were read during the first step's last slots)
#pragma unroll
                    for (int f = 0; f < PD; ++f) fr[f] = fragT(f, 0);
                }
                exp_pair(0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 14; ++i) {
                    if ((i & 1) == 0 && i <= 10) {      // slot pair of fragment i / 2: read fragment i / 2 + PD (this step's, else the next step's)
                        const int f2 = (i >> 1) + PD;
                        if (f2 <= 5) fr[(R0T + f2) % NFR] = fragT(f2, SECOND ? 1 : 0);
                        else if (!SECOND && f2 - 6 < PD) fr[(R0T + f2) % NFR] = fragT(f2 - 6, 1);
                    }
                    // STAT kernels pin the order INSIDE a slot as well (fragment read, exp2 pair, pack, max, lane swap, then the MFMA):
                    // left to the compiler the order changes with unrelated code (r4: 1 % either way between builds)
                    if (STAT) __builtin_amdgcn_sched_barrier(0);
                    if (!STAT) mmaT(i);
                    exp_pair(i + 1);
                    if (STAT) __builtin_amdgcn_sched_barrier(0);
                    cvt_pair(i);
                    if (STAT) __builtin_amdgcn_sched_barrier(0);
                    if ((i & 1) && !(VAR & 2048) && CHK) mq[0] = pk_max3(mq[0], word(i - 1), word(i));
                    if (i >= 4 && i < 12) swapT(i - 4);
                    if (STAT) { __builtin_amdgcn_sched_barrier(0); mmaT(i); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                exp_pair(15);
                cvt_pair(14);
                cvt_pair(15);
                if (!(VAR & 2048) && CHK) mq[0] = pk_max3(mq[0], word(14), word(15));
            } else if (VAR & 1) {
                if (!SECOND) {       // (the second step's first two fragments were read during the first step's last slots)
                    fr[0] = frag(0, KB, G0);
                    fr[1] = frag(1, KB, G0);
                }
                exp_pair(0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 14; ++i) {
                    if ((i & 1) == 0) {
                        const int f2 = (i >> 1) + 2;
                        if (f2 <= 6) fr[(R0 + f2) % 3] = frag(f2, KB, G0);
                        else if (!SECOND) fr[f2 % 3] = frag(f2 - 7, 1, 2);       // next step's fragments 0 and 1
                    }
                    mma(i, fr[(R0 + (i >> 1)) % 3]);
                    exp_pair(i + 1);
                    cvt_pair(i);
                    if ((i & 1) && !(VAR & 2048)) mq[0] = pk_max3(mq[0], word(i - 1), word(i));
                    __builtin_amdgcn_sched_barrier(0);
                }
                exp_pair(15);
                cvt_pair(14);
                cvt_pair(15);
                if (!(VAR & 2048)) mq[0] = pk_max3(mq[0], word(14), word(15));
            } else {
#pragma unroll
                for (int i = 0; i < 14; ++i) mma(i, frag(i >> 1, KB, G0));
#pragma unroll
                for (int pr = 0; pr < 16; ++pr) exp_pair(pr);
#pragma unroll
                for (int pr = 0; pr < 16; ++pr) cvt_pair(pr);
#pragma unroll
                for (int w = 0; w < 16; w += 2) mq[(w >> 3) & 1] = pk_max3(mq[(w >> 3) & 1], word(w), word(w + 1));
            }
            if constexpr (CHK) {
            uint32_t mm = pk_max3(mq[0], mq[1], mq[1]);
            // pin the speculative exp2 / pack work in THIS basic block: without it hipcc sinks it below the `forced` test
            // (its results are dead on the exact path) and the MFMAs above lose the VALU work they are meant to hide
            asm volatile("" : "+v"(mm));
            const uint32_t top = max(mm & 0xffffu, mm >> 16);
            // first block (j == 0) or ragged / past-the-end block (j >= L / 32): ONE unsigned compare, folded into the overflow test's
            // threshold so that the hot path has a single branch per step (forced: every lane passes `top >= 0`)
            const bool forced = (uint32_t)(j - 1) >= (uint32_t)((L >> 5) - 1);
            const uint32_t thr1 = forced ? 0u : (uint32_t)E::fromf(__builtin_exp2f(OFFS_THR)) + 1u;
            if (__builtin_expect(__any(top >= thr1), 0)) {
                // ---- exact path (first block, ragged / past-the-end block, or a score ran past the deferred maximum): the
                // block's scores are recomputed from K in global memory (L2), with the pad slot as it stands ----
                const int krow = j * 32 + swap23(col);
                uint4 kfs[NKT];
#pragma unroll
                for (int tk = 0; tk < NKT; ++tk)
                    kfs[tk] = buf_load16(rs_k, (uint32_t)(krow * (DPK * 2) + tk * 32 + hi * 16));    // rows >= L read as 0
                if (hi == QPAD_HI) kfs[QPAD_T].x = (kfs[QPAD_T].x & 0xffff0000u) | (uint32_t)E::fromf(1.0f);   // K[:, 40] = 1
#pragma unroll
                for (int tk = 0; tk < NKT; ++tk)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) sc[qb] = E::mfma(kfs[tk], qf[qb][tk], tk == 0 ? zero16 : sc[qb]);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    if (j * 32 + 32 > L) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = j * 32 + 8 * hi + (r & 7) + 16 * (r >> 3);
                            if (key >= L) sc[qb][r] = -INFINITY;
                        }
                    }
                    float mx = sc[qb][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[qb][r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    const bool first = (j == 0);
                    if (first || __any(mx > OFFS_THR)) {
                        const float want = m_ref[qb] + (first ? mx + FIRST_BIAS : fmaxf(mx, 0.f));
                        const float nref = E::tof(E::fromf(want));          // what the 16-bit Q slot can carry
                        const float delta = nref - m_ref[qb];
                        if (!first) {       // on the first block O is still 0 (and delta may be hugely negative: 2^-delta = inf)
                            const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                            for (int dt = 0; dt < (TAIL ? 1 : 2); ++dt)
#pragma unroll
                                for (int r = 0; r < 16; ++r) o[qb][dt][r] *= alpha;
                            if constexpr (TAIL) {       // tail accumulators: lane l holds query 16 hq + (l & 15) of the block
#pragma unroll
                                for (int hq = 0; hq < 2; ++hq) {
                                    const float at = __shfl(alpha, (lane & 15) + 16 * hq);
#pragma unroll
                                    for (int r = 0; r < 4; ++r) ot[qb][hq][r] *= at;
                                }
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) { sc[qb][r] -= delta; sn[qb][r] -= delta; }
                        m_ref[qb] = nref;
                        if (hi == QPAD_HI) qf[qb][QPAD_T].x = (qf[qb][QPAD_T].x & 0xffff0000u) | (uint32_t)E::fromf(-nref);
                    }
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        pc[qb][g].x = E::pack2(__builtin_amdgcn_exp2f(sc[qb][8 * g + 0]), __builtin_amdgcn_exp2f(sc[qb][8 * g + 1]));
                        pc[qb][g].y = E::pack2(__builtin_amdgcn_exp2f(sc[qb][8 * g + 2]), __builtin_amdgcn_exp2f(sc[qb][8 * g + 3]));
                        pc[qb][g].z = E::pack2(__builtin_amdgcn_exp2f(sc[qb][8 * g + 4]), __builtin_amdgcn_exp2f(sc[qb][8 * g + 5]));
                        pc[qb][g].w = E::pack2(__builtin_amdgcn_exp2f(sc[qb][8 * g + 6]), __builtin_amdgcn_exp2f(sc[qb][8 * g + 7]));
                    }
                }
            }
            }       // CHK
            if (VAR & 4096) tm_steps += 1;
        };

        {
        reset_state(); reset_p(); init_running();
        constexpr bool checked_run = !UNCHK;
        // ---- prologue: unit -1 (K block 0 in its second half) -> S_a = QK^T(block 0); unit 0 staged behind it ----
        if (DMA) {
            dma_unit(-1, 2);
            dma_unit(0, 0);
            dma_unit(1, 1);
            dma_wait();
        } else {
            load_unit(-1);
            store_unit(1);
            load_unit(0);
        }
        __syncthreads();
        {
            const char* Ks = smem + (DMA ? 2 : 1) * BUFB + VBY;           // unit -1
#pragma unroll
            for (int tk = 0; tk < NKT; ++tk) {
                const uint4 kf = DMA ? *reinterpret_cast<const uint4*>(Ks + 32 * KSTR_D + kfd[tk])
                                     : *reinterpret_cast<const uint4*>(Ks + 32 * KSTR + kfrag + tk * 32);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) sa[qb] = E::mfma(kf, qf[qb][tk], tk == 0 ? zero16 : sa[qb]);
            }
        }
        if (!DMA) store_unit(0);
        __syncthreads();

        // units 0 .. NUF - 1 hold two real key blocks each; the last unit holds block J - 1 (J odd) and / or only owes the
        // P.V of block J - 1 ("drain"): handled after the loop so that the loop body has no per-step range logic
        // (register staging keeps the simpler form: NU full iterations, past-the-end blocks masked to P = 0 on the exact path)
        const int NUF = DMA ? (J >> 1) : NU;
        auto iteration_sync = [&]() __attribute__((always_inline)) {
            if (DMA) { if (NPIECE == 3) dma_wait_keep3(); else dma_wait_keep2(); }   // this wave's pieces of unit u + 1 have landed (u + 2 stays in flight) ...
            if (!(VAR & 8)) __syncthreads();                        // ... and so have everybody else's
        };
        int ring = 0;                                               // u % NRING
        unsigned long long t_loop0 = 0;
        if (VAR & 4096) t_loop0 = __builtin_amdgcn_s_memtime();
        if (STAT && !(UNCHK && checked_run)) {
            auto body = [&](auto slot_c, int u, auto chk_c) __attribute__((always_inline)) {      // iteration u on ring slot u % 3 = slot_c (compile-time)
                constexpr int SLOT = decltype(slot_c)::value;
                dma_static(std::integral_constant<int, (SLOT + 2) % 3>{});          // unit u + 2; its buffer was last read in iteration u - 1
                const char* Vs = smem + SLOT * BUFB;
                step(Vs, std::false_type{}, 2 * u, sa, sb, pa, pb, chk_c);
                step(Vs, std::true_type{}, 2 * u + 1, sb, sa, pb, pa, chk_c);
                iteration_sync();
            };
            const std::integral_constant<int, 0> c0{}; const std::integral_constant<int, 1> c1{}; const std::integral_constant<int, 2> c2{};
            constexpr std::integral_constant<bool, !UNCHK> interior_chk{};          // UNCHK: interior iterations run without the per-step test
            int u = 0;
            if constexpr (UNCHK) {
                // iteration 0 (first block: forced exact path) and the last one to three (a ragged block may sit there) keep the checked step
                if (NUF > 0) { body(c0, 0, std::true_type{}); u = 1; ring = 1; }
                for (; u + 3 <= NUF - 1; u += 3) { body(c1, u, interior_chk); body(c2, u + 1, interior_chk); body(c0, u + 2, interior_chk); }
                if (u < NUF) {
                    body(c1, u, std::true_type{}); ++u; ring = 2;
                    if (u < NUF) {
                        body(c2, u, std::true_type{}); ++u; ring = 0;
                        if (u < NUF) { body(c0, u, std::true_type{}); ++u; ring = 1; }
                    }
                }
            } else {
                for (; u + 3 <= NUF; u += 3) { body(c0, u, std::true_type{}); body(c1, u + 1, std::true_type{}); body(c2, u + 2, std::true_type{}); }
                if (u < NUF) {
                    body(c0, u, std::true_type{}); ++u; ring = 1;
                    if (u < NUF) { body(c1, u, std::true_type{}); ++u; ring = 2; }
                }
            }
        } else
        for (int u = 0; u < NUF; ++u) {
            if (DMA) {
                if (!(VAR & 16)) dma_unit(u + 2, ring == 0 ? 2 : ring - 1);      // buffer (u+2) % 3 was last read in iteration u-1
            } else if (!(VAR & 16)) load_unit(u + 1);
            const char* Vs = smem + (DMA ? ring : (u & 1)) * BUFB;
            ring = ring == NRING - 1 ? 0 : ring + 1;
            step(Vs, std::false_type{}, 2 * u, sa, sb, pa, pb, std::true_type{});
            step(Vs, std::true_type{}, 2 * u + 1, sb, sa, pb, pa, std::true_type{});
            if (!DMA && !(VAR & 16)) store_unit((u + 1) & 1);
            unsigned long long t_s = 0;
            if (VAR & 4096) t_s = __builtin_amdgcn_s_memtime();
            iteration_sync();
            if (VAR & 4096) tm_sync += __builtin_amdgcn_s_memtime() - t_s;
        }
        if (VAR & 4096) tm_loop += __builtin_amdgcn_s_memtime() - t_loop0;
        if (DMA) {
            const char* Vs = smem + ring * BUFB;                     // unit NUF
            auto drain = [&](int g0, uint4 (&pp)[2][2]) {      // O += V^T P^T of the last block; nothing left to score
#pragma unroll
                for (int dt = 0; dt < (TAIL ? 1 : 2); ++dt)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const uint4 vf = DMA ? *reinterpret_cast<const uint4*>(Vs + dt * 32 * VSTR_D + vfd[g0 + g])
                                             : *reinterpret_cast<const uint4*>(Vs + dt * 32 * VSTR + vfrag + (g0 + g) * 32);
#pragma unroll
                        for (int qb = 0; qb < 2; ++qb) o[qb][dt] = E::mfma(vf, pp[qb][g], o[qb][dt]);
                    }
                if constexpr (TAIL) {
                    const uint4 tf = *reinterpret_cast<const uint4*>(Vs + tfd[g0 >> 1]);
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        uint32_t* a = reinterpret_cast<uint32_t*>(&pp[qb][0]);
                        uint32_t* b = reinterpret_cast<uint32_t*>(&pp[qb][1]);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const auto r = __builtin_amdgcn_permlane16_swap(a[c], b[c], false, false);
                            a[c] = r[0]; b[c] = r[1];
                        }
#pragma unroll
                        for (int hq = 0; hq < 2; ++hq) ot[qb][hq] = E::mfma16(tf, pp[qb][hq], ot[qb][hq]);
                    }
                }
            };
            if (J & 1) {
                step(Vs, std::false_type{}, J - 1, sa, sb, pa, pb, std::true_type{});
                drain(2, pa);
            } else {
                drain(0, pb);
            }
            dma_wait();               // pieces of units past the end are still landing: the next phase restages every buffer
            __syncthreads();
        }
        if constexpr (UNCHK) {
            // row 40 of the tail accumulators (reg 0 of lanes 32..47) holds the denominators of this lane's queries: not finite, or so
            // large (>= 2^100) that a numerator sum |P.v| <= denominator * max|v| may have left fp32's range
            if (lane >= 32 && lane < 48) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq) bad |= (__float_as_uint(ot[qb][hq][0]) & 0x7f800000u) >= (227u << 23);
            }
        }
        }

        // ---- end of phase: normalise; phase 0 of a two-phase row is parked in LDS rounded to the element type (the
        // reference adds two half-precision SDPA outputs, attention_processor.py:612) and read back by the same lane ----
        if constexpr (TAIL) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                // softmax denominator of query 16 hq + n: accumulator row 40 = reg 0 of lane 32 + n in tail accumulator hq
                const float l0 = __shfl(ot[qb][0][0], 32 + (lane & 15));
                const float l1 = __shfl(ot[qb][1][0], 32 + (lane & 15));
                const float inv = ((ph == 1) ? w2 : 1.0f) / ((col & 16) ? l1 : l0);
                uint32_t pk[12];
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) {               // jj < 4: rows 8 jj + 4 hi + 0..3 of query `col`;  4, 5: tail half hq = jj - 4,
                    float v[4];                                //          rows 32 + 4 (lane >> 4) + 0..3 of query 16 hq + (lane & 15)
                    if (jj < 4) {
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) v[e2] = o[qb][0][4 * jj + e2] * inv;
                    } else {
                        const float it = __shfl(inv, (lane & 15) + 16 * (jj - 4));
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) v[e2] = ot[qb][jj - 4][e2] * it;
                    }
                    if (ph == 1) {
                        const uint2 prev = *reinterpret_cast<const uint2*>(park + (qb * 6 + jj) / 2 * 1024 + ((qb * 6 + jj) & 1) * 8);
                        v[0] += E::lo(prev.x); v[1] += E::hi(prev.x); v[2] += E::lo(prev.y); v[3] += E::hi(prev.y);
                    }
                    pk[2 * jj] = E::pack2(v[0], v[1]);
                    pk[2 * jj + 1] = E::pack2(v[2], v[3]);
                }
                if constexpr (DUP) {
                    if (ph == 0) {                             // the paired (uncond) row's output: this row's self-attention result
                        const int q = q0 + qb * 32 + col;
                        if (q < p.N) {
                            bf16_t* orow = p.out_dup + ((size_t)b * p.N + q) * p.out_ld + h * D;
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
                                *reinterpret_cast<uint2*>(orow + 8 * jj + 4 * hi) = make_uint2(pk[2 * jj], pk[2 * jj + 1]);
                        }
                        if (lane < 32) {
#pragma unroll
                            for (int hq = 0; hq < 2; ++hq) {
                                const int qt = q0 + qb * 32 + 16 * hq + (lane & 15);
                                if (qt < p.N)
                                    *reinterpret_cast<uint2*>(p.out_dup + ((size_t)b * p.N + qt) * p.out_ld + h * D + 32 + 4 * (lane >> 4)) =
                                        make_uint2(pk[8 + 2 * hq], pk[9 + 2 * hq]);
                            }
                        }
                    }
                }
                if (ph == 0 && nph == 2) {
#pragma unroll
                    for (int jj = 0; jj < 6; ++jj)
                        *reinterpret_cast<uint2*>(park + (qb * 6 + jj) / 2 * 1024 + ((qb * 6 + jj) & 1) * 8) = make_uint2(pk[2 * jj], pk[2 * jj + 1]);
                } else {
                    const int q = q0 + qb * 32 + col;
                    if constexpr (PROJ) {                      // write-through (sc0 sc1): the projecting workgroup may sit on another XCD
                        const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0x80000000u, 0x00020000);
                        if (q < p.N) {
                            const int off = (int)((((size_t)b * p.N + q) * p.out_ld + h * D) * 2);
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const v2u v = {pk[2 * jj], pk[2 * jj + 1]};
                                __builtin_amdgcn_raw_buffer_store_b64(v, rs_o, off + (8 * jj + 4 * hi) * 2, 0, AUX_WT);
                            }
                        }
                        if (lane < 32) {
#pragma unroll
                            for (int hq = 0; hq < 2; ++hq) {
                                const int qt = q0 + qb * 32 + 16 * hq + (lane & 15);
                                if (qt < p.N) {
                                    const v2u v = {pk[8 + 2 * hq], pk[9 + 2 * hq]};
                                    __builtin_amdgcn_raw_buffer_store_b64(v, rs_o, (int)((((size_t)b * p.N + qt) * p.out_ld + h * D + 32 + 4 * (lane >> 4)) * 2), 0, AUX_WT);
                                }
                            }
                        }
                    } else {
                    if (q < p.N) {
                        bf16_t* orow = p.out + ((size_t)b * p.N + q) * p.out_ld + h * D;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
                            *reinterpret_cast<uint2*>(orow + 8 * jj + 4 * hi) = make_uint2(pk[2 * jj], pk[2 * jj + 1]);
                    }
                    if (lane < 32) {                           // lanes 32..47 hold rows 40..43 (the denominators), 48..63 rows 44..47
#pragma unroll
                        for (int hq = 0; hq < 2; ++hq) {
                            const int qt = q0 + qb * 32 + 16 * hq + (lane & 15);
                            if (qt < p.N)
                                *reinterpret_cast<uint2*>(p.out + ((size_t)b * p.N + qt) * p.out_ld + h * D + 32 + 4 * (lane >> 4)) =
                                    make_uint2(pk[8 + 2 * hq], pk[9 + 2 * hq]);
                        }
                    }
                    }
                }
            }
        } else {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float mine = o[qb][L_DT][L_REG];             // accumulator row 40 (the softmax denominator): lanes with hi == L_HI
            const float other = __shfl_xor(mine, 32);
            const float lt = (hi == L_HI) ? mine : other;
            const float inv = ((ph == 1) ? w2 : 1.0f) / lt;
            uint32_t pk[10];
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) {                   // 5 quads of valid head-dim rows per lane: 8 jj + 4 hi + 0..3
                const int dt = jj >> 2, r0 = 4 * (jj & 3);
                float v[4];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) v[e2] = o[qb][dt][r0 + e2] * inv;
                if (ph == 1) {
                    const uint2 prev = *reinterpret_cast<const uint2*>(park + (qb * 5 + jj) / 2 * 1024 + ((qb * 5 + jj) & 1) * 8);
                    v[0] += E::lo(prev.x); v[1] += E::hi(prev.x); v[2] += E::lo(prev.y); v[3] += E::hi(prev.y);
                }
                pk[2 * jj] = E::pack2(v[0], v[1]);
                pk[2 * jj + 1] = E::pack2(v[2], v[3]);
            }
            if (ph == 0 && nph == 2) {
#pragma unroll
                for (int jj = 0; jj < 5; ++jj)
                    *reinterpret_cast<uint2*>(park + (qb * 5 + jj) / 2 * 1024 + ((qb * 5 + jj) & 1) * 8) = make_uint2(pk[2 * jj], pk[2 * jj + 1]);
            } else {
                const int q = q0 + qb * 32 + col;
                if (q < p.N && !(VAR & 4096)) {
                    bf16_t* orow = p.out + ((size_t)b * p.N + q) * p.out_ld + h * D;
#pragma unroll
                    for (int jj = 0; jj < 5; ++jj)
                        *reinterpret_cast<uint2*>(orow + 8 * jj + 4 * hi) = make_uint2(pk[2 * jj], pk[2 * jj + 1]);
                }
            }
        }
        }
    }
    if constexpr (PROJ) {
        __shared__ int s_ticket;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's write-through O stores have been acknowledged
        __syncthreads();                                        // ... everybody's in the workgroup (and the loop's LDS is dead)
        int* counter = p.proj_counters + b * (int)gridDim.x + wx;
        if (tid == 0) s_ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket == p.H - 1) {                              // the last head of this (batch, 256-row block): all 320 channels of O are in memory
            if (tid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef PROJ_T_SKIP_GEMM        // (timing probe: hand-off only)
            attn_out_proj<F16>(p, b, wx * NT, min(NT, p.N - wx * NT), smem);
#endif
        }
    }
    if ((VAR & 4096) && lane == 0) {       // (the kernel's real output is garbage in this variant: the counters overwrite its head)
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.out);
        atomicAdd(dbg + 0, tm_loop); atomicAdd(dbg + 1, tm_slots); atomicAdd(dbg + 2, tm_check); atomicAdd(dbg + 3, tm_sync);
        atomicAdd(dbg + 4, tm_steps); atomicAdd(dbg + 5, 1ull);
    }
    if constexpr (UNCHK) return __syncthreads_or((int)bad) != 0;      // (also: everybody is done with the LDS before a re-run restages it)
    return false;
}

// UNCHK kernels run WITHOUT the per-step overflow test on interior steps (the deferred maximum stays the first block's: P = 2^(s - m_ref)
// is unbounded above, which bf16's 8 exponent bits and the fp32 accumulators absorb up to 2^127).  The softmax denominators are checked
// when a phase is done; should one of them not be finite, the whole workgroup runs again with the checked step on every block and
// overwrites its output (exact, slow, never seen on real data).  The two runs share nothing but the kernel arguments, so the cold
// checked body costs the hot one no registers (a retry loop around the phase did: 244 SGPR + 63 VGPR spills, +6 % time).
template <bool F16, int THR, int VAR>
__global__ __launch_bounds__((VAR & 32768) ? 512 : 256, 2) void attn40_kernel(const AttnParams p) {
    if constexpr ((VAR & 4194304) != 0) {
        if (attn40_body<F16, THR, VAR>(p)) attn40_body<F16, THR, VAR & ~4194304>(p);
    } else {
        attn40_body<F16, THR, VAR>(p);
    }
}

template <bool F16, int THR, int VAR>
int launch_attn40(const AttnParams& p, hipStream_t s) {
    auto kern = attn40_kernel<F16, THR, VAR>;
    constexpr int NW = (VAR & 32768) ? 8 : 4;
    constexpr int PARKB = NW * ((VAR & 8192) ? PARK_T : PARK) / 4;
    constexpr int LDS_BYTES = (VAR & 512) ? 100 * 1024 : (VAR & 128) ? NRING * BUF_D + PARKB + 1024 : 2 * BUF + PARKB;
    static_assert(!(VAR & 1048576) || LDS_BYTES >= PJ_LDS, "the out-projection's chunk buffers must fit the loop's LDS");
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), LDS_BYTES, "attention(d=40)")) return rc_attr;
    dim3 grid((p.N + NW * 64 - 1) / (NW * 64), p.H, p.B);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), LDS_BYTES, s, p);
    return imd_check_launch("attention(d=40)");
}

}  // namespace

// variant: 10 (default) = software-pipelined kernel, pinned MFMA / VALU interleave, head-dim rows 32..40 of P.V on
//          v_mfma_f32_16x16x32 (VAR & 8192), K / V^T staged by LDS-DMA when the caller guarantees K's pad column
//          (imd_attn_params.k_pad_one), through registers otherwise;
//          9 = round-2 default (P.V as two 32x32x16 row blocks), same staging rule;  7 = 9 with register staging always;
//          6 = 7 with the compiler's own interleave;  8 = 7 with the deferred-maximum bound 2^12 instead of 2^8
//          20..49 (only when built with -DIMD_ABLATIONS): timing ablations with WRONG results (tools/attn_bench.py);
//          50..54 (same builds): correct variants measured no faster -- 8-wave workgroups, DMA pieces without their leading wait
//          states, LDS fragment reads 3 / 4 fragments ahead
int imd_launch_attention_d40(const AttnParams& p, int variant, hipStream_t s) {
    const bool h = p.dtype == IMD_DTYPE_F16;
    switch (variant) {
        case 6: return h ? launch_attn40<true, 8, 0>(p, s) : launch_attn40<false, 8, 0>(p, s);
        case 8: return h ? launch_attn40<true, 12, 1>(p, s) : launch_attn40<false, 12, 1>(p, s);
        case 9:             // LDS-DMA staging: needs the caller's guarantee that K's pad column holds 1.0
            if (p.k_pad_one) return h ? launch_attn40<true, 8, 1 | 128>(p, s) : launch_attn40<false, 8, 1 | 128>(p, s);
            return h ? launch_attn40<true, 8, 1>(p, s) : launch_attn40<false, 8, 1>(p, s);
        case 7: return h ? launch_attn40<true, 8, 1>(p, s) : launch_attn40<false, 8, 1>(p, s);
        case 11:            // the 16x16x32 tail with register staging always
            return h ? launch_attn40<true, 8, 1 | 8192>(p, s) : launch_attn40<false, 8, 1 | 8192>(p, s);
#ifdef IMD_ABLATIONS
        // timing ablations of variant 7 (WRONG results; tools/attn_bench.py --variants ...)
        case 20: return launch_attn40<false, 8, 1 | 4>(p, s);         // no exp2
        case 21: return launch_attn40<false, 8, 1 | 8>(p, s);         // no loop barrier
        case 22: return launch_attn40<false, 8, 1 | 16>(p, s);        // no loads / LDS stores in the loop
        case 23: return launch_attn40<false, 8, 1 | 32>(p, s);        // no P.V MFMAs
        case 24: return launch_attn40<false, 8, 1 | 64>(p, s);        // 2 of 6 QK^T MFMAs
        case 25: return launch_attn40<false, 8, 1 | 8 | 16>(p, s);    // no barrier, no staging
        case 26: return launch_attn40<false, 8, 1 | 256>(p, s);       // one LDS fragment read per step
        case 27: return launch_attn40<false, 8, 1 | 512>(p, s);       // one workgroup per CU
        case 28: return launch_attn40<false, 8, 1 | 4 | 16 | 256>(p, s);   // MFMAs + pack only
        case 29: return launch_attn40<false, 8, 1 | 32 | 64 | 16>(p, s);   // 2 MFMAs per step, everything else
        case 30: return launch_attn40<false, 8, 1 | 128 | 2048>(p, s);     // LDS-DMA variant without the packed-max / overflow test
        case 31: return launch_attn40<false, 8, 1 | 128 | 4096>(p, s);     // cycle counters (tools/attn_bench.py --cycles)
        case 32: return launch_attn40<false, 8, 1 | 128 | 4096 | 512>(p, s);   // ... with one workgroup per CU
        // one workgroup per CU (a lone wave per SIMD): what is that wave's step made of
        case 33: return launch_attn40<false, 8, 1 | 128 | 512>(p, s);
        case 34: return launch_attn40<false, 8, 1 | 128 | 512 | 4>(p, s);              // no exp2
        case 35: return launch_attn40<false, 8, 1 | 128 | 512 | 16>(p, s);             // no staging
        case 36: return launch_attn40<false, 8, 1 | 128 | 512 | 32>(p, s);             // no P.V MFMAs
        case 37: return launch_attn40<false, 8, 1 | 128 | 512 | 32 | 64>(p, s);        // 2 MFMAs per step
        case 38: return launch_attn40<false, 8, 1 | 128 | 512 | 4 | 16 | 2048>(p, s);  // MFMAs, packs and fragment reads only
        case 39: return launch_attn40<false, 8, 1 | 128 | 512 | 8>(p, s);              // no barrier
        // ablations of the default kernel (variant 10: 16x16x32 tail + LDS-DMA staging)
        case 40: return launch_attn40<false, 8, 1 | 128 | 8192 | 2048>(p, s);          // no packed-max / overflow test
        case 41: return launch_attn40<false, 8, 1 | 128 | 8192 | 8>(p, s);             // no loop barrier
        case 42: return launch_attn40<false, 8, 1 | 128 | 8192 | 16>(p, s);            // no staging in the loop
        case 43: return launch_attn40<false, 8, 1 | 128 | 8192 | 4>(p, s);             // exp2 -> move
        case 44: return launch_attn40<false, 8, 1 | 128 | 8192 | 256>(p, s);           // one LDS fragment read per step
        case 45: return launch_attn40<false, 8, 1 | 128 | 8192 | 8 | 16>(p, s);        // no barrier, no staging
        case 46: return launch_attn40<false, 8, 1 | 128 | 8192 | 8 | 16 | 2048>(p, s); // ... and no overflow test
        case 47: return launch_attn40<false, 8, 1 | 128 | 8192 | 8 | 16 | 2048 | 256>(p, s);   // ... and one fragment read
        case 48: return launch_attn40<false, 8, 1 | 128 | 8192 | 512>(p, s);           // one workgroup per CU
        // round-3 variants that compute CORRECT results and measured no faster than 10 (profiles/r3c_*, r3d_*): the kernel is power-capped
        case 50:            // 10 with 8-wave (512-query) workgroups
            if (p.k_pad_one) return h ? launch_attn40<true, 8, 1 | 128 | 8192 | 32768>(p, s) : launch_attn40<false, 8, 1 | 128 | 8192 | 32768>(p, s);
            return h ? launch_attn40<true, 8, 1 | 8192>(p, s) : launch_attn40<false, 8, 1 | 8192>(p, s);
        case 51:            // 50 without the five leading wait states of the loop's DMA pieces
            if (p.k_pad_one) return h ? launch_attn40<true, 8, 1 | 128 | 8192 | 32768 | 65536>(p, s) : launch_attn40<false, 8, 1 | 128 | 8192 | 32768 | 65536>(p, s);
            return h ? launch_attn40<true, 8, 1 | 8192>(p, s) : launch_attn40<false, 8, 1 | 8192>(p, s);
        case 52:            // 10 without them
            if (p.k_pad_one) return h ? launch_attn40<true, 8, 1 | 128 | 8192 | 65536>(p, s) : launch_attn40<false, 8, 1 | 128 | 8192 | 65536>(p, s);
            return h ? launch_attn40<true, 8, 1 | 8192>(p, s) : launch_attn40<false, 8, 1 | 8192>(p, s);
        case 53:            // 10 with fragment reads three fragments ahead
            if (p.k_pad_one) return h ? launch_attn40<true, 8, 1 | 128 | 8192 | 131072>(p, s) : launch_attn40<false, 8, 1 | 128 | 8192 | 131072>(p, s);
            return h ? launch_attn40<true, 8, 1 | 8192>(p, s) : launch_attn40<false, 8, 1 | 8192>(p, s);
        case 54:            // ... four fragments ahead
            if (p.k_pad_one) return h ? launch_attn40<true, 8, 1 | 128 | 8192 | 262144>(p, s) : launch_attn40<false, 8, 1 | 128 | 8192 | 262144>(p, s);
            return h ? launch_attn40<true, 8, 1 | 8192>(p, s) : launch_attn40<false, 8, 1 | 8192>(p, s);
#endif
        case 10:            // the round-3 default: one loop body, ring slot and staging bookkeeping at run time
            if (p.proj_w == nullptr) {
                if (p.k_pad_one) return h ? launch_attn40<true, 8, 1 | 128 | 8192>(p, s) : launch_attn40<false, 8, 1 | 128 | 8192>(p, s);
                return h ? launch_attn40<true, 8, 1 | 8192>(p, s) : launch_attn40<false, 8, 1 | 8192>(p, s);
            }
            [[fallthrough]];
        case 13:            // round 4 (default): 12 without the per-step overflow test on interior steps; fp16 (round 5) with the reference maximum biased by 2^4 (FIRST_BIAS)
            if (p.out_dup != nullptr)       // (validated by imd_launch_attention: k_pad_one, no fused out-projection)
                return h ? launch_attn40<true, 8, 1 | 128 | 8192 | 2097152 | 4194304 | 8388608>(p, s) : launch_attn40<false, 8, 1 | 128 | 8192 | 2097152 | 4194304 | 8388608>(p, s);
            if (p.proj_w == nullptr && p.k_pad_one)
                return h ? launch_attn40<true, 8, 1 | 128 | 8192 | 2097152 | 4194304>(p, s) : launch_attn40<false, 8, 1 | 128 | 8192 | 2097152 | 4194304>(p, s);
            [[fallthrough]];
        case 12:            // round 4: the LDS-DMA kernel with the main loop unrolled over the three ring slots -- compile-time LDS
        default:            // addresses, three-instruction staging pieces, no first / ragged-block test on interior steps
            if (p.out_dup != nullptr)
                return h ? launch_attn40<true, 8, 1 | 128 | 8192 | 2097152 | 8388608>(p, s) : launch_attn40<false, 8, 1 | 128 | 8192 | 2097152 | 8388608>(p, s);
            if (p.proj_w == nullptr && p.k_pad_one)
                return h ? launch_attn40<true, 8, 1 | 128 | 8192 | 2097152>(p, s) : launch_attn40<false, 8, 1 | 128 | 8192 | 2097152>(p, s);
            if (p.proj_w != nullptr) {      // fused out-projection (validated by imd_launch_attention)
                if (p.k_pad_one) return h ? launch_attn40<true, 8, 1 | 128 | 8192 | 1048576>(p, s) : launch_attn40<false, 8, 1 | 128 | 8192 | 1048576>(p, s);
                return h ? launch_attn40<true, 8, 1 | 8192 | 1048576>(p, s) : launch_attn40<false, 8, 1 | 8192 | 1048576>(p, s);
            }
            if (p.k_pad_one) return h ? launch_attn40<true, 8, 1 | 128 | 8192>(p, s) : launch_attn40<false, 8, 1 | 128 | 8192>(p, s);
            return h ? launch_attn40<true, 8, 1 | 8192>(p, s) : launch_attn40<false, 8, 1 | 8192>(p, s);
    }
}
