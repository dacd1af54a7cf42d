// Fused dual-softmax ("hybrid") attention for gfx950: bf16 MFMA, fp32 online softmax.
//
//   O[b, q, h*D:(h+1)*D] = softmax(Q K1^T) V1  +  s2[b] * softmax(Q K2^T) V2
//
// which is the core of the reference's hybrid processor
// (/root/reference/adapter/attention_processor.py:589-612, RefSAttnProcessor2_0: frozen self
// attention over the image tokens, PLUS an independently normalised cross attention over the
// garment UNet's tokens, summed before the out-projection), and -- with K2/V2 = the 4 IP-Adapter
// face tokens -- of LoRAIPAttnProcessor2_0 (:833-856).  With k2 == nullptr or s2[b] == 0 it is the
// plain attention of the uncond pass / CAttnProcessor2_0 / the Perceiver resampler
// (adapter/resampler.py:71-74; its softmax is fp32 there as it is here).
//
// Layouts (produced by the head-split epilogue of conv_gemm.hip):
//   Q   [B , H, N , DPK]   already multiplied by  D^-1/2 * log2(e)   (softmax runs on exp2)
//   K   [Bk, H, L , DPK]   rows = keys, zero padded from D to DPK (multiple of 16)
//   V^T [Bk, H, DPV, LP]   rows = head-dim, keys contiguous, LP = L rounded up to 64 (zero padded)
// The kv batch entry used by batch b is b / kv_bdiv (stride-0 style sharing: the garment K/V are
// computed ONCE per garment and shared by every image of the batch; text K/V once per prompt).
//
// Structure: one workgroup = 4 waves = 4*QW blocks of 32 query rows of one (batch, head);
// K / V^T tiles of 64 keys are staged global -> registers -> LDS (double buffered, one barrier
// per tile).  Per 32x32 block the wave computes S^T = K Q^T with the MFMA rows *permuted*
// (swap23) so that each lane ends up holding, for ITS query column, 8 consecutive keys per
// register octet: after exp2 and bf16 packing those registers ARE the B-operand of the
// O^T += V^T P^T MFMA -- no LDS round trip, no cross-lane shuffles for P.  Softmax statistics
// (running max m, running sum l) are per-lane scalars because a lane owns one query column in
// both MFMAs; the two half-waves that share a query exchange one value per tile (max) and one
// per phase (sum).
#include "common.h"
#include "imd_kernels.h"

namespace {

constexpr int KT = 64;                 // keys per tile
constexpr int VSTR = KT * 2 + 16;      // bytes per V^T LDS row (9 x 16 B: conflict-free b128 reads)

template <int D> struct AttnCfg {
    static constexpr int DPK = (D + 15) / 16 * 16;
    static constexpr int DPV = (D + 31) / 32 * 32;
    static constexpr int NKT = DPK / 16;
    static constexpr int NDT = DPV / 32;
    static constexpr int KSTR = DPK * 2 + 16;
    // V^T rows kept in LDS: the D real rows plus (when D < DPV) the all-ones row D that yields the softmax denominator.
    // The remaining pad rows of the last 32-row MFMA block are NOT stored: their fragment reads run into the K tile
    // that follows (finite garbage in, accumulator rows nobody reads out) -- 26 KB instead of 32 KB per workgroup for
    // d = 40, i.e. 6 resident workgroups per CU instead of 5.
    static constexpr int VROWS = DPV > D ? D + 1 : DPV;
    static constexpr int VBYTES = VROWS * VSTR;
    static constexpr int BUF = VBYTES + KT * KSTR;               // [V^T rows][K rows]
    static_assert((DPV - VROWS) * VSTR <= KT * KSTR, "phantom V^T rows must stay inside the K tile");
    static constexpr int KVECS = (KT * (DPK / 8) + 255) / 256;
    static constexpr int VVECS = (D * (KT / 8) + 255) / 256;
};

typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4u;
__device__ __forceinline__ uint4 buf_load16(const __amdgpu_buffer_rsrc_t& rs, uint32_t byte_off) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
constexpr uint32_t OOB = 0xffffffffu;

// packed 3-input maximum of two fp16 lanes: on non-negative 16-bit float patterns (fp16 OR bf16) it is the maximum of
// the patterns as integers, with inf / NaN patterns propagating
__device__ __forceinline__ uint32_t pk_max3_f16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <bool F16, int D, int QW, int MINW, int KB, bool SPEC, int SCHED>
__global__ __launch_bounds__(256, MINW) void attn_kernel(const AttnParams p) {
    using C = AttnCfg<D>;
    using E = El<F16>;
    // When the V^T tile has pad rows (D < DPV) row D is set to ONE: the P.V MFMA then accumulates the softmax
    // denominator sum_k P[q][k] in accumulator row D for free (and from the same rounded P as the numerator).
    constexpr bool LSUM_MFMA = C::DPV > D;
    // When Q/K rows have pad slots (D < DPK) the running-max subtraction is done BY the QK^T MFMA: K's pad
    // column D is set to 1 in LDS and Q's pad slot D carries -m_ref, so S comes out as q.k - m_ref and the
    // softmax needs no per-score subtract.  m_ref is a deferred max: it is raised (and O rescaled) only when a
    // row's tile maximum exceeds it by more than OFFS_THR (base-2 units), which keeps P <= 2^OFFS_THR.
    constexpr bool OFFS = C::DPK > D;
    constexpr float OFFS_THR = 8.0f;
    constexpr int QPAD_T = D / 16;                    // Q fragment that holds pad slot D ...
    constexpr int QPAD_HI = (D % 16) / 8;             // ... in the lanes of this half-wave ...
    static_assert(!OFFS || (D % 8) == 0, "pad slot must start a 16-bit pair");   // ... as element 0 of the fragment
    constexpr int L_DT = D / 32, L_REG = ((D % 32) & 3) + 4 * (((D % 32) >> 3)), L_HI = ((D % 32) >> 2) & 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hi = lane >> 5;
    const int col = lane & 31;
    // XCD-aware work mapping: hardware block L runs on XCD L % 8 (private 4 MiB L2 each).  Give every XCD a
    // contiguous slice of the (batch, head, q-tile) work list so that all q-tiles of one (batch, head) -- which
    // re-read the same K / V^T -- hit the same L2 instead of pulling it through all eight.
    int wx, h, b;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gridDim.x * gridDim.y * gridDim.z;
        const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const unsigned xcd = L & 7u, slot = L >> 3;
        const unsigned q8 = total >> 3, r8 = total & 7u;          // bijective also when total % 8 != 0
        unsigned w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
        if (p.flags & 1) w = L;                                    // tuning: plain dispatch order
        // work list order: q-tile fastest, then BATCH, then head -- an XCD's slice then holds whole heads across
        // all batch rows, i.e. the same mix of 2-phase (garment) and 1-phase rows as every other XCD
        const unsigned gz = gridDim.z;
        wx = (int)(w % gx);
        b = (int)((w / gx) % gz);
        h = (int)(w / (gx * gz));
    }
    // phase-split launch (imd_attn_params.phase2_rows): batch entries >= B run the SECOND softmax of row b - B only (fp32 result to phase2_out, added to the
    // first by a follow-up elementwise launch), entries < B the first one only
    const bool split = p.phase2_rows > 0;
    const bool second_only = split && b >= p.B;
    if (second_only) b -= p.B;
    const int q0 = (wx * 4 + wave) * (QW * 32);

    // V^T row D (the all-ones row) of both LDS buffers, written once and never restaged
    if (C::DPV > D) {
        constexpr int PADV = VSTR / 16;
        const uint32_t one2 = E::pack2(1.0f, 1.0f);
        for (int v = tid; v < 2 * PADV; v += 256) {
            const int bufi = v / PADV, r = v % PADV;
            *reinterpret_cast<uint4*>(smem + bufi * C::BUF + D * VSTR + r * 16) = make_uint4(one2, one2, one2, one2);
        }
    }

    // ---- Q fragments (B operand of S^T = K Q^T): lane = query column, 8 head-dim values ----
    uint4 qf[QW][C::NKT];
#pragma unroll
    for (int qb = 0; qb < QW; ++qb) {
        const int q = q0 + qb * 32 + col;
#pragma unroll
        for (int t = 0; t < C::NKT; ++t) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q < p.N) v = *reinterpret_cast<const uint4*>(p.q + ((size_t)(b * p.H + h) * p.N + q) * C::DPK + t * 16 + hi * 8);
            qf[qb][t] = v;
        }
    }

    const bool causal = p.causal != 0;
    float w2 = 0.f;
    if (p.k2 != nullptr && p.scale2 != nullptr) w2 = p.scale2[b];
    const int nph = split ? (second_only ? 2 : 1) : ((w2 != 0.f) ? 2 : 1);

    f32x16 o[QW][C::NDT];
    float m_run[QW], l_run[QW];

    for (int ph = second_only ? 1 : 0; ph < nph; ++ph) {
#pragma unroll
        for (int qb = 0; qb < QW; ++qb) {
            m_run[qb] = OFFS ? 0.f : -INFINITY; l_run[qb] = 0.f;       // OFFS: m_run is m_ref (starts at 0)
            if (OFFS && hi == QPAD_HI) qf[qb][QPAD_T].x &= 0xffff0000u;  // Q pad slot = -m_ref = 0
#pragma unroll
            for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][dt][r] = 0.f;
        }
        const int L = ph ? p.L2 : p.L1;
        const int LP = ph ? p.L2P : p.L1P;
        const int kvb = ph ? (b / p.kv2_bdiv) : (b / p.kv1_bdiv);
        const bf16_t* kbase = (ph ? p.k2 : p.k1) + (size_t)(kvb * p.H + h) * L * C::DPK;
        const bf16_t* vbase = (ph ? p.v2t : p.v1t) + (size_t)(kvb * p.H + h) * C::DPV * LP;
        const int ntiles = (L + KT - 1) / KT;
        // buffer descriptors over this (batch, head)'s K rows / V^T rows: tails read as zero, no branches
        const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kbase), 0, (uint32_t)L * C::DPK * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vbase), 0, (uint32_t)D * LP * 2, 0x00020000);

        uint4 kreg[C::KVECS], vreg[C::VVECS];
        auto load_tile = [&](int t) {
            const int kt0 = t * KT;
#pragma unroll
            for (int i = 0; i < C::KVECS; ++i) {
                const int v = tid + i * 256;                       // K tile is one contiguous span of KT rows
                const bool ok = v < KT * (C::DPK / 8);
                kreg[i] = buf_load16(rs_k, ok ? (uint32_t)(kt0 * C::DPK * 2 + v * 16) : OOB);
            }
#pragma unroll
            for (int i = 0; i < C::VVECS; ++i) {
                const int v = tid + i * 256;
                const int row = v / (KT / 8), vc = v % (KT / 8);
                const bool ok = v < D * (KT / 8);
                vreg[i] = buf_load16(rs_v, ok ? (uint32_t)((row * LP + kt0 + vc * 8) * 2) : OOB);
            }
        };
        auto store_tile = [&](int bufi) {
            char* Vs = smem + bufi * C::BUF;
            char* Ks = Vs + C::VBYTES;
#pragma unroll
            for (int i = 0; i < C::KVECS; ++i) {
                const int v = tid + i * 256;
                const int row = v / (C::DPK / 8), vc = v % (C::DPK / 8);
                if (OFFS && vc == D / 8) kreg[i].x = (kreg[i].x & 0xffff0000u) | (uint32_t)E::fromf(1.0f);   // K[:, D] = 1
                if (v < KT * (C::DPK / 8)) *reinterpret_cast<uint4*>(Ks + row * C::KSTR + vc * 16) = kreg[i];
            }
#pragma unroll
            for (int i = 0; i < C::VVECS; ++i) {
                const int v = tid + i * 256;
                const int row = v / (KT / 8), vc = v % (KT / 8);
                if (v < D * (KT / 8)) *reinterpret_cast<uint4*>(Vs + row * VSTR + vc * 16) = vreg[i];
            }
        };

        load_tile(0);
        store_tile(0);
        __syncthreads();

        const int kfrag = swap23(col) * C::KSTR + hi * 16;   // permuted key row of this lane
        const int vfrag = col * VSTR + hi * 16;
        for (int t = 0; t < ntiles; ++t) {
            // SCHED & 2: issue the next tile's loads unconditionally (past the last tile the K offsets are out of range and
            // read as 0, the V^T ones alias rows that are never stored): a conditional load makes the staging registers
            // loop-carried, and hipcc then copies them right after the loads, i.e. waits for the data at the TOP of the tile
            // instead of at the LDS writes at its end
            if ((SCHED & 2) || t + 1 < ntiles) load_tile(t + 1);
            const char* Vs = smem + (t & 1) * C::BUF;
            const char* Ks = Vs + C::VBYTES;
            const bool ragged = (t + 1) * KT > L;

            // The 64-key tile is consumed KB 32-key blocks at a time (KB = 2: one softmax update per tile, more
            // independent MFMAs in flight; KB = 1: half the live S / P registers -> one more resident wave per SIMD).
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k0 = 0; k0 < 2; k0 += KB) {
            // a block that lies entirely past the last key (77 text tokens: keys 96..127 of the second tile) contributes nothing
            if (t * KT + k0 * 32 >= L) continue;
            // ---- S^T = K Q^T: every K fragment is read from LDS once and used for all QW query blocks ----
            f32x16 s[QW][KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int tk = 0; tk < C::NKT; ++tk) {
                    const uint4 kf = *reinterpret_cast<const uint4*>(Ks + (k0 + kb) * 32 * C::KSTR + kfrag + tk * 32);
#pragma unroll
                    for (int qb = 0; qb < QW; ++qb)      // first k step takes the inline constant 0 as C: no accumulator clears
                        s[qb][kb] = E::mfma(kf, qf[qb][tk], tk == 0 ? zero16 : s[qb][kb]);
                }

            uint4 pf[QW][2 * KB];
            // With several query blocks per wave the P.V MFMAs of block qb are issued right after ITS softmax, so that
            // they execute underneath the VALU work of block qb+1 (V^T fragments are read once, up front).
            constexpr bool PVSPLIT = QW > 1 && (SCHED & 1);
            uint4 vfr[PVSPLIT ? C::NDT : 1][PVSPLIT ? 2 * KB : 1];
            if (PVSPLIT) {
#pragma unroll
                for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
                    for (int g = 0; g < 2 * KB; ++g)
                        vfr[dt][g] = *reinterpret_cast<const uint4*>(Vs + dt * 32 * VSTR + vfrag + (2 * k0 + g) * 32);
            }
            auto softmax_block = [&](int qb) {
                const bool first = (t == 0) && (k0 == 0);
                auto pack_p = [&]() {      // P^T fragments: register octet g of block kb = keys 16g+8hi..+7 of that block
#pragma unroll
                    for (int g = 0; g < 2 * KB; ++g) {
                        const int kb = g >> 1, r0 = (g & 1) * 8;
                        pf[qb][g].x = E::pack2(s[qb][kb][r0 + 0], s[qb][kb][r0 + 1]);
                        pf[qb][g].y = E::pack2(s[qb][kb][r0 + 2], s[qb][kb][r0 + 3]);
                        pf[qb][g].z = E::pack2(s[qb][kb][r0 + 4], s[qb][kb][r0 + 5]);
                        pf[qb][g].w = E::pack2(s[qb][kb][r0 + 6], s[qb][kb][r0 + 7]);
                    }
                };
                if (OFFS && SPEC) {
                    // Speculative softmax step: s already holds q.k - m_ref (deferred max, folded into the MFMA), so
                    // exponentiate and pack straight away; the packed 16-bit P patterns are non-negative, hence ordered
                    // like integers / like fp16 patterns, and ONE packed 3-input max per two registers finds the largest
                    // P of the block (inf / NaN patterns included).  Only when some P exceeds 2^OFFS_THR (or on the
                    // first / ragged block) is the block redone on the exact path below.  Saves the 32-value fp32 max
                    // tree and the cross-half-wave exchange on every other block.
                    bool redo = first || ragged || causal;
                    if (!redo) {
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) s[qb][kb][r] = __builtin_amdgcn_exp2f(s[qb][kb][r]);
                        pack_p();
                        uint32_t m = pk_max3_f16(pf[qb][0].x, pf[qb][0].y, pf[qb][0].z);
                        m = pk_max3_f16(m, pf[qb][0].w, pf[qb][1].x);
                        m = pk_max3_f16(m, pf[qb][1].y, pf[qb][1].z);
#pragma unroll
                        for (int g = 2; g < 2 * KB; g += 2) {
                            m = pk_max3_f16(m, pf[qb][g - 1].w, pf[qb][g].x);
                            m = pk_max3_f16(m, pf[qb][g].y, pf[qb][g].z);
                            m = pk_max3_f16(m, pf[qb][g].w, pf[qb][g + 1].x);
                            m = pk_max3_f16(m, pf[qb][g + 1].y, pf[qb][g + 1].z);
                        }
                        m = pk_max3_f16(m, pf[qb][2 * KB - 1].w, pf[qb][2 * KB - 1].w);
                        const uint32_t top = max(m & 0xffffu, m >> 16);
                        redo = __any(top > (uint32_t)E::fromf(256.0f));          // 2^OFFS_THR
                        if (redo) {          // rare: recompute this block's scores (K fragments are still in LDS)
#pragma unroll
                            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                                for (int tk = 0; tk < C::NKT; ++tk) {
                                    const uint4 kf = *reinterpret_cast<const uint4*>(Ks + (k0 + kb) * 32 * C::KSTR + kfrag + tk * 32);
                                    s[qb][kb] = E::mfma(kf, qf[qb][tk], tk == 0 ? zero16 : s[qb][kb]);
                                }
                        }
                    }
                    if (!redo) return;
                }
                if (ragged || causal) {     // keys >= L of the last tile (and, causal: keys after the query) contribute nothing
                    const int qidx = causal ? q0 + qb * 32 + col : 0x7fffffff;
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = t * KT + (k0 + kb) * 32 + 8 * hi + (r & 7) + 16 * (r >> 3);
                            if (key >= L || key > qidx) s[qb][kb][r] = -INFINITY;
                        }
                }
                // ---- online softmax (base 2; Q carries the scale) ----
                float mx = s[qb][0][0];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][kb][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (OFFS) {
                    // s already holds q.k - m_ref.  Raise m_ref only on the first block or past the threshold.
                    if (first || __any(mx > OFFS_THR)) {
                        const float want = m_run[qb] + (first ? mx : fmaxf(mx, 0.f));
                        const float nref = E::tof(E::fromf(want));          // what the 16-bit Q slot can carry
                        const float delta = nref - m_run[qb];
                        if (!first) {       // on the first block O is still 0 (and delta may be hugely negative: 2^-delta = inf)
                            const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                            for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
                                for (int r = 0; r < 16; ++r) o[qb][dt][r] *= alpha;
                        }
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) s[qb][kb][r] -= delta;
                        m_run[qb] = nref;
                        if (hi == QPAD_HI) qf[qb][QPAD_T].x = (qf[qb][QPAD_T].x & 0xffff0000u) | (uint32_t)E::fromf(-nref);
                    }
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[qb][kb][r] = __builtin_amdgcn_exp2f(s[qb][kb][r]);
                } else {
                    const float m_new = fmaxf(m_run[qb], mx);
                    // rescale only when some row's running max actually moved (exact: alpha == 1 otherwise)
                    if (__any(m_new != m_run[qb])) {
                        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
                        if (!LSUM_MFMA) l_run[qb] *= alpha;
#pragma unroll
                        for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) o[qb][dt][r] *= alpha;
                        m_run[qb] = m_new;
                    }
                    float psum = 0.f;
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float pe = __builtin_amdgcn_exp2f(s[qb][kb][r] - m_new);
                            s[qb][kb][r] = pe;
                            if (!LSUM_MFMA) psum += pe;
                        }
                    if (!LSUM_MFMA) l_run[qb] += psum;
                }
                pack_p();
            };
#pragma unroll
            for (int qb = 0; qb < QW; ++qb) {
                softmax_block(qb);
                if (PVSPLIT) {
#pragma unroll
                    for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
                        for (int g = 0; g < 2 * KB; ++g) o[qb][dt] = E::mfma(vfr[dt][g], pf[qb][g], o[qb][dt]);
                        }
            }
            // ---- O^T += V^T P^T: every V^T fragment is read once and used for all QW query blocks ----
            if (!PVSPLIT) {
#pragma unroll
                for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
                    for (int g = 0; g < 2 * KB; ++g) {
                        const uint4 vf = *reinterpret_cast<const uint4*>(Vs + dt * 32 * VSTR + vfrag + (2 * k0 + g) * 32);
#pragma unroll
                        for (int qb = 0; qb < QW; ++qb) o[qb][dt] = E::mfma(vf, pf[qb][g], o[qb][dt]);
                    }
            }
            }
            if (t + 1 < ntiles) store_tile((t + 1) & 1);
            __syncthreads();
        }

        // ---- end of phase: normalise and combine.  out = O1/l1 (phase 0) [+ w2 * O2/l2 (phase 1)]; the phase-0
        // result is parked in the output buffer (rounded to the 16-bit element type, like the reference's two
        // half-precision SDPA outputs that are added at attention_processor.py:612) and re-read by the SAME lanes.
#pragma unroll
        for (int qb = 0; qb < QW; ++qb) {
            float lt;
            if (LSUM_MFMA) {
                const float mine = o[qb][L_DT][L_REG];            // accumulator row D lives in lanes with hi == L_HI
                const float other = __shfl_xor(mine, 32);
                lt = (hi == L_HI) ? mine : other;
            } else {
                lt = l_run[qb] + __shfl_xor(l_run[qb], 32);
            }
            const float inv = ((ph == 1) ? w2 : 1.0f) / lt;
            const int q = q0 + qb * 32 + col;
            if (q >= p.N) continue;
            bf16_t* orow = p.out + ((size_t)b * p.N + q) * p.out_ld + h * D;
#pragma unroll
            for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int dd = dt * 32 + 8 * j + 4 * hi;
                    if (dd >= D) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = o[qb][dt][4 * j + e] * inv;
                    if (second_only) {          // w2 * O2 / l2 in fp32: the follow-up launch adds it to the stored first phase
                        *reinterpret_cast<float4*>(p.phase2_out + ((size_t)b * p.N + q) * (p.H * D) + h * D + dd) = make_float4(v[0], v[1], v[2], v[3]);
                        continue;
                    }
                    if (ph == 1) {
                        const uint2 prev = *reinterpret_cast<const uint2*>(orow + dd);
                        v[0] += E::lo(prev.x); v[1] += E::hi(prev.x); v[2] += E::lo(prev.y); v[3] += E::hi(prev.y);
                    }
                    *reinterpret_cast<uint2*>(orow + dd) = make_uint2(E::pack2(v[0], v[1]), E::pack2(v[2], v[3]));
                }
        }
    }
}

// out[r, c] = round16(out[r, c] + add[r, c]) over `rows` rows of C channels (4 per thread): the phase-split launch's addition of the second softmax
template <bool F16>
__global__ __launch_bounds__(256) void attn_phase_add_kernel(bf16_t* out, int out_ld, const float* __restrict__ add, int C, long rows) {
    using E = El<F16>;
    const int vpr = C / 4;
    const long total = rows * vpr;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const long r = i / vpr;
        const int c = (int)(i - r * vpr) * 4;
        const float4 a = *reinterpret_cast<const float4*>(add + r * C + c);
        uint2* dst = reinterpret_cast<uint2*>(out + r * out_ld + c);
        const uint2 prev = *dst;
        *dst = make_uint2(E::pack2(a.x + E::lo(prev.x), a.y + E::hi(prev.x)), E::pack2(a.z + E::lo(prev.y), a.w + E::hi(prev.y)));
    }
}

template <bool F16, int D, int QW, int MINW, int KB = 2, bool SPEC = false, int SCHED = 0>
int launch_attn(const AttnParams& p, hipStream_t s) {
    using C = AttnCfg<D>;
    constexpr int lds = 2 * C::BUF;
    auto kern = attn_kernel<F16, D, QW, MINW, KB, SPEC, SCHED>;
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), lds, "attention")) return rc_attr;
    const int rows = 4 * QW * 32;
    dim3 grid((p.N + rows - 1) / rows, p.H, p.B + (p.phase2_rows > 0 ? p.phase2_rows : 0));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    int rc = imd_check_launch("attention");
    if (rc || p.phase2_rows <= 0) return rc;
    // the second launch of the phase-split form: out[r] = round(out[r] + phase2_out[r]) for the rows with a second softmax
    const long vecs = (long)p.phase2_rows * p.N * (p.H * D / 4);
    const int blocks = (int)((vecs + 255) / 256 < 4096 ? (vecs + 255) / 256 : 4096);
    hipLaunchKernelGGL(attn_phase_add_kernel<F16>, dim3(blocks), dim3(256), 0, s, p.out, p.out_ld, p.phase2_out, p.H * D, (long)p.phase2_rows * p.N);
    return imd_check_launch("attention (phase add)");
}

}  // namespace

// head-dim-40 kernel variant (tuning knob 0, imd_set_tuning(0, v)); all variants give the same result up to fp32 order:
//   13 (default): 12 with the overflow test of the deferred maximum only on the first and last steps of a phase and a check of the
//      softmax denominators when the phase is done (a workgroup that finds one not finite runs again as variant 12); fp16
//      operands (round 5) with the phase's reference maximum biased by 2^4 so that P = inf needs a score 20 base-2 units above the first
//      block's maximum; the fused out-projection runs 12;  12: 10 with the main loop unrolled over the three ring slots
//      (compile-time LDS addresses, three-instruction staging pieces) and the order inside every MFMA slot pinned;
//   10 (the round-3 default): attention_d40.hip, software-pipelined steps with the MFMA / VALU interleave written out (N >= 512; shorter
//      sequences and the causal mask fall through to variant 4 below), head-dim rows 32..40 of P.V on v_mfma_f32_16x16x32,
//      K / V^T staged by LDS-DMA when the caller guarantees K's pad column (imd_attn_params.k_pad_one), through registers
//      otherwise;  11: as 10, always through registers;  9 / 7: the round-2 kernel (P.V as two 32x32x16 row blocks) with the
//      same two staging rules;  6: as 7 without the pinned interleave;  8: as 7 with the deferred-maximum bound 2^12;
//      20..39: timing ablations, -DIMD_ABLATIONS builds only
//   5: as 2 with the next tile's loads issued unconditionally (+7 % over 2)
//   2: 2 query blocks per wave, 32-key softmax blocks, speculative exp (round-1 default)
//   4: 1 query block per wave, 32-key blocks, speculative exp      3: 1 query block, 64-key blocks, exact max every block
//   1: as 3 with speculative exp
int g_attn_qw40 = 13;

int imd_attn_dpk(int D) { return (D + 15) / 16 * 16; }
int imd_attn_dpv(int D) { return (D + 31) / 32 * 32; }

int g_attn_xcd = 1;
#ifdef IMD_ATTN_SWEEP
int g_attn_v80 = 0, g_attn_v160 = 0;      // (sweep builds: template parameters of the generic kernel at head dims 80 / 160, imd_set_tuning(3 / 4, .))
#endif

int imd_attention_dup_supported(int H, int N, int D) { return (D == 40 && N >= 512 && H > 0) ? 1 : 0; }
int imd_attention_phase_split_supported(int D) { return (D == 64 || D == 80 || D == 160) ? 1 : 0; }

int imd_launch_attention(const AttnParams& p_in, hipStream_t s) {
    AttnParams p = p_in;
    // tuning of this call: the caller's (IMD_TUNING_PER_CALL in flags on entry: bits 0..7 head-dim-40 variant, bit 8 plain work order) or the
    // process-wide knobs 0 / 1
    const unsigned tag = (unsigned)p_in.flags & IMD_TUNING_TAG_MASK;
    if (tag != 0 && tag != (unsigned)IMD_TUNING_PER_CALL)
        return imd_set_error("attention: flags = 0x%x on entry is neither 0 nor IMD_TUNING_PER_CALL | bits (an uninitialised parameter block?)", (unsigned)p_in.flags);
    const bool per_call = tag == (unsigned)IMD_TUNING_PER_CALL;
    const int qw40 = (per_call && (p_in.flags & 255)) ? (p_in.flags & 255) : g_attn_qw40;
#ifdef IMD_ABLATIONS
    constexpr int QW40_MAX = 54;
#else
    constexpr int QW40_MAX = 13;            // (the range imd_set_tuning(0, .) accepts in this build)
#endif
    if (qw40 < 1 || qw40 > QW40_MAX) return imd_set_error("attention: per-call head-dim-40 variant %d out of range 1..%d", qw40, QW40_MAX);
    p.flags = (per_call ? !((p_in.flags >> 8) & 1) : g_attn_xcd != 0) ? 0 : 1;
    if (p.B <= 0 || p.H <= 0 || p.N <= 0 || p.L1 <= 0) return imd_set_error("attention: empty problem B=%d H=%d N=%d L1=%d", p.B, p.H, p.N, p.L1);
    if (p.L1P % 64 || p.L1P < p.L1) return imd_set_error("attention: L1P (%d) must be a multiple of 64 and >= L1 (%d)", p.L1P, p.L1);
    if (p.k2 && (p.L2 <= 0 || p.L2P % 64 || p.L2P < p.L2)) return imd_set_error("attention: bad second key set L2=%d L2P=%d", p.L2, p.L2P);
    if (p.kv1_bdiv <= 0 || (p.k2 && p.kv2_bdiv <= 0)) return imd_set_error("attention: kv batch divisors must be positive");
    if (p.H > 65535 || p.B > 65535) return imd_set_error("attention: H/B exceed grid limits");
    if (p.dtype != IMD_DTYPE_BF16 && p.dtype != IMD_DTYPE_F16) return imd_set_error("attention: unknown dtype %d", p.dtype);
    if (p.causal && (p.k2 != nullptr || p.L1 != p.N)) return imd_set_error("attention: the causal mask needs a single key set with L1 == N");
    const bool h = p.dtype == IMD_DTYPE_F16;
    if ((p.phase2_rows != 0) != (p.phase2_out != nullptr)) return imd_set_error("attention: phase2_rows and phase2_out go together");
    if (p.phase2_rows != 0) {           // phase-split launch (ABI v9): the generic kernel only
        if (!imd_attention_phase_split_supported(p.D) || p.k2 == nullptr || p.scale2 == nullptr || p.causal || p.proj_w != nullptr || p.out_dup != nullptr ||
            p.phase2_rows < 0 || p.phase2_rows > p.B || (p.out_ld % 4) || p.B + p.phase2_rows > 65535)
            return imd_set_error("attention: phase2_rows needs head dim 64 / 80 / 160, a second key set with scale2, 0 < rows <= B, no causal mask / fused out-projection / "
                                 "out_dup (got D=%d rows=%d B=%d)", p.D, p.phase2_rows, p.B);
        if ((reinterpret_cast<uintptr_t>(p.phase2_out) & 15) || (reinterpret_cast<uintptr_t>(p.out) & 7))
            return imd_set_error("attention: phase2_out must be 16-byte aligned (out 8-byte)");
    }
    if (p.out_dup != nullptr) {         // duplicated first-phase output (ABI v9): the static-ring d = 40 kernel only
        if (!imd_attention_dup_supported(p.H, p.N, p.D) || !p.k_pad_one || p.causal || p.proj_w != nullptr)
            return imd_set_error("attention: out_dup needs head dim 40, N >= 512, k_pad_one, no causal mask and no fused out-projection (got D=%d N=%d k_pad_one=%d)", p.D, p.N, p.k_pad_one);
        if (p.out_dup == p.out) return imd_set_error("attention: out_dup must not alias out");
        return imd_launch_attention_d40(p, qw40 == 12 ? 12 : 13, s);      // (every other variant is an A/B form of these two)
    }
    if (p.proj_w != nullptr) {          // fused out-projection: the d = 40 kernel (variant 10) only
        if (p.D != 40 || p.H * p.D != 320 || p.N < 512 || p.causal)
            return imd_set_error("attention: the fused out-projection needs head dim 40, 8 heads, N >= 512, no causal mask (got D=%d H=%d N=%d)", p.D, p.H, p.N);
        if (!p.proj_out || !p.proj_counters) return imd_set_error("attention: fused out-projection without proj_out / proj_counters");
        if ((p.out_ld % 8) || (p.proj_out_ld % 8) || (p.proj_res && (p.proj_res_ld % 8)))
            return imd_set_error("attention: fused out-projection needs out_ld, proj_out_ld and proj_res_ld %% 8 == 0");
        if ((reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.proj_w) | reinterpret_cast<uintptr_t>(p.proj_out) |
             reinterpret_cast<uintptr_t>(p.proj_res) | reinterpret_cast<uintptr_t>(p.proj_b)) & 15)
            return imd_set_error("attention: fused out-projection needs 16-byte aligned out / proj_w / proj_out / proj_res / proj_b");
        if ((size_t)p.B * p.N * p.out_ld * 2 >= 0x80000000ull) return imd_set_error("attention: fused out-projection: out buffer beyond 2 GiB");
        return imd_launch_attention_d40(p, 10, s);
    }
    switch (p.D) {
        case 40:
            if (qw40 >= 6 && p.N >= 512 && !p.causal) return imd_launch_attention_d40(p, qw40, s);
            if (qw40 == 2 && p.N >= 512) return h ? launch_attn<true, 40, 2, 2, 1, true, 1>(p, s) : launch_attn<false, 40, 2, 2, 1, true, 1>(p, s);
            if (qw40 == 5 && p.N >= 512) return h ? launch_attn<true, 40, 2, 2, 1, true, 3>(p, s) : launch_attn<false, 40, 2, 2, 1, true, 3>(p, s);
            if (qw40 == 3) return h ? launch_attn<true, 40, 1, 3, 2, false>(p, s) : launch_attn<false, 40, 1, 3, 2, false>(p, s);
            if (qw40 == 4) return h ? launch_attn<true, 40, 1, 3, 1, true>(p, s) : launch_attn<false, 40, 1, 3, 1, true>(p, s);
            if (qw40 == 1 && p.N >= 512) return h ? launch_attn<true, 40, 2, 2, 1, true, 0>(p, s) : launch_attn<false, 40, 2, 2, 1, true, 0>(p, s);
            return h ? launch_attn<true, 40, 1, 3, 1, true>(p, s) : launch_attn<false, 40, 1, 3, 1, true>(p, s);
        case 64: return h ? launch_attn<true, 64, 1, 2>(p, s) : launch_attn<false, 64, 1, 2>(p, s);
        case 80:
#ifdef IMD_ATTN_SWEEP
            switch (g_attn_v80) {
                case 1: return h ? launch_attn<true, 80, 1, 2, 2, false, 0>(p, s) : launch_attn<false, 80, 1, 2, 2, false, 0>(p, s);      // (the default before round 6)
                case 2: return h ? launch_attn<true, 80, 1, 3, 1, false, 0>(p, s) : launch_attn<false, 80, 1, 3, 1, false, 0>(p, s);
                case 3: return h ? launch_attn<true, 80, 1, 3, 1, false, 2>(p, s) : launch_attn<false, 80, 1, 3, 1, false, 2>(p, s);
                case 4: return h ? launch_attn<true, 80, 2, 1, 2, false, 0>(p, s) : launch_attn<false, 80, 2, 1, 2, false, 0>(p, s);
                case 5: return h ? launch_attn<true, 80, 2, 1, 2, false, 3>(p, s) : launch_attn<false, 80, 2, 1, 2, false, 3>(p, s);
                case 6: return h ? launch_attn<true, 80, 2, 1, 1, false, 3>(p, s) : launch_attn<false, 80, 2, 1, 1, false, 3>(p, s);
                case 7: return h ? launch_attn<true, 80, 1, 2, 1, false, 2>(p, s) : launch_attn<false, 80, 1, 2, 1, false, 2>(p, s);
                default: break;
            }
#endif
            // (round 6: the next tile's loads issued unconditionally, SCHED = 2 -- 53.6 -> 51.7 us at the 32x32 level, 29.0 -> 27.5 us at the 16x16 level,
            // tools/attn_generic_sweep.py, profiles/r6ab_*; two query blocks per wave lose 60 % here)
            return h ? launch_attn<true, 80, 1, 2, 2, false, 2>(p, s) : launch_attn<false, 80, 1, 2, 2, false, 2>(p, s);
        case 160:
#ifdef IMD_ATTN_SWEEP
            switch (g_attn_v160) {
                case 1: return h ? launch_attn<true, 160, 1, 1, 2, false, 0>(p, s) : launch_attn<false, 160, 1, 1, 2, false, 0>(p, s);      // (the default before round 6)
                case 2: return h ? launch_attn<true, 160, 1, 1, 1, false, 2>(p, s) : launch_attn<false, 160, 1, 1, 1, false, 2>(p, s);
                case 3: return h ? launch_attn<true, 160, 1, 2, 1, false, 0>(p, s) : launch_attn<false, 160, 1, 2, 1, false, 0>(p, s);
                default: break;
            }
#endif
            return h ? launch_attn<true, 160, 1, 1, 2, false, 2>(p, s) : launch_attn<false, 160, 1, 1, 2, false, 2>(p, s);
        default: return imd_set_error("attention: unsupported head dim %d (supported: 40, 64, 80, 160)", p.D);
    }
}
