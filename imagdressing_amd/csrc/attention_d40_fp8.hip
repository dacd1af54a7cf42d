// Hybrid attention for head dim 40 on the MX block-scaled FP8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3 x e4m3, fp32
// accumulate) -- BASELINE.json configs[4]: "ControlNet-inpainting path, 768x576 ... with fp8 MFMA attention".
//
//   O[b, q, h*40:(h+1)*40] = softmax(Q K1^T) V1  +  s2[b] * softmax(Q K2^T) V2        (attention_processor.py:589-612)
//
// Why the block-scaled form: plain fp8 MFMA (32x32x16) issues at the bf16 rate on gfx950; the scaled K = 64 instruction
// does 4x the contraction in 2x the cycles, and one instruction covers the whole padded head dim of QK^T (40 -> 64), so a
// 32 x 32 score block costs ONE 64-cycle MFMA instead of three 32-cycle ones, and P.V over 64 keys costs two instead of
// eight (matrix-pipe time per key x query: 0.57 of the bf16 kernel's).  The E8M0 block scales are used as plain
// power-of-two operand scales (every 32-element block of an operand carries the same exponent), chosen by the quantiser.
//
// Operands (produced by imd_attn_quantize_fp8 from the 16-bit head-split layouts):
//   Q8   [B , H, N, 64]  e4m3: q * 2^eq in columns 0..39 (q already carries d^-1/2 log2 e), zeros elsewhere
//   K8   [Bk, H, L, 64]  e4m3: k * 2^ek in columns 0..39, 2^(eq+ek) in columns 40 and 41 (the two slots through which the
//                        deferred row maximum enters the MFMA: Q's slots carry -m_ref as a two-term fp8 sum hi + lo)
//   V8^T [Bk, H, 64, LP] e4m3: v * 2^ev, rows = head dim (40 used), keys contiguous but PERMUTED inside every group of 64
//                        (key k sits at position 32 ((k >> 3) & 1) + 8 ((k >> 4) & 3) + (k & 7)): that is the order in
//                        which a lane's exponentiated scores come out of the swapped S^T = K Q^T MFMA, so the packed P
//                        bytes ARE the B operand of O^T += V^T P^T, as in the 16-bit kernels.
// Operand lane map of the 32x32x64 instruction (measured, tools/probes/mx_layout_probe.hip): lane l holds row / column
// l & 31 and contraction slots 32 (l >> 5) .. +31 in byte order; the scale operand's low byte b multiplies by 2^(b - 127).
//
// Softmax: exact row maximum per 32-key block (8 v_max3), P = exp2(S) with S = q.k - m_ref straight out of the MFMA;
// m_ref is kept 8 below the running maximum so that the largest P is ~2^8 and e4m3's range (2^-9 .. 448) reaches 17
// binades below it; it is raised (and O rescaled) only when a score passes m_ref + 8.25.  The softmax denominator comes out
// of the P.V MFMA through an all-2^ev V^T row (same rounded P as the numerator, the 2^ev cancels).
//
// Schedule: 32-key steps, software-pipelined in the source like attention_d40.hip -- step j issues QK^T of block j + 1 and,
// on even steps, P.V of the previous 64-key group, next to the softmax of block j; K / V^T units of 64 keys are staged
// through registers into a double-buffered LDS tile (80-byte rows: conflict-free 2 x b128 fragment reads), one barrier per
// unit; the phase-0 result of a two-phase row is parked in LDS.
#include "common.h"
#include "imd_kernels.h"

namespace {

constexpr int D = 40;
constexpr int RS = 80;                       // LDS row stride (64 data bytes + 16: odd number of 16-byte slots)
constexpr int VROWS = D + 1;                 // 40 head-dim rows + the all-2^ev row (softmax denominator)
constexpr int VBYTES = VROWS * RS;           // 3280 (rows 41..63 of the second MFMA row block read into the K rows behind)
constexpr int KBYTES = 64 * RS;              // 5120
constexpr int BUF = VBYTES + KBYTES;         // 8400
constexpr int PARK = 4 * 5 * 1024;
constexpr int LDS_BYTES = 2 * BUF + PARK;
static_assert((64 - VROWS) * RS <= KBYTES, "phantom V^T rows must stay inside the K rows");

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4u;
__device__ __forceinline__ uint4 buf_load16(const __amdgpu_buffer_rsrc_t& rs, uint32_t byte_off) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
constexpr uint32_t OOB = 0xffffffffu;

__device__ __forceinline__ v8i_t frag32(const char* p) {          // 32 bytes of one LDS row: two conflict-free b128 reads
    const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 16);
    v8i_t r;
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    return r;
}
__device__ __forceinline__ float fp8_to_f32(uint32_t byte) { return __builtin_amdgcn_cvt_f32_fp8((int)byte, 0); }
__device__ __forceinline__ uint32_t f32_to_fp8(float v) { return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xffu; }
__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}

template <bool F16>
__global__ __launch_bounds__(256, 2) void attn40_fp8_kernel(const AttnParams p, const int eq, const int ek, const int ev) {
    using E = El<F16>;
    constexpr float TARGET = 8.0f;            // the running maximum maps to P = 2^8 ...
    constexpr float THR = 8.25f;              // ... and m_ref moves when a score passes 2^8.25 (e4m3 tops out at 448 = 2^8.8)
    constexpr int L_REG = 4;                  // accumulator row 40: second row block, register 4, lanes with hi == 0
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, col = lane & 31;
    int wx, h, b;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, total = gx * gy * gz;
        const unsigned Lb = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const unsigned xcd = Lb & 7u, slot = Lb >> 3, q8 = total >> 3, r8 = total & 7u;
        unsigned w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
        if (p.flags & 1) w = Lb;
        wx = (int)(w % gx);
        b = (int)((w / gx) % gz);
        h = (int)(w / (gx * gz));
    }
    const int q0 = (wx * 4 + wave) * 64;
    const int sa = 127 - ek, sb = 127 - eq;                       // E8M0 bytes: A = K (2^-ek), B = Q (2^-eq)

    {   // V^T row 40 = 2^ev in both buffers, written once
        const uint32_t one = f32_to_fp8(__builtin_exp2f((float)ev)) * 0x01010101u;
        for (int v = tid; v < 2 * (RS / 16); v += 256)
            *reinterpret_cast<uint4*>(smem + (v / (RS / 16)) * BUF + D * RS + (v % (RS / 16)) * 16) = make_uint4(one, one, one, one);
    }
    const unsigned char* q8p = reinterpret_cast<const unsigned char*>(p.q);
    v8i_t qf[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + qb * 32 + col;
        uint4 a = make_uint4(0, 0, 0, 0), c = a;
        if (q < p.N) {
            const unsigned char* src = q8p + ((size_t)(b * p.H + h) * p.N + q) * 64 + hi * 32;
            a = *reinterpret_cast<const uint4*>(src);
            c = *reinterpret_cast<const uint4*>(src + 16);
        }
        qf[qb][0] = a.x; qf[qb][1] = a.y; qf[qb][2] = a.z; qf[qb][3] = a.w; qf[qb][4] = c.x; qf[qb][5] = c.y; qf[qb][6] = c.z; qf[qb][7] = c.w;
    }
    float w2 = 0.f;
    if (p.k2 != nullptr && p.scale2 != nullptr) w2 = p.scale2[b];
    const int nph = (w2 != 0.f) ? 2 : 1;
    const int kfrag = swap23(col) * RS + hi * 32, vfrag = col * RS + hi * 32;
    char* park = smem + 2 * BUF + wave * 5120 + lane * 16;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int ph = 0; ph < nph; ++ph) {
        f32x16 o[2][2];
        float m_ref[2] = {0.f, 0.f};
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (hi == 1) qf[qb][2] &= (int)0xffff0000u;           // Q slots 40, 41 (bytes 8, 9 of the upper half) = -m_ref = 0
            o[qb][0] = zero16; o[qb][1] = zero16;
        }
        const int L = ph ? p.L2 : p.L1, LP = ph ? p.L2P : p.L1P;
        const int kvb = ph ? (b / p.kv2_bdiv) : (b / p.kv1_bdiv);
        const unsigned char* kbase = reinterpret_cast<const unsigned char*>(ph ? p.k2 : p.k1) + (size_t)(kvb * p.H + h) * L * 64;
        const unsigned char* vbase = reinterpret_cast<const unsigned char*>(ph ? p.v2t : p.v1t) + (size_t)(kvb * p.H + h) * 64 * LP;
        const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(kbase), 0, (uint32_t)L * 64, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(vbase), 0, (uint32_t)D * LP, 0x00020000);
        const int J = (L + 31) >> 5;                  // 32-key blocks
        const int NU = ((J - 1) >> 1) + 2;            // units: P.V of group g runs in unit g + 1

        // staging: unit u = K rows [64u+32, 64u+96) + V^T columns [64(u-1), 64u) (positions inside a 64-group are already permuted)
        const int kr = tid >> 2, kc = tid & 3, vr = tid >> 2, vc = tid & 3;
        const bool vok = tid < D * 4;
        uint4 kreg, vreg;
        auto load_unit = [&](int u) {
            const int krow = u * 64 + 32 + kr;
            kreg = buf_load16(rs_k, krow >= 0 ? (uint32_t)(krow * 64 + kc * 16) : OOB);
            const int c = (u - 1) * 64 + vc * 16;
            vreg = buf_load16(rs_v, (vok && c >= 0 && c < LP) ? (uint32_t)(vr * LP + c) : OOB);
        };
        auto store_unit = [&](int bufi) {
            char* Vs = smem + bufi * BUF;
            *reinterpret_cast<uint4*>(Vs + VBYTES + kr * RS + kc * 16) = kreg;
            if (vok) *reinterpret_cast<uint4*>(Vs + vr * RS + vc * 16) = vreg;
        };

        f32x16 sa_[2], sb_[2];
        uint32_t pcur[2][8], pprev[2][8];             // packed P of the current / previous 64-key group: 32 bytes per query block
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int i = 0; i < 8; ++i) { pcur[qb][i] = 0u; pprev[qb][i] = 0u; }

        // ---- one 32-key step: S_n = QK^T(block j + 1); even steps: O += V^T P^T (previous 64-key group); softmax of block j ----
        auto step = [&](const char* Vs, int half, int j, f32x16 (&sc)[2], f32x16 (&sn)[2]) {
            const char* Ks = Vs + VBYTES;
            {
                const v8i_t kf = frag32(Ks + half * 32 * RS + kfrag);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    sn[qb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf[qb], zero16, 0, 0, 0, sa, 0, sb);
            }
            if (half == 0) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const v8i_t vf = frag32(Vs + dt * 32 * RS + vfrag);
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        v8i_t pb;
#pragma unroll
                        for (int i = 0; i < 8; ++i) pb[i] = (int)pprev[qb][i];
                        o[qb][dt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pb, o[qb][dt], 0, 0, 0, 127, 0, 127);
                    }
                }
            }
            const bool ragged = j * 32 + 32 > L;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if (ragged) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = j * 32 + 8 * hi + (r & 7) + 16 * (r >> 3);
                        if (key >= L) sc[qb][r] = -INFINITY;
                    }
                }
                float m0 = fmaxf(fmaxf(sc[qb][0], sc[qb][1]), sc[qb][2]);
                float m1 = fmaxf(fmaxf(sc[qb][3], sc[qb][4]), sc[qb][5]);
                m0 = fmaxf(fmaxf(m0, sc[qb][6]), sc[qb][7]);
                m1 = fmaxf(fmaxf(m1, sc[qb][8]), sc[qb][9]);
                m0 = fmaxf(fmaxf(m0, sc[qb][10]), sc[qb][11]);
                m1 = fmaxf(fmaxf(m1, sc[qb][12]), sc[qb][13]);
                float mx = fmaxf(fmaxf(m0, m1), fmaxf(sc[qb][14], sc[qb][15]));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const bool first = (j == 0);
                if (__builtin_expect(first || __any(mx > THR), 0)) {
                    // move m_ref so that this block's maximum maps to 2^TARGET; carried by Q slots 40 / 41 as hi + lo (two e4m3 terms)
                    const float want = m_ref[qb] + (first ? mx : fmaxf(mx, TARGET)) - TARGET;
                    const uint32_t bh = f32_to_fp8(-want);
                    const float fh = fp8_to_f32(bh);
                    const uint32_t bl = f32_to_fp8(-want - fh);
                    const float nref = -(fh + fp8_to_f32(bl));
                    const float delta = nref - m_ref[qb];
                    if (!first) {
                        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) o[qb][dt][r] *= alpha;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) { sc[qb][r] -= delta; sn[qb][r] -= delta; }
                    m_ref[qb] = nref;
                    if (hi == 1) qf[qb][2] = (int)(((uint32_t)qf[qb][2] & 0xffff0000u) | bh | (bl << 8));
                    if (!first && half == 1) {        // the first half of the current group was packed at the old scale
                        // (rare) rescale it through fp32; its bytes are this lane's own
                        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t w = pcur[qb][i];
                            pcur[qb][i] = pack4_fp8(fp8_to_f32(w & 0xff) * alpha, fp8_to_f32((w >> 8) & 0xff) * alpha,
                                                    fp8_to_f32((w >> 16) & 0xff) * alpha, fp8_to_f32(w >> 24) * alpha);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[qb][r] = __builtin_amdgcn_exp2f(sc[qb][r]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    pcur[qb][half * 4 + i] = pack4_fp8(sc[qb][4 * i], sc[qb][4 * i + 1], sc[qb][4 * i + 2], sc[qb][4 * i + 3]);
            }
            if (half == 1) {                          // the group is complete: it becomes "previous" for the next unit's P.V
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int i = 0; i < 8; ++i) pprev[qb][i] = pcur[qb][i];
            }
        };

        load_unit(-1);
        store_unit(1);
        load_unit(0);
        __syncthreads();
        {
            const v8i_t kf = frag32(smem + BUF + VBYTES + 32 * RS + kfrag);          // K block 0 = second half of unit -1
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
                sa_[qb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf[qb], zero16, 0, 0, 0, sa, 0, sb);
        }
        store_unit(0);
        __syncthreads();
        for (int u = 0; u < NU; ++u) {
            load_unit(u + 1);
            const char* Vs = smem + (u & 1) * BUF;
            step(Vs, 0, 2 * u, sa_, sb_);
            step(Vs, 1, 2 * u + 1, sb_, sa_);
            store_unit((u + 1) & 1);
            __syncthreads();
        }

#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float mine = o[qb][1][L_REG];
            const float other = __shfl_xor(mine, 32);
            const float lt = (hi == 0) ? mine : other;
            const float inv = ((ph == 1) ? w2 : 1.0f) / lt;
            uint32_t pk[10];
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) {
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
                if (q < p.N) {
                    bf16_t* orow = p.out + ((size_t)b * p.N + q) * p.out_ld + h * D;
#pragma unroll
                    for (int jj = 0; jj < 5; ++jj)
                        *reinterpret_cast<uint2*>(orow + 8 * jj + 4 * hi) = make_uint2(pk[2 * jj], pk[2 * jj + 1]);
                }
            }
        }
    }
}

// ---- quantisers: 16-bit head-split operands -> e4m3 operands of the kernel above -------------------------------------------
// kind 0: rows [rows, 48] -> [rows, 64]: columns 0..39 scaled by 2^e, columns 40 and 41 = pad_val, the rest 0
template <bool F16>
__global__ void quant_rows_kernel(const bf16_t* src, unsigned char* dst, long rows, float scale, float pad_val) {
    using E = El<F16>;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;        // one thread = 16 output bytes
    const long row = i >> 2;
    const int part = (int)(i & 3);
    if (row >= rows) return;
    uint4 outv = make_uint4(0, 0, 0, 0);
    if (part < 3) {
        const uint4 a = *reinterpret_cast<const uint4*>(src + row * 48 + part * 16), c = *reinterpret_cast<const uint4*>(src + row * 48 + part * 16 + 8);
        float f[16];
        unpack8<F16>(a, f); unpack8<F16>(c, f + 8);
#pragma unroll
        for (int e = 0; e < 16; ++e) f[e] *= scale;
        if (part == 2) {
            f[8] = pad_val; f[9] = pad_val;
#pragma unroll
            for (int e = 10; e < 16; ++e) f[e] = 0.f;
        }
        outv = make_uint4(pack4_fp8(f[0], f[1], f[2], f[3]), pack4_fp8(f[4], f[5], f[6], f[7]), pack4_fp8(f[8], f[9], f[10], f[11]),
                          pack4_fp8(f[12], f[13], f[14], f[15]));
    }
    *reinterpret_cast<uint4*>(dst + row * 64 + part * 16) = outv;
}
// kind 1: V^T [G, 64, LP] 16-bit -> [G, 64, LP] e4m3 (rows >= 40 are not read by the kernel), keys permuted inside 64-groups
template <bool F16>
__global__ void quant_vt_kernel(const bf16_t* src, unsigned char* dst, long groups, int LP, float scale) {
    using E = El<F16>;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;        // one thread = 8 keys of one row
    const int per_row = LP >> 3;
    const long rowi = i / per_row;
    const int k0 = (int)(i - rowi * per_row) << 3;
    if (rowi >= groups * D) return;
    const long g = rowi / D;
    const int d = (int)(rowi - g * D);
    const uint4 a = *reinterpret_cast<const uint4*>(src + ((size_t)g * 64 + d) * LP + k0);
    float f[8];
    unpack8<F16>(a, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= scale;
    const int pos = (k0 & ~63) + 32 * ((k0 >> 3) & 1) + 8 * ((k0 >> 4) & 3);
    *reinterpret_cast<uint2*>(dst + ((size_t)g * 64 + d) * LP + pos) = make_uint2(pack4_fp8(f[0], f[1], f[2], f[3]), pack4_fp8(f[4], f[5], f[6], f[7]));
}

}  // namespace

int imd_launch_attn_quantize_fp8(const bf16_t* src, unsigned char* dst, int kind, long rows_or_groups, int LP, int exp2_scale,
                                 float pad_val, int dtype, hipStream_t s) {
    const bool h = dtype == IMD_DTYPE_F16;
    const float scale = __builtin_exp2f((float)exp2_scale);
    if (kind == 0) {
        const long n = rows_or_groups * 4;
        const unsigned blocks = (unsigned)((n + 255) / 256);
        if (h) hipLaunchKernelGGL(quant_rows_kernel<true>, dim3(blocks), dim3(256), 0, s, src, dst, rows_or_groups, scale, pad_val);
        else hipLaunchKernelGGL(quant_rows_kernel<false>, dim3(blocks), dim3(256), 0, s, src, dst, rows_or_groups, scale, pad_val);
    } else {
        if (LP % 64) return imd_set_error("attn_quantize_fp8: LP (%d) must be a multiple of 64", LP);
        const long n = rows_or_groups * D * (LP >> 3);
        const unsigned blocks = (unsigned)((n + 255) / 256);
        if (h) hipLaunchKernelGGL(quant_vt_kernel<true>, dim3(blocks), dim3(256), 0, s, src, dst, rows_or_groups, LP, scale);
        else hipLaunchKernelGGL(quant_vt_kernel<false>, dim3(blocks), dim3(256), 0, s, src, dst, rows_or_groups, LP, scale);
    }
    return imd_check_launch("attn_quantize_fp8");
}

int imd_launch_attention_fp8(const AttnParams& p, int eq, int ek, int ev, hipStream_t s) {
    if (p.D != D) return imd_set_error("attention_fp8: head dim %d (the fp8 kernel is the d = 40, UNet level-0 kernel)", p.D);
    if (p.B <= 0 || p.H <= 0 || p.N <= 0 || p.L1 <= 0) return imd_set_error("attention_fp8: empty problem");
    if (p.L1P % 64 || p.L1P < p.L1 || (p.k2 && (p.L2 <= 0 || p.L2P % 64 || p.L2P < p.L2))) return imd_set_error("attention_fp8: bad key-set sizes");
    if (p.kv1_bdiv <= 0 || (p.k2 && p.kv2_bdiv <= 0) || p.causal) return imd_set_error("attention_fp8: bad kv divisors / causal unsupported");
    if (eq + ek < 0 || eq + ek > 8) return imd_set_error("attention_fp8: eq + ek must be in [0, 8] (the pad slots hold 2^(eq+ek) in e4m3)");
    const bool h = p.dtype == IMD_DTYPE_F16;
    const void* kern = h ? reinterpret_cast<const void*>(attn40_fp8_kernel<true>) : reinterpret_cast<const void*>(attn40_fp8_kernel<false>);
    if (int rc_attr = imd_lds_attr(kern, LDS_BYTES, "attention_fp8")) return rc_attr;
    dim3 grid((p.N + 255) / 256, p.H, p.B);
    if (h) hipLaunchKernelGGL(attn40_fp8_kernel<true>, grid, dim3(256), LDS_BYTES, s, p, eq, ek, ev);
    else hipLaunchKernelGGL(attn40_fp8_kernel<false>, grid, dim3(256), LDS_BYTES, s, p, eq, ek, ev);
    return imd_check_launch("attention_fp8");
}
