// Row-resident linear layer for the 16x16-level token matrix (K = 1280, N a multiple of 160): the K = N = 1280 projections of the
// level-2 transformer blocks (25 launches per denoising step, M = 2048 rows) and `norm2 -> attn2.to_q` as one launch.  Third
// member of the family (row_linear.hip: K = 320, row_linear_k640.hip: K = 640); same reference arithmetic, same motivation.
//
// With M = 2048 there are only 64 blocks of 32 rows, and a 32-channel chunk of 1280-wide weight rows is 80 KB -- no room for a
// ring in LDS.  So this one runs on v_mfma_f32_16x16x32 (a weight chunk is 16 rows = 40 KB) and cuts K four ways:
//   * grid = (M / 64 row blocks) x (N / 160 channel groups) = 32 x 8 = 256 workgroups for the 1280 -> 1280 layers;
//   * wave (tb, kq) holds token block tb (32 rows, as two 16-token MFMA operands) x K-QUARTER kq (320 k) in 80 VGPRs and
//     produces partial sums of a 16-channel chunk for its 32 tokens with 20 MFMAs (every weight fragment feeds both token
//     halves: 0.5 KB of LDS reads per MFMA, the same bytes per FLOP as the 32x32 kernels);
//   * the 160 weight rows of the group stream through LDS in 10 chunks of 16 rows x 1280 k (40 KB, LDS-DMA, 3-slot ring, piece p
//     of row r stored at p ^ (r & 15));
//   * reduction: wave kq = 0 owns token half 0, wave kq = 1 token half 1; every wave parks the partial sums it does not own in
//     LDS (double buffered, published by the next chunk's barrier); the two owners add the three foreign partials, bias, scale,
//     residual (requested up front with the activations: a chunk step is shorter than a trip to HBM) and store 4 channels per
//     token straight from registers while the next chunk runs;
//   * LayerNorm prologue: the four K quarters exchange (sum, squared deviations) through LDS.
#include "gemm_common.h"
#include "lds_dma.h"

namespace {

constexpr int R12_K = 1280, R12_KQ = 320;
constexpr int R12_STEPS = R12_KQ / 32;            // 10 k32-steps per wave and chunk
constexpr int R12_ROWB = R12_K * 2;               // 2560 bytes per weight row = 160 pieces
constexpr int R12_CHUNK = 16 * R12_ROWB;          // 40960
constexpr int R12_RING = 3;
constexpr int R12_NG = 160;                       // channels per workgroup
constexpr int R12_NC = R12_NG / 16;               // 10 chunks
constexpr int R12_OFF_RED = R12_RING * R12_CHUNK;             // 122880: partial sums, 2 buffers x [2 tb][4 kq][2 th] x 1 KB
constexpr int R12_OFF_BIAS = R12_OFF_RED + 2 * 16 * 1024;     // 155648
constexpr int R12_OFF_LN = R12_OFF_BIAS + R12_NG * 4;         // 156288: LayerNorm partials: 8 waves x 64 lanes x 2 token halves x float
constexpr int R12_LDS = R12_OFF_LN + 8 * 64 * 8;              // 160384

template <bool F16> struct Mfma16;
template <> struct Mfma16<false> {
    static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma16<true> {
    static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
// v_mfma_f32_16x16x32: A lane l = row l & 15, k slots 8 (l >> 4) .. + 7;  B lane l = column l & 15, same k slots;
// D lane l = column l & 15, rows 4 (l >> 4) + r in register r.  Issued with A = weight rows (channels), B = token rows.

// GN (round 6): GroupNorm (+ SiLU) of the rows from the statistic partials of x (gn_in_*; see row_linear.hip)
template <bool F16, bool LN, bool GN = false>
__global__ __launch_bounds__(512, 1) void row_linear_k1280_kernel(const ConvGemmParams p, const float ln_eps) {
    static_assert(!(LN && GN), "one prologue at a time");
    using E = El<F16>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    const int tb = wave & 1, kq = wave >> 1;
    // hardware workgroup b runs on XCD b % 8: whole row blocks per XCD (all channel groups of a row block share one L2)
    const int n_groups = p.N / R12_NG;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int mblk = (slot / n_groups) * 8 + xcd, grp = slot % n_groups;
    if (mblk * 64 >= p.M) return;
    const int m0 = mblk * 64 + tb * 32, n0 = grp * R12_NG;

    GnInReq<R12_K> gnreq;
    if constexpr (GN) gn_in_request<R12_K>(p, (mblk * 64) / (p.Hout * p.Wout), gnreq);      // (ahead of the activation loads: they come back first)
    // ---- activations: 2 x 16 rows x 320 k (this wave's K quarter) straight into B-operand fragments ----
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    uint4 xf[2][R12_STEPS];
#pragma unroll
    for (int th = 0; th < 2; ++th) {
        const int m = m0 + th * 16 + col;
        const uint32_t xoff = (uint32_t)m * (uint32_t)(p.x_pix_stride * 2) + (uint32_t)(kq * (R12_KQ * 2) + g * 16);
#pragma unroll
        for (int s = 0; s < R12_STEPS; ++s) xf[th][s] = buf_load16(rs_x, m < p.M ? xoff + s * 64 : OOB);
    }
    float bias_v = 0.f;
    if (tid < R12_NG && p.bias) bias_v = p.bias[n0 + tid];
    // residual of the 4 channels x 10 chunks this lane will store (owner waves only), requested with the activations: a chunk
    // step is shorter than a trip to HBM here, so fetching it one chunk ahead (as the wider kernels do) stalls every step
    const bool owner = kq < 2;
    const int mo = m0 + (kq & 1) * 16 + col;
    const bool has_res = !GN && owner && p.res != nullptr;      // (GN: row-major output without residual only, see row_linear.hip)
    uint2 rres[R12_NC];
    if (has_res) {
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.res), 0, 0x80000000u, 0x00020000);
        const uint32_t roff = (uint32_t)mo * (uint32_t)(p.res_ld * 2) + (uint32_t)((n0 + 4 * g) * 2);
#pragma unroll
        for (int c = 0; c < R12_NC; ++c) rres[c] = buf_load8(rs_r, mo < p.M ? roff + (uint32_t)(c * 32) : OOB);
    }

    // ---- weight stream ----
    const v4i_t ds_w = raw_rsrc(p.w, p.w_bytes);
    uint32_t woff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int q = (j * 8 + wave) * 64 + lane;
        const int row = q / 160, pos = q - row * 160;
        woff[j] = (uint32_t)((n0 + row) * R12_ROWB + ((pos ^ (row & 15)) << 4));
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    auto stage = [&](int c) {
        const uint32_t base = lds0 + (uint32_t)((c % R12_RING) * R12_CHUNK) + (uint32_t)wave * 1024u;
#pragma unroll
        for (int j = 0; j < 5; ++j) dma16(ds_w, base + j * 8192u, woff[j] + (uint32_t)c * R12_CHUNK);
    };
    stage(0);
    stage(1);
#pragma unroll
    for (int th = 0; th < 2; ++th)
#pragma unroll
        for (int s = 0; s < R12_STEPS; ++s) asm volatile("" : "+v"(xf[th][s].x), "+v"(xf[th][s].y), "+v"(xf[th][s].z), "+v"(xf[th][s].w));
    if (tid < R12_NG) reinterpret_cast<float*>(smem + R12_OFF_BIAS)[tid] = bias_v;

    if constexpr (GN) {      // (scratch = ring slot 2: nothing lands there before stage(2), issued behind the chunk loop's first barrier; the 64 rows lie in ONE image)
        const float *ga, *gs;
        gn_in_coeffs<R12_K>(p, (mblk * 64) / (p.Hout * p.Wout), gnreq, reinterpret_cast<float*>(smem + 2 * R12_CHUNK), ga, gs);
        const bool silu = p.gn_in_silu != 0;
#pragma unroll
        for (int th = 0; th < 2; ++th)
#pragma unroll
            for (int s = 0; s < R12_STEPS; ++s) {  // (one fragment at a time: see row_linear.hip)
                xf[th][s] = gn_in_apply8<F16>(xf[th][s], ga, gs, kq * R12_KQ + 32 * s + 8 * g, silu);
                asm volatile("" ::: "memory");        // (keeps the coefficient reads of the next fragment behind this one)
            __builtin_amdgcn_sched_barrier(0);
            }
    }

    if constexpr (LN) {      // LayerNorm without affine over all 1280 channels: a token's row is spread over 4 lanes (g) x 4 waves (kq)
        float* lnx = reinterpret_cast<float*>(smem + R12_OFF_LN);           // [8 waves][2 th][64 lanes]
        float mean[2], rstd[2];
        float part[2];
#pragma unroll
        for (int th = 0; th < 2; ++th) {
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < R12_STEPS; ++s) {
                float f[8];
                unpack8<F16>(xf[th][s], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += f[e];
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            part[th] = sum;
            lnx[(wave * 2 + th) * 64 + lane] = sum;
        }
        __syncthreads();
#pragma unroll
        for (int th = 0; th < 2; ++th) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) t += lnx[((tb + 2 * k) * 2 + th) * 64 + lane];
            mean[th] = t * (1.0f / R12_K);
        }
        __syncthreads();
#pragma unroll
        for (int th = 0; th < 2; ++th)
#pragma unroll
            for (int s = 0; s < R12_STEPS; ++s) asm volatile("" : "+v"(xf[th][s].x), "+v"(xf[th][s].y), "+v"(xf[th][s].z), "+v"(xf[th][s].w));
#pragma unroll
        for (int th = 0; th < 2; ++th) {
            float sq = 0.f;
#pragma unroll
            for (int s = 0; s < R12_STEPS; ++s) {
                float f[8];
                unpack8<F16>(xf[th][s], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[e] - mean[th]; sq = fmaf(d, d, sq); }
            }
            sq += __shfl_xor(sq, 16);
            sq += __shfl_xor(sq, 32);
            lnx[(wave * 2 + th) * 64 + lane] = sq;
        }
        __syncthreads();
#pragma unroll
        for (int th = 0; th < 2; ++th) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) t += lnx[((tb + 2 * k) * 2 + th) * 64 + lane];
            rstd[th] = rsqrtf(t * (1.0f / R12_K) + ln_eps);
        }
        (void)part;
#pragma unroll
        for (int th = 0; th < 2; ++th)
#pragma unroll
            for (int s = 0; s < R12_STEPS; ++s) asm volatile("" : "+v"(xf[th][s].x), "+v"(xf[th][s].y), "+v"(xf[th][s].z), "+v"(xf[th][s].w));
#pragma unroll
        for (int th = 0; th < 2; ++th) {
            const float shift = -mean[th] * rstd[th];
#pragma unroll
            for (int s = 0; s < R12_STEPS; ++s) {
                float f[8];
                unpack8<F16>(xf[th][s], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], rstd[th], shift);
                xf[th][s] = pack8<F16>(f);
            }
        }
    }

    // ---- output side: waves kq = 0 / 1 own token half 0 / 1: channels n0 + 16 c + 4 g .. + 3 of token m0 + 16 kq + col ----
    const int HWo = p.Hout * p.Wout;
    const bool heads = !GN && p.mode == OUT_HEADS;
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(heads ? (void*)p.hd[0].ptr : p.out, 0, 0x80000000u, 0x00020000);
    uint32_t obase = OOB;
    if (owner && mo < p.M) {
        if (heads) { const int bi = mo / HWo, tok = mo - bi * HWo; obase = (uint32_t)(((size_t)bi * p.hH * p.hd[0].L + tok) * p.hd[0].DP * 2); }
        else obase = (uint32_t)mo * (uint32_t)(p.out_ld * 2);
    }
    float4* red = reinterpret_cast<float4*>(smem + R12_OFF_RED);          // [2 buffers][2 tb][4 kq][2 th][64 lanes]
    const float* bias_s = reinterpret_cast<const float*>(smem + R12_OFF_BIAS);
    const float osc = heads ? p.out_scale * p.hd[0].scale : p.out_scale;
    f32x4 own;                         // the owner's own partial sums of the previous chunk
    own[0] = own[1] = own[2] = own[3] = 0.f;
    auto emit = [&](int c) {           // owners only: chunk c = own + three foreign partials, bias, scale, residual, one 8-byte store
        float4 t = make_float4(own[0], own[1], own[2], own[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k == kq) continue;
            const float4 o = red[((((c & 1) * 2 + tb) * 4 + k) * 2 + kq) * 64 + lane];
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        const int nl = 16 * c + 4 * g;
        const float4 bb = *reinterpret_cast<const float4*>(bias_s + nl);
        float v0 = (t.x + bb.x) * osc, v1 = (t.y + bb.y) * osc, v2 = (t.z + bb.z) * osc, v3 = (t.w + bb.w) * osc;
        if (has_res) {
            v0 += E::lo(rres[c].x); v1 += E::hi(rres[c].x);
            v2 += E::lo(rres[c].y); v3 += E::hi(rres[c].y);
        }
        const int n = n0 + nl;
        uint32_t off;
        if (heads) { const int h = n / p.hD, dd = n - h * p.hD; off = (uint32_t)((h * p.hd[0].L * p.hd[0].DP + dd) * 2); }
        else off = (uint32_t)(n * 2);
        typedef __attribute__((__vector_size__(2 * sizeof(uint32_t)))) uint32_t v2u;
        const v2u pk = {E::pack2(v0, v1), E::pack2(v2, v3)};
        __builtin_amdgcn_raw_buffer_store_b64(pk, rs_o, (int)(obase == OOB ? OOB : obase + off), 0, 0);
    };

    const uint32_t pb = (uint32_t)(kq * 40 + g), sw = (uint32_t)col;       // fragment piece = kq * 40 + 4 s + g, stored at piece ^ row
    const char* wlane = smem + col * R12_ROWB;
#pragma unroll
    for (int c = 0; c <= R12_NC; ++c) {
        dma_wait();                    // stores are in flight with the DMA pieces: no counted wait (see row_linear.hip)
        __syncthreads();               // chunk c landed for everybody; the partials of chunk c - 1 are published; slot of chunk c - 1 is free
        if (c > 0 && owner) emit(c - 1);
        if (c + 2 < R12_NC) stage(c + 2);
        if (c < R12_NC) {
            f32x4 acc[2];
#pragma unroll
            for (int th = 0; th < 2; ++th) acc[th][0] = acc[th][1] = acc[th][2] = acc[th][3] = 0.f;
            const char* Ws = wlane + (c % R12_RING) * R12_CHUNK;
#pragma unroll
            for (int s = 0; s < R12_STEPS; ++s) {
                const uint4 wf = *reinterpret_cast<const uint4*>(Ws + (((pb + 4 * s) ^ sw) << 4));
                acc[0] = Mfma16<F16>::run(wf, xf[0][s], acc[0]);
                acc[1] = Mfma16<F16>::run(wf, xf[1][s], acc[1]);
            }
            // token half th belongs to wave kq == th: everything else is parked for its owner
#pragma unroll
            for (int th = 0; th < 2; ++th) {
                if (th == kq) own = acc[th];
                else red[((((c & 1) * 2 + tb) * 4 + kq) * 2 + th) * 64 + lane] = make_float4(acc[th][0], acc[th][1], acc[th][2], acc[th][3]);
            }
        }
    }
}

template <bool F16, bool LN, bool GN = false>
int launch_r12(const ConvGemmParams& p, float eps, hipStream_t s) {
    auto kern = row_linear_k1280_kernel<F16, LN, GN>;
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), R12_LDS, "row_linear_k1280")) return rc_attr;
    const unsigned grid = (unsigned)((((p.M + 63) / 64 + 7) / 8) * 8 * (p.N / R12_NG));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), R12_LDS, s, p, eps);
    return imd_check_launch("row_linear_k1280");
}

}  // namespace

bool imd_row_linear_k1280_supported(const ConvGemmParams& p) {
    const bool direct = p.act == ACT_NONE && !p.out_f32 && p.rowvec == nullptr &&
                        (p.mode == OUT_ROWMAJOR || (p.hd[0].kind == 0 && p.hd[0].ptr != nullptr && p.N == p.hC));
    return direct && p.taps == 1 && p.K == R12_K && p.Cin == R12_K && p.stride == 1 && !p.ups && p.Hin == p.Hout && p.Win == p.Wout &&
           p.N >= R12_NG && (p.N % R12_NG) == 0 && p.split_k <= 1 && p.gn_a == nullptr && (p.x_pix_stride % 8) == 0 &&
           (p.mode != OUT_HEADS || (p.hD % 4) == 0);
}

int imd_launch_row_linear_k1280(const ConvGemmParams& p_in, int ln, float ln_eps, hipStream_t s) {
    ConvGemmParams p = p_in;
    if (p_in.res_rows != 0) return imd_set_error("row_linear_k1280: a periodic residual (res_rows) exists in the K = 320 row-resident projection only");
    if (!imd_row_linear_k1280_supported(p))
        return imd_set_error("row_linear_k1280: needs a plain linear layer with K = 1280, N a multiple of 160 and a bias / scale / residual epilogue "
                             "(got N=%d K=%d taps=%d act=%d)", p.N, p.K, p.taps, p.act);
    if (p.dtype != IMD_DTYPE_BF16 && p.dtype != IMD_DTYPE_F16) return imd_set_error("row_linear_k1280: unknown dtype %d", p.dtype);
    const size_t xb = ((size_t)(p.M - 1) * p.x_pix_stride + p.K) * 2, wb = (size_t)p.N * p.K * 2;
    const size_t ob = p.mode == OUT_HEADS ? (size_t)(p.M / (p.Hout * p.Wout)) * p.hH * p.hd[0].L * p.hd[0].DP * 2 : ((size_t)(p.M - 1) * p.out_ld + p.N) * 2;
    const size_t rb = p.res ? ((size_t)(p.M - 1) * p.res_ld + p.N) * 2 : 0;
    if (xb >= 0xffffffffull || ob >= 0x80000000ull || rb >= 0x80000000ull) return imd_set_error("row_linear_k1280: operand too large");
    p.x_bytes = (uint32_t)xb;
    p.w_bytes = (uint32_t)wb;
    p.split_k = 1;
    p.flags = 0;
    const bool h = p.dtype == IMD_DTYPE_F16;
    if (p.gn_in_partial != nullptr) {
        if (ln || !gn_in_ok(p, R12_K, 64))
            return imd_set_error("row_linear_k1280: gn_in_* needs K = 1280, K %% groups == 0, groups <= 64, H W %% 64 == 0 and no LayerNorm prologue (ask imd_row_linear_gn_in_supported())");
        return h ? launch_r12<true, false, true>(p, ln_eps, s) : launch_r12<false, false, true>(p, ln_eps, s);
    }
    if (ln) return h ? launch_r12<true, true>(p, ln_eps, s) : launch_r12<false, true>(p, ln_eps, s);
    return h ? launch_r12<true, false>(p, ln_eps, s) : launch_r12<false, false>(p, ln_eps, s);
}
