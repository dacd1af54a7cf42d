// Row-resident linear layer for the 32x32-level token matrix (K = 640, N a multiple of 160): the K = N = 640 projections of
// the level-1 transformer blocks (proj_in, attn1.to_out, attn2.to_q, attn2.to_out, proj_out -- 25 launches per denoising step,
// M = 8192 rows) and `norm2 -> attn2.to_q` as one launch.  Same reference arithmetic and same reasons as row_linear.hip (the
// tiled kernel walks K in ten dependent round trips to memory per workgroup and runs at ~0.11 of the MFMA peak on this shape).
//
// A full 640-wide row block does not fit a wave's registers next to the accumulators, and M = 8192 gives only 64 blocks of
// 128 rows, so the work is cut differently from the K = 320 kernel:
//   * grid = (M / 128 row blocks) x (N / 160 channel groups) -- 64 x 4 = 256 workgroups for the 640 -> 640 layers;
//   * inside a workgroup, wave (tb, kh) holds token block tb (32 rows) x K-HALF kh (320 k) in 80 VGPRs (B-operand layout,
//     loaded straight from global memory) and multiplies it with the matching half of every weight row: 20 MFMAs per
//     32-channel chunk, partial sums;
//   * the 160 weight rows of the group stream through LDS in 5 chunks of 32 rows x 640 k (40 KB, LDS-DMA, 3-slot ring, rows
//     unpadded with piece p of row r stored at p ^ (r & 15) -> conflict-free ds_read_b128);
//   * the two waves (tb, 0) and (tb, 1) own HALF of the 32 x 32 result each (register quads {0,1} / {2,3}): each hands the
//     quads it does not own to its partner through 2 KB of LDS (double buffered, published by the next chunk's barrier), adds
//     what it receives and stores its 2 x 4 channels per token straight from registers -- bias from LDS, residual fetched one
//     chunk ahead -- while the next chunk is being multiplied;
//   * LayerNorm prologue: each wave sees half a row; the two halves exchange (sum, sum of squared deviations) through LDS.
#include "gemm_common.h"
#include "lds_dma.h"

namespace {

constexpr int R6_K = 640, R6_KH = 320;
constexpr int R6_STEPS = R6_KH / 16;              // 20 k-steps per wave and chunk
constexpr int R6_ROWB = R6_K * 2;                 // 1280 bytes per weight row = 80 pieces
constexpr int R6_CHUNK = 32 * R6_ROWB;            // 40960
constexpr int R6_RING = 3;
constexpr int R6_NG = 160;                        // channels per workgroup
constexpr int R6_NC = R6_NG / 32;                 // 5 chunks
constexpr int R6_OFF_RED = R6_RING * R6_CHUNK;    // 122880: partial-sum exchange, 2 buffers x 8 waves x 2 KB
constexpr int R6_OFF_BIAS = R6_OFF_RED + 2 * 8 * 2048;     // 155648
constexpr int R6_OFF_LN = R6_OFF_BIAS + R6_NG * 4;         // 156288: LayerNorm partials, 8 waves x 64 lanes x 8 B
constexpr int R6_LDS = R6_OFF_LN + 8 * 64 * 8;             // 160384 <= 163840

// GN (round 6): GroupNorm (+ SiLU) of the rows from the statistic partials of x (gn_in_*; see row_linear.hip)
template <bool F16, bool LN, bool GN = false>
__global__ __launch_bounds__(512, 1) void row_linear_k640_kernel(const ConvGemmParams p, const float ln_eps) {
    static_assert(!(LN && GN), "one prologue at a time");
    using E = El<F16>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, col = lane & 31;
    const int tb = wave & 3, kh = wave >> 2;
    // hardware workgroup b runs on XCD b % 8: give every XCD whole row blocks (all channel groups of a row block share one L2,
    // so the activation rows cross the fabric once)
    const int n_groups = p.N / R6_NG;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int mblk = (slot / n_groups) * 8 + xcd, grp = slot % n_groups;
    if (mblk * 128 >= p.M) return;
    const int m0 = mblk * 128, n0 = grp * R6_NG;
    const int m = m0 + tb * 32 + col;

    GnInReq<R6_K> gnreq;
    if constexpr (GN) gn_in_request<R6_K>(p, m0 / (p.Hout * p.Wout), gnreq);      // (ahead of the activation loads: they come back first)
    // ---- activations: 32 rows x 320 k (this wave's K half) straight into B-operand fragments ----
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    const uint32_t xoff = (uint32_t)m * (uint32_t)(p.x_pix_stride * 2) + (uint32_t)(kh * (R6_KH * 2) + hi * 16);
    uint4 xf[R6_STEPS];
#pragma unroll
    for (int s = 0; s < R6_STEPS; ++s) xf[s] = buf_load16(rs_x, m < p.M ? xoff + s * 32 : OOB);
    float bias_v = 0.f;
    if (tid < R6_NG && p.bias) bias_v = p.bias[n0 + tid];

    // ---- weight stream ----
    const v4i_t ds_w = raw_rsrc(p.w, p.w_bytes);
    uint32_t woff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int q = (j * 8 + wave) * 64 + lane;
        const int row = q / 80, pos = q - row * 80;
        woff[j] = (uint32_t)((n0 + row) * R6_ROWB + ((pos ^ (row & 15)) << 4));
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    auto stage = [&](int c) {
        const uint32_t base = lds0 + (uint32_t)((c % R6_RING) * R6_CHUNK) + (uint32_t)wave * 1024u;
#pragma unroll
        for (int j = 0; j < 5; ++j) dma16(ds_w, base + j * 8192u, woff[j] + (uint32_t)c * R6_CHUNK);
    };
    stage(0);
    stage(1);
    // hipcc counts only its own loads: pin their wait here (it also covers chunks 0 and 1, requested with them)
#pragma unroll
    for (int s = 0; s < R6_STEPS; ++s) asm volatile("" : "+v"(xf[s].x), "+v"(xf[s].y), "+v"(xf[s].z), "+v"(xf[s].w));
    if (tid < R6_NG) reinterpret_cast<float*>(smem + R6_OFF_BIAS)[tid] = bias_v;

    if constexpr (LN) {      // LayerNorm without affine over all 640 channels: the K halves meet in LDS twice (sum, squared deviations)
        float2* lnx = reinterpret_cast<float2*>(smem + R6_OFF_LN);
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < R6_STEPS; ++s) {
            float f[8];
            unpack8<F16>(xf[s], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += f[e];
        }
        sum += __shfl_xor(sum, 32);
        lnx[wave * 64 + lane].x = sum;
        __syncthreads();
        const float mean = (sum + lnx[(wave ^ 4) * 64 + lane].x) * (1.0f / R6_K);
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < R6_STEPS; ++s) asm volatile("" : "+v"(xf[s].x), "+v"(xf[s].y), "+v"(xf[s].z), "+v"(xf[s].w));
#pragma unroll
        for (int s = 0; s < R6_STEPS; ++s) {
            float f[8];
            unpack8<F16>(xf[s], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; sq = fmaf(d, d, sq); }
        }
        sq += __shfl_xor(sq, 32);
        lnx[wave * 64 + lane].y = sq;
        __syncthreads();
        const float rstd = rsqrtf((sq + lnx[(wave ^ 4) * 64 + lane].y) * (1.0f / R6_K) + ln_eps);
        const float shift = -mean * rstd;
#pragma unroll
        for (int s = 0; s < R6_STEPS; ++s) asm volatile("" : "+v"(xf[s].x), "+v"(xf[s].y), "+v"(xf[s].z), "+v"(xf[s].w));
#pragma unroll
        for (int s = 0; s < R6_STEPS; ++s) {
            float f[8];
            unpack8<F16>(xf[s], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], rstd, shift);
            xf[s] = pack8<F16>(f);
        }
    }

    if constexpr (GN) {      // (scratch = ring slot 2: nothing lands there before stage(2), issued behind the chunk loop's first barrier)
        const float *ga, *gs;
        gn_in_coeffs<R6_K>(p, m0 / (p.Hout * p.Wout), gnreq, reinterpret_cast<float*>(smem + 2 * R6_CHUNK), ga, gs);
        const bool silu = p.gn_in_silu != 0;
#pragma unroll
        for (int s = 0; s < R6_STEPS; ++s) {       // (one fragment at a time: see row_linear.hip)
            xf[s] = gn_in_apply8<F16>(xf[s], ga, gs, kh * R6_KH + 16 * s + 8 * hi, silu);
            asm volatile("" ::: "memory");        // (keeps the coefficient reads of the next fragment behind this one)
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- output side: this wave stores quads {2 kh, 2 kh + 1} of every chunk: channels n0 + 32 c + 8 q + 4 hi .. + 3 ----
    const int HWo = p.Hout * p.Wout;
    const bool heads = !GN && p.mode == OUT_HEADS;        // (GN: row-major output without residual only, see row_linear.hip)
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(heads ? (void*)p.hd[0].ptr : p.out, 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.res), 0, 0x80000000u, 0x00020000);
    uint32_t obase = OOB;
    if (m < p.M) {
        if (heads) { const int bi = m / HWo, tok = m - bi * HWo; obase = (uint32_t)(((size_t)bi * p.hH * p.hd[0].L + tok) * p.hd[0].DP * 2); }
        else obase = (uint32_t)m * (uint32_t)(p.out_ld * 2);
    }
    const bool has_res = !GN && p.res != nullptr;
    const uint32_t roff = (uint32_t)m * (uint32_t)(p.res_ld * 2);
    uint4 rraw[2];                     // residual of a chunk: (quad 2 kh | quad 2 kh + 1) in accumulator layout or, wide, 8 consecutive channels
    const bool wide = GN || (p.flags & 1024) != 0;      // 16-byte stores / residual loads of 8 consecutive channels (row_linear.hip: "WIDE stores"); knob 2 bit 10 = the 8-byte form
    auto load_res = [&](int c) {
        if (wide) rraw[c & 1] = buf_load16(rs_r, m < p.M ? roff + (uint32_t)((n0 + 32 * c + 16 * kh + 8 * hi) * 2) : OOB);
        else {
            const uint2 a = buf_load8(rs_r, m < p.M ? roff + (uint32_t)((n0 + 32 * c + 8 * (2 * kh) + 4 * hi) * 2) : OOB);
            const uint2 b = buf_load8(rs_r, m < p.M ? roff + (uint32_t)((n0 + 32 * c + 8 * (2 * kh + 1) + 4 * hi) * 2) : OOB);
            rraw[c & 1] = make_uint4(a.x, a.y, b.x, b.y);
        }
    };
    float4* red = reinterpret_cast<float4*>(smem + R6_OFF_RED);          // [2][8 waves][2 quads][64 lanes]
    const float* bias_s = reinterpret_cast<const float*>(smem + R6_OFF_BIAS);
    const float osc = heads ? p.out_scale * p.hd[0].scale : p.out_scale;
    float own[8];                      // this wave's two quads of the previous chunk (its own partial sums)
    auto emit = [&](int c) {           // chunk c: own partial + the partner's, bias, scale, residual, one 16-byte (or two 8-byte) stores
        typedef __attribute__((__vector_size__(2 * sizeof(uint32_t)))) uint32_t v2u;
        typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4u_t;
        v2u pk[2];
        uint2 rres[2];                 // residual in accumulator layout
        if (has_res) {
            uint4 r = rraw[c & 1];
            if (wide) {
                const auto sx = __builtin_amdgcn_permlane32_swap(r.x, r.z, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(r.y, r.w, false, false);
                r = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
            rres[0] = make_uint2(r.x, r.y); rres[1] = make_uint2(r.z, r.w);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 o = red[((c & 1) * 8 + (wave ^ 4)) * 128 + j * 64 + lane];
            const int nl = 32 * c + 8 * (2 * kh + j) + 4 * hi;       // channel inside the group
            const float4 bb = *reinterpret_cast<const float4*>(bias_s + nl);
            float v0 = (own[4 * j] + o.x + bb.x) * osc, v1 = (own[4 * j + 1] + o.y + bb.y) * osc;
            float v2 = (own[4 * j + 2] + o.z + bb.z) * osc, v3 = (own[4 * j + 3] + o.w + bb.w) * osc;
            if (has_res) {
                v0 += E::lo(rres[j].x); v1 += E::hi(rres[j].x);
                v2 += E::lo(rres[j].y); v3 += E::hi(rres[j].y);
            }
            pk[j] = v2u{E::pack2(v0, v1), E::pack2(v2, v3)};
        }
        auto offset_of = [&](int n) -> uint32_t {
            if (heads) { const int h = n / p.hD, dd = n - h * p.hD; return (uint32_t)((h * p.hd[0].L * p.hd[0].DP + dd) * 2); }
            return (uint32_t)(n * 2);
        };
        if (wide) {                    // quads 2 kh (channels 16 kh + 4 hi ..) and 2 kh + 1 (16 kh + 8 + 4 hi ..) -> channels 16 kh + 8 hi + 0..7 of the chunk
            const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
            const v4u_t w = {r0[0], r1[0], r0[1], r1[1]};
            __builtin_amdgcn_raw_buffer_store_b128(w, rs_o, (int)(obase == OOB ? OOB : obase + offset_of(n0 + 32 * c + 16 * kh + 8 * hi)), 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_buffer_store_b64(pk[j], rs_o, (int)(obase == OOB ? OOB : obase + offset_of(n0 + 32 * c + 8 * (2 * kh + j) + 4 * hi)), 0, 0);
        }
    };

    const uint32_t pb = (uint32_t)(kh * 40 + hi), sw = (uint32_t)(col & 15);
    const char* wlane = smem + col * R6_ROWB;
#pragma unroll
    for (int c = 0; c <= R6_NC; ++c) {
        dma_wait();                    // stores are in flight with the DMA pieces: no counted wait (see row_linear.hip)
        __syncthreads();               // chunk c landed for everybody; the partials of chunk c - 1 are published; slot of chunk c - 1 is free
        if (c > 0) emit(c - 1);
        if (c + 2 < R6_NC) stage(c + 2);
        if (c < R6_NC) {
            if (has_res) load_res(c);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const char* Ws = wlane + (c % R6_RING) * R6_CHUNK;
#pragma unroll
            for (int s = 0; s < R6_STEPS; ++s) {
                const uint4 wf = *reinterpret_cast<const uint4*>(Ws + (((pb + 2 * s) ^ sw) << 4));
                acc = E::mfma(wf, xf[s], acc);
            }
            // keep quads {2 kh, 2 kh + 1}, hand the other two to the partner wave (tb, 1 - kh)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 lo = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
                const float4 hi4 = make_float4(acc[8 + 4 * j], acc[9 + 4 * j], acc[10 + 4 * j], acc[11 + 4 * j]);
                const float4 mine = kh ? hi4 : lo, theirs = kh ? lo : hi4;
                own[4 * j] = mine.x; own[4 * j + 1] = mine.y; own[4 * j + 2] = mine.z; own[4 * j + 3] = mine.w;
                red[((c & 1) * 8 + wave) * 128 + j * 64 + lane] = theirs;
            }
        }
    }
}

template <bool F16, bool LN, bool GN = false>
int launch_r6(const ConvGemmParams& p, float eps, hipStream_t s) {
    auto kern = row_linear_k640_kernel<F16, LN, GN>;
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), R6_LDS, "row_linear_k640")) return rc_attr;
    const unsigned grid = (unsigned)((((p.M + 127) / 128 + 7) / 8) * 8 * (p.N / R6_NG));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), R6_LDS, s, p, eps);
    return imd_check_launch("row_linear_k640");
}

}  // namespace

bool imd_row_linear_k640_supported(const ConvGemmParams& p) {
    const bool direct = p.act == ACT_NONE && !p.out_f32 && p.rowvec == nullptr &&
                        (p.mode == OUT_ROWMAJOR || (p.hd[0].kind == 0 && p.hd[0].ptr != nullptr && p.N == p.hC));
    return direct && p.taps == 1 && p.K == R6_K && p.Cin == R6_K && p.stride == 1 && !p.ups && p.Hin == p.Hout && p.Win == p.Wout &&
           p.N >= R6_NG && (p.N % R6_NG) == 0 && p.split_k <= 1 && p.gn_a == nullptr && (p.x_pix_stride % 8) == 0 &&
           (p.mode != OUT_HEADS || (p.hD % 4) == 0);
}

int imd_launch_row_linear_k640(const ConvGemmParams& p_in, int ln, float ln_eps, hipStream_t s) {
    ConvGemmParams p = p_in;
    if (p_in.res_rows != 0) return imd_set_error("row_linear_k640: a periodic residual (res_rows) exists in the K = 320 row-resident projection only");
    if (!imd_row_linear_k640_supported(p))
        return imd_set_error("row_linear_k640: needs a plain linear layer with K = 640, N a multiple of 160 and a bias / scale / residual epilogue "
                             "(got N=%d K=%d taps=%d act=%d)", p.N, p.K, p.taps, p.act);
    if (p.dtype != IMD_DTYPE_BF16 && p.dtype != IMD_DTYPE_F16) return imd_set_error("row_linear_k640: unknown dtype %d", p.dtype);
    const size_t xb = ((size_t)(p.M - 1) * p.x_pix_stride + p.K) * 2, wb = (size_t)p.N * p.K * 2;
    const size_t ob = p.mode == OUT_HEADS ? (size_t)(p.M / (p.Hout * p.Wout)) * p.hH * p.hd[0].L * p.hd[0].DP * 2 : ((size_t)(p.M - 1) * p.out_ld + p.N) * 2;
    const size_t rb = p.res ? ((size_t)(p.M - 1) * p.res_ld + p.N) * 2 : 0;
    if (xb >= 0xffffffffull || ob >= 0x80000000ull || rb >= 0x80000000ull) return imd_set_error("row_linear_k640: operand too large");
    p.x_bytes = (uint32_t)xb;
    p.w_bytes = (uint32_t)wb;
    p.split_k = 1;
    p.flags = (g_gemm_flags & 1024) ? 0 : 1024;        // bit 10: wide (16-byte) stores of the direct epilogue
    const bool h = p.dtype == IMD_DTYPE_F16;
    if (p.gn_in_partial != nullptr) {
        if (ln || !gn_in_ok(p, R6_K, 128))
            return imd_set_error("row_linear_k640: gn_in_* needs K = 640, K %% groups == 0, groups <= 64, H W %% 128 == 0 and no LayerNorm prologue (ask imd_row_linear_gn_in_supported())");
        return h ? launch_r6<true, false, true>(p, ln_eps, s) : launch_r6<false, false, true>(p, ln_eps, s);
    }
    if (ln) return h ? launch_r6<true, true>(p, ln_eps, s) : launch_r6<false, true>(p, ln_eps, s);
    return h ? launch_r6<true, false>(p, ln_eps, s) : launch_r6<false, false>(p, ln_eps, s);
}
