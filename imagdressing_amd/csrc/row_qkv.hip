// norm1 -> attn1.to_q / to_k / to_v of the 64x64-level transformer blocks (C = 320, 8 heads x 40) as ONE launch, writing the
// three operands straight into the layouts the attention kernel consumes (Q / K [B, H, N, 48] row-major per head, Q pre-scaled;
// V^T [B, H, 64, pad64(N)] transposed).  Row-resident scheme of row_linear.hip (32 token rows x all 320 k of a wave in 80
// VGPRs, normalised in place; the 960 weight rows stream through LDS by DMA in 15 chunks of 64 rows), with two differences:
//   * only two accumulator blocks are alive (the chunk being multiplied and the one being stored): N = 960 does not fit;
//   * the V chunks are multiplied with the MFMA operands SWAPPED, D[token][channel] instead of D[channel][token]: a lane then
//     holds 4 consecutive TOKENS of one channel per accumulator quad, which is a contiguous 8-byte piece of a V^T row.
// Reference arithmetic: diffusers==0.24.0 BasicTransformerBlock.norm1 + Attention.to_q/to_k/to_v (un-vendored; call sites
// /root/reference/adapter/attention_processor.py:568-588) -- the LayerNorm affine is folded into the weights by the host.
#include "gemm_common.h"
#include "lds_dma.h"

namespace {

constexpr int RQ_K = 320, RQ_STEPS = 20, RQ_ROWB = 640, RQ_CH = 64, RQ_CHUNK = RQ_CH * RQ_ROWB, RQ_RING = 3;
constexpr int RQ_N = 960, RQ_NC = RQ_N / RQ_CH;        // 15 chunks: 0..4 Q, 5..9 K, 10..14 V
constexpr int RQ_LDS = RQ_RING * RQ_CHUNK + RQ_N * 4;  // + the bias vector

template <bool F16, bool LN>
__global__ __launch_bounds__(512, 1) void row_qkv_kernel(const ConvGemmParams p, const float ln_eps) {
    using E = El<F16>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, col = lane & 31;
    const int rb = wave & 3, chh = wave >> 2;
    const int m0 = blockIdx.x * 128;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    const int m = m0 + rb * 32 + col;
    const uint32_t xoff = (uint32_t)m * (uint32_t)(p.x_pix_stride * 2) + hi * 16;
    uint4 xf[RQ_STEPS];
#pragma unroll
    for (int s = 0; s < RQ_STEPS; ++s) xf[s] = buf_load16(rs_x, m < p.M ? xoff + s * 32 : OOB);
    float bias_v[2] = {0.f, 0.f};
    if (p.bias) {
        bias_v[0] = p.bias[tid];
        if (tid + 512 < RQ_N) bias_v[1] = p.bias[tid + 512];
    }

    const v4i_t ds_w = raw_rsrc(p.w, p.w_bytes);
    uint32_t woff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int q = (j * 8 + wave) * 64 + lane;
        const int row = q / 40, pos = q - row * 40;
        woff[j] = (uint32_t)(row * RQ_ROWB + ((pos ^ ((row >> 1) & 7)) << 4));
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    auto stage = [&](int c) {
        const uint32_t base = lds0 + (uint32_t)((c % RQ_RING) * RQ_CHUNK) + (uint32_t)wave * 1024u;
#pragma unroll
        for (int j = 0; j < 5; ++j) dma16(ds_w, base + j * 8192u, woff[j] + (uint32_t)c * RQ_CHUNK);
    };
    stage(0);
    stage(1);
#pragma unroll
    for (int s = 0; s < RQ_STEPS; ++s) asm volatile("" : "+v"(xf[s].x), "+v"(xf[s].y), "+v"(xf[s].z), "+v"(xf[s].w));
    {
        float* bs = reinterpret_cast<float*>(smem + RQ_RING * RQ_CHUNK);
        bs[tid] = bias_v[0];
        if (tid + 512 < RQ_N) bs[tid + 512] = bias_v[1];
    }

    if constexpr (LN) {
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < RQ_STEPS; ++s) {
            float f[8];
            unpack8<F16>(xf[s], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += f[e];
        }
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / RQ_K);
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < RQ_STEPS; ++s) asm volatile("" : "+v"(xf[s].x), "+v"(xf[s].y), "+v"(xf[s].z), "+v"(xf[s].w));
#pragma unroll
        for (int s = 0; s < RQ_STEPS; ++s) {
            float f[8];
            unpack8<F16>(xf[s], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; sq = fmaf(d, d, sq); }
        }
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq * (1.0f / RQ_K) + ln_eps);
        const float shift = -mean * rstd;
#pragma unroll
        for (int s = 0; s < RQ_STEPS; ++s) asm volatile("" : "+v"(xf[s].x), "+v"(xf[s].y), "+v"(xf[s].z), "+v"(xf[s].w));
#pragma unroll
        for (int s = 0; s < RQ_STEPS; ++s) {
            float f[8];
            unpack8<F16>(xf[s], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], rstd, shift);
            xf[s] = pack8<F16>(f);
        }
    }

    // ---- output addressing.  The 128 rows of a workgroup lie in ONE image (HW % 128 == 0, checked by the launcher). ----
    const int HWo = p.Hout * p.Wout;
    const int bi = m0 / HWo, tok0 = m0 - bi * HWo + rb * 32;          // first token of this wave's block inside its image
    const float* bs = reinterpret_cast<const float*>(smem + RQ_RING * RQ_CHUNK);
    typedef __attribute__((__vector_size__(2 * sizeof(uint32_t)))) uint32_t v2u;
    auto emit = [&](int c, const f32x16& acc) {
        const int which = c / 5;                                      // 0 Q, 1 K, 2 V (compile-time after unrolling)
        const HeadsDest hd = p.hd[which];
        if (hd.ptr == nullptr) return;
        const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(hd.ptr, 0, 0x80000000u, 0x00020000);
        if (which < 2) {          // D[channel][token]: lane = token, quad j = channels 8 j + 4 hi .. + 3
            const bool ok = m < p.M;
            const uint32_t obase = (uint32_t)(((size_t)bi * p.hH * hd.L + (tok0 + col)) * hd.DP * 2);
            // (round 5) 16-byte stores of 8 consecutive channels: one v_permlane32_swap per packed register pair (row_linear.hip: "WIDE stores");
            // an 8-channel group never straddles a head (head dim 40 = 5 x 8)
            typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4u_t;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint32_t pk[2][2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * t + jj, n = c * 64 + chh * 32 + 8 * j + 4 * hi;     // channel in [0, 960)
                    const float4 bb = *reinterpret_cast<const float4*>(bs + n);
                    pk[jj][0] = E::pack2((acc[4 * j] + bb.x) * hd.scale, (acc[4 * j + 1] + bb.y) * hd.scale);
                    pk[jj][1] = E::pack2((acc[4 * j + 2] + bb.z) * hd.scale, (acc[4 * j + 3] + bb.w) * hd.scale);
                }
                const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                const v4u_t w = {r0[0], r1[0], r0[1], r1[1]};
                const int n8 = c * 64 + chh * 32 + 16 * t + 8 * hi;
                const int nc = n8 - which * RQ_K, h = nc / p.hD, dd = nc - h * p.hD;
                __builtin_amdgcn_raw_buffer_store_b128(w, rs_o, (int)(ok ? obase + (uint32_t)((h * hd.L * hd.DP + dd) * 2) : OOB), 0, 0);
            }
        } else {                  // swapped MFMA, D[token][channel]: lane = channel, quad j = tokens 8 j + 4 hi .. + 3
            const int n = c * 64 + chh * 32 + col;
            const int nc = n - 2 * RQ_K, h = nc / p.hD, dd = nc - h * p.hD;
            const float b = bs[n];
            const uint32_t obase = (uint32_t)((((size_t)bi * p.hH + h) * hd.DP + dd) * hd.L * 2);       // hd.L = padded row length
            typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4u_t;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {      // quads 2 tt, 2 tt + 1 -> 8 consecutive tokens 16 tt + 8 hi .. of this lane's channel: one 16-byte store
                uint32_t pk[2][2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * tt + jj;
                    pk[jj][0] = E::pack2((acc[4 * j] + b) * hd.scale, (acc[4 * j + 1] + b) * hd.scale);
                    pk[jj][1] = E::pack2((acc[4 * j + 2] + b) * hd.scale, (acc[4 * j + 3] + b) * hd.scale);
                }
                const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                const v4u_t w = {r0[0], r1[0], r0[1], r1[1]};
                const int t = 16 * tt + 8 * hi;
                __builtin_amdgcn_raw_buffer_store_b128(w, rs_o, (int)(m0 + rb * 32 + t < p.M ? obase + (uint32_t)((tok0 + t) * 2) : OOB), 0, 0);
            }
        }
    };

    const int wrow = chh * 32 + col;
    const uint32_t a16 = (uint32_t)((hi ^ ((wrow >> 1) & 7)) << 4);
    const char* wlane = smem + wrow * RQ_ROWB;
    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < RQ_NC; ++c) {
        dma_wait();                    // stores are in flight with the DMA pieces: no counted wait (see row_linear.hip)
        __syncthreads();
        if (c > 0) emit(c - 1, acc[(c - 1) & 1]);
        if (c + 2 < RQ_NC) stage(c + 2);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c & 1][r] = 0.f;
        const char* Ws = wlane + (c % RQ_RING) * RQ_CHUNK;
#pragma unroll
        for (int s = 0; s < RQ_STEPS; ++s) {
            const uint4 wf = *reinterpret_cast<const uint4*>(Ws + ((uint32_t)(s * 32) ^ a16));
            if (c < 10) acc[c & 1] = E::mfma(wf, xf[s], acc[c & 1]);
            else acc[c & 1] = E::mfma(xf[s], wf, acc[c & 1]);
        }
    }
    emit(RQ_NC - 1, acc[(RQ_NC - 1) & 1]);
}

template <bool F16, bool LN>
int launch_rq(const ConvGemmParams& p, float eps, hipStream_t s) {
    auto kern = row_qkv_kernel<F16, LN>;
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), RQ_LDS, "row_qkv")) return rc_attr;
    hipLaunchKernelGGL(kern, dim3((unsigned)((p.M + 127) / 128)), dim3(512), RQ_LDS, s, p, eps);
    return imd_check_launch("row_qkv");
}

}  // namespace

// the fused q / k / v projection of a 320-channel self-attention layer: head-split epilogue with (Q row-major, K row-major, V^T)
bool imd_row_qkv_supported(const ConvGemmParams& p) {
    if (!(p.taps == 1 && p.K == RQ_K && p.Cin == RQ_K && p.N == RQ_N && p.mode == OUT_HEADS && p.hC == RQ_K && p.stride == 1 && !p.ups &&
          p.Hin == p.Hout && p.Win == p.Wout && p.split_k <= 1 && p.act == ACT_NONE && !p.out_f32 && p.rowvec == nullptr && p.res == nullptr &&
          p.gn_a == nullptr && (p.x_pix_stride % 8) == 0 && p.out_scale == 1.0f && (p.hD % 8) == 0)) return false;      // (hD % 8: the 16-byte Q / K stores write 8 consecutive channels of ONE head)
    const int HWo = p.Hout * p.Wout;
    if (HWo % 128) return false;                                   // a 128-row workgroup must not straddle two images
    for (int i = 0; i < 3; ++i) {
        if (p.hd[i].ptr == nullptr) continue;
        if (p.hd[i].kind != (i == 2 ? 1 : 0)) return false;
        if (i == 2 && (p.hd[i].L % 4)) return false;
    }
    return true;
}

int imd_launch_row_qkv(const ConvGemmParams& p_in, int ln, float ln_eps, hipStream_t s) {
    ConvGemmParams p = p_in;
    if (p_in.res_rows != 0) return imd_set_error("row_qkv: a periodic residual (res_rows) exists in the K = 320 row-resident projection only");
    if (!imd_row_qkv_supported(p)) return imd_set_error("row_qkv: needs the 320 -> 960 head-split q/k/v projection (Q, K row-major, V transposed), HW %% 128 == 0");
    if (p.dtype != IMD_DTYPE_BF16 && p.dtype != IMD_DTYPE_F16) return imd_set_error("row_qkv: unknown dtype %d", p.dtype);
    const size_t xb = ((size_t)(p.M - 1) * p.x_pix_stride + p.K) * 2;
    const size_t B = (size_t)(p.M / (p.Hout * p.Wout));
    for (int i = 0; i < 3; ++i)
        if (p.hd[i].ptr && B * p.hH * p.hd[i].L * p.hd[i].DP * 2 >= 0x80000000ull) return imd_set_error("row_qkv: operand too large");
    if (xb >= 0xffffffffull) return imd_set_error("row_qkv: operand too large");
    p.x_bytes = (uint32_t)xb;
    p.w_bytes = (uint32_t)((size_t)p.N * p.K * 2);
    p.split_k = 1;
    p.flags = 0;
    const bool h = p.dtype == IMD_DTYPE_F16;
    if (ln) return h ? launch_rq<true, true>(p, ln_eps, s) : launch_rq<false, true>(p, ln_eps, s);
    return h ? launch_rq<true, false>(p, ln_eps, s) : launch_rq<false, false>(p, ln_eps, s);
}
