// Implicit-GEMM convolution / linear layer for gfx950 (bf16 MFMA, fp32 accumulate).
//
//   out[m, n] = epilogue( sum_k A(m, k) * W[n, k] )        m = output pixel / token, n = channel
//
// * A is gathered on the fly from an NHWC activation tensor (3x3 / 1x1 taps, stride 1|2, zero
//   padding, optional fused nearest-2x upsample of the input) -- for a plain linear layer it is
//   simply the [M, K] row-major matrix (taps = 1, H = W = 1).
// * W is the pre-packed weight [N][K] with k = tap * Cin + ci  (torch Linear layout; conv weights
//   are repacked once at load time from [Cout][Cin][3][3] to [Cout][ky][kx][Cin]).
// * Tiles are staged global -> registers -> LDS (double buffered, one barrier per K tile; the
//   register hop is what lets the gather zero-fill the padding halo), rows padded by 16 B so
//   that the 16-lane groups of a ds_read_b128 hit 16 distinct 16-B bank slots.
// * The MFMA is issued "swapped" (A-operand = weight rows, B-operand = activation rows) so that
//   in the accumulator a lane owns ONE output row m and 4 consecutive channels per register
//   quad: bias / time-embedding / residual / SiLU / GEGLU / head-split epilogues are all
//   lane-local and stores are 8-byte packed bf16x4.
//
// Reference arithmetic this replaces (diffusers==0.24.0, un-vendored; call sites
// /root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline.py:466,499,511):
// ResnetBlock2D conv1/conv2/conv_shortcut, Down/Upsample2D conv, Transformer2DModel proj_in/
// proj_out, Attention to_q/to_k/to_v/to_out, GEGLU FeedForward, TimestepEmbedding, and the
// to_k_ref/to_v_ref garment projections of adapter/attention_processor.py:600-601.
#include "common.h"
#include "imd_kernels.h"

namespace {

template <bool F16, int BM, int BN, int BK, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemmParams p) {
    using E = El<F16>;
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = BN / WAVES_N / 32;
    constexpr int STRIDE = BK * 2 + 16;        // bytes per LDS row (odd number of 16-B slots)
    constexpr int VPR = BK / 8;                // 16-B vectors per row
    constexpr int RSTEP = 256 / VPR;           // rows covered by one pass of the 256 threads
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
    const int tile_m = blockIdx.x / n_tiles;
    const int tile_n = blockIdx.x - tile_m * n_tiles;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const int kc = tid % VPR;          // this thread's 16-B column inside a K tile
    const int r0 = tid / VPR;          // first row it stages

    // ---- per-row gather setup (rows are fixed for the whole K loop) ----
    const int HWo = p.Hout * p.Wout;
    const int pad = (p.taps == 9) ? 1 : 0;
    const int Hl = p.ups ? p.Hin * 2 : p.Hin;   // logical (post-upsample) input size
    const int Wl = p.ups ? p.Win * 2 : p.Win;
    int a_base[A_VECS];    // pixel index of (b, 0, 0), or -1 when the row is out of range
    int a_yx[A_VECS];      // (iy0 << 16) | (ix0 & 0xffff), top-left tap position
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) {
        const int m = m0 + r0 + i * RSTEP;
        if (m < p.M) {
            const int b = m / HWo;
            const int rem = m - b * HWo;
            const int oy = rem / p.Wout;
            const int ox = rem - oy * p.Wout;
            a_base[i] = b * p.Hin * p.Win;
            a_yx[i] = (int)(((unsigned)(oy * p.stride - pad)) << 16) | ((ox * p.stride - pad) & 0xffff);
        } else {
            a_base[i] = -1;
            a_yx[i] = 0;
        }
    }

    uint4 a_reg[A_VECS], w_reg[W_VECS];

    auto load_tile = [&](int k0) {
        const int k = k0 + kc * 8;
        const bool kv = k < p.K;
        int tap = 0, ci = k;
        if (p.taps == 9) { tap = k / p.Cin; ci = k - tap * p.Cin; }
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kv && a_base[i] >= 0) {
                int iy = (a_yx[i] >> 16) + ky;
                int ix = (int)(short)(a_yx[i] & 0xffff) + kx;
                if (iy >= 0 && iy < Hl && ix >= 0 && ix < Wl) {
                    if (p.ups) { iy >>= 1; ix >>= 1; }
                    const size_t off = (size_t)(a_base[i] + iy * p.Win + ix) * (size_t)p.x_pix_stride + ci;
                    v = *reinterpret_cast<const uint4*>(p.x + off);
                }
            }
            a_reg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < W_VECS; ++i) {
            const int n = n0 + r0 + i * RSTEP;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kv && n < p.N) v = *reinterpret_cast<const uint4*>(p.w + (size_t)n * p.K + k);
            w_reg[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        char* As = smem + buf * BUF;
        char* Ws = As + BM * STRIDE;
#pragma unroll
        for (int i = 0; i < A_VECS; ++i)
            *reinterpret_cast<uint4*>(As + (r0 + i * RSTEP) * STRIDE + kc * 16) = a_reg[i];
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

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frag_off = (lane & 31) * STRIDE + (lane >> 5) * 16;
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) load_tile((t + 1) * BK);          // global loads fly during the MFMAs
        const char* As = smem + (t & 1) * BUF;
        const char* Ws = As + BM * STRIDE;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
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
        }
        if (t + 1 < nk) store_tile((t + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns row m, register quads own 4 consecutive channels ----
    const int hi = lane >> 5;
    const int col = lane & 31;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m0 + wm0 + b * 32 + col;
        if (m >= p.M) continue;
        const int bi = m / HWo;          // batch index (per-batch row vector, head split)
#pragma unroll
        for (int a = 0; a < TN; ++a) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn0 + a * 32 + 8 * j + 4 * hi;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * j + e];
                if (p.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (p.rowvec) {
                    const float4 rv = *reinterpret_cast<const float4*>(p.rowvec + (size_t)bi * p.rowvec_stride + n);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
                if (p.out_scale != 1.0f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= p.out_scale;
                }
                if (p.res) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(p.res + (size_t)m * p.res_ld + n);
                    v[0] += E::lo(rr.x); v[1] += E::hi(rr.x); v[2] += E::lo(rr.y); v[3] += E::hi(rr.y);
                }
                if (p.act == ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                }
                if (p.act == ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                }
                if (p.mode == OUT_HEADS) {
                    const int which = n / p.hC;
                    const int c = n - which * p.hC;
                    const int h = c / p.hD;
                    const int dd = c - h * p.hD;
                    const int tok = m - bi * HWo;
                    const HeadsDest hdst = p.hd[which];
                    if (hdst.ptr == nullptr) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= hdst.scale;
                    if (hdst.kind == 0) {          // [B, H, L, DP] row-major per head
                        bf16_t* dst = hdst.ptr + ((size_t)(bi * p.hH + h) * hdst.L + tok) * hdst.DP + dd;
                        *reinterpret_cast<uint2*>(dst) = make_uint2(E::pack2(v[0], v[1]), E::pack2(v[2], v[3]));
                    } else {                       // [B, H, DP, L] transposed (keys contiguous)
                        bf16_t* dst = hdst.ptr + ((size_t)(bi * p.hH + h) * hdst.DP + dd) * hdst.L + tok;
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst[(size_t)e * hdst.L] = E::fromf(v[e]);
                    }
                } else if (p.act == ACT_GEGLU) {   // interleaved (value, gate) channel pairs
                    const float o0 = v[0] * gelu_erf_f(v[1]);
                    const float o1 = v[2] * gelu_erf_f(v[3]);
                    bf16_t* dst = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.out_ld + (n >> 1);
                    *reinterpret_cast<uint32_t*>(dst) = E::pack2(o0, o1);
                } else if (p.out_f32) {
                    float* dst = reinterpret_cast<float*>(p.out) + (size_t)m * p.out_ld + n;
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    bf16_t* dst = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.out_ld + n;
                    *reinterpret_cast<uint2*>(dst) = make_uint2(E::pack2(v[0], v[1]), E::pack2(v[2], v[3]));
                }
            }
        }
    }
}

template <bool F16, int BM, int BN, int BK, int WM, int WN>
int launch_cfg(const ConvGemmParams& p, hipStream_t s) {
    constexpr int lds = 2 * (BM + BN) * (BK * 2 + 16);
    static bool attr_set = false;
    auto kern = conv_gemm_kernel<F16, BM, BN, BK, WM, WN>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return imd_set_error("conv_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const long mt = (p.M + BM - 1) / BM, nt = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3((unsigned)(mt * nt)), dim3(256), lds, s, p);
    return imd_check_launch("conv_gemm");
}

}  // namespace

int imd_conv_gemm_choose_cfg(int M, int N) {
    const long b128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (b128 < 192) return 2;
    if ((N % 128) != 0 && (N % 64) == 0 && N < 640) return 1;
    return 0;
}

int imd_launch_conv_gemm(const ConvGemmParams& p, int cfg, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return imd_set_error("conv_gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    if ((p.K % 8) || (p.Cin % 8) || (p.x_pix_stride % 8))
        return imd_set_error("conv_gemm: K (%d), Cin (%d) and pixel stride (%d) must be multiples of 8", p.K, p.Cin, p.x_pix_stride);
    if (p.N % 4) return imd_set_error("conv_gemm: N (%d) must be a multiple of 4", p.N);
    if (p.taps != 1 && p.taps != 9) return imd_set_error("conv_gemm: taps must be 1 or 9 (got %d)", p.taps);
    if (p.K != p.taps * p.Cin) return imd_set_error("conv_gemm: K (%d) != taps*Cin (%d)", p.K, p.taps * p.Cin);
    if (cfg < 0) cfg = imd_conv_gemm_choose_cfg(p.M, p.N);
    if (p.dtype != IMD_DTYPE_BF16 && p.dtype != IMD_DTYPE_F16) return imd_set_error("conv_gemm: unknown dtype %d", p.dtype);
    const bool h = p.dtype == IMD_DTYPE_F16;
    switch (cfg) {
        case 0: return h ? launch_cfg<true, 128, 128, 64, 2, 2>(p, s) : launch_cfg<false, 128, 128, 64, 2, 2>(p, s);
        case 1: return h ? launch_cfg<true, 256, 64, 32, 4, 1>(p, s) : launch_cfg<false, 256, 64, 32, 4, 1>(p, s);
        case 2: return h ? launch_cfg<true, 64, 64, 64, 2, 2>(p, s) : launch_cfg<false, 64, 64, 64, 2, 2>(p, s);
        default: return imd_set_error("conv_gemm: unknown tile config %d", cfg);
    }
}
