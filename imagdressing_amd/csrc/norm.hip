// GroupNorm(+SiLU) and LayerNorm for NHWC / token-major bf16 activations (HBM-bound kernels).
//
// Reference arithmetic (diffusers==0.24.0, un-vendored): ResnetBlock2D norm1/norm2 + SiLU,
// Transformer2DModel.norm (eps 1e-6, no activation), conv_norm_out + SiLU; BasicTransformerBlock
// norm1/2/3 and the LayerNorms of /root/reference/adapter/resampler.py:16,43-44,199.
//
// GroupNorm runs as two launches over the same tensor: (1) per-(batch, pixel-chunk, group)
// fp32 partial sums -- every thread owns a FIXED 8-channel vector column so its accumulators stay
// in registers and all loads are 16-byte coalesced -- written to a small workspace (deterministic,
// no atomics); (2) normalise + affine (+ SiLU), each block first folding the partials of its batch
// entry.  Statistics are fp32.
#include "common.h"
#include "imd_kernels.h"

namespace {

constexpr int GN_THREADS = 320;          // 5 waves: divides evenly for C/8 = 40, 80, 160, 320

// pixels per block ("chunk"): sized so that the grid has ~1000 blocks even for the 8x8 / 16x16 levels
// (a fixed 64-pixel chunk left those launches with 8..32 blocks: pure latency), never below two passes
// of the block's pixel lanes.
__host__ __device__ inline int gn_pix_per_chunk(int B, int HW, int C) {
    const int vpp = C / 8;
    const int cols = vpp < GN_THREADS ? vpp : GN_THREADS;
    const int plan = GN_THREADS / cols;
#ifndef GN_TARGET_BLOCKS
#define GN_TARGET_BLOCKS 512      // A/B on one box (profiles/r3z_gn_ab.txt): 2048 -> +1.6 %, 1024 -> 0, 512 -> -0.5 %, 128 -> +1.3 % of the step
#endif
#ifndef GN_MAX_PPC
#define GN_MAX_PPC 128
#endif
    int ppc = (int)(((long)B * HW + GN_TARGET_BLOCKS - 1) / GN_TARGET_BLOCKS);
    ppc = (ppc + plan - 1) / plan * plan;
    if (ppc < 2 * plan) ppc = 2 * plan;
    if (ppc > GN_MAX_PPC) ppc = GN_MAX_PPC;
    return ppc;
}
__host__ __device__ inline int gn_chunks(int B, int HW, int C) {
    const int ppc = gn_pix_per_chunk(B, HW, C);
    return (HW + ppc - 1) / ppc;
}

// One thread: vector column `vec` (channels 8*vec .. 8*vec+7), pixel lanes interleaved.
template <bool F16>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const GroupNormParams p) {
    __shared__ float red[GN_THREADS][4];
    const int vpp = p.C / 8;                       // vectors per pixel
    const int tid = threadIdx.x;
    const int cpg = p.C / p.G;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int ppc = gn_pix_per_chunk(p.B, p.HW, p.C);
    const int pix0 = chunk * ppc;
    const int pix1 = min(p.HW, pix0 + ppc);
    const int nchunks = gn_chunks(p.B, p.HW, p.C);

    // columns are processed in passes of GN_THREADS / ppb ... keep it simple: loop over
    // column blocks of width `cols` = min(vpp, GN_THREADS); pixel lanes = GN_THREADS / cols.
    const int cols = min(vpp, GN_THREADS);
    const int plan = GN_THREADS / cols;            // pixel lanes
    const int my_col = tid % cols, my_pl = tid / cols;
    const bool active = my_pl < plan;

    for (int cbase = 0; cbase < vpp; cbase += cols) {
        const int vec = cbase + my_col;
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
        const int c0 = vec * 8;
        const int g0 = c0 / cpg;
        const int split = min(8, (g0 + 1) * cpg - c0);   // channels [0,split) -> g0, rest -> g0+1
        if (active && vec < vpp) {
            const bf16_t* xb = p.x + (size_t)b * p.HW * p.x_ld + c0;
            // 4 independent 16-byte loads in flight per thread (memory-level parallelism)
            for (int pix = pix0 + my_pl; pix < pix1; pix += 4 * plan) {
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int px = pix + u * plan;
                    v[u] = (px < pix1) ? *reinterpret_cast<const uint4*>(xb + (size_t)px * p.x_ld) : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float f[8];
                    unpack8<F16>(v[u], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (e < split) { s0 += f[e]; q0 += f[e] * f[e]; }
                        else           { s1 += f[e]; q1 += f[e] * f[e]; }
                    }
                }
            }
        }
        red[tid][0] = s0; red[tid][1] = q0; red[tid][2] = s1; red[tid][3] = q1;
        __syncthreads();
        // thread g (< G) folds every (pixel lane, column) entry that touches group g
        if (tid < p.G) {
            const int g = tid;
            const int cfirst = g * cpg, clast = (g + 1) * cpg - 1;
            int vlo = cfirst / 8, vhi = clast / 8;
            vlo = max(vlo, cbase); vhi = min(vhi, min(vpp, cbase + cols) - 1);
            float S = 0.f, Q = 0.f;
            for (int v = vlo; v <= vhi; ++v) {
                const int vg0 = (v * 8) / cpg;
                for (int pl = 0; pl < plan; ++pl) {
                    const float* e = red[pl * cols + (v - cbase)];
                    if (vg0 == g) { S += e[0]; Q += e[1]; }
                    else if (vg0 + 1 == g) { S += e[2]; Q += e[3]; }
                }
            }
            float* dst = p.partial + (((size_t)b * nchunks + chunk) * p.G + g) * 2;
            if (cbase == 0) { dst[0] = S; dst[1] = Q; }
            else { dst[0] += S; dst[1] += Q; }
        }
        __syncthreads();
    }
}

// gn_stats_kernel over out = cat([a, b (+ b_add)], channel), WRITING out on the way (round 6): the skip concatenation of an up block and the
// statistics pass of the ResnetBlock2D.norm1 that follows it were two sweeps over the same tensor.  Same chunking, same thread -> (vector
// column, pixel lane) map and same summation order as gn_stats_kernel on the finished tensor, and the sums are taken from the ROUNDED 16-bit
// values that are stored: partials bit-identical to the two-launch form.  b may hold B / k images (one skip tensor for both CFG halves).
struct ConcatSrc { const bf16_t* a; int Ca; const bf16_t* b; int Cb; const bf16_t* b_add; int b_B; };

template <bool F16>
__global__ __launch_bounds__(GN_THREADS) void concat2_stats_kernel(const GroupNormParams p, const ConcatSrc q) {
    __shared__ float red[GN_THREADS][4];
    const int vpp = p.C / 8;
    const int tid = threadIdx.x;
    const int cpg = p.C / p.G;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int ppc = gn_pix_per_chunk(p.B, p.HW, p.C);
    const int pix0 = chunk * ppc;
    const int pix1 = min(p.HW, pix0 + ppc);
    const int nchunks = gn_chunks(p.B, p.HW, p.C);
    const int cols = min(vpp, GN_THREADS);
    const int plan = GN_THREADS / cols;
    const int my_col = tid % cols, my_pl = tid / cols;
    const bool active = my_pl < plan;

    for (int cbase = 0; cbase < vpp; cbase += cols) {
        const int vec = cbase + my_col;
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
        const int c0 = vec * 8;
        const int g0 = c0 / cpg;
        const int split = min(8, (g0 + 1) * cpg - c0);
        if (active && vec < vpp) {
            const bool from_a = c0 < q.Ca;
            const int cs = from_a ? c0 : c0 - q.Ca;
            const int ld = from_a ? q.Ca : q.Cb;
            const bf16_t* src = from_a ? q.a + (size_t)b * p.HW * q.Ca + cs : q.b + (size_t)(b % q.b_B) * p.HW * q.Cb + cs;
            const bf16_t* add = (!from_a && q.b_add) ? q.b_add + (size_t)b * p.HW * q.Cb + cs : nullptr;
            bf16_t* ob = p.y + (size_t)b * p.HW * p.C + c0;
            for (int pix = pix0 + my_pl; pix < pix1; pix += 4 * plan) {
                uint4 v[4], w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int px = pix + u * plan;
                    v[u] = (px < pix1) ? *reinterpret_cast<const uint4*>(src + (size_t)px * ld) : make_uint4(0, 0, 0, 0);
                    if (add) w[u] = (px < pix1) ? *reinterpret_cast<const uint4*>(add + (size_t)px * ld) : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int px = pix + u * plan;
                    float f[8];
                    if (add) {       // (the ControlNet residual: summed in fp32, rounded once -- concat2_kernel's arithmetic)
                        float x[8], y[8];
                        unpack8<F16>(v[u], x);
                        unpack8<F16>(w[u], y);
                        v[u] = make_uint4(El<F16>::pack2(x[0] + y[0], x[1] + y[1]), El<F16>::pack2(x[2] + y[2], x[3] + y[3]),
                                          El<F16>::pack2(x[4] + y[4], x[5] + y[5]), El<F16>::pack2(x[6] + y[6], x[7] + y[7]));
                    }
                    if (px < pix1) *reinterpret_cast<uint4*>(ob + (size_t)px * p.C) = v[u];
                    unpack8<F16>(v[u], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (e < split) { s0 += f[e]; q0 += f[e] * f[e]; }
                        else           { s1 += f[e]; q1 += f[e] * f[e]; }
                    }
                }
            }
        }
        red[tid][0] = s0; red[tid][1] = q0; red[tid][2] = s1; red[tid][3] = q1;
        __syncthreads();
        if (tid < p.G) {
            const int g = tid;
            const int cfirst = g * cpg, clast = (g + 1) * cpg - 1;
            int vlo = cfirst / 8, vhi = clast / 8;
            vlo = max(vlo, cbase); vhi = min(vhi, min(vpp, cbase + cols) - 1);
            float S = 0.f, Q = 0.f;
            for (int v = vlo; v <= vhi; ++v) {
                const int vg0 = (v * 8) / cpg;
                for (int pl = 0; pl < plan; ++pl) {
                    const float* e = red[pl * cols + (v - cbase)];
                    if (vg0 == g) { S += e[0]; Q += e[1]; }
                    else if (vg0 + 1 == g) { S += e[2]; Q += e[3]; }
                }
            }
            float* dst = p.partial + (((size_t)b * nchunks + chunk) * p.G + g) * 2;
            if (cbase == 0) { dst[0] = S; dst[1] = Q; }
            else { dst[0] += S; dst[1] += Q; }
        }
        __syncthreads();
    }
}

template <bool F16>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const GroupNormParams p) {
    __shared__ float s_mean[64], s_rstd[64];
    __shared__ float red_s[GN_THREADS], red_q[GN_THREADS];
    const int vpp = p.C / 8;
    const int tid = threadIdx.x;
    const int cpg = p.C / p.G;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int nchunks = p.nparts > 0 ? p.nparts : gn_chunks(p.B, p.HW, p.C);       // (nparts: partials written by the producer of x)
    {   // fold this batch entry's per-chunk partials: ALL threads take part (group = tid % G, every parts-th
        // chunk each) so the ~100 partial loads of a block are independent and in flight together
        const int parts = GN_THREADS / p.G;
        const int g = tid % p.G, part = tid / p.G;
        float S = 0.f, Q = 0.f;
        // (round 6) a thread's partials are requested TOGETHER and summed in the same ascending order afterwards (gemm_common.h::gn_in_coeffs does the
        // same): as a run-time loop the fold is a chain of ~10 dependent trips to the L2 in front of every block's streaming phase
        constexpr int MAXP = 13;
        if (nchunks <= MAXP * parts) {
            float2 pv[MAXP];
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const int c = part + i * parts;
                pv[i] = (part < parts && c < nchunks) ? *reinterpret_cast<const float2*>(p.partial + (((size_t)b * nchunks + c) * p.G + g) * 2) : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const int c = part + i * parts;
                if (part < parts && c < nchunks) { S += pv[i].x; Q += pv[i].y; }
            }
        } else if (part < parts) {
            for (int c = part; c < nchunks; c += parts) {
                const float2 v = *reinterpret_cast<const float2*>(p.partial + (((size_t)b * nchunks + c) * p.G + g) * 2);
                S += v.x; Q += v.y;
            }
        }
        red_s[tid] = S; red_q[tid] = Q;
        __syncthreads();
        if (tid < p.G) {
            float St = 0.f, Qt = 0.f;
            for (int k = 0; k < parts; ++k) { St += red_s[k * p.G + tid]; Qt += red_q[k * p.G + tid]; }
            const float n = (float)p.HW * (float)cpg;
            const float mean = St / n;
            const float var = fmaxf(Qt / n - mean * mean, 0.f);
            s_mean[tid] = mean;
            s_rstd[tid] = rsqrtf(var + p.eps);
        }
    }
    __syncthreads();
    const int ppc = gn_pix_per_chunk(p.B, p.HW, p.C);
    const int pix0 = chunk * ppc;
    const int pix1 = min(p.HW, pix0 + ppc);
    const int cols = min(vpp, GN_THREADS);
    const int plan = GN_THREADS / cols;
    const int my_col = tid % cols, my_pl = tid / cols;
    if (my_pl >= plan) return;
    for (int cbase = 0; cbase < vpp; cbase += cols) {
        const int vec = cbase + my_col;
        if (vec >= vpp) continue;
        const int c0 = vec * 8;
        float a[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c0 + e) / cpg;
            const float ga = p.gamma[c0 + e] * s_rstd[g];
            a[e] = ga;
            sh[e] = p.beta[c0 + e] - s_mean[g] * ga;
        }
        const bf16_t* xb = p.x + (size_t)b * p.HW * p.x_ld + c0;
        bf16_t* yb = p.y + (size_t)b * p.HW * p.y_ld + c0;
        for (int pix = pix0 + my_pl; pix < pix1; pix += 4 * plan) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int px = pix + u * plan;
                v[u] = (px < pix1) ? *reinterpret_cast<const uint4*>(xb + (size_t)px * p.x_ld) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int px = pix + u * plan;
                if (px >= pix1) continue;
                float f[8];
                unpack8<F16>(v[u], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = f[e] * a[e] + sh[e];
                    if (p.silu) y = silu_f(y);
                    f[e] = y;
                }
                *reinterpret_cast<uint4*>(yb + (size_t)px * p.y_ld) = pack8<F16>(f);
            }
        }
    }
}

// Fold the partials of batch entry blockIdx.x and write the per-(batch, channel) affine form of the normalisation,
// y = x * a[b][c] + sh[b][c]: consumed by the fused GroupNorm prologue of conv_patch.hip.
__global__ __launch_bounds__(GN_THREADS) void gn_coeffs_kernel(const GroupNormParams p, float* __restrict__ ca, float* __restrict__ cb) {
    __shared__ float s_mean[64], s_rstd[64];
    __shared__ float red_s[GN_THREADS], red_q[GN_THREADS];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int cpg = p.C / p.G;
    const int nchunks = p.nparts > 0 ? p.nparts : gn_chunks(p.B, p.HW, p.C);       // (nparts: partials written by the producer of x, as in gn_apply_kernel)
    const int parts = GN_THREADS / p.G;
    const int g = tid % p.G, part = tid / p.G;
    float S = 0.f, Q = 0.f;
    if (part < parts) {
        for (int c = part; c < nchunks; c += parts) {
            const float2 v = *reinterpret_cast<const float2*>(p.partial + (((size_t)b * nchunks + c) * p.G + g) * 2);
            S += v.x; Q += v.y;
        }
    }
    red_s[tid] = S; red_q[tid] = Q;
    __syncthreads();
    if (tid < p.G) {
        float St = 0.f, Qt = 0.f;
        for (int k = 0; k < parts; ++k) { St += red_s[k * p.G + tid]; Qt += red_q[k * p.G + tid]; }
        const float n = (float)p.HW * (float)cpg;
        const float mean = St / n;
        const float var = fmaxf(Qt / n - mean * mean, 0.f);
        s_mean[tid] = mean;
        s_rstd[tid] = rsqrtf(var + p.eps);
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += GN_THREADS) {
        const int gg = c / cpg;
        const float ga = p.gamma[c] * s_rstd[gg];
        ca[(size_t)b * p.C + c] = ga;
        cb[(size_t)b * p.C + c] = p.beta[c] - s_mean[gg] * ga;
    }
}

// Normalise with precomputed per-(batch, channel) coefficients (gn_coeffs_kernel): used for very large feature maps
// (the VAE's 256x256 / 512x512 levels), where folding thousands of per-chunk partials in every apply block would cost
// more than the normalisation itself.
template <bool F16>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_coeffs_kernel(const GroupNormParams p, const float* __restrict__ ca, const float* __restrict__ cb) {
    const int vpp = p.C / 8;
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int ppc = gn_pix_per_chunk(p.B, p.HW, p.C);
    const int pix0 = chunk * ppc;
    const int pix1 = min(p.HW, pix0 + ppc);
    const int cols = min(vpp, GN_THREADS);
    const int plan = GN_THREADS / cols;
    const int my_col = tid % cols, my_pl = tid / cols;
    if (my_pl >= plan) return;
    for (int cbase = 0; cbase < vpp; cbase += cols) {
        const int vec = cbase + my_col;
        if (vec >= vpp) continue;
        const int c0 = vec * 8;
        float a[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = ca[(size_t)b * p.C + c0 + e]; sh[e] = cb[(size_t)b * p.C + c0 + e]; }
        const bf16_t* xb = p.x + (size_t)b * p.HW * p.x_ld + c0;
        bf16_t* yb = p.y + (size_t)b * p.HW * p.y_ld + c0;
        for (int pix = pix0 + my_pl; pix < pix1; pix += 4 * plan) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int px = pix + u * plan;
                v[u] = (px < pix1) ? *reinterpret_cast<const uint4*>(xb + (size_t)px * p.x_ld) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int px = pix + u * plan;
                if (px >= pix1) continue;
                float f[8];
                unpack8<F16>(v[u], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = f[e] * a[e] + sh[e];
                    if (p.silu) y = silu_f(y);
                    f[e] = y;
                }
                *reinterpret_cast<uint4*>(yb + (size_t)px * p.y_ld) = pack8<F16>(f);
            }
        }
    }
}

// One wave per row; C <= 8 * 64 * LN_MAXV.
constexpr int LN_MAXV = 4;
template <bool F16>
__global__ __launch_bounds__(256) void layernorm_kernel(const LayerNormParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int nv = p.C / 8;
    float f[LN_MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int v = lane + i * 64;
        if (v < nv) {
            const uint4 x = *reinterpret_cast<const uint4*>(p.x + (size_t)row * p.x_ld + v * 8);
            unpack8<F16>(x, f[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += f[i][e];
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)p.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int v = lane + i * 64;
        if (v < nv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = rsqrtf(q / (float)p.C + p.eps);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int v = lane + i * 64;
        if (v < nv) {
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = (f[i][e] - mean) * rstd * p.gamma[v * 8 + e] + p.beta[v * 8 + e];
            *reinterpret_cast<uint4*>(p.y + (size_t)row * p.y_ld + v * 8) = pack8<F16>(y);
        }
    }
}

// Row softmax, fp32 -> 16 bit: one workgroup per row; the row lives in registers (NV values per thread, loops fully
// unrolled so that the array is never indexed dynamically).
template <bool F16, int NV>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s_in, int s_ld, bf16_t* __restrict__ p_out, int p_ld, int cols) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = s_in + (size_t)blockIdx.x * s_ld;
    bf16_t* out = p_out + (size_t)blockIdx.x * p_ld;
    float v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tid + i * 256;
        v[i] = c < cols ? row[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = __builtin_amdgcn_exp2f((v[i] - mx) * LOG2E);       // exp2(-inf) = 0 for the columns past the row
        sum += v[i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tid + i * 256;
        if (c < cols) out[c] = El<F16>::fromf(v[i] * inv);
    }
}

template <bool F16>
void launch_softmax(const float* s_in, int s_ld, bf16_t* p_out, int p_ld, int rows, int cols, hipStream_t s) {
    const int nv = (cols + 255) / 256;
    if (nv <= 4) hipLaunchKernelGGL((softmax_rows_kernel<F16, 4>), dim3(rows), dim3(256), 0, s, s_in, s_ld, p_out, p_ld, cols);
    else if (nv <= 16) hipLaunchKernelGGL((softmax_rows_kernel<F16, 16>), dim3(rows), dim3(256), 0, s, s_in, s_ld, p_out, p_ld, cols);
    else if (nv <= 32) hipLaunchKernelGGL((softmax_rows_kernel<F16, 32>), dim3(rows), dim3(256), 0, s, s_in, s_ld, p_out, p_ld, cols);
    else hipLaunchKernelGGL((softmax_rows_kernel<F16, 64>), dim3(rows), dim3(256), 0, s, s_in, s_ld, p_out, p_ld, cols);
}

}  // namespace

int imd_launch_softmax_rows(const float* s_in, int s_ld, bf16_t* p_out, int p_ld, int rows, int cols, int dtype, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return imd_set_error("softmax_rows: empty matrix");
    if (cols > 256 * 64) return imd_set_error("softmax_rows: at most %d columns (got %d)", 256 * 64, cols);
    if (s_ld < cols || p_ld < cols) return imd_set_error("softmax_rows: row strides shorter than the row");
    if (dtype != IMD_DTYPE_BF16 && dtype != IMD_DTYPE_F16) return imd_set_error("softmax_rows: unknown dtype %d", dtype);
    if (dtype == IMD_DTYPE_F16) launch_softmax<true>(s_in, s_ld, p_out, p_ld, rows, cols, s);
    else launch_softmax<false>(s_in, s_ld, p_out, p_ld, rows, cols, s);
    return imd_check_launch("softmax_rows");
}

constexpr int GN_TWO_LEVEL_CHUNKS = 256;   // more per-chunk partials than this: fold once (coefficients), then apply

int imd_groupnorm_workspace_floats(int B, int HW, int C, int G) {
    return B * gn_chunks(B, HW, C) * G * 2 + 2 * B * C;
}

static int gn_validate(const GroupNormParams& p) {
    if (p.B <= 0 || p.HW <= 0 || p.C <= 0) return imd_set_error("groupnorm: empty tensor");
    if (p.C % 8 || p.C % p.G || p.G > 64) return imd_set_error("groupnorm: C (%d) must be a multiple of 8 and of G (%d <= 64)", p.C, p.G);
    if ((p.C / p.G) < 4 || ((p.C / p.G) < 8 && (8 % (p.C / p.G)) != 0))
        return imd_set_error("groupnorm: channels per group (%d) must be 4 or >= 8 (an 8-channel vector may span two groups)", p.C / p.G);
    if (p.x_ld % 8 || p.y_ld % 8) return imd_set_error("groupnorm: pixel strides must be multiples of 8");
    if (p.dtype != IMD_DTYPE_BF16 && p.dtype != IMD_DTYPE_F16) return imd_set_error("groupnorm: unknown dtype %d", p.dtype);
    return 0;
}

int imd_groupnorm_parts_of(int B, int HW, int C) {
    if (B <= 0 || HW <= 0 || C <= 0 || C % 8) return 0;
    return gn_chunks(B, HW, C);
}

int imd_launch_concat2_gn_stats(const bf16_t* a, int Ca, const bf16_t* b, int Cb, const bf16_t* b_add, bf16_t* out, int B, int HW, int b_B, int G,
                                float* partial, int dtype, hipStream_t s) {
    if (Ca <= 0 || Cb <= 0 || Ca % 8 || Cb % 8) return imd_set_error("concat2 + statistics: channel counts must be positive multiples of 8 (got %d + %d)", Ca, Cb);
    if (b_B <= 0 || B % b_B) return imd_set_error("concat2 + statistics: b holds %d images, which must divide B = %d", b_B, B);
    GroupNormParams p{};
    p.y = out; p.partial = partial;
    p.B = B; p.HW = HW; p.C = Ca + Cb; p.G = G; p.x_ld = p.y_ld = Ca + Cb; p.dtype = dtype;
    if (G <= 0) return imd_set_error("concat2 + statistics: G must be positive");
    int rc = gn_validate(p);
    if (rc) return rc;
    const ConcatSrc q{a, Ca, b, Cb, b_add, b_B};
    dim3 grid(gn_chunks(B, HW, p.C), B);
    if (dtype == IMD_DTYPE_F16) hipLaunchKernelGGL(concat2_stats_kernel<true>, grid, dim3(GN_THREADS), 0, s, p, q);
    else hipLaunchKernelGGL(concat2_stats_kernel<false>, grid, dim3(GN_THREADS), 0, s, p, q);
    return imd_check_launch("concat2 + groupnorm statistics");
}

int imd_launch_groupnorm_coeffs(const GroupNormParams& p, float* ca, float* cb, hipStream_t s) {
    int rc = gn_validate(p);
    if (rc) return rc;
    if (p.nparts > 0) {              // statistics came with the tensor (convolution epilogue / finish launch): fold them, same order as gn_apply_kernel
        if (p.nparts > 4096) return imd_set_error("groupnorm coeffs: nparts %d is implausible", p.nparts);
        hipLaunchKernelGGL(gn_coeffs_kernel, dim3(p.B), dim3(GN_THREADS), 0, s, p, ca, cb);
        return imd_check_launch("groupnorm coeffs (producer statistics)");
    }
    dim3 grid(gn_chunks(p.B, p.HW, p.C), p.B);
    if (p.dtype == IMD_DTYPE_F16) hipLaunchKernelGGL(gn_stats_kernel<true>, grid, dim3(GN_THREADS), 0, s, p);
    else hipLaunchKernelGGL(gn_stats_kernel<false>, grid, dim3(GN_THREADS), 0, s, p);
    rc = imd_check_launch("groupnorm stats");
    if (rc) return rc;
    hipLaunchKernelGGL(gn_coeffs_kernel, dim3(p.B), dim3(GN_THREADS), 0, s, p, ca, cb);
    return imd_check_launch("groupnorm coeffs");
}

int imd_launch_groupnorm(const GroupNormParams& p, hipStream_t s) {
    int rc0 = gn_validate(p);
    if (rc0) return rc0;
    const int chunks = gn_chunks(p.B, p.HW, p.C);
    dim3 grid(chunks, p.B);
    const bool h = p.dtype == IMD_DTYPE_F16;
    if (p.nparts > 0) {              // statistics came with the tensor (convolution epilogue): normalise only
        if (p.nparts > 4096) return imd_set_error("groupnorm: nparts %d is implausible", p.nparts);
        if (h) hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(GN_THREADS), 0, s, p);
        else hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(GN_THREADS), 0, s, p);
        return imd_check_launch("groupnorm apply (producer statistics)");
    }
    if (chunks > GN_TWO_LEVEL_CHUNKS) {
        float* ca = p.partial + (size_t)p.B * chunks * p.G * 2;
        float* cb = ca + (size_t)p.B * p.C;
        int rc = imd_launch_groupnorm_coeffs(p, ca, cb, s);
        if (rc) return rc;
        if (h) hipLaunchKernelGGL(gn_apply_coeffs_kernel<true>, grid, dim3(GN_THREADS), 0, s, p, ca, cb);
        else hipLaunchKernelGGL(gn_apply_coeffs_kernel<false>, grid, dim3(GN_THREADS), 0, s, p, ca, cb);
        return imd_check_launch("groupnorm apply (coefficients)");
    }
    if (h) hipLaunchKernelGGL(gn_stats_kernel<true>, grid, dim3(GN_THREADS), 0, s, p);
    else hipLaunchKernelGGL(gn_stats_kernel<false>, grid, dim3(GN_THREADS), 0, s, p);
    int rc = imd_check_launch("groupnorm stats");
    if (rc) return rc;
    if (h) hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(GN_THREADS), 0, s, p);
    else hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(GN_THREADS), 0, s, p);
    return imd_check_launch("groupnorm apply");
}

int imd_launch_layernorm(const LayerNormParams& p, hipStream_t s) {
    if (p.rows <= 0) return imd_set_error("layernorm: no rows");
    if (p.C % 8 || p.C > 8 * 64 * LN_MAXV) return imd_set_error("layernorm: C (%d) must be a multiple of 8 and <= %d", p.C, 8 * 64 * LN_MAXV);
    if (p.x_ld % 8 || p.y_ld % 8) return imd_set_error("layernorm: row strides must be multiples of 8");
    if (p.dtype != IMD_DTYPE_BF16 && p.dtype != IMD_DTYPE_F16) return imd_set_error("layernorm: unknown dtype %d", p.dtype);
    if (p.dtype == IMD_DTYPE_F16) hipLaunchKernelGGL(layernorm_kernel<true>, dim3((p.rows + 3) / 4), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(layernorm_kernel<false>, dim3((p.rows + 3) / 4), dim3(256), 0, s, p);
    return imd_check_launch("layernorm");
}
