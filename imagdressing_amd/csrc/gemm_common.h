// Pieces shared by the implicit-GEMM kernels (conv_gemm.hip, conv_patch.hip): raw buffer loads with
// out-of-range => 0 semantics and the fused 8-channel epilogue.
#pragma once
#include "common.h"
#include "imd_kernels.h"

namespace {

typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4u;

__device__ __forceinline__ uint4 buf_load16(const __amdgpu_buffer_rsrc_t& rs, uint32_t byte_off) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
// same, served from L2 without allocating in the CU's 32 KiB vector L1 (sc1): used for the weight stream so that
// the L1 keeps the activation lines that neighbouring 3x3 taps re-read
__device__ __forceinline__ uint2 buf_load8(const __amdgpu_buffer_rsrc_t& rs, uint32_t byte_off) {
    typedef __attribute__((__vector_size__(2 * sizeof(uint32_t)))) uint32_t v2u;
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)byte_off, 0, 0);
    return make_uint2(v[0], v[1]);
}
__device__ __forceinline__ uint4 buf_load16_nl1(const __amdgpu_buffer_rsrc_t& rs, uint32_t byte_off) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16);
    return make_uint4(v[0], v[1], v[2], v[3]);
}

constexpr uint32_t OOB = 0xffffffffu;   // any offset past num_records reads as zero

// XCD-aware tile order.  Hardware workgroup x of a grid row runs on XCD (x + const) % 8, each XCD with a private 4 MiB L2.
// With the plain order the (row tile, channel tile) pairs that share an operand tile are sprayed over all eight L2s,
// so every activation tile is pulled through the fabric once per channel tile (and every weight tile once per row
// tile).  Remap x so that each XCD owns a CONTIGUOUS range of the tile list, ordered
//   flags & 4: row-tile major    -> all channel tiles of a row tile share one L2 (activations fetched once; the right
//                                   choice when the activation operand is the big one: the 64x64 / 32x32 levels)
//   flags & 8: channel-tile major -> all row tiles of a channel tile share one L2 (weights fetched once: the 8x8 / 16x16
//                                   levels, where K = 11520..23040 weight rows dominate)
// The launcher picks the cheaper of the two from the operand sizes (imd_gemm_pick_order).
__device__ __forceinline__ void xcd_tile_order(int flags, int m_tiles, int n_tiles, int& tile_m, int& tile_n) {
    unsigned w = blockIdx.x;
    if (flags & 12) {
        const unsigned gx = gridDim.x, k = w & 7u, slot = w >> 3, q8 = gx >> 3, r8 = gx & 7u;
        w = (k < r8 ? k * (q8 + 1) : r8 * (q8 + 1) + (k - r8) * q8) + slot;       // bijective also when gx % 8 != 0
    }
    if (flags & 8) { tile_n = (int)(w / (unsigned)m_tiles); tile_m = (int)(w - (unsigned)tile_n * m_tiles); }
    else if ((flags & 4) && (flags & 16)) {
        // grouped order inside the XCD's range: 8 row tiles x all channel tiles, row tile fastest -- the workgroups that
        // are resident together then share 8 activation tiles and a handful of weight tiles instead of sweeping the whole
        // weight matrix once per row tile (which overflows the 4 MiB L2 when N*K*2 does: the GEGLU / qkv projections)
        constexpr unsigned GM = 8;
        const unsigned per_group = GM * (unsigned)n_tiles;
        const unsigned g = w / per_group, r = w - g * per_group;
        const unsigned rows = min(GM, (unsigned)m_tiles - g * GM);
        tile_n = (int)(r / rows);
        tile_m = (int)(g * GM + (r - (unsigned)tile_n * rows));
    }
    else { tile_m = (int)(w / (unsigned)n_tiles); tile_n = (int)(w - (unsigned)tile_m * n_tiles); }
}

// ------------------------------------------------------------------------------------------
// shared epilogue on 8 consecutive channels [n, n+8) of row m (nv = number of valid channels: 4 or 8)
// ------------------------------------------------------------------------------------------
// GroupNorm of the INPUT rows inside a row-resident projection (imd_conv_gemm_params.gn_in_*, round 6): per-channel coefficients of image `b`
// into LDS scratch -- a[k] = gamma[k] rstd[g(k)], sh[k] = beta[k] - mean[g(k)] a[k] -- from the statistic partials of x, folded in EXACTLY the order
// gn_apply_kernel (norm.hip, 320 threads: group = tid % G, every (320 / G)-th partial each, then a serial sum over the 320 / G lanes) folds them, so
// that the fused launch is bit-identical to imd_groupnorm + the projection.  All threads of the workgroup call it (three barriers inside).
// scratch: 2 * 320 + 128 + 2 * K floats.  Returns the coefficient arrays through a / sh.
// Two halves so that the requests go out AHEAD of the kernel's activation loads and come back first (loads return in order): gn_in_request at the very
// top of the kernel, gn_in_coeffs once the activations have been requested too -- the fold then waits for the partials only.
constexpr int GN_IN_MAXP = 13;            // partials per thread requested as one burst (the usual 96 .. 128 partials per image are 10 .. 13 per thread)
template <int K>
struct GnInReq {
    float2 pv[GN_IN_MAXP];
    float gam[(K + 511) / 512], bet[(K + 511) / 512];      // gamma / beta of channels tid, tid + 512, ... (512-thread workgroups)
    bool burst;
};

template <int K>
__device__ __forceinline__ void gn_in_request(const ConvGemmParams& p, int b, GnInReq<K>& r) {
    constexpr int T = 320;
    const int tid = threadIdx.x;
    const int Gn = p.gn_in_groups, nchunks = p.gn_in_nparts;
    const int parts = T / Gn;
    const int g = tid % Gn, part = tid / Gn;
    r.burst = nchunks <= GN_IN_MAXP * parts;
#pragma unroll
    for (int i = 0; i < GN_IN_MAXP; ++i) {
        const int c = part + i * parts;
        r.pv[i] = (r.burst && tid < T && part < parts && c < nchunks) ? *reinterpret_cast<const float2*>(p.gn_in_partial + (((size_t)b * nchunks + c) * Gn + g) * 2)
                                                                          : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < (K + 511) / 512; ++i) {
        const int c = tid + 512 * i;
        r.gam[i] = c < K ? p.gn_in_gamma[c] : 0.f;
        r.bet[i] = c < K ? p.gn_in_beta[c] : 0.f;
    }
}

template <int K>
__device__ __forceinline__ void gn_in_coeffs(const ConvGemmParams& p, int b, const GnInReq<K>& r, float* scratch, const float*& a, const float*& sh) {
    constexpr int T = 320;
    float* red_s = scratch;
    float* red_q = scratch + T;
    float* s_mean = scratch + 2 * T;
    float* s_rstd = s_mean + 64;
    float* s_a = s_rstd + 64;
    float* s_sh = s_a + K;
    const int tid = threadIdx.x;
    const int Gn = p.gn_in_groups, cpg = K / Gn, nchunks = p.gn_in_nparts;
    if (tid < T) {
        const int parts = T / Gn;
        const int g = tid % Gn, part = tid / Gn;
        float S = 0.f, Q = 0.f;
        // the partials were requested TOGETHER (gn_in_request) and are summed in the same ascending order: as a run-time loop the fold is a chain of ~10
        // dependent trips to the L2 that every workgroup of the launch waits out
        if (r.burst) {
#pragma unroll
            for (int i = 0; i < GN_IN_MAXP; ++i) {
                const int c = part + i * parts;
                if (part < parts && c < nchunks) { S += r.pv[i].x; Q += r.pv[i].y; }
            }
        } else if (part < parts) {
            for (int c = part; c < nchunks; c += parts) {
                const float2 v = *reinterpret_cast<const float2*>(p.gn_in_partial + (((size_t)b * nchunks + c) * Gn + g) * 2);
                S += v.x; Q += v.y;
            }
        }
        red_s[tid] = S; red_q[tid] = Q;
    }
    __syncthreads();
    if (tid < Gn) {
        const int parts = T / Gn;
        float St = 0.f, Qt = 0.f;
        for (int k = 0; k < parts; ++k) { St += red_s[k * Gn + tid]; Qt += red_q[k * Gn + tid]; }
        const float n = (float)(p.Hout * p.Wout) * (float)cpg;
        const float mean = St / n;
        const float var = fmaxf(Qt / n - mean * mean, 0.f);
        s_mean[tid] = mean;
        s_rstd[tid] = rsqrtf(var + p.gn_in_eps);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (K + 511) / 512; ++i) {
        const int c = tid + 512 * i;
        if (c < K) {
            const int g = c / cpg;
            const float ga = r.gam[i] * s_rstd[g];
            s_a[c] = ga;
            s_sh[c] = r.bet[i] - s_mean[g] * ga;
        }
    }
    __syncthreads();
    a = s_a; sh = s_sh;
}

// one 8-channel fragment (channels k0 .. k0 + 7) of a row through y = x a + sh (+ SiLU), rounded back to the element type -- gn_apply_kernel's arithmetic
template <bool F16>
__device__ __forceinline__ uint4 gn_in_apply8(const uint4& raw, const float* a, const float* sh, int k0, bool silu) {
    const float4 a0 = *reinterpret_cast<const float4*>(a + k0), a1 = *reinterpret_cast<const float4*>(a + k0 + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(sh + k0), s1 = *reinterpret_cast<const float4*>(sh + k0 + 4);
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    float f[8];
    unpack8<F16>(raw, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float y = f[e] * av[e] + sv[e];
        if (silu) y = silu_f(y);
        f[e] = y;
    }
    return pack8<F16>(f);
}

// can a row-resident kernel with `rows` rows per workgroup and K input channels take p's gn_in_* request?
inline bool gn_in_ok(const ConvGemmParams& p, int K, int rows) {
    if (p.gn_in_partial == nullptr) return true;
    const int HW = p.Hout * p.Wout;
    // (row-major output, no residual: the GN instantiations fix the epilogue's run-time forks at compile time -- Transformer2DModel.proj_in has neither)
    return p.mode == OUT_ROWMAJOR && p.res == nullptr && p.gn_in_gamma != nullptr && p.gn_in_beta != nullptr && p.gn_in_groups > 0 && p.gn_in_groups <= 64 && (K % p.gn_in_groups) == 0 &&
           p.gn_in_nparts > 0 && p.gn_in_nparts <= 4096 && HW > 0 && (HW % rows) == 0 && (p.M % HW) == 0 && p.K == K;
}

// ------------------------------------------------------------------------------------------
// per-column addends of 8 channels [n, n+8): bias[n..] (+ the per-batch vector of batch entry `bi` when given)
__device__ __forceinline__ void load_col_addends(const ConvGemmParams& p, int bi, int n, int nv, float4& a0, float4& a1) {
    a0 = make_float4(0, 0, 0, 0); a1 = a0;
    const bool full = nv == 8;
    if (p.bias) {
        a0 = *reinterpret_cast<const float4*>(p.bias + n);
        if (full) a1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
    }
    if (p.rowvec && bi >= 0) {
        const float* rv = p.rowvec + (size_t)bi * p.rowvec_stride + n;
        const float4 r0 = *reinterpret_cast<const float4*>(rv);
        a0.x += r0.x; a0.y += r0.y; a0.z += r0.z; a0.w += r0.w;
        if (full) { const float4 r1 = *reinterpret_cast<const float4*>(rv + 4); a1.x += r1.x; a1.y += r1.y; a1.z += r1.z; a1.w += r1.w; }
    }
}

// pre0 / pre1: bias (+ per-batch vector) of these 8 columns fetched by the caller ONCE per thread (use_pre false: fetched here).  In the
// row-major chunk loop a thread keeps the same 8 columns for every row it emits, so the tile kernels hoist these loads out
// of the loop (isolated 320 -> 2560 projection: 123 -> 117 us; neutral end to end).  The values travel BY VALUE: handing
// epilogue8 a pointer to a caller-side array demoted that array to scratch memory (48 B/lane) and cost 2.4 % end to end.
// use_rpre / rpre: the 8 residual values of this chunk fetched by the caller AHEAD of the epilogue loop (the one-workgroup-per-CU tile
// kernels have nothing else resident to hide the residual's latency behind: conv_patch3.hip fetches all of a thread's rows at once)
template <bool F16>
__device__ __forceinline__ void epilogue8(const ConvGemmParams& p, float* v, int m, int n, int nv, int HWo, bool use_pre = false,
                                          float4 pre0 = float4{0, 0, 0, 0}, float4 pre1 = float4{0, 0, 0, 0}, bool use_rpre = false,
                                          uint4 rpre = uint4{0, 0, 0, 0}, uint4* pk_out = nullptr) {
    // pk_out (gemm_dma256.hip): a row-major 16-bit result -- 8 channels, or the 4 GEGLU outputs in .x / .y -- is handed back packed instead of
    // being stored; the caller stores it (after a lane transpose that makes the stores whole lines).  Other output forms ignore it.
    using E = El<F16>;
    const int bi = m / HWo;
    float4 b0 = make_float4(0, 0, 0, 0), b1 = b0, r0 = b0, r1 = b0;
    uint4 rr = make_uint4(0, 0, 0, 0);
    const bool full = nv == 8;
    // issue every load first (they are independent), consume afterwards
    if (use_pre) { b0 = pre0; b1 = pre1; }          // (by value: a pointer to a caller-side array would push it into scratch)
    else load_col_addends(p, bi, n, nv, b0, b1);
    if (p.res) {
        if (use_rpre) rr = rpre;
        else {
            const bf16_t* rp = p.res + (size_t)m * p.res_ld + n;
            if (full) rr = *reinterpret_cast<const uint4*>(rp);
            else { const uint2 t = *reinterpret_cast<const uint2*>(rp); rr.x = t.x; rr.y = t.y; }
        }
    }
    v[0] += b0.x + r0.x; v[1] += b0.y + r0.y; v[2] += b0.z + r0.z; v[3] += b0.w + r0.w;
    v[4] += b1.x + r1.x; v[5] += b1.y + r1.y; v[6] += b1.z + r1.z; v[7] += b1.w + r1.w;
    if (p.out_scale != 1.0f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
    }
    if (p.res) {
        float f[8];
        unpack8<F16>(rr, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += f[e];
    }
    if (p.act == ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
    } else if (p.act == ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf_f(v[e]);
    } else if (p.act == ACT_QUICK_GELU) {      // x * sigmoid(1.702 x) = silu(1.702 x) / 1.702
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu_f(1.702f * v[e]) * (1.0f / 1.702f);
    }
    if (p.mode == OUT_HEADS) {
        const int which = n / p.hC;
        const int c = n - which * p.hC;
        const int h = c / p.hD;
        const int dd = c - h * p.hD;
        const int tok = m - bi * HWo;
        const HeadsDest hdst = p.hd[which];
        if (hdst.ptr == nullptr) return;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= hdst.scale;
        if (hdst.kind == 0) {          // [B, H, L, DP] row-major per head
            bf16_t* dst = hdst.ptr + ((size_t)(bi * p.hH + h) * hdst.L + tok) * hdst.DP + dd;
            if (full) *reinterpret_cast<uint4*>(dst) = pack8<F16>(v);
            else *reinterpret_cast<uint2*>(dst) = make_uint2(E::pack2(v[0], v[1]), E::pack2(v[2], v[3]));
        } else {                       // [B, H, DP, L] transposed (keys contiguous)
            bf16_t* dst = hdst.ptr + ((size_t)(bi * p.hH + h) * hdst.DP + dd) * hdst.L + tok;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < nv) dst[(size_t)e * hdst.L] = E::fromf(v[e]);
        }
    } else if (p.act == ACT_GEGLU) {   // interleaved (value, gate) channel pairs -> nv/2 outputs
        bf16_t* dst = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.out_ld + (n >> 1);
        const uint32_t o0 = E::pack2(v[0] * gelu_erf_f(v[1]), v[2] * gelu_erf_f(v[3]));
        if (full) {
            const uint32_t o1 = E::pack2(v[4] * gelu_erf_f(v[5]), v[6] * gelu_erf_f(v[7]));
            if (pk_out) { pk_out->x = o0; pk_out->y = o1; return; }
            *reinterpret_cast<uint2*>(dst) = make_uint2(o0, o1);
        } else {
            *reinterpret_cast<uint32_t*>(dst) = o0;
        }
    } else if (p.out_f32) {
        float* dst = reinterpret_cast<float*>(p.out) + (size_t)m * p.out_ld + n;
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        if (full) *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        bf16_t* dst = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.out_ld + n;
        if (full && pk_out) { *pk_out = pack8<F16>(v); return; }
        if (full) *reinterpret_cast<uint4*>(dst) = pack8<F16>(v);
        else *reinterpret_cast<uint2*>(dst) = make_uint2(E::pack2(v[0], v[1]), E::pack2(v[2], v[3]));
    }
}

// ------------------------------------------------------------------------------------------
// split-K without a second launch.  Every workgroup of a K slice writes its fp32 partial tile to its slab; the LAST workgroup
// to arrive at the tile's counter sums the slabs of the tile in slice order 0 .. S-1 -- the order the finish kernel uses, so
// the result is bit-identical whoever arrives last -- and runs the epilogue.
// Coherence: the slices of a tile may run on different XCDs, whose L2s only meet at the memory side.  A release / acquire
// fence pair at device scope would do (`__threadfence()`), but on this part it writes back and invalidates the WHOLE L2 of the
// XCD in every workgroup: measured 659 -> 973 ms end to end.  Instead the slab traffic itself is made coherent: partial tiles
// are stored with sc0 sc1 (write-through to the memory side), the arrival counter is a device-scope atomic issued after the
// workgroup's stores have been acknowledged (vmcnt(0) + barrier), and the last arrival reads the slabs with sc0 sc1 loads
// (served from the memory side, never from a stale L1 / L2 line).  Nothing else of the kernel's traffic is touched.
// ------------------------------------------------------------------------------------------
constexpr int AUX_SYSTEM = 17;         // buffer cache policy bits sc0 (1) | sc1 (16)

__device__ __forceinline__ void slab_store8(float* slab_base, size_t elem_off, const float4& v0, const float4& v1, bool full, bool coherent) {
    if (coherent) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab_base, 0, 0x80000000u, 0x00020000);
        const v4u a = {__float_as_uint(v0.x), __float_as_uint(v0.y), __float_as_uint(v0.z), __float_as_uint(v0.w)};
        __builtin_amdgcn_raw_buffer_store_b128(a, rs, (int)(elem_off * 4), 0, AUX_SYSTEM);
        if (full) {
            const v4u b = {__float_as_uint(v1.x), __float_as_uint(v1.y), __float_as_uint(v1.z), __float_as_uint(v1.w)};
            __builtin_amdgcn_raw_buffer_store_b128(b, rs, (int)(elem_off * 4 + 16), 0, AUX_SYSTEM);
        }
    } else {
        float* dst = slab_base + elem_off;
        *reinterpret_cast<float4*>(dst) = v0;
        if (full) *reinterpret_cast<float4*>(dst + 4) = v1;
    }
}

// Returns true in exactly one workgroup per tile; the counter is left at zero for the next launch.
__device__ __forceinline__ bool splitk_last_arrival(int* counters, int tile_id, int split_k, int tid) {
    __shared__ int s_ticket;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's write-through slab stores have been acknowledged
    __syncthreads();                                       // ... everybody's in the workgroup
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = s_ticket == split_k - 1;
    if (last && tid == 0) __hip_atomic_store(counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return last;
}

// sum of the K slices of 8 consecutive channels [n, n+8) of row m, in slice order (what splitk_finish_kernel computes), read
// from the memory side
__device__ __forceinline__ void splitk_sum8(const ConvGemmParams& p, int m, int n, int nv, float* v) {
    const size_t slab = (size_t)p.M * p.N;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    const int off = (int)(((size_t)m * p.N + n) * 4);
    for (int s = 0; s < p.split_k; ++s) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.splitk_ws + s * slab, 0, 0x80000000u, 0x00020000);
        const v4u a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX_SYSTEM);
        v[0] += __uint_as_float(a[0]); v[1] += __uint_as_float(a[1]); v[2] += __uint_as_float(a[2]); v[3] += __uint_as_float(a[3]);
        if (nv == 8) {
            const v4u b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, AUX_SYSTEM);
            v[4] += __uint_as_float(b[0]); v[5] += __uint_as_float(b[1]); v[6] += __uint_as_float(b[2]); v[7] += __uint_as_float(b[3]);
        }
    }
}

}  // namespace
