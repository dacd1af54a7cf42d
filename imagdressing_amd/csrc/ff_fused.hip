// Fused feed-forward of a BasicTransformerBlock on the 64x64 level (C = 320, inner = 1280):
//
//   out[m, :] = x[m, :] + b2 + W2 . geglu( W1 . LN(x[m, :]) + b1 )          geglu(v, g) = v * gelu(g)
//
// i.e. norm3 -> ff.net[0] (GEGLU) -> ff.net[2] -> + residual of diffusers==0.24.0 BasicTransformerBlock (un-vendored; call
// sites /root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline.py:499,511) as ONE launch.  As four launches
// (layernorm, 320 -> 2560 GEGLU projection, 1280 -> 320 projection + residual) the block's feed-forward takes ~188 us of
// which the 84 MB GEGLU tensor written and read back, the 21 MB normalised tensor and three lock-step fill / drain phases are
// pure overhead; here the 128 x 1280 activations of a workgroup never leave the CU.
//
// Structure (same row-resident scheme as row_linear.hip):
//   * a workgroup (8 waves) owns 128 token rows; wave (rb, hh) holds the 32 rows of block rb -- all 320 k -- in 80 VGPRs in
//     MFMA B-operand layout, normalises them in place (LayerNorm affine folded into W1 / b1 by the host), and owns the
//     accumulators of output channels [160 hh, 160 hh + 160) of those rows (80 VGPRs);
//   * the inner dimension is walked in 40 chunks of 32 channels.  Chunk c needs 64 rows of W1 (32 value + 32 gate rows,
//     40 KB) and a 320 x 32 slab of W2 (20 KB); both stream through LDS by DMA (2-slot rings, one barrier per chunk, source-side
//     XOR swizzles for conflict-free ds_read_b128).  The host packs W1 so that value_j and gate_j of a 16-channel block land in
//     the SAME lane of the MFMA result (rows i and i + 8), and W2's columns so that the lane's 8 GEGLU outputs are, in register
//     order, exactly one 16-byte B-operand fragment of the second GEMM: h never needs a cross-lane shuffle;
//   * per chunk the two waves (rb, 0) and (rb, 1) each compute ONE 16-channel block of h for their 32 rows (20 MFMAs + the GELU
//     on 8 values per lane), swap the packed fragments through 1 KB of LDS, and both run the second GEMM for their own 160
//     output channels over the chunk's 32 inner channels (10 MFMAs).  The loop is skewed -- iteration c does FF1(c) and
//     FF2(c - 1) -- so there is ONE barrier per chunk and the GELU VALU work sits next to independent MFMAs.
// Arithmetic is that of the unfused path: h is rounded to the 16-bit element type before the second GEMM (as the GEGLU tensor
// was when it went through memory), fp32 accumulation everywhere, erf-based GELU (common.h).
#include <type_traits>
#include "gemm_common.h"
#include "lds_dma.h"

namespace {

constexpr int FF_C = 320, FF_I = 1280;
constexpr int FF_STEPS = FF_C / 16;                 // 20 k-steps of the first GEMM
constexpr int FF_NCH = FF_I / 32;                   // 40 chunks of 32 inner channels
constexpr int FF_W1_CHUNK = 64 * FF_C * 2;          // 40960: 2 blocks x 32 packed rows x 640 B
constexpr int FF_W2_CHUNK = FF_C * 32 * 2;          // 20480: 320 rows x 64 B
constexpr int FF_OFF_W2 = 2 * FF_W1_CHUNK;          // 81920
constexpr int FF_OFF_DUMP = FF_OFF_W2 + 2 * FF_W2_CHUNK;   // 122880 (1 KB landing zone of the empty DMA pieces)
constexpr int FF_OFF_HX = FF_OFF_DUMP + 1024;       // 123904: h exchange, 2 buffers x 8 waves x 1 KB
constexpr int FF_OFF_B1 = FF_OFF_HX + 2 * 8 * 1024; // 140288: packed b1, 2560 floats
constexpr int FF_OFF_B2 = FF_OFF_B1 + 2 * FF_I * 4; // 150528: b2, 320 floats
constexpr int FF_LDS = FF_OFF_B2 + FF_C * 4;        // 151808

template <bool F16>
__global__ __launch_bounds__(512, 1) void ff_geglu320_kernel(const imd_ff_params p) {
    using E = El<F16>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, col = lane & 31;
    const int rb = wave & 3, hh = wave >> 2;
    const int m0 = blockIdx.x * 128;
    const int m = m0 + rb * 32 + col;

    // ---- activations: 32 rows x 320 k per wave, straight into B-operand fragments ----
    const uint32_t x_bytes = (uint32_t)(((size_t)(p.M - 1) * p.x_ld + FF_C) * 2);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, x_bytes, 0x00020000);
    const uint32_t xoff = (uint32_t)m * (uint32_t)(p.x_ld * 2) + hi * 16;
    uint4 xf[FF_STEPS];
#pragma unroll
    for (int s = 0; s < FF_STEPS; ++s) xf[s] = buf_load16(rs_x, m < p.M ? xoff + s * 32 : OOB);
    float b1v[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) b1v[i] = p.b1[tid + 512 * i];
    const float b2v = tid < FF_C ? p.b2[tid] : 0.f;

    // ---- weight streams ----
    const v4i_t ds_w1 = raw_rsrc(p.w1, (uint32_t)(2 * FF_I * FF_C * 2)), ds_w2 = raw_rsrc(p.w2, (uint32_t)(FF_C * FF_I * 2));
    uint32_t w1off[5], w2off[3];
#pragma unroll
    for (int j = 0; j < 5; ++j) {                   // W1 chunk: 64 rows x 40 pieces, piece p of row r at p ^ ((r >> 1) & 7)
        const int q = (j * 8 + wave) * 64 + lane;
        const int row = q / 40, pos = q - row * 40;
        w1off[j] = (uint32_t)(row * (FF_C * 2) + ((pos ^ ((row >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {                   // W2 chunk: 320 rows x 4 pieces, piece p of row r at p ^ ((r >> 2) & 3)
        const int id = j * 8 + wave;
        const int q = id * 64 + lane;
        const int row = q >> 2, pos = q & 3;
        w2off[j] = id < 20 ? (uint32_t)(row * 64 + ((pos ^ ((row >> 2) & 3)) << 4)) : OOB;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    auto stage_w1 = [&](int c) {
        const uint32_t base = lds0 + (uint32_t)((c & 1) * FF_W1_CHUNK) + (uint32_t)wave * 1024u;
#pragma unroll
        for (int j = 0; j < 5; ++j) dma16(ds_w1, base + j * 8192u, w1off[j] + (uint32_t)c * FF_W1_CHUNK);
    };
    auto stage_w2 = [&](int c) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int id = j * 8 + wave;
            const uint32_t dst = id < 20 ? lds0 + FF_OFF_W2 + (uint32_t)((c & 1) * FF_W2_CHUNK) + (uint32_t)id * 1024u : lds0 + FF_OFF_DUMP;
            dma16(ds_w2, dst, w2off[j] == OOB ? OOB : w2off[j] + (uint32_t)c * FF_W2_CHUNK);
        }
    };
    stage_w1(0);
    // hipcc counts only its own loads: pin their wait here (it also covers W1 chunk 0, requested with them)
#pragma unroll
    for (int s = 0; s < FF_STEPS; ++s) asm volatile("" : "+v"(xf[s].x), "+v"(xf[s].y), "+v"(xf[s].z), "+v"(xf[s].w));
    {
        float* b1s = reinterpret_cast<float*>(smem + FF_OFF_B1);
#pragma unroll
        for (int i = 0; i < 5; ++i) b1s[tid + 512 * i] = b1v[i];
        if (tid < FF_C) reinterpret_cast<float*>(smem + FF_OFF_B2)[tid] = b2v;
    }

    if (p.ln) {      // LayerNorm without affine, in place (two-pass fp32; lanes l and l ^ 32 share a row)
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < FF_STEPS; ++s) {
            float f[8];
            unpack8<F16>(xf[s], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += f[e];
        }
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / FF_C);
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < FF_STEPS; ++s) asm volatile("" : "+v"(xf[s].x), "+v"(xf[s].y), "+v"(xf[s].z), "+v"(xf[s].w));
#pragma unroll
        for (int s = 0; s < FF_STEPS; ++s) {
            float f[8];
            unpack8<F16>(xf[s], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; sq = fmaf(d, d, sq); }
        }
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq * (1.0f / FF_C) + p.ln_eps);
        const float shift = -mean * rstd;
#pragma unroll
        for (int s = 0; s < FF_STEPS; ++s) asm volatile("" : "+v"(xf[s].x), "+v"(xf[s].y), "+v"(xf[s].z), "+v"(xf[s].w));
#pragma unroll
        for (int s = 0; s < FF_STEPS; ++s) {
            float f[8];
            unpack8<F16>(xf[s], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], rstd, shift);
            xf[s] = pack8<F16>(f);
        }
    }

    f32x16 acc_out[5];
#pragma unroll
    for (int cb = 0; cb < 5; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_out[cb][r] = 0.f;

    // fragment addresses
    const int w1row = hh * 32 + col;                                            // this wave's block of the W1 chunk
    const uint32_t a16 = (uint32_t)((hi ^ ((w1row >> 1) & 7)) << 4);
    const char* w1lane = smem + w1row * (FF_C * 2);
    const uint32_t w2sw = (uint32_t)((col >> 2) & 3);
    const char* w2lane = smem + FF_OFF_W2 + (hh * 5 * 32 + col) * 64;           // + cb * 2048 + ((2 t + hi) ^ w2sw) * 16
    uint4* hx = reinterpret_cast<uint4*>(smem + FF_OFF_HX);                      // [2][8][64]
    const float* b1s = reinterpret_cast<const float*>(smem + FF_OFF_B1);
    uint4 h_own = make_uint4(0, 0, 0, 0);

    // one chunk step, three stages deep: FF1(c) (DO1), the GELU of chunk c - 1 (DOG) and FF2(c - 2) (DO2).  The first GEMM is
    // a single dependent accumulator chain and the GELU is ~130 VALU instructions per wave: written out interleaved (two
    // first-GEMM MFMAs, one second-GEMM MFMA, one GELU) so that the VALU work issues in the shadow of independent MFMAs
    // instead of between the barrier and the next chunk (measured: GELU 32 us, first GEMM 40 us, second 24 us, all additive).
    f32x16 ahp;                      // first-GEMM result of the previous chunk, waiting for its GELU
#pragma unroll
    for (int r = 0; r < 16; ++r) ahp[r] = 0.f;
    auto step = [&](int c, auto do1, auto dog, auto do2) {
        constexpr bool DO1 = decltype(do1)::value, DOG = decltype(dog)::value, DO2 = decltype(do2)::value;
        dma_wait();                      // W1(c) and W2(c - 2), requested one iteration ago, have landed (this wave's pieces)
        __syncthreads();                 // ... everybody's; h(c - 2) is published; the slots of W1(c - 1) / W2(c - 3) are free
        if (DO1 && c + 1 < FF_NCH) stage_w1(c + 1);
        if (DOG) stage_w2(c - 1);        // needed by FF2(c - 1) in the next iteration
        uint4 h_par = make_uint4(0, 0, 0, 0);
        if (DO2) h_par = hx[(c & 1) * 512 + (wave ^ 4) * 64 + lane];
        const char* W2s = w2lane + (c & 1) * FF_W2_CHUNK;
        const uint4 hB0 = hh == 0 ? h_own : h_par, hB1 = hh == 0 ? h_par : h_own;
        const uint32_t po0 = (((uint32_t)hi) ^ w2sw) << 4, po1 = (((uint32_t)(2 + hi)) ^ w2sw) << 4;
        f32x16 ah;
        if (DO1) {
            const float* bb = b1s + (2 * c + hh) * 32 + 4 * hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = *reinterpret_cast<const float4*>(bb + 8 * q);
                ah[4 * q] = b.x; ah[4 * q + 1] = b.y; ah[4 * q + 2] = b.z; ah[4 * q + 3] = b.w;
            }
        }
        const char* W1s = w1lane + (c & 1) * FF_W1_CHUNK;
        float h[8];
        // Slot schedule, pinned with sched_barrier(0) (left alone hipcc issues the 30 MFMAs first, each behind the wait for its
        // own fragment, and the whole GELU afterwards).  Slot s: fragment reads two slots ahead, first-GEMM MFMA s, on odd slots a
        // second-GEMM MFMA, on even slots one GELU -- VALU and LDS latency sit in the shadow of MFMAs that do not depend on them.
        uint4 w1f[3], w2f[2];
        auto ld1 = [&](int st) { return *reinterpret_cast<const uint4*>(W1s + ((uint32_t)(st * 32) ^ a16)); };
        auto ld2 = [&](int i) { const int t = i / 5, cb = i - 5 * t; return *reinterpret_cast<const uint4*>(W2s + cb * 2048 + (t ? po1 : po0)); };
        if (DO1) { w1f[0] = ld1(0); w1f[1] = ld1(1); }
        if (DO2) w2f[0] = ld2(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < FF_STEPS; ++s) {
            if (DO1 && s + 2 < FF_STEPS) w1f[(s + 2) % 3] = ld1(s + 2);
            if (DO1) ah = E::mfma(w1f[s % 3], xf[s], ah);
            if (DO2 && (s & 1)) {
                const int i = s >> 1, t = i / 5, cb = i - 5 * t;          // ten MFMAs of the second GEMM, one per two k-steps
                if (i + 1 < 10) w2f[(i + 1) & 1] = ld2(i + 1);
                acc_out[cb] = E::mfma(w2f[i & 1], t ? hB1 : hB0, acc_out[cb]);
            }
            if (DOG && !(s & 1) && s < 16) {                              // eight GELUs, one per two k-steps
                const int e = s >> 1;
                float gate = e < 4 ? ahp[4 + e] : ahp[8 + e];
                asm volatile("" : "+v"(gate));                            // opaque: the GELU may neither be hoisted to the top ...
                h[e] = (e < 4 ? ahp[e] : ahp[4 + e]) * gelu_erf_f(gate);
                asm volatile("" : "+v"(h[e]));                            // ... nor sunk below the last MFMA (pure VALU floats across sched_barrier)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DOG) {                       // h(c - 1): own half stays in registers for FF2(c - 1), the partner finds it in LDS
            h_own = pack8<F16>(h);
            hx[((c - 1) & 1) * 512 + wave * 64 + lane] = h_own;
        }
        if (DO1) ahp = ah;
    };
    using T_ = std::true_type; using F_ = std::false_type;
    step(0, T_{}, F_{}, F_{});
    step(1, T_{}, T_{}, F_{});
#pragma unroll 1
    for (int c = 2; c < FF_NCH; ++c) step(c, T_{}, T_{}, T_{});
    step(FF_NCH, F_{}, T_{}, T_{});
    step(FF_NCH + 1, F_{}, F_{}, T_{});

    // ---- epilogue: + b2 + residual (the un-normalised input rows, re-read in accumulator layout), 16-byte stores ----
    const float* b2s = reinterpret_cast<const float*>(smem + FF_OFF_B2);
    const uint32_t o_bytes = (uint32_t)(((size_t)(p.M - 1) * p.out_ld + FF_C) * 2);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, o_bytes, 0x00020000);
    const uint32_t rbase = (uint32_t)m * (uint32_t)(p.x_ld * 2), obase = (uint32_t)m * (uint32_t)(p.out_ld * 2);
    uint2 res[5][4];
#pragma unroll
    for (int cb = 0; cb < 5; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = hh * 160 + cb * 32 + 8 * q + 4 * hi;
            res[cb][q] = buf_load8(rs_x, m < p.M ? rbase + (uint32_t)(n * 2) : OOB);
        }
    // (round 5) 16-byte stores: one v_permlane32_swap per packed register pair turns the accumulator layout's two 4-channel groups into 8
    // consecutive channels per lane -- half as many store requests, 32 contiguous bytes per row (row_linear.hip: "WIDE stores")
#pragma unroll
    for (int cb = 0; cb < 5; ++cb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            uint32_t pk[2][2];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * t + qq, n = hh * 160 + cb * 32 + 8 * q + 4 * hi;
                const float4 bb = *reinterpret_cast<const float4*>(b2s + n);
                const float v0 = acc_out[cb][4 * q] + bb.x + E::lo(res[cb][q].x), v1 = acc_out[cb][4 * q + 1] + bb.y + E::hi(res[cb][q].x);
                const float v2 = acc_out[cb][4 * q + 2] + bb.z + E::lo(res[cb][q].y), v3 = acc_out[cb][4 * q + 3] + bb.w + E::hi(res[cb][q].y);
                pk[qq][0] = E::pack2(v0, v1); pk[qq][1] = E::pack2(v2, v3);
            }
            const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
            typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4u_t;
            const v4u_t w = {r0[0], r1[0], r0[1], r1[1]};
            const int n8 = hh * 160 + cb * 32 + 16 * t + 8 * hi;
            __builtin_amdgcn_raw_buffer_store_b128(w, rs_o, (int)(m < p.M ? obase + (uint32_t)(n8 * 2) : OOB), 0, 0);
        }
}

}  // namespace

int imd_launch_ff_geglu(const imd_ff_params& p, hipStream_t s) {
    if (p.C != FF_C || p.inner != FF_I) return imd_set_error("ff_geglu: built for C = 320, inner = 1280 (got %d, %d)", p.C, p.inner);
    if (p.M <= 0 || (p.x_ld % 8) || (p.out_ld % 4) || p.x_ld < FF_C || p.out_ld < FF_C) return imd_set_error("ff_geglu: bad geometry M=%d x_ld=%d out_ld=%d", p.M, p.x_ld, p.out_ld);
    if (((size_t)(p.M - 1) * p.x_ld + FF_C) * 2 >= 0xffffffffull || ((size_t)(p.M - 1) * p.out_ld + FF_C) * 2 >= 0xffffffffull)
        return imd_set_error("ff_geglu: operand larger than 4 GiB");
    if (p.dtype != IMD_DTYPE_BF16 && p.dtype != IMD_DTYPE_F16) return imd_set_error("ff_geglu: unknown dtype %d", p.dtype);
    const bool h = p.dtype == IMD_DTYPE_F16;
    const void* kern = h ? reinterpret_cast<const void*>(ff_geglu320_kernel<true>) : reinterpret_cast<const void*>(ff_geglu320_kernel<false>);
    if (int rc_attr = imd_lds_attr(kern, FF_LDS, "ff_geglu")) return rc_attr;
    const dim3 grid((unsigned)((p.M + 127) / 128));
    if (h) hipLaunchKernelGGL(ff_geglu320_kernel<true>, grid, dim3(512), FF_LDS, s, p);
    else hipLaunchKernelGGL(ff_geglu320_kernel<false>, grid, dim3(512), FF_LDS, s, p);
    return imd_check_launch("ff_geglu");
}
