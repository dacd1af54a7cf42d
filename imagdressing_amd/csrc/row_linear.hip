// Row-resident linear layer for the level-0 token matrix (K = 320; N = 64 .. 320 in steps of 64):
//
//   out[m, :] = epilogue( LN?(x[m, :]) . W^T )          m = token, M ~ 32768, K = 320
//
// The tiled implicit-GEMM kernel (conv_gemm.hip) walks K in 32/64-element steps: on this shape every step is a round
// trip to memory for 64-byte row pieces with two tiles of lead, a barrier per step, and the kernel ends up at ~0.10 of the
// MFMA peak although it is not even HBM-bound (profiles/r2h_*).  Here the roles are turned around:
//   * a workgroup (8 waves) owns 128 complete token rows.  Each wave requests ITS 32 rows x 640 bytes -- one contiguous
//     20 KB span -- with 20 back-to-back 16-byte loads per lane, straight into the MFMA B-operand layout (lane = token,
//     8 consecutive k per 16-k step), and keeps them in 80 VGPRs for the whole kernel: all activation bytes of the launch
//     are in flight at once, there is no K loop on the activation side and no barrier coupled to memory latency.
//   * the weight matrix streams through LDS in chunks of 64 output channels x 320 k (40 KB) by LDS-DMA
//     (buffer_load ... lds, 3-slot ring, counted vmcnt, one barrier per chunk), shared by the 8 waves; rows are stored
//     unpadded with the 16-byte pieces of row r XOR-ed by (r >> 1) & 7 on the SOURCE side, which makes the ds_read_b128
//     fragment reads conflict-free.
//   * all N <= 320 output channels of a wave's 32 tokens stay in accumulators (80 VGPRs) until the end, so no store is in
//     flight while DMA pieces are being counted; the fp32 tile then crosses LDS once (the ring is dead by then) and leaves
//     through the same fused 8-channel epilogue as the tiled kernel (bias / residual / head-split Q layout ...).
//   * optional LayerNorm prologue (LN): mean / variance of each token row are taken from the registers (two-pass, fp32;
//     the two lanes that share a row exchange one partial each) and the rows are normalised in place; the affine part is
//     folded into the weights by the caller (W' = W diag(gamma), b' = b + W beta), so `layernorm -> linear` is ONE launch
//     and the normalised tensor never exists in memory.
// Reference arithmetic: diffusers==0.24.0 BasicTransformerBlock / Transformer2DModel (un-vendored; call sites
// /root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline.py:499,511): proj_in, attn.to_out[0] (+ residual),
// norm2 -> attn2.to_q, proj_out (+ residual), and adapter/attention_processor.py:568 (to_q), :617 (to_out) on the 64x64 level.
#include "gemm_common.h"
#include "lds_dma.h"

namespace {

constexpr int RL_K = 320;
constexpr int RL_STEPS = RL_K / 16;              // 20 MFMA k-steps
constexpr int RL_ROWB = RL_K * 2;                // bytes per weight row
constexpr int RL_CH = 64;                        // output channels per weight chunk
constexpr int RL_CHUNK = RL_CH * RL_ROWB;        // 40960 bytes
constexpr int RL_PIECES = RL_CHUNK / (8 * 1024); // DMA pieces per wave per chunk: 5
constexpr int RL_RING = 3;
constexpr int RL_BM = 128;
constexpr int RL_CLD = 320 + 4;                  // fp32 epilogue tile leading dimension
constexpr int RL_LDS = RL_RING * RL_CHUNK;       // 122880 >= 64 * RL_CLD * 4 = 82944
constexpr int RL_LDS_TOTAL = RL_LDS + 320 * 4;   // + the bias vector of the direct epilogue
static_assert(RL_PIECES == 5, "dma_wait_keep5 assumes five pieces per chunk");

// GN (round 6): GroupNorm (+ SiLU) of the rows from the statistic partials of x (imd_conv_gemm_params.gn_in_*) -- Transformer2DModel.norm -> proj_in in one launch
template <bool F16, int NC, bool LN, bool DIRECT, bool GN = false>
__global__ __launch_bounds__(512, 1) void row_linear_kernel(const ConvGemmParams p, const float ln_eps) {
    static_assert(!(LN && GN), "one prologue at a time");
    using E = El<F16>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, col = lane & 31;
    const int rb = wave & 3;            // 32-token block of the workgroup's 128 rows
    const int chh = wave >> 2;          // which 32-channel half of every 64-channel weight chunk
    const int m0 = blockIdx.x * RL_BM;

    GnInReq<RL_K> gnreq;
    if constexpr (GN) gn_in_request<RL_K>(p, m0 / (p.Hout * p.Wout), gnreq);      // (ahead of the activation loads: they come back first)
    // ---- activations: this wave's 32 rows, all of K, straight into B-operand fragments ----
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    const int m = m0 + rb * 32 + col;
    const uint32_t xoff = (uint32_t)m * (uint32_t)(p.x_pix_stride * 2) + hi * 16;
    uint4 xf[RL_STEPS];
#pragma unroll
    for (int s = 0; s < RL_STEPS; ++s) xf[s] = buf_load16(rs_x, m < p.M ? xoff + s * 32 : OOB);

    // DIRECT epilogue (row-major 16-bit output or head-split Q; bias / scale / residual only): every chunk's 32 x 32 block
    // leaves straight from the accumulators while the NEXT chunk is being multiplied, so the output stream overlaps the
    // MFMA phase instead of following it (all 256 workgroups run in lock-step: an epilogue at the end is fully exposed --
    // measured 9 us of the 21; HBM absorbs writes at only ~2.3 TB/s, so the 21 MB of output are the longest phase and
    // everything else should hide under them).  The residual of chunk c is requested while chunk c is multiplied and
    // consumed one chunk later, so its 21 MB ride under the output stream as well; the bias sits in LDS behind the ring.
    uint4 rraw[2][2];                                            // residual of a chunk: per register-group pair t, (group 2 t | group 2 t + 1) or, wide, 8 consecutive channels
    // (GN: the launcher sends row-major outputs without residual only -- Transformer2DModel.proj_in -- so the epilogue's run-time forks are compile-time
    // there: with them AND the prologue in one body hipcc spilled 190 registers)
    const bool wide = GN || (p.flags & 1024) != 0;               // 16-byte stores / residual loads (see emit)
    const bool has_res = DIRECT && !LN && !GN && p.res != nullptr;      // (LayerNorm + residual: staged epilogue)
    const bool heads = !GN && p.mode == OUT_HEADS;
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.res), 0, 0x80000000u, 0x00020000);
    // (periodic residual, res_rows > 0: row m adds res[m % res_rows]; res_rows is a multiple of the 128-row block, so the shift is per workgroup)
    const int res_shift = p.res_rows > 0 ? (m0 / p.res_rows) * p.res_rows : 0;
    const uint32_t roff = (uint32_t)(m - res_shift) * (uint32_t)(p.res_ld * 2) + (uint32_t)((wave >> 2) * 64);
    auto load_res = [&](int c) {        // residual of chunk c: lane = token; 8 consecutive channels per 16-byte load (wide; put into the accumulator layout
                                        // by emit), or 4 per 8-byte load in accumulator layout
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (wide) rraw[c & 1][t] = buf_load16(rs_r, m < p.M ? roff + (uint32_t)(c * 128 + t * 32 + hi * 16) : OOB);
            else {
                const uint2 a = buf_load8(rs_r, m < p.M ? roff + (uint32_t)(c * 128 + (2 * t) * 16 + hi * 8) : OOB);
                const uint2 b = buf_load8(rs_r, m < p.M ? roff + (uint32_t)(c * 128 + (2 * t + 1) * 16 + hi * 8) : OOB);
                rraw[c & 1][t] = make_uint4(a.x, a.y, b.x, b.y);
            }
        }
    };
    float bias_v = 0.f;              // requested behind the activation rows; parked in LDS once they have arrived (below)
    if (DIRECT && tid < NC * RL_CH && p.bias) bias_v = p.bias[tid];

    // ---- weight stream: source offsets of this lane's five pieces of a chunk (piece q of the chunk lands at LDS slot q) ----
    const v4i_t ds_w = raw_rsrc(p.w, p.w_bytes);
    uint32_t woff[RL_PIECES];
#pragma unroll
    for (int j = 0; j < RL_PIECES; ++j) {
        const int q = (j * 8 + wave) * 64 + lane;
        const int row = q / 40, pos = q - row * 40;
        woff[j] = (uint32_t)(row * RL_ROWB + ((pos ^ ((row >> 1) & 7)) << 4));
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;      // LDS aperture offset of the dynamic array
    auto stage = [&](int c) {
        const uint32_t base = lds0 + (uint32_t)((c % RL_RING) * RL_CHUNK) + (uint32_t)wave * 1024u;
#pragma unroll
        for (int j = 0; j < RL_PIECES; ++j) dma16(ds_w, base + j * 8192u, woff[j] + (uint32_t)c * RL_CHUNK);
    };
    stage(0);
    if (NC > 1) stage(1);
    // hipcc counts only its own (activation) loads: pin their wait HERE, where it also covers chunks 0 and 1 that were
    // requested with them, instead of in front of the last MFMA of chunk 0 where it would drain chunk 2 as well
#pragma unroll
    for (int s = 0; s < RL_STEPS; ++s) asm volatile("" : "+v"(xf[s].x), "+v"(xf[s].y), "+v"(xf[s].z), "+v"(xf[s].w));
    if (DIRECT && tid < NC * RL_CH) reinterpret_cast<float*>(smem + RL_LDS)[tid] = bias_v;     // published by the first barrier of the chunk loop

    if constexpr (LN) {      // rows normalised in place (affine folded into W / bias by the caller)
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < RL_STEPS; ++s) {
            float f[8];
            unpack8<F16>(xf[s], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += f[e];
        }
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / RL_K);
        float sq = 0.f;
        // (opaque touch: keeps hipcc from holding all 160 unpacked values of the row alive across the passes)
#pragma unroll
        for (int s = 0; s < RL_STEPS; ++s) asm volatile("" : "+v"(xf[s].x), "+v"(xf[s].y), "+v"(xf[s].z), "+v"(xf[s].w));
#pragma unroll
        for (int s = 0; s < RL_STEPS; ++s) {
            float f[8];
            unpack8<F16>(xf[s], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; sq = fmaf(d, d, sq); }
        }
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq * (1.0f / RL_K) + ln_eps);
        const float shift = -mean * rstd;
#pragma unroll
        for (int s = 0; s < RL_STEPS; ++s) asm volatile("" : "+v"(xf[s].x), "+v"(xf[s].y), "+v"(xf[s].z), "+v"(xf[s].w));
#pragma unroll
        for (int s = 0; s < RL_STEPS; ++s) {
            float f[8];
            unpack8<F16>(xf[s], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], rstd, shift);
            xf[s] = pack8<F16>(f);
        }
    }

    if constexpr (GN) {      // rows normalised in place with the per-channel coefficients of this workgroup's image (its 128 rows lie in ONE image)
        // scratch = ring slot 2: no DMA piece lands there before stage(2), which is issued behind the first barrier of the chunk loop
        const float *ga, *gs;
        gn_in_coeffs<RL_K>(p, m0 / (p.Hout * p.Wout), gnreq, reinterpret_cast<float*>(smem + 2 * RL_CHUNK), ga, gs);
        const bool silu = p.gn_in_silu != 0;
#pragma unroll
        for (int s = 0; s < RL_STEPS; ++s) {       // (one fragment at a time: left alone hipcc hoists all 80 coefficient reads and spills 190 registers)
            xf[s] = gn_in_apply8<F16>(xf[s], ga, gs, 16 * s + 8 * hi, silu);
            asm volatile("" ::: "memory");        // (keeps the coefficient reads of the next fragment behind this one)
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    f32x16 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    // direct epilogue of chunk c: channels c*64 + chh*32 + 8j + 4hi .. +3 of token m
    const int HWo = p.Hout * p.Wout;
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
        heads ? (void*)p.hd[0].ptr : p.out, 0, 0x80000000u, 0x00020000);
    uint32_t obase = OOB;            // byte offset of (token m, channel 0) in the row-major case / of (bi, head 0, tok, 0) for head-split Q
    if (DIRECT && m < p.M) {
        if (heads) { const int bi = m / HWo, tok = m - bi * HWo; obase = (uint32_t)(((size_t)bi * p.hH * p.hd[0].L + tok) * p.hd[0].DP * 2); }
        else obase = (uint32_t)m * (uint32_t)(p.out_ld * 2);
    }
    // WIDE stores (round 5): the accumulator layout gives a lane 4 channels of a row, so an 8-byte store instruction puts 16 contiguous bytes
    // into each of 32 rows -- fragments the L2 takes at its REQUEST rate (measured on the GEMM epilogues: 16-byte fragments drain at 2.5 TB/s, a
    // plain fill writes at 6.2, profiles/r5d_write_bw_probe.jsonl).  One v_permlane32_swap per packed register pair turns two 4-channel groups
    // into 8 consecutive channels per lane: half as many store requests, 32 contiguous bytes per row.  (Tuning knob 2 bit 10 = the 8-byte form.)
    auto emit = [&](int c) {
        const float* bias_s = reinterpret_cast<const float*>(smem + RL_LDS);
        uint2 rres[4];                     // residual in accumulator layout: group j = channels 8 j + 4 hi .. + 3
        if (has_res) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint4 r = rraw[c & 1][t];
                if (wide) {                // lane hi = 0 holds channels 16 t + 0..7, hi = 1 holds 16 t + 8..15: swap hi = 0's upper half with hi = 1's lower half
                    const auto sx = __builtin_amdgcn_permlane32_swap(r.x, r.z, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(r.y, r.w, false, false);
                    r = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                }
                rres[2 * t] = make_uint2(r.x, r.y); rres[2 * t + 1] = make_uint2(r.z, r.w);
            }
        }
        const float osc = heads ? p.out_scale * p.hd[0].scale : p.out_scale;
        typedef __attribute__((__vector_size__(2 * sizeof(uint32_t)))) uint32_t v2u;
        typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4u_t;
        v2u pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = c * 64 + chh * 32 + 8 * j + 4 * hi;
            const float4 bb = *reinterpret_cast<const float4*>(bias_s + n);
            float v0 = (acc[c][4 * j] + bb.x) * osc, v1 = (acc[c][4 * j + 1] + bb.y) * osc;
            float v2 = (acc[c][4 * j + 2] + bb.z) * osc, v3 = (acc[c][4 * j + 3] + bb.w) * osc;
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
        if (wide) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {      // groups j = 2 t (channels 16 t + 4 hi ..) and 2 t + 1 (16 t + 8 + 4 hi ..) -> channels 16 t + 8 hi + 0..7
                const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * t][0], pk[2 * t + 1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * t][1], pk[2 * t + 1][1], false, false);
                const v4u_t w = {r0[0], r1[0], r0[1], r1[1]};
                const uint32_t off = offset_of(c * 64 + chh * 32 + 16 * t + 8 * hi);
                __builtin_amdgcn_raw_buffer_store_b128(w, rs_o, (int)(obase == OOB ? OOB : obase + off), 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b64(pk[j], rs_o, (int)(obase == OOB ? OOB : obase + offset_of(c * 64 + chh * 32 + 8 * j + 4 * hi)), 0, 0);
        }
    };

    const int wrow = chh * 32 + col;                                     // weight row inside a chunk
    const uint32_t a16 = (uint32_t)((hi ^ ((wrow >> 1) & 7)) << 4);      // piece 2s + hi of row wrow sits at ((2s) ^ (hi ^ f)) * 16
    const char* wlane = smem + wrow * RL_ROWB;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        // this wave's pieces of chunk c have landed (chunk c + 1 may still fly).  With the direct epilogue stores are in
        // flight too, and loads and stores do not retire in order with each other: no counted wait, drain everything (the
        // youngest operations are one whole chunk old by now).
        if (DIRECT || c + 1 >= NC) dma_wait(); else dma_wait_keep5();
        __syncthreads();                                          // ... and everybody else's; all waves are done with chunk c - 1
        // order matters: hipcc drains the memory counter in front of emit()'s use of the residual registers (loads and stores
        // pending together) -- at this point nothing is in flight, after stage() the fresh DMA pieces would be
        if (DIRECT && c > 0) emit(c - 1);
        if (c + 2 < NC) stage(c + 2);                             // into the slot chunk c - 1 just vacated
        if (has_res) load_res(c);
        const char* Ws = wlane + (c % RL_RING) * RL_CHUNK;
#pragma unroll
        for (int s = 0; s < RL_STEPS; ++s) {
            const uint4 wf = *reinterpret_cast<const uint4*>(Ws + ((uint32_t)(s * 32) ^ a16));
            acc[c] = E::mfma(wf, xf[s], acc[c]);
        }
    }

    if constexpr (DIRECT) { emit(NC - 1); return; }
    // ---- staged epilogue (activations, fp32 output, V^T head layouts ...): 64 rows at a time through LDS (fp32), then 8
    // consecutive channels of a row per thread through the shared epilogue8 ----
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CPR = NC * 8;                  // 8-channel chunks per row
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if ((rb >> 1) == half) {
            float* dst0 = Cs + ((rb & 1) * 32 + col) * RL_CLD + chh * 32 + 4 * hi;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float4*>(dst0 + c * 64 + 8 * j) = make_float4(acc[c][4 * j], acc[c][4 * j + 1], acc[c][4 * j + 2], acc[c][4 * j + 3]);
        }
        __syncthreads();
        for (int t = tid; t < 64 * CPR; t += 512) {
            const int row = t / CPR, cc = (t - row * CPR) * 8;
            const int mm = m0 + half * 64 + row;
            if (mm >= p.M) continue;
            const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * RL_CLD + cc);
            const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * RL_CLD + cc + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (p.res_rows > 0) {      // (staged form of the periodic residual: the same shared epilogue on a parameter block whose residual base is shifted)
                ConvGemmParams q = p;
                q.res = p.res - (size_t)res_shift * p.res_ld;
                epilogue8<F16>(q, v, mm, cc, 8, HWo);
            } else
            epilogue8<F16>(p, v, mm, cc, 8, HWo);
        }
    }
}

template <bool F16, int NC, bool LN, bool DIRECT, bool GN = false>
int launch_rl_d(const ConvGemmParams& p, float eps, hipStream_t s) {
    auto kern = row_linear_kernel<F16, NC, LN, DIRECT, GN>;
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), RL_LDS_TOTAL, "row_linear")) return rc_attr;
    hipLaunchKernelGGL(kern, dim3((unsigned)((p.M + RL_BM - 1) / RL_BM)), dim3(512), RL_LDS_TOTAL, s, p, eps);
    return imd_check_launch("row_linear");
}

template <bool F16, int NC, bool LN>
int launch_rl(const ConvGemmParams& p, float eps, hipStream_t s) {
    const bool direct = p.act == ACT_NONE && !p.out_f32 && p.rowvec == nullptr && (g_gemm_flags & 256) == 0 &&
                        (p.mode == OUT_ROWMAJOR || (p.hd[0].kind == 0 && p.hd[0].ptr != nullptr && p.N == p.hC));
    const size_t ob = p.mode == OUT_HEADS ? (size_t)(p.M / (p.Hout * p.Wout)) * p.hH * p.hd[0].L * p.hd[0].DP * 2 : ((size_t)(p.M - 1) * p.out_ld + p.N) * 2;
    const size_t rb = p.res ? ((size_t)(p.M - 1) * p.res_ld + p.N) * 2 : 0;
    if (p.gn_in_partial != nullptr) {        // (never together with LN: refused by the launcher)
        if constexpr (!LN) {
            if (direct && ob < 0x80000000ull && rb < 0x80000000ull) return launch_rl_d<F16, NC, false, true, true>(p, eps, s);
            return launch_rl_d<F16, NC, false, false, true>(p, eps, s);
        }
    }
    if (direct && !(LN && p.res) && ob < 0x80000000ull && rb < 0x80000000ull) return launch_rl_d<F16, NC, LN, true>(p, eps, s);
    return launch_rl_d<F16, NC, LN, false>(p, eps, s);
}

template <bool F16, bool LN>
int launch_rl_n(const ConvGemmParams& p, float eps, hipStream_t s) {
    switch (p.N / RL_CH) {
        case 1: return launch_rl<F16, 1, LN>(p, eps, s);
        case 2: return launch_rl<F16, 2, LN>(p, eps, s);
        case 3: return launch_rl<F16, 3, LN>(p, eps, s);
        case 4: return launch_rl<F16, 4, LN>(p, eps, s);
        default: return launch_rl<F16, 5, LN>(p, eps, s);
    }
}

}  // namespace

bool imd_row_linear_supported(const ConvGemmParams& p) {
    return p.taps == 1 && p.K == RL_K && p.Cin == RL_K && p.stride == 1 && !p.ups && p.Hin == p.Hout && p.Win == p.Wout &&
           p.N >= RL_CH && p.N <= 320 && (p.N % RL_CH) == 0 && p.split_k <= 1 && p.act != ACT_GEGLU && p.gn_a == nullptr && (p.x_pix_stride % 8) == 0;
}

// ln != 0: LayerNorm (no affine) over the K = 320 channels of every row is applied to x on the fly
int imd_launch_row_linear(const ConvGemmParams& p_in, int ln, float ln_eps, hipStream_t s) {
    ConvGemmParams p = p_in;
    if (!imd_row_linear_supported(p))
        return imd_set_error("row_linear: needs a plain linear layer with K = 320 and N = 64..320 in steps of 64 (got N=%d K=%d taps=%d split=%d)", p.N, p.K, p.taps, p.split_k);
    if (p.gn_in_partial != nullptr && (ln || !gn_in_ok(p, RL_K, RL_BM)))
        return imd_set_error("row_linear: gn_in_* needs K = 320, K %% groups == 0, groups <= 64, H W %% 128 == 0 and no LayerNorm prologue (ask imd_row_linear_gn_in_supported())");
    if (p.res_rows != 0 && (p.res == nullptr || p.res_rows < 0 || p.res_rows % RL_BM || p.M % p.res_rows))
        return imd_set_error("row_linear: res_rows (%d) needs a residual, a multiple of %d rows and a divisor of M = %d", p.res_rows, RL_BM, p.M);
    if (p.dtype != IMD_DTYPE_BF16 && p.dtype != IMD_DTYPE_F16) return imd_set_error("row_linear: unknown dtype %d", p.dtype);
    const size_t xb = ((size_t)(p.M - 1) * p.x_pix_stride + p.K) * 2, wb = (size_t)p.N * p.K * 2;
    if (xb >= 0xffffffffull) return imd_set_error("row_linear: operand larger than 4 GiB");
    p.x_bytes = (uint32_t)xb;
    p.w_bytes = (uint32_t)wb;
    p.split_k = 1;
    p.flags = (g_gemm_flags & 1024) ? 0 : 1024;        // bit 10 of the kernel's flags: wide (16-byte) stores of the direct epilogue
    const bool h = p.dtype == IMD_DTYPE_F16;
    if (ln) return h ? launch_rl_n<true, true>(p, ln_eps, s) : launch_rl_n<false, true>(p, ln_eps, s);
    return h ? launch_rl_n<true, false>(p, ln_eps, s) : launch_rl_n<false, false>(p, ln_eps, s);
}
