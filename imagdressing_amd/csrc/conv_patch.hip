// 3x3 / stride-1 convolution with the input HALO PATCH resident in LDS (gfx950, bf16 / fp16 MFMA).
//
// The gather kernel (conv_gemm.hip) re-fetches the activation tile from L2 for each of the nine taps.  Here a
// workgroup owns an 8 x 16 block of output pixels of one image x 128 output channels and, per 32-channel chunk,
// stages the (8+2) x (16+2) input patch in LDS ONCE; the nine taps then read their MFMA operand fragments from
// that patch at a constant row offset ((ky*18 + kx) rows), so only the weight tile streams per tap.
// Operand traffic per tap drops from 16 KB to ~9.6 KB and the per-tap address arithmetic disappears.
//
// GroupNorm + SiLU fusion: when gn_a / gn_b are given (per-(batch, channel) scale and shift produced from the
// GroupNorm statistics), the normalisation and the SiLU are applied while the patch is written to LDS -- once per
// input element instead of as a separate read-modify-write pass over the whole tensor
// (diffusers ResnetBlock2D: norm -> SiLU -> conv; call sites IMAGDressing_v1_pipeline.py:466,499,511).
// The zero halo stays zero (padding applies AFTER norm + activation in the reference).
//
// Pipeline: A patch double-buffered across channel chunks (fetched during tap 0, written after tap 8), weight tile
// double-buffered across taps, one barrier per tap; epilogue shared with conv_gemm.hip (bias / time-embedding
// vector / residual / activation fused, coalesced 16-byte stores); optional split over channel chunks.
#include <type_traits>

#include "gemm_common.h"
#include "lds_dma.h"

namespace {

constexpr int TH = 8, TW = 16;                 // output pixels per workgroup: 8 rows x 16 columns
constexpr int PW = TW + 2, PH = TH + 2;        // halo patch
constexpr int NPIX = PH * PW;                  // 180 patch pixels
constexpr int CK = 32;                         // channels per chunk
constexpr int RSTR = CK * 2 + 16;              // 80-byte LDS rows (5 x 16 B: odd slot count)
constexpr int BN = 128;
constexpr int A_BYTES = NPIX * RSTR;           // 14,400
constexpr int W_BYTES = BN * RSTR;             // 10,240
constexpr int MAIN_LDS = 2 * A_BYTES + 2 * W_BYTES;   // 49,280
constexpr int CLD = BN + 4;
constexpr int EROWS = 64;
constexpr int EPI_LDS = EROWS * CLD * 4;       // 33,792
constexpr int PATCH_LDS = MAIN_LDS > EPI_LDS ? MAIN_LDS : EPI_LDS;
constexpr int A_VECS = (NPIX * (CK / 8) + 255) / 256;   // 3
// MFMA column (lane & 31) -> pixel of the wave's 2 x 16 pixel block.  ds_read_b128 is serviced in the 16-lane groups
// {0-3,12-15,20-27} / {4-11,16-19,28-31}; with the identity mapping the second image row (patch rows +18) lands two
// lanes of a group on one 16-byte bank slot (2-way conflict on every activation fragment).  This permutation gives
// each group 16 patch rows that are distinct mod 16, i.e. conflict-free at the 80-byte row stride.
__device__ constexpr unsigned char kColPix[32] = {0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 15, 4, 5, 6, 7,
                                                  30, 31, 16, 17, 22, 23, 24, 25, 26, 27, 28, 29, 18, 19, 20, 21};
constexpr int W_VECS = BN * (CK / 8) / 256;             // 2

// DMA (round 3): BOTH operands reach LDS by LDS-DMA (`buffer_load ... lds`), nothing is staged through registers.  The halo patch
// of a 32-channel chunk (180 pixel rows x 64 B, 12 one-KB pieces) and the weight tile of a tap (128 rows x 64 B, 8 pieces) are
// written lane-linear, i.e. as unpadded 64-byte rows; fragment reads stay conflict-free through a SOURCE-side swizzle -- piece
// c of row r sits at position c ^ ((r >> 2) & 3), so the 16 rows of a ds_read_b128 lane group (any 8 + 8 consecutive rows of the
// patch, or 4 aligned row quadruples of the weight tile) fall on 16 distinct 16-byte bank slots.  Weight tiles run through a
// ring of three taps (tap t + 2 in flight while tap t is multiplied, counted s_waitcnt); the next chunk's patch is fetched at
// tap 5 into the other patch buffer; out-of-image halo pixels and rows / taps past the end are out-of-range offsets that the
// DMA turns into zeros.  Against the register-staged form: no VGPR -> LDS stores (the slow LDS write path: ~79 B/clk) and no
// staging registers.  48 KB of LDS -> 3 workgroups per CU as before.  The fused GroupNorm prologue needs the values in
// registers and keeps the register-staged kernel.
constexpr int AB_D = 12 * 1024, WB_D = 8 * 1024, NWR_D = 3;
constexpr int PATCH_DMA_LDS = 2 * AB_D + NWR_D * WB_D;            // 49,152 (>= EPI_LDS)
static_assert(PATCH_DMA_LDS >= EPI_LDS, "the epilogue tile must fit the main-loop LDS");
// CKD = 64 (tile config 29, round 4): 64-channel chunks = 128-BYTE rows.  The L2 hands a CU whole 128-byte lines (tools/probes/staging_probe.hip:
// 62 GB/s per CU in 64-byte segments, 113 in 128-byte ones); the 64-byte rows above use half of every line they pull.  Patch 180 x 128 B (23 pieces,
// six per wave with one empty), weight tile 128 x 128 B (16 pieces, four per wave), piece c of row r at c ^ ((r >> 1) & 7); 16 MFMAs per wave and
// tap.  80 KB of LDS with a TWO-slot weight ring (the next tap's tile lands while this tap is multiplied) -> two workgroups per CU.
template <int CKD> struct PD {
    static constexpr int RB = CKD * 2, LPR = RB / 16;                       // row bytes, 16-byte pieces per row
    static constexpr int APIECES = (NPIX * RB + 1023) / 1024;               // 12 | 23
    static constexpr int APW = (APIECES + 3) / 4;                           // patch pieces per wave: 3 | 6
    static constexpr int AB = APW * 4 * 1024;                               // 12,288 | 24,576
    static constexpr int WPW = BN * RB / 1024 / 4;                          // weight pieces per wave and tap: 2 | 4
    static constexpr int WB = WPW * 4 * 1024;                               // 8,192 | 16,384
    static constexpr int NWR = CKD == 32 ? 3 : 2;
    static constexpr int LDS = 2 * AB + NWR * WB;                           // 49,152 | 81,920
    static_assert(LDS >= EPI_LDS, "the epilogue tile must fit the main-loop LDS");
    static __device__ __forceinline__ int swz(int r) { return CKD == 32 ? (r >> 2) & 3 : (r >> 1) & 7; }
};

template <bool F16, bool DMA, int CKD = 32>
__global__ __launch_bounds__(256, CKD == 32 ? 3 : 2) void conv3x3_patch_kernel(const ConvGemmParams p) {
    static_assert(CKD == 32 || DMA, "64-channel chunks exist for the LDS-DMA form only");
    using G = PD<CKD>;
    using E = El<F16>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Abuf = smem;
    char* const Wbuf = smem + 2 * A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave >> 1) * 64;          // 2 x 2 waves, each 64 pixels x 64 channels
    const int wn0 = (wave & 1) * 64;
    const int hi = lane >> 5, col = lane & 31;
    const int cpix = kColPix[col];             // pixel (0..31) this lane's MFMA column stands for

    const int H = p.Hout, W = p.Wout;          // output map = logical input map (fused nearest-2x upsample: twice the stored input, DMA path only)
    // ragged maps (96 x 72 latents of the 768 x 576 configuration: W = 72, 36, 18): the last tile row / column hangs over the
    // edge; its patch pixels outside the image read as zero like any halo pixel and its output pixels are not stored
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int n_tiles = (p.N + BN - 1) / BN;
    int bid, tile_n;
    xcd_tile_order(p.flags, (int)(gridDim.x / n_tiles), n_tiles, bid, tile_n);      // bid = pixel-tile index
    const int tile_id = bid * n_tiles + tile_n;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = tile_n * BN;

    const int nchunks = p.Cin / CKD;
    const int split = blockIdx.y;
    const int per = (nchunks + p.split_k - 1) / p.split_k;
    const int c_begin = split * per;
#ifdef PATCH_T_NOLOOP
    const int c_end = c_begin;                  // timing probe: launch + prologue + epilogue only
#else
    const int c_end = min(nchunks, c_begin + per);
#endif

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- per-thread staging assignments ----
    uint32_t a_off[A_VECS];      // byte offset of (patch pixel, channel 8*vc) at chunk 0, or OOB (halo outside the image)
    int a_lds[A_VECS];           // LDS byte offset inside an A buffer, or -1
    int a_ch[A_VECS];            // channel offset 8*vc (for the fused GroupNorm coefficients)
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) {
        // staging map: 16 consecutive lanes write 4 rows x 4 pieces; with the 80-byte row stride rows R, R+1, R+2, R+3 put three
        // of the sixteen 16-byte pieces on an occupied bank slot, rows R, R+4, R+8, R+12 do not (5 r mod 16 = 0, 4, 8, 12)
        const int v = tid + i * 256;
        const int vq = v & 63, vc = vq & 3;
        const int pp = (v >> 6) * 16 + (vq >> 4) + 4 * ((vq >> 2) & 3);
        a_lds[i] = -1; a_off[i] = OOB; a_ch[i] = vc * 8;
        if (pp < NPIX) {
            const int iy = y0 - 1 + pp / PW, ix = x0 - 1 + pp % PW;
            a_lds[i] = pp * RSTR + vc * 16;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                a_off[i] = (uint32_t)(((b * H + iy) * W + ix) * p.x_pix_stride + vc * 8) * 2u;
        }
    }
    uint32_t w_off[W_VECS];
    int w_lds[W_VECS];
#pragma unroll
    for (int i = 0; i < W_VECS; ++i) {
        const int v = tid + i * 256;
        const int vq = v & 63, vc = vq & 3;
        const int row = (v >> 6) * 16 + (vq >> 4) + 4 * ((vq >> 2) & 3);      // (same conflict-free staging map)
        w_lds[i] = row * RSTR + vc * 16;
        w_off[i] = (n0 + row < p.N) ? (uint32_t)(((size_t)(n0 + row) * p.K + vc * 8) * 2) : OOB;
    }

    // weight tiles are fetched TWO taps ahead into alternating register sets (w_r0 / w_r1): one tap of MFMA work
    // (~256 matrix-core cycles per wave) is shorter than the L2 latency, two taps x 3 resident workgroups are not
    uint4 a_reg[A_VECS], w_r0[W_VECS], w_r1[W_VECS];
    auto load_patch = [&](int c) {
#pragma unroll
        for (int i = 0; i < A_VECS; ++i)
            a_reg[i] = buf_load16(rs_x, (a_off[i] != OOB && c < c_end) ? a_off[i] + (uint32_t)(c * CK * 2) : OOB);
    };
    auto store_patch = [&](int buf, int c) {
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
            if (a_lds[i] < 0) continue;
            uint4 v = a_reg[i];
            if (p.gn_a != nullptr && a_off[i] != OOB) {        // fused GroupNorm (+SiLU); the zero halo stays zero
                const int ch = c * CK + a_ch[i];
                const float* ga = p.gn_a + (size_t)b * p.Cin + ch;
                const float* gb = p.gn_b + (size_t)b * p.Cin + ch;
                const float4 s0 = *reinterpret_cast<const float4*>(ga), s1 = *reinterpret_cast<const float4*>(ga + 4);
                const float4 t0 = *reinterpret_cast<const float4*>(gb), t1 = *reinterpret_cast<const float4*>(gb + 4);
                float f[8];
                unpack8<F16>(v, f);
                f[0] = fmaf(f[0], s0.x, t0.x); f[1] = fmaf(f[1], s0.y, t0.y); f[2] = fmaf(f[2], s0.z, t0.z); f[3] = fmaf(f[3], s0.w, t0.w);
                f[4] = fmaf(f[4], s1.x, t1.x); f[5] = fmaf(f[5], s1.y, t1.y); f[6] = fmaf(f[6], s1.z, t1.z); f[7] = fmaf(f[7], s1.w, t1.w);
                if (p.gn_silu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
                }
                v = pack8<F16>(f);
            }
            *reinterpret_cast<uint4*>(Abuf + buf * A_BYTES + a_lds[i]) = v;
        }
    };
    // it = flattened (chunk, tap) index relative to c_begin; past the end => zero-fill loads that touch no memory
    const int total = (c_end - c_begin) * 9;
    auto load_w = [&](uint4 (&wr)[W_VECS], int it) {
        const int c = c_begin + it / 9, t = it % 9;
        const uint32_t koff = (uint32_t)((t * p.Cin + c * CK) * 2);
#pragma unroll
        for (int i = 0; i < W_VECS; ++i)
            wr[i] = buf_load16_nl1(rs_w, (w_off[i] != OOB && it < total) ? w_off[i] + koff : OOB);
    };
    auto store_w = [&](const uint4 (&wr)[W_VECS], int buf) {
#pragma unroll
        for (int i = 0; i < W_VECS; ++i) *reinterpret_cast<uint4*>(Wbuf + buf * W_BYTES + w_lds[i]) = wr[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][bb][r] = 0.f;

    // A-fragment base rows: output pixel q = wm0 + bb*32 + col -> patch row (q/16)*18 + q%16 (+ tap offset)
    int a_frag[2];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
        const int q = wm0 + bb * 32 + cpix;
        a_frag[bb] = ((q / TW) * PW + (q % TW)) * RSTR + hi * 16;
    }
    const int w_frag = (wn0 + col) * RSTR + hi * 16;
    // the last channel tile of N = 320 / 960 ... is half empty: waves that own no valid channel skip the matrix work
    const bool wave_live = n0 + wn0 < p.N;

    if constexpr (DMA) {
        // Round 4: the loop's bookkeeping cut to the bone -- it was 500 instructions per 32-channel chunk for 72 MFMAs (62 address adds, 42
        // out-of-range selects, 63 s_mov + 42 s_nop of M0 save / restore and wait states, a divergent-branch skeleton around every tap):
        //   * every LDS address of the loop is a compile-time offset from a loop-invariant register: the 9 x 2 x 2 activation-fragment
        //     addresses (tap, pixel block, 16-deep slice) are computed ONCE, the chunk loop is unrolled over its two patch buffers and the
        //     nine taps over the three weight-ring slots;
        //   * a staging piece is three instructions (s_add m0 / s_nop / buffer_load ... lds) on a RUNNING source offset (+ one add per tap);
        //     out-of-image halo pixels and channel rows past N carry the offset 2^31, which stays out of range under the running adds (the
        //     operands are < 2 GiB: checked by the launcher); pieces staged past the last tap / chunk read in-range bytes nobody multiplies;
        //   * the "this wave owns no valid channel" test (last channel tile of N = 320) is a scalar branch.
        // Same MFMA order per accumulator as before: bit-identical results.
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
        const v4i_t dx = raw_rsrc(p.x, p.x_bytes), dw = raw_rsrc(p.w, p.w_bytes);
        constexpr uint32_t FAR = 0x80000000u;
        uint32_t acur[G::APW], wcur[G::WPW];  // running source offsets: patch pieces of the NEXT chunk to stage, weight pieces of the next tap
        uint32_t adst[G::APW], wdst[G::WPW];  // LDS byte addresses of the pieces inside patch buffer 0 / ring slot 0 (wave-uniform)
#pragma unroll
        for (int i = 0; i < G::APW; ++i) {
            const int slot = (wv * G::APW + i) * 64 + lane, pp = slot / G::LPR, piece = (slot % G::LPR) ^ G::swz(pp);
            acur[i] = FAR;
            if (pp < NPIX) {
                const int iy = y0 - 1 + pp / PW, ix = x0 - 1 + pp % PW;        // logical pixel; the zero halo is applied AFTER the upsample
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;     // nearest-2x: source pixel = logical pixel >> 1
                    acur[i] = (uint32_t)(((b * p.Hin + sy) * p.Win + sx) * p.x_pix_stride + piece * 8 + c_begin * CKD) * 2u;
                }
            }
            adst[i] = smem_base + (uint32_t)((wv * G::APW + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < G::WPW; ++i) {
            const int slot = (wv * G::WPW + i) * 64 + lane, row = slot / G::LPR, piece = (slot % G::LPR) ^ G::swz(row);
            wcur[i] = (n0 + row < p.N) ? (uint32_t)(((size_t)(n0 + row) * p.K + piece * 8 + c_begin * CKD) * 2) : FAR;
            wdst[i] = smem_base + (uint32_t)(2 * G::AB + (wv * G::WPW + i) * 1024);
        }
        const uint32_t w_tap = (uint32_t)(p.Cin * 2), w_chunk = (uint32_t)(CKD * 2) - 8u * w_tap;      // next tap / tap 8 -> tap 0 of the next chunk
        auto dma_patch = [&](auto buf_c) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < G::APW; ++i) {
                asm volatile("s_add_u32 m0, %1, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                             : : "v"(acur[i]), "s"(adst[i]), "s"(dx), "n"(decltype(buf_c)::value * G::AB) : "memory", "scc");
                acur[i] += (uint32_t)(CKD * 2);
            }
        };
        auto dma_w = [&](auto ring_c, auto cross_c) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < G::WPW; ++i) {
                asm volatile("s_add_u32 m0, %1, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                             : : "v"(wcur[i]), "s"(wdst[i]), "s"(dw), "n"(decltype(ring_c)::value * G::WB) : "memory", "scc");
                wcur[i] += decltype(cross_c)::value ? w_chunk : w_tap;
            }
        };
        // fragment addresses (bytes inside a patch buffer / a ring slot), all loop-invariant
        int w_fr[2], xa[9][2];            // 16-deep slice kk = 0; slice kk is the same address ^ (kk << 5) (one VALU in the loop instead of more live registers)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int row = wn0 + a * 32 + col;
            w_fr[a] = row * G::RB + ((hi ^ G::swz(row)) << 4);
        }
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            const int q = wm0 + bb * 32 + cpix, r0 = (q / TW) * PW + (q % TW);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int rw = r0 + (t / 3) * PW + (t % 3);
                xa[t][bb] = rw * G::RB + ((hi ^ G::swz(rw)) << 4);
            }
        }
        const bool live = __builtin_amdgcn_readfirstlane((int)(n0 + (wv & 1) * 64 < p.N)) != 0;      // scalar: the whole wave or nothing
        const std::integral_constant<int, 0> i0{}; const std::integral_constant<int, 1> i1{}; const std::integral_constant<int, 2> i2{};
        if (total > 0) {
            dma_patch(i0);
            dma_w(i0, std::false_type{});
            if (G::NWR == 3) dma_w(i1, std::false_type{});
        }
        dma_wait();
        __syncthreads();
        auto chunk = [&](auto ab_c) __attribute__((always_inline)) {       // one chunk out of patch buffer ab_c
            constexpr int AB = decltype(ab_c)::value;
            const char* As = smem + AB * G::AB;
#pragma unroll
            for (int t = 0; t < 9; ++t) {                  // ring slots are compile-time: 9 taps = 3 turns of a three-slot ring; two slots alternate over a chunk PAIR
                constexpr int LEAD = G::NWR - 1;           // the tile staged at tap t is tap t + LEAD's
                const int cur = G::NWR == 3 ? t % 3 : (9 * AB + t) % 2;
                if (G::NWR == 3) {
                    if (t % 3 == 0) dma_w(i2, std::false_type{});
                    else if (t % 3 == 1) dma_w(i0, std::false_type{});
                    else dma_w(i1, std::false_type{});
                } else {
                    if (((9 * AB + t) & 1) == 0) dma_w(i1, std::false_type{}); else dma_w(i0, std::false_type{});
                }
                if (t == 8 - LEAD) {                       // (the piece just staged was tap 8: the next one is tap 0 of the next chunk)
#pragma unroll
                    for (int i = 0; i < G::WPW; ++i) wcur[i] += w_chunk - w_tap;
                }
                if (t == 5) dma_patch(std::integral_constant<int, AB ^ 1>{});              // (always APW pieces: the counted waits rely on it)
                if (live) {
                    const char* Ws = smem + 2 * G::AB + cur * G::WB;
#pragma unroll
                    for (int k2 = 0; k2 < CKD / 32; ++k2) {            // 32 channels at a time: eight fragment reads, eight MFMAs
                        uint4 wf[2][2], xf[2][2];
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                            for (int a = 0; a < 2; ++a) wf[kk][a] = *reinterpret_cast<const uint4*>(Ws + (w_fr[a] ^ ((2 * k2 + kk) * 32)));
#pragma unroll
                            for (int bb = 0; bb < 2; ++bb) xf[kk][bb] = *reinterpret_cast<const uint4*>(As + (xa[t][bb] ^ ((2 * k2 + kk) * 32)));
                        }
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                            for (int a = 0; a < 2; ++a)
#pragma unroll
                                for (int bb = 0; bb < 2; ++bb) acc[a][bb] = E::mfma(wf[kk][a], xf[kk][bb], acc[a][bb]);
                    }
                }
                // the next tap's weight pieces have landed.  Three slots: everything but this tap's pieces (and, at taps 5 and 6, the patch pieces
                // issued behind them at tap 5) may stay in flight.  Two slots: the next tap's tile is the one just staged -- only the patch pieces
                // issued behind it at tap 5 may stay
                if (G::NWR == 3) { if (t == 5 || t == 6) dma_wait_keep_n<G::WPW + G::APW>(); else dma_wait_keep_n<G::WPW>(); }
                else { if (t == 5) dma_wait_keep_n<G::APW>(); else dma_wait(); }
                __syncthreads();
            }
        };
        const int nch = c_end - c_begin;
        int cc = 0;
#pragma unroll 1
        for (; cc + 2 <= nch; cc += 2) { chunk(i0); chunk(i1); }
        if (cc < nch) chunk(i0);
        dma_wait();                  // pieces staged past the end are still landing: the epilogue reuses this LDS
        __syncthreads();
    } else {
    if (total > 0) {
        load_patch(c_begin);
        load_w(w_r0, 0);
        load_w(w_r1, 1);
        store_patch(0, c_begin);
        store_w(w_r0, 0);
    }
    __syncthreads();

    // one tap: prefetch the weights of tap it+2 into `wl`, multiply tap `it` out of W buffer WB, park tap it+1 (`ws`)
    auto step = [&](int it, auto WBc, uint4 (&wl)[W_VECS], const uint4 (&ws)[W_VECS]) {
        constexpr int WB = decltype(WBc)::value;
        const int cc = it / 9, t = it - cc * 9;
        const int c = c_begin + cc;
        const int ab = cc & 1;
        if (t == 0) load_patch(c + 1);       // (before the weight prefetch: vmcnt retires in order)
        load_w(wl, it + 2);
        if (wave_live) {
            const int tap_off = ((t / 3) * PW + (t % 3)) * RSTR;
            const char* As = Abuf + ab * A_BYTES + tap_off;
            const char* Ws = Wbuf + WB * W_BYTES;
#pragma unroll
            for (int kk = 0; kk < CK / 16; ++kk) {
                uint4 wf[2], xf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) wf[a] = *reinterpret_cast<const uint4*>(Ws + w_frag + a * 32 * RSTR + kk * 32);
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) xf[bb] = *reinterpret_cast<const uint4*>(As + a_frag[bb] + kk * 32);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) acc[a][bb] = E::mfma(wf[a], xf[bb], acc[a][bb]);
            }
        }
        store_w(ws, WB ^ 1);
        if (t == 8 && c + 1 < c_end) store_patch(ab ^ 1, c + 1);
        __syncthreads();
    };
#pragma unroll 1
    for (int it = 0; it < total; it += 2) {
        step(it, std::integral_constant<int, 0>{}, w_r0, w_r1);
        if (it + 1 < total) step(it + 1, std::integral_constant<int, 1>{}, w_r1, w_r0);
    }
    }      // !DMA

    // ---- epilogue (same scheme as conv_gemm.hip): one 64-pixel wave-row group at a time through LDS ----
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CPR = BN / 8;
    constexpr int CHUNKS = EROWS * CPR;
    const int HW = H * W;
    float* slab = (p.split_k > 1) ? p.splitk_ws + (size_t)split * p.M * p.N : nullptr;
    // bias / time-embedding vector of this thread's 8 columns, fetched once (see epilogue8): the tile lies in ONE image
    float4 col_pre0 = make_float4(0, 0, 0, 0), col_pre1 = col_pre0;
    bool use_col_pre = false;
    if (slab == nullptr && (p.bias || p.rowvec)) {
        const int n = n0 + (tid % CPR) * 8;
        if (n < p.N) {
            load_col_addends(p, p.rowvec ? b : -1, n, (n + 8 <= p.N) ? 8 : 4, col_pre0, col_pre1);
            use_col_pre = true;
        }
    }
    // GroupNorm statistics of the output tile (gn_stats_out): a thread keeps the same 8 channels for every row it emits, so it
    // accumulates (sum, sum of squares) of its final values for the at most two groups those channels belong to
    const bool want_stats = p.gn_stats_out != nullptr && slab == nullptr;
    const int cpg = want_stats ? p.N / p.gn_stats_groups : 1;
    const int st_n = n0 + (tid % CPR) * 8;
    const int st_split = min(8, (st_n / cpg + 1) * cpg - st_n);       // channels [0, split) of the chunk -> its first group
    float st[4] = {0.f, 0.f, 0.f, 0.f};
    // (round 5) The residual of ALL eight chunks this thread emits is fetched before its first store.  A load issued between two stores makes
    // the wait for the load a wait for the store (vmcnt counts in order): the loop form load -> compute -> store paid a store round trip per chunk
    // (DESIGN section 6, round-5 findings).  CHUNKS == 4 * 256: every thread owns chunks tid + 256 i of both 64-row passes.
    static_assert(CHUNKS == 4 * 256, "the epilogue's residual prefetch assumes four chunks per thread and pass");
    const bool pre_res = slab == nullptr && p.res != nullptr;
    uint4 rpre[2][4];
    auto chunk_geom = [&](int wr, int i, int& m, int& n) -> bool {
        const int ch = tid + 256 * i;
        const int row = ch / CPR, cc = (ch - row * CPR) * 8;
        const int q = wr * EROWS + row;
        const int oy = y0 + q / TW, ox = x0 + q % TW;
        m = (b * H + oy) * W + ox;
        n = n0 + cc;
        return n < p.N && oy < H && ox < W;
    };
    if (pre_res) {
#pragma unroll
        for (int wr = 0; wr < 2; ++wr)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int m, n;
                rpre[wr][i] = make_uint4(0, 0, 0, 0);
                if (chunk_geom(wr, i, m, n) && n + 8 <= p.N) rpre[wr][i] = *reinterpret_cast<const uint4*>(p.res + (size_t)m * p.res_ld + n);
            }
    }
#pragma unroll
    for (int wr = 0; wr < 2; ++wr) {
        if ((wave >> 1) == wr) {
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* dst = Cs + (bb * 32 + cpix) * CLD + wn0 + a * 32 + 8 * j + 4 * hi;
                        *reinterpret_cast<float4*>(dst) = make_float4(acc[a][bb][4 * j], acc[a][bb][4 * j + 1], acc[a][bb][4 * j + 2], acc[a][bb][4 * j + 3]);
                    }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = tid + 256 * i;
            const int row = ch / CPR, cc = (ch - row * CPR) * 8;
            int m, n;
            if (!chunk_geom(wr, i, m, n)) continue;
            const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * CLD + cc);
            const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * CLD + cc + 4);
            if (slab) {
                slab_store8(slab, (size_t)m * p.N + n, v0, v1, n + 8 <= p.N, p.splitk_counters != nullptr);
            } else {
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                const int nv = (n + 8 <= p.N) ? 8 : 4;
                epilogue8<F16>(p, v, m, n, nv, HW, use_col_pre, col_pre0, col_pre1, pre_res && nv == 8, rpre[wr][i]);
                if (want_stats) {               // v now holds the final values (bias / vector / residual / activation applied)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (e < nv) {
                            // statistics of the STORED tensor (the value after rounding to 16 bits): what the standalone gn_stats_kernel and the
                            // reference's GroupNorm see, so the result does not depend on which kernel produced the tensor
                            const float r = E::tof(E::fromf(v[e]));
                            if (e < st_split) { st[0] += r; st[1] += r * r; }
                            else { st[2] += r; st[3] += r * r; }
                        }
                    }
                }
            }
        }
        if (wr == 0) __syncthreads();
    }
    if (want_stats) {
        __syncthreads();                        // the fp32 tile in LDS is dead: reuse its head for the 256 x 4 partials
        float* red = reinterpret_cast<float*>(smem);
        *reinterpret_cast<float4*>(red + tid * 4) = make_float4(st[0], st[1], st[2], st[3]);
        __syncthreads();
        const int G = p.gn_stats_groups;
        if (tid < G) {                          // fixed summation order: column chunk, then row lane (deterministic)
            const int g = tid;
            float S = 0.f, Q = 0.f;
            for (int j = 0; j < CPR; ++j) {
                const int nj = n0 + 8 * j;
                if (nj >= p.N) break;
                const int gj = nj / cpg;
                if (gj == g || gj + 1 == g) {
                    const int o = (gj == g) ? 0 : 2;
                    for (int rl = 0; rl < 256 / CPR; ++rl) { S += red[(j + CPR * rl) * 4 + o]; Q += red[(j + CPR * rl) * 4 + o + 1]; }
                }
            }
            const int nparts = tiles_y * tiles_x * n_tiles;
            const int part = (ty * tiles_x + tx) * n_tiles + tile_n;
            float* dst = p.gn_stats_out + (((size_t)b * nparts + part) * G + g) * 2;
            dst[0] = S; dst[1] = Q;
        }
    }
    // K slices summed in-kernel by the tile's last-arriving workgroup (gemm_common.h::splitk_last_arrival)
    if (slab != nullptr && p.splitk_counters != nullptr) {
        if (splitk_last_arrival(p.splitk_counters, tile_id, p.split_k, tid)) {
            for (int ch = tid; ch < TH * TW * CPR; ch += 256) {
                const int q = ch / CPR, cc = (ch - q * CPR) * 8;
                const int oy = y0 + q / TW, ox = x0 + q % TW;
                const int n = n0 + cc;
                if (n >= p.N || oy >= H || ox >= W) continue;
                const int m = (b * H + oy) * W + ox;
                const int nv = (n + 8 <= p.N) ? 8 : 4;
                float v[8];
                splitk_sum8(p, m, n, nv, v);
                epilogue8<F16>(p, v, m, n, nv, HW);
            }
        }
    }
}

}  // namespace

// statistic partials per image written through gn_stats_out (0: this launch cannot produce them)
int imd_conv_patch_stats_parts_of(const ConvGemmParams& p) {
    if (!imd_conv_patch_supported(p) || p.split_k > 1 || p.out_f32 || p.gn_stats_groups <= 0 || p.gn_stats_groups > 64 ||
        p.N % p.gn_stats_groups || (p.N / p.gn_stats_groups) < 8)
        return 0;
    return ((p.Hout + TH - 1) / TH) * ((p.Wout + TW - 1) / TW) * ((p.N + BN - 1) / BN);
}

bool imd_conv_patch_supported(const ConvGemmParams& p) {
    // the fused nearest-2x upsample (Upsample2D: interpolate -> conv) is a source-pixel map of the LDS-DMA staging only
    const bool geom = p.ups ? (p.Hout == 2 * p.Hin && p.Wout == 2 * p.Win && p.gn_a == nullptr && !(g_gemm_flags & 512))
                            : (p.Hin == p.Hout && p.Win == p.Wout);
    return p.taps == 9 && p.stride == 1 && !p.pad_br_only && geom && p.Hout >= TH && p.Wout >= TW && (p.Cin % CK) == 0 &&
           p.mode == OUT_ROWMAJOR && p.act != ACT_GEGLU;
}

// tile config 29: the LDS-DMA form with 64-channel chunks (128-byte rows)
bool imd_conv_patch64_supported(const ConvGemmParams& p) {
    return imd_conv_patch_supported(p) && (p.Cin % 64) == 0 && p.gn_a == nullptr && !(g_gemm_flags & 512) && p.x_bytes < 0x80000000u && p.w_bytes < 0x80000000u;
}

int imd_launch_conv_patch64(const ConvGemmParams& p, hipStream_t s) {
    if (!imd_conv_patch64_supported(p)) return imd_set_error("conv_patch (128-byte rows): unsupported problem (needs 3x3 stride 1, H >= 8, W >= 16, Cin %% 64 == 0, operands < 2 GiB, no fused GroupNorm)");
    const bool h = p.dtype == IMD_DTYPE_F16;
    typedef void (*kern_t)(const ConvGemmParams);
    const kern_t kern = h ? conv3x3_patch_kernel<true, true, 64> : conv3x3_patch_kernel<false, true, 64>;
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), PD<64>::LDS, "conv_patch")) return rc_attr;
    const int B = p.M / (p.Hout * p.Wout);
    const long blocks = (long)B * ((p.Hout + TH - 1) / TH) * ((p.Wout + TW - 1) / TW) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)p.split_k), dim3(256), PD<64>::LDS, s, p);
    return imd_check_launch("conv_patch (128-byte rows)");
}

int imd_launch_conv_patch(const ConvGemmParams& p, hipStream_t s) {
    if (!imd_conv_patch_supported(p)) return imd_set_error("conv_patch: unsupported geometry (needs 3x3 stride 1, H >= 8, W >= 16, Cin %% 32 == 0)");
    const bool h = p.dtype == IMD_DTYPE_F16;
    // LDS-DMA staging of both operands unless the fused GroupNorm prologue (values needed in registers) is asked for, or tuning knob 2
    // bit 9 selects the round-1/2 register-staged form (A/B)
    const bool dma = p.gn_a == nullptr && !(g_gemm_flags & 512) && (p.ups || (p.x_bytes < 0x80000000u && p.w_bytes < 0x80000000u));   // (the DMA loop marks out-of-range pieces with offset 2^31)
    if (p.ups && (p.x_bytes >= 0x80000000u || p.w_bytes >= 0x80000000u)) return imd_set_error("conv_patch: fused upsample needs operands < 2 GiB");
    typedef void (*kern_t)(const ConvGemmParams);
    const kern_t kern = dma ? (h ? conv3x3_patch_kernel<true, true> : conv3x3_patch_kernel<false, true>)
                            : (h ? conv3x3_patch_kernel<true, false> : conv3x3_patch_kernel<false, false>);
    const int lds = dma ? PATCH_DMA_LDS : PATCH_LDS;
    if (int rc_attr = imd_lds_attr(reinterpret_cast<const void*>(kern), lds, "conv_patch")) return rc_attr;
    const int B = p.M / (p.Hout * p.Wout);
    const long blocks = (long)B * ((p.Hout + TH - 1) / TH) * ((p.Wout + TW - 1) / TW) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)p.split_k), dim3(256), lds, s, p);
    return imd_check_launch("conv_patch");
}
