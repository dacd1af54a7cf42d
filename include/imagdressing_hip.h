/* imagdressing_hip.h -- C ABI of libimagdressing_hip.so (MI355X / gfx950 only).
 *
 * The drop-in boundary of the IMAGDressing-v1 denoising hot path.  The reference is pure Python
 * on this path (its device code is whatever torch dispatches to), so the "FFI" a maintainer binds
 * is ctypes: see INTEGRATION.md for the stub.  Every entry point below names the reference
 * interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - All pointers are DEVICE pointers into HBM unless stated otherwise; 16-bit tensors (bf16 or fp16, chosen
 *     per call by `dtype`) are passed as uint16_t*.  Inputs are borrowed and must stay alive until the stream work completes.
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); kernels are only
 *     enqueued, never synchronised; no allocation happens inside any call (hipGraph-capturable).
 *   - Return value: 0 on success, non-zero on error; imd_last_error() returns the message
 *     (thread-local).  There is NO CPU fallback: argument/shape/arch errors are reported, never
 *     papered over.
 *   - Activations are NHWC / token-major ([B, H*W, C]) bf16; weights are [N][K] bf16 with
 *     k = (ky*3 + kx) * Cin + ci for 3x3 convolutions.
 *   - ABI v8: the first field of every parameter block is `struct_bytes` = sizeof(the struct) as the CALLER compiled it.  An
 *     entry point that receives any other value returns an error instead of reading fields the caller never wrote (the v7 blocks
 *     grew at the tail; a binding that mirrored an older header handed the library a truncated struct).  Zero-initialise the
 *     block, set struct_bytes, then fill what you need: every optional pointer is NULL-off.
 */
#ifndef IMAGDRESSING_HIP_H
#define IMAGDRESSING_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMD_ABI_VERSION 9

enum { IMD_ACT_NONE = 0, IMD_ACT_SILU = 1, IMD_ACT_GEGLU = 2, IMD_ACT_GELU = 3, IMD_ACT_QUICK_GELU = 4 /* x * sigmoid(1.702 x): CLIP text MLP */ };
enum { IMD_OUT_ROWMAJOR = 0, IMD_OUT_HEADS = 1 };
/* 16-bit element type of activations and weights (both run v_mfma_f32_32x32x16_* at the same rate) */
enum { IMD_DTYPE_BF16 = 0, IMD_DTYPE_F16 = 1 };

/* destination of one split (Q, K or V) of a head-split projection */
typedef struct imd_heads_dest {
    uint16_t* ptr; /* NULL: drop this split */
    int kind;      /* 0: [B, H, L, DP] (rows = tokens)   1: [B, H, DP, L] (rows = head dim, keys contiguous) */
    int DP;        /* kind 0: padded head dim (row length)   kind 1: rows per head */
    int L;         /* kind 0: tokens per batch entry         kind 1: padded token count (row length) */
    float scale;   /* multiplied in fp32 before rounding (softmax scale * log2 e folded into Q) */
} imd_heads_dest;

/* Per-call tuning (ABI v8; tag widened in v9): `flags` of imd_conv_gemm_params / imd_attn_params is an INPUT.  0 (what a zero-initialised
 * block carries) = use the process-wide knobs of imd_set_tuning().  A caller whose flags carry the 8-bit tag IMD_TUNING_PER_CALL in bits
 * 24..31 -- (flags & IMD_TUNING_TAG_MASK) == IMD_TUNING_PER_CALL -- chooses for THIS call only:
 *   imd_conv_gemm:  bits 0..4 = bits 0..4 of tuning knob 2 (tap-inner K order, weight loads past L1, XCD-aware tile order, grouped order)
 *   imd_attention:  bits 0..7 = head-dim-40 kernel variant (tuning knob 0; 0 = the process-wide value; validated against the same range
 *                   as imd_set_tuning(0, .) of this build), bit 8 = 1: plain work order (tuning knob 1 = 0)
 * so two pipelines in one process can run different settings without touching global state (imagdressing_amd.ops.tuning_scope).  Any
 * other non-zero value in bits 24..31 is refused (v9): a block whose flags were left uninitialised does not silently select a tuning.
 * The fp8 attention, row-resident projection and fused feed-forward entry points have no per-call choice and ignore the field. */
#define IMD_TUNING_PER_CALL 0x5A000000
#define IMD_TUNING_TAG_MASK 0xFF000000u

typedef struct imd_conv_gemm_params {
    uint32_t struct_bytes; /* sizeof(imd_conv_gemm_params) in the caller's view (ABI v8); checked on entry */
    const uint16_t* x; /* NHWC activations (pixel stride x_pix_stride) or [M, K] rows */
    const uint16_t* w; /* [N][K] */
    void* out;         /* bf16 (fp32 when out_f32) [M, out_ld] */
    int M, N, K;
    int Cin, taps, Hin, Win, Hout, Wout, stride, ups;
    int x_pix_stride;
    int out_ld, res_ld;
    const float* bias;   /* [N] or NULL */
    const float* rowvec; /* per-batch vector [B][rowvec_stride] added to every pixel of batch b, or NULL */
    int rowvec_stride;   /* floats between the vectors of consecutive batch entries; 0 = ONE vector for the whole batch (a shared timestep) */
    const uint16_t* res; /* residual [M, res_ld] or NULL */
    float out_scale;     /* applied after bias/rowvec, before the residual add */
    int act;             /* IMD_ACT_* */
    int out_f32;
    int mode;            /* IMD_OUT_* */
    int hC, hH, hD;      /* head split: channels per split, heads, head dim */
    imd_heads_dest hd[3];
    int dtype;           /* IMD_DTYPE_* */
    int split_k;         /* K slices (<= 1: none); > 1 needs splitk_ws and a row-major epilogue */
    float* splitk_ws;    /* split_k * M * N floats of scratch */
    uint32_t x_bytes, w_bytes; /* filled in by the library (buffer-descriptor extents) */
    int flags;           /* filled in by the library (tuning bits); on entry: 0, or IMD_TUNING_PER_CALL | per-call tuning bits (see above) */
    /* fused GroupNorm(+SiLU) prologue (cfg 5 only): x is normalised as x*gn_a[b][c] + gn_b[b][c] (then SiLU when
     * gn_silu) while it is staged; coefficients come from imd_groupnorm_coeffs().  NULL: plain convolution. */
    const float* gn_a;
    const float* gn_b;
    int gn_silu;
    int pad_br_only;     /* 3x3 taps: 0 = symmetric zero padding 1 (UNet); 1 = padding on the bottom / right edge only, i.e.
                          * F.pad(x, (0, 1, 0, 1)) + conv(padding = 0): the stride-2 Downsample2D of the VAE encoder */
    int* splitk_counters; /* split_k > 1 only.  NULL: the K slices are summed by a second launch (fixed order).  Otherwise >=
                          * IMD_SPLITK_COUNTERS ints that are ZERO on entry and are left zero: every output tile's last-arriving
                          * workgroup sums the slices itself, in the same fixed order (bit-identical results, one launch less) */
    /* GroupNorm statistics of the OUTPUT (row-major 16-bit output, N % gn_stats_groups == 0, >= 8 channels per group): the fp32 (sum, sum of
     * squares) per group of the FINAL (rounded) values, written to gn_stats_out[((b * nparts + part) * G + g) * 2].  Size the buffer with
     * nparts = imd_conv_gemm_stats_parts(p, cfg) -- NOT imd_conv_patch_stats_parts() -- because the layout depends on who writes it:
     *   - tile configs 5 / 22 / 23 / 29 with split_k == 1: the halo-patch kernel's own epilogue, part = (pixel tile of the image) * n_tiles +
     *     channel tile; groups outside a tile get zeros;
     *   - ANY K-sliced launch that finishes with the second launch (split_k > 1, splitk_counters == NULL): the finish launch, part = a block
     *     of consecutive pixels of image b (all channels), nparts = ceil(HW / rows per part).
     * imd_conv_gemm_stats_parts() answers 0 when this (p, cfg) cannot produce them; a launch that asks anyway is refused.  The next
     * imd_groupnorm on that tensor passes the buffer as `partial` with `nparts` and skips its statistics pass (ResnetBlock2D: conv1 -> norm2,
     * conv2 -> the next block's norm). */
    float* gn_stats_out;
    int gn_stats_groups;
    /* GroupNorm (+ SiLU) of the OUTPUT inside the finish launch of a K-sliced problem (ABI v9; imd_conv_gemm_gn_out_supported()): when
     * gn_out_gamma is given, `out` receives  act(GroupNorm_G(epilogue(sum of the K slices)))  instead of the epilogue's result -- the tensor
     * ResnetBlock2D.norm2 + nonlinearity would produce from conv1's output, which nothing else reads (diffusers-0.24 ResnetBlock2D.forward:
     * conv1 -> + temb -> norm2 -> SiLU -> conv2).  One workgroup owns all pixels of one (image, group), so the statistics are complete in
     * the launch: the un-normalised tensor never exists in memory and the separate imd_groupnorm launch disappears.  Statistics are those of
     * the value ROUNDED to the element type (what imd_groupnorm would read back), fp32, fixed summation order.  Needs split_k > 1 without
     * splitk_counters, a row-major 16-bit output without activation / GEGLU, N % groups == 0 with 4 | N / groups, and
     * Hout * Wout * (N / groups) <= 12288 (the 16x16 and 8x8 levels); gn_stats_out must be NULL (there is nothing left to normalise). */
    /* GroupNorm (+ SiLU) of the INPUT rows inside the row-resident projections (ABI v9; imd_row_linear_gn_in_supported(p, cfg); tile configs
     * 12 / 13 / 14): Transformer2DModel.norm -> proj_in as ONE launch -- the normalised tensor never exists in memory.  gn_in_partial holds the
     * (sum, sum of squares) partials of x exactly as imd_groupnorm takes them (`partial` + `nparts`: [B][gn_in_nparts][gn_in_groups][2], written by
     * the producer of x through gn_stats_out or by a statistics pass); every workgroup folds its image's partials in imd_groupnorm's order and
     * applies  y = x * (gamma rstd) + (beta - mean gamma rstd)  (+ SiLU) to its rows in registers, rounded to the element type -- bit-identical
     * to imd_groupnorm followed by the same projection.  Needs K = Cin = 320 / 640 / 1280, K % gn_in_groups == 0, gn_in_groups <= 64 and
     * Hout * Wout a multiple of the kernel's row block (128 / 128 / 64). */
    const float* gn_in_partial; /* NULL: off */
    const float* gn_in_gamma;   /* [K] */
    const float* gn_in_beta;    /* [K] */
    int gn_in_nparts, gn_in_groups, gn_in_silu;
    float gn_in_eps;
    const float* gn_out_gamma; /* [N] or NULL */
    const float* gn_out_beta;  /* [N] */
    float gn_out_eps;
    int gn_out_silu;
    int gn_out_groups;
    /* PERIODIC residual (ABI v9, the K = 320 row-resident projection -- tile config 12 -- only; any other kernel refuses a non-zero value):
     * `res` holds res_rows rows and output row m adds res[m % res_rows].  The first hybrid block of a CFG batch, whose block input is the same
     * for the cond and the uncond rows (one copy kept): attn1's out-projection and Transformer2DModel.proj_out add it to both halves without a
     * materialised torch.cat([x, x]).  Needs res_rows % 128 == 0 and M % res_rows == 0; 0 = one residual row per output row. */
    int res_rows;
} imd_conv_gemm_params;
#define IMD_SPLITK_COUNTERS 16384

typedef struct imd_attn_params {
    uint32_t struct_bytes; /* sizeof(imd_attn_params) in the caller's view (ABI v8); checked on entry */
    const uint16_t* q;   /* [B, H, N, DPK], pre-scaled by D^-1/2 * log2(e) */
    const uint16_t* k1;  /* [B1, H, L1, DPK] */
    const uint16_t* v1t; /* [B1, H, DPV, L1P] */
    const uint16_t* k2;  /* optional second key/value set (garment tokens / IP tokens) or NULL */
    const uint16_t* v2t;
    const float* scale2; /* [B] weight of the second softmax branch per batch entry (0 => skipped) */
    uint16_t* out;       /* [B, N, out_ld] with head h at columns h*D .. */
    int B, H, N, D;
    int L1, L1P, kv1_bdiv; /* kv batch entry of batch b is b / kv1_bdiv */
    int L2, L2P, kv2_bdiv;
    int out_ld;
    int dtype;
    int flags;           /* filled in by the library (tuning bits); on entry: 0, or IMD_TUNING_PER_CALL | per-call tuning bits (see above) */
    int causal;          /* 1: query i attends keys 0..i of the first key set only (CLIP text encoder); needs k2 == NULL, D != 40 */
    int k_pad_one;       /* 1: the caller guarantees that pad column D of EVERY K row (k1 and k2) holds 1.0 (D < DPK only, i.e.
                          * D = 40: the slot through which the deferred row maximum enters the QK^T MFMA).  The kernels that
                          * stage K through registers write that 1 themselves (0 or 1 in memory are both fine); with the
                          * guarantee the d = 40 kernel may stage K / V^T by LDS-DMA, which cannot patch data in flight. */
    /* Fused out-projection (ABI v7; head dim 40 with H * D == 320 and N >= 512 only -- the UNet's 64x64-level blocks): when proj_w is
     * given the launch also computes  proj_out = proj_res + proj_b + out . proj_w^T  (Attention.to_out[0] + the block residual,
     * attention_processor.py:614-622), i.e. the hybrid block's third launch disappears.  `out` is still written (it is the hand-off
     * buffer between the heads of a row block: stored write-through, read back by the head that finishes last). */
    const uint16_t* proj_w;   /* NULL, or [H*D, H*D] row-major (out channel, in channel) */
    const float* proj_b;      /* [H*D] or NULL */
    const uint16_t* proj_res; /* [B, N, proj_res_ld] or NULL */
    uint16_t* proj_out;       /* [B, N, proj_out_ld] */
    int proj_res_ld, proj_out_ld;
    int* proj_counters;       /* >= B * ceil(N / 256) ints, ZERO on entry, left zero (one per batch entry and 256-row block) */
    /* Duplicated first-phase output (ABI v9; imd_attention_dup_supported(): head dim 40, N >= 512, k_pad_one, no causal mask, no fused
     * out-projection): when out_dup is given, softmax(Q K1^T) V1 of batch entry b -- rounded to the element type exactly as a row
     * WITHOUT a second key set stores it -- is also written to out_dup[b, :, :] (same out_ld).  For the first hybrid block of a CFG
     * batch, where the cond and uncond rows of an image still have identical Q / K / V: one launch over the B cond rows produces the
     * cond rows (`out`) and the uncond rows (`out_dup`), bit-identical to a 2B-row launch (RefSAttnProcessor2_0 with and without
     * sa_hidden_states on the same hidden states, attention_processor.py:589-612). */
    uint16_t* out_dup;
    /* Phase-split launch of the hybrid attention (ABI v9; imd_attention_phase_split_supported(): head dims 64 / 80 / 160, k2 given, no causal mask):
     * the two softmaxes of a row with a second key set are independent until their results are added, and at the 32x32 / 16x16 / 8x8 levels a
     * launch has too few workgroups to hide one row's 2 x L / 64 sequential key tiles.  With phase2_rows = R > 0 the rows [0, R) are the ones
     * whose scale2 is non-zero (the cond half of a CFG batch) and the grid grows to B + R batch entries: entry b < B runs the FIRST softmax of
     * row b only and stores it as a row without a second key set would; entry B + r runs the SECOND softmax of row r and stores
     * scale2[r] * softmax(Q K2^T) V2 as fp32 to phase2_out[r, :, :] ([R, N, H * D]); a second, elementwise launch then writes
     * out[r] = round(out[r] + phase2_out[r]) -- the same arithmetic as the one-workgroup form (first result rounded to the element type,
     * second added in fp32, rounded once): bit-identical, with every workgroup half as long and 1.5x as many of them. */
    float* phase2_out;   /* NULL: off */
    int phase2_rows;
} imd_attn_params;

/* Fused feed-forward of a transformer block on the 64x64 level (ff_fused.hip), C = 320, inner = 1280:
 *   out = x + b2 + W2 . geglu(W1 . LN(x) + b1)         (norm3 -> ff.net[0] GEGLU -> ff.net[2] -> + residual, ONE launch)
 * w1 / b1 / w2 are PACKED by the host (imagdressing_amd/ops.py::pack_ff_fused; layout documented there and in ff_fused.hip):
 *   w1 [80 blocks][32 rows][320]: block b holds inner channels 16 b .. 16 b + 15; row i is the value row (bit 3 of i clear)
 *      or the gate row (bit 3 set) of inner channel 16 b + (i & 7) + 8 (i >> 4); LayerNorm gamma folded in (W1 diag(gamma));
 *   b1 [80][32] fp32 in the same order, LayerNorm beta folded in (b1 + W1 beta);
 *   w2 [40 chunks][320 rows][32]: chunk c holds inner channels 32 c .. 32 c + 31 as two 16-groups whose members are stored in
 *      the order 0-3, 8-11, 4-7, 12-15 (the register order of the GEGLU outputs of a lane). */
typedef struct imd_ff_params {
    uint32_t struct_bytes; /* sizeof(imd_ff_params) in the caller's view (ABI v8); checked on entry */
    const uint16_t* x;   /* [M, x_ld] block input (un-normalised when ln) -- also the residual */
    const uint16_t* w1;
    const float* b1;
    const uint16_t* w2;
    const float* b2;     /* [C] */
    uint16_t* out;       /* [M, out_ld] */
    int M, C, inner;
    int x_ld, out_ld;
    int ln;              /* 1: LayerNorm without affine (eps = ln_eps) on every row of x first */
    float ln_eps;
    int dtype;
} imd_ff_params;

typedef struct imd_groupnorm_params {
    uint32_t struct_bytes; /* sizeof(imd_groupnorm_params) in the caller's view (ABI v8); checked on entry */
    const uint16_t* x; uint16_t* y; const float* gamma; const float* beta;
    float* partial;      /* workspace of imd_groupnorm_workspace_floats() floats */
    int B, HW, C, G, x_ld, y_ld;
    float eps;
    int silu;
    int dtype;
    int nparts;          /* 0: imd_groupnorm computes the statistics itself (two launches).  > 0: `partial` already holds
                          * [B][nparts][G][2] fp32 (sum, sum of squares) partials of x written by the PRODUCER of x -- the
                          * convolution epilogue, imd_conv_gemm_params.gn_stats_out -- and only the normalise pass is launched */
} imd_groupnorm_params;

typedef struct imd_layernorm_params {
    uint32_t struct_bytes; /* sizeof(imd_layernorm_params) in the caller's view (ABI v8); checked on entry */
    const uint16_t* x; uint16_t* y; const float* gamma; const float* beta;
    int rows, C, x_ld, y_ld;
    float eps;
    int dtype;
} imd_layernorm_params;

typedef struct imd_ddim_params {
    uint32_t struct_bytes; /* sizeof(imd_ddim_params) in the caller's view (ABI v8); checked on entry */
    float* z;             /* [B, HW, 4] fp32 latent state, updated in place */
    const float* eps;     /* [2B, HW, 4] fp32: rows [0,B) cond pass, [B,2B) uncond pass */
    uint16_t* x_next;     /* [2B, HW, 8] bf16 next UNet input (channels 4..7 = 0) or NULL */
    int B, HW;
    float guidance, sqrt_a_t, sqrt_1m_a_t, sqrt_a_prev, sqrt_1m_a_prev;
    const float* mask;    /* [B, HW] inpaint mask or NULL */
    const float* z_img;   /* [B, HW, 4] */
    const float* noise;   /* [B, HW, 4] */
    float sqrt_a_next, sqrt_1m_a_next;
    int dtype;            /* element type of x_next */
    const float* coefs;   /* NULL, or DEVICE pointer to 6 fp32 {sqrt_a_t, sqrt_1m_a_t, sqrt_a_prev, sqrt_1m_a_prev, sqrt_a_next,
                           * sqrt_1m_a_next} read by the kernel INSTEAD of the host scalars above: lets one captured HIP graph of a
                           * denoising step serve every timestep (the host refreshes 24 bytes per step, stream-ordered) */
    const float* var_noise; /* NULL, or [B, HW, 4] fp32 standard-normal noise of the stochastic DDIM step (eta > 0, DDIMScheduler.step's
                           * `variance_noise`): z_prev += sigma * var_noise BEFORE the inpaint blend; the caller then passes
                           * sqrt_1m_a_prev = sqrt(1 - a_prev - sigma^2) (the direction coefficient of diffusers' step) */
    float sigma;          /* eta * sqrt((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)); read only with var_noise */
} imd_ddim_params;

/* library / device */
int imd_abi_version(void);
const char* imd_last_error(void);
/* 0 iff `device` is a gfx950 part this library was built for. */
int imd_device_check(int device);

/* Implicit-GEMM convolution / linear layer with fused epilogue.
 * Replaces (diffusers==0.24.0, called from dressing_sd/pipelines/IMAGDressing_v1_pipeline.py:466,499,511):
 * ResnetBlock2D.conv1/conv2/conv_shortcut/time_emb_proj, Down/Upsample2D.conv, Transformer2DModel.proj_in/out,
 * Attention.to_q/to_k/to_v/to_out[0], FeedForward (GEGLU), TimestepEmbedding, ControlNet zero-convs; and
 * RefSAttnProcessor2_0.to_k_ref/to_v_ref (adapter/attention_processor.py:600-601), to_k_ip/to_v_ip (:841-842),
 * the nn.Linear layers of adapter/resampler.py.  cfg (rows x channels x K-depth of a workgroup tile): -1 auto, 0: 128x128x64, 1: 128x64x64, 2: 64x64x64,
 * 3: 64x64x64 with 4 K tiles in flight, 4: 128x128x32, 5: 3x3 halo-patch kernel, 6: 64x320x32 (N % 320 == 0: no idle columns),
 * 7: 64x64x32 / 4 in flight, 8: 128x128x32 / 4 in flight, 9: 256x128x32, 10: 256x256x32 and 11: 128x320x64 (8-wave workgroups:
 * fewer operand bytes fetched per MFMA for the wide projections; 11 covers N = 320 k with full-width row blocks); 12..15: row-resident
 * kernels; 16: 256x256x64 and 17: 128x128x32 with a three-stage ring, both operands staged by LDS-DMA (plain linear layers, K % 64 == 0);
 * 18: the gathering form of 17 for 3x3 convolutions (stride 1 | 2, Cin % 32 == 0; K slices allowed for 17 and 18);
 * 19 / 20: 17 / 18 with a four-stage ring (64 KB, two workgroups per CU, three tiles of lead);
 * 21: the halo-patch kernel with 16 x 16 pixel tiles (256 pixels x 128 channels per workgroup, wave tiles 128 x 64: half the weight bytes per MAC);
 * 22: the halo-patch kernel with 8 x 16 pixels x 160 channels per workgroup (wave tiles 32 x 160: N = 320 k runs without idle waves or padded MFMAs);
 * 23: 22 with eight waves = 16 x 16 pixels x 160 channels per workgroup (one staged weight tile serves 256 pixels: 154 instead of 290 bytes of LDS-DMA per MFMA).
 * 25 / 27: the 128 x 128 LDS-DMA tiles of 17 / 19 with 128-BYTE rows (K tile 64), two / three ring stages: the L2 hands a CU whole 128-byte lines, 64-byte
 *     rows use half of each (tools/probes/staging_probe.hip); 26 / 28: the 3x3 gathering form of the same (one tap x 64 channels per tile, Cin % 64 == 0).
 * 29: the halo-patch kernel (5) with 64-channel chunks = 128-byte rows and a two-slot weight ring (80 KB of LDS, two workgroups per CU): measured slower
 *     than 5 on every shape of the UNet (the ring and the third workgroup matter more than the whole lines), kept as a tuning candidate.
 * 24: 3x3 stride-1 convolutions of maps 8 pixels wide (the UNet's lowest level): a workgroup owns all pixels of up to 8 images x 64 channels x one
 *     K slice, so every weight byte is fetched once; K-sliced only (split_k >= 2).
 * Results are identical up to fp32 summation order. */
int imd_conv_gemm(const imd_conv_gemm_params* p, int cfg, void* stream);
int imd_conv_gemm_auto_cfg(int M, int N);
/* suggested number of K slices for tile config `cfg` (1 = do not split) */
int imd_conv_gemm_auto_split(int M, int N, int K, int cfg);

/* Fused dual-softmax attention.  Replaces the two F.scaled_dot_product_attention calls + add of
 * RefSAttnProcessor2_0.__call__ (adapter/attention_processor.py:589-612), LoRAIPAttnProcessor2_0 (:833-856),
 * the single SDPA of CAttnProcessor2_0 (:277-279) / CacheAttnProcessor2_0 (:80-82), and
 * PerceiverAttention's softmax(QK^T)V (adapter/resampler.py:71-74). */
int imd_attention(const imd_attn_params* p, void* stream);
/* The same fused dual-softmax attention for head dim 40 on the MX block-scaled FP8 MFMA (BASELINE.json configs[4]: the
 * 768x576 ControlNet-inpainting configuration "with fp8 MFMA attention").  q / k1 / v1t / k2 / v2t of `p` point to e4m3
 * operands produced by imd_attn_quantize_fp8 (Q8, K8: [.., rows, 64] bytes; V8^T: [.., 64, LP] bytes, keys permuted inside
 * groups of 64); `out` and `dtype` are the 16-bit output.  eq / ek / ev: the power-of-two exponents the operands were
 * scaled by (q * 2^eq ...), 0 <= eq + ek <= 8.  Same reference lines as imd_attention (attention_processor.py:589-612). */
int imd_attention_fp8(const imd_attn_params* p, int eq, int ek, int ev, void* stream);
/* 16-bit head-split attention operands -> the e4m3 operands of imd_attention_fp8.
 * kind 0: Q or K rows [count, 48] -> [count, 64]: columns 0..39 scaled by 2^exp2_scale, columns 40 and 41 = pad_val
 *         (Q: 0; K: 2^(eq+ek), the slots that carry the deferred row maximum), the rest 0;
 * kind 1: V^T [count, 64, LP] -> [count, 64, LP] bytes (rows 0..39), scaled by 2^exp2_scale, key k of every 64-group stored
 *         at position 32 ((k >> 3) & 1) + 8 ((k >> 4) & 3) + (k & 7) (the order the kernel's packed P comes out in). */
int imd_attn_quantize_fp8(const uint16_t* src, uint8_t* dst, int kind, long count, int LP, int exp2_scale, float pad_val, int dtype,
                          void* stream);
/* 1 iff tile config cfg (12 / 13 / 14: the row-resident projections) can normalise its input rows as *p's gn_in_* fields ask */
int imd_row_linear_gn_in_supported(const imd_conv_gemm_params* p, int cfg);
/* 1 iff the finish launch of *p (split_k filled in as imd_conv_gemm will see it) can apply GroupNorm(gn_out_groups) (+ SiLU) to its output */
int imd_conv_gemm_gn_out_supported(const imd_conv_gemm_params* p);
/* 1 iff imd_attention accepts imd_attn_params.out_dup for this head count / query count / head dim (with k_pad_one = 1) */
int imd_attention_dup_supported(int H, int N, int D);
/* 1 iff imd_attention runs phase2_rows / phase2_out at this head dim (the generic kernel: 64, 80, 160) */
int imd_attention_phase_split_supported(int D);
/* padded head dims of the Q/K rows (dpk) and V^T rows (dpv) for head dim D */
int imd_attn_padded_dims(int D, int* dpk, int* dpv);
/* performance knobs (results are identical for every accepted setting up to fp32 summation order).  knob 0: head-dim-40 attention
 * kernel variant (default 13: the static-ring kernel 12 with the deferred-maximum overflow test on a phase's first and last steps
 * only and a workgroup-level re-run as 12 when a softmax denominator comes out non-finite -- bf16, fp16 runs 12; 12: the LDS-DMA
 * kernel with compile-time ring addresses; 10: software-pipelined kernel of attention_d40.hip with the P.V tail on
 * v_mfma_f32_16x16x32, LDS-DMA staging when imd_attn_params.k_pad_one; 11: the same with register staging; 9 / 7: the round-2 kernel with the same two staging
 * rules; 6, 8: scheduling variants of 7; 1..5: round-1 kernel shapes.  Values 20..39 (timing ablations that compute WRONG
 * results) are accepted only by a library built with -DIMD_ABLATIONS; a normal build rejects them);
 * knob 1: XCD-aware work mapping of the attention grid (0|1);
 * knob 2: GEMM operand-fetch bits (bit0: tap-inner K order for 3x3 convs, bit1: weight loads bypass the L1,
 * bit2: XCD-aware tile order -- each XCD's L2 owns whole row tiles or whole channel tiles, whichever moves fewer bytes;
 * bit4: row tiles visited in groups of 8 inside an XCD's range; A/B switches with identical results: bit8 row_linear's staged epilogue,
 * bit9 the halo-patch conv staged through registers, bit10 8-byte stores in the row kernels.  Bits 5..7 are timing ablations of
 * gemm_dma256.hip that compute WRONG results: compiled into, and accepted by, -DIMD_ABLATIONS builds only -- a normal build rejects
 * them, and any bit above 10). */
int imd_set_tuning(int knob, int value);
/* current value of a knob (-1: unknown knob): lets a harness snapshot and restore the process-global settings around an A/B */
int imd_get_tuning(int knob);

/* GroupNorm (+SiLU) over NHWC: diffusers ResnetBlock2D.norm1/norm2, Transformer2DModel.norm, conv_norm_out. */
int imd_groupnorm(const imd_groupnorm_params* p, void* stream);
int imd_groupnorm_workspace_floats(int B, int HW, int C, int G);
/* Statistics only: writes the normalisation of ResnetBlock2D.norm1/norm2 as per-(batch, channel) fp32 coefficients
 * coef_a[B][C], coef_b[B][C] (y = x*a + b; p->y is unused) for imd_conv_gemm's fused prologue (gn_a / gn_b, cfg 5). */
int imd_groupnorm_coeffs(const imd_groupnorm_params* p, float* coef_a, float* coef_b, void* stream);
/* 1 iff tile config 5 (LDS-resident halo patch: 3x3, stride 1, H % 8 == 0, W % 16 == 0, Cin % 32 == 0) can run *p. */
int imd_conv_patch_supported(const imd_conv_gemm_params* p);
/* 1 iff tile config 21 (the halo-patch kernel with 16 x 16 pixel tiles, conv_patch2.hip) takes this geometry */
int imd_conv_patch2_supported(const imd_conv_gemm_params* p);
/* 1 iff tile config 22 (the halo-patch kernel with 160-channel tiles, conv_patch3.hip: N = 320 k without idle waves) takes this geometry */
int imd_conv_patch3_supported(const imd_conv_gemm_params* p);
/* 1 iff tile config 23 (the same kernel with eight waves: 16 x 16 pixels x 160 channels per workgroup) takes this geometry */
int imd_conv_patch4_supported(const imd_conv_gemm_params* p);
/* 1 iff tile config 24 (conv_img.hip: whole maps 8 pixels wide x 64 channels x one K slice per workgroup, split_k >= 2) takes this problem;
 * split_k must be filled in as imd_conv_gemm will see it */
int imd_conv_img_supported(const imd_conv_gemm_params* p);
/* number of statistic partials per image the halo-patch kernel writes for this geometry (gn_stats_out), 0 if it cannot */
int imd_conv_patch_stats_parts(const imd_conv_gemm_params* p);
/* the same for ANY launch: partials per image that imd_conv_gemm(p, cfg) writes through gn_stats_out -- the halo-patch epilogue (cfg 5, no K
 * slices) or, with p->split_k > 1 and a separate finish launch (splitk_counters == NULL), the finish launch itself, which then sums the
 * slices, runs the epilogue AND emits the statistics of the tensor it stores (row-major 16-bit outputs, N / groups >= 8); the halo-patch
 * kernels with 160-channel tiles (cfg 22 / 23) emit them from their own epilogue like cfg 5; 0: none */
int imd_conv_gemm_stats_parts(const imd_conv_gemm_params* p, int cfg);
/* 1 iff tile config 16 (256 x 256 x 64 LDS-DMA tile kernel, gemm_dma.hip: plain linear layer, K % 64 == 0, no K split) can run *p. */
int imd_gemm_dma_supported(const imd_conv_gemm_params* p);

/* Row-resident linear layers: K = 320, N = 64..320 in steps of 64 (row_linear.hip, the 64x64-level token matrix; tile config
 * 12 of imd_conv_gemm) and K = 640 / 1280, N a multiple of 160 with a bias / scale / residual / head-split-Q
 * epilogue (row_linear_k640.hip / row_linear_k1280.hip, the 32x32 / 16x16 levels; tile configs 13 / 14); and the fused q / k / v
 * projection of a 320-channel self-attention layer (K = 320, N = 960, head-split epilogue with Q, K row-major and V transposed,
 * Hout * Wout a multiple of 128: row_qkv.hip, tile config 15 -- norm1 -> to_q / to_k / to_v as one launch when ln != 0).  Same parameter block and fused epilogue as imd_conv_gemm (taps = 1).  Every wave keeps its 32 token rows in registers, the weights stream through LDS by DMA.  ln != 0: LayerNorm
 * WITHOUT affine (eps = ln_eps) is applied to each row of x on the fly -- BasicTransformerBlock.norm2 -> attn2.to_q as one
 * launch; the caller folds the affine part into the layer: W' = W diag(gamma), b' = b + W beta. */
int imd_row_linear(const imd_conv_gemm_params* p, int ln, float ln_eps, void* stream);
int imd_row_linear_supported(const imd_conv_gemm_params* p);

/* BasicTransformerBlock.norm3 + ff (GEGLU feed-forward) + residual as one launch; see imd_ff_params. */
int imd_ff_geglu(const imd_ff_params* p, void* stream);

/* LayerNorm over the last dim: BasicTransformerBlock.norm1/2/3; adapter/resampler.py:16,43-44,199. */
int imd_layernorm(const imd_layernorm_params* p, void* stream);
/* Row softmax, fp32 in -> 16-bit out: p[r][c] = softmax_c(s[r][:cols]) (the upcast softmax of the single-head, d = 512
 * attention in the VAE mid block -- diffusers AttnProcessor on AutoencoderKL.{encoder,decoder}.mid_block.attentions.0,
 * reached from IMAGDressing_v1_pipeline.py:457-458,:544).  cols <= 16384, row strides in elements. */
int imd_softmax_rows(const float* s, int s_ld, uint16_t* p, int p_ld, int rows, int cols, int dtype, void* stream);

/* CFG combine + DDIM step (+ inpaint blend) + next UNet input:
 * IMAGDressing_v1_pipeline.py:483-488,521-532; ..._pipeline_controlnet_inpainting.py:487-500. */
int imd_ddim_cfg_step(const imd_ddim_params* p, void* stream);

/* diffusers Timesteps(dim, flip_sin_to_cos=True, freq_shift=0): out[B, dim] fp32 = [cos | sin]. */
int imd_timestep_embedding(const float* t, float* out, int B, int dim, void* stream);

/* out = a + b_scale * b over [rows, C] with row strides (ControlNet residual add,
 * ..._pipeline_ipa_controlnet.py:676-677,687-688; skip + residual). */
int imd_add(const uint16_t* a, int a_ld, const uint16_t* b, int b_ld, uint16_t* out, int out_ld, long rows, int C, float b_scale, int dtype, void* stream);
/* strided 2-D copy (channel concat of UNet skip connections). */
/* Token + position embedding lookup (CLIPTextEmbeddings): out[r][:] = table[ids[r]][:] + pos[r % T][:]; ids are int64. */
int imd_embed_tokens(const uint16_t* table, int vocab, const uint16_t* pos, int T, const int64_t* ids, uint16_t* out, long rows, int C,
                     int dtype, void* stream);
/* ViT sequence assembly (CLIPVisionEmbeddings): out[b][0] = cls + pos[0]; out[b][1 + p] = patches[b][p] + pos[1 + p]. */
int imd_vit_assemble(const uint16_t* patches, const uint16_t* cls, const uint16_t* pos, uint16_t* out, int B, int P, int C, int dtype,
                     void* stream);
/* out[i] = sum_j coefs[j] * xs[j][i] over n <= 8 fp32 tensors of `numel` elements (xs / coefs are HOST arrays; out may alias an
 * input): the latent arithmetic of the multistep UniPC sampler (x0 prediction with CFG, predictor and corrector updates). */
int imd_lincomb(const float* const* xs, const float* coefs, int n, float* out, long numel, void* stream);
int imd_copy2d(const uint16_t* a, int a_ld, uint16_t* out, int out_ld, long rows, int C, void* stream);
/* out[r, 0:Ca] = a[r, :], out[r, Ca:Ca+Cb] = b[r, :] (+ b_add[r, :] when given): torch.cat([x, skip (+ ControlNet residual)], dim=1) of
 * an up block (diffusers UNet2DConditionModel up path; ..._pipeline_ipa_controlnet.py:676-688) in one launch; contiguous rows.
 * b_rows: 0 (= rows), or a divisor of rows -- b then holds b_rows rows and row r reads b[r % b_rows] (one skip tensor serving both
 * halves of a CFG batch whose halves are identical up to that layer). */
int imd_concat2(const uint16_t* a, int Ca, const uint16_t* b, int Cb, const uint16_t* b_add, uint16_t* out, long rows, long b_rows, int dtype, void* stream);
/* imd_concat2 over B images of HW pixels that ALSO writes the GroupNorm(G) statistics of its output -- the ResnetBlock2D.norm1 that
 * follows every skip concatenation of the up path (diffusers unet_2d_blocks.py CrossAttnUpBlock2D / UpBlock2D.forward: cat, then resnet) --
 * as partial[B][imd_groupnorm_parts(B, HW, Ca + Cb)][G][2] fp32 (sum, sum of squares), the layout imd_groupnorm_params.nparts takes.
 * Same chunking and summation order as imd_groupnorm's own statistics launch on the finished tensor: bit-identical partials, one sweep
 * less.  b holds b_B images (a divisor of B; image i reads b[i % b_B]). */
int imd_concat2_gn_stats(const uint16_t* a, int Ca, const uint16_t* b, int Cb, const uint16_t* b_add, uint16_t* out, int B, int HW, int b_B, int G,
                         float* partial, int dtype, void* stream);
/* number of per-image statistic partials imd_groupnorm's own statistics launch (and imd_concat2_gn_stats) writes for this tensor; 0 if C % 8 */
int imd_groupnorm_parts(int B, int HW, int C);
/* fp32 -> 16-bit element cast (round to nearest even). */
int imd_f32_to_16(const float* a, uint16_t* out, long n, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IMAGDRESSING_HIP_H */
