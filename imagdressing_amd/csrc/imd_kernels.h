// Internal (C++) interface between the C-ABI layer (capi.cpp) and the HIP kernels.
// The parameter structs ARE the public C structs of include/imagdressing_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/imagdressing_hip.h"

typedef uint16_t bf16_t;
typedef imd_heads_dest HeadsDest;
typedef imd_conv_gemm_params ConvGemmParams;
typedef imd_attn_params AttnParams;
typedef imd_groupnorm_params GroupNormParams;
typedef imd_layernorm_params LayerNormParams;
typedef imd_ddim_params DdimParams;

enum { ACT_NONE = IMD_ACT_NONE, ACT_SILU = IMD_ACT_SILU, ACT_GEGLU = IMD_ACT_GEGLU, ACT_GELU = IMD_ACT_GELU, ACT_QUICK_GELU = IMD_ACT_QUICK_GELU };
enum { OUT_ROWMAJOR = IMD_OUT_ROWMAJOR, OUT_HEADS = IMD_OUT_HEADS };

// error plumbing (thread-local message, surfaced by imd_last_error())
int imd_set_error(const char* fmt, ...);
int imd_check_launch(const char* what);
// hipFuncAttributeMaxDynamicSharedMemorySize of `kern` on the CURRENT device, set once per (device, kernel): the attribute is per device, a
// process may drive several (round-5 advisor finding: a per-process flag left the second device's first launch without it).  0 = ok.
int imd_lds_attr(const void* kern, int bytes, const char* what);
// CU count of the current device rounded down to a multiple of 8 (one share per XCD), at least 8; -1 on error (persistent grids)
int imd_cu_count8();

int imd_conv_gemm_choose_cfg(int M, int N);
int imd_conv_gemm_choose_split(int M, int N, int K, int cfg);
int imd_launch_conv_gemm(const ConvGemmParams& p, int cfg, hipStream_t s);
bool imd_conv_gemm_fill_extents(ConvGemmParams& p);          // x_bytes / w_bytes from the geometry (false: an operand exceeds 4 GiB)
bool imd_conv_patch_supported(const ConvGemmParams& p);
bool imd_conv_patch2_supported(const ConvGemmParams& p);      // conv_patch2.hip: 16 x 16 pixel tiles (tile config 21)
int imd_launch_conv_patch2(const ConvGemmParams& p, hipStream_t s);
bool imd_conv_patch3_supported(const ConvGemmParams& p);      // conv_patch3.hip: 8 x 16 pixels x 160 channels (tile config 22)
int imd_launch_conv_patch3(const ConvGemmParams& p, hipStream_t s);
bool imd_conv_patch4_supported(const ConvGemmParams& p);      // conv_patch3.hip with eight waves: 16 x 16 pixels x 160 channels (tile config 23)
int imd_launch_conv_patch4(const ConvGemmParams& p, hipStream_t s);
int imd_conv_patch3_stats_parts_of(const ConvGemmParams& p, int nw);
bool imd_conv_patch64_supported(const ConvGemmParams& p);    // conv_patch.hip with 64-channel chunks = 128-byte rows (tile config 29)
int imd_launch_conv_patch64(const ConvGemmParams& p, hipStream_t s);
bool imd_conv_img_supported(const ConvGemmParams& p);         // conv_img.hip: whole 8-wide maps x 64 channels x one K slice per workgroup (tile config 24)
int imd_launch_conv_img(const ConvGemmParams& p, hipStream_t s);
int imd_conv_patch_stats_parts_of(const ConvGemmParams& p);
int imd_conv_gemm_stats_parts_of(const ConvGemmParams& p, int cfg);
bool imd_conv_gemm_gn_out_supported_of(const ConvGemmParams& p, int cfg);
bool imd_row_linear_gn_in_supported_of(const ConvGemmParams& p, int cfg);
int imd_launch_conv_patch(const ConvGemmParams& p, hipStream_t s);
bool imd_row_linear_supported(const ConvGemmParams& p);                                    // row_linear.hip
int imd_launch_row_linear(const ConvGemmParams& p, int ln, float ln_eps, hipStream_t s);
bool imd_row_linear_k640_supported(const ConvGemmParams& p);                               // row_linear_k640.hip
int imd_launch_row_linear_k640(const ConvGemmParams& p, int ln, float ln_eps, hipStream_t s);
bool imd_row_linear_k1280_supported(const ConvGemmParams& p);                              // row_linear_k1280.hip
int imd_launch_row_linear_k1280(const ConvGemmParams& p, int ln, float ln_eps, hipStream_t s);
bool imd_row_qkv_supported(const ConvGemmParams& p);                                        // row_qkv.hip
int imd_launch_row_qkv(const ConvGemmParams& p, int ln, float ln_eps, hipStream_t s);
bool imd_gemm_dma_supported(const ConvGemmParams& p);                                       // gemm_dma.hip
int imd_launch_gemm_dma(const ConvGemmParams& p, hipStream_t s);
int imd_launch_gemm_dma128(const ConvGemmParams& p, int stages, hipStream_t s);      // stages: 3 | 4 ring stages
int imd_launch_gemm_dma256(const ConvGemmParams& p, int form, hipStream_t s);        // gemm_dma256.hip: 0 = 256x128 persistent, 1 = 256x128 one item per workgroup (tile configs 30 / 31)
bool imd_conv_dma_supported(const ConvGemmParams& p);                         // gemm_dma.hip: 128 x 128 x 32, 3-stage ring (tile config 17)
int imd_launch_ff_geglu(const imd_ff_params& p, hipStream_t s);                              // ff_fused.hip
int imd_launch_attention(const AttnParams& p, hipStream_t s);
int imd_launch_attention_d40(const AttnParams& p, int variant, hipStream_t s);
int imd_launch_attention_fp8(const AttnParams& p, int eq, int ek, int ev, hipStream_t s);                       // attention_d40_fp8.hip
int imd_launch_attn_quantize_fp8(const bf16_t* src, unsigned char* dst, int kind, long rows_or_groups, int LP, int exp2_scale,
                                 float pad_val, int dtype, hipStream_t s);   // attention_d40.hip: software-pipelined level-0 kernel
extern int g_attn_qw40;
extern int g_attn_xcd;
#ifdef IMD_ATTN_SWEEP
extern int g_attn_v80, g_attn_v160;
#endif
extern int g_gemm_flags;
int imd_attn_dpk(int D);
int imd_attn_dpv(int D);
int imd_launch_groupnorm(const GroupNormParams& p, hipStream_t s);
int imd_launch_groupnorm_coeffs(const GroupNormParams& p, float* ca, float* cb, hipStream_t s);
int imd_groupnorm_parts_of(int B, int HW, int C);
int imd_launch_concat2_gn_stats(const bf16_t* a, int Ca, const bf16_t* b, int Cb, const bf16_t* b_add, bf16_t* out, int B, int HW, int b_B, int G,
                                float* partial, int dtype, hipStream_t s);
int imd_launch_layernorm(const LayerNormParams& p, hipStream_t s);
int imd_launch_softmax_rows(const float* s_in, int s_ld, bf16_t* p_out, int p_ld, int rows, int cols, int dtype, hipStream_t s);
int imd_launch_ddim_cfg_step(const DdimParams& p, hipStream_t s);
int imd_launch_timestep_embedding(const float* t, float* out, int B, int dim, hipStream_t s);
int imd_launch_add(const bf16_t* a, int a_ld, const bf16_t* b, int b_ld, bf16_t* out, int out_ld, long rows, int C, float b_scale, int dtype, hipStream_t s);
int imd_launch_embed_tokens(const bf16_t* table, int vocab, const bf16_t* pos, int T, const int64_t* ids, bf16_t* out, long rows, int C, int dtype, hipStream_t s);
int imd_launch_vit_assemble(const bf16_t* patches, const bf16_t* cls, const bf16_t* pos, bf16_t* out, int B, int P, int C, int dtype, hipStream_t s);
int imd_launch_lincomb(const float* const* xs, const float* coefs, int n, float* out, long numel, hipStream_t s);
int imd_launch_copy2d(const bf16_t* a, int a_ld, bf16_t* out, int out_ld, long rows, int C, hipStream_t s);
int imd_launch_concat2(const bf16_t* a, int Ca, const bf16_t* b, int Cb, const bf16_t* b_add, bf16_t* out, long rows, long b_rows, int dtype, hipStream_t s);
int imd_launch_f32_to_16(const float* a, bf16_t* out, long n, int dtype, hipStream_t s);
