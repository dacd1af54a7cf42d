// C-ABI entry points of libimagdressing_hip.so (see include/imagdressing_hip.h).
#include <atomic>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "imd_kernels.h"

namespace {
thread_local char g_err[512] = "";
}

int imd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

int imd_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return imd_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

#define IMD_REQUIRE(cond, ...) do { if (!(cond)) return imd_set_error(__VA_ARGS__); } while (0)
// ABI v8: every parameter block starts with the size the CALLER believes it has
#define IMD_REQUIRE_SIZE(p, what) IMD_REQUIRE((p)->struct_bytes == sizeof(*(p)), \
    "%s: parameter block is %u bytes in the caller's view, this library (ABI v%d) expects %zu: the binding mirrors another version of include/imagdressing_hip.h (set struct_bytes = sizeof(struct) after zero-initialising it)", \
    what, (unsigned)(p)->struct_bytes, IMD_ABI_VERSION, sizeof(*(p)))

// ---- per-device launcher state (see imd_kernels.h) ----
namespace {
struct LdsAttrEnt { const void* kern; int dev; };
constexpr int kMaxLdsAttr = 512;
LdsAttrEnt g_lds_attr[kMaxLdsAttr];
std::atomic<int> g_lds_attr_n{0};
std::mutex g_lds_attr_mu;
int g_cu8[64] = {};
}  // namespace

int imd_lds_attr(const void* kern, int bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return imd_set_error("%s: cannot identify the current device", what);
    const int n = g_lds_attr_n.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i)
        if (g_lds_attr[i].kern == kern && g_lds_attr[i].dev == dev) return 0;
    std::lock_guard<std::mutex> lock(g_lds_attr_mu);
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return imd_set_error("%s: hipFuncSetAttribute failed: %s", what, hipGetErrorString(e));
    const int m = g_lds_attr_n.load(std::memory_order_relaxed);
    if (m < kMaxLdsAttr) {                 // (a full table only means the attribute is set again on later launches)
        g_lds_attr[m] = LdsAttrEnt{kern, dev};
        g_lds_attr_n.store(m + 1, std::memory_order_release);
    }
    return 0;
}

int imd_cu_count8() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    if (g_cu8[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
        g_cu8[dev] = cus / 8 * 8 < 8 ? 8 : cus / 8 * 8;
    }
    return g_cu8[dev];
}

extern "C" {

int imd_abi_version(void) { return IMD_ABI_VERSION; }
const char* imd_last_error(void) { return g_err; }

int imd_device_check(int device) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return imd_set_error("device_check: hipGetDeviceProperties(%d) failed: %s", device, hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return imd_set_error("device_check: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    return 0;
}

int imd_conv_gemm(const imd_conv_gemm_params* p, int cfg, void* stream) {
    IMD_REQUIRE(p != nullptr, "conv_gemm: null params");
    IMD_REQUIRE_SIZE(p, "conv_gemm");
    IMD_REQUIRE(p->x && p->w, "conv_gemm: null input/weight pointer");
    IMD_REQUIRE(p->mode == IMD_OUT_HEADS || p->out != nullptr, "conv_gemm: null output pointer");
    IMD_REQUIRE(p->Hout > 0 && p->Wout > 0 && p->Hin > 0 && p->Win > 0, "conv_gemm: bad geometry");
    IMD_REQUIRE(p->stride == 1 || p->stride == 2, "conv_gemm: stride must be 1 or 2");
    IMD_REQUIRE(p->M % (p->Hout * p->Wout) == 0, "conv_gemm: M (%d) is not a multiple of Hout*Wout (%d)", p->M, p->Hout * p->Wout);
    if (p->mode == IMD_OUT_HEADS) {
        IMD_REQUIRE(p->hC > 0 && p->hH > 0 && p->hD > 0 && p->hC == p->hH * p->hD, "conv_gemm: bad head split C=%d H=%d D=%d", p->hC, p->hH, p->hD);
        IMD_REQUIRE(p->N % p->hC == 0 && p->N / p->hC <= 3, "conv_gemm: N (%d) must be 1..3 splits of %d channels", p->N, p->hC);
        IMD_REQUIRE(p->hD % 8 == 0, "conv_gemm: head dim must be a multiple of 8");
        IMD_REQUIRE(p->act != IMD_ACT_GEGLU && !p->out_f32, "conv_gemm: head-split output excludes GEGLU / fp32 output");
    }
    if (p->act == IMD_ACT_GEGLU) IMD_REQUIRE(!p->out_f32 && !p->res, "conv_gemm: GEGLU excludes fp32 output and residual");
    return imd_launch_conv_gemm(*p, cfg, (hipStream_t)stream);
}

int imd_conv_gemm_auto_cfg(int M, int N) { return imd_conv_gemm_choose_cfg(M, N); }
int imd_conv_gemm_auto_split(int M, int N, int K, int cfg) { return imd_conv_gemm_choose_split(M, N, K, cfg < 0 ? imd_conv_gemm_choose_cfg(M, N) : cfg); }

int imd_attention(const imd_attn_params* p, void* stream) {
    IMD_REQUIRE(p != nullptr, "attention: null params");
    IMD_REQUIRE_SIZE(p, "attention");
    IMD_REQUIRE(p->q && p->k1 && p->v1t && p->out, "attention: null q/k/v/out pointer");
    IMD_REQUIRE((p->k2 == nullptr) == (p->v2t == nullptr), "attention: k2 and v2t must be given together");
    IMD_REQUIRE(p->out_ld >= p->H * p->D && p->out_ld % 4 == 0, "attention: bad out_ld %d", p->out_ld);
    return imd_launch_attention(*p, (hipStream_t)stream);
}

int imd_attention_fp8(const imd_attn_params* p, int eq, int ek, int ev, void* stream) {
    IMD_REQUIRE(p != nullptr, "attention_fp8: null params");
    IMD_REQUIRE_SIZE(p, "attention_fp8");
    IMD_REQUIRE(p->q && p->k1 && p->v1t && p->out, "attention_fp8: null q/k/v/out pointer");
    IMD_REQUIRE((p->k2 == nullptr) == (p->v2t == nullptr), "attention_fp8: k2 and v2t must be given together");
    IMD_REQUIRE(p->out_ld >= p->H * p->D && p->out_ld % 4 == 0, "attention_fp8: bad out_ld %d", p->out_ld);
    IMD_REQUIRE(p->dtype == IMD_DTYPE_BF16 || p->dtype == IMD_DTYPE_F16, "attention_fp8: unknown output dtype %d", p->dtype);
    return imd_launch_attention_fp8(*p, eq, ek, ev, (hipStream_t)stream);
}

int imd_attn_quantize_fp8(const uint16_t* src, uint8_t* dst, int kind, long count, int LP, int exp2_scale, float pad_val, int dtype,
                          void* stream) {
    IMD_REQUIRE(src && dst && count > 0, "attn_quantize_fp8: null pointer / empty");
    IMD_REQUIRE(kind == 0 || kind == 1, "attn_quantize_fp8: kind must be 0 (Q / K rows) or 1 (V^T)");
    IMD_REQUIRE(exp2_scale >= -16 && exp2_scale <= 16, "attn_quantize_fp8: exponent out of range");
    return imd_launch_attn_quantize_fp8(src, dst, kind, count, LP, exp2_scale, pad_val, dtype, (hipStream_t)stream);
}

int imd_set_tuning(int knob, int value) {
    switch (knob) {
        case 0:
#ifdef IMD_ABLATIONS
            IMD_REQUIRE(value >= 1 && value <= 54, "set_tuning: attention variant for head dim 40 must be 1..54 (20..49: timing ablations, WRONG results; 50..54: correct but no faster)");
#else
            IMD_REQUIRE(value >= 1 && value <= 13, "set_tuning: attention variant for head dim 40 must be 1..13 (the timing ablations 20..49 and the measured no-gain variants 50..54 exist only in -DIMD_ABLATIONS builds)");
#endif
            g_attn_qw40 = value; return 0;
        case 1: g_attn_xcd = value ? 1 : 0; return 0;
        case 2:                                          // (bit 8: row_linear staged epilogue, bit 9: halo-patch conv staged through registers, bit 10: row kernels store 8 bytes per lane; A/B only)
#ifndef IMD_ABLATIONS
            IMD_REQUIRE((value & 224) == 0, "set_tuning: bits 5..7 of knob 2 are the timing ablations of gemm_dma256.hip (WRONG results); they exist only in -DIMD_ABLATIONS builds");
#endif
            IMD_REQUIRE((value & ~2047) == 0, "set_tuning: knob 2 has bits 0..10 only (got %d)", value);
            g_gemm_flags = value; return 0;
#ifdef IMD_ATTN_SWEEP
        case 3: g_attn_v80 = value; return 0;
        case 4: g_attn_v160 = value; return 0;
#endif
        default: return imd_set_error("set_tuning: unknown knob %d", knob);
    }
}

int imd_get_tuning(int knob) { return knob == 0 ? g_attn_qw40 : knob == 1 ? g_attn_xcd : knob == 2 ? g_gemm_flags : -1; }

int imd_attn_padded_dims(int D, int* dpk, int* dpv) {
    IMD_REQUIRE(D == 40 || D == 64 || D == 80 || D == 160, "attn_padded_dims: unsupported head dim %d", D);
    if (dpk) *dpk = imd_attn_dpk(D);
    if (dpv) *dpv = imd_attn_dpv(D);
    return 0;
}

int imd_groupnorm(const imd_groupnorm_params* p, void* stream) {
    IMD_REQUIRE(p != nullptr, "groupnorm: null params");
    IMD_REQUIRE_SIZE(p, "groupnorm");
    IMD_REQUIRE(p->x && p->y && p->gamma && p->beta && p->partial, "groupnorm: null pointer");
    return imd_launch_groupnorm(*p, (hipStream_t)stream);
}

int imd_groupnorm_coeffs(const imd_groupnorm_params* p, float* coef_a, float* coef_b, void* stream) {
    IMD_REQUIRE(p != nullptr, "groupnorm_coeffs: null params");
    IMD_REQUIRE_SIZE(p, "groupnorm_coeffs");
    IMD_REQUIRE(p->x && p->gamma && p->beta && p->partial && coef_a && coef_b, "groupnorm_coeffs: null pointer");
    return imd_launch_groupnorm_coeffs(*p, coef_a, coef_b, (hipStream_t)stream);
}

static bool sized(const imd_conv_gemm_params* p) { return p && p->struct_bytes == sizeof(*p); }      // (queries answer 0 for a foreign layout)
// the queries see the block as the launcher will: with the operand extents filled in (a caller's block still has them at zero, which would
// let the "< 2 GiB" clauses of the LDS-DMA kernels pass for any size); an operand beyond 4 GiB is "not supported"
static int query(const imd_conv_gemm_params* p, bool (*pred)(const imd_conv_gemm_params&)) {
    if (!sized(p)) return 0;
    imd_conv_gemm_params q = *p;
    if (!imd_conv_gemm_fill_extents(q)) return 0;
    return pred(q) ? 1 : 0;
}
int imd_conv_patch_supported(const imd_conv_gemm_params* p) { return query(p, static_cast<bool (*)(const imd_conv_gemm_params&)>(imd_conv_patch_supported)); }
int imd_conv_patch2_supported(const imd_conv_gemm_params* p) { return query(p, static_cast<bool (*)(const imd_conv_gemm_params&)>(imd_conv_patch2_supported)); }
int imd_conv_patch3_supported(const imd_conv_gemm_params* p) { return query(p, static_cast<bool (*)(const imd_conv_gemm_params&)>(imd_conv_patch3_supported)); }
int imd_conv_patch4_supported(const imd_conv_gemm_params* p) { return query(p, static_cast<bool (*)(const imd_conv_gemm_params&)>(imd_conv_patch4_supported)); }
int imd_conv_img_supported(const imd_conv_gemm_params* p) { return query(p, static_cast<bool (*)(const imd_conv_gemm_params&)>(imd_conv_img_supported)); }
int imd_conv_patch_stats_parts(const imd_conv_gemm_params* p) { return sized(p) ? imd_conv_patch_stats_parts_of(*p) : 0; }
int imd_conv_gemm_stats_parts(const imd_conv_gemm_params* p, int cfg) { return sized(p) ? imd_conv_gemm_stats_parts_of(*p, cfg) : 0; }
int imd_row_linear_gn_in_supported(const imd_conv_gemm_params* p, int cfg) { return (sized(p) && p->gn_in_partial != nullptr && imd_row_linear_gn_in_supported_of(*p, cfg)) ? 1 : 0; }
int imd_conv_gemm_gn_out_supported(const imd_conv_gemm_params* p) { return (sized(p) && imd_conv_gemm_gn_out_supported_of(*p, -1)) ? 1 : 0; }
int imd_gemm_dma_supported(const imd_conv_gemm_params* p) { return query(p, static_cast<bool (*)(const imd_conv_gemm_params&)>(imd_gemm_dma_supported)); }

int imd_row_linear_supported(const imd_conv_gemm_params* p) { return (sized(p) && (imd_row_linear_supported(*p) || imd_row_linear_k640_supported(*p) || imd_row_linear_k1280_supported(*p) || imd_row_qkv_supported(*p))) ? 1 : 0; }

int imd_row_linear(const imd_conv_gemm_params* p, int ln, float ln_eps, void* stream) {
    IMD_REQUIRE(p != nullptr, "row_linear: null params");
    IMD_REQUIRE_SIZE(p, "row_linear");
    IMD_REQUIRE(p->x && p->w, "row_linear: null input/weight pointer");
    IMD_REQUIRE(p->mode == IMD_OUT_HEADS || p->out != nullptr, "row_linear: null output pointer");
    IMD_REQUIRE(p->Hout > 0 && p->Wout > 0 && p->M > 0 && p->M % (p->Hout * p->Wout) == 0, "row_linear: bad geometry");
    if (p->mode == IMD_OUT_HEADS) {
        IMD_REQUIRE(p->hC > 0 && p->hH > 0 && p->hD > 0 && p->hC == p->hH * p->hD && p->N % p->hC == 0 && p->N / p->hC <= 3 && p->hD % 8 == 0,
                    "row_linear: bad head split C=%d H=%d D=%d", p->hC, p->hH, p->hD);
        IMD_REQUIRE(!p->out_f32, "row_linear: head-split output excludes fp32 output");
    }
    IMD_REQUIRE(!ln || ln_eps > 0.f, "row_linear: LayerNorm needs eps > 0");
    if (p->K == 320 && p->N == 960 && p->mode == IMD_OUT_HEADS) {
        IMD_REQUIRE(p->gn_in_partial == nullptr, "row_linear: the fused q / k / v projection has no GroupNorm prologue (gn_in_*)");
        return imd_launch_row_qkv(*p, ln, ln_eps, (hipStream_t)stream);
    }
    if (p->K == 640) return imd_launch_row_linear_k640(*p, ln, ln_eps, (hipStream_t)stream);
    if (p->K == 1280) return imd_launch_row_linear_k1280(*p, ln, ln_eps, (hipStream_t)stream);
    return imd_launch_row_linear(*p, ln, ln_eps, (hipStream_t)stream);
}

int imd_ff_geglu(const imd_ff_params* p, void* stream) {
    IMD_REQUIRE(p != nullptr, "ff_geglu: null params");
    IMD_REQUIRE_SIZE(p, "ff_geglu");
    IMD_REQUIRE(p->x && p->w1 && p->b1 && p->w2 && p->b2 && p->out, "ff_geglu: null pointer");
    IMD_REQUIRE(!p->ln || p->ln_eps > 0.f, "ff_geglu: LayerNorm needs eps > 0");
    return imd_launch_ff_geglu(*p, (hipStream_t)stream);
}

int imd_layernorm(const imd_layernorm_params* p, void* stream) {
    IMD_REQUIRE(p != nullptr, "layernorm: null params");
    IMD_REQUIRE_SIZE(p, "layernorm");
    IMD_REQUIRE(p->x && p->y && p->gamma && p->beta, "layernorm: null pointer");
    return imd_launch_layernorm(*p, (hipStream_t)stream);
}

int imd_embed_tokens(const uint16_t* table, int vocab, const uint16_t* pos, int T, const int64_t* ids, uint16_t* out, long rows, int C,
                     int dtype, void* stream) {
    IMD_REQUIRE(table && pos && ids && out, "embed_tokens: null pointer");
    return imd_launch_embed_tokens(table, vocab, pos, T, ids, out, rows, C, dtype, (hipStream_t)stream);
}

int imd_vit_assemble(const uint16_t* patches, const uint16_t* cls, const uint16_t* pos, uint16_t* out, int B, int P, int C, int dtype,
                     void* stream) {
    IMD_REQUIRE(patches && cls && pos && out, "vit_assemble: null pointer");
    return imd_launch_vit_assemble(patches, cls, pos, out, B, P, C, dtype, (hipStream_t)stream);
}

int imd_lincomb(const float* const* xs, const float* coefs, int n, float* out, long numel, void* stream) {
    IMD_REQUIRE(xs && coefs && out, "lincomb: null pointer");
    return imd_launch_lincomb(xs, coefs, n, out, numel, (hipStream_t)stream);
}

int imd_softmax_rows(const float* s, int s_ld, uint16_t* p, int p_ld, int rows, int cols, int dtype, void* stream) {
    IMD_REQUIRE(s && p, "softmax_rows: null pointer");
    return imd_launch_softmax_rows(s, s_ld, p, p_ld, rows, cols, dtype, (hipStream_t)stream);
}

int imd_ddim_cfg_step(const imd_ddim_params* p, void* stream) {
    IMD_REQUIRE(p != nullptr, "ddim_cfg_step: null params");
    IMD_REQUIRE_SIZE(p, "ddim_cfg_step");
    IMD_REQUIRE(p->z && p->eps, "ddim_cfg_step: null pointer");
    IMD_REQUIRE(p->coefs != nullptr || p->sqrt_a_t > 0.f, "ddim_cfg_step: sqrt(alpha_t) must be positive");
    return imd_launch_ddim_cfg_step(*p, (hipStream_t)stream);
}

int imd_timestep_embedding(const float* t, float* out, int B, int dim, void* stream) {
    IMD_REQUIRE(t && out, "timestep_embedding: null pointer");
    return imd_launch_timestep_embedding(t, out, B, dim, (hipStream_t)stream);
}

int imd_add(const uint16_t* a, int a_ld, const uint16_t* b, int b_ld, uint16_t* out, int out_ld, long rows, int C, float b_scale, int dtype, void* stream) {
    IMD_REQUIRE(a && b && out, "add: null pointer");
    return imd_launch_add(a, a_ld, b, b_ld, out, out_ld, rows, C, b_scale, dtype, (hipStream_t)stream);
}

int imd_copy2d(const uint16_t* a, int a_ld, uint16_t* out, int out_ld, long rows, int C, void* stream) {
    IMD_REQUIRE(a && out, "copy2d: null pointer");
    return imd_launch_copy2d(a, a_ld, out, out_ld, rows, C, (hipStream_t)stream);
}

int imd_concat2(const uint16_t* a, int Ca, const uint16_t* b, int Cb, const uint16_t* b_add, uint16_t* out, long rows, long b_rows, int dtype, void* stream) {
    IMD_REQUIRE(a && b && out, "concat2: null pointer");
    return imd_launch_concat2(a, Ca, b, Cb, b_add, out, rows, b_rows, dtype, (hipStream_t)stream);
}

int imd_groupnorm_parts(int B, int HW, int C) { return imd_groupnorm_parts_of(B, HW, C); }

int imd_concat2_gn_stats(const uint16_t* a, int Ca, const uint16_t* b, int Cb, const uint16_t* b_add, uint16_t* out, int B, int HW, int b_B, int G,
                         float* partial, int dtype, void* stream) {
    IMD_REQUIRE(a && b && out && partial, "concat2 + statistics: null pointer");
    return imd_launch_concat2_gn_stats(a, Ca, b, Cb, b_add, out, B, HW, b_B, G, partial, dtype, (hipStream_t)stream);
}

int imd_f32_to_16(const float* a, uint16_t* out, long n, int dtype, void* stream) {
    IMD_REQUIRE(a && out, "f32_to_16: null pointer");
    return imd_launch_f32_to_16(a, out, n, dtype, (hipStream_t)stream);
}

}  // extern "C"
