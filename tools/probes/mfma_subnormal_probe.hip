// Does the matrix pipe keep fp16 SUBNORMAL inputs, and do the packing converts produce them?  Decides how far the fp16 attention kernel may
// bias its softmax reference maximum downwards (attention_d40.hip FIRST_BIAS: P = 2^(s - m_first - c) moves the small weights of a row
// into fp16's subnormal range, 2^-15 .. 2^-24).
//   A[i][k] = a (fp16 bit pattern built on the device: 2^e for e = -14 .. -25 via v_cvt_pk_f16_f32 from fp32), B = 1.0
//   -> D[i][j] = 16 * a if the MFMA reads subnormals, 0 if it flushes them.
// Prints, per exponent: the fp16 word the convert produced and D[0][0] / 16 from v_mfma_f32_32x32x16_f16 and (x 32) from v_mfma_f32_16x16x32_f16.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_subnormal_probe.hip -o gpurun_out/mfma_subnormal_probe && gpurun_out/mfma_subnormal_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* in, float* out, uint32_t* words, int n) {
    for (int e = 0; e < n; ++e) {
        f32x2 v = {in[e], in[e]};
        const f16x2 h = __builtin_convertvector(v, f16x2);               // v_cvt_pk_f16_f32 (what El<true>::pack2 compiles to)
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = h[0]; b[i] = (_Float16)1.0f; }
        f32x16 c = {};
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        f32x4 c4 = {};
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
        if (threadIdx.x == 0) {
            out[2 * e] = c[0] / 16.0f;
            out[2 * e + 1] = c4[0] / 32.0f;
            words[e] = (uint32_t)__builtin_bit_cast(uint16_t, h[0]);
        }
    }
}

int main() {
    const int n = 14;
    float hin[n], hout[2 * n];
    uint32_t hw[n];
    for (int e = 0; e < n; ++e) hin[e] = __builtin_ldexpf(1.5f, -13 - e);        // 1.5 x 2^-13 (normal) ... 1.5 x 2^-26 (below half the smallest subnormal)
    float *din, *dout;
    uint32_t* dw;
    hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(hout)); hipMalloc(&dw, sizeof(hw));
    hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, dout, dw, n);
    hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
    hipMemcpy(hw, dw, sizeof(hw), hipMemcpyDeviceToHost);
    int kept = 0, subn = 0;
    for (int e = 0; e < n; ++e) {
        const bool is_sub = (hw[e] & 0x7c00u) == 0 && (hw[e] & 0x3ffu) != 0;
        printf("{\"probe\": \"mfma_subnormal\", \"input\": %.6e, \"fp16_word\": \"0x%04x\", \"fp16_subnormal\": %s, \"mfma_32x32x16_per_term\": %.6e, "
               "\"mfma_16x16x32_per_term\": %.6e}\n", hin[e], hw[e], is_sub ? "true" : "false", hout[2 * e], hout[2 * e + 1]);
        if (is_sub) { ++subn; if (hout[2 * e] != 0.f && hout[2 * e + 1] != 0.f) ++kept; }
    }
    printf("{\"probe\": \"mfma_subnormal\", \"summary\": \"%d of %d subnormal fp16 inputs reach the accumulator non-zero on both MFMA shapes\", \"subnormals_preserved\": %s}\n",
           kept, subn, kept == subn && subn > 0 ? "true" : "false");
    return 0;
}
