// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3) on gfx950: which (row, k) does byte j of lane l of the
// A / B operand stand for, and what do the E8M0 scale operands do.  Prints the decoded maps; used once to write
// attention_d40_fp8.hip (no guide in this image documents the 32-byte operand layout).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mx_layout_probe.hip -o gpurun_out/mx_probe && gpurun_out/mx_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void probe(const uint8_t* a, const uint8_t* b, float* c, int sa, int sb) {      // a, b: [64 lanes][32 bytes]
    const int l = threadIdx.x;
    v8i A, B;
    memcpy(&A, a + l * 32, 32);
    memcpy(&B, b + l * 32, 32);
    v16f acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) c[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];      // C[row][col]
}

static uint8_t f8(int v) {       // small non-negative integers 0..8 as e4m3 (bias 7)
    static const uint8_t t[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
    return t[v];
}

int main() {
    uint8_t *a, *b; float* c;
    hipMallocManaged(&a, 64 * 32); hipMallocManaged(&b, 64 * 32); hipMallocManaged(&c, 32 * 32 * 4);
    // ---- A operand map: one-hot byte (l, j) in A; B holds k-codes under the HYPOTHESIS that B lane l, byte j = (k = 32 (l>>5) + j, col = l & 31)
    // -> first find B's own map with A = all ones in ONE row-set: do it symmetrically in two passes
    const int samples[][2] = {{0, 0}, {0, 1}, {0, 7}, {0, 8}, {0, 15}, {0, 16}, {0, 17}, {0, 31}, {1, 0}, {31, 3}, {32, 0}, {32, 1}, {32, 15}, {32, 16}, {32, 31}, {45, 9}};
    for (int which = 0; which < 2; ++which) {       // 0: probe A's map, 1: probe B's map
        printf("%s operand: (lane, byte) -> (%s, k)\n", which ? "B" : "A", which ? "col" : "row");
        for (auto& s : samples) {
            int code[2];
            int idx = -1;
            for (int pass = 0; pass < 2; ++pass) {  // pass 0: other operand = 1 + (byte index & 7); pass 1: 1 + (byte index >> 3) + 4 * (lane >> 5)
                memset(a, 0, 64 * 32); memset(b, 0, 64 * 32);
                uint8_t* hot = which ? b : a; uint8_t* oth = which ? a : b;
                hot[s[0] * 32 + s[1]] = f8(1);
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 32; ++j) oth[l * 32 + j] = f8(pass == 0 ? 1 + (j & 7) : 1 + (j >> 3) + 4 * (l >> 5));
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, c, 127, 127);
                hipDeviceSynchronize();
                // the one-hot element contributes to one row (A) / one column (B) of C; all entries of it are equal to the other operand's code
                int found = -1; float val = 0;
                for (int i = 0; i < 32 && found < 0; ++i)
                    for (int n = 0; n < 32; ++n) {
                        const float v = which ? c[n * 32 + i] : c[i * 32 + n];
                        if (v != 0) { found = i; val = v; break; }
                    }
                idx = found; code[pass] = (int)val;
            }
            // the other operand's byte (j', half h') that multiplied the hot byte: j' & 7 = code0 - 1, (j' >> 3) + 4 h' = code1 - 1
            const int jj = (code[0] - 1) + 8 * ((code[1] - 1) & 3), hh = (code[1] - 1) >> 2;
            printf("  (%2d, %2d) -> (%2d, pairs with other operand's half-wave %d byte %2d)\n", s[0], s[1], idx, hh, jj);
        }
    }
    // ---- scales: A = B = ones -> C = 64 at scale bytes (127, 127); what do (128, 127), (127, 126), (0x7f7f7f80 & opsel) give
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { a[l * 32 + j] = f8(1); b[l * 32 + j] = f8(1); }
    const int sc[][2] = {{127, 127}, {128, 127}, {127, 126}, {130, 125}, {0x80 << 8 | 127, 127}};
    for (auto& s : sc) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, c, s[0], s[1]);
        hipDeviceSynchronize();
        printf("scale_a=0x%x scale_b=0x%x: C[0][0]=%g C[31][31]=%g\n", s[0], s[1], c[0], c[31 * 32 + 31]);
    }
    return 0;
}
