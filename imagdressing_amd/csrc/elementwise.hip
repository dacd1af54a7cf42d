// HBM-bound elementwise kernels of the sampling loop (gfx950).
//
// ddim_cfg_step fuses, over the [B, HW, 4] latent:
//   classifier-free guidance      eps = eps_u + g (eps_c - eps_u)
//       (/root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline.py:521-527)
//   the DDIM update (eta = 0)     x0 = (z - sqrt(1-a_t) eps) / sqrt(a_t);  z' = sqrt(a_prev) x0 + sqrt(1-a_prev) eps
//       (diffusers==0.24.0 DDIMScheduler.step, call site :530-532)
//   the inpainting blend          z' = (1-m) add_noise(z_img, noise, t_next) + m z'
//       (..._pipeline_controlnet_inpainting.py:487-500)
//   and the next step's UNet input: bf16, channels padded 4 -> 8, duplicated for the cond and
//   uncond halves (torch.cat([latents]*2), :483-488; scale_model_input is the identity for DDIM).
// Algorithmic traffic per latent element: 3 fp32 reads + 1 fp32 write (+3 reads with inpaint)
// + 2 x 4 B of bf16 next-input writes.
#include "common.h"
#include "imd_kernels.h"

namespace {

template <bool F16>
__global__ __launch_bounds__(256) void ddim_cfg_step_kernel(const DdimParams p) {
    const long total = (long)p.B * p.HW;                   // one thread per pixel (4 channels = 16 B)
    float sa_t = p.sqrt_a_t, s1_t = p.sqrt_1m_a_t, sa_p = p.sqrt_a_prev, s1_p = p.sqrt_1m_a_prev, sa_n = p.sqrt_a_next, s1_n = p.sqrt_1m_a_next;
    if (p.coefs) {                                         // schedule coefficients from device memory (HIP-graph replay of a step)
        sa_t = p.coefs[0]; s1_t = p.coefs[1]; sa_p = p.coefs[2]; s1_p = p.coefs[3]; sa_n = p.coefs[4]; s1_n = p.coefs[5];
    }
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const float4 z = reinterpret_cast<const float4*>(p.z)[i];
        const float4 ec = reinterpret_cast<const float4*>(p.eps)[i];
        const float4 eu = reinterpret_cast<const float4*>(p.eps)[i + total];
        float zz[4] = {z.x, z.y, z.z, z.w};
        const float c[4] = {ec.x, ec.y, ec.z, ec.w};
        const float u[4] = {eu.x, eu.y, eu.z, eu.w};
        float mk = 1.f;
        float zi[4] = {0, 0, 0, 0}, nz[4] = {0, 0, 0, 0}, vn[4] = {0, 0, 0, 0};
        if (p.var_noise) {                                 // stochastic DDIM (eta > 0): sigma * noise, added before the blend
            const float4 n = reinterpret_cast<const float4*>(p.var_noise)[i];
            vn[0] = p.sigma * n.x; vn[1] = p.sigma * n.y; vn[2] = p.sigma * n.z; vn[3] = p.sigma * n.w;
        }
        if (p.mask) {
            mk = p.mask[i];
            const float4 a = reinterpret_cast<const float4*>(p.z_img)[i];
            const float4 n = reinterpret_cast<const float4*>(p.noise)[i];
            zi[0] = a.x; zi[1] = a.y; zi[2] = a.z; zi[3] = a.w;
            nz[0] = n.x; nz[1] = n.y; nz[2] = n.z; nz[3] = n.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float eps = u[e] + p.guidance * (c[e] - u[e]);
            const float x0 = (zz[e] - s1_t * eps) / sa_t;
            float zn = sa_p * x0 + s1_p * eps + vn[e];
            if (p.mask) {
                const float proper = sa_n * zi[e] + s1_n * nz[e];
                zn = (1.f - mk) * proper + mk * zn;
            }
            zz[e] = zn;
        }
        reinterpret_cast<float4*>(p.z)[i] = make_float4(zz[0], zz[1], zz[2], zz[3]);
        if (p.x_next) {
            const uint4 o = make_uint4(El<F16>::pack2(zz[0], zz[1]), El<F16>::pack2(zz[2], zz[3]), 0u, 0u);
            reinterpret_cast<uint4*>(p.x_next)[i] = o;
            reinterpret_cast<uint4*>(p.x_next)[i + total] = o;
        }
    }
}

// diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]
__global__ void timestep_embedding_kernel(const float* t, float* out, int B, int dim) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i - b * half;
    const float freq = expf(-9.210340371976184f * (float)j / (float)half);   // ln(10000)
    const float arg = t[b] * freq;
    out[(size_t)b * dim + j] = cosf(arg);
    out[(size_t)b * dim + half + j] = sinf(arg);
}

// out[r, c] = a[r, c] + b_scale * b[r, c]   (strided rows; 8 channels per thread)
template <bool F16>
__global__ __launch_bounds__(256) void add_kernel(const bf16_t* a, int a_ld, const bf16_t* b, int b_ld, bf16_t* out, int out_ld,
                                                   long rows, int C, float b_scale) {
    const int vpr = C / 8;
    const long total = rows * vpr;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const long r = i / vpr;
        const int c = (int)(i - r * vpr) * 8;
        float fa[8], fb[8];
        unpack8<F16>(*reinterpret_cast<const uint4*>(a + r * a_ld + c), fa);
        unpack8<F16>(*reinterpret_cast<const uint4*>(b + r * b_ld + c), fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[e] += b_scale * fb[e];
        *reinterpret_cast<uint4*>(out + r * out_ld + c) = pack8<F16>(fa);
    }
}

__global__ __launch_bounds__(256) void copy2d_kernel(const bf16_t* a, int a_ld, bf16_t* out, int out_ld, long rows, int C) {
    const int vpr = C / 8;
    const long total = rows * vpr;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const long r = i / vpr;
        const int c = (int)(i - r * vpr) * 8;
        *reinterpret_cast<uint4*>(out + r * out_ld + c) = *reinterpret_cast<const uint4*>(a + r * a_ld + c);
    }
}

// out[r, 0:Ca] = a[r], out[r, Ca:Ca+Cb] = b[r] (+ b_add[r]): the skip concat of an up block (+ the ControlNet residual) in ONE launch
template <bool F16>
__global__ __launch_bounds__(256) void concat2_kernel(const bf16_t* a, int Ca, const bf16_t* b, int Cb, const bf16_t* b_add, bf16_t* out, long rows, long b_rows) {
    const int va = Ca / 8, vpr = (Ca + Cb) / 8;
    const long total = rows * vpr;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const long r = i / vpr;
        const int v = (int)(i - r * vpr);
        uint4 val;
        if (v < va) val = *reinterpret_cast<const uint4*>(a + r * Ca + v * 8);
        else {
            const long off = r * Cb + (v - va) * 8;
            val = *reinterpret_cast<const uint4*>(b + (r % b_rows) * Cb + (v - va) * 8);     // (b_rows < rows: b repeats, e.g. one copy for both CFG halves)
            if (b_add) {
                float x[8], y[8];
                unpack8<F16>(val, x);
                unpack8<F16>(*reinterpret_cast<const uint4*>(b_add + off), y);
                val = make_uint4(El<F16>::pack2(x[0] + y[0], x[1] + y[1]), El<F16>::pack2(x[2] + y[2], x[3] + y[3]),
                                 El<F16>::pack2(x[4] + y[4], x[5] + y[5]), El<F16>::pack2(x[6] + y[6], x[7] + y[7]));
            }
        }
        *reinterpret_cast<uint4*>(out + r * (long)(Ca + Cb) + v * 8) = val;
    }
}

// out[r] = table[ids[r]] + pos[r % T]      (8 channels per thread; ids outside [0, vocab) read row 0)
template <bool F16>
__global__ __launch_bounds__(256) void embed_tokens_kernel(const bf16_t* table, int vocab, const bf16_t* pos, int T, const int64_t* ids,
                                                           bf16_t* out, long rows, int C) {
    const int vpr = C / 8;
    const long total = rows * vpr;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const long r = i / vpr;
        const int c = (int)(i - r * vpr) * 8;
        long id = ids[r];
        if (id < 0 || id >= vocab) id = 0;
        float a[8], b[8];
        unpack8<F16>(*reinterpret_cast<const uint4*>(table + id * C + c), a);
        unpack8<F16>(*reinterpret_cast<const uint4*>(pos + (r % T) * C + c), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        *reinterpret_cast<uint4*>(out + r * C + c) = pack8<F16>(a);
    }
}

// out[b][0] = cls + pos[0]; out[b][1 + p] = patches[b][p] + pos[1 + p]
template <bool F16>
__global__ __launch_bounds__(256) void vit_assemble_kernel(const bf16_t* patches, const bf16_t* cls, const bf16_t* pos, bf16_t* out, int B, int P, int C) {
    const int vpr = C / 8;
    const long total = (long)B * (P + 1) * vpr;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const long r = i / vpr;
        const int c = (int)(i - r * vpr) * 8;
        const int b = (int)(r / (P + 1)), t = (int)(r - (long)b * (P + 1));
        float a[8], q[8];
        unpack8<F16>(t == 0 ? *reinterpret_cast<const uint4*>(cls + c) : *reinterpret_cast<const uint4*>(patches + ((long)b * P + t - 1) * C + c), a);
        unpack8<F16>(*reinterpret_cast<const uint4*>(pos + (long)t * C + c), q);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += q[e];
        *reinterpret_cast<uint4*>(out + r * C + c) = pack8<F16>(a);
    }
}

struct LincombArgs { const float* x[8]; float c[8]; int n; };
__global__ __launch_bounds__(256) void lincomb_kernel(const LincombArgs a, float* out, long n4, long numel) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        const long e = i * 4;
        if (e + 4 <= numel) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < a.n) {
                    const float4 v = *reinterpret_cast<const float4*>(a.x[j] + e);
                    acc.x = fmaf(a.c[j], v.x, acc.x); acc.y = fmaf(a.c[j], v.y, acc.y); acc.z = fmaf(a.c[j], v.z, acc.z); acc.w = fmaf(a.c[j], v.w, acc.w);
                }
            *reinterpret_cast<float4*>(out + e) = acc;
        } else {
            for (long k = e; k < numel; ++k) {
                float acc = 0.f;
                for (int j = 0; j < a.n; ++j) acc = fmaf(a.c[j], a.x[j][k], acc);
                out[k] = acc;
            }
        }
    }
}

template <bool F16>
__global__ __launch_bounds__(256) void f32_to_16_kernel(const float* a, bf16_t* out, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) out[i] = El<F16>::fromf(a[i]);
}

inline unsigned grid_for(long work_items) {
    long blocks = (work_items + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

int imd_launch_ddim_cfg_step(const DdimParams& p, hipStream_t s) {
    if (p.B <= 0 || p.HW <= 0) return imd_set_error("ddim_cfg_step: empty latent");
    if (p.mask && (!p.z_img || !p.noise)) return imd_set_error("ddim_cfg_step: inpaint mask given without image latents / noise");
    if (p.var_noise && p.coefs) return imd_set_error("ddim_cfg_step: the stochastic step (var_noise) takes host coefficients, not the device table");
    if (p.dtype == IMD_DTYPE_F16) hipLaunchKernelGGL(ddim_cfg_step_kernel<true>, dim3(grid_for((long)p.B * p.HW)), dim3(256), 0, s, p);
    else if (p.dtype == IMD_DTYPE_BF16) hipLaunchKernelGGL(ddim_cfg_step_kernel<false>, dim3(grid_for((long)p.B * p.HW)), dim3(256), 0, s, p);
    else return imd_set_error("ddim_cfg_step: unknown dtype %d", p.dtype);
    return imd_check_launch("ddim_cfg_step");
}

int imd_launch_timestep_embedding(const float* t, float* out, int B, int dim, hipStream_t s) {
    if (B <= 0 || dim <= 0 || (dim & 1)) return imd_set_error("timestep_embedding: bad shape B=%d dim=%d", B, dim);
    const int n = B * dim / 2;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t, out, B, dim);
    return imd_check_launch("timestep_embedding");
}

int imd_launch_add(const bf16_t* a, int a_ld, const bf16_t* b, int b_ld, bf16_t* out, int out_ld, long rows, int C, float b_scale, int dtype, hipStream_t s) {
    if (rows <= 0 || C <= 0) return imd_set_error("add: empty tensor");
    if (C % 8 || a_ld % 8 || b_ld % 8 || out_ld % 8) return imd_set_error("add: C and row strides must be multiples of 8");
    if (dtype == IMD_DTYPE_F16) hipLaunchKernelGGL(add_kernel<true>, dim3(grid_for(rows * (C / 8))), dim3(256), 0, s, a, a_ld, b, b_ld, out, out_ld, rows, C, b_scale);
    else if (dtype == IMD_DTYPE_BF16) hipLaunchKernelGGL(add_kernel<false>, dim3(grid_for(rows * (C / 8))), dim3(256), 0, s, a, a_ld, b, b_ld, out, out_ld, rows, C, b_scale);
    else return imd_set_error("add: unknown dtype %d", dtype);
    return imd_check_launch("add");
}

int imd_launch_embed_tokens(const bf16_t* table, int vocab, const bf16_t* pos, int T, const int64_t* ids, bf16_t* out, long rows, int C, int dtype, hipStream_t s) {
    if (rows <= 0 || C <= 0 || vocab <= 0 || T <= 0) return imd_set_error("embed_tokens: empty problem");
    if (C % 8) return imd_set_error("embed_tokens: C (%d) must be a multiple of 8", C);
    if (dtype == IMD_DTYPE_F16) hipLaunchKernelGGL(embed_tokens_kernel<true>, dim3(grid_for(rows * (C / 8))), dim3(256), 0, s, table, vocab, pos, T, ids, out, rows, C);
    else if (dtype == IMD_DTYPE_BF16) hipLaunchKernelGGL(embed_tokens_kernel<false>, dim3(grid_for(rows * (C / 8))), dim3(256), 0, s, table, vocab, pos, T, ids, out, rows, C);
    else return imd_set_error("embed_tokens: unknown dtype %d", dtype);
    return imd_check_launch("embed_tokens");
}

int imd_launch_vit_assemble(const bf16_t* patches, const bf16_t* cls, const bf16_t* pos, bf16_t* out, int B, int P, int C, int dtype, hipStream_t s) {
    if (B <= 0 || P <= 0 || C <= 0) return imd_set_error("vit_assemble: empty problem");
    if (C % 8) return imd_set_error("vit_assemble: C (%d) must be a multiple of 8", C);
    const long work = (long)B * (P + 1) * (C / 8);
    if (dtype == IMD_DTYPE_F16) hipLaunchKernelGGL(vit_assemble_kernel<true>, dim3(grid_for(work)), dim3(256), 0, s, patches, cls, pos, out, B, P, C);
    else if (dtype == IMD_DTYPE_BF16) hipLaunchKernelGGL(vit_assemble_kernel<false>, dim3(grid_for(work)), dim3(256), 0, s, patches, cls, pos, out, B, P, C);
    else return imd_set_error("vit_assemble: unknown dtype %d", dtype);
    return imd_check_launch("vit_assemble");
}

int imd_launch_lincomb(const float* const* xs, const float* coefs, int n, float* out, long numel, hipStream_t s) {
    if (n < 1 || n > 8) return imd_set_error("lincomb: 1..8 inputs (got %d)", n);
    if (numel <= 0) return imd_set_error("lincomb: empty tensor");
    LincombArgs a;
    for (int j = 0; j < 8; ++j) { a.x[j] = j < n ? xs[j] : nullptr; a.c[j] = j < n ? coefs[j] : 0.f; }
    a.n = n;
    for (int j = 0; j < n; ++j)
        if (a.x[j] == nullptr || (reinterpret_cast<uintptr_t>(a.x[j]) & 15)) return imd_set_error("lincomb: input %d is null or not 16-byte aligned", j);
    if (reinterpret_cast<uintptr_t>(out) & 15) return imd_set_error("lincomb: output is not 16-byte aligned");
    const long n4 = (numel + 3) / 4;
    hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for(n4)), dim3(256), 0, s, a, out, n4, numel);
    return imd_check_launch("lincomb");
}

int imd_launch_copy2d(const bf16_t* a, int a_ld, bf16_t* out, int out_ld, long rows, int C, hipStream_t s) {
    if (rows <= 0 || C <= 0) return imd_set_error("copy2d: empty tensor");
    if (C % 8 || a_ld % 8 || out_ld % 8) return imd_set_error("copy2d: C and row strides must be multiples of 8");
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, s, a, a_ld, out, out_ld, rows, C);
    return imd_check_launch("copy2d");
}

int imd_launch_concat2(const bf16_t* a, int Ca, const bf16_t* b, int Cb, const bf16_t* b_add, bf16_t* out, long rows, long b_rows, int dtype, hipStream_t s) {
    if (rows <= 0 || Ca <= 0 || Cb <= 0) return imd_set_error("concat2: empty tensor");
    if (b_rows <= 0) b_rows = rows;
    if (rows % b_rows) return imd_set_error("concat2: b_rows (%ld) must divide rows (%ld)", b_rows, rows);
    if (Ca % 8 || Cb % 8) return imd_set_error("concat2: channel counts must be multiples of 8 (got %d + %d)", Ca, Cb);
    const long work = rows * ((Ca + Cb) / 8);
    if (dtype == IMD_DTYPE_F16) hipLaunchKernelGGL(concat2_kernel<true>, dim3(grid_for(work)), dim3(256), 0, s, a, Ca, b, Cb, b_add, out, rows, b_rows);
    else if (dtype == IMD_DTYPE_BF16) hipLaunchKernelGGL(concat2_kernel<false>, dim3(grid_for(work)), dim3(256), 0, s, a, Ca, b, Cb, b_add, out, rows, b_rows);
    else return imd_set_error("concat2: unknown dtype %d", dtype);
    return imd_check_launch("concat2");
}

int imd_launch_f32_to_16(const float* a, bf16_t* out, long n, int dtype, hipStream_t s) {
    if (n <= 0) return imd_set_error("f32_to_16: empty tensor");
    if (dtype == IMD_DTYPE_F16) hipLaunchKernelGGL(f32_to_16_kernel<true>, dim3(grid_for(n)), dim3(256), 0, s, a, out, n);
    else if (dtype == IMD_DTYPE_BF16) hipLaunchKernelGGL(f32_to_16_kernel<false>, dim3(grid_for(n)), dim3(256), 0, s, a, out, n);
    else return imd_set_error("f32_to_16: unknown dtype %d", dtype);
    return imd_check_launch("f32_to_16");
}
