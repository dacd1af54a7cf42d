// What does a launch cost before it does anything?  Back-to-back launches of a kernel that stores one word per workgroup, as a function of
// the dynamic LDS allocation, the workgroup size and the grid -- the geometries of the tile kernels (48 KB x 256 threads, three per CU), the
// row-resident kernels (124..160 KB x 512, one per CU) and conv_img.hip (136 KB x 256).  Reports us per launch in a stream (HIP events
// over 400 launches): the floor under every short kernel of the denoising step (~23,600 launches per image batch).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/launch_ramp_probe.hip -o gpurun_out/launch_ramp_probe && gpurun_out/launch_ramp_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void touch(uint32_t* out) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) { smem[0] = (char)blockIdx.x; out[blockIdx.x] = (uint32_t)smem[0]; }
}

int main() {
    uint32_t* out;
    hipMalloc(&out, 1 << 20);
    hipFuncSetAttribute(reinterpret_cast<const void*>(touch), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int lds[] = {0, 48 * 1024, 100 * 1024, 136 * 1024, 160 * 1024};
    const int threads[] = {256, 512};
    const int grids[] = {64, 240, 512, 2048};
    for (int t : threads)
        for (int g : grids)
            for (int l : lds) {
                for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(touch, dim3(g), dim3(t), l, 0, out);
                hipDeviceSynchronize();
                hipEventRecord(e0, 0);
                for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(touch, dim3(g), dim3(t), l, 0, out);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                printf("{\"threads\": %d, \"grid\": %d, \"lds_kb\": %d, \"us_per_launch\": %.2f}\n", t, g, l / 1024, ms * 1000.f / 400.f);
            }
    return 0;
}
