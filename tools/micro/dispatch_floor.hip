// How long does a launch of many small workgroups take when the workgroups do (almost) nothing?  The floor under env_step_kernel's
// launches: (workgroups, threads, dynamic LDS bytes) -> us per launch (HIP events over 200 back-to-back launches).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/dispatch_floor.hip -o /tmp/dispatch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void tiny(int *out, int spin) {
    extern __shared__ int sm[];
    if (threadIdx.x == 0) sm[0] = blockIdx.x;
    __syncthreads();
    int v = sm[0];
    for (int i = 0; i < spin; ++i) v = v * 1664525 + 1013904223;  // dependent chain: ~spin * 8 cycles
    if (v == 12345 && threadIdx.x == 0) out[0] = v;
}
int main() {
    int *out;
    hipMalloc(&out, 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int cfg[][3] = {{4096, 128, 9424}, {4096, 128, 0}, {2048, 256, 18848}, {1024, 512, 37696}, {4096, 64, 7000}, {8192, 64, 4000}, {2048, 64, 6000},
                          {16384, 128, 9424}, {256, 128, 9424}};
    for (int spin : {0, 1000}) {
        for (auto &c : cfg) {
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(tiny, dim3(c[0]), dim3(c[1]), c[2], 0, out, spin);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny, dim3(c[0]), dim3(c[1]), c[2], 0, out, spin);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("spin %4d: %5d workgroups x %3d threads, %5d B LDS: %.2f us per launch\n", spin, c[0], c[1], c[2], ms * 1000 / 200);
        }
    }
    return 0;
}
