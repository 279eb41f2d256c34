// Latency of the FIRST global loads of a launch when every wave of the chip issues them at the same moment (what env_step_kernel's
// start phase waits for): 4096 workgroups x 128 threads, each wave loads 4 bytes per lane from a small (L2-resident) array, then a
// dependent second load; s_memtime around them.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/first_load.hip -o first_load.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void probe(const int *a, const int *b, unsigned long long *out, int nloads) {
    unsigned long long t0, t1, t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const int idx = (blockIdx.x * 40 + (threadIdx.x % 40)) & 0x3FFFF;
    int v = 0;
    for (int k = 0; k < nloads; ++k) v += a[(idx + k * 4096) & 0x3FFFF];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(v) : "memory");
    int w = b[(v & 0xFFFF) + threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) : "v"(w) : "memory");
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = t1 - t0;
        out[blockIdx.x * 2 + 1] = t2 - t1;
    }
    if (w == 123456789) out[0] = 0;
}
int main() {
    int *a, *b;
    unsigned long long *out;
    hipMalloc(&a, 1 << 20);
    hipMalloc(&b, 1 << 20);
    hipMemset(a, 0, 1 << 20);
    hipMemset(b, 0, 1 << 20);
    for (int blocks : {256, 1024, 4096, 16384}) {
        for (int nl : {1, 12}) {
            hipMalloc(&out, blocks * 16);
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe, dim3(blocks), dim3(128), 0, 0, a, b, out, nl);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(blocks * 2);
            hipMemcpy(h.data(), out, blocks * 16, hipMemcpyDeviceToHost);
            std::vector<unsigned long long> f, s;
            for (int i = 0; i < blocks; ++i) {
                f.push_back(h[2 * i]);
                s.push_back(h[2 * i + 1]);
            }
            std::sort(f.begin(), f.end());
            std::sort(s.begin(), s.end());
            printf("%5d workgroups x 128, %2d first loads per lane: first round p10/p50/p90 = %llu / %llu / %llu cycles; dependent second load p50 = %llu\n",
                   blocks, nl, f[blocks / 10], f[blocks / 2], f[blocks * 9 / 10], s[blocks / 2]);
            hipFree(out);
        }
    }
    return 0;
}
