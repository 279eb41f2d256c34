// mfma_valu_overlap.hip -- can ONE wave per SIMD run the GRU cell's pointwise math (exp / rcp / fma on registers) under its own MFMA
// chain?  (round 5: the premise of a four-wave, 512-register recurrence cell.)  Three kernels over the same work per wave --
// NM v_mfma_f32_16x16x32_bf16 on 12 independent accumulators and NV "gate" evaluations (2 sigmoids + 1 tanh each) on independent data --
// (a) MFMAs only, (b) pointwise only, (c) both in one basic block, interleaved by sched_group_barrier; 1 or 2 waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int NM = 96, NV = 12, REP = 64;

__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return __builtin_fmaf(2.f, __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)), -1.f); }

template <int MODE>
__global__ void __launch_bounds__(512) k(const float *in, float *out, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(in[lane + i] * 0.01f), b[i] = (__bf16)(in[64 + lane + i] * 0.01f);
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float g[NV][4];
    for (int i = 0; i < NV; ++i)
        for (int j = 0; j < 4; ++j) g[i][j] = in[128 + lane + i * 4 + j];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
        if (MODE & 1) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m % 12] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % 12], 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float rg = sigm(g[i][0]), zg = sigm(g[i][1]), ng = tanh_(g[i][2] + rg * g[i][3]);
                const float o = __builtin_fmaf(1.f - zg, ng, zg * g[i][0]);
                g[i][0] = o;
                g[i][1] = rg + o;
                g[i][2] = zg - o;
            }
        }
        if (MODE == 3) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // three VALU
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < NV; ++i) s += g[i][0] + g[i][1] + g[i][2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float *in, *out;
    unsigned long long *cyc;
    hipMalloc(&in, 4096 * 4);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 37) % 101) / 50.f - 1.f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
        for (int mode = 1; mode <= 3; ++mode) {
            for (int it = 0; it < 3; ++it) {
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, in, out, cyc);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, in, out, cyc);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, in, out, cyc);
            }
            hipDeviceSynchronize();
            unsigned long long c[256];
            hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
            double m = 0;
            for (int i = 0; i < 256; ++i) m += (double)c[i];
            std::printf("%d waves per SIMD, %s: %.0f cycles per repetition (%d MFMAs + %d gate evaluations x 4 per wave)\n", threads / 256,
                        mode == 1 ? "MFMA only      " : mode == 2 ? "pointwise only " : "both interleaved", m / 256 / REP, NM, NV);
        }
    }
    return 0;
}
