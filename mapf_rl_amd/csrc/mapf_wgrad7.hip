// mapf_wgrad7.hip -- weight gradient of the encoder's 1x1 head convolution (reference model.py:160 `nn.Conv2d(128, 16, 1)`,
// backward via worker.py:316):
//     dW7[co][ci] = sum over rows r = (observation, position) of gz7[r][co] * a6[r][ci],   co < 16, ci < 128,
// with R = M * 49 rows (6.0 M at the config-2 learner batch, 19.3 M at 128 agents).  As a library GEMM this [16 x R] x [R x 128]
// product got a 256 x 192 macro tile for its 16 x 128 output -- 94 % wasted MFMA work on a split over 8192-row slabs: 5.5 ms of a
// 38.8 ms update at config 2, 24 ms of 120 ms at 128 agents (profiles/r01_l_learner_kernel_stats.md, r02_c5_learner_*).  The
// product is 4 KFLOP per 288 bytes of input: HBM-bound (a6 is read exactly once: 1.5 GB at config 2), so this is a streaming
// kernel on the vector ALU, not an MFMA tiling: a wavefront owns a contiguous range of rows, lane l holds input channels 2 l and
// 2 l + 1 of the current row (one coalesced 256-byte row per wave load) and 16 x 2 fp32 sums; the row of gz7 (16 values, the same
// for every lane) comes through the scalar cache.  UNROLL rows are in flight per wave.  Partial sums per workgroup (the 4 waves
// are added through LDS), fp32 [MAPF_ENC_WGRAD7_PARTS][16][128], summed by the caller (deterministic, no atomics).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "mapf_dqn.h"
#include "mapf_env.h"

namespace {

constexpr int PARTS = MAPF_ENC_WGRAD7_PARTS, NTHR = 256, NW = NTHR / 64, UNROLL = 4;  // (8 rows in flight: 64 scalar registers of gz7, spills)

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }

__global__ void __launch_bounds__(NTHR) conv7_wgrad_kernel(const uint32_t *__restrict__ gz7, const uint32_t *__restrict__ a6, long long R,
                                                          float *__restrict__ partial) {
    __shared__ float red[NW][16][128];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // rows of this wave: R split evenly over PARTS * NW waves
    const long long nwaves = (long long)PARTS * NW, wave = (long long)blockIdx.x * NW + w;
    const long long r0 = R * wave / nwaves, r1 = R * (wave + 1) / nwaves;
    float acc0[16], acc1[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc0[c] = acc1[c] = 0.f;
    long long r = r0;
    for (; r + UNROLL <= r1; r += UNROLL) {
        uint32_t av[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) av[u] = a6[(r + u) * 64 + lane];  // 2 channels of row r + u
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t *g = gz7 + (r + u) * 8;  // wave-uniform address: 8 dwords = 16 bf16 through the scalar cache
            const float a_lo = bf16_lo(av[u]), a_hi = bf16_hi(av[u]);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t gw = g[q];
                const float g0 = bf16_lo(gw), g1 = bf16_hi(gw);
                acc0[2 * q] = fmaf(g0, a_lo, acc0[2 * q]);
                acc1[2 * q] = fmaf(g0, a_hi, acc1[2 * q]);
                acc0[2 * q + 1] = fmaf(g1, a_lo, acc0[2 * q + 1]);
                acc1[2 * q + 1] = fmaf(g1, a_hi, acc1[2 * q + 1]);
            }
        }
    }
    for (; r < r1; ++r) {
        const uint32_t av = a6[r * 64 + lane];
        const uint32_t *g = gz7 + r * 8;
        const float a_lo = bf16_lo(av), a_hi = bf16_hi(av);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t gw = g[q];
            const float g0 = bf16_lo(gw), g1 = bf16_hi(gw);
            acc0[2 * q] = fmaf(g0, a_lo, acc0[2 * q]);
            acc1[2 * q] = fmaf(g0, a_hi, acc1[2 * q]);
            acc0[2 * q + 1] = fmaf(g1, a_lo, acc0[2 * q + 1]);
            acc1[2 * q + 1] = fmaf(g1, a_hi, acc1[2 * q + 1]);
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) *reinterpret_cast<float2 *>(&red[w][c][2 * lane]) = make_float2(acc0[c], acc1[c]);
    __syncthreads();
    float *out = partial + (size_t)blockIdx.x * 16 * 128;
    for (int i = threadIdx.x; i < 16 * 128; i += NTHR) {
        const int c = i >> 7, ci = i & 127;
        float s = red[0][c][ci];
#pragma unroll
        for (int k = 1; k < NW; ++k) s += red[k][c][ci];
        out[i] = s;
    }
}

}  // namespace

extern "C" {

int mapf_encoder_wgrad7(const uint16_t *gz7_dev, const uint16_t *in_dev, int64_t M, float *partial_dev, void *stream) {
    if (M < 0 || !partial_dev || (M > 0 && (!gz7_dev || !in_dev))) return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(gz7_dev) & 15) || (reinterpret_cast<uintptr_t>(in_dev) & 15) || (reinterpret_cast<uintptr_t>(partial_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    // every partition writes its slab (zeros when it has no rows), so the caller's sum is always defined
    hipLaunchKernelGGL(conv7_wgrad_kernel, dim3(PARTS), dim3(NTHR), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const uint32_t *>(gz7_dev),
                       reinterpret_cast<const uint32_t *>(in_dev), (long long)M * 49, partial_dev);
    if (hipGetLastError() != hipSuccess) {
        std::fprintf(stderr, "mapf_encoder_wgrad7: launch failed\n");
        return MAPF_ERR_HIP;
    }
    return MAPF_OK;
}

}  // extern "C"
