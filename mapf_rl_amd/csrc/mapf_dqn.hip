// mapf_dqn.hip -- fused conv epilogues of the observation encoder (see include/mapf_dqn.h).
// HBM-bound elementwise kernels: 16 bytes (8 bf16) per lane, fully coalesced; one read + one write of the
// activation instead of the 3-4 passes of separate bias / add / ReLU kernels.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "mapf_dqn.h"
#include "mapf_env.h"

namespace {

__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16(float f) {  // round to nearest even (inputs are finite)
    uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

__global__ void __launch_bounds__(256) bias_res_relu_fwd_kernel(uint4 *y, const float *bias, const uint4 *res, long long nvec,
                                                                int C) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const int c0 = (int)((v * 8) % C);
        uint4 a = y[v];
        uint4 r = make_uint4(0, 0, 0, 0);
        if (res) r = res[v];
        const float4 b0 = *reinterpret_cast<const float4 *>(bias + c0);
        const float4 b1 = *reinterpret_cast<const float4 *>(bias + c0 + 4);
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        uint32_t aw[4] = {a.x, a.y, a.z, a.w}, rw[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float lo = bf16_to_f32(aw[k] & 0xFFFFu) + bb[2 * k] + bf16_to_f32(rw[k] & 0xFFFFu);
            float hi = bf16_to_f32(aw[k] >> 16) + bb[2 * k + 1] + bf16_to_f32(rw[k] >> 16);
            lo = lo > 0.f ? lo : 0.f;
            hi = hi > 0.f ? hi : 0.f;
            aw[k] = f32_to_bf16(lo) | (f32_to_bf16(hi) << 16);
        }
        y[v] = make_uint4(aw[0], aw[1], aw[2], aw[3]);
    }
}

// every thread keeps the same 8-channel group (grid stride is a multiple of C/8), accumulates its bias
// gradient in registers, then one LDS reduction and C atomics per block
__global__ void __launch_bounds__(256) bias_res_relu_bwd_kernel(const uint4 *g, const uint4 *y, uint4 *gx, float *gbias,
                                                                long long nvec, int C) {
    __shared__ float s_part[256 * 8];
    const long long stride = (long long)gridDim.x * blockDim.x;  // multiple of C/8 (host guarantees)
    const long long v0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long v = v0; v < nvec; v += stride) {
        const uint4 gg = g[v], yy = y[v];
        uint32_t gw[4] = {gg.x, gg.y, gg.z, gg.w}, yw[4] = {yy.x, yy.y, yy.z, yy.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // y is a ReLU output: > 0 iff its bf16 bits are a positive non-zero number
            const uint32_t ylo = yw[k] & 0xFFFFu, yhi = yw[k] >> 16;
            const uint32_t mlo = (ylo != 0 && !(ylo & 0x8000u)) ? 0xFFFFu : 0u;
            const uint32_t mhi = (yhi != 0 && !(yhi & 0x8000u)) ? 0xFFFF0000u : 0u;
            gw[k] &= (mlo | mhi);
            acc[2 * k] += bf16_to_f32(gw[k] & 0xFFFFu);
            acc[2 * k + 1] += bf16_to_f32(gw[k] >> 16);
        }
        gx[v] = make_uint4(gw[0], gw[1], gw[2], gw[3]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s_part[threadIdx.x * 8 + k] = acc[k];
    __syncthreads();
    // thread t < C sums channel t over the threads whose group covers it
    const int groups = C / 8;  // threads with equal (tid % groups) share a channel group
    if ((int)threadIdx.x < C) {
        const int c = threadIdx.x, grp = c / 8, k = c % 8;
        // v0 % groups == (blockIdx*256 + tid) % groups; 256 % groups == 0 for groups in {2, 16}, so tid % groups decides
        float s = 0.f;
        for (int t = 0; t < 256; ++t) {
            const long long vt = (long long)blockIdx.x * blockDim.x + t;
            if ((int)(vt % groups) == grp) s += s_part[t * 8 + k];
        }
        atomicAdd(&gbias[c], s);
    }
}


// ---- communication mask (reference model.py:195-208): j is inside i's FOV square AND among i's `max_comm`
// nearest agents by Euclidean distance (itself included, ties -> lowest index).  One block per environment,
// one thread per agent i; the environment's positions sit in LDS.  Replaces a [E,N,N] int64 topk + scatter
// (1.3 ms per 4096x40 actor step) by one ~10 us pass. ----
constexpr int COMM_MAX_K = 8;

__global__ void __launch_bounds__(128) comm_mask_kernel(const short2 *__restrict__ pos, int N, int radius, int k,
                                                        uint8_t *__restrict__ mask, int32_t *__restrict__ packed, int cw,
                                                        const int4 *__restrict__ envtab) {
    __shared__ short2 s_pos[128];
    const int e = blockIdx.x, i = threadIdx.x;
    long long row0 = (long long)e * N, moff = (long long)e * N * N;
    if (envtab) {  // environments of different agent counts, their agents' rows back to back: {agents, first row, byte offset of the mask, -}
        const int4 d = envtab[e];
        N = d.x, row0 = d.y, moff = d.z;
        k = k < N ? k : N;
    }
    if (i < N) s_pos[i] = pos[row0 + i];
    __syncthreads();
    if (i >= N) return;
    const int px = s_pos[i].x, py = s_pos[i].y;
    int best[COMM_MAX_K];  // the k smallest keys d2*N + j, ascending
#pragma unroll
    for (int t = 0; t < COMM_MAX_K; ++t) best[t] = 0x7FFFFFFF;
    for (int j = 0; j < N; ++j) {
        const int dx = px - s_pos[j].x, dy = py - s_pos[j].y;
        int key = (dx * dx + dy * dy) * N + j;
#pragma unroll
        for (int t = 0; t < COMM_MAX_K; ++t) {  // insertion into the sorted list
            if (t < k) {
                const int lo = min(best[t], key);
                key = max(best[t], key);
                best[t] = lo;
            }
        }
    }
    int kth = 0;
#pragma unroll
    for (int t = 0; t < COMM_MAX_K; ++t)
        if (t == k - 1) kth = best[t];
    uint8_t *mrow = mask ? mask + moff + (long long)i * N : nullptr;
    int32_t *prow = packed ? packed + (row0 + i) * cw : nullptr;
    uint32_t word = 0;
    for (int j = 0; j < N; ++j) {
        const int dx = px - s_pos[j].x, dy = py - s_pos[j].y;
        const bool in_fov = abs(dx) <= radius && abs(dy) <= radius;
        const bool on = in_fov && ((dx * dx + dy * dy) * N + j) <= kth;
        if (mrow) mrow[j] = on ? 1 : 0;
        word |= (on ? 1u : 0u) << (j & 31);
        if ((j & 31) == 31 || j == N - 1) {
            if (prow) prow[j >> 5] = (int32_t)word;
            word = 0;
        }
    }
    if (prow)
        for (int w = (N + 31) / 32; w < cw; ++w) prow[w] = 0;
}

// ---- dueling Q head of the policy's forward (reference model.py:218: q = V + A - mean(A)) + its arg-max: sixteen lanes per agent row.
// hidden bf16 [rows][256]; adv.weight f32 [5][256], adv.bias [5], state.weight [256], state.bias [1]; fp32 accumulation.  As PyTorch
// operations under autocast this was a dozen launches of a few microseconds each (two linears, mean, add / sub, casts, arg-max) -- a
// sixth of an actor iteration at curriculum shapes. ----
// Sixteen lanes (one DPP row) per agent row, 16 channels each: the six dot products are reduced with DPP row rotations / quad
// permutations (VALU only).  (The first version spent a whole wavefront per row and reduced with 36 __shfl_xor = LDS-pipe
// instructions per row: 63 us for the 163,840 rows of config 2, whose 84 MB take 17 us to read.)
template <int CTRL>
__device__ __forceinline__ float qh_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float qh_row_sum16(float v) {
    v = qh_dpp_add<0x128>(v);  // row_ror:8
    v = qh_dpp_add<0x124>(v);  // row_ror:4
    v = qh_dpp_add<0x4E>(v);   // quad_perm:[2,3,0,1]
    return qh_dpp_add<0xB1>(v);  // quad_perm:[1,0,3,2]
}
__global__ void __launch_bounds__(256) q_head_kernel(const uint16_t *__restrict__ hidden, long long rows, const float *__restrict__ w_adv,
                                                     const float *__restrict__ b_adv, const float *__restrict__ w_st, const float *__restrict__ b_st,
                                                     float *__restrict__ q, long long *__restrict__ act) {
    const int q16 = threadIdx.x & 15;  // this lane's 16 channels: 16 q16 .. 16 q16 + 15
    const long long stride = (long long)gridDim.x * 16;
    float wa[5][16], ws[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#pragma unroll
        for (int k = 0; k < 5; ++k) wa[k][j] = w_adv[k * 256 + 16 * q16 + j];
        ws[j] = w_st[16 * q16 + j];
    }
    for (long long r0 = (long long)blockIdx.x * 16; r0 < rows; r0 += stride) {
        const long long r = r0 + (threadIdx.x >> 4);
        const bool live = r < rows;  // (the row's 16 lanes agree; the DPP reductions below run for every lane)
        uint4 hv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        if (live) {
            const uint4 *src = reinterpret_cast<const uint4 *>(hidden + r * 256 + 16 * q16);
            hv[0] = src[0];
            hv[1] = src[1];
        }
        const uint32_t hw[8] = {hv[0].x, hv[0].y, hv[0].z, hv[0].w, hv[1].x, hv[1].y, hv[1].z, hv[1].w};
        float d[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float h = (j & 1) ? __uint_as_float(hw[j >> 1] & 0xFFFF0000u) : __uint_as_float(hw[j >> 1] << 16);
#pragma unroll
            for (int k = 0; k < 5; ++k) d[k] += h * wa[k][j];
            d[5] += h * ws[j];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = qh_row_sum16(d[k]);
        if (live && q16 == 0) {
            float a[5], mean = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                a[k] = d[k] + b_adv[k];
                mean += a[k];
            }
            mean *= 0.2f;
            const float v = d[5] + b_st[0];
            int best = 0;
            float qb = v + a[0] - mean;
            q[r * 5] = qb;
#pragma unroll
            for (int k = 1; k < 5; ++k) {
                const float qk = v + a[k] - mean;
                q[r * 5 + k] = qk;
                if (qk > qb) {  // first maximum, as torch.argmax
                    qb = qk;
                    best = k;
                }
            }
            if (act) act[r] = best;
        }
    }
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            std::fprintf(stderr, "mapf_dqn: %s failed: %s\n", #expr, hipGetErrorString(_e)); \
            return MAPF_ERR_HIP;                                                           \
        }                                                                                  \
    } while (0)

int pick_grid(long long nvec, int groups) {
    long long blocks = (nvec + 255) / 256;
    if (blocks > 2048) blocks = 2048;  // grid-stride the rest (cdna guide: cap memory-bound grids at ~8 blocks/CU)
    // stride = blocks*256 must be a multiple of `groups`: 256 % groups == 0 for every supported C
    (void)groups;
    return (int)(blocks < 1 ? 1 : blocks);
}

// ---- which entries of a training window can reach its output (see mapf_dqn.h: mapf_window_relevance) ----
// One workgroup per window, thread j = agent j.  Walks the steps backwards: the set starts as {agent 0} at the window's last
// step, grows by two hops along the step's communication mask (agent i reads agent j where comm[i][j]), and is carried to the
// step before through every agent's own recurrent state.  As ~20 PyTorch launches per step this cost more host time than the
// pruned encoder launches it makes possible.
__global__ void __launch_bounds__(128) window_relevance_kernel(const uint8_t *__restrict__ comm, const long long *__restrict__ steps, int T, int B,
                                                               int N, uint8_t *__restrict__ rel) {
    // the set as two 64-bit words held by every thread (identical in all of them): a hop visits the few agents IN the set instead
    // of all N rows, and the only exchange is one ballot per wave and hop
    __shared__ unsigned long long s_ballot[2];
    const int b = blockIdx.x, j = threadIdx.x, wv = j >> 6;
    const long long last = steps[b] - 1;
    unsigned long long r0 = 0, r1 = 0;
    for (int t = T - 1; t >= 0; --t) {
        if (t == last) r0 |= 1ull;
        if (t <= last) {  // (uniform: before the window's last step the set is empty)
            const uint8_t *m = comm + ((size_t)b * T + t) * N * N;
            for (int round = 0; round < 2; ++round) {
                bool v = false;
                for (unsigned long long w = r0; w != 0 && j < N; w &= w - 1) v |= m[(__ffsll((long long)w) - 1) * N + j] != 0;
                for (unsigned long long w = r1; w != 0 && j < N; w &= w - 1) v |= m[(63 + __ffsll((long long)w)) * N + j] != 0;
                const unsigned long long bal = __ballot(v);
                __syncthreads();  // (the previous hop's words have been read)
                if ((j & 63) == 0) s_ballot[wv] = bal;
                __syncthreads();
                r0 |= s_ballot[0];
                r1 |= s_ballot[1];
            }
        }
        if (j < N) rel[((size_t)t * B + b) * N + j] = (uint8_t)(((j < 64 ? r0 >> j : r1 >> (j - 64)) & 1ull) != 0);
    }
}

}  // namespace

extern "C" {

int mapf_bias_res_relu_fwd(uint16_t *y_dev, const float *bias_dev, const uint16_t *res_dev, int64_t n, int C, void *stream) {
    if (!y_dev || !bias_dev || n < 0 || C < 8 || C % 8 || (256 % (C / 8)) || n % C) return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(y_dev) & 15) || (reinterpret_cast<uintptr_t>(res_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(bias_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    if (n == 0) return MAPF_OK;
    const long long nvec = n / 8;
    hipLaunchKernelGGL(bias_res_relu_fwd_kernel, dim3(pick_grid(nvec, C / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<uint4 *>(y_dev), bias_dev, reinterpret_cast<const uint4 *>(res_dev), nvec, C);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_bias_res_relu_bwd(const uint16_t *g_dev, const uint16_t *y_dev, uint16_t *gx_dev, float *gbias_dev, int64_t n, int C,
                           void *stream) {
    if (!g_dev || !y_dev || !gx_dev || !gbias_dev || n < 0 || C < 8 || C % 8 || C > 256 || (256 % (C / 8)) || n % C)
        return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(g_dev) & 15) || (reinterpret_cast<uintptr_t>(y_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(gx_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    if (n == 0) return MAPF_OK;
    const long long nvec = n / 8;
    hipLaunchKernelGGL(bias_res_relu_bwd_kernel, dim3(pick_grid(nvec, C / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint4 *>(g_dev), reinterpret_cast<const uint4 *>(y_dev),
                       reinterpret_cast<uint4 *>(gx_dev), gbias_dev, nvec, C);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_comm_mask(const int16_t *pos_dev, int E, int N, int obs_radius, int max_comm, uint8_t *mask_dev, int32_t *packed_dev,
                   int cw, void *stream) {
    if (!pos_dev || E < 0 || N < 1 || N > 128 || obs_radius < 0 || max_comm < 1 || max_comm > COMM_MAX_K) return MAPF_ERR_INVALID_ARG;
    if (!mask_dev && !packed_dev) return MAPF_ERR_INVALID_ARG;
    if (packed_dev && cw < (N + 31) / 32) return MAPF_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(pos_dev) & 3) return MAPF_ERR_INVALID_ARG;
    if (E == 0) return MAPF_OK;
    const int k = max_comm < N ? max_comm : N;
    hipLaunchKernelGGL(comm_mask_kernel, dim3(E), dim3(128), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const short2 *>(pos_dev), N, obs_radius, k, mask_dev, packed_dev, cw, (const int4 *)nullptr);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_comm_mask_multi(const int16_t *pos_dev, int E, const int32_t *envtab_dev, int obs_radius, int max_comm, uint8_t *mask_dev,
                         int32_t *packed_dev, int cw, void *stream) {
    if (E < 0 || obs_radius < 0 || max_comm < 1 || max_comm > COMM_MAX_K || !pos_dev || !envtab_dev || (!mask_dev && !packed_dev) || (packed_dev && cw < 1) ||
        (reinterpret_cast<uintptr_t>(pos_dev) & 3) || (reinterpret_cast<uintptr_t>(envtab_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    if (E == 0) return MAPF_OK;
    hipLaunchKernelGGL(comm_mask_kernel, dim3(E), dim3(128), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const short2 *>(pos_dev), 128,
                       obs_radius, max_comm, mask_dev, packed_dev, cw, reinterpret_cast<const int4 *>(envtab_dev));
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_q_head(const uint16_t *hidden_dev, int64_t rows, const float *adv_weight_dev, const float *adv_bias_dev, const float *state_weight_dev,
                const float *state_bias_dev, float *q_dev, int64_t *action_dev, void *stream) {
    if (rows < 0 || !hidden_dev || !adv_weight_dev || !adv_bias_dev || !state_weight_dev || !state_bias_dev || !q_dev ||
        (reinterpret_cast<uintptr_t>(hidden_dev) & 15) || (reinterpret_cast<uintptr_t>(action_dev) & 7))
        return MAPF_ERR_INVALID_ARG;
    if (rows == 0) return MAPF_OK;
    long long blocks = (rows + 15) / 16;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(q_head_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), hidden_dev, (long long)rows, adv_weight_dev,
                       adv_bias_dev, state_weight_dev, state_bias_dev, q_dev, reinterpret_cast<long long *>(action_dev));
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_window_relevance(const uint8_t *comm_dev, const int64_t *steps_dev, int T, int B, int N, uint8_t *rel_dev, void *stream) {
    if (T < 1 || B < 0 || N < 1 || N > 128 || !comm_dev || !steps_dev || !rel_dev) return MAPF_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(steps_dev) & 7) return MAPF_ERR_INVALID_ARG;
    if (B == 0) return MAPF_OK;
    hipLaunchKernelGGL(window_relevance_kernel, dim3(B), dim3(128), 0, static_cast<hipStream_t>(stream), comm_dev,
                       reinterpret_cast<const long long *>(steps_dev), T, B, N, rel_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

}  // extern "C"
