// mapf_inproj.hip -- the recurrent cell's input projection for the ACTOR's step (reference model.py:191 `self.recurrent(latent, hidden)`:
// the W_ih x half of the GRUCell; its bias is added inside csrc/mapf_recur.hip), for a LIST of rows:
//     gi[row] = W_ih latent[row]      for row in row_list[0 .. *row_count)    (or every row)
//
// Why not the library GEMM that the learner's forward uses: gi depends on the latent only, and an agent whose observation did not
// change keeps its latent (fused.LatentCache) -- so it keeps its gi row too, and the actor's projection shrinks with the number of
// agents that moved, like its encoder pass (the row list and its device-side count are the ones mapf_obs_changed produced; no host
// read).  hipBLASLt has no device-side row count or gather, ran this shape at 0.55 PF/s in 32 k-row chunks (a larger call selects a
// stream-K kernel whose workgroups spin on peers: fused.mm_rows), and cost 36 us of a 280 us curriculum iteration on 12 k rows.
//
// One workgroup (8 waves) per 64 listed rows: the rows' latents are gathered into LDS (bf16 [64][800], K padded from 784 with zeros;
// row pitch 1616 B = an odd number of 16-byte chunks: conflict-free ds_read_b128 over 16 rows), wave w owns output tiles 6 w .. 6 w + 5
// of the 48 (16 channels each) for all four row tiles: v_mfma_f32_16x16x32_bf16 with A = weights in fragment order (one wave load = a
// contiguous 1 KiB, next k-step requested before the current one's MFMAs), B = latents from LDS; an accumulator lane holds 4
// consecutive channels of one row = one 8-byte store.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "mapf_dqn.h"
#include "mapf_env.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int K_IN = 784, K_PAD = 800, KS = K_PAD / 32;  // 25 k-steps
constexpr int N_OUT = 768, TILES = N_OUT / 16;            // 48 output tiles
constexpr int RB = 64;                                    // rows per workgroup
constexpr int NTHR = 512;
constexpr int ROWP = K_PAD * 2 + 16;                      // 1616 bytes: 101 chunks (odd)
constexpr int LDS_BYTES = RB * ROWP + RB * 4;
constexpr long long INPROJ_SPLIT_ROWS = 32768;  // up to here two workgroups share a row block, each with half of the output tiles: at 12.5 k rows
                                                // 14 / 19 / 33 us for 10 % / 60 % / all rows listed against 21 / 25 / 27 undivided (tools/micro/inproj_bench.py)
static_assert(TILES * KS * 512 == MAPF_INPROJ_PACKED_ELEMS, "header constant out of date");
static_assert((ROWP / 16) % 2 == 1, "row pitch must be an odd number of 16-byte chunks");

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// fp32 W_ih [768][784] -> bf16 fragments [tile][k-step][lane][8]: element j of lane l = W[16 tile + (l & 15)][32 kstep + 8 (l >> 4) + j]
__global__ void __launch_bounds__(256) inproj_pack_kernel(const float *__restrict__ w, uint16_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= TILES * KS * 512) return;
    const int j = i & 7, l = (i >> 3) & 63, kk = (i >> 9) % KS, t = (i >> 9) / KS;
    const int o = 16 * t + (l & 15), k = 32 * kk + 8 * (l >> 4) + j;
    const float v = k < K_IN ? w[o * K_IN + k] : 0.f;
    out[i] = (uint16_t)(pack2_bf16(v, 0.f) & 0xFFFFu);
}

// TPW output tiles per wave: 6 (a workgroup computes all 768 channels of its rows) or 2 / 3 (grid.y = 3 / 2 workgroups share a row
// block, each streaming a third / half of the weights: shorter workgroups for short lists, at the price of gathering the rows 3 / 2 x)
template <int TPW>
__global__ void __launch_bounds__(NTHR, 1) inproj_rows_kernel(const uint16_t *__restrict__ lat, long long num_rows, const int32_t *__restrict__ row_list,
                                                              const int32_t *__restrict__ row_count, const uint16_t *__restrict__ wp,
                                                              uint16_t *__restrict__ gi) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lh = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long total = num_rows;
    if (row_list != nullptr) {
        const long long c = (long long)*row_count;
        total = c < num_rows ? c : num_rows;
    }
    int32_t *srow = reinterpret_cast<int32_t *>(smem + RB * ROWP);
    // the grid is sized for num_rows; with a list the workgroups WALK it (most of them would only find out that there is nothing left)
    for (long long base = (long long)blockIdx.x * RB; base < total; base += (long long)gridDim.x * RB) {
        const int nrows = (int)((total - base) < RB ? (total - base) : RB);
        if (tid < RB) srow[tid] = tid < nrows ? (row_list ? row_list[base + tid] : (int32_t)(base + tid)) : -1;
        __syncthreads();
        // ---- gather the rows' latents: 98 chunks of data + 2 of zeros (K padding) per row; rows beyond nrows read as zeros ----
        for (int i = tid; i < RB * 100; i += NTHR) {
            const int r = i / 100, ch = i - 100 * r, row = srow[r];
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row >= 0 && ch < 98) v = *reinterpret_cast<const uint4 *>(lat + (long long)row * K_IN + ch * 8);
            *reinterpret_cast<uint4 *>(smem + r * ROWP + ch * 16) = v;
        }
        __syncthreads();
        // ---- TPW output tiles x 4 row tiles per wave ----
        const int tile0 = (int)blockIdx.y * 8 * TPW + TPW * w;  // this wave's first output tile
        f32x4 acc[TPW][4];
#pragma unroll
        for (int c = 0; c < TPW; ++c)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[c][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        int opaque0 = 0;  // (the weight addresses are the same in every pass of the walk: an opaque offset keeps the compiler from hoisting
        asm volatile("" : "+s"(opaque0));  //  all 150 fragment loads in front of the loop, as in csrc/mapf_recur.hip)
        const bf16x8 *wv = reinterpret_cast<const bf16x8 *>(wp + opaque0) + (long long)tile0 * KS * 64 + lane;
        bf16x8 a[2][TPW];
#pragma unroll
        for (int c = 0; c < TPW; ++c) a[0][c] = wv[(c * KS) * 64];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk + 1 < KS) {
#pragma unroll
                for (int c = 0; c < TPW; ++c) a[(kk + 1) & 1][c] = wv[(c * KS + kk + 1) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);  // (the next k-step's loads stay in front of this one's MFMAs, and no further ahead)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(smem + (16 * rt + lr) * ROWP + (32 * kk + 8 * lh) * 2);
#pragma unroll
                for (int c = 0; c < TPW; ++c) acc[c][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk & 1][c], b, acc[c][rt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- store: lane (lr, lh) holds channels 16 tile + 4 lh .. + 3 of row 16 rt + lr ----
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int row = srow[16 * rt + lr];
            if (row < 0) continue;
            uint16_t *dst = gi + (long long)row * N_OUT + 16 * tile0 + 4 * lh;
#pragma unroll
            for (int c = 0; c < TPW; ++c)
                *reinterpret_cast<uint2 *>(dst + 16 * c) = make_uint2(pack2_bf16(acc[c][rt][0], acc[c][rt][1]), pack2_bf16(acc[c][rt][2], acc[c][rt][3]));
        }
        __syncthreads();  // every wave is done with the LDS image before the next pass overwrites it
    }
}

}  // namespace

extern "C" {

int mapf_input_proj_pack(const float *w_ih_dev, uint16_t *packed_dev, void *stream) {
    if (!w_ih_dev || !packed_dev || (reinterpret_cast<uintptr_t>(packed_dev) & 15)) return MAPF_ERR_INVALID_ARG;
    hipLaunchKernelGGL(inproj_pack_kernel, dim3((TILES * KS * 512 + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w_ih_dev, packed_dev);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_input_proj_rows(const uint16_t *latent_dev, int64_t num_rows, const int32_t *row_list_dev, const int32_t *row_count_dev,
                         const uint16_t *packed_dev, uint16_t *gi_dev, void *stream) {
    if (num_rows < 0 || !latent_dev || !packed_dev || !gi_dev || (row_list_dev != nullptr) != (row_count_dev != nullptr) ||
        (reinterpret_cast<uintptr_t>(latent_dev) & 15) || (reinterpret_cast<uintptr_t>(packed_dev) & 15) || (reinterpret_cast<uintptr_t>(gi_dev) & 7))
        return MAPF_ERR_INVALID_ARG;
    if (num_rows == 0) return MAPF_OK;
    long long blocks = (num_rows + RB - 1) / RB;
    if (row_list_dev != nullptr && blocks > 1024) blocks = 1024;  // (walked: see the kernel)
    if (blocks > 0x7FFFFFFFLL) return MAPF_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    static const int force = std::getenv("MAPF_INPROJ_TPW") ? std::atoi(std::getenv("MAPF_INPROJ_TPW")) : 0;  // (tuning only)
    const int tpw = force ? force : (num_rows <= INPROJ_SPLIT_ROWS ? 3 : 6);
    if (tpw == 2)
        hipLaunchKernelGGL(inproj_rows_kernel<2>, dim3((unsigned)blocks, 3), dim3(NTHR), 0, st, latent_dev, (long long)num_rows, row_list_dev, row_count_dev,
                           packed_dev, gi_dev);
    else if (tpw == 3)
        hipLaunchKernelGGL(inproj_rows_kernel<3>, dim3((unsigned)blocks, 2), dim3(NTHR), 0, st, latent_dev, (long long)num_rows, row_list_dev, row_count_dev,
                           packed_dev, gi_dev);
    else
        hipLaunchKernelGGL(inproj_rows_kernel<6>, dim3((unsigned)blocks, 1), dim3(NTHR), 0, st, latent_dev, (long long)num_rows, row_list_dev, row_count_dev,
                           packed_dev, gi_dev);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

}  // extern "C"
