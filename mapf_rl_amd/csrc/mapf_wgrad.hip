// mapf_wgrad.hip -- weight gradient of the encoder's 3x3 128->128 convolutions (reference model.py:30-42, the
// backward of ResBlock.block1/block2 that autograd derives for `Learner.train`, worker.py:316) as one streaming
// MFMA kernel per layer (see include/mapf_dqn.h: mapf_encoder_wgrad).
//
//     dW[co][ky][kx][ci] = sum over observations m and positions (y,x) of
//                          gz[m][y][x][co] * a[m][y+ky-1][x+kx-1][ci]            (zero outside the 7x7 image)
//
// gz = ReLU-masked pre-activation gradient of the layer (mapf_encoder_backward_data), a = the layer's input
// (mapf_encoder_forward_save), both bf16 [M][49][128].  As a GEMM: 128 (co) x 1152 (tap, ci) outputs, K = all
// positions of all observations -- tiny output, enormous K, so the output is held in registers and the
// operands stream through LDS:
//  * A workgroup (256 threads, ONE wave per SIMD, up to 512 VGPRs) owns the slab [128 co] x [one ky: 3 kx x 128 ci]
//    for a partition of the observations: wave w holds 128 co x 96 columns = 8 x 6 tiles of 16x16 (192
//    accumulator registers).  LDS bytes per MFMA stay at 29 % of the LDS peak; a smaller per-wave block would
//    be LDS-bound.
//  * Both operands need K (= position) along the fragment's register axis while memory has channels contiguous:
//    ds_read_b64_tr_b16 (hardware transpose read) delivers a [4 positions x 16 channels] block column-major.  Rows
//    are 288 B apart (256 + 32) so that the 8 consecutive rows a half-wave reads fall in distinct banks.
//  * Positions live in the same zero-bordered 8-wide image as the forward kernel (row 64*obs + 8*y + x for gz,
//    +9 for a), so the input row of tap (ky,kx) is a constant offset 8*ky + kx from the gz row; border rows are zero
//    in gz, which makes the padded K positions contribute nothing.
//  * Two observations per stage, double-buffered: the next stage's 50 KB are loaded to registers while the current
//    stage's 192 MFMAs per wave run, and written to the other LDS buffer afterwards (one barrier per stage).
//  * The three ky-slabs of one observation partition run on the same XCD (ids i, i+8, i+16 share an L2), so the
//    operands come from HBM once.
//  * Output: per-partition partial sums fp32 [P][128][3][3][128]; the caller adds the P slabs (deterministic).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "mapf_dqn.h"
#include "mapf_env.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int WG_OBS = 2;                    // observations per stage
constexpr int WROW = 288;                    // LDS bytes per position row
constexpr int GZ_ROWS = 64 * WG_OBS;         // 128 rows = K per stage (4 k-steps of 32)
constexpr int IN_ROWS = 64 * WG_OBS + 18;    // input image rows reachable through the 9 taps
constexpr int GZ_BYTES = GZ_ROWS * WROW;
constexpr int STAGE_BYTES = (GZ_ROWS + IN_ROWS) * WROW;  // 78,912
static_assert(2 * STAGE_BYTES <= 160 * 1024, "LDS budget (double buffer)");
constexpr int ROWS_PER_STAGE = 49 * WG_OBS;              // rows of each tensor actually loaded
constexpr int CHUNKS = 2 * ROWS_PER_STAGE * 16;          // 16-byte chunks per stage (gz + input)
constexpr int LOADS = (CHUNKS + 255) / 256;              // per thread: 13
constexpr int PARTS_PER_XCD = MAPF_ENC_WGRAD_PARTS / 8;
static_assert(MAPF_ENC_WGRAD_PARTS % 8 == 0, "");

__device__ __forceinline__ bf16x8 tr_read2(const unsigned char *p0, const unsigned char *p1) {
    // two transposed 4-row blocks -> the 8 k-elements of one MFMA fragment
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p1));
    union {
        s16x4 h[2];
        bf16x8 v;
    } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
}

__global__ void __launch_bounds__(256, 1) encoder_wgrad_kernel(const uint16_t *__restrict__ gz, const uint16_t *__restrict__ ain,
                                                              long long M, float *__restrict__ ws) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int ky = slot % 3, part = (slot / 3) * 8 + xcd;

    // observation pairs of this partition
    const long long pairs = (M + WG_OBS - 1) / WG_OBS;
    const long long per = (pairs + MAPF_ENC_WGRAD_PARTS - 1) / MAPF_ENC_WGRAD_PARTS;
    const long long pair0 = per * part;
    long long nst = pairs - pair0;
    nst = nst < 0 ? 0 : (nst > per ? per : nst);

    for (int i = tid; i < 2 * STAGE_BYTES / 16; i += 256) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);

    // ---- staging geometry of this thread: chunk c = tid + 256 i -> (tensor, row, 16-byte column) ----
    int dst[LOADS];  // LDS byte offset inside a stage | observation-in-pair << 20 | tensor << 21; -1: no chunk
    int src[LOADS];  // bf16-element offset inside the stage's 49*WG_OBS rows of that tensor
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int c = tid + 256 * i;
        const bool have = c < CHUNKS;
        const int t = c / (ROWS_PER_STAGE * 16), cc = c - t * (ROWS_PER_STAGE * 16);
        const int rowi = cc >> 4, ch = cc & 15;
        const int o = rowi / 49, q = rowi - 49 * o, y = q / 7, x = q - 7 * y;
        const int off = (t ? GZ_BYTES + (64 * o + 8 * y + x + 9) * WROW : (64 * o + 8 * y + x) * WROW) + ch * 16;
        dst[i] = have ? (off | (o << 20) | (t << 21)) : -1;
        src[i] = cc * 8;
    }
    uint4 stg[LOADS];
    auto load_stage = [&](long long pair) {
        const long long ob = pair * WG_OBS;
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            stg[i] = make_uint4(0, 0, 0, 0);
            if (dst[i] >= 0 && ob + ((dst[i] >> 20) & 1) < M) {
                const uint16_t *base = ((dst[i] >> 21) & 1) ? ain : gz;
                stg[i] = *reinterpret_cast<const uint4 *>(base + ob * 6272 + src[i]);
            }
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i)
            if (dst[i] >= 0) *reinterpret_cast<uint4 *>(smem + buf * STAGE_BYTES + (dst[i] & 0xFFFFF)) = stg[i];
    };

    // ---- fragment addresses of this lane (ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies row q, columns 4p..) ----
    const int li = lane & 15, lh = lane >> 4, q4 = li >> 2, p4 = li & 3;
    // k-slot (lh, j) of a 32-row k-step: row 4 lh + j for j < 4 (first read), 16 + 4 lh + (j - 4) for the second
    const int a_base = (4 * lh + q4) * WROW + 8 * p4;
    int b_base[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int n0 = 96 * w + 16 * t, kx = n0 >> 7, ci0 = n0 & 127;
        b_base[t] = GZ_BYTES + (4 * lh + q4 + 8 * ky + kx) * WROW + 8 * p4 + ci0 * 2;
    }

    f32x4 acc[8][6];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();  // zero fill done
    if (nst > 0) {
        load_stage(pair0);
        store_stage(0);
    }
    __syncthreads();

    for (long long st = 0; st < nst; ++st) {
        const int buf = (int)(st & 1);
        const unsigned char *sb = smem + buf * STAGE_BYTES;
        const bool more = st + 1 < nst;
        if (more) load_stage(pair0 + st + 1);  // in flight during this stage's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 af[2][8], bfr[2][6];
        auto read_a = [&](int ks, int c) { return tr_read2(sb + a_base + (32 * ks) * WROW + c * 32, sb + a_base + (32 * ks + 16) * WROW + c * 32); };
        auto read_b = [&](int ks, int t) { return tr_read2(sb + b_base[t] + (32 * ks) * WROW, sb + b_base[t] + (32 * ks + 16) * WROW); };
#pragma unroll
        for (int c = 0; c < 8; ++c) af[0][c] = read_a(0, c);
#pragma unroll
        for (int t = 0; t < 6; ++t) bfr[0][t] = read_b(0, t);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    // next k-step's 14 fragments: one behind every third MFMA
                    const int i = c * 6 + t;
                    if (ks < 3 && i % 3 == 0 && i / 3 < 14) {
                        const int f = i / 3;
                        if (f < 8)
                            af[nxt][f] = read_a(ks + 1, f);
                        else
                            bfr[nxt][f - 8] = read_b(ks + 1, f - 8);
                    }
                    acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cur][c], bfr[cur][t], acc[c][t], 0, 0, 0);
                    if (i % 3 == 2) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) store_stage(buf ^ 1);
        __syncthreads();
    }

    // ---- partial sums of this partition: ws[part][co][ky][kx][ci] ----
    float *out = ws + (long long)part * (128 * 9 * 128);
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int n0 = 96 * w + 16 * t, kx = n0 >> 7, ci = (n0 & 127) + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * c + 4 * lh + r;
                out[((co * 3 + ky) * 3 + kx) * 128 + ci] = acc[c][t][r];
            }
        }
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            std::fprintf(stderr, "mapf_wgrad: %s failed: %s\n", #expr, hipGetErrorString(_e)); \
            return MAPF_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

}  // namespace

extern "C" {

int mapf_encoder_wgrad(const uint16_t *gz_dev, const uint16_t *in_dev, int64_t M, float *partial_dev, void *stream) {
    if (M < 0 || !partial_dev || (M > 0 && (!gz_dev || !in_dev))) return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(gz_dev) & 15) || (reinterpret_cast<uintptr_t>(in_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(partial_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    // every partition writes its slab (zeros when it has no observations), so the caller's sum is always defined
    hipLaunchKernelGGL(encoder_wgrad_kernel, dim3(3 * MAPF_ENC_WGRAD_PARTS), dim3(256), 0, static_cast<hipStream_t>(stream), gz_dev,
                       in_dev, (long long)M, partial_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

}  // extern "C"
