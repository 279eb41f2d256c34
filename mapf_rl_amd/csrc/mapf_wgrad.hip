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
//  * One observation (64 K rows = 2 k-steps, 96 MFMAs per wave) per step through a ring of three LDS buffers: while
//    observation i is multiplied, i+1 is already resident (its first fragments are prefetched during i's last k-step,
//    so the hand-over exposes nothing but the barrier) and i+2 travels HBM -> registers -> the third buffer.
//  * The three ky-slabs of one observation partition run on the same XCD (ids i, i+8, i+16 share an L2), so the
//    operands come from HBM once.
//  * Output: per-partition partial sums fp32 [P][128][3][3][128]; the caller adds the P slabs (deterministic).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <type_traits>

#include "mapf_dqn.h"
#include "mapf_env.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int V>
using I = std::integral_constant<int, V>;

#ifndef MAPF_WGRAD_SETS  // register sets of the HBM -> LDS staging: 1 = prefetch distance 2 observations, 2 = distance 3
#define MAPF_WGRAD_SETS 1
#endif
#ifndef MAPF_WGRAD_ABLATE  // diagnostic builds only (tools/micro/wgrad_ablate.py): 1 = no staging, 2 = no fragment reads, 4 = no barrier
#define MAPF_WGRAD_ABLATE 0
#endif
constexpr int WROW = 288;                    // LDS bytes per position row
constexpr int GZ_ROWS = 64;                  // one observation = 64 rows of K (2 k-steps of 32), 49 of them non-zero
constexpr int IN_ROWS = 64 + 18;             // input image rows reachable through the 9 taps
constexpr int GZ_BYTES = GZ_ROWS * WROW;
constexpr int OBS_BYTES = (GZ_ROWS + IN_ROWS) * WROW;  // 42,048
constexpr int NBUF = 3;                      // ring: computing obs i, obs i+1 resident, obs i+2 being written
static_assert(NBUF * OBS_BYTES <= 160 * 1024, "LDS budget");
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
    __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * OBS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int ky = slot % 3, part = (slot / 3) * 8 + xcd;

    // observations of this partition
    const long long per = (M + MAPF_ENC_WGRAD_PARTS - 1) / MAPF_ENC_WGRAD_PARTS;
    const long long ob0 = per * part;
    long long nob = M - ob0;
    nob = nob < 0 ? 0 : (nob > per ? per : nob);

    for (int i = tid; i < NBUF * OBS_BYTES / 16; i += 256) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);

    // ---- staging: an observation is 784 16-byte chunks of gz and 784 of the input (49 positions x 16) ----
    // loads 0-2 / 4-6: chunk tid + 256 i of gz / of the input (wave-uniform base + tid*16: no per-load address math);
    // load 3: the last 16 chunks of each tensor (position 48) on threads 0-15 / 16-31, a harmless duplicate elsewhere.
    // At one wave per SIMD every VALU instruction of the staging code delays an MFMA, hence this shape.
    int dst[3];  // LDS byte offset of gz chunk tid + 256 i inside an observation buffer; the input chunk sits GZ_BYTES + 9 rows further
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = (tid >> 4) + 16 * i, y = q / 7, x = q - 7 * y;
        dst[i] = (8 * y + x) * WROW + (tid & 15) * 16;
    }
    constexpr int IN_SHIFT = GZ_BYTES + 9 * WROW;
    const bool tail_in = (tid >> 4) & 1;  // load 3: threads 16-31 carry the input tensor's position 48
    const int dst3 = (8 * 6 + 6) * WROW + (tid & 15) * 16 + (tail_in ? IN_SHIFT : 0);
    const int v16 = tid * 16, v16t = (tid & 15) * 16;
    // (plain structs of seven named registers and compile-time set selection: hipcc puts an ARRAY that is passed to a
    // lambda or indexed by a lambda parameter into scratch memory or LDS)
    struct Stage {
        u32x4 g0, g1, g2, t3, a0, a1, a2;  // native vectors: first-class SSA values (HIP's uint4 is a struct with unions)
    };
    Stage stg0;
#if MAPF_WGRAD_SETS == 2
    Stage stg1;  // observation j travels in set j % 2 and is loaded TWO steps before it is written to LDS
#endif
    auto load_into = [&](Stage &st, long long ob) __attribute__((always_inline)) {
        const char *g = reinterpret_cast<const char *>(gz + ob * 6272), *a = reinterpret_cast<const char *>(ain + ob * 6272);
        st.g0 = *reinterpret_cast<const u32x4 *>(g + v16);
        st.g1 = *reinterpret_cast<const u32x4 *>(g + 4096 + v16);
        st.g2 = *reinterpret_cast<const u32x4 *>(g + 8192 + v16);
        st.t3 = *reinterpret_cast<const u32x4 *>((tail_in ? a : g) + 12288 + v16t);
        st.a0 = *reinterpret_cast<const u32x4 *>(a + v16);
        st.a1 = *reinterpret_cast<const u32x4 *>(a + 4096 + v16);
        st.a2 = *reinterpret_cast<const u32x4 *>(a + 8192 + v16);
    };
    auto store_from = [&](const Stage &st, int buf) __attribute__((always_inline)) {
        unsigned char *base = smem + buf * OBS_BYTES;
        *reinterpret_cast<u32x4 *>(base + dst[0]) = st.g0;
        *reinterpret_cast<u32x4 *>(base + dst[1]) = st.g1;
        *reinterpret_cast<u32x4 *>(base + dst[2]) = st.g2;
        *reinterpret_cast<u32x4 *>(base + dst[0] + IN_SHIFT) = st.a0;
        *reinterpret_cast<u32x4 *>(base + dst[1] + IN_SHIFT) = st.a1;
        *reinterpret_cast<u32x4 *>(base + dst[2] + IN_SHIFT) = st.a2;
        if (tid < 32) *reinterpret_cast<u32x4 *>(base + dst3) = st.t3;
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
    if (nob > 0) {
        load_into(stg0, ob0);
        store_from(stg0, 0);
    }
    if (nob > 1) {
        load_into(stg0, ob0 + 1);
        store_from(stg0, 1);
    }
#if MAPF_WGRAD_SETS == 2
    if (nob > 2) load_into(stg0, ob0 + 2);  // written to buffer 2 at the end of step 0, like every later observation
#endif  // written to buffer 2 at the end of step 0, like every later observation
    __syncthreads();

    bf16x8 af[2][8], bfr[2][6];
    auto read_a = [&](const unsigned char *sb, int ks, int c) { return tr_read2(sb + a_base + (32 * ks) * WROW + c * 32, sb + a_base + (32 * ks + 16) * WROW + c * 32); };
    auto read_b = [&](const unsigned char *sb, int ks, int t) { return tr_read2(sb + b_base[t] + (32 * ks) * WROW, sb + b_base[t] + (32 * ks + 16) * WROW); };
    if (nob > 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) af[0][c] = read_a(smem, 0, c);
#pragma unroll
        for (int t = 0; t < 6; ++t) bfr[0][t] = read_b(smem, 0, t);
    }

    // One observation (step i): 2 k-steps x 48 MFMAs on buffer `b`; the fragments of its second k-step, then of the
    // next observation's first k-step (buffer b1, complete since the last barrier), are read behind the MFMAs.
    // Staging: observation i + 2 is loaded HBM -> registers at the start of the step and written to the free buffer
    // b2 after the MFMAs, before the barrier.
    auto one_obs = [&](long long i, auto B, auto B1, auto B2, auto PAR) __attribute__((always_inline)) {
        constexpr int b = decltype(B)::value, b1 = decltype(B1)::value, b2 = decltype(B2)::value, par = decltype(PAR)::value;
        const unsigned char *sb = smem + b * OBS_BYTES, *sb1 = smem + b1 * OBS_BYTES;
#if MAPF_WGRAD_SETS == 2
        if (i + 3 < nob && !(MAPF_WGRAD_ABLATE & 1)) {
            if constexpr (par == 0)
                load_into(stg1, ob0 + i + 3);
            else
                load_into(stg0, ob0 + i + 3);
        }
#else
        if (i + 2 < nob && !(MAPF_WGRAD_ABLATE & 1)) load_into(stg0, ob0 + i + 2);  // in flight during this observation's MFMAs
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int cur = ks, nxt = ks ^ 1;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const int m = c * 6 + t;  // one of the 14 next fragments behind every third MFMA
                    if (m % 3 == 0 && m / 3 < 14 && !(MAPF_WGRAD_ABLATE & 2)) {
                        const int f = m / 3;
                        const unsigned char *src_buf = ks == 0 ? sb : sb1;  // past the last observation sb1 holds stale but valid LDS: harmless
                        if (f < 8)
                            af[nxt][f] = read_a(src_buf, nxt, f);
                        else
                            bfr[nxt][f - 8] = read_b(src_buf, nxt, f - 8);
                    }
                    acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cur][c], bfr[cur][t], acc[c][t], 0, 0, 0);
                    if (m % 3 == 2) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#if MAPF_WGRAD_SETS == 2
        if (i + 2 < nob && !(MAPF_WGRAD_ABLATE & 1)) {
            if constexpr (par == 0)
                store_from(stg0, b2);
            else
                store_from(stg1, b2);
        }
#else
        if (i + 2 < nob && !(MAPF_WGRAD_ABLATE & 1)) store_from(stg0, b2);
#endif
        if (!(MAPF_WGRAD_ABLATE & 4)) __syncthreads();
    };
    // Six observations per loop iteration (two turns of the buffer ring): every buffer offset is a constant, and the
    // accumulator shuffle hipcc emits on the loop back-edge (it does not tie an MFMA's
    // destination to its source: 192 v_accvgpr_mov per iteration, which HALVED the MFMA rate with one observation per
    // iteration) is paid once per 576 MFMAs.
    long long i = 0;
    for (; i + 6 <= nob; i += 6) {
        one_obs(i, I<0>{}, I<1>{}, I<2>{}, I<0>{});
        one_obs(i + 1, I<1>{}, I<2>{}, I<0>{}, I<1>{});
        one_obs(i + 2, I<2>{}, I<0>{}, I<1>{}, I<0>{});
        one_obs(i + 3, I<0>{}, I<1>{}, I<2>{}, I<1>{});
        one_obs(i + 4, I<1>{}, I<2>{}, I<0>{}, I<0>{});
        one_obs(i + 5, I<2>{}, I<0>{}, I<1>{}, I<1>{});
    }
    // tail (i % 6 == 0 here, so the ring and the register sets are in their initial phase)
    if (i < nob) one_obs(i, I<0>{}, I<1>{}, I<2>{}, I<0>{});
    if (i + 1 < nob) one_obs(i + 1, I<1>{}, I<2>{}, I<0>{}, I<1>{});
    if (i + 2 < nob) one_obs(i + 2, I<2>{}, I<0>{}, I<1>{}, I<0>{});
    if (i + 3 < nob) one_obs(i + 3, I<0>{}, I<1>{}, I<2>{}, I<1>{});
    if (i + 4 < nob) one_obs(i + 4, I<1>{}, I<2>{}, I<0>{}, I<0>{});

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
