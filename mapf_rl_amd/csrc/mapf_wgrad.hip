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
//  * A workgroup (512 threads = 8 waves, two per SIMD) owns one of TWO column slabs [128 co] x [576 of the 1152
//    (tap, ci) columns] for a partition of the observations; wave w holds 64 co x 144 columns = 4 x 9 tiles of 16x16
//    (144 accumulator registers).  Three narrower slabs at one wave per SIMD (the first version) pull the same 25 KB
//    per observation into 1.5x as many CUs -- the kernel is bound by that L2 -> CU stream, not by MFMA or LDS.
//  * The MFMA is inline asm with the destination TIED to the accumulator: hipcc does not tie the builtin's, and a loop
//    that carries its accumulators then needs register-shuffle space (v_accvgpr_mov storms at one wave per SIMD,
//    hundreds of spills at 256 registers).  asm is opaque to hipcc's hazard padding: operands come from LDS reads (the
//    wait-count pass still sees the registers), and two s_nop 15 precede the epilogue's accumulator reads.
//  * Both operands need K (= position) along the fragment's register axis while memory has channels contiguous:
//    ds_read_b64_tr_b16 (hardware transpose read) delivers a [4 positions x 16 channels] block column-major.  Rows
//    are 288 B apart (256 + 32) so that the 8 consecutive rows a half-wave reads fall in distinct banks.
//  * K slot k = 8*y + x of an observation (64 slots = 2 k-steps, 49 of them real).  The INPUT lives in the same
//    zero-bordered 8-wide image as in the forward kernel (row 9 + 8*y + x), so the row of tap (ky,kx) is a constant
//    offset 8*ky + kx from the slot; gz is stored dense (row 7*y + x) and every lane points the slots that are
//    padding (x = 7, y = 7) at one all-zero row -- the transposed read takes a row address per lane anyway.
//  * One observation (72 MFMAs per wave) per step through a ring of FOUR LDS buffers filled by global_load_lds_dwordx4
//    (HBM -> LDS without passing through registers; lane i of a wave writes LDS bytes [16 i, 16 i + 16) of a 1-KiB
//    chunk from ANY global address, masked lanes write nothing -- tools/micro/lds_direct_load.hip): while observation i
//    is multiplied, i+1 and i+2 are resident and i+3 is in flight for two whole steps (~2.5 us, the HBM latency under
//    this load).  The first version staged through 16 registers per lane and a third buffer: the data of i+2 had one
//    step to arrive and the kernel sat at 1.53 ms per layer against 0.99 ms for its MFMAs alone.  The loads are inline
//    asm: hipcc's wait-count pass would make every later LDS read wait for a builtin LDS-DMA load it cannot disambiguate.
//    Every wave issues exactly 4 chunk loads per observation, so `s_waitcnt vmcnt(4)` at the end of a step means "all but
//    the newest observation have landed".
//  * The two slabs of one observation partition run on the same XCD (ids i, i+8 share an L2), so the
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

#ifndef MAPF_WGRAD_ABLATE  // diagnostic builds only (tools/micro/wgrad_ablate.py): 1 = no staging, 2 = no fragment reads, 4 = no barrier
#define MAPF_WGRAD_ABLATE 0
#endif
constexpr int SLABS = MAPF_ENC_WGRAD_SLABS;   // column slabs of the 128 x 1152 output: one workgroup each per partition (2)
constexpr int NTHR = 512;                    // 8 waves = 2 co halves x 4 column quarters of the slab
constexpr int CT = 4;                        // 16-row co tiles per wave (64 co)
constexpr int NTN = 1152 / SLABS / 4 / 16;    // 16-column tiles per wave (9: 144 columns); 144 accumulator registers
static_assert(NTN * 16 * 4 * SLABS == 1152 && (2 * NTN) % 3 == 0 && NTN >= CT, "");
constexpr int WROW = 288;                    // LDS bytes per position row
constexpr int GZ_ZERO_ROW = 49;              // dense gz rows 0..48 = positions 7*y + x; row 49 stays zero (padding K slots)
constexpr int IN_ROWS = 64 + 18;             // input image rows reachable through the 9 taps
constexpr int GZ_BYTES = 15 * 1024;          // 50 rows of 288 B, rounded up to whole 1-KiB load chunks
constexpr int OBS_BYTES = GZ_BYTES + IN_ROWS * WROW;  // 38,976
constexpr int NBUF = 4;                      // ring: computing obs i, obs i+1 and i+2 resident, obs i+3 in flight
static_assert(NBUF * OBS_BYTES <= 160 * 1024 && (GZ_ZERO_ROW + 1) * WROW <= GZ_BYTES, "LDS budget");
// direct-to-LDS chunks per observation: gz rows 0..48 = bytes [0, 14112) -> chunks 0..13; input rows 9..63 (the interior)
// = bytes [2592, 18432) of the input region -> its chunks 2..17; 30 chunks + 2 harmless repeats = 8 waves x 4
constexpr int LPW = 4;
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

__global__ void __launch_bounds__(NTHR, 1) encoder_wgrad_kernel(const uint16_t *__restrict__ gz, const uint16_t *__restrict__ ain,
                                                              long long M, float *__restrict__ ws) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * OBS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int slab = slot % SLABS, part = (slot / SLABS) * 8 + xcd;
    const int chalf = w & 1, nq = w >> 1;  // this wave's 64 output channels / 144 columns of the slab

    // observations of this partition
    const long long per = (M + MAPF_ENC_WGRAD_PARTS - 1) / MAPF_ENC_WGRAD_PARTS;
    const long long ob0 = per * part;
    long long nob = M - ob0;
    nob = nob < 0 ? 0 : (nob > per ? per : nob);

    for (int i = tid; i < NBUF * OBS_BYTES / 16; i += NTHR) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);

    // ---- staging: wave w issues chunk-loads j = w, w + 8, w + 16, w + 24 of every observation ----
    // per lane and load: the global byte offset inside the observation (of gz for j < 14 and the two repeats, of the input
    // otherwise) of the 16 bytes this lane's LDS slot holds, or -1 where the slot is row padding / a border row
    int soff[LPW];
#pragma unroll
    for (int k = 0; k < LPW; ++k) {
        const int jj = w + 8 * k;
        const bool isg = jj < 14 || jj >= 30;
        const int chunk = jj < 14 ? jj : (jj >= 30 ? jj - 30 : jj - 12);  // chunk index inside its region
        const int L = 1024 * chunk + 16 * lane, row = L / WROW, col = L - row * WROW;
        int off = -1;
        if (col < 256) {
            if (isg) {
                if (row < 49) off = row * 256 + col;
            } else {
                const int r = row - 9;
                if (r >= 0 && r < 56 && (r & 7) != 7) off = (7 * (r >> 3) + (r & 7)) * 256 + col;
            }
        }
        soff[k] = off;
    }
    typedef __attribute__((address_space(3))) unsigned char *lds_byte_ptr;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte_ptr)smem;
    auto issue_loads = [&](long long ob, int buf) __attribute__((always_inline)) {
        const uint16_t *gsrc = gz + ob * 6272, *asrc = ain + ob * 6272;
#pragma unroll
        for (int k = 0; k < LPW; ++k) {
            const int jj = w + 8 * k;
            const bool isg = jj < 14 || jj >= 30;
            const int chunk = jj < 14 ? jj : (jj >= 30 ? jj - 30 : jj - 12);
            const uint32_t dst = lds0 + buf * OBS_BYTES + (isg ? 0 : GZ_BYTES) + 1024 * chunk;
            const uint16_t *src = isg ? gsrc : asrc;
            if (soff[k] >= 0)
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(soff[k]), "s"(src) : "memory");
        }
    };

    // ---- fragment addresses of this lane (ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies row q, columns 4p..) ----
    const int li = lane & 15, lh = lane >> 4, q4 = li >> 2, p4 = li & 3;
    // k-slot (lh, j) of a 32-slot k-step: slot 4 lh + j for j < 4 (first read), 16 + 4 lh + (j - 4) for the second;
    // gz row of slot 8 y + x: 7 y + x, or the zero row for the padding slots
    int a_row[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int slot = 32 * ks + 16 * blk + 4 * lh + q4, y = slot >> 3, x = slot & 7;
            a_row[ks][blk] = ((x < 7 && y < 7) ? 7 * y + x : GZ_ZERO_ROW) * WROW + 8 * p4 + (CT * chalf) * 32;
        }
    // output column = (tap, ci): this wave's tile t covers columns col0(t) .. +16
    auto col0 = [&](int t) { return (1152 / SLABS) * slab + (16 * NTN) * nq + 16 * t; };
    int b_base[NTN];
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
        const int c = col0(t), tap = c >> 7, ci0 = c & 127;
        b_base[t] = GZ_BYTES + (4 * lh + q4 + 8 * (tap / 3) + tap % 3) * WROW + 8 * p4 + ci0 * 2;
    }

    f32x4 acc[CT][NTN];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < NTN; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();  // zero fill done
    if (!(MAPF_WGRAD_ABLATE & 1)) {
        if (nob > 0) issue_loads(ob0, 0);
        if (nob > 1) issue_loads(ob0 + 1, 1);
        if (nob > 2) {
            issue_loads(ob0 + 2, 2);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // observations 0 and 1 have landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();

    // fragments: the CT gz (A) tiles of the current k-step and of the next one; the input (B) tiles pass through a ring of
    // three (tile t's MFMAs run while tile t+2 is being read)
    bf16x8 af[2][CT], br[3];
    auto read_a = [&](const unsigned char *sb, int ks, int c) { return tr_read2(sb + a_row[ks][0] + c * 32, sb + a_row[ks][1] + c * 32); };
    auto read_b = [&](const unsigned char *sb, int ks, int t) { return tr_read2(sb + b_base[t] + (32 * ks) * WROW, sb + b_base[t] + (32 * ks + 16) * WROW); };
    if (nob > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) af[0][c] = read_a(smem, 0, c);
        br[0] = read_b(smem, 0, 0);
        br[1] = read_b(smem, 0, 1);
    }

    // One observation (step i): 2 k-steps x 48 MFMAs on buffer `b`; the fragments of its second k-step, then of the
    // next observation's first k-step (buffer b1, complete since the last barrier), are read behind the MFMAs.
    // Staging: the loads of observation i + 3 are issued at the start of the step into the buffer that step i - 1 read
    // (free since its barrier); the step ends by waiting for observation i + 2 (issued one step earlier) and a barrier.
    auto one_obs = [&](long long i, auto B, auto B1, auto B3) __attribute__((always_inline)) {
        constexpr int b = decltype(B)::value, b1 = decltype(B1)::value, b3 = decltype(B3)::value;
        const unsigned char *sb = smem + b * OBS_BYTES, *sb1 = smem + b1 * OBS_BYTES;
        const bool more = i + 3 < nob && !(MAPF_WGRAD_ABLATE & 1);
        if (more) issue_loads(ob0 + i + 3, b3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int cur = ks, nxt = ks ^ 1;
            // the k-step after this one: (this observation, ks 1) or (next observation, ks 0; past the last observation sb1
            // holds stale but valid LDS: harmless)
            const unsigned char *nbuf = ks == 0 ? sb : sb1;
#pragma unroll
            for (int t = 0; t < NTN; ++t) {
                const int g = ks * NTN + t;  // 2 * NTN tiles per observation, a multiple of 3: ring slots repeat per observation
                if (!(MAPF_WGRAD_ABLATE & 2)) {
                    if (t + 2 < NTN)
                        br[(g + 2) % 3] = read_b(sb, ks, t + 2);
                    else
                        br[(g + 2) % 3] = read_b(nbuf, nxt, t + 2 - NTN);
                    if (t < CT) af[nxt][t] = read_a(nbuf, nxt, t);
                }
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    // tied destination: hipcc does not tie the builtin's, and shuffles 4 registers per MFMA on the loop back-edge
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[c][t]) : "v"(af[cur][c]), "v"(br[g % 3]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // everything but the 4 loads just issued
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(MAPF_WGRAD_ABLATE & 4)) __syncthreads();
    };
    // Four observations per loop iteration (one turn of the buffer ring): every buffer offset is a constant, and the
    // accumulator shuffle hipcc emits on the loop back-edge (it does not tie an MFMA's destination to its source:
    // 192 v_accvgpr_mov per iteration, which HALVED the MFMA rate with one observation per iteration) is paid once
    // per 288 MFMAs.
    long long i = 0;
    for (; i + 4 <= nob; i += 4) {
        one_obs(i, I<0>{}, I<1>{}, I<3>{});
        one_obs(i + 1, I<1>{}, I<2>{}, I<0>{});
        one_obs(i + 2, I<2>{}, I<3>{}, I<1>{});
        one_obs(i + 3, I<3>{}, I<0>{}, I<2>{});
    }
    // tail (i % 4 == 0 here, so the ring is in its initial phase)
    if (i < nob) one_obs(i, I<0>{}, I<1>{}, I<3>{});
    if (i + 1 < nob) one_obs(i + 1, I<1>{}, I<2>{}, I<0>{});
    if (i + 2 < nob) one_obs(i + 2, I<2>{}, I<3>{}, I<1>{});

    // ---- partial sums of this partition: ws[part][co][tap = ky*3 + kx][ci] ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the asm MFMAs are opaque to hipcc's hazard padding: let the last ones retire
    float *out = ws + (long long)part * (128 * 9 * 128);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < NTN; ++t) {
            const int cc = col0(t), tap = cc >> 7, ci = (cc & 127) + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * (CT * chalf + c) + 4 * lh + r;
                out[(co * 9 + tap) * 128 + ci] = acc[c][t][r];
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
    hipLaunchKernelGGL(encoder_wgrad_kernel, dim3(SLABS * MAPF_ENC_WGRAD_PARTS), dim3(NTHR), 0, static_cast<hipStream_t>(stream), gz_dev,
                       in_dev, (long long)M, partial_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

}  // extern "C"
