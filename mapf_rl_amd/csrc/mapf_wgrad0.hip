// mapf_wgrad0.hip -- weight gradient of the encoder's FIRST convolution (reference model.py:148, Conv2d(6, 128, 3, 1),
// valid padding on the 9x9 observation; the backward autograd derives for `Learner.train`, worker.py:316):
//
//     dW0[co][ci][ky][kx] = sum over observations m and output positions (y,x) in 7x7 of
//                           gz0[m][y][x][co] * obs[m][ci][y+ky][x+kx]
//
// gz0 = ReLU-masked pre-activation gradient of the layer (mapf_encoder_backward, layer 0), f16 [M][49][128]; obs the raw
// observations (uint8 / bool bytes or bf16 [M][6][9][9]).  As a GEMM the output is 128 x 54 with K = 49 M: far too small
// for a BLAS call per se, and the im2col matrix [49 M][54] a library GEMM wants is 0.65 GB written and read again (0.9 ms
// per update at 122,880 observations).  Here a workgroup keeps its 128 x 64 slab of the output in registers and streams
// its partition of the observations through LDS -- the kernel reads gz0 exactly once (1.5 GB) and is bound by that:
//  * gz0 of an observation travels HBM -> LDS with global_load_lds_dwordx4 (as in csrc/mapf_wgrad.hip: lane i of a wave
//    fills LDS bytes [16 i, 16 i + 16) of a 1-KiB chunk from any global address), the raw observation with
//    global_load_lds_ushort / _dword (one LDS dword per lane).  gz0 runs two observations ahead through a ring of three
//    15-KB slots, the raw bytes three ahead through a ring of four (the patch matrix is built one step early), 67 KB in
//    all: two workgroups per CU, one building patches or waiting at its barrier while the other multiplies.  Every wave
//    issues exactly 5 loads per step, so `s_waitcnt vmcnt(5)` = "everything issued before this step has landed".
//  * The patch matrix of an observation is built TRANSPOSED in LDS, PT[j = ci*9 + ky*3 + kx][k = 7y + x] (f16, 64 x 64,
//    rows >= 54 and positions >= 49 zero; columns permuted inside each 32 to the k order of the transposed read), one step ahead of its use (double buffer), so that the B fragment of the MFMA
//    (8 consecutive k of one column j) is a plain ds_read_b128; the A fragment (gz0^T: 8 consecutive k of one co) comes
//    from the position-major gz0 image through ds_read_b64_tr_b16, padding k slots pointed at an all-zero row.
//  * Wave w owns co tiles 2w, 2w+1 x the 4 column tiles: 8 accumulator tiles, 16 MFMAs per observation.
//  * Output: per-partition partial sums fp32 [P][128][64] (column j, 54 used); the caller adds the P slabs.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "mapf_dqn.h"
#include "mapf_env.h"

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 el8;  // the encoder kernels' element type is f16 (mapf_encoder.hip)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int NTHR = 256;
constexpr int PARTS = MAPF_ENC_WGRAD0_PARTS;
constexpr int WROW = 288;                 // gz image: bytes per position row (256 + 32: conflict-free transposed reads)
constexpr int GZ_ZERO_ROW = 49;
constexpr int GZ_BYTES = 15 * 1024;       // 50 rows, rounded up to whole 1-KiB load chunks
constexpr int RAW_BYTES = 1024;           // raw observation: 243 dwords (u8: two bytes per dword; bf16: two elements per dword)
constexpr int NGZ = 3, NRAW = 4;
constexpr int OFF_RAW = NGZ * GZ_BYTES;
constexpr int PT_ROW = 64 * 2 + 16;       // 144
constexpr int PT_BYTES = 64 * PT_ROW;
constexpr int OFF_PT = OFF_RAW + NRAW * RAW_BYTES;
constexpr int LDS_BYTES = OFF_PT + 2 * PT_BYTES;
static_assert(2 * LDS_BYTES <= 160 * 1024, "LDS budget: two workgroups per CU");

__device__ __forceinline__ el8 tr_read2(const unsigned char *p0, const unsigned char *p1) {
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p1));
    union {
        s16x4 h[2];
        el8 v;
    } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
}

__device__ __forceinline__ uint32_t f32_to_el_bits(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ uint32_t u8_to_el_bits(uint32_t v) { return f32_to_el_bits((float)v); }  // exact for 0..255
__device__ __forceinline__ uint32_t bf16_to_el_bits(uint32_t h) {  // observation given as bf16: clamped into f16's range
    return f32_to_el_bits(__builtin_amdgcn_fmed3f(__uint_as_float(h << 16), -65504.f, 65504.f));
}

template <typename InT>
__global__ void __launch_bounds__(NTHR, 2) conv0_wgrad_kernel(const uint16_t *__restrict__ gz, const InT *__restrict__ obs, long long M,
                                                            float *__restrict__ ws, const uint32_t *__restrict__ grad_scale,
                                                            const int32_t *__restrict__ valid_rows) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = blockIdx.x;
    if (valid_rows != nullptr) {  // (mapf_encoder_wgrad0_bounded: only the first *valid_rows <= M observations carry a gradient)
        const long long cnt = (long long)*valid_rows;
        M = cnt < M ? (cnt < 0 ? 0 : cnt) : M;
    }
    const long long per = (M + PARTS - 1) / PARTS;
    const long long ob0 = per * part;
    long long nob = M - ob0;
    nob = nob < 0 ? 0 : (nob > per ? per : nob);

    for (int i = tid; i < LDS_BYTES / 16; i += NTHR) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);

    // ---- staging: wave w issues gz chunk-loads jj = w, w + 4, w + 8, w + 12 (chunks 0..13 hold rows 0..48; 14, 15 repeat
    // chunks 0, 1) and dwords [64 w, 64 w + 64) of the raw observation ----
    int soff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int jj = w + 4 * k, chunk = jj < 14 ? jj : jj - 14;
        const int L = 1024 * chunk + 16 * lane, row = L / WROW, col = L - row * WROW;
        soff[k] = (col < 256 && row < 49) ? row * 256 + col : -1;
    }
    const int rdw = 64 * w + lane;  // raw dword of this lane (243 per observation)
    constexpr int RAW_STRIDE = sizeof(InT) == 1 ? 2 : 4;  // global bytes behind one LDS dword
    typedef __attribute__((address_space(3))) unsigned char *lds_byte_ptr;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte_ptr)smem;
    auto issue_gz = [&](long long ob, int slot) __attribute__((always_inline)) {
        const uint16_t *gsrc = gz + ob * 6272;
        const uint32_t base = lds0 + slot * GZ_BYTES;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jj = w + 4 * k, chunk = jj < 14 ? jj : jj - 14;
            const uint32_t dst = base + 1024 * chunk;
            if (soff[k] >= 0) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(soff[k]), "s"(gsrc) : "memory");
        }
    };
    auto issue_raw = [&](long long ob, int slot) __attribute__((always_inline)) {
        const InT *osrc = obs + ob * 486;
        const uint32_t rdst = lds0 + OFF_RAW + slot * RAW_BYTES + 256 * w;
        const int roff = rdw * RAW_STRIDE;
        if (rdw < 243) {
            if (sizeof(InT) == 1)
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_ushort %1, %2" ::"s"(rdst), "v"(roff), "s"(osrc) : "memory");
            else
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" ::"s"(rdst), "v"(roff), "s"(osrc) : "memory");
        }
    };
    // element e (0..485) of the staged observation as f16 bits
    auto raw_elem = [&](const unsigned char *raw, int e) -> uint32_t {
        if (sizeof(InT) == 1) return u8_to_el_bits(raw[(e >> 1) * 4 + (e & 1)]);
        return bf16_to_el_bits(*reinterpret_cast<const uint16_t *>(raw + 2 * e));
    };
    // PT[j][8 kc .. 8 kc + 8) for this thread's tasks (54 rows x 8 chunks = 432 tasks over 256 threads)
    auto build_pt = [&](int slot, int pb) __attribute__((always_inline)) {
        const unsigned char *raw = smem + OFF_RAW + slot * RAW_BYTES;
        unsigned char *pt = smem + OFF_PT + pb * PT_BYTES;
        for (int task = tid; task < 54 * 8; task += NTHR) {
            const int j = task >> 3, kc = task & 7;
            const int ci = j / 9, tap = j - 9 * ci, ky = tap / 3, kx = tap - 3 * ky;
            const int e0 = ci * 81 + ky * 9 + kx;
            uint32_t v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                // physical column 8 kc + q holds the k the A fragment has in the same register: the transposed read delivers
                // slots 4 lh + (0..3) and 16 + 4 lh + (0..3) of a 32-slot k-step to lane group lh = kc & 3
                const int k = 32 * (kc >> 2) + 4 * (kc & 3) + (q < 4 ? q : 12 + q), y = k / 7, x = k - 7 * y;
                v[q] = k < 49 ? raw_elem(raw, e0 + 9 * y + x) : 0u;
            }
            *reinterpret_cast<uint4 *>(pt + j * PT_ROW + kc * 16) =
                make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
        }
    };

    // ---- fragment addresses of this lane ----
    const int li = lane & 15, lh = lane >> 4, q4 = li >> 2, p4 = li & 3;
    int a_row[2][2];  // gz row (bytes) of k-slot 32 ks + 16 blk + 4 lh + q4: dense positions, padding slots -> the zero row
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int k = 32 * ks + 16 * blk + 4 * lh + q4;
            a_row[ks][blk] = (k < 49 ? k : GZ_ZERO_ROW) * WROW + 8 * p4 + (2 * w) * 32;
        }
    const int b_off = li * PT_ROW + lh * 16;  // + jt * 16 * PT_ROW + ks * 64

    f32x4 acc[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();  // zero fill done
    // prologue: gz 0, 1 and raw 0, 1, 2 (the step loop issues gz s + 2 and raw s + 3)
    for (int o = 0; o < 3; ++o) {
        if (o < 2 && o < nob) issue_gz(ob0 + o, o);
        if (o < nob) issue_raw(ob0 + o, o);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nob > 0) build_pt(0, 0);
    __syncthreads();

    int gs = 0;  // gz slot of observation s (s mod 3)
    for (long long s = 0; s < nob; ++s) {
        const int pb = (int)(s & 1);
        const int gs2 = gs == 0 ? 2 : gs - 1;  // (s + 2) mod 3: the slot step s - 1 read
        // loads of this step: exactly 5 per wave while the stream lasts (gz s + 2: 4, raw s + 3: 1); the tail issues fewer and
        // waits for everything
        const bool full = s + 3 < nob;
        if (s + 2 < nob) issue_gz(ob0 + s + 2, gs2);
        if (s + 3 < nob) issue_raw(ob0 + s + 3, (int)((s + 3) & 3));
        if (s + 1 < nob) build_pt((int)((s + 1) & 3), pb ^ 1);
        const unsigned char *gzb = smem + gs * GZ_BYTES, *ptb = smem + OFF_PT + pb * PT_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            el8 a[2], b[4];
#pragma unroll
            for (int c = 0; c < 2; ++c) a[c] = tr_read2(gzb + a_row[ks][0] + c * 32, gzb + a_row[ks][1] + c * 32);
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const el8 *>(ptb + b_off + t * 16 * PT_ROW + ks * 64);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c], b[t], acc[c][t], 0, 0, 0);
        }
        if (full)
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  // gz s + 1 and raw s + 2 (issued one step ago) have landed
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        gs = gs == 2 ? 0 : gs + 1;
    }

    // ---- partial sums of this partition: ws[part][co][j] ----
    const float inv_scale = grad_scale ? __uint_as_float(grad_scale[1]) : 1.f;  // gz carries the backward chain's loss scale
    float *out = ws + (long long)part * (128 * 64);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * (2 * w + c) + 4 * lh + r) * 64 + 16 * t + li] = acc[c][t][r] * inv_scale;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            std::fprintf(stderr, "mapf_wgrad0: %s failed: %s\n", #expr, hipGetErrorString(_e)); \
            return MAPF_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

}  // namespace

extern "C" {

static int wgrad0_launch(const uint16_t *gz0_dev, const void *obs_dev, int obs_dtype, int64_t M, const int32_t *valid_rows_dev,
                         const uint32_t *grad_scale_dev, float *partial_dev, void *stream) {
    if (M < 0 || !partial_dev || (M > 0 && (!gz0_dev || !obs_dev))) return MAPF_ERR_INVALID_ARG;
    if (obs_dtype != MAPF_ENC_OBS_U8 && obs_dtype != MAPF_ENC_OBS_BF16) return MAPF_ERR_INVALID_ARG;
    // u8 observations are fetched two bytes at a time (486 bytes each: every observation starts on an even address iff the base does)
    if ((reinterpret_cast<uintptr_t>(gz0_dev) & 15) || (reinterpret_cast<uintptr_t>(partial_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(obs_dev) & (obs_dtype == MAPF_ENC_OBS_U8 ? 1 : 3)))
        return MAPF_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // every partition writes its slab (zeros when it has no observations), so the caller's sum is always defined
    if (obs_dtype == MAPF_ENC_OBS_U8)
        hipLaunchKernelGGL(conv0_wgrad_kernel<uint8_t>, dim3(PARTS), dim3(NTHR), 0, st, gz0_dev, static_cast<const uint8_t *>(obs_dev),
                           (long long)M, partial_dev, grad_scale_dev, valid_rows_dev);
    else
        hipLaunchKernelGGL(conv0_wgrad_kernel<uint16_t>, dim3(PARTS), dim3(NTHR), 0, st, gz0_dev, static_cast<const uint16_t *>(obs_dev),
                           (long long)M, partial_dev, grad_scale_dev, valid_rows_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_encoder_wgrad0(const uint16_t *gz0_dev, const void *obs_dev, int obs_dtype, int64_t M, const uint32_t *grad_scale_dev,
                        float *partial_dev, void *stream) {
    return wgrad0_launch(gz0_dev, obs_dev, obs_dtype, M, nullptr, grad_scale_dev, partial_dev, stream);
}

int mapf_encoder_wgrad0_bounded(const uint16_t *gz0_dev, const void *obs_dev, int obs_dtype, int64_t M, const int32_t *valid_rows_dev,
                                const uint32_t *grad_scale_dev, float *partial_dev, void *stream) {
    if (!valid_rows_dev || (reinterpret_cast<uintptr_t>(valid_rows_dev) & 3)) return MAPF_ERR_INVALID_ARG;
    return wgrad0_launch(gz0_dev, obs_dev, obs_dtype, M, valid_rows_dev, grad_scale_dev, partial_dev, stream);
}

}  // extern "C"
