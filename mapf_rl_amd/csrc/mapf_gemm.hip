// mapf_gemm.hip -- the learner's remaining dense products and reductions as this library's own kernels (round 5; rounds 1-4 sent them
// to hipBLASLt / torch reductions: 12 Tensile launches + 18 reduce launches = 1.7 of 8.9 ms of an update's kernel time, and a
// stream-K library GEMM that spun on peers beside a second stream had to be chunked around, fused.mm_rows).  The backward pass of
// `Learner.train` (reference worker.py:312-324: loss.backward()) needs, besides the recurrence / encoder kernels:
//
//   mapf_tall_tn     weight gradients  dW[m][n] = sum_k dY[k][m] X[k][n]  with K = 10^4 .. 10^6 rows and a small output (the GRU cells,
//                    q|k|v, W_O, the input projection, the encoder's 1x1 head): both operands stored row-major with K as the ROW index,
//                    so both MFMA fragments come out of LDS through the transposing read (ds_read_b64_tr_b16); split over K into
//                    partitions whose fp32 partial slabs a second, small launch sums in partition order (deterministic);
//   mapf_sum_parts   out[i] = scale * sum_p parts[p][i]: the encoder's per-partition weight-gradient slabs (6 layers in one launch),
//                    its bias partials, conv0's slab with the column -> [co][ky][kx][ci] permutation;
//   mapf_proj_rows   y[row] = W x[row] for all rows, W a packed fragment image: the GRU's input projection W_ih (768 x 784) forward --
//                    csrc/mapf_inproj.hip's kernel, which this file generalises -- and its transpose for the gradient w.r.t. the
//                    latents (784 x 768).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "mapf_dqn.h"
#include "mapf_env.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// ------------------------------------------------------------------------------------------------------------------
// tall-skinny TN product
// ------------------------------------------------------------------------------------------------------------------
// Four waves in a WM x WN grid, each MT x NT tiles of 16 x 16: a workgroup owns an (WM MT 16) x (WN NT 16) slab of the output for one
// partition of K.  The slab shape sets the traffic: every slab re-streams its operand columns, so the large outputs use 128 x 128
// (64 x 64 slabs measured 186 us where the library took 54 on [26624] x 768 x 256: 327 MB of operand re-reads), the narrow ones
// shapes that fit them (16 x 128 for the encoder's 1x1 head, 64 x 128 for W_O, 128 x 64 for the update cell's input weight).
constexpr int T_THR = 256;
constexpr int TK = 32;  // K rows per block = one MFMA k-step; double-buffered in LDS
constexpr int TD = 4;   // blocks in flight per workgroup (registers)

template <int WM, int WN, int MT, int NT>
struct Slab {
    static constexpr int SM = WM * MT * 16, SN = WN * NT * 16;
    static constexpr int PA = SM * 2 + 32, PB = SN * 2 + 32;  // row pitch: the 8 rows a 32-lane group of ds_read_b64_tr_b16 touches land on distinct banks
    static constexpr int A_BYTES = TK * PA, B_BYTES = TK * PB, STAGE = A_BYTES + B_BYTES, LDS = 2 * STAGE;
    static constexpr int CA = SM / 8, CB = SN / 8;            // 16-byte chunks per row
    static constexpr int QA = (TK * CA + T_THR - 1) / T_THR, QB = (TK * CB + T_THR - 1) / T_THR;
    static_assert(WM * WN == 4 && LDS <= 64 * 1024, "");
};

// two transposed 4-row reads = the 8 k-elements of one MFMA fragment (lane 4 q + p of a 16-lane group supplies row q, columns 4 p ..
// of a [4 rows][16 columns] block and receives column (lane & 15)'s four rows): k-slot numbering  {4 lh + j | 16 + 4 lh + j}, the same
// for both operands
__device__ __forceinline__ uint4 tr_read2(const unsigned char *p0, const unsigned char *p1) {
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p1));
    union {
        s16x4 h[2];
        uint4 v;
    } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
}

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
    if (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// grid (slabs over n, slabs over m, P partitions of K).  ws: [slab][P][SM][SN] f32 (P > 1)
template <bool F16, int WM, int WN, int MT, int NT>
__global__ void __launch_bounds__(T_THR) tall_tn_kernel(const uint16_t *__restrict__ A, long long lda, const uint16_t *__restrict__ B, long long ldb,
                                                        long long K, int m, int n, float *__restrict__ ws,
                                                        float *__restrict__ out, const uint32_t *__restrict__ scale_bits, int accumulate) {
    using S = Slab<WM, WN, MT, NT>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S::LDS];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lh = lane >> 4, q4 = li >> 2, p4 = li & 3;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w / WN, wn = w - WN * wm;
    const int n0 = S::SN * blockIdx.x, m0 = S::SM * blockIdx.y, P = gridDim.z, part = blockIdx.z;
    const int slab = blockIdx.y * gridDim.x + blockIdx.x;
    const long long nblk = (K + TK - 1) / TK;
    const long long b_lo = nblk * part / P, b_hi = nblk * (part + 1) / P;

    // staging through registers, TD blocks deep: block b travels through slot b % TD into LDS buffer b & 1
    uint4 ra[TD][S::QA], rb[TD][S::QB];
    auto load_block = [&](long long blk, int slot) {
#pragma unroll
        for (int q = 0; q < S::QA; ++q) {
            const int c = tid + T_THR * q, row = c / S::CA, ch = c - S::CA * row;
            const long long k = blk * TK + row;
            ra[slot][q] = make_uint4(0, 0, 0, 0);
            if (row < TK && k < K && m0 + 8 * ch < m) ra[slot][q] = *reinterpret_cast<const uint4 *>(A + k * lda + m0 + 8 * ch);
        }
#pragma unroll
        for (int q = 0; q < S::QB; ++q) {
            const int c = tid + T_THR * q, row = c / S::CB, ch = c - S::CB * row;
            const long long k = blk * TK + row;
            rb[slot][q] = make_uint4(0, 0, 0, 0);
            if (row < TK && k < K && n0 + 8 * ch < n) rb[slot][q] = *reinterpret_cast<const uint4 *>(B + k * ldb + n0 + 8 * ch);
        }
    };
    auto store_block = [&](int slot) {  // register slot -> LDS buffer slot & 1
        unsigned char *sa = smem + (slot & 1) * S::STAGE, *sb = sa + S::A_BYTES;
#pragma unroll
        for (int q = 0; q < S::QA; ++q) {
            const int c = tid + T_THR * q, row = c / S::CA, ch = c - S::CA * row;
            if (row < TK) *reinterpret_cast<uint4 *>(sa + row * S::PA + 16 * ch) = ra[slot][q];
        }
#pragma unroll
        for (int q = 0; q < S::QB; ++q) {
            const int c = tid + T_THR * q, row = c / S::CB, ch = c - S::CB * row;
            if (row < TK) *reinterpret_cast<uint4 *>(sb + row * S::PB + 16 * ch) = rb[slot][q];
        }
    };

    f32x4 acc[MT][NT];  // a lane holds out[m = 16 i + li][n = 16 j + 4 lh .. + 3] of this wave's tiles
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fa_off = (4 * lh + q4) * S::PA + 8 * p4 + (16 * MT * wm) * 2, fb_off = (4 * lh + q4) * S::PB + 8 * p4 + (16 * NT * wn) * 2;
    auto compute = [&](int buf) {
        const unsigned char *sa = smem + buf * S::STAGE + fa_off, *sb = smem + buf * S::STAGE + S::A_BYTES + fb_off;
        uint4 fa[MT], fb[NT];
#pragma unroll
        for (int t = 0; t < MT; ++t) fa[t] = tr_read2(sa + 32 * t, sa + 16 * S::PA + 32 * t);
#pragma unroll
        for (int t = 0; t < NT; ++t) fb[t] = tr_read2(sb + 32 * t, sb + 16 * S::PB + 32 * t);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<F16>(fb[j], fa[i], acc[i][j]);  // D[M = n][N = m]: 4 consecutive n per lane
    };
    // Step of block b: block b + 1 goes from its slot into the other LDS buffer, block b + 1 + TD is requested into that slot, block b is
    // multiplied, barrier.  A request has TD steps to arrive: with few workgroups per CU (small outputs) nothing else hides the ~1 us
    // of a load from the Infinity Cache (two steps deep the loop ran at half a load latency per 32 K rows).
    const long long nb = b_hi - b_lo;
#pragma unroll
    for (int u = 0; u < TD; ++u)
        if (u < nb) load_block(b_lo + u, u);
    if (nb > 0) {
        store_block(0);
        if (TD < nb) load_block(b_lo + TD, 0);
    }
    __syncthreads();
    for (long long i = 0; i < nb; i += TD) {
#pragma unroll
        for (int u = 0; u < TD; ++u) {
            const long long b = i + u;
            if (b < nb) {  // (workgroup-uniform)
                if (b + 1 < nb) store_block((u + 1) % TD);
                if (b + 1 + TD < nb) load_block(b_lo + b + 1 + TD, (u + 1) % TD);
                compute(u & 1);
                __syncthreads();
            }
        }
    }

    const float scale = scale_bits ? __uint_as_float(scale_bits[1]) : 1.f;
    auto finish = [&](f32x4 v, int mm, int nn) {  // out[mm][nn .. nn + 3]
        if (mm >= m) return;
        float *o = out + (long long)mm * n + nn;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (nn + r < n) o[r] = (accumulate ? o[r] : 0.f) + v[r] * scale;
    };
    if (P == 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) finish(acc[i][j], m0 + 16 * (MT * wm + i) + li, n0 + 16 * (NT * wn + j) + 4 * lh);
        return;
    }
    // partial slab -> workspace; tall_sum_kernel (the next launch) adds the P partials in partition order.  (Round 5 first let the
    // last-arriving workgroup of a slab do it behind an agent-scope fence + counter: correct, and ~100 us slower -- every workgroup's
    // release fence writes its XCD's L2 back so that the other seven XCDs may read the slab.)
    constexpr int SLAB = S::SM * S::SN;
    float *mine = ws + ((long long)slab * P + part) * SLAB;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            *reinterpret_cast<f32x4 *>(mine + (16 * (MT * wm + i) + li) * S::SN + 16 * (NT * wn + j) + 4 * lh) = acc[i][j];
}

// out = [out +] scale * sum_p ws[slab][p]: grid (SM * SN / 1024, slabs); a thread owns one float4 of the slab
__global__ void __launch_bounds__(256) tall_sum_kernel(const float *__restrict__ ws, int P, int SM, int SN, int slabs_n, int m, int n,
                                                          float *__restrict__ out, const uint32_t *__restrict__ scale_bits, int accumulate) {
    const int slab = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x, row = c / (SN / 4), col = 4 * (c - (SN / 4) * row);
    const int mm = SM * (slab / slabs_n) + row, nn = SN * (slab % slabs_n) + col;
    if (row >= SM || mm >= m || nn >= n) return;
    const long long SLAB = (long long)SM * SN;
    const float *src = ws + (long long)slab * P * SLAB + (long long)row * SN + col;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int p = 0;
    for (; p + 8 <= P; p += 8) {  // eight loads in flight; added in partition order
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(src + (long long)(p + u) * SLAB);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < P; ++p) s += *reinterpret_cast<const f32x4 *>(src + (long long)p * SLAB);
    const float scale = scale_bits ? __uint_as_float(scale_bits[1]) : 1.f;
    float *o = out + (long long)mm * n + nn;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (nn + r < n) o[r] = (accumulate ? o[r] : 0.f) + s[r] * scale;
}

// which slab shape an output gets, and into how many partitions K is split
struct TallPlan {
    int cfg, sm, sn, slabs, parts;
};
inline TallPlan tall_plan(long long K, int m, int n) {
    TallPlan p;
    if (m <= 16) p.cfg = 1, p.sm = 16, p.sn = 128;         // the encoder's 1x1 head
    else if (m <= 64) p.cfg = 2, p.sm = 64, p.sn = 128;    // W_O
    else if (n <= 64) p.cfg = 3, p.sm = 128, p.sn = 64;    // update cell, input weight
    else p.cfg = 0, p.sm = 128, p.sn = 128;
    p.slabs = ((m + p.sm - 1) / p.sm) * ((n + p.sn - 1) / p.sn);
    const long long nblk = (K + TK - 1) / TK;
    static const int target = std::getenv("MAPF_TALL_WGS") ? std::atoi(std::getenv("MAPF_TALL_WGS")) : 384;  // (tuning; swept 256..1024 at the learner's shapes, tools/micro/tall_gemm_bench.py)
    long long P = (target + p.slabs - 1) / p.slabs;              // workgroups over the chip
    if (P > nblk / 8) P = nblk / 8;                              // at least eight K blocks per partition
    if (P > MAPF_TALL_TN_MAX_PARTS) P = MAPF_TALL_TN_MAX_PARTS;
    if (P < 1) P = 1;
    p.parts = (int)P;
    return p;
}

// ------------------------------------------------------------------------------------------------------------------
// sums over partitions
// ------------------------------------------------------------------------------------------------------------------
// out[g][i] = scale * sum_p parts[g][p][i], i < n (n a multiple of 4), G groups with their own output pointers
struct SumGroups {
    const float *parts[8];
    float *out[8];
};
__global__ void __launch_bounds__(256) sum_parts_kernel(SumGroups gr, int P, long long n, const uint32_t *__restrict__ scale_bits) {
    const long long i = 4 * ((long long)blockIdx.x * 256 + threadIdx.x);
    if (i >= n) return;
    const float *src = gr.parts[blockIdx.y] + i;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int p = 0;
    for (; p + 8 <= P; p += 8) {  // eight loads in flight; summed in partition order
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(src + (long long)(p + u) * n);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < P; ++p) s += *reinterpret_cast<const f32x4 *>(src + (long long)p * n);
    const float scale = scale_bits ? __uint_as_float(scale_bits[1]) : 1.f;
    *reinterpret_cast<f32x4 *>(gr.out[blockIdx.y] + i) = s * scale;
}

// conv0: ws0 [P][128][64] (columns j = ci*9 + ky*3 + kx, 54 used) -> the weight's memory [co][ky][kx][ci] (ci = 6)
__global__ void __launch_bounds__(256) sum_conv0_kernel(const float *__restrict__ ws0, int P, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // output element: co * 54 + (ky*3 + kx) * 6 + ci
    if (i >= 128 * 54) return;
    const int co = i / 54, r = i - 54 * co, tap = r / 6, ci = r - 6 * tap;
    const float *src = ws0 + co * 64 + ci * 9 + tap;
    float s = 0.f;
    int p = 0;
    for (; p + 16 <= P; p += 16) {  // sixteen loads in flight (one per iteration ran at a load latency per partition: 0.22 ms for 512)
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[(long long)(p + u) * (128 * 64)];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; p < P; ++p) s += src[(long long)p * (128 * 64)];
    out[i] = s;
}

// bias partials, two stages (one workgroup per output row walking all partials took 0.9 ms at 6,000 partials: a load latency per two of them).
// Stage 1, grid (8, BS): rows 0..6 = gb_part [7][nblk][128], row 7 = gb7_part [rows7][16] read as [rows7 / 8][128]; workgroup (r, s) adds
// partials s, s + BS, ... of its row: thread (sub, quad) takes every 8th of those for 4 columns, eight loads in flight, then the 8
// sub-sums are added through LDS in order -> scratch [8][BS][128].  Stage 2, grid 8: the BS slices in order -> the outputs.
constexpr int BS = 64;
__global__ void __launch_bounds__(256) sum_bias1_kernel(const float *__restrict__ gb_part, long long nblk, const float *__restrict__ gb7_part,
                                                        long long rows7, float *__restrict__ scratch) {
    __shared__ f32x4 red[256];
    const int tid = threadIdx.x, quad = tid & 31, sub = tid >> 5, r = blockIdx.x, sl = blockIdx.y;
    const long long n = r < 7 ? nblk : rows7 / 8;   // partial rows of 128 floats
    const float *src = (r < 7 ? gb_part + (long long)r * nblk * 128 : gb7_part) + 4 * quad;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    long long b = sl + (long long)BS * sub;
    const long long step = 8ll * BS;
    for (; b + 7 * step < n; b += 8 * step) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(src + (b + u * step) * 128);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < n; b += step) s += *reinterpret_cast<const f32x4 *>(src + b * 128);
    red[tid] = s;
    __syncthreads();
    if (sub == 0) {
        f32x4 t = red[quad];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[32 * k + quad];
        *reinterpret_cast<f32x4 *>(scratch + ((long long)r * BS + sl) * 128 + 4 * quad) = t;
    }
}
__global__ void __launch_bounds__(128) sum_bias2_kernel(const float *__restrict__ scratch, const float *__restrict__ gb7_part, long long rows7,
                                                        float *__restrict__ out7, float *__restrict__ out1) {
    __shared__ float red[128];
    const int c = threadIdx.x, r = blockIdx.x;
    float s = 0.f;
#pragma unroll 16
    for (int k = 0; k < BS; ++k) s += scratch[((long long)r * BS + k) * 128 + c];
    if (r < 7) {
        out7[r * 128 + c] = s;
        return;
    }
    // row 7: column c of the [rows7 / 8][128] view is column c & 15 of every 8th row (+ c >> 4) of gb7_part; the rows7 % 8 tail rows follow
    red[c] = s;
    __syncthreads();
    if (c < 16) {
        float t = 0.f;
        for (int k = 0; k < 8; ++k) t += red[16 * k + c];
        for (long long b = rows7 / 8 * 8; b < rows7; ++b) t += gb7_part[b * 16 + c];
        out1[c] = t;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// y[row] = W x[row], all rows (csrc/mapf_inproj.hip's scheme, for any K_IN / N_OUT; W as packed fragments [tile][k-step][lane][8])
// ------------------------------------------------------------------------------------------------------------------
template <int K_IN, int N_OUT>
struct Proj {
    static constexpr int K_PAD = (K_IN + 31) / 32 * 32, KS = K_PAD / 32, TILES = (N_OUT + 15) / 16;
    static constexpr int RB = 64, NTHR = 512;
    static constexpr int CH = K_IN / 8, CHP = K_PAD / 8;                       // 16-byte chunks of a row: data / padded
    static constexpr int ROWP = K_PAD * 2 + (((K_PAD / 8) % 2 == 0) ? 16 : 0);  // an odd number of 16-byte chunks: conflict-free ds_read_b128 over 16 rows
    static constexpr int LDS = RB * ROWP;
    static_assert(K_IN % 8 == 0 && (ROWP / 16) % 2 == 1 && LDS <= 160 * 1024, "");
};

// fp32 W [rows][cols] -> bf16 fragments of W (TRANS = false: outputs = rows of W, K = its columns) or of W^T (TRANS = true)
template <int K_IN, int N_OUT, bool TRANS>
__global__ void __launch_bounds__(256) proj_pack_kernel(const float *__restrict__ w, uint16_t *__restrict__ out) {
    using PJ = Proj<K_IN, N_OUT>;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= PJ::TILES * PJ::KS * 512) return;
    const int j = i & 7, l = (i >> 3) & 63, kk = (i >> 9) % PJ::KS, t = (i >> 9) / PJ::KS;
    const int o = 16 * t + (l & 15), k = 32 * kk + 8 * (l >> 4) + j;
    float v = 0.f;
    if (o < N_OUT && k < K_IN) v = TRANS ? w[(long long)k * N_OUT + o] : w[(long long)o * K_IN + k];
    out[i] = (uint16_t)(pack2_bf16(v, 0.f) & 0xFFFFu);
}

// TPW output tiles per wave; grid (row blocks, ceil(TILES / (8 TPW)))
template <int K_IN, int N_OUT, int TPW>
__global__ void __launch_bounds__(512, 1) proj_rows_kernel(const uint16_t *__restrict__ x, long long num_rows, const uint16_t *__restrict__ wp,
                                                           uint16_t *__restrict__ y) {
    using PJ = Proj<K_IN, N_OUT>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[PJ::LDS];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lh = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long base = (long long)blockIdx.x * PJ::RB;
    const int nrows = (int)((num_rows - base) < PJ::RB ? (num_rows - base) : PJ::RB);
    for (int i = tid; i < PJ::RB * PJ::CHP; i += PJ::NTHR) {
        const int r = i / PJ::CHP, ch = i - PJ::CHP * r;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < nrows && ch < PJ::CH) v = *reinterpret_cast<const uint4 *>(x + (base + r) * K_IN + ch * 8);
        *reinterpret_cast<uint4 *>(smem + r * PJ::ROWP + ch * 16) = v;
    }
    __syncthreads();
    const int tile0 = (int)blockIdx.y * 8 * TPW + TPW * w;  // this wave's first output tile
    if (tile0 >= PJ::TILES) return;
    f32x4 acc[TPW][4];
#pragma unroll
    for (int c = 0; c < TPW; ++c)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[c][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 *wv = reinterpret_cast<const bf16x8 *>(wp) + (long long)tile0 * PJ::KS * 64 + lane;
    bf16x8 a[2][TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c) a[0][c] = tile0 + c < PJ::TILES ? wv[(c * PJ::KS) * 64] : bf16x8{};
#pragma unroll
    for (int kk = 0; kk < PJ::KS; ++kk) {
        if (kk + 1 < PJ::KS) {
#pragma unroll
            for (int c = 0; c < TPW; ++c) a[(kk + 1) & 1][c] = tile0 + c < PJ::TILES ? wv[(c * PJ::KS + kk + 1) * 64] : bf16x8{};
        }
        __builtin_amdgcn_sched_barrier(0);  // (the next k-step's loads stay in front of this one's MFMAs, and no further ahead)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(smem + (16 * rt + lr) * PJ::ROWP + (32 * kk + 8 * lh) * 2);
#pragma unroll
            for (int c = 0; c < TPW; ++c) acc[c][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk & 1][c], b, acc[c][rt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int r = 16 * rt + lr;
        if (r >= nrows) continue;
        uint16_t *dst = y + (base + r) * N_OUT + 16 * tile0 + 4 * lh;
#pragma unroll
        for (int c = 0; c < TPW; ++c)
            if (tile0 + c < PJ::TILES)
                *reinterpret_cast<uint2 *>(dst + 16 * c) = make_uint2(pack2_bf16(acc[c][rt][0], acc[c][rt][1]), pack2_bf16(acc[c][rt][2], acc[c][rt][3]));
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int mapf_tall_tn_plan(int64_t K, int m, int n, int *slabs_out, int *parts_out, int64_t *ws_elems_out) {
    if (K < 0 || m < 1 || n < 1) return MAPF_ERR_INVALID_ARG;
    const TallPlan p = tall_plan(K, m, n);
    if (slabs_out) *slabs_out = p.slabs;
    if (parts_out) *parts_out = p.parts;
    if (ws_elems_out) *ws_elems_out = p.parts > 1 ? (int64_t)p.slabs * p.parts * p.sm * p.sn : 0;
    return MAPF_OK;
}

int mapf_tall_tn(const uint16_t *a_dev, int64_t lda, const uint16_t *b_dev, int64_t ldb, int64_t K, int m, int n, int f16, float *out_dev,
                 const uint32_t *scale_dev, int accumulate, float *ws_dev, int64_t ws_elems, void *stream) {
    if (K < 0 || m < 1 || n < 1 || !out_dev || (K > 0 && (!a_dev || !b_dev)) || lda < m || ldb < n) return MAPF_ERR_INVALID_ARG;
    if ((lda & 7) || (ldb & 7) || (m & 7) || (n & 7) || !aligned16(a_dev) || !aligned16(b_dev) || !aligned16(out_dev) || !aligned16(ws_dev))
        return MAPF_ERR_INVALID_ARG;
    const TallPlan p = tall_plan(K, m, n);
    if (p.parts > 1 && (!ws_dev || ws_elems < (int64_t)p.slabs * p.parts * p.sm * p.sn))
        return MAPF_ERR_INVALID_ARG;
    const dim3 grid((n + p.sn - 1) / p.sn, (m + p.sm - 1) / p.sm, p.parts);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define TALL_LAUNCH(F, WM, WN, MT, NT)                                                                                                  \
    hipLaunchKernelGGL((tall_tn_kernel<F, WM, WN, MT, NT>), grid, dim3(T_THR), 0, st, a_dev, (long long)lda, b_dev, (long long)ldb, (long long)K, m, \
                       n, ws_dev, out_dev, scale_dev, accumulate)
    if (p.cfg == 0) {
        if (f16) TALL_LAUNCH(true, 2, 2, 4, 4); else TALL_LAUNCH(false, 2, 2, 4, 4);
    } else if (p.cfg == 1) {
        if (f16) TALL_LAUNCH(true, 1, 4, 1, 2); else TALL_LAUNCH(false, 1, 4, 1, 2);
    } else if (p.cfg == 2) {
        if (f16) TALL_LAUNCH(true, 1, 4, 4, 2); else TALL_LAUNCH(false, 1, 4, 4, 2);
    } else {
        if (f16) TALL_LAUNCH(true, 4, 1, 2, 4); else TALL_LAUNCH(false, 4, 1, 2, 4);
    }
#undef TALL_LAUNCH
    if (p.parts > 1)
        hipLaunchKernelGGL(tall_sum_kernel, dim3(p.sm * p.sn / 1024, p.slabs), dim3(256), 0, st, ws_dev, p.parts, p.sm, p.sn, (n + p.sn - 1) / p.sn, m, n,
                           out_dev, scale_dev, accumulate);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_sum_parts(const float *const *parts_dev, float *const *out_dev, int groups, int P, int64_t n, const uint32_t *scale_dev, void *stream) {
    if (groups < 1 || groups > 8 || P < 1 || n < 4 || (n & 3) || !parts_dev || !out_dev) return MAPF_ERR_INVALID_ARG;
    SumGroups gr{};
    for (int g = 0; g < groups; ++g) {
        if (!parts_dev[g] || !out_dev[g] || !aligned16(parts_dev[g]) || !aligned16(out_dev[g])) return MAPF_ERR_INVALID_ARG;
        gr.parts[g] = parts_dev[g];
        gr.out[g] = out_dev[g];
    }
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((n / 4 + 255) / 256), groups), dim3(256), 0, static_cast<hipStream_t>(stream), gr, P, (long long)n,
                       scale_dev);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_encoder_small_grads(const float *gb_part_dev, int64_t nblk, float *bias7_out_dev, const float *gb7_part_dev, int64_t rows7,
                             float *bias1_out_dev, const float *ws0_dev, int parts0, float *w0_out_dev, float *scratch_dev, void *stream) {
    if (!gb_part_dev || !bias7_out_dev || !gb7_part_dev || !bias1_out_dev || !ws0_dev || !w0_out_dev || !scratch_dev || nblk < 0 || rows7 < 0 || parts0 < 1)
        return MAPF_ERR_INVALID_ARG;
    if (!aligned16(gb_part_dev) || !aligned16(gb7_part_dev) || !aligned16(scratch_dev)) return MAPF_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sum_bias1_kernel, dim3(8, BS), dim3(256), 0, st, gb_part_dev, (long long)nblk, gb7_part_dev, (long long)rows7, scratch_dev);
    hipLaunchKernelGGL(sum_bias2_kernel, dim3(8), dim3(128), 0, st, scratch_dev, gb7_part_dev, (long long)rows7, bias7_out_dev, bias1_out_dev);
    hipLaunchKernelGGL(sum_conv0_kernel, dim3((128 * 54 + 255) / 256), dim3(256), 0, st, ws0_dev, parts0, w0_out_dev);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_latent_grad_pack(const float *w_ih_dev, uint16_t *packed_dev, void *stream) {
    if (!w_ih_dev || !packed_dev || !aligned16(packed_dev)) return MAPF_ERR_INVALID_ARG;
    using PJ = Proj<768, 784>;
    static_assert(PJ::TILES * PJ::KS * 512 == MAPF_LATGRAD_PACKED_ELEMS, "header constant out of date");
    hipLaunchKernelGGL((proj_pack_kernel<768, 784, true>), dim3((PJ::TILES * PJ::KS * 512 + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       w_ih_dev, packed_dev);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_latent_grad_rows(const uint16_t *d_gi_dev, int64_t num_rows, const uint16_t *packed_dev, uint16_t *g_lat_dev, void *stream) {
    if (num_rows < 0 || !d_gi_dev || !packed_dev || !g_lat_dev || !aligned16(d_gi_dev) || !aligned16(packed_dev) || (reinterpret_cast<uintptr_t>(g_lat_dev) & 7))
        return MAPF_ERR_INVALID_ARG;
    if (num_rows == 0) return MAPF_OK;
    const long long blocks = (num_rows + 63) / 64;
    if (blocks > 0x7FFFFFFFLL) return MAPF_ERR_INVALID_ARG;
    // 49 output tiles: 7 per wave over 7 waves (<= 32 k rows: two workgroups per row block with 4 / 3 tiles per wave, as csrc/mapf_inproj.hip)
    if (num_rows <= 32768)
        hipLaunchKernelGGL((proj_rows_kernel<768, 784, 4>), dim3((unsigned)blocks, 2), dim3(512), 0, static_cast<hipStream_t>(stream), d_gi_dev,
                           (long long)num_rows, packed_dev, g_lat_dev);
    else
        hipLaunchKernelGGL((proj_rows_kernel<768, 784, 7>), dim3((unsigned)blocks, 1), dim3(512), 0, static_cast<hipStream_t>(stream), d_gi_dev,
                           (long long)num_rows, packed_dev, g_lat_dev);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

}  // extern "C"
