// mapf_recur_bwd.hip -- backward through time of the recurrence that csrc/mapf_recur.hip runs forward
// (reference: autograd through model.py:242-249 `bootstrap`, i.e. GRUCell + 2 x CommBlock round per step, driven by
// `loss.backward()` at worker.py:316).  One workgroup per environment walks the steps in reverse with DH, the gradient
// w.r.t. the hidden state, resident in LDS (see include/mapf_dqn.h: mapf_recurrent_backward).
//
// Per step, in reverse: for round 1, 0 of the CommBlock
//   (1) update-cell backward, elementwise on the saved gates:  d = upd ? DH : 0,
//         dn = d (1-z)(1-n^2),  dz = d (h - n) z (1-z),  dr = dn hn r (1-r);   d_gi = (dr,dz,dn), d_gh = (dr,dz,dn r)
//       are written to global memory (they are outputs: the caller's weight-gradient GEMMs need them) and, as rows
//       [dr | dz | dn | dn r] of the LDS image G, kept for (2); DH <- upd ? d z : DH;
//   (2) DH += U_hh^T d_gh,  d_info = U_ih^T d_gi   (A = transposed weights packed in fragment order, B = rows of G: every
//       wave needs every row, so reading them back from global would cost 16 x the bytes of the weights themselves);
//   (3) d_ctx = W_O^T d_info;
//   (4) attention backward, one head at a time on LDS images of q, k, v, P (saved by the forward): dP = d_ctx v^T,
//       dS = P (dP - rowsum(dP P)) / 8, dv = P^T d_ctx, dq = dS k, dk = dS^T q.  Every product is an MFMA with both
//       operands row-major in LDS: an operand whose reduction index is the ROW of its image is read with
//       ds_read_b64_tr_b16 (blocks at rows 8 lh and 8 lh + 4, so the fragment's k order is that of a plain 16-byte read);
//   (5) DH += W_qkv^T d_qkv;
// then the recurrent cell's backward (d_gi of it is the gradient w.r.t. the GRU input projection, an output) and
// DH += W_hh^T d_gh; agent 0's external gradient of the previous step is added and the loop continues.
// Weight gradients are NOT formed here: they are tall GEMMs over (saved input, d_*) rows that the caller runs once per
// update.  Bias gradients are column sums of the d_* rows: accumulated in LDS on the way (fp32) and written as one
// partial vector per environment, which spares the caller five more passes over 0.2-0.4 GB each.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "mapf_dqn.h"
#include "mapf_env.h"
#include "mapf_recur_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#ifndef MAPF_RBWD_ABLATE  // diagnostic builds only (tools/micro/recur_bwd_ablate.py): 1 update-cell elementwise, 2 its GEMMs, 4 W_O + attention,
#define MAPF_RBWD_ABLATE 0  // 8 W_qkv GEMM, 16 recurrent cell; results are wrong, only the time matters
#endif
#ifndef MAPF_RECUR_NT  // as in csrc/mapf_recur.hip: 3 agent tiles here, 1 in csrc/mapf_recur_bwd_nt1.hip (which includes this file)
#define MAPF_RECUR_NT 3
#endif
constexpr int NT = MAPF_RECUR_NT, NA = 16 * NT, D = 256, HD = 64, NTHR = 512;
#ifdef MAPF_RECUR_TRACE  // diagnostic builds only (tools/micro/recur_bwd_trace.py): cycle stamps of wave 0 of one workgroup, as in csrc/mapf_recur.hip
#ifndef MAPF_RECUR_TRACE_WG
#define MAPF_RECUR_TRACE_WG 0
#endif
__device__ unsigned long long g_btrace[128];
__device__ int g_btrace_n;
__shared__ unsigned long long btrace_lds[128];
__shared__ int btrace_lds_n;
#define TRACE_POINT(id)                                                                                                  \
    do {                                                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x == MAPF_RECUR_TRACE_WG) {                                                     \
            const int k_ = btrace_lds_n;                                                                                 \
            if (k_ < 126) {                                                                                              \
                btrace_lds[k_] = ((unsigned long long)(id) << 56) | (__builtin_readcyclecounter() & 0xFFFFFFFFFFFFFFull); \
                btrace_lds_n = k_ + 1;                                                                                   \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)
#define TRACE_BEGIN() do { if (threadIdx.x == 0) btrace_lds_n = 0; } while (0)
#define TRACE_END()                                                                    \
    do {                                                                               \
        if (threadIdx.x == 0 && blockIdx.x == MAPF_RECUR_TRACE_WG) {                   \
            for (int k_ = 0; k_ < btrace_lds_n; ++k_) g_btrace[k_] = btrace_lds[k_];   \
            g_btrace_n = btrace_lds_n;                                                 \
        }                                                                              \
    } while (0)
#else
#define TRACE_POINT(id) do { } while (0)
#define TRACE_BEGIN() do { } while (0)
#define TRACE_END() do { } while (0)
#endif
constexpr int H_ROW = D * 2 + 32;     // 544
constexpr int INFO_ROW = 64 * 2 + 32;  // 160
constexpr int CTX_ROW = 128 * 2 + 32;  // 288
constexpr int QKV_ROW = 384 * 2 + 32;  // 800
constexpr int IMG_ROW = 64 * 2 + 32;   // per-head q / k / v / P / dS images: 64 rows x 64 columns
constexpr int SF_ROW = 52;             // fp32 dP rows
constexpr int G_ROW = 1024 * 2 + 32;   // [dr | dz | dn | dn r] rows of one GRU-cell backward
constexpr int OFF_DH = 0;
constexpr int OFF_DINFO = OFF_DH + NA * H_ROW;
constexpr int OFF_DCTX = OFF_DINFO + NA * INFO_ROW;      // 64 rows: rows >= 48 stay zero (K padding of the transposed reads)
constexpr int OFF_G = OFF_DCTX + 64 * CTX_ROW;           // G and the attention images / d_qkv / dP are never live together
constexpr int OFF_DQKV = OFF_G;
constexpr int OFF_QI = OFF_DQKV + NA * QKV_ROW;
constexpr int OFF_KI = OFF_QI + 64 * IMG_ROW;
constexpr int OFF_VI = OFF_KI + 64 * IMG_ROW;
constexpr int OFF_PI = OFF_VI + 64 * IMG_ROW;
constexpr int OFF_DSI = OFF_PI + 64 * IMG_ROW;
constexpr int OFF_SF = OFF_DSI + 64 * IMG_ROW;
constexpr int UNION_BYTES = (OFF_SF + NA * SF_ROW * 4 - OFF_G) > NA * G_ROW ? (OFF_SF + NA * SF_ROW * 4 - OFF_G) : NA * G_ROW;
constexpr int OFF_UPD = OFF_G + UNION_BYTES;
constexpr int OFF_RIDX = OFF_UPD + 64 * 4;  // global row of every agent at this step (-1: none), as in the forward kernel
constexpr int OFF_BSUM = OFF_RIDX + 64 * 4;  // fp32 column sums over (steps, agents): [update cell | recurrent cell][dr|dz|dn|dn r][256], d_qkv[384]
constexpr int NBSUM = MAPF_RECUR_BSUM_ELEMS;
static_assert(NBSUM == 2 * 1024 + 384, "header constant out of date");
constexpr int LDS_BYTES = OFF_BSUM + NBSUM * 4;
static_assert(LDS_BYTES <= 160 * 1024 && OFF_SF % 16 == 0 && OFF_UPD % 16 == 0, "LDS budget / alignment");

// transposed-weight buffer (bf16 elements, fragment order [tile][k-step][lane][8], gates outermost for the GRU matrices)
constexpr int WT_UIH = 0;                         // [3][4][8]   U_ih^T: 64 outputs, K = 768 (3 gates x 256)
constexpr int WT_UHH = WT_UIH + 3 * 4 * 8 * 512;  // [3][16][8]  U_hh^T: 256 outputs
constexpr int WT_WHH = WT_UHH + 3 * 16 * 8 * 512; // [3][16][8]  W_hh^T
constexpr int WT_WO = WT_WHH + 3 * 16 * 8 * 512;  // [8][2]      W_O^T: 128 outputs, K = 64
constexpr int WT_QKV = WT_WO + 8 * 2 * 512;       // [16][12]    W_qkv^T: 256 outputs, K = 384
constexpr int WT_TOTAL = WT_QKV + 16 * 12 * 512;
static_assert(WT_TOTAL == MAPF_RECUR_WEIGHT_ELEMS, "header constant out of date");

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ void unpack4(const uint2 v, float (&f)[4]) {
    f[0] = bf16_lo(v.x);
    f[1] = bf16_hi(v.x);
    f[2] = bf16_lo(v.y);
    f[3] = bf16_hi(v.y);
}
__device__ __forceinline__ uint2 pack4(const float (&f)[4]) { return make_uint2(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3])); }

// Sum over the 16 lanes of a DPP row (every lane ends up with the total): row rotations by 8 and 4, then the two quad
// permutations.  VALU-only: __shfl_xor compiles to ds_bpermute_b32, an LDS-pipe instruction with LDS latency.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row_sum16(float v) {
    v = dpp_add<0x128>(v);  // row_ror:8
    v = dpp_add<0x124>(v);  // row_ror:4
    v = dpp_add<0x4E>(v);   // quad_perm:[2,3,0,1]
    return dpp_add<0xB1>(v);  // quad_perm:[1,0,3,2]
}
// fragment (16 "columns" col0.. of the image, k = rows row0 .. row0+32) of an operand whose reduction index is the image ROW
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *img, int row_bytes, int row0, int col0, int lane) {
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const int li = lane & 15, lh = lane >> 4, q4 = li >> 2, p4 = li & 3;
    const unsigned char *p0 = img + (row0 + 8 * lh + q4) * row_bytes + (col0 + 4 * p4) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0 + 4 * row_bytes));
    union {
        s16x4 h[2];
        bf16x8 v;
    } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
}
// fragment of an operand whose reduction index runs along the image row (row = 16 tile + lr, k = k0 + 8 lh .. +8)
__device__ __forceinline__ bf16x8 row_frag(const unsigned char *img, int row_bytes, int row, int k0, int lane) {
    return *reinterpret_cast<const bf16x8 *>(img + row * row_bytes + (k0 + 8 * (lane >> 4)) * 2);
}

// A pointer that went through an opaque `asm volatile("" : "+s"(p))` comes back GENERIC, and hipcc then emits FLAT loads -- which
// also count on the LDS counter, so every LDS wait waits for the weight stream too.  Back into the global address space:
#define GLOBAL_FRAG(p) ((const __attribute__((address_space(1))) bf16x8 *)(p))
// acc[n] += (packed tile `wp`, KS k-steps) * B^T, B rows in an LDS image: row(agent) = X + agent * xrow, columns 0 .. 32 KS
template <int KS>
__device__ __forceinline__ void gemm_lB(f32x4 (&acc)[NT], const unsigned char *__restrict__ wt, int tile_kstep, const unsigned char *X, int xrow,
                                        int lane) {
    const int lr = lane & 15, lh = lane >> 4;
    // scalar base + 32-bit lane offset: the load takes its address as (SGPR pair, VGPR offset) instead of a 64-bit VGPR
    // pair per tile, which the compiler would hoist out of the time loop and spill
    const unsigned char *wp = wt + (size_t)tile_kstep * 1024;
    const uint32_t voff = (uint32_t)lane * 16u;
    bf16x8 a[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const unsigned char *pk = wp + kk * 1024;
        asm volatile("" : "+s"(pk));
        a[kk] = *GLOBAL_FRAG(pk + voff);
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(X + (16 * n + lr) * xrow + (32 * kk + 8 * lh) * 2);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk], b, acc[n], 0, 0, 0);
        }
}

// two output tiles at once: both tiles' A fragments (2 KS KiB per wave) are requested before the first MFMA -- the kernel is
// bound by the L2 -> CU weight stream, i.e. by the bytes it keeps in flight -- and every B fragment is read from LDS once
template <int KS>
__device__ __forceinline__ void gemm2_lB(f32x4 (&acc0)[NT], f32x4 (&acc1)[NT], const unsigned char *__restrict__ wt, int tk0, int tk1,
                                         const unsigned char *X, int xrow, int lane) {
    const int lr = lane & 15, lh = lane >> 4;
    const unsigned char *w0 = wt + (size_t)tk0 * 1024, *w1 = wt + (size_t)tk1 * 1024;
    const uint32_t voff = (uint32_t)lane * 16u;
    bf16x8 a0[KS], a1[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        // (laundered scalar bases: otherwise hipcc folds the lane offset into one 64-bit VGPR address per tile, hoists all of
        // them out of the time loop and spills them)
        const unsigned char *p0 = w0 + kk * 1024, *p1 = w1 + kk * 1024;
        asm volatile("" : "+s"(p0), "+s"(p1));
        a0[kk] = *GLOBAL_FRAG(p0 + voff);
        a1[kk] = *GLOBAL_FRAG(p1 + voff);
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(X + (16 * n + lr) * xrow + (32 * kk + 8 * lh) * 2);
            acc0[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[kk], b, acc0[n], 0, 0, 0);
            acc1[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[kk], b, acc1[n], 0, 0, 0);
        }
}

struct BwdArgs {
    // saved by the forward (see RecurSave in csrc/mapf_recur.hip)
    const uint16_t *hin0, *g1, *hr, *qkv, *ctx_unused, *info_unused, *g2, *P;
    const uint8_t *comm;   // [T][E][N][N]
    const uint16_t *dA0;   // [T][E][256] gradient w.r.t. agent 0's state after every step
    const uint16_t *WT;    // transposed weights, fragment order
    // outputs, rows as in the forward's saved tensors
    uint16_t *d_gi1, *d_gh1;  // [R][768]
    uint16_t *d_gi2, *d_gh2;  // [2][R][768]
    uint16_t *d_info;         // [2][R][64]
    uint16_t *d_qkv;          // [2][R][384]
    float *bsum;              // [E][NBSUM] per-environment column sums (bias gradients)
    int T, E, N;
    const int32_t *rowidx;    // [T][E][N] compact rows (see csrc/mapf_recur.hip) or nullptr: dense
    long long nrows;
    int a0s;                  // > 0: the environment is a tile of several windows of a0s agent rows each; dA0 [T][E][N / a0s][256] -> rows 0, a0s, ...
};

__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
    f[0] = bf16_lo(v.x);
    f[1] = bf16_hi(v.x);
    f[2] = bf16_lo(v.y);
    f[3] = bf16_hi(v.y);
    f[4] = bf16_lo(v.z);
    f[5] = bf16_hi(v.z);
    f[6] = bf16_lo(v.w);
    f[7] = bf16_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return make_uint4(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]), pack2_bf16(f[4], f[5]), pack2_bf16(f[6], f[7]));
}

// GRU cell backward, elementwise.  Nothing here needs the MFMA lane layout (DH and G are LDS images), so the work is cut
// into (agent, 8 consecutive channels) tasks: every global access is a 16-byte piece of a fully used 512-byte row segment.
// (In the accumulator layout -- 4 channels of 16 different agents per load -- this phase was 45 % of the kernel.)
// Reads DH (LDS), the saved gates and input state (global); writes the d_gi / d_gh rows (global), rows [dr|dz|dn|dn r] of
// G (LDS; zero for agents >= N: G shares its LDS with the attention images) and DH <- (upd ? d z : DH).
__device__ __forceinline__ void gru_bwd_elementwise(unsigned char *DH, unsigned char *G, const uint16_t *__restrict__ gates,
                                                    const uint16_t *__restrict__ hin, const int *upd, uint16_t *__restrict__ dgi,
                                                    uint16_t *__restrict__ dgh, const int *ridx, int tid) {
    constexpr int NTASK = NA * 32 / NTHR;  // 3
    // all 15 global loads of the thread's three tasks are issued before any of them is used (the phase sits between two
    // barriers with nothing to overlap: one exposed HBM latency instead of three); agents without a row read agent 0's rows and
    // are zeroed afterwards, so that the loads carry no control dependence
    uint4 vr[NTASK], vz[NTASK], vn[NTASK], vh[NTASK], vx[NTASK];
#pragma unroll
    for (int it = 0; it < NTASK; ++it) {
        const int task = tid + it * NTHR, agent = task >> 5, c0 = 8 * (task & 31);
        const int src = ridx[agent] >= 0 ? ridx[agent] : 0;  // (any valid row: the values are discarded)
        const uint16_t *g = gates + (long long)src * 1024 + c0;
        vr[it] = *reinterpret_cast<const uint4 *>(g);
        vz[it] = *reinterpret_cast<const uint4 *>(g + 256);
        vn[it] = *reinterpret_cast<const uint4 *>(g + 512);
        vh[it] = *reinterpret_cast<const uint4 *>(g + 768);
        vx[it] = *reinterpret_cast<const uint4 *>(hin + (long long)src * D + c0);
    }
#pragma unroll
    for (int it = 0; it < NTASK; ++it) {
        const int task = tid + it * NTHR, agent = task >> 5, c0 = 8 * (task & 31);
        uint4 *grow = reinterpret_cast<uint4 *>(G + agent * G_ROW + c0 * 2);  // gate g at + 32 g (512 bytes apart)
        const int row = ridx[agent];
        if (row < 0) {
            grow[0] = grow[32] = grow[64] = grow[96] = make_uint4(0, 0, 0, 0);
            continue;
        }
        float r[8], z[8], nn[8], hn[8], h[8], d[8];
        unpack8(vr[it], r);
        unpack8(vz[it], z);
        unpack8(vn[it], nn);
        unpack8(vh[it], hn);
        unpack8(vx[it], h);
        uint4 *dcell = reinterpret_cast<uint4 *>(DH + agent * H_ROW + c0 * 2);
        unpack8(*dcell, d);
        const bool on = upd == nullptr || upd[agent] != 0;
        float dr[8], dz[8], dn[8], dnr[8], dpass[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float dd = on ? d[k] : 0.f;
            dn[k] = dd * (1.f - z[k]) * (1.f - nn[k] * nn[k]);
            dz[k] = dd * (h[k] - nn[k]) * z[k] * (1.f - z[k]);
            dr[k] = dn[k] * hn[k] * r[k] * (1.f - r[k]);
            dnr[k] = dn[k] * r[k];
            dpass[k] = on ? dd * z[k] : d[k];
        }
        uint16_t *gi = dgi + (long long)row * 768 + c0, *gh = dgh + (long long)row * 768 + c0;
        const uint4 pr = pack8(dr), pz = pack8(dz), pn = pack8(dn), pnr = pack8(dnr);
        *reinterpret_cast<uint4 *>(gi) = pr;
        *reinterpret_cast<uint4 *>(gi + 256) = pz;
        *reinterpret_cast<uint4 *>(gi + 512) = pn;
        *reinterpret_cast<uint4 *>(gh) = pr;
        *reinterpret_cast<uint4 *>(gh + 256) = pz;
        *reinterpret_cast<uint4 *>(gh + 512) = pnr;
        grow[0] = pr;
        grow[32] = pz;
        grow[64] = pn;
        grow[96] = pnr;
        *dcell = pack8(dpass);
    }
}

// bias gradients: column sums of G over the agents (rows >= N are zero), two columns per thread, fixed order -> repeatable;
// runs in the GEMM phase that only reads G
__device__ __forceinline__ void bias_colsum(const unsigned char *G, float *bsum, int tid) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
    for (int a = 0; a < NA; ++a) {
        const uint32_t wv = *reinterpret_cast<const uint32_t *>(G + a * G_ROW + tid * 4);
        s0 += bf16_lo(wv);
        s1 += bf16_hi(wv);
    }
    bsum[2 * tid] += s0;
    bsum[2 * tid + 1] += s1;
}

// DH[agent][16 tile + 4 lh ..] += acc  (this lane's own cells)
__device__ __forceinline__ void add_to_dh(unsigned char *DH, const f32x4 (&acc)[NT], int tile, int lr, int lh) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        uint2 *cell = reinterpret_cast<uint2 *>(DH + (16 * n + lr) * H_ROW + (16 * tile + 4 * lh) * 2);
        float d[4];
        unpack4(*cell, d);
        const float o[4] = {d[0] + acc[n][0], d[1] + acc[n][1], d[2] + acc[n][2], d[3] + acc[n][3]};
        *cell = pack4(o);
    }
}

__global__ void __launch_bounds__(NTHR, 1) recurrent_bwd_kernel(BwdArgs A) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lh = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: weight-tile addresses become scalar base + lane offset
    const int e = blockIdx.x, T = A.T, E = A.E, N = A.N;
    const long long RTOT = A.rowidx ? A.nrows : (long long)T * E * N;
    unsigned char *DH = smem + OFF_DH;
    int *upd = reinterpret_cast<int *>(smem + OFF_UPD);
    int *ridx = reinterpret_cast<int *>(smem + OFF_RIDX);
    float *SF = reinterpret_cast<float *>(smem + OFF_SF);
    const unsigned char *WTB = reinterpret_cast<const unsigned char *>(A.WT);

    for (int i = tid; i < LDS_BYTES / 16; i += NTHR) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    TRACE_BEGIN();
    for (int t = T - 1; t >= 0; --t) {
        const long long row0 = ((long long)t * E + e) * N;
        // ---- external gradient of agent 0's state after step t; partner counts of this step's mask ----
        if (tid < 64) {
            const int K = A.a0s > 0 ? (N + A.a0s - 1) / A.a0s : 1;
            for (int k = 0; k < K; ++k) {
                const uint2 g = *reinterpret_cast<const uint2 *>(A.dA0 + (((long long)t * E + e) * K + k) * D + tid * 4);
                uint2 *cell = reinterpret_cast<uint2 *>(DH + (k * A.a0s) * H_ROW + tid * 8);
                float a[4], b[4];
                unpack4(*cell, a);
                unpack4(g, b);
                const float o[4] = {a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
                *cell = pack4(o);
            }
        } else if (tid < 128) {
            upd[tid - 64] = 0;
        } else if (tid < 192) {
            const int a = tid - 128;
            ridx[a] = a < N ? (A.rowidx ? A.rowidx[row0 + a] : (int)(row0 + a)) : -1;
        }
        TRACE_POINT(1);
        __syncthreads();
        TRACE_POINT(101);
        {
            const uint8_t *cm = A.comm + ((long long)t * E + e) * N * N;
            for (int idx = tid; idx < N * N; idx += NTHR)
                if (cm[idx] != 0) atomicAdd(&upd[idx / N], 1);
        }
        TRACE_POINT(2);
        __syncthreads();
        TRACE_POINT(102);
        if (tid < 64) upd[tid] = upd[tid] > 1 ? 1 : 0;  // model.py:103
        TRACE_POINT(3);
        __syncthreads();
        TRACE_POINT(103);

        for (int q = 1; q >= 0; --q) {
            const long long rq = (long long)q * RTOT;  // first row of round q's tensors
            uint16_t *dgi2 = A.d_gi2 + rq * 768, *dgh2 = A.d_gh2 + rq * 768;
            // (1) update-cell backward
            if (!(MAPF_RBWD_ABLATE & 1)) gru_bwd_elementwise(DH, smem + OFF_G, A.g2 + rq * 1024, A.hr + rq * D, upd, dgi2, dgh2, ridx, tid);
            TRACE_POINT(4);
            __syncthreads();
            TRACE_POINT(104);
            // (2) DH += U_hh^T d_gh (2 output tiles per wave); d_info = U_ih^T d_gi (waves 0-3, one tile each)
            if (!(MAPF_RBWD_ABLATE & 1)) bias_colsum(smem + OFF_G, reinterpret_cast<float *>(smem + OFF_BSUM), tid);
            if (!(MAPF_RBWD_ABLATE & 2)) {
                f32x4 acc0[NT], acc1[NT], acci[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) acc0[n] = acc1[n] = acci[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const unsigned char *gh = smem + OFF_G + (g == 2 ? 768 : 256 * g) * 2, *gi = smem + OFF_G + 256 * g * 2;
                    gemm2_lB<8>(acc0, acc1, WTB, WT_UHH / 512 + ((g * 16 + w) * 8), WT_UHH / 512 + ((g * 16 + w + 8) * 8), gh, G_ROW, lane);
                    if (w < 4) gemm_lB<8>(acci, WTB, WT_UIH / 512 + ((g * 4 + w) * 8), gi, G_ROW, lane);
                }
                add_to_dh(DH, acc0, w, lr, lh);
                add_to_dh(DH, acc1, w + 8, lr, lh);
                if (w < 4) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int agent = 16 * n + lr;
                        const float o[4] = {acci[n][0], acci[n][1], acci[n][2], acci[n][3]};
                        const uint2 v = pack4(o);
                        *reinterpret_cast<uint2 *>(smem + OFF_DINFO + agent * INFO_ROW + (16 * w + 4 * lh) * 2) = v;
                        if (ridx[agent] >= 0) *reinterpret_cast<uint2 *>(A.d_info + (rq + ridx[agent]) * 64 + 16 * w + 4 * lh) = v;
                    }
                }
            }
            TRACE_POINT(5);
            __syncthreads();
            TRACE_POINT(105);
            // (3) d_ctx = W_O^T d_info: 8 output tiles, K = 64
            if (!(MAPF_RBWD_ABLATE & 4)) {
                f32x4 acc[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm_lB<2>(acc, WTB, WT_WO / 512 + (w * 2), smem + OFF_DINFO, INFO_ROW, lane);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float o[4] = {acc[n][0], acc[n][1], acc[n][2], acc[n][3]};
                    *reinterpret_cast<uint2 *>(smem + OFF_DCTX + (16 * n + lr) * CTX_ROW + (16 * w + 4 * lh) * 2) = pack4(o);
                }
            }
            TRACE_POINT(6);
            __syncthreads();
            TRACE_POINT(106);
            // (4) attention backward, one head at a time
            for (int hd = 0; hd < 2 && !(MAPF_RBWD_ABLATE & 4); ++hd) {
                // images: q, k, v rows [agent][64] of this head (zero rows for agents >= N); P rows [agent i][64 slots j] (zero rows
                // 48..63).  The four 16-byte loads of a thread are issued together, then stored: one exposed latency per head.
                {
                    const uint16_t *ps = A.P + ((((long long)q * T + t) * E + e) * 2 + hd) * (NA * 64);
                    const int a = (tid >> 3) & 63, ch = tid & 7;  // every thread: row a, chunk ch of q, of k, of v and of P
                    uint4 vq = make_uint4(0, 0, 0, 0), vk = vq, vv = vq, vp = vq;
                    if (ridx[a] >= 0) {
                        const uint16_t *row = A.qkv + (rq + ridx[a]) * 384 + hd * HD + ch * 8;
                        vq = *reinterpret_cast<const uint4 *>(row);
                        vk = *reinterpret_cast<const uint4 *>(row + 128);
                        vv = *reinterpret_cast<const uint4 *>(row + 256);
                    }
                    if (a < NA) vp = *reinterpret_cast<const uint4 *>(ps + a * 64 + ch * 8);
                    unsigned char *dst = smem + a * IMG_ROW + ch * 16;
                    *reinterpret_cast<uint4 *>(dst + OFF_QI) = vq;
                    *reinterpret_cast<uint4 *>(dst + OFF_KI) = vk;
                    *reinterpret_cast<uint4 *>(dst + OFF_VI) = vv;
                    *reinterpret_cast<uint4 *>(dst + OFF_PI) = vp;
                }
                TRACE_POINT(7);
                __syncthreads();
                TRACE_POINT(107);
                // dP[i][j] = sum_d d_ctx[i][d] v[j][d]
                for (int job = w; job < NT * NT; job += NTHR / 64) {
                    const int ti = job / NT, tj = job % NT;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(smem + OFF_DCTX, CTX_ROW, 16 * ti + lr, hd * HD + 32 * kk, lane),
                                                                      row_frag(smem + OFF_VI, IMG_ROW, 16 * tj + lr, 32 * kk, lane), acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) SF[(16 * ti + 4 * lh + r) * SF_ROW + 16 * tj + lr] = acc[r];
                }
                TRACE_POINT(8);
                __syncthreads();
                TRACE_POINT(108);
                // softmax backward per row i, 4 lanes x 12 columns each: dS = P (dP - sum_j dP P) / 8; slots 48..63 and rows 48..63 of
                // the dS image are zero (K padding)
                if (tid < 4 * NA) {
                    const int i = tid >> 2, part = tid & 3;
                    const float4 *dp = reinterpret_cast<const float4 *>(SF + i * SF_ROW + 12 * part);
                    const uint2 *pp = reinterpret_cast<const uint2 *>(smem + OFF_PI + i * IMG_ROW + 24 * part);
                    float P[12], dP[12];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float4 x = dp[c];
                        dP[4 * c] = x.x;
                        dP[4 * c + 1] = x.y;
                        dP[4 * c + 2] = x.z;
                        dP[4 * c + 3] = x.w;
                        float pv[4];
                        unpack4(pp[c], pv);
                        P[4 * c] = pv[0];
                        P[4 * c + 1] = pv[1];
                        P[4 * c + 2] = pv[2];
                        P[4 * c + 3] = pv[3];
                    }
                    float dot = 0.f;
#pragma unroll
                    for (int j = 0; j < 12; ++j) dot += dP[j] * P[j];
                    dot = dpp_add<0x4E>(dot);  // the 4 lanes of a row are one quad
                    dot = dpp_add<0xB1>(dot);
                    uint2 *ds = reinterpret_cast<uint2 *>(smem + OFF_DSI + i * IMG_ROW + 24 * part);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float o[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = P[4 * c + k] * (dP[4 * c + k] - dot) * 0.125f;
                        ds[c] = pack4(o);
                    }
                    if (part == 0) {
                        uint4 *pad = reinterpret_cast<uint4 *>(smem + OFF_DSI + i * IMG_ROW + 96);
                        pad[0] = pad[1] = make_uint4(0, 0, 0, 0);
                    }
                } else if (tid < 4 * NA + (64 - NA) * 8) {  // rows NA .. 63 of the image stay zero
                    const int k = tid - 4 * NA;
                    *reinterpret_cast<uint4 *>(smem + OFF_DSI + (NA + (k >> 3)) * IMG_ROW + (k & 7) * 16) = make_uint4(0, 0, 0, 0);
                }
                TRACE_POINT(9);
                __syncthreads();
                TRACE_POINT(109);
                // dv, dq, dk: 3 products x 4 d-tiles x 3 agent tiles, each K = 64 (image rows / slots 48..63 are zero)
                for (int job = w; job < 3 * 4 * NT; job += NTHR / 64) {
                    const int prod = job / (4 * NT), td = (job / NT) % 4, ta = job % NT;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        bf16x8 a, b;
                        if (prod == 0) {  // dv^T[d][j] = sum_i d_ctx[i][d] P[i][j]
                            a = tr_frag(smem + OFF_DCTX, CTX_ROW, 32 * kk, hd * HD + 16 * td, lane);
                            b = tr_frag(smem + OFF_PI, IMG_ROW, 32 * kk, 16 * ta, lane);
                        } else if (prod == 1) {  // dq^T[d][i] = sum_j k[j][d] dS[i][j]
                            a = tr_frag(smem + OFF_KI, IMG_ROW, 32 * kk, 16 * td, lane);
                            b = row_frag(smem + OFF_DSI, IMG_ROW, 16 * ta + lr, 32 * kk, lane);
                        } else {  // dk^T[d][j] = sum_i q[i][d] dS[i][j]
                            a = tr_frag(smem + OFF_QI, IMG_ROW, 32 * kk, 16 * td, lane);
                            b = tr_frag(smem + OFF_DSI, IMG_ROW, 32 * kk, 16 * ta, lane);
                        }
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
                    }
                    const int colbase = (prod == 0 ? 256 : (prod == 1 ? 0 : 128)) + hd * HD + 16 * td + 4 * lh;
                    const float o[4] = {acc[0], acc[1], acc[2], acc[3]};
                    *reinterpret_cast<uint2 *>(smem + OFF_DQKV + (16 * ta + lr) * QKV_ROW + colbase * 2) = pack4(o);
                }
                TRACE_POINT(10);
                __syncthreads();
                TRACE_POINT(110);
            }
            // (5) d_qkv rows -> global; DH += W_qkv^T d_qkv (2 output tiles per wave, K = 384)
            if (tid < 384) {  // column sums of d_qkv (rows >= N are zero)
                float v = 0.f;
                for (int a = 0; a < N; ++a) v += bf16_lo(*reinterpret_cast<const uint16_t *>(smem + OFF_DQKV + a * QKV_ROW + tid * 2));
                reinterpret_cast<float *>(smem + OFF_BSUM)[2048 + tid] += v;
            }
            for (int i = tid; i < N * 48; i += NTHR) {
                const int a = i / 48, ch = i - a * 48;
                if (ridx[a] >= 0)
                    *reinterpret_cast<uint4 *>(A.d_qkv + (rq + ridx[a]) * 384 + ch * 8) = *reinterpret_cast<const uint4 *>(smem + OFF_DQKV + a * QKV_ROW + ch * 16);
            }
            if (!(MAPF_RBWD_ABLATE & 8)) {
                f32x4 acc0[NT], acc1[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) acc0[n] = acc1[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm2_lB<12>(acc0, acc1, WTB, WT_QKV / 512 + (w * 12), WT_QKV / 512 + ((w + 8) * 12), smem + OFF_DQKV, QKV_ROW, lane);
                add_to_dh(DH, acc0, w, lr, lh);
                add_to_dh(DH, acc1, w + 8, lr, lh);
            }
            TRACE_POINT(11);
            __syncthreads();
            TRACE_POINT(111);
        }
        // ---- recurrent cell backward: d_gi1 is the gradient w.r.t. the GRU input projection ----
        if (!(MAPF_RBWD_ABLATE & 16)) gru_bwd_elementwise(DH, smem + OFF_G, A.g1, A.hin0, nullptr, A.d_gi1, A.d_gh1, ridx, tid);
        TRACE_POINT(12);
        __syncthreads();
        TRACE_POINT(112);
        if (!(MAPF_RBWD_ABLATE & 16)) {
            bias_colsum(smem + OFF_G, reinterpret_cast<float *>(smem + OFF_BSUM) + 1024, tid);
            f32x4 acc0[NT], acc1[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) acc0[n] = acc1[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const unsigned char *gh = smem + OFF_G + (g == 2 ? 768 : 256 * g) * 2;
                gemm2_lB<8>(acc0, acc1, WTB, WT_WHH / 512 + ((g * 16 + w) * 8), WT_WHH / 512 + ((g * 16 + w + 8) * 8), gh, G_ROW, lane);
            }
            add_to_dh(DH, acc0, w, lr, lh);
            add_to_dh(DH, acc1, w + 8, lr, lh);
        }
        TRACE_POINT(13);
        __syncthreads();
        TRACE_POINT(113);
    }
    for (int i = tid; i < NBSUM; i += NTHR) A.bsum[(long long)e * NBSUM + i] = reinterpret_cast<const float *>(smem + OFF_BSUM)[i];
    TRACE_END();
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            std::fprintf(stderr, "mapf_recur_bwd: %s failed: %s\n", #expr, hipGetErrorString(_e)); \
            return MAPF_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)

}  // namespace

#if MAPF_RECUR_NT == 3
extern "C" {
#define RECUR_ENTRY(name) name
#else  // the <= 16- / <= 32-agent builds: an internal symbol, reached from the entry point of the 48-agent build (mapf_recur_internal.h)
#define RECUR_PASTE2(a, b) a##b
#define RECUR_PASTE(a, b) RECUR_PASTE2(a, b)
#define RECUR_ENTRY(name) __attribute__((visibility("hidden"))) RECUR_PASTE(name, MAPF_RECUR_SUFFIX)
#endif

int RECUR_ENTRY(mapf_recurrent_backward_packed)(const uint16_t *const *saved_dev, const uint8_t *comm_dev, const uint16_t *d_agent0_dev,
                            const uint16_t *weights_t_dev, int T, int E, int N, void *const *out_dev, const int32_t *row_index_dev,
                            int64_t num_rows, int agent0_stride, void *stream) {
    if (T < 1 || E < 0 || N < 1 || N > MAPF_RECUR_MAX_AGENTS || !saved_dev || !comm_dev || !d_agent0_dev || !weights_t_dev || !out_dev)
        return MAPF_ERR_INVALID_ARG;
    for (int i = 0; i < 8; ++i)
        if (!saved_dev[i] || (reinterpret_cast<uintptr_t>(saved_dev[i]) & 15)) return MAPF_ERR_INVALID_ARG;
    for (int i = 0; i < 7; ++i)
        if (!out_dev[i] || (reinterpret_cast<uintptr_t>(out_dev[i]) & 15)) return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(d_agent0_dev) & 15) || (reinterpret_cast<uintptr_t>(weights_t_dev) & 15)) return MAPF_ERR_INVALID_ARG;
    if (agent0_stride < 0 || agent0_stride > N) return MAPF_ERR_INVALID_ARG;
    if (E == 0) return MAPF_OK;
    if (agent0_stride > 0 && N > MAPF_RECUR_NARROW_AGENTS) return MAPF_ERR_UNSUPPORTED;  // tiles of several windows: the <= 48-agent kernels only
#if MAPF_RECUR_NT == 3
    if (N <= MAPF_RECUR_SMALL_AGENTS)  // one agent tile: the same kernel built for 16 agents (and what the forward of these shapes saved)
        return mapf_recurrent_backward_packed_nt1(saved_dev, comm_dev, d_agent0_dev, weights_t_dev, T, E, N, out_dev, row_index_dev, num_rows, agent0_stride,
                                                  stream);
    if (N <= 2 * MAPF_RECUR_SMALL_AGENTS)  // two tiles
        return mapf_recurrent_backward_packed_nt2(saved_dev, comm_dev, d_agent0_dev, weights_t_dev, T, E, N, out_dev, row_index_dev, num_rows, agent0_stride,
                                                  stream);
#endif
    BwdArgs a;
    a.hin0 = saved_dev[0];
    a.g1 = saved_dev[1];
    a.hr = saved_dev[2];
    a.qkv = saved_dev[3];
    a.ctx_unused = saved_dev[4];
    a.info_unused = saved_dev[5];
    a.g2 = saved_dev[6];
    a.P = saved_dev[7];
    a.comm = comm_dev;
    a.dA0 = d_agent0_dev;
    a.WT = weights_t_dev;
    a.d_gi1 = static_cast<uint16_t *>(out_dev[0]);
    a.d_gh1 = static_cast<uint16_t *>(out_dev[1]);
    a.d_gi2 = static_cast<uint16_t *>(out_dev[2]);
    a.d_gh2 = static_cast<uint16_t *>(out_dev[3]);
    a.d_info = static_cast<uint16_t *>(out_dev[4]);
    a.d_qkv = static_cast<uint16_t *>(out_dev[5]);
    a.bsum = static_cast<float *>(out_dev[6]);
    a.T = T;
    a.E = E;
    a.N = N;
    a.rowidx = row_index_dev;
    a.nrows = num_rows;
    a.a0s = agent0_stride;
    if (row_index_dev && (N > NA || num_rows < 1)) return MAPF_ERR_UNSUPPORTED;  // compact rows: the <= 48-agent kernels only
    if (N > NA) {  // 49..128 agents: csrc/mapf_recur_wide_bwd.hip
        RecurBwdArgs b{};
        b.hin0 = a.hin0, b.g1 = a.g1, b.hr = a.hr, b.qkv = a.qkv, b.ctx_unused = a.ctx_unused, b.info_unused = a.info_unused, b.g2 = a.g2, b.P = a.P;
        b.comm = a.comm, b.dA0 = a.dA0, b.WT = a.WT;
        b.d_gi1 = a.d_gi1, b.d_gh1 = a.d_gh1, b.d_gi2 = a.d_gi2, b.d_gh2 = a.d_gh2, b.d_info = a.d_info, b.d_qkv = a.d_qkv, b.bsum = a.bsum;
        b.T = T, b.E = E, b.N = N;
        return mapf_recur_wide_backward(b, static_cast<hipStream_t>(stream));
    }
    hipLaunchKernelGGL(recurrent_bwd_kernel, dim3(E), dim3(NTHR), 0, static_cast<hipStream_t>(stream), a);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

#if MAPF_RECUR_NT == 3
int mapf_recurrent_backward(const uint16_t *const *saved_dev, const uint8_t *comm_dev, const uint16_t *d_agent0_dev, const uint16_t *weights_t_dev, int T,
                            int E, int N, void *const *out_dev, const int32_t *row_index_dev, int64_t num_rows, void *stream) {
    return mapf_recurrent_backward_packed(saved_dev, comm_dev, d_agent0_dev, weights_t_dev, T, E, N, out_dev, row_index_dev, num_rows, 0, stream);
}
}  // extern "C"
#endif
#ifdef MAPF_RECUR_TRACE
#define RBWD_TRACE_NAME2(a, b) a##b
#define RBWD_TRACE_NAME(a, b) RBWD_TRACE_NAME2(a, b)
extern "C" int RBWD_TRACE_NAME(mapf_recur_btrace_read_nt, MAPF_RECUR_NT)(unsigned long long *out, int reset) {
    int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_btrace_n), sizeof(int)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_btrace), sizeof(unsigned long long) * 128) != hipSuccess) return -1;
    if (reset) {
        const int zero = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_btrace_n), &zero, sizeof(int)) != hipSuccess) return -1;
    }
    return n;
}
#endif
