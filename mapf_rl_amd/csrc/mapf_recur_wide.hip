// mapf_recur_wide.hip -- the recurrence of csrc/mapf_recur.hip (reference model.py:186-218 `Network.step`, model.py:242-249
// `bootstrap`: GRUCell, then 2 x [MultiHeadAttention over the agents of one environment, update GRUCell where the agent has a
// partner]) for environments of 49..128 agents: the reference's 64-agent fixture (test64_40_0.3.pkl, test.py:82-145) and
// BASELINE config 5 (128 agents).  Same inputs, outputs, saved tensors and rounding points as the <= 48-agent kernel; what
// differs is what fits: at 128 agents the hidden states alone take 68 KB of the 160 KB LDS, so
//   * the GRU cells update the hidden image IN PLACE: a wave holds the new values of its (channel block, agent half) jobs in
//     registers until every wave has read the old state (one barrier), then stores them;
//   * attention runs one head at a time on 64-channel q / k / v images, and everything behind the q|k|v projection of a head
//     -- scores, masked softmax, P V, W_O -- is ONE wave-local job per tile of 16 queries with no LDS round trip and no
//     barrier: the scores are accumulated transposed (S^T = K Q^T: key on the accumulator row, query on the lane), so a
//     query's softmax row lives in one lane column (row max / sum = in-lane + two cross-lane steps), the normalised P^T
//     accumulators ARE the B operand of ctx^T = V^T P^T (k index permuted consistently on the A side: V is read with
//     ds_read_b64_tr_b16 from its row-major image, rows 4 lh .. 4 lh + 3 and 16 + 4 lh .. of each 32-key step), and the ctx^T
//     accumulators are in turn the B operand of info^T += W_O[:, head] ctx^T (W_O fragments re-ordered by two 8-byte loads).
//     Nothing of size N x N ever exists outside registers.
// One workgroup (8 waves) per environment; NT = 4 (<= 64 agents) or 8 (<= 128) agent tiles of 16.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "mapf_dqn.h"
#include "mapf_env.h"
#include "mapf_recur_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
// A pointer that went through an opaque `asm volatile("" : "+s"(p))` (to keep step-invariant loads inside the step loop) comes
// back GENERIC, and hipcc then emits FLAT loads/stores -- which also count on the LDS counter, so every LDS wait waits for the
// weight stream too.  GLOBAL_PTR puts it back into the global address space.
#define GLOBAL_PTR(T, p) ((__attribute__((address_space(1))) T *)(p))
// (through native vector types: HIP's float4 / uint2 are classes, and copying one takes a generic `this`)
typedef __attribute__((ext_vector_type(4))) float gf32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int gu32x2;
__device__ __forceinline__ float4 gload_f4(const void *p) {
    const gf32x4 v = *GLOBAL_PTR(const gf32x4, p);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint2 gload_u2(const void *p) {
    const gu32x2 v = *GLOBAL_PTR(const gu32x2, p);
    return make_uint2(v[0], v[1]);
}
__device__ __forceinline__ void gstore_u2(void *p, uint2 v) { *GLOBAL_PTR(gu32x2, p) = gu32x2{v.x, v.y}; }
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int D = 256, HD = 64, NTHR = 512;
constexpr int H_ROW = D * 2 + 32;   // 544: hidden image rows (conflict-free ds_read_b128 over 16 consecutive rows)
constexpr int A_ROW = HD * 2 + 32;  // 160: 64-channel images q, k, v (one head) and info

template <int NT>
struct Lay {
    static constexpr int NA = 16 * NT, MW = NT / 2;  // agents (padded), mask words per agent row
    static constexpr int OFF_H = 0;
    static constexpr int OFF_Q = OFF_H + NA * H_ROW;
    static constexpr int OFF_K = OFF_Q + NA * A_ROW;
    static constexpr int OFF_V = OFF_K + NA * A_ROW;
    static constexpr int OFF_INFO = OFF_V + NA * A_ROW;
    static constexpr int OFF_MB = OFF_INFO + NA * A_ROW;  // comm mask of the step as bits
    static constexpr int OFF_UPD = OFF_MB + NA * MW * 4;
    static constexpr int BYTES = OFF_UPD + NA * 4;
    static_assert(BYTES <= 160 * 1024 && OFF_MB % 16 == 0, "LDS budget / alignment");
};

// weight / bias buffer layout (include/mapf_dqn.h: mapf_recurrent_infer)
constexpr int W_HH = 0, W_QKV = W_HH + 768 * 256, W_O = W_QKV + 384 * 256, U_IH = W_O + 64 * 128, U_HH = U_IH + 768 * 64;
static_assert(U_HH + 768 * 256 == MAPF_RECUR_WEIGHT_ELEMS, "header constant out of date");
constexpr int B_IH = 0, B_HH = 768, B_QKV = 1536, UB_IH = 1920, UB_HH = 2688;
static_assert(UB_HH + 768 == MAPF_RECUR_BIAS_ELEMS, "header constant out of date");

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ uint2 pack4(const f32x4 v) { return make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])); }
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
// (as in csrc/mapf_recur.hip: v_rcp_f32 instead of an IEEE division -- the cells' pointwise math is VALU time of the order of their MFMA time)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return __builtin_fmaf(2.f, __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)), -1.f); }
// two accumulator tiles (4 consecutive k each) -> one B / A fragment of 8 bf16
__device__ __forceinline__ bf16x8 frag_from_acc(const f32x4 lo, const f32x4 hi) {
    union {
        uint32_t u[4];
        bf16x8 v;
    } x;
    x.u[0] = pack2_bf16(lo[0], lo[1]);
    x.u[1] = pack2_bf16(lo[2], lo[3]);
    x.u[2] = pack2_bf16(hi[0], hi[1]);
    x.u[3] = pack2_bf16(hi[2], hi[3]);
    return x.v;
}

// acc{0,1,2}[n] += (tiles t0, t1, t2 of 16 rows of W) * X^T for NTH agent tiles starting at agent tile `tile0`; W packed in MFMA
// A-fragment order [tile][k-step][lane][8] (one wave load = one contiguous 1 KiB), X an LDS image with `xrow` bytes per agent
// row.  The A fragments are requested KC k-steps at a time (3 KC KiB in flight per wave) before the MFMAs that use them.
template <int KS, int KC, int NTH>
__device__ __forceinline__ void gemm3(f32x4 (&acc0)[NTH], f32x4 (&acc1)[NTH], f32x4 (&acc2)[NTH], const uint16_t *__restrict__ W, int t0,
                                      int t1, int t2, const unsigned char *X, int xrow, int tile0, int lane) {
    static_assert(KS % KC == 0, "");
    const int lr = lane & 15, lh = lane >> 4;
    // scalar tile bases + one 32-bit lane offset: as 64-bit per-lane addresses the compiler hoists every tile's pointer out of
    // the step loop and spills them (csrc/mapf_recur_bwd.hip: gemm2_lB)
    const unsigned char *w0 = reinterpret_cast<const unsigned char *>(W) + (size_t)t0 * KS * 1024;
    const unsigned char *w1 = reinterpret_cast<const unsigned char *>(W) + (size_t)t1 * KS * 1024;
    const unsigned char *w2 = reinterpret_cast<const unsigned char *>(W) + (size_t)t2 * KS * 1024;
    const uint32_t voff = (uint32_t)lane * 16u;
    const unsigned char *xb = X + (16 * tile0 + lr) * xrow + 16 * lh;
#pragma unroll
    for (int kc = 0; kc < KS; kc += KC) {
        bf16x8 a[KC][3];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const unsigned char *p0 = w0 + (kc + kk) * 1024, *p1 = w1 + (kc + kk) * 1024, *p2 = w2 + (kc + kk) * 1024;
            asm volatile("" : "+s"(p0), "+s"(p1), "+s"(p2));
            a[kk][0] = *GLOBAL_PTR(const bf16x8, p0 + voff);
            a[kk][1] = *GLOBAL_PTR(const bf16x8, p1 + voff);
            a[kk][2] = *GLOBAL_PTR(const bf16x8, p2 + voff);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KC; ++kk)
#pragma unroll
            for (int n = 0; n < NTH; ++n) {
                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(xb + 16 * n * xrow + 64 * (kc + kk));
                acc0[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][0], b, acc0[n], 0, 0, 0);
                acc1[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][1], b, acc1[n], 0, 0, 0);
                acc2[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][2], b, acc2[n], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// One GRU cell job: 16-channel block `cblk`, agent tiles tile0 .. tile0 + NTH - 1.  Returns the new state of the lane's
// (agent, 4 channels) cells in `outv` (the caller stores them after the barrier that ends every wave's reads of the old state).
//   r = s(gi_r + b_ir + W_hr h + b_hr), z likewise, n = tanh(gi_n + b_in + r (W_hn h + b_hn)), h' = (1-z) n + z h
// gi_* comes from global memory (GI_GLOBAL: the precomputed input projection, bf16 [agent][768]) or from a GEMM of Wi (KI k-steps)
// with the LDS image Xi.  `upd` (LDS int per agent, or nullptr): keep h where it is 0.  gsave: r, z, n, W_hn h + b_hn for the backward.
template <int NTH, bool GI_GLOBAL, int KI>
__device__ __forceinline__ void gru_job(uint2 (&outv)[NTH], int cblk, int tile0, const uint16_t *__restrict__ gi_glob,
                                        const uint16_t *__restrict__ Wi, const unsigned char *Xi, int xirow, const uint16_t *__restrict__ Wh,
                                        const float *__restrict__ bi, const float *__restrict__ bh, const unsigned char *H, const int *upd,
                                        int nagents, int lane, uint16_t *__restrict__ gsave) {
    const int lr = lane & 15, lh = lane >> 4;
    f32x4 ar[NTH], az[NTH], ani[NTH], anh[NTH];
    const int c0 = 16 * cblk + 4 * lh;  // this lane's 4 channels
    // (opaque scalar bases: the bias values do not depend on the step, and the compiler would otherwise load all of them -- 24
    // registers per job -- once in front of the step loop and keep them in scratch)
    asm volatile("" : "+s"(bi), "+s"(bh), "+s"(gi_glob), "+s"(gsave));
    const float4 bir = gload_f4(bi + c0), biz = gload_f4(bi + 256 + c0), bin = gload_f4(bi + 512 + c0);
    const float4 bhr = gload_f4(bh + c0), bhz = gload_f4(bh + 256 + c0), bhn = gload_f4(bh + 512 + c0);
#pragma unroll
    for (int n = 0; n < NTH; ++n) {
        ar[n] = f32x4{bir.x + bhr.x, bir.y + bhr.y, bir.z + bhr.z, bir.w + bhr.w};
        az[n] = f32x4{biz.x + bhz.x, biz.y + bhz.y, biz.z + bhz.z, biz.w + bhz.w};
        ani[n] = f32x4{bin.x, bin.y, bin.z, bin.w};
        anh[n] = f32x4{bhn.x, bhn.y, bhn.z, bhn.w};
        if (GI_GLOBAL) {
            const int agent = 16 * (tile0 + n) + lr;
            if (agent < nagents) {
                const unsigned char *g = reinterpret_cast<const unsigned char *>(gi_glob) + (uint32_t)(agent * 768 + c0) * 2u;
                const uint2 gr = gload_u2(g), gz = gload_u2(g + 512), gn = gload_u2(g + 1024);
                ar[n] += f32x4{bf16_lo(gr.x), bf16_hi(gr.x), bf16_lo(gr.y), bf16_hi(gr.y)};
                az[n] += f32x4{bf16_lo(gz.x), bf16_hi(gz.x), bf16_lo(gz.y), bf16_hi(gz.y)};
                ani[n] += f32x4{bf16_lo(gn.x), bf16_hi(gn.x), bf16_lo(gn.y), bf16_hi(gn.y)};
            }
        }
    }
    if (!GI_GLOBAL) gemm3<KI, KI, NTH>(ar, az, ani, Wi, cblk, 16 + cblk, 32 + cblk, Xi, xirow, tile0, lane);
    gemm3<8, 4, NTH>(ar, az, anh, Wh, cblk, 16 + cblk, 32 + cblk, H, H_ROW, tile0, lane);
#pragma unroll
    for (int n = 0; n < NTH; ++n) {
        const int agent = 16 * (tile0 + n) + lr;
        const uint2 hv = *reinterpret_cast<const uint2 *>(H + agent * H_ROW + c0 * 2);
        const float h[4] = {bf16_lo(hv.x), bf16_hi(hv.x), bf16_lo(hv.y), bf16_hi(hv.y)};
        f32x4 o, rg4, zg4, ng4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rg = sigmoidf_(ar[n][r]), zg = sigmoidf_(az[n][r]);
            const float ng = tanhf_(ani[n][r] + rg * anh[n][r]);
            o[r] = __builtin_fmaf(1.f - zg, ng, zg * h[r]);
            rg4[r] = rg;
            zg4[r] = zg;
            ng4[r] = ng;
        }
        if (gsave != nullptr && agent < nagents) {
            unsigned char *gs = reinterpret_cast<unsigned char *>(gsave) + (uint32_t)(agent * 1024 + c0) * 2u;
            gstore_u2(gs, pack4(rg4));
            gstore_u2(gs + 512, pack4(zg4));
            gstore_u2(gs + 1024, pack4(ng4));
            gstore_u2(gs + 1536, pack4(anh[n]));
        }
        const bool keep = upd != nullptr && upd[agent] == 0;
        outv[n] = keep ? hv : pack4(o);
    }
}

// A whole GRU cell over all agents, in place on the hidden image H: wave w owns agent half w >> 2 and the four channel blocks
// (w & 3) + 4 k (walked in an order rotated by `rot`: workgroups run in step and would otherwise request the same weight lines
// at the same moment).
template <int NT, bool GI_GLOBAL, int KI>
__device__ __forceinline__ void gru_phase(unsigned char *H, const uint16_t *__restrict__ gi_glob, const uint16_t *__restrict__ Wi,
                                          const unsigned char *Xi, int xirow, const uint16_t *__restrict__ Wh, const float *__restrict__ bi,
                                          const float *__restrict__ bh, const int *upd, int nagents, int w, int lane, int rot,
                                          uint16_t *__restrict__ gsave) {
    constexpr int NTH = NT / 2;
    const int lr = lane & 15, lh = lane >> 4;
    const int tile0 = (w >> 2) * NTH;
    uint2 outv[4][NTH];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cblk = ((w & 3) + 4 * k + rot) & 15;
        gru_job<NTH, GI_GLOBAL, KI>(outv[k], cblk, tile0, gi_glob, Wi, Xi, xirow, Wh, bi, bh, H, upd, nagents, lane, gsave);
    }
    __syncthreads();  // every wave has read the old state of every agent
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cblk = ((w & 3) + 4 * k + rot) & 15;
#pragma unroll
        for (int n = 0; n < NTH; ++n)
            *reinterpret_cast<uint2 *>(H + (16 * (tile0 + n) + lr) * H_ROW + (16 * cblk + 4 * lh) * 2) = outv[k][n];
    }
    __syncthreads();
}

// Attention of one head for the 16 queries of tile `ti` (see the file header): returns ctx of this head in c[td] (accumulator
// layout: lane column = query 16 ti + lr, rows = channels 16 td + 4 lh + r).  Q, K, V: this head's images; mb: mask bit rows.
template <int NT>
__device__ __forceinline__ void attention_tile(f32x4 (&c)[4], int ti, const unsigned char *Q, const unsigned char *K, const unsigned char *V,
                                               const uint32_t *mb, int lane) {
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    constexpr int MW = NT / 2;
    const int lr = lane & 15, lh = lane >> 4;
    const int i = 16 * ti + lr;
    // S^T[j][i] = k_j . q_i: NT key tiles x 2 k-steps
    bf16x8 bq[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) bq[kk] = *reinterpret_cast<const bf16x8 *>(Q + i * A_ROW + (32 * kk + 8 * lh) * 2);
    f32x4 s[NT];
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
        s[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(K + (16 * tj + lr) * A_ROW + (32 * kk + 8 * lh) * 2);
            s[tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[kk], s[tj], 0, 0, 0);
        }
    }
    // masked softmax over the keys of query i: s[tj][r] is key j = 16 tj + 4 lh + r  (model.py:75-78: fp32 scores, masked_fill(-1e9);
    // bits of keys >= N are 0, so padded keys get weight 0 as long as the row holds one allowed key; rows without any are
    // discarded by the caller: the agent has no partner, model.py:103)
    uint32_t mw[MW];
#pragma unroll
    for (int q = 0; q < MW; ++q) mw[q] = mb[i * MW + q];
    float mx = -3.0e38f;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
        const uint32_t bits = mw[tj >> 1] >> (16 * (tj & 1) + 4 * lh);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t m = 0u - ((bits >> r) & 1u);
            s[tj][r] = __uint_as_float((__float_as_uint(s[tj][r] * 0.125f) & m) | (__float_as_uint(-1e9f) & ~m));
            mx = fmaxf(mx, s[tj][r]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[tj][r] = __expf(s[tj][r] - mx);
            sum += s[tj][r];
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    // ctx^T[d][i] = sum_j v[j][d] P[i][j]: per 32-key step the P^T accumulators of key tiles 2 s2, 2 s2 + 1 are the B fragment
    // (element jj < 4: key 32 s2 + 4 lh + jj, else key 32 s2 + 16 + 4 lh + jj - 4); the A fragment takes the same keys from the
    // row-major V image with two transposed reads (lane 4 q + p of a 16-lane group addresses row q, columns 4 p .. 4 p + 3 of the
    // group's 4-row block and receives column lr of the 4 rows)
#pragma unroll
    for (int td = 0; td < 4; ++td) c[td] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < NT / 2; ++s2) {
        const bf16x8 bp = frag_from_acc(s[2 * s2] * inv, s[2 * s2 + 1] * inv);
#pragma unroll
        for (int td = 0; td < 4; ++td) {
            const unsigned char *p0 = V + (32 * s2 + 4 * lh + (lr >> 2)) * A_ROW + (16 * td + 4 * (lr & 3)) * 2;
            union {
                s16x4 h[2];
                bf16x8 v;
            } a;
            a.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
            a.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0 + 16 * A_ROW));
            c[td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, bp, c[td], 0, 0, 0);
        }
    }
}

// acc_info[ot] += W_O[16 ot .. +15][64 hd + d] ctx[i][d] with the ctx^T accumulators c[] as B operand (two 32-channel k-steps);
// the W_O fragment is re-ordered to the accumulators' k order by two 8-byte loads from the packed [tile][k-step][lane][8] image
__device__ __forceinline__ void wo_accumulate(f32x4 (&acc_info)[4], const f32x4 (&c)[4], const uint16_t *__restrict__ Wo, int hd, int lane) {
    const int lr = lane & 15, lh = lane >> 4;
    asm volatile("" : "+s"(Wo));  // step-invariant loads: keep them inside the step loop (see gru_job)
    const unsigned char *base = reinterpret_cast<const unsigned char *>(Wo) + (uint32_t)((16 * (lh >> 1) + lr) * 16 + 8 * (lh & 1));
#pragma unroll
    for (int s3 = 0; s3 < 2; ++s3) {
        const bf16x8 bc = frag_from_acc(c[2 * s3], c[2 * s3 + 1]);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            const unsigned char *p = base + (size_t)((ot * 4 + 2 * hd + s3) * 64) * 16;
            union {
                uint2 u[2];
                bf16x8 v;
            } a;
            a.u[0] = gload_u2(p);            // k = 4 lh + jj
            a.u[1] = gload_u2(p + 32 * 16);  // k = 16 + 4 lh + jj
            acc_info[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, bc, acc_info[ot], 0, 0, 0);
        }
    }
}

template <int NT, bool SAVE>
__global__ void __launch_bounds__(NTHR) recurrent_wide_kernel(const uint16_t *__restrict__ gi, const uint16_t *__restrict__ h0,
                                                             const uint8_t *__restrict__ comm, const uint16_t *__restrict__ W,
                                                             const float *__restrict__ bias, int T, int E, int N,
                                                             uint16_t *__restrict__ h_out, uint16_t *__restrict__ agent0_out, RecurSave sv) {
    using L = Lay<NT>;
    constexpr int NA = L::NA, NTH = NT / 2, MW = L::MW;
    __shared__ __attribute__((aligned(16))) unsigned char smem[L::BYTES];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = blockIdx.x, rot = blockIdx.x;
    unsigned char *H = smem + L::OFF_H;
    int *upd = reinterpret_cast<int *>(smem + L::OFF_UPD);
    uint32_t *mb = reinterpret_cast<uint32_t *>(smem + L::OFF_MB);

    // hidden state of this environment (rows >= N stay zero-initialised: they are computed like real agents and never stored)
    for (int i = tid; i < L::BYTES / 16; i += NTHR) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (h0 != nullptr)
        for (int i = tid; i < N * 32; i += NTHR) {  // 32 chunks of 16 B per agent
            const int a = i >> 5, ch = i & 31;
            *reinterpret_cast<uint4 *>(H + a * H_ROW + ch * 16) = *reinterpret_cast<const uint4 *>(h0 + ((long long)e * N + a) * D + ch * 8);
        }
    __syncthreads();

    const long long RTOT = (long long)T * E * N;
    auto save_hidden = [&](uint16_t *dst_rows) {  // N rows of 256 bf16 from the hidden image
        for (int i = tid; i < N * 32; i += NTHR) {
            const int a = i >> 5, ch = i & 31;
            *reinterpret_cast<uint4 *>(dst_rows + (long long)a * D + ch * 8) = *reinterpret_cast<const uint4 *>(H + a * H_ROW + ch * 16);
        }
    };
    for (int t = 0; t < T; ++t) {
        const long long row0 = ((long long)t * E + e) * N;  // first saved row of this (step, environment)
        // the lane index is made opaque once per step (and per round): every per-lane address below is then recomputed where it
        // is used -- a few integer instructions -- instead of being hoisted in front of the step loop as ~100 registers of
        // step-invariant addresses that end up in scratch
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        if (SAVE) save_hidden(sv.hin0 + row0 * D);
        // ---- this step's communication mask -> bit rows in LDS ----
        const uint8_t *comm_t = comm + ((long long)t * E + e) * N * N;
        for (int i = tid; i < NA * MW; i += NTHR) mb[i] = 0u;
        __syncthreads();
        for (int idx = tid; idx < N * N; idx += NTHR)
            if (comm_t[idx] != 0) {
                const int i = idx / N, j = idx - i * N;
                atomicOr(&mb[i * MW + (j >> 5)], 1u << (j & 31));
            }
        // ---- recurrent GRU cell, in place (its first barrier also publishes the mask bits) ----
        gru_phase<NT, true, 1>(H, gi + row0 * 768, nullptr, nullptr, 0, W + W_HH, bias + B_IH, bias + B_HH, nullptr, N, w, lane, rot,
                               SAVE ? sv.g1 + row0 * 1024 : nullptr);
        if (tid < NA) {  // model.py:103: an agent is updated by the communication block iff its mask row holds a partner besides itself
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < MW; ++q) cnt += __popc(mb[tid * MW + q]);
            upd[tid] = (tid < N && cnt > 1) ? 1 : 0;
        }
        // ---- two communication rounds (shared weights) ----
        for (int round = 0; round < 2; ++round) {
            const long long rq = (long long)round * RTOT + row0;
            asm volatile("" : "+v"(lane));
            const int lr = lane & 15, lh = lane >> 4;
            if (SAVE) save_hidden(sv.hr + rq * D);
            f32x4 acc_info[4];
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) acc_info[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int hd = 0; hd < 2; ++hd) {
                // q | k | v of this head = W h + b: wave w -> channel tile ct of each (q, k, v), agent half hf
                {
                    const int ct = (w + rot) & 3, hf = w >> 2, tile0 = hf * NTH;
                    f32x4 acc[3][NTH];
                    const float *bq = bias + B_QKV;
                    asm volatile("" : "+s"(bq));
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const float4 b4 = gload_f4(bq + 128 * g + 64 * hd + 16 * ct + 4 * lh);
#pragma unroll
                        for (int n = 0; n < NTH; ++n) acc[g][n] = f32x4{b4.x, b4.y, b4.z, b4.w};
                    }
                    gemm3<8, 4, NTH>(acc[0], acc[1], acc[2], W + W_QKV, 4 * hd + ct, 8 + 4 * hd + ct, 16 + 4 * hd + ct, H, H_ROW, tile0, lane);
#pragma unroll
                    for (int n = 0; n < NTH; ++n) {
                        const int agent = 16 * (tile0 + n) + lr, c0 = 16 * ct + 4 * lh;
                        const uint2 vq = pack4(acc[0][n]), vk = pack4(acc[1][n]), vv = pack4(acc[2][n]);
                        *reinterpret_cast<uint2 *>(smem + L::OFF_Q + agent * A_ROW + c0 * 2) = vq;
                        *reinterpret_cast<uint2 *>(smem + L::OFF_K + agent * A_ROW + c0 * 2) = vk;
                        *reinterpret_cast<uint2 *>(smem + L::OFF_V + agent * A_ROW + c0 * 2) = vv;
                        if (SAVE && agent < N) {
                            uint16_t *qs = sv.qkv + (rq + agent) * 384 + 64 * hd + c0;
                            *reinterpret_cast<uint2 *>(qs) = vq;
                            *reinterpret_cast<uint2 *>(qs + 128) = vk;
                            *reinterpret_cast<uint2 *>(qs + 256) = vv;
                        }
                    }
                }
                __syncthreads();
                if (w < NT) {  // one tile of 16 queries per wave (waves >= NT idle here when NT = 4)
                    f32x4 c[4];
                    attention_tile<NT>(c, w, smem + L::OFF_Q, smem + L::OFF_K, smem + L::OFF_V, mb, lane);
                    if (SAVE && 16 * w + lr < N) {
#pragma unroll
                        for (int td = 0; td < 4; ++td)
                            *reinterpret_cast<uint2 *>(sv.ctx + (rq + 16 * w + lr) * 128 + 64 * hd + 16 * td + 4 * lh) = pack4(c[td]);
                    }
                    wo_accumulate(acc_info, c, W + W_O, hd, lane);
                }
                __syncthreads();  // the next head's projection overwrites the images
            }
            if (w < NT) {
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) {
                    const uint2 v = pack4(acc_info[ot]);
                    *reinterpret_cast<uint2 *>(smem + L::OFF_INFO + (16 * w + lr) * A_ROW + (16 * ot + 4 * lh) * 2) = v;
                    if (SAVE && 16 * w + lr < N) *reinterpret_cast<uint2 *>(sv.info + (rq + 16 * w + lr) * 64 + 16 * ot + 4 * lh) = v;
                }
            }
            __syncthreads();
            // update cell, in place, where the agent has a partner
            gru_phase<NT, false, 2>(H, nullptr, W + U_IH, smem + L::OFF_INFO, A_ROW, W + U_HH, bias + UB_IH, bias + UB_HH, upd, N, w, lane, rot,
                                    SAVE ? sv.g2 + rq * 1024 : nullptr);
        }
        if (agent0_out != nullptr && tid < 32)  // agent 0's state after this step (model.py:248)
            *reinterpret_cast<uint4 *>(agent0_out + ((long long)t * E + e) * D + tid * 8) = *reinterpret_cast<const uint4 *>(H + tid * 16);
    }
    for (int i = tid; i < N * 32; i += NTHR) {
        const int a = i >> 5, ch = i & 31;
        *reinterpret_cast<uint4 *>(h_out + ((long long)e * N + a) * D + ch * 8) = *reinterpret_cast<const uint4 *>(H + a * H_ROW + ch * 16);
    }
}

}  // namespace

int mapf_recur_wide_forward(const uint16_t *gi, const uint16_t *h0, const uint8_t *comm, const uint16_t *W, const float *bias, int T, int E,
                            int N, uint16_t *h_out, uint16_t *agent0_out, const RecurSave *sv, hipStream_t stream) {
    if (N <= MAPF_RECUR_NARROW_AGENTS || N > MAPF_RECUR_MAX_AGENTS) return MAPF_ERR_INVALID_ARG;
    const RecurSave s = sv ? *sv : RecurSave{};
    if (N <= 64) {
        if (sv) hipLaunchKernelGGL((recurrent_wide_kernel<4, true>), dim3(E), dim3(NTHR), 0, stream, gi, h0, comm, W, bias, T, E, N, h_out, agent0_out, s);
        else hipLaunchKernelGGL((recurrent_wide_kernel<4, false>), dim3(E), dim3(NTHR), 0, stream, gi, h0, comm, W, bias, T, E, N, h_out, agent0_out, s);
    } else {
        if (sv) hipLaunchKernelGGL((recurrent_wide_kernel<8, true>), dim3(E), dim3(NTHR), 0, stream, gi, h0, comm, W, bias, T, E, N, h_out, agent0_out, s);
        else hipLaunchKernelGGL((recurrent_wide_kernel<8, false>), dim3(E), dim3(NTHR), 0, stream, gi, h0, comm, W, bias, T, E, N, h_out, agent0_out, s);
    }
    if (hipGetLastError() != hipSuccess) {
        std::fprintf(stderr, "mapf_recur_wide_forward: launch failed\n");
        return MAPF_ERR_HIP;
    }
    return MAPF_OK;
}
