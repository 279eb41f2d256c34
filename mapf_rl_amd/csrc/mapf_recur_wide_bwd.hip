// mapf_recur_wide_bwd.hip -- backward through time of csrc/mapf_recur_wide.hip (environments of 49..128 agents); the same
// inputs, outputs and math as csrc/mapf_recur_bwd.hip (autograd through reference model.py:242-249, driven by worker.py:316).
// One workgroup per environment walks the steps in reverse with DH, the gradient w.r.t. the hidden states, resident in LDS
// (68 KB at 128 agents).  What the <= 48-agent kernel keeps in LDS whole does not fit beside it, so
//   * GRU-cell backward is cut along the CHANNELS: per chunk of 32 channels the elementwise part (channel-local: dn, dz, dr
//     from the saved gates of the same channel) writes rows [dr | dz | dn | dn r] x 32 of a double-buffered 36 KB image, and the
//     products DH += W_hh^T d_gh, d_info = U_ih^T d_gi take exactly ONE k-step (32 channels) per gate from it, accumulating in
//     registers across the 8 chunks -- every transposed weight is still streamed once per cell;
//   * attention backward runs one head at a time on 64-channel images (q, k, v, d_ctx) in two wave-local passes with the
//     N x N quantities in registers only: pass A (a tile of 16 queries per wave) recomputes S^T = K Q^T and the softmax exactly as
//     the forward does, forms dP^T = V d_ctx^T, the row dot products, dS^T, and dq^T = K^T dS^T with the dS^T accumulators as
//     B operand; it leaves row max / 1/sum / dot in LDS.  Pass B (a tile of 16 keys per wave) recomputes S, P, dP, dS with the
//     query on the accumulator rows and forms dv^T = d_ctx^T P, dk^T = Q^T dS the same way (A operands through
//     ds_read_b64_tr_b16 from the row-major images).  The attention weights are NOT saved by the wide forward.
//   * DH += W_qkv^T d_qkv is taken per head from the dq / dk / dv images that replace q / k / v, so d_qkv never needs an LDS
//     image of its own; d_info goes through its global output rows (re-read by the same workgroup behind a barrier).
// Bias-gradient column sums accumulate in the per-environment output vector (one owner thread per column).
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

constexpr int D = 256, HD = 64, NTHR = 512;
constexpr int H_ROW = D * 2 + 32;    // 544
constexpr int A_ROW = HD * 2 + 32;   // 160: 64-channel images
constexpr int G_ROW = 128 * 2 + 32;  // 288: [dr | dz | dn | dn r] x 32 channels of one chunk
constexpr int NBSUM = MAPF_RECUR_BSUM_ELEMS;

template <int NT>
struct BL {
    static constexpr int NA = 16 * NT, MW = NT / 2;
    static constexpr int OFF_DH = 0;
    static constexpr int OFF_U = OFF_DH + NA * H_ROW;  // union: chunk images of the GRU backward | attention images
    static constexpr int OFF_G0 = OFF_U, OFF_G1 = OFF_U + NA * G_ROW;
    static constexpr int OFF_QI = OFF_U, OFF_KI = OFF_QI + NA * A_ROW, OFF_VI = OFF_KI + NA * A_ROW, OFF_CI = OFF_VI + NA * A_ROW;
    static constexpr int U_BYTES = (2 * NA * G_ROW > 4 * NA * A_ROW) ? 2 * NA * G_ROW : 4 * NA * A_ROW;
    static constexpr int OFF_ST = OFF_U + U_BYTES;  // per query: softmax max, 1 / sum, dot  (3 x NA floats)
    static constexpr int OFF_UPD = OFF_ST + 3 * NA * 4;
    static constexpr int OFF_MB = OFF_UPD + NA * 4;
    static constexpr int BYTES = OFF_MB + NA * MW * 4;
    static_assert(BYTES <= 160 * 1024 && OFF_ST % 16 == 0 && OFF_MB % 16 == 0, "LDS budget / alignment");
};

// transposed-weight buffer, in 1-KiB fragment units [tile][k-step] (include/mapf_dqn.h: mapf_recurrent_backward)
constexpr int WT_UIH = 0;                     // [3 gates][4 out tiles][8]   U_ih^T
constexpr int WT_UHH = WT_UIH + 3 * 4 * 8;    // [3][16][8]                  U_hh^T
constexpr int WT_WHH = WT_UHH + 3 * 16 * 8;   // [3][16][8]                  W_hh^T
constexpr int WT_WO = WT_WHH + 3 * 16 * 8;    // [8][2]                      W_O^T
constexpr int WT_QKV = WT_WO + 8 * 2;         // [16][12]                    W_qkv^T
static_assert((WT_QKV + 16 * 12) * 512 == MAPF_RECUR_WEIGHT_ELEMS, "header constant out of date");

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ uint2 pack4(const f32x4 v) { return make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])); }
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
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
// A fragment whose k index runs along the ROWS of a row-major LDS image, in the accumulator-chained k order of
// csrc/mapf_recur_wide.hip: rows r0 + 4 lh + jj (jj < 4) and r0 + 16 + 4 lh + jj - 4; 16 "columns" c0 .. c0 + 15 of the image
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char *img, int row_bytes, int r0, int c0, int lane) {
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const int lr = lane & 15, lh = lane >> 4;
    const unsigned char *p0 = img + (r0 + 4 * lh + (lr >> 2)) * row_bytes + (c0 + 4 * (lr & 3)) * 2;
    union {
        s16x4 h[2];
        bf16x8 v;
    } a;
    a.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
    a.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0 + 16 * row_bytes));
    return a.v;
}
__device__ __forceinline__ bf16x8 row_frag(const unsigned char *img, int row_bytes, int row, int k0, int lane) {
    return *reinterpret_cast<const bf16x8 *>(img + row * row_bytes + (k0 + 8 * (lane >> 4)) * 2);
}
// one 1-KiB weight fragment: scalar base (kept out of vector registers) + 16 bytes per lane
__device__ __forceinline__ bf16x8 wfrag(const unsigned char *WTB, int unit, uint32_t voff) {
    const unsigned char *p = WTB + (size_t)unit * 1024;
    asm volatile("" : "+s"(p));
    // (behind the asm the pointer is generic; back to the global address space, or hipcc emits FLAT loads, which also count on the
    // LDS counter and make every LDS wait wait for the weight stream)
    return *((const __attribute__((address_space(1))) bf16x8 *)(p + voff));
}

template <int NT>
__device__ __forceinline__ void add_to_dh(unsigned char *DH, const f32x4 (&acc)[NT], int tile, int lr, int lh) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        uint2 *cell = reinterpret_cast<uint2 *>(DH + (16 * n + lr) * H_ROW + (16 * tile + 4 * lh) * 2);
        const uint2 c = *cell;
        const f32x4 o = {bf16_lo(c.x) + acc[n][0], bf16_hi(c.x) + acc[n][1], bf16_lo(c.y) + acc[n][2], bf16_hi(c.y) + acc[n][3]};
        *cell = pack4(o);
    }
}

// GRU cell backward over all agents, 8 chunks of 32 channels (see the file header).
//   d = upd ? DH : 0;  dn = d (1-z)(1-n^2);  dz = d (h - n) z (1-z);  dr = dn hn r (1-r);  DH <- upd ? d z : DH;
//   d_gi = (dr, dz, dn), d_gh = (dr, dz, dn r) -> global rows (operands of the caller's weight-gradient GEMMs);
//   DH += Whh^T d_gh;  HAS_IH: d_info = U_ih^T d_gi -> global rows.
template <int NT, bool HAS_IH>
__device__ __forceinline__ void gru_bwd_phase(unsigned char *smem, const uint16_t *__restrict__ gates, const uint16_t *__restrict__ hin,
                                              const int *upd, uint16_t *__restrict__ dgi, uint16_t *__restrict__ dgh,
                                              const unsigned char *WTB, int wt_hh, uint16_t *__restrict__ d_info, float *__restrict__ bsum,
                                              int N, int w, int tid, int lane) {
    using L = BL<NT>;
    constexpr int NA = L::NA;
    asm volatile("" : "+v"(lane), "+v"(tid));  // per-lane addresses are recomputed here, not hoisted out of the step loop
    const int lr = lane & 15, lh = lane >> 4;
    const uint32_t voff = (uint32_t)lane * 16u;
    unsigned char *DH = smem + L::OFF_DH;
    f32x4 accH0[NT], accH1[NT], accI[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) accH0[n] = accH1[n] = accI[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int cc = 0; cc < 8; ++cc) {
        unsigned char *Gc = smem + ((cc & 1) ? L::OFF_G1 : L::OFF_G0);
        // ---- elementwise: (agent, 8 channels) tasks ----
        for (int task = tid; task < NA * 4; task += NTHR) {
            const int a = task >> 2, pc = task & 3, c0 = 32 * cc + 8 * pc;
            uint4 *grow = reinterpret_cast<uint4 *>(Gc + a * G_ROW + pc * 16);  // gate g at + 64 g bytes
            if (a >= N) {
                grow[0] = grow[4] = grow[8] = grow[12] = make_uint4(0, 0, 0, 0);
                continue;
            }
            const uint16_t *g = gates + (size_t)a * 1024 + c0;
            const uint4 vr = *reinterpret_cast<const uint4 *>(g), vz = *reinterpret_cast<const uint4 *>(g + 256),
                        vn = *reinterpret_cast<const uint4 *>(g + 512), vh = *reinterpret_cast<const uint4 *>(g + 768),
                        vx = *reinterpret_cast<const uint4 *>(hin + (size_t)a * D + c0);
            float r[8], z[8], nn[8], hn[8], h[8], d[8];
            unpack8(vr, r);
            unpack8(vz, z);
            unpack8(vn, nn);
            unpack8(vh, hn);
            unpack8(vx, h);
            uint4 *dcell = reinterpret_cast<uint4 *>(DH + a * H_ROW + c0 * 2);
            unpack8(*dcell, d);
            const bool on = upd == nullptr || upd[a] != 0;
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
            uint16_t *gi = dgi + (size_t)a * 768 + c0, *gh = dgh + (size_t)a * 768 + c0;
            const uint4 pr = pack8(dr), pz = pack8(dz), pn = pack8(dn), pnr = pack8(dnr);
            *reinterpret_cast<uint4 *>(gi) = pr;
            *reinterpret_cast<uint4 *>(gi + 256) = pz;
            *reinterpret_cast<uint4 *>(gi + 512) = pn;
            *reinterpret_cast<uint4 *>(gh) = pr;
            *reinterpret_cast<uint4 *>(gh + 256) = pz;
            *reinterpret_cast<uint4 *>(gh + 512) = pnr;
            grow[0] = pr;
            grow[4] = pz;
            grow[8] = pn;
            grow[12] = pnr;
            *dcell = pack8(dpass);
        }
        __syncthreads();
        // ---- bias gradients: column sums of the chunk over the agents (rows >= N are zero); column tid = 32 gate + j ----
        if (tid < 128) {
            float s = 0.f;
#pragma unroll 8
            for (int a = 0; a < NA; ++a) s += bf16_lo(*reinterpret_cast<const uint16_t *>(Gc + a * G_ROW + tid * 2));
            bsum[(tid >> 5) * 256 + 32 * cc + (tid & 31)] += s;
        }
        // ---- one k-step per gate: DH tiles w and w + 8 (+ d_info tile w on waves 0-3) ----
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const bf16x8 a0 = wfrag(WTB, wt_hh + (g * 16 + w) * 8 + cc, voff), a1 = wfrag(WTB, wt_hh + (g * 16 + w + 8) * 8 + cc, voff);
            bf16x8 ai = a0;
            if (HAS_IH && w < 4) ai = wfrag(WTB, WT_UIH + (g * 4 + w) * 8 + cc, voff);
            const int gh_col = (g == 2 ? 96 : 32 * g), gi_col = 32 * g;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const bf16x8 bh = row_frag(Gc, G_ROW, 16 * n + lr, gh_col, lane);
                accH0[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bh, accH0[n], 0, 0, 0);
                accH1[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bh, accH1[n], 0, 0, 0);
                if (HAS_IH && w < 4) {
                    const bf16x8 bi = g == 2 ? row_frag(Gc, G_ROW, 16 * n + lr, gi_col, lane) : bh;
                    accI[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ai, bi, accI[n], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();  // every chunk's in-place DH update is done
    add_to_dh<NT>(DH, accH0, w, lr, lh);
    add_to_dh<NT>(DH, accH1, w + 8, lr, lh);
    if (HAS_IH && w < 4) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int agent = 16 * n + lr;
            if (agent < N) *reinterpret_cast<uint2 *>(d_info + (size_t)agent * 64 + 16 * w + 4 * lh) = pack4(accI[n]);
        }
    }
    __syncthreads();
}

// masked score -> softmax numerator input: model.py:77 masked_fill(-1e9) as a bit select
__device__ __forceinline__ float masked(float s, uint32_t bit) {
    const uint32_t m = 0u - (bit & 1u);
    return __uint_as_float((__float_as_uint(s * 0.125f) & m) | (__float_as_uint(-1e9f) & ~m));
}

// attention backward of one head (file header).  d_info_rows: this (round, step, environment)'s global d_info rows.
template <int NT>
__device__ __forceinline__ void attn_bwd_head(unsigned char *smem, int hd, const uint16_t *__restrict__ qkv_rows, const uint16_t *__restrict__ d_info_rows,
                                              uint16_t *__restrict__ d_qkv_rows, const unsigned char *WTB, float *__restrict__ bsum, int N, int w,
                                              int tid, int lane) {
    using L = BL<NT>;
    constexpr int NA = L::NA, NTH = NT / 2, MW = L::MW;
    asm volatile("" : "+v"(lane), "+v"(tid));
    const int lr = lane & 15, lh = lane >> 4;
    const uint32_t voff = (uint32_t)lane * 16u;
    unsigned char *DH = smem + L::OFF_DH, *QI = smem + L::OFF_QI, *KI = smem + L::OFF_KI, *VI = smem + L::OFF_VI, *CI = smem + L::OFF_CI;
    float *st_m = reinterpret_cast<float *>(smem + L::OFF_ST), *st_inv = st_m + NA, *st_dot = st_inv + NA;
    const uint32_t *mb = reinterpret_cast<const uint32_t *>(smem + L::OFF_MB);
    // ---- (a) images q, k, v of this head (zero rows for agents >= N); d_ctx = W_O[:, head]^T d_info ----
    for (int i = tid; i < NA * 8; i += NTHR) {
        const int a = i >> 3, ch = i & 7;
        uint4 vq = make_uint4(0, 0, 0, 0), vk = vq, vv = vq;
        if (a < N) {
            const uint16_t *row = qkv_rows + (size_t)a * 384 + hd * HD + ch * 8;
            vq = *reinterpret_cast<const uint4 *>(row);
            vk = *reinterpret_cast<const uint4 *>(row + 128);
            vv = *reinterpret_cast<const uint4 *>(row + 256);
        }
        *reinterpret_cast<uint4 *>(QI + a * A_ROW + ch * 16) = vq;
        *reinterpret_cast<uint4 *>(KI + a * A_ROW + ch * 16) = vk;
        *reinterpret_cast<uint4 *>(VI + a * A_ROW + ch * 16) = vv;
    }
    {
        const int ot = w & 3, tile0 = (w >> 2) * NTH;
        f32x4 acc[NTH];
#pragma unroll
        for (int n = 0; n < NTH; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 a = wfrag(WTB, WT_WO + (4 * hd + ot) * 2 + kk, voff);
#pragma unroll
            for (int n = 0; n < NTH; ++n) {
                const int agent = 16 * (tile0 + n) + lr;
                uint4 raw = make_uint4(0, 0, 0, 0);
                if (agent < N) raw = *reinterpret_cast<const uint4 *>(d_info_rows + (size_t)agent * 64 + 32 * kk + 8 * lh);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, raw), acc[n], 0, 0, 0);
            }
        }
#pragma unroll
        for (int n = 0; n < NTH; ++n) *reinterpret_cast<uint2 *>(CI + (16 * (tile0 + n) + lr) * A_ROW + (16 * ot + 4 * lh) * 2) = pack4(acc[n]);
    }
    __syncthreads();
    f32x4 dq[4], dk[4], dv[4];
#pragma unroll
    for (int td = 0; td < 4; ++td) dq[td] = dk[td] = dv[td] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- (b) pass A: queries 16 w .. 16 w + 15 on the lane column ----
    if (w < NT) {
        const int i = 16 * w + lr;
        f32x4 s[NT], dp[NT];
        {
            const bf16x8 bq0 = row_frag(QI, A_ROW, i, 0, lane), bq1 = row_frag(QI, A_ROW, i, 32, lane);
            const bf16x8 bc0 = row_frag(CI, A_ROW, i, 0, lane), bc1 = row_frag(CI, A_ROW, i, 32, lane);
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
                s[tj] = dp[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
                s[tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(KI, A_ROW, 16 * tj + lr, 0, lane), bq0, s[tj], 0, 0, 0);
                s[tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(KI, A_ROW, 16 * tj + lr, 32, lane), bq1, s[tj], 0, 0, 0);
                dp[tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(VI, A_ROW, 16 * tj + lr, 0, lane), bc0, dp[tj], 0, 0, 0);
                dp[tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(VI, A_ROW, 16 * tj + lr, 32, lane), bc1, dp[tj], 0, 0, 0);
            }
        }
        uint32_t mw[MW];
#pragma unroll
        for (int q = 0; q < MW; ++q) mw[q] = mb[i * MW + q];
        float mx = -3.0e38f;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const uint32_t bits = mw[tj >> 1] >> (16 * (tj & 1) + 4 * lh);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[tj][r] = masked(s[tj][r], bits >> r);
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
        float dot = 0.f;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            // the forward multiplies v by P rounded to bf16: differentiate what was computed
            const uint2 pk = pack4(s[tj] * inv);
            s[tj] = f32x4{bf16_lo(pk.x), bf16_hi(pk.x), bf16_lo(pk.y), bf16_hi(pk.y)};
#pragma unroll
            for (int r = 0; r < 4; ++r) dot += s[tj][r] * dp[tj][r];
        }
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        if (lh == 0) {
            st_m[i] = mx;
            st_inv[i] = inv;
            st_dot[i] = dot;
        }
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[tj][r] = s[tj][r] * (dp[tj][r] - dot) * 0.125f;  // dS^T
        // dq^T[d][i] = sum_j k[j][d] dS[i][j]
#pragma unroll
        for (int s2 = 0; s2 < NT / 2; ++s2) {
            const bf16x8 b = frag_from_acc(dp[2 * s2], dp[2 * s2 + 1]);
#pragma unroll
            for (int td = 0; td < 4; ++td) dq[td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(KI, A_ROW, 32 * s2, 16 * td, lane), b, dq[td], 0, 0, 0);
        }
    }
    __syncthreads();  // row statistics visible
    // ---- (c) pass B: keys 16 w .. 16 w + 15 on the lane column, queries on the accumulator rows ----
    if (w < NT) {
        const int j = 16 * w + lr;
        f32x4 s[NT], dp[NT];
        {
            const bf16x8 bk0 = row_frag(KI, A_ROW, j, 0, lane), bk1 = row_frag(KI, A_ROW, j, 32, lane);
            const bf16x8 bv0 = row_frag(VI, A_ROW, j, 0, lane), bv1 = row_frag(VI, A_ROW, j, 32, lane);
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                s[ti] = dp[ti] = f32x4{0.f, 0.f, 0.f, 0.f};
                s[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(QI, A_ROW, 16 * ti + lr, 0, lane), bk0, s[ti], 0, 0, 0);
                s[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(QI, A_ROW, 16 * ti + lr, 32, lane), bk1, s[ti], 0, 0, 0);
                dp[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(CI, A_ROW, 16 * ti + lr, 0, lane), bv0, dp[ti], 0, 0, 0);
                dp[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(CI, A_ROW, 16 * ti + lr, 32, lane), bv1, dp[ti], 0, 0, 0);
            }
        }
        const int wsel = w >> 1, bsh = 16 * (w & 1) + lr;  // key j's bit inside a query's mask words
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const int i0 = 16 * ti + 4 * lh;
            const float4 m4 = *reinterpret_cast<const float4 *>(st_m + i0), v4 = *reinterpret_cast<const float4 *>(st_inv + i0),
                         d4 = *reinterpret_cast<const float4 *>(st_dot + i0);
            const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, iv[4] = {v4.x, v4.y, v4.z, v4.w}, dt[4] = {d4.x, d4.y, d4.z, d4.w};
            f32x4 p;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = __expf(masked(s[ti][r], mb[(i0 + r) * MW + wsel] >> bsh) - mm[r]) * iv[r];
            const uint2 pk = pack4(p);
            p = f32x4{bf16_lo(pk.x), bf16_hi(pk.x), bf16_lo(pk.y), bf16_hi(pk.y)};
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[ti][r] = p[r] * (dp[ti][r] - dt[r]) * 0.125f;  // dS
            s[ti] = p;
        }
        // dv^T[d][j] = sum_i d_ctx[i][d] P[i][j];  dk^T[d][j] = sum_i q[i][d] dS[i][j]
#pragma unroll
        for (int s2 = 0; s2 < NT / 2; ++s2) {
            const bf16x8 bp = frag_from_acc(s[2 * s2], s[2 * s2 + 1]), bs = frag_from_acc(dp[2 * s2], dp[2 * s2 + 1]);
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                dv[td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(CI, A_ROW, 32 * s2, 16 * td, lane), bp, dv[td], 0, 0, 0);
                dk[td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(QI, A_ROW, 32 * s2, 16 * td, lane), bs, dk[td], 0, 0, 0);
            }
        }
    }
    __syncthreads();  // every read of the q / k / v / d_ctx images is done
    // ---- (d) dq / dk / dv replace q / k / v; global rows of d_qkv ----
    if (w < NT) {
        const int a = 16 * w + lr;
#pragma unroll
        for (int td = 0; td < 4; ++td) {
            const uint2 vq = pack4(dq[td]), vk = pack4(dk[td]), vv = pack4(dv[td]);
            const int c = 16 * td + 4 * lh;
            *reinterpret_cast<uint2 *>(QI + a * A_ROW + c * 2) = vq;
            *reinterpret_cast<uint2 *>(KI + a * A_ROW + c * 2) = vk;
            *reinterpret_cast<uint2 *>(VI + a * A_ROW + c * 2) = vv;
            if (a < N) {
                uint16_t *row = d_qkv_rows + (size_t)a * 384 + hd * HD + c;
                *reinterpret_cast<uint2 *>(row) = vq;
                *reinterpret_cast<uint2 *>(row + 128) = vk;
                *reinterpret_cast<uint2 *>(row + 256) = vv;
            }
        }
    }
    __syncthreads();
    // ---- (e) DH += W_qkv^T[:, this head's q | k | v columns] d_qkv; bias column sums ----
    if (tid < 192) {
        const unsigned char *img = smem + L::OFF_QI + (tid >> 6) * NA * A_ROW;
        float sacc = 0.f;
        for (int a = 0; a < N; ++a) sacc += bf16_lo(*reinterpret_cast<const uint16_t *>(img + a * A_ROW + (tid & 63) * 2));
        bsum[2048 + 128 * (tid >> 6) + 64 * hd + (tid & 63)] += sacc;
    }
    {
        f32x4 acc0[NT], acc1[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc0[n] = acc1[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            const unsigned char *img = smem + L::OFF_QI + part * NA * A_ROW;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ks = 4 * part + 2 * hd + kk;
                const bf16x8 a0 = wfrag(WTB, WT_QKV + w * 12 + ks, voff), a1 = wfrag(WTB, WT_QKV + (w + 8) * 12 + ks, voff);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const bf16x8 b = row_frag(img, A_ROW, 16 * n + lr, 32 * kk, lane);
                    acc0[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc0[n], 0, 0, 0);
                    acc1[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc1[n], 0, 0, 0);
                }
            }
        }
        add_to_dh<NT>(DH, acc0, w, lr, lh);
        add_to_dh<NT>(DH, acc1, w + 8, lr, lh);
    }
    __syncthreads();
}

template <int NT>
__global__ void __launch_bounds__(NTHR) recurrent_wide_bwd_kernel(RecurBwdArgs A) {
    using L = BL<NT>;
    constexpr int NA = L::NA, MW = L::MW;
    __shared__ __attribute__((aligned(16))) unsigned char smem[L::BYTES];
    const int tid0 = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int e = blockIdx.x, T = A.T, E = A.E, N = A.N;
    const long long RTOT = (long long)T * E * N;
    unsigned char *DH = smem + L::OFF_DH;
    int *upd = reinterpret_cast<int *>(smem + L::OFF_UPD);
    uint32_t *mb = reinterpret_cast<uint32_t *>(smem + L::OFF_MB);
    const unsigned char *WTB = reinterpret_cast<const unsigned char *>(A.WT);
    float *bsum = A.bsum + (long long)e * NBSUM;

    for (int i = tid0; i < L::BYTES / 16; i += NTHR) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid0; i < NBSUM; i += NTHR) bsum[i] = 0.f;
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        const long long row0 = ((long long)t * E + e) * N;
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        // ---- external gradient of agent 0's state after step t; this step's mask bits and partner flags ----
        if (tid < 64) {
            const uint2 g = *reinterpret_cast<const uint2 *>(A.dA0 + ((long long)t * E + e) * D + tid * 4);
            uint2 *cell = reinterpret_cast<uint2 *>(DH + tid * 8);
            const uint2 c = *cell;
            const f32x4 o = {bf16_lo(c.x) + bf16_lo(g.x), bf16_hi(c.x) + bf16_hi(g.x), bf16_lo(c.y) + bf16_lo(g.y), bf16_hi(c.y) + bf16_hi(g.y)};
            *cell = pack4(o);
        }
        for (int i = tid; i < NA * MW; i += NTHR) mb[i] = 0u;
        __syncthreads();
        {
            const uint8_t *cm = A.comm + ((long long)t * E + e) * N * N;
            for (int idx = tid; idx < N * N; idx += NTHR)
                if (cm[idx] != 0) {
                    const int i = idx / N, j = idx - i * N;
                    atomicOr(&mb[i * MW + (j >> 5)], 1u << (j & 31));
                }
        }
        __syncthreads();
        if (tid < NA) {
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < MW; ++q) cnt += __popc(mb[tid * MW + q]);
            upd[tid] = (tid < N && cnt > 1) ? 1 : 0;  // model.py:103
        }
        __syncthreads();

        for (int q = 1; q >= 0; --q) {
            const long long rq = (long long)q * RTOT + row0;
            // update-cell backward: DH, d_gi2 / d_gh2 rows, d_info rows
            gru_bwd_phase<NT, true>(smem, A.g2 + rq * 1024, A.hr + rq * D, upd, A.d_gi2 + rq * 768, A.d_gh2 + rq * 768, WTB, WT_UHH,
                                    A.d_info + rq * 64, bsum, N, w, tid, lane);
            // attention backward + DH += W_qkv^T d_qkv, one head at a time
#pragma unroll 1
            for (int hd = 0; hd < 2; ++hd)
                attn_bwd_head<NT>(smem, hd, A.qkv + rq * 384, A.d_info + rq * 64, A.d_qkv + rq * 384, WTB, bsum, N, w, tid, lane);
        }
        // recurrent cell backward: d_gi1 is the gradient w.r.t. the GRU input projection
        gru_bwd_phase<NT, false>(smem, A.g1 + row0 * 1024, A.hin0 + row0 * D, nullptr, A.d_gi1 + row0 * 768, A.d_gh1 + row0 * 768, WTB, WT_WHH,
                                 nullptr, bsum + 1024, N, w, tid, lane);
    }
}

}  // namespace

int mapf_recur_wide_backward(const RecurBwdArgs &a, hipStream_t stream) {
    if (a.N <= MAPF_RECUR_NARROW_AGENTS || a.N > MAPF_RECUR_MAX_AGENTS) return MAPF_ERR_INVALID_ARG;
    if (a.N <= 64) hipLaunchKernelGGL(recurrent_wide_bwd_kernel<4>, dim3(a.E), dim3(NTHR), 0, stream, a);
    else hipLaunchKernelGGL(recurrent_wide_bwd_kernel<8>, dim3(a.E), dim3(NTHR), 0, stream, a);
    if (hipGetLastError() != hipSuccess) {
        std::fprintf(stderr, "mapf_recur_wide_backward: launch failed\n");
        return MAPF_ERR_HIP;
    }
    return MAPF_OK;
}
