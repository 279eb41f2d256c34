// mapf_recur.hip -- inference recurrence of the DQN behind the encoder (reference model.py:186-218 `Network.step`
// and the no-gradient `bootstrap` of the target network, model.py:242-249 called from worker.py:300-303):
//     hidden = GRUCell(latent, hidden);  2 x [ info = MultiHeadAttention(hidden, comm_mask);
//                                              hidden = where(has_partner, GRUCell(info, hidden), hidden) ]
// for T consecutive steps, an environment (or two, see the LDS maps) at a time per workgroup (see include/mapf_dqn.h: mapf_recurrent_infer).
//
// Everything after the GRU's input projection is independent between environments (attention only mixes the <= 48
// agents of one environment), so a workgroup keeps its environment's hidden states in LDS for all T steps and only
// streams the weights (1.1 MB, L2-resident).  Through PyTorch this is ~14 small launches per step (GEMMs on 160 K x
// 256 rows, a fused GRU cell that also writes a 5x workspace, SDPA, where, copies): 1.9 ms of a 12 ms actor
// iteration, 18 x that in the target network's bootstrap.
//
// Conventions (as in csrc/mapf_encoder.hip): v_mfma_f32_16x16x32_bf16 with A = weights (16 output channels x 32 k,
// pre-packed in fragment order so that one wave load is a contiguous 1 KiB) and
// B = activations (32 k x 16 agents, rows of an LDS image with one row per agent), so an accumulator lane holds
// 4 consecutive channels of one agent = one 8-byte LDS store.  LDS rows are 32 bytes longer than a multiple of 256
// (conflict-free ds_read_b128 over 16 consecutive rows, tools/micro/lds_b128_bank.hip).
// The GRU gates of a 16-channel block (r, z, n) are accumulated by the same wave, so the cell's pointwise math runs
// on registers; gi (the input projection W_ih x, one large GEMM over all steps) is an input.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "mapf_dqn.h"
#include "mapf_env.h"
#include "mapf_recur_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

#ifndef MAPF_RECUR_ABLATE  // diagnostic builds only (tools/micro/recur_ablate.py): 1 GRU, 2 QKV, 4 attention, 8 W_O, 16 update cell
#define MAPF_RECUR_ABLATE 0
#endif
#ifndef MAPF_RECUR_NT  // agent tiles of 16 the kernels of this translation unit are built for: 3 (up to 48 agents per environment) in
#define MAPF_RECUR_NT 3  // csrc/mapf_recur.hip itself, 1 (up to 16) in csrc/mapf_recur_nt1.hip, which includes this file
#endif
constexpr int NT = MAPF_RECUR_NT;
#ifdef MAPF_RECUR_TRACE  // diagnostic builds only (tools/micro/recur_trace.py): cycle stamps of wave 0 of workgroup MAPF_RECUR_TRACE_WG
#ifndef MAPF_RECUR_TRACE_WG
#define MAPF_RECUR_TRACE_WG 0
#endif
__device__ unsigned long long g_trace[128];
__device__ int g_trace_n;
// stamps go to LDS (s_memtime and ds_write wait on lgkmcnt only): a stamp that touched global memory would wait for every weight load
// in flight (vmcnt retires in order) and so serialise exactly what is being measured
__shared__ unsigned long long trace_lds[128];
__shared__ int trace_lds_n;
#define TRACE_POINT(id)                                                                                                 \
    do {                                                                                                                \
        if (threadIdx.x == 0 && blockIdx.x == MAPF_RECUR_TRACE_WG) {                                                    \
            const int k_ = trace_lds_n;                                                                                 \
            if (k_ < 126) {                                                                                             \
                trace_lds[k_] = ((unsigned long long)(id) << 56) | (__builtin_readcyclecounter() & 0xFFFFFFFFFFFFFFull); \
                trace_lds_n = k_ + 1;                                                                                   \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
#define TRACE_BEGIN() do { if (threadIdx.x == 0) trace_lds_n = 0; } while (0)
#define TRACE_END()                                                                  \
    do {                                                                             \
        if (threadIdx.x == 0 && blockIdx.x == MAPF_RECUR_TRACE_WG) {                 \
            for (int k_ = 0; k_ < trace_lds_n; ++k_) g_trace[k_] = trace_lds[k_];    \
            g_trace_n = trace_lds_n;                                                 \
        }                                                                            \
    } while (0)
#else
#define TRACE_POINT(id) do { } while (0)
#define TRACE_BEGIN() do { } while (0)
#define TRACE_END() do { } while (0)
#endif
constexpr int NA = 16 * NT;
constexpr int D = 256;           // hidden size (config.latent_dim)
constexpr int HD = 64;           // attention head dim (comm output_dim)
// MAPF_RECUR_WAVES: 8 (what is built) = two waves per SIMD with 256 registers each and the weight stream held one job ahead in registers.
// 16 (round 5, A/B builds only: tools/micro/recur_multi.py) = four waves per SIMD with 128 registers each -- one channel block per wave,
// fragments requested a k-step ahead -- so that the cells' pointwise math of one wave would run under the MFMAs of its three SIMD
// partners.  Same bits; SLOWER: 0.560 -> 0.603 ms per 4096 x 40 step, 0.509 -> 0.533 per 18 x 192 x 40 pass (27-49 spilled registers at
// three agent tiles; requesting 2 / 3 k-steps ahead: 0.64 / 0.73).  profiles/r05_recurrence_16waves.txt
#ifndef MAPF_RECUR_WAVES
#define MAPF_RECUR_WAVES 8
#endif
constexpr int NTHR = 64 * MAPF_RECUR_WAVES;
constexpr int NWV = MAPF_RECUR_WAVES;
static_assert(NWV == 8 || NWV == 16, "");

// LDS image (bytes)
constexpr int H_ROW = D * 2 + 32;          // 544
constexpr int H_BYTES = NA * H_ROW;        // one hidden buffer
constexpr int QK_ROW = 256 * 2 + 32;       // q (2 heads x 64) | k (2 heads x 64)
constexpr int VT_ROW = 64 * 2 + 32;        // v transposed: row = (head, d), 64 agent slots
constexpr int CTX_ROW = 128 * 2 + 32;
constexpr int INFO_ROW = 64 * 2 + 32;
constexpr int OFF_H0 = 0, OFF_H1 = OFF_H0 + H_BYTES;
constexpr int OFF_QK = OFF_H1 + H_BYTES;
constexpr int OFF_VT = OFF_QK + NA * QK_ROW;
// (round 5: the attention of a (head, agent tile) runs in one wave's registers -- no score / softmax images; ctx is written while
// other waves still read q | k, so it has its own bytes)
constexpr int OFF_CTX = OFF_VT + 128 * VT_ROW;
constexpr int OFF_INFO = OFF_CTX + NA * CTX_ROW;
constexpr int OFF_UPD = OFF_INFO + NA * INFO_ROW;
constexpr int OFF_MB = OFF_UPD + 64 * 4;     // comm mask of the step as bits: 2 words per agent row
constexpr int OFF_RIDX = OFF_MB + NA * 2 * 4;  // global row of every agent at this step (-1: none), see recurrent_infer_kernel
constexpr int OFF_BSUM = OFF_RIDX + NA * 4;   // gate biases of both cells as the accumulators want them: [cell][r: b_ir + b_hr | z: b_iz + b_hz | b_in | b_hn][256] f32
constexpr int LDS_BYTES = OFF_BSUM + 2 * 4 * 256 * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(OFF_CTX % 16 == 0 && OFF_INFO % 16 == 0 && OFF_UPD % 16 == 0 && OFF_BSUM % 16 == 0 && LDS_BYTES % 16 == 0, "");
// TWO environments per workgroup (round 5, recurrent_infer_kernel<false, true>; the one-step launches of the actor, where a CU has many
// environments to walk): every weight fragment a wave fetches serves both environments' MFMAs -- a step is bound by the L2 -> CU weight
// stream (1.8 MB per environment step at 64 B/clk), this halves it.  Two of the images above do not fit in 160 KB, so per environment only
//   [ H a | H b | v^T ]   (H a / H b: hidden state, current and next, trading places as above)
// is kept and the rest lives in bytes that are dead at the time:
//   q | k      in the NEXT-state buffer (dead between a cell and the update cell behind it; same 544-byte rows),
//   ctx        over the q half of the same rows: the wave of a (head, agent tile) is the only reader of that tile's q of that head, and has
//              it in registers before its first ctx store (one wave per (head, tile) in this mode, whatever NT),
//   info       over the first NA rows of v^T (dead between the attention and the next q|k|v; the slots >= NA of those rows, which the
//              attention multiplies by P = 0, are zeroed again in front of the next q|k|v: 0 x garbage could be 0 x Inf).
constexpr int P_OFF_VT = 2 * H_BYTES;
constexpr int P_ENV = P_OFF_VT + 128 * VT_ROW;
constexpr int P_OFF_UPD = 2 * P_ENV;                 // [2][64] int
constexpr int P_OFF_MB = P_OFF_UPD + 2 * 64 * 4;     // [2][NA][2] words
constexpr int P_OFF_RIDX = P_OFF_MB + 2 * NA * 2 * 4;
constexpr int P_OFF_BSUM = P_OFF_RIDX + 2 * NA * 4;
constexpr int P_LDS_BYTES = P_OFF_BSUM + 2 * 4 * 256 * 4;
static_assert(P_LDS_BYTES <= 160 * 1024, "LDS budget (two environments)");
static_assert(P_ENV % 16 == 0 && P_OFF_BSUM % 16 == 0 && QK_ROW == H_ROW && INFO_ROW == VT_ROW, "");

// weight buffer (bf16 elements) and bias buffer (f32 elements), see mapf_dqn.h
constexpr int W_HH = 0, W_QKV = W_HH + 768 * 256, W_O = W_QKV + 384 * 256, U_IH = W_O + 64 * 128, U_HH = U_IH + 768 * 64;
constexpr int W_TOTAL = U_HH + 768 * 256;
static_assert(W_TOTAL == MAPF_RECUR_WEIGHT_ELEMS, "header constant out of date");
constexpr int B_IH = 0, B_HH = 768, B_QKV = 1536, UB_IH = 1920, UB_HH = 2688, B_TOTAL = 3456;
static_assert(B_TOTAL == MAPF_RECUR_BIAS_ELEMS, "header constant out of date");

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
// v_rcp_f32 (1 ulp) instead of an IEEE division (ten instructions): the cells' pointwise math is VALU time of the same order as their
// MFMA time (tools/micro/recur_trace.py), and every result is rounded to bf16 next
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return __builtin_fmaf(2.f, __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)), -1.f); }

// value of another lane of the same DPP quad (0x4E: lanes 2,3,0,1; 0xB1: lanes 1,0,3,2) -- VALU only, no LDS-pipe shuffle
template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

// acc[n] += (tile `t` of 16 rows of W) * X^T for the NT agent tiles; W packed [tile][k-step][lane][8] in MFMA A-fragment order (one wave load = one
// contiguous 1 KiB = 8 full cache lines; row-major weights cost 16 half lines per load and ran the L2 -> CU path at a quarter of its rate),
// X an LDS image with `xrow` bytes per agent row.  All KS A fragments are loaded before the first MFMA.
template <int KS>
__device__ __forceinline__ void gemm16(f32x4 (&acc)[NT], const uint16_t *__restrict__ W, int t, const unsigned char *X, int xrow, int lane) {
    const int lr = lane & 15, lh = lane >> 4;
    bf16x8 a[KS];
    const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(W) + (long long)t * KS * 64 + lane;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) a[kk] = wp[kk * 64];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(X + (16 * n + lr) * xrow + (32 * kk + 8 * lh) * 2);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk], b, acc[n], 0, 0, 0);
        }
}

// ---- the weight stream ----
// A wave's weight fragments for one phase (three output tiles x 8 k-steps = 24 KiB) live in one register array, `wf`, for the whole
// kernel.  Whoever consumes a k-step of it requests the same k-step of the NEXT job of this wave into the registers just freed
// (stream_mfma<.., true>), across the phase barriers: second channel block of a cell, the q|k|v tiles behind a cell, the update
// cell's tiles behind q|k|v (they arrive during the attention phases), the next step's first block behind the last update cell.
// The L2 -> CU stream, which a step is bound by together with its barriers, then runs under the pointwise math and the attention
// phases instead of starting at the top of every phase (tools/micro/recur_trace.py: 2.7-8 k cycles of exposed load time per phase
// before, of ~100 k per step at 40 agents).
struct Frag3 {  // the fragments of three 16-row tiles of a packed matrix, already at this lane: k-step kk of tile g is p[g][kk * 64]
    const bf16x8 *p[3];
};
__device__ __forceinline__ Frag3 frag3(const uint16_t *__restrict__ W, int t0, int t1, int t2, int KS, int lane) {
    const bf16x8 *b = reinterpret_cast<const bf16x8 *>(W) + lane;
    return Frag3{{b + (long long)t0 * KS * 64, b + (long long)t1 * KS * 64, b + (long long)t2 * KS * 64}};
}
__device__ __forceinline__ Frag3 gate_frags(const uint16_t *__restrict__ W, int cblk, int KS, int lane) {  // gate tiles r, z, n of a channel block
    return frag3(W, cblk, 16 + cblk, 32 + cblk, KS, lane);
}

template <int KS>
__device__ __forceinline__ void load_frags(bf16x8 (&a)[KS][3], const Frag3 &f) {
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int g = 0; g < 3; ++g) a[kk][g] = f.p[g][kk * 64];
}

// acc_g[n] += a[kk][g] x X for all k-steps; with RELOAD every k-step's fragments are replaced by `next`'s once used
template <int KS, bool RELOAD, bool PIN = false>
__device__ __forceinline__ void stream_mfma(f32x4 (&acc0)[NT], f32x4 (&acc1)[NT], f32x4 (&acc2)[NT], bf16x8 (&a)[KS][3], const unsigned char *X,
                                            int xrow, int lane, const Frag3 &next) {
    const int lr = lane & 15, lh = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(X + (16 * n + lr) * xrow + (32 * kk + 8 * lh) * 2);
            acc0[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][0], b, acc0[n], 0, 0, 0);
            acc1[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][1], b, acc1[n], 0, 0, 0);
            acc2[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][2], b, acc2[n], 0, 0, 0);
        }
        if (RELOAD) {  // pinned behind this k-step's MFMAs: left to itself the scheduler hoists every reload to the top (into NEW registers: 430 spills)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 3; ++g) a[kk][g] = next.p[g][kk * 64];
            __builtin_amdgcn_sched_barrier(0);
        } else if (PIN) {  // (two environments: the first one's k-steps stay in order too, or all their activation reads are hoisted to the top)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- GRU cell of a 16-channel block, all agents: gates accumulated in registers, pointwise math, new state to Hout ----
//   r = s(gi_r + b_ir + W_hr h + b_hr), z likewise, n = tanh(gi_n + b_in + r (W_hn h + b_hn)), h' = (1-z) n + z h
// gi_* comes either from global memory (GI_GLOBAL: precomputed input projection, bf16 [row][768]) or from a GEMM of
// W_i (KI*32 columns) with the LDS image Xi.  `upd` (LDS int per agent, or nullptr): keep h where it is 0.
// `ridx` (LDS int per agent): the agent's row in gsave at this step, -1 = none (nothing saved).
// A wave owns TWO channel blocks per cell (16 blocks, 8 waves): gru_pair.

// what a block's gate accumulators start from besides the biases (LDS, OFF_BSUM): with GI_GLOBAL, the precomputed input projection
struct GruInit {
    uint2 gr[NT], gz[NT], gn[NT];
    int row[NT];
};

// row[n]: the row of agent 16 n + lr in gi_glob, -1 = none
__device__ __forceinline__ void gru_fetch_gi(GruInit &s, int cblk, const uint16_t *__restrict__ gi_glob, const int (&row)[NT], int lh) {
    const int c0 = 16 * cblk + 4 * lh;  // this lane's 4 channels
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        s.row[n] = row[n];
        s.gr[n] = s.gz[n] = s.gn[n] = make_uint2(0u, 0u);
        if (row[n] >= 0) {
            const uint16_t *g = gi_glob + (long long)row[n] * 768 + c0;
            s.gr[n] = *reinterpret_cast<const uint2 *>(g);
            s.gz[n] = *reinterpret_cast<const uint2 *>(g + 256);
            s.gn[n] = *reinterpret_cast<const uint2 *>(g + 512);
        }
    }
}

// bsum: the cell's [4][256] f32 block of OFF_BSUM
template <bool GI_GLOBAL>
__device__ __forceinline__ void gru_start(const GruInit &s, const float *bsum, int c0, f32x4 (&ar)[NT], f32x4 (&az)[NT], f32x4 (&ani)[NT],
                                          f32x4 (&anh)[NT]) {
    const f32x4 br = *reinterpret_cast<const f32x4 *>(bsum + c0), bz = *reinterpret_cast<const f32x4 *>(bsum + 256 + c0),
                bni = *reinterpret_cast<const f32x4 *>(bsum + 512 + c0), bnh = *reinterpret_cast<const f32x4 *>(bsum + 768 + c0);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        ar[n] = br;
        az[n] = bz;
        ani[n] = bni;
        anh[n] = bnh;
        if (GI_GLOBAL && s.row[n] >= 0) {
            ar[n] += f32x4{bf16_lo(s.gr[n].x), bf16_hi(s.gr[n].x), bf16_lo(s.gr[n].y), bf16_hi(s.gr[n].y)};
            az[n] += f32x4{bf16_lo(s.gz[n].x), bf16_hi(s.gz[n].x), bf16_lo(s.gz[n].y), bf16_hi(s.gz[n].y)};
            ani[n] += f32x4{bf16_lo(s.gn[n].x), bf16_hi(s.gn[n].x), bf16_lo(s.gn[n].y), bf16_hi(s.gn[n].y)};
        }
    }
}

__device__ __forceinline__ void gru_finish(int cblk, const f32x4 (&ar)[NT], const f32x4 (&az)[NT], const f32x4 (&ani)[NT], const f32x4 (&anh)[NT],
                                           const unsigned char *Hin, unsigned char *Hout, const int *upd, const int *ridx, int lr, int lh,
                                           uint16_t *__restrict__ gsave) {
    const int c0 = 16 * cblk + 4 * lh;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int agent = 16 * n + lr;
        const uint2 hv = *reinterpret_cast<const uint2 *>(Hin + agent * H_ROW + c0 * 2);
        const float h[4] = {bf16_lo(hv.x), bf16_hi(hv.x), bf16_lo(hv.y), bf16_hi(hv.y)};
        float o[4], rg4[4], zg4[4], ng4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rg = sigmoidf_(ar[n][r]), zg = sigmoidf_(az[n][r]);
            const float ng = tanhf_(ani[n][r] + rg * anh[n][r]);
            o[r] = __builtin_fmaf(1.f - zg, ng, zg * h[r]);  // (spelled out: which product the compiler fuses must not depend on the code around it)
            rg4[r] = rg;
            zg4[r] = zg;
            ng4[r] = ng;
        }
        if (gsave != nullptr && ridx[agent] >= 0) {  // training forward: r, z, n, W_hn h + b_hn of this cell for the backward pass
            uint16_t *gs = gsave + (long long)ridx[agent] * 1024 + c0;
            *reinterpret_cast<uint2 *>(gs) = make_uint2(pack2_bf16(rg4[0], rg4[1]), pack2_bf16(rg4[2], rg4[3]));
            *reinterpret_cast<uint2 *>(gs + 256) = make_uint2(pack2_bf16(zg4[0], zg4[1]), pack2_bf16(zg4[2], zg4[3]));
            *reinterpret_cast<uint2 *>(gs + 512) = make_uint2(pack2_bf16(ng4[0], ng4[1]), pack2_bf16(ng4[2], ng4[3]));
            *reinterpret_cast<uint2 *>(gs + 768) = make_uint2(pack2_bf16(anh[n][0], anh[n][1]), pack2_bf16(anh[n][2], anh[n][3]));
        }
        const bool keep = upd != nullptr && upd[agent] == 0;
        *reinterpret_cast<uint2 *>(Hout + agent * H_ROW + c0 * 2) = keep ? hv : make_uint2(pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]));
    }
}

// The cell for the wave's two channel blocks cA, cB, of NE environments (2: the weight fragments of a block serve both environments'
// MFMAs before the next block's replace them).  On entry wf holds W_h's gate tiles of cA (and wi W_i's, KI k-steps, when the
// input projection is computed here); on exit wf holds `next` (wi is spent).  sA / sB: the blocks' input projection rows (GI_GLOBAL).
// (Tried with two environments: block cB's input projection rows requested inside the cell, into the registers block cA's had just left --
// four GruInit at once cost spills at three agent tiles.  Slower at every shape: the rows come from HBM, and the second request wave pays
// its latency again: 0.585 -> 0.629 ms at 4096 x 40, 0.328 -> 0.362 at 4096 x 24.)
template <bool GI_GLOBAL, int KI, int NE>
__device__ __forceinline__ void gru_pair(int cA, int cB, const GruInit (&sA)[NE], const GruInit (&sB)[NE], bf16x8 (&wf)[8][3], bf16x8 (&wi)[KI][3],
                                         const uint16_t *__restrict__ Wi, const unsigned char *const (&Xi)[NE], int xirow,
                                         const uint16_t *__restrict__ Wh, const float *bsum, const unsigned char *const (&Hin)[NE],
                                         unsigned char *const (&Hout)[NE], const int *const (&upd)[NE], const int *const (&ridx)[NE], int lr, int lh,
                                         uint16_t *__restrict__ gsave, const Frag3 &next) {
    const int lane = 16 * lh + lr;
    f32x4 ar[NT], az[NT], ani[NT], anh[NT];
    TRACE_POINT(20);
    // (Tried, round 5: waves 4-7 -- the SIMD partners of 0-3 -- entering a cell 640 / 1280 / 1920 cycles late (s_sleep), and the partners at
    // different issue priorities (s_setprio), so that one's MFMA chain would run under the other's pointwise math: 0.550 -> 0.544 / 0.546 /
    // 0.554 ms per 4096 x 40 step, nothing at 18 x 192 x 40.  profiles/r05_recurrence_stagger.txt)
#pragma unroll
    for (int s = 0; s < NE; ++s) {
        gru_start<GI_GLOBAL>(sA[s], bsum, 16 * cA + 4 * lh, ar, az, ani, anh);
        if (s == NE - 1) {
            if (!GI_GLOBAL) stream_mfma<KI, true>(ar, az, ani, wi, Xi[s], xirow, lane, gate_frags(Wi, cB, KI, lane));
            stream_mfma<8, true>(ar, az, anh, wf, Hin[s], H_ROW, lane, gate_frags(Wh, cB, 8, lane));
        } else {
            if (!GI_GLOBAL) stream_mfma<KI, false, true>(ar, az, ani, wi, Xi[s], xirow, lane, next);
            stream_mfma<8, false, true>(ar, az, anh, wf, Hin[s], H_ROW, lane, next);
        }
        if (s == 0) TRACE_POINT(21);
        gru_finish(cA, ar, az, ani, anh, Hin[s], Hout[s], upd[s], ridx[s], lr, lh, gsave);
        __builtin_amdgcn_sched_barrier(0);
    }
    TRACE_POINT(22);
#pragma unroll
    for (int s = 0; s < NE; ++s) {
        gru_start<GI_GLOBAL>(sB[s], bsum, 16 * cB + 4 * lh, ar, az, ani, anh);
        if (!GI_GLOBAL) stream_mfma<KI, false, (NE > 1)>(ar, az, ani, wi, Xi[s], xirow, lane, next);
        if (s == NE - 1)
            stream_mfma<8, true>(ar, az, anh, wf, Hin[s], H_ROW, lane, next);
        else
            stream_mfma<8, false, true>(ar, az, anh, wf, Hin[s], H_ROW, lane, next);
        if (s == 0) TRACE_POINT(23);
        gru_finish(cB, ar, az, ani, anh, Hin[s], Hout[s], upd[s], ridx[s], lr, lh, gsave);
        __builtin_amdgcn_sched_barrier(0);
    }
    TRACE_POINT(24);
}

// ---- sixteen waves: the weight fragments of a k-step are requested one k-step ahead, nothing is held across phases ----
// acc_g[n] += (gate tiles f.p[g] of a packed matrix) x X for KS k-steps
#ifndef MAPF_RECUR_JIT_AHEAD
#define MAPF_RECUR_JIT_AHEAD 1  // k-steps a fragment is requested ahead of its MFMAs
#endif
template <int KS>
__device__ __forceinline__ void jit_mfma(f32x4 (&acc0)[NT], f32x4 (&acc1)[NT], f32x4 (&acc2)[NT], const Frag3 &f, const unsigned char *X, int xrow,
                                         int lane) {
    const int lr = lane & 15, lh = lane >> 4;
    constexpr int AH = MAPF_RECUR_JIT_AHEAD < KS ? MAPF_RECUR_JIT_AHEAD : KS - 1, RING = AH + 1;
    bf16x8 a[RING][3];
#pragma unroll
    for (int k0 = 0; k0 < AH; ++k0)
#pragma unroll
        for (int g = 0; g < 3; ++g) a[k0][g] = f.p[g][k0 * 64];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        if (kk + AH < KS) {
#pragma unroll
            for (int g = 0; g < 3; ++g) a[(kk + AH) % RING][g] = f.p[g][(kk + AH) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);  // (the requests stay in front of this k-step's MFMAs, and no further ahead)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(X + (16 * n + lr) * xrow + (32 * kk + 8 * lh) * 2);
            acc0[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk % RING][0], b, acc0[n], 0, 0, 0);
            acc1[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk % RING][1], b, acc1[n], 0, 0, 0);
            acc2[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk % RING][2], b, acc2[n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the cell for this wave's ONE channel block
template <bool GI_GLOBAL, int KI>
__device__ __forceinline__ void gru_one(int cA, const GruInit &sA, const uint16_t *__restrict__ Wi, const unsigned char *Xi, int xirow,
                                        const uint16_t *__restrict__ Wh, const float *bsum, const unsigned char *Hin, unsigned char *Hout,
                                        const int *upd, const int *ridx, int lr, int lh, uint16_t *__restrict__ gsave) {
    const int lane = 16 * lh + lr;
    f32x4 ar[NT], az[NT], ani[NT], anh[NT];
    TRACE_POINT(20);
    gru_start<GI_GLOBAL>(sA, bsum, 16 * cA + 4 * lh, ar, az, ani, anh);
    if (!GI_GLOBAL) jit_mfma<KI>(ar, az, ani, gate_frags(Wi, cA, KI, lane), Xi, xirow, lane);
    jit_mfma<8>(ar, az, anh, gate_frags(Wh, cA, 8, lane), Hin, H_ROW, lane);
    TRACE_POINT(21);
    gru_finish(cA, ar, az, ani, anh, Hin, Hout, upd, ridx, lr, lh, gsave);
    TRACE_POINT(24);
}

// Barrier between two phases of a step.  The weight loads are loads from read-only, non-aliased memory: the instruction scheduler moves
// them up across s_barrier (and then spills what it fetched early) unless the barrier is also a scheduling boundary.
__device__ __forceinline__ void phase_sync(int id = 0) {
    __builtin_amdgcn_sched_barrier(0);
    TRACE_POINT(id);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    TRACE_POINT(id + 100);
}

template <bool SAVE, bool PAIR>
__global__ void __launch_bounds__(NTHR, 1) recurrent_infer_kernel(const uint16_t *__restrict__ gi, const uint16_t *__restrict__ h0,
                                                                  const uint8_t *__restrict__ comm, const uint16_t *__restrict__ W,
                                                                  const float *__restrict__ bias, int T, int E, int N_arg,
                                                                  uint16_t *__restrict__ h_out, uint16_t *__restrict__ agent0_out,
                                                                  RecurSave sv, const int32_t *__restrict__ rowidx, long long nrows,
                                                                  const int4 *__restrict__ envtab, int a0s) {
    // NE environments side by side in this workgroup (PAIR: two, see the LDS map; never with SAVE, envtab or sixteen waves)
    constexpr int NE = PAIR ? 2 : 1;
    static_assert(!(PAIR && SAVE) && !(PAIR && NWV != 8), "");
    __shared__ __attribute__((aligned(16))) unsigned char smem[PAIR ? P_LDS_BYTES : LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lh = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction; said so, its tiles' addresses stay in scalar registers
    // Persistent over environments (round 5): workgroup b steps environments b, b + gridDim.x, ... (PAIR: the pairs 2 b | 2 b + 1,
    // 2 (b + gridDim.x) | ...): the LDS fill, the bias sums and the
    // first weight fragments are paid once per workgroup instead of once per environment, and a CU never waits for a new workgroup to
    // be placed.  (envtab launches keep one environment per workgroup: their agent counts differ.)
    int e = NE * blockIdx.x;
    // envtab (mapf_recurrent_infer_multi: one step of environments of DIFFERENT agent counts -- the curriculum's levels -- in one
    // launch): per environment {agents, first row of its agents in gi / h0 / h_out, byte offset of its mask in comm, -}
    int n_env = N_arg, n_real = N_arg;
    long long hrow0 = (long long)e * N_arg, coff = 0;
    if (!PAIR && envtab != nullptr) {
        const int4 d = envtab[e];
        n_env = __builtin_amdgcn_readfirstlane(d.x);
        hrow0 = __builtin_amdgcn_readfirstlane(d.y);
        coff = __builtin_amdgcn_readfirstlane(d.z);
        n_real = __builtin_amdgcn_readfirstlane(d.w);
        if (n_real <= 0) n_real = n_env;
    }
    const int N = n_env;   // agent rows this workgroup steps (per environment)
    const int NR = n_real;  // ... of which every NR consecutive ones are one environment (their masks [NR][NR] back to back): see below
    // agents of environment e + s: N, or 0 when E is odd and the last pair has no second member (all of its rows are then padding rows:
    // computed like real agents, never loaded or stored)
    auto agents = [&](int s) { return (!PAIR || e + s < E) ? N : 0; };
    const int rot = blockIdx.x;  // workgroups walk the weight tiles in rotated order: they run in step, and would otherwise all
                                 // request the same cache lines at the same moment

    TRACE_BEGIN();
    TRACE_POINT(30);
    // the weight stream (see stream_mfma): this wave's channel blocks / q|k|v tiles are the same in every phase of every step
    const int cA = (w + rot) & 15, cB = (w + 8 + rot) & 15, wq = (w + rot) & 7;
#if MAPF_RECUR_WAVES == 8
    bf16x8 wf[8][3], wi[2][3];
    load_frags<8>(wf, gate_frags(W + W_HH, cA, 8, lane));  // first job: the recurrent cell's block cA; under way during the prologue
#endif

    // What a step reads from global memory besides the weights: its mask bytes and the recurrent cell's input projection rows of this
    // lane's agents, for both channel blocks (from HBM -- 60 KB per 40-agent environment, by all workgroups at about the same time: a
    // one-step launch waited 18 k cycles for them behind its prologue).  Requested before anything else of the step; step 0's before
    // the prologue.  The mask bytes first: vmcnt retires in order, requested behind the input projection they would wait for it.
    constexpr int CI = (NA * NA + NTHR - 1) / NTHR;
    uint8_t cbyte[NE][CI];
    GruInit sA[NE], sB[NE];
    auto fetch_inputs = [&](int t) {
#pragma unroll
        for (int s = 0; s < NE; ++s) {
            const int Ns = agents(s);
            const uint8_t *comm_t = comm + (envtab ? coff : ((long long)t * E + e + s) * N * N);
#pragma unroll
            for (int q = 0; q < CI; ++q) {
                const int idx = tid + q * NTHR;
                cbyte[s][q] = idx < Ns * NR ? comm_t[idx] : (uint8_t)0;
            }
        }
#pragma unroll
        for (int s = 0; s < NE; ++s) {
            const int Ns = agents(s);
            const long long row0 = envtab ? hrow0 : ((long long)t * E + e + s) * N;
            int grow[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int a = 16 * n + lr;
                grow[n] = a < Ns ? (rowidx ? rowidx[row0 + a] : (int)(row0 + a)) : -1;
            }
            if (!(MAPF_RECUR_ABLATE & 1)) {
                gru_fetch_gi(sA[s], cA, gi, grow, lh);
                if (NWV == 8) gru_fetch_gi(sB[s], cB, gi, grow, lh);
            }
        }
    };
    fetch_inputs(0);
    // initial hidden rows of an environment: 32 chunks of 16 B per agent, HP per thread, through registers into the LDS image
    constexpr int HP = (NA * 32 + NTHR - 1) / NTHR;
    uint4 hpre[NE][HP];
    auto fetch_h0 = [&]() {
#pragma unroll
        for (int s = 0; s < NE; ++s)
#pragma unroll
            for (int q = 0; q < HP; ++q) {
                const int i = tid + q * NTHR, a = i >> 5, ch = i & 31;
                hpre[s][q] = make_uint4(0, 0, 0, 0);
                if (h0 != nullptr && a < agents(s)) hpre[s][q] = *reinterpret_cast<const uint4 *>(h0 + (hrow0 + (long long)s * N + a) * D + ch * 8);
            }
    };
    // LDS images of environment s (see the maps above)
    auto envb = [&](int s) -> unsigned char * { return smem + (PAIR ? s * P_ENV : 0); };
    unsigned char *Hc_[NE], *Hn_[NE];
#pragma unroll
    for (int s = 0; s < NE; ++s) Hc_[s] = envb(s), Hn_[s] = envb(s) + H_BYTES;
    auto Hcur = [&](int s) -> unsigned char * { return Hc_[s]; };
    auto Hnxt = [&](int s) -> unsigned char * { return Hn_[s]; };
    auto swap_states = [&]() {
#pragma unroll
        for (int s = 0; s < NE; ++s) {
            unsigned char *tmp = Hc_[s];
            Hc_[s] = Hn_[s];
            Hn_[s] = tmp;
        }
    };
    auto QKp = [&](int s) -> unsigned char * { return PAIR ? Hnxt(s) : smem + OFF_QK; };
    auto VTp = [&](int s) -> unsigned char * { return PAIR ? envb(s) + P_OFF_VT : smem + OFF_VT; };
    auto CTXp = [&](int s) -> unsigned char * { return PAIR ? Hnxt(s) : smem + OFF_CTX; };
    auto INFOp = [&](int s) -> unsigned char * { return PAIR ? envb(s) + P_OFF_VT : smem + OFF_INFO; };
    constexpr int CTX_ROW_ = PAIR ? QK_ROW : CTX_ROW;
    auto updp = [&](int s) -> int * { return reinterpret_cast<int *>(smem + (PAIR ? P_OFF_UPD + s * 256 : OFF_UPD)); };
    auto mbp = [&](int s) -> uint32_t * { return reinterpret_cast<uint32_t *>(smem + (PAIR ? P_OFF_MB + s * NA * 8 : OFF_MB)); };
    auto ridxp = [&](int s) -> int * { return reinterpret_cast<int *>(smem + (PAIR ? P_OFF_RIDX + s * NA * 4 : OFF_RIDX)); };
    auto store_h0 = [&]() {  // rows >= N keep what they hold (zeros, or the padding rows' own bounded trajectory)
#pragma unroll
        for (int s = 0; s < NE; ++s)
#pragma unroll
            for (int q = 0; q < HP; ++q) {
                const int i = tid + q * NTHR, a = i >> 5, ch = i & 31;
                if (a < agents(s)) *reinterpret_cast<uint4 *>(Hcur(s) + a * H_ROW + ch * 16) = hpre[s][q];
            }
    };
    fetch_h0();
    // (Tried and dropped, round 5: pulling the NEXT environment's input lines into L2 ahead of time -- one dword per 128-byte line,
    // either by every wave in front of the last update cell, or by waves 6-7, idle in the attention phase.  The first made the cell's
    // own weight reloads wait for HBM (vmcnt retires in order): 0.62 -> 0.68 ms per 4096 x 40 step; the second cost 200 more spilled
    // registers in a kernel that sits at its 256-register ceiling: 0.93 ms.  Holding the rows themselves in registers across the cell:
    // 300 spills.  Workgroups starting spread over a 15 .. 60 us window, so that their per-environment input bursts do not coincide:
    // -1 .. -2 % at 40 agents, +2 .. +15 % at 24 / 16 / 6.  What is kept: the persistent loop, and the initial hidden rows through
    // registers in front of the LDS fill -- 0.61 -> 0.55 ms.  profiles/r05_recurrence_ab_persistent.txt)

    // hidden state of this environment (rows >= N stay zero: they are computed like real agents and never stored)
    for (int i = tid; i < (PAIR ? P_LDS_BYTES : LDS_BYTES) / 16; i += NTHR) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    float *bsum = reinterpret_cast<float *>(smem + (PAIR ? P_OFF_BSUM : OFF_BSUM));
    for (int i = tid; i < 2 * 256; i += NTHR) {  // same sums, same order as the accumulators used to start from
        const int cell = i >> 8, c = i & 255;
        const float *bi = bias + (cell ? UB_IH : B_IH), *bh = bias + (cell ? UB_HH : B_HH);
        float *d = bsum + cell * 1024 + c;
        d[0] = bi[c] + bh[c];
        d[256] = bi[256 + c] + bh[256 + c];
        d[512] = bi[512 + c];
        d[768] = bh[512 + c];
    }
    store_h0();
    __syncthreads();

    TRACE_POINT(31);
    const float scale = 0.125f;  // 1 / sqrt(64)

    // Rows of gi and of the saved tensors: dense (rowidx == nullptr) -- agent a of (step t, environment e) is row (t E + e) N + a of
    // T E N rows -- or compact: rowidx [T][E][N] names the row of every entry that matters among `nrows` rows, -1 for the others
    // (include/mapf_dqn.h: mapf_plan_rows' gidx; such an agent gets no input projection and nothing of it is saved).
    const long long RTOT = rowidx ? nrows : (long long)T * E * N;
    auto save_hidden = [&](uint16_t *dst, const unsigned char *H) {  // the step's rows of 256 bf16 from an LDS hidden image (SAVE: one environment)
        const int *ridx = ridxp(0);
        for (int i = tid; i < N * 32; i += NTHR) {
            const int a = i >> 5, ch = i & 31, row = ridx[a];
            if (row >= 0) *reinterpret_cast<uint4 *>(dst + (long long)row * D + ch * 8) = *reinterpret_cast<const uint4 *>(H + a * H_ROW + ch * 16);
        }
    };
    const uint16_t *const W_arg = W;
    const float *const bias_arg = bias;
    const int G = gridDim.x;
    for (;;) {  // environments of this workgroup
    for (int t = 0;;) {  // (T >= 1)
        // The weight and bias addresses of a wave are the same at every step; opaque copies of the base pointers keep the compiler from
        // hoisting those loads out of the step loop (it did, once the cells' channel blocks became loop-invariant: 430 spilled registers).
        int opaque0 = 0;  // (an opaque OFFSET: through the asm the pointers themselves would lose their address space -> flat loads)
        asm volatile("" : "+s"(opaque0));
        const uint16_t *__restrict__ W = W_arg + opaque0;
        const float *__restrict__ bias = bias_arg + opaque0;
        // ---------------- this step's communication mask -> bit rows in LDS (the softmax loops must not touch global memory:
        // 120 dependent byte loads per row made the first version 10x slower than its MFMAs) ----------------
#pragma unroll
        for (int s = 0; s < NE; ++s) {
            const long long row0 = envtab ? hrow0 : ((long long)t * E + e + s) * N;  // first dense row of this (step, environment)
            if (tid < NA * 2) mbp(s)[tid] = 0u;
            if (tid < NA) ridxp(s)[tid] = tid < agents(s) ? (rowidx ? rowidx[row0 + tid] : (int)(row0 + tid)) : -1;
        }
        phase_sync(1);
        if (SAVE) save_hidden(sv.hin0, Hcur(0));
        // (mapf_recurrent_infer_multi packs several small environments of one level into one workgroup -- their agents' rows are
        // consecutive everywhere, and the weights, which is what a step streams, are fetched once for all of them: row i = agent i % NR
        // of environment i / NR reads that environment's mask row, its partners are the columns of that environment: a block-diagonal
        // mask.  NR == N: one environment, the plain case.)
#pragma unroll
        for (int s = 0; s < NE; ++s) {
            uint32_t *mb = mbp(s);
#pragma unroll
            for (int q = 0; q < CI; ++q)
                if (cbyte[s][q] != 0) {
                    const int idx = tid + q * NTHR, i = idx / NR, j = (i / NR) * NR + (idx - i * NR);
                    atomicOr(&mb[2 * i + (j >> 5)], 1u << (j & 31));
                }
        }
        // ---------------- recurrent GRU cell: current -> next state (the barrier behind it also publishes the mask bits) ----------------
        const unsigned char *Hin[NE], *Xi[NE];
        unsigned char *Hout[NE];
        const int *updv[NE], *noupd[NE], *ridxv[NE];
#pragma unroll
        for (int s = 0; s < NE; ++s) {
            Hin[s] = Hcur(s);
            Hout[s] = Hnxt(s);
            Xi[s] = nullptr;
            noupd[s] = nullptr;
            updv[s] = updp(s);
            ridxv[s] = ridxp(s);
        }
#if MAPF_RECUR_WAVES == 8
        if (!(MAPF_RECUR_ABLATE & 1))
            gru_pair<true, 2, NE>(cA, cB, sA, sB, wf, wi, nullptr, Xi, 0, W + W_HH, bsum, Hin, Hout, noupd, ridxv, lr, lh, SAVE ? sv.g1 : nullptr,
                                  frag3(W + W_QKV, wq, wq + 8, wq + 16, 8, lane));
#else
        if (!(MAPF_RECUR_ABLATE & 1))
            gru_one<true, 2>(cA, sA[0], nullptr, nullptr, 0, W + W_HH, bsum, Hin[0], Hout[0], nullptr, ridxv[0], lr, lh, SAVE ? sv.g1 : nullptr);
#endif
        phase_sync(2);
        swap_states();
        // ---------------- two communication rounds (shared weights): current -> next -> swap ----------------
        for (int round = 0; round < 2; ++round) {
            asm volatile("" : "+s"(opaque0));  // (both rounds read the same weights: without this the second round's loads are "the first round's
            W = W_arg + opaque0;               //  values", kept in registers across the whole round)
            bias = bias_arg + opaque0;
            if (SAVE) save_hidden(sv.hr + (long long)round * RTOT * D, Hcur(0));
            // q | k | v = W_qkv h + b: 24 output tiles of 16
            auto store_qkv = [&](int s, int tile, const f32x4 (&acc)[NT]) {  // tile 0..7 q, 8..15 k, 16..23 v (16 channels each)
                const int g = tile >> 3, c0 = 16 * (tile & 7) + 4 * lh;
                unsigned char *QK = QKp(s), *VT = VTp(s);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int agent = 16 * n + lr;
                    const uint32_t p01 = pack2_bf16(acc[n][0], acc[n][1]), p23 = pack2_bf16(acc[n][2], acc[n][3]);
                    if (g < 2) {  // q (channels 0..127) and k (128..255): [agent][channel]
                        *reinterpret_cast<uint2 *>(QK + agent * QK_ROW + (128 * g + c0) * 2) = make_uint2(p01, p23);
                    } else {      // v transposed: row (head, d) = c0 + r, column = agent
                        uint16_t *vt = reinterpret_cast<uint16_t *>(VT + c0 * VT_ROW) + agent;
                        vt[0] = (uint16_t)(p01 & 0xFFFFu);
                        vt[VT_ROW / 2] = (uint16_t)(p01 >> 16);
                        vt[2 * (VT_ROW / 2)] = (uint16_t)(p23 & 0xFFFFu);
                        vt[3 * (VT_ROW / 2)] = (uint16_t)(p23 >> 16);
                    }
                    if (SAVE && ridxp(0)[agent] >= 0)
                        *reinterpret_cast<uint2 *>(sv.qkv + ((long long)round * RTOT + ridxp(0)[agent]) * 384 + 128 * g + c0) = make_uint2(p01, p23);
                }
            };
            if (PAIR) {  // the previous round's info lay over the first NA rows of v^T: their agent slots >= NA are zero again before P meets them
                constexpr int ZW = (128 - 2 * NA) / 16;  // 16-byte words per row
                for (int i = tid; i < NE * NA * ZW; i += NTHR) {
                    const int s = i / (NA * ZW), r = (i - s * NA * ZW) / ZW, q = i - (s * NA + r) * ZW;
                    *reinterpret_cast<uint4 *>(VTp(s) + r * VT_ROW + 2 * NA + 16 * q) = make_uint4(0, 0, 0, 0);
                }
            }
#if MAPF_RECUR_WAVES == 8
            if (!(MAPF_RECUR_ABLATE & 2)) {  // (one job per wave and environment)
#pragma unroll
                for (int s = 0; s < NE; ++s) {
                    f32x4 acc[3][NT];  // tiles wq (q), wq + 8 (k), wq + 16 (v)
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const float4 b4 = *reinterpret_cast<const float4 *>(bias + B_QKV + 16 * (wq + 8 * g) + 4 * lh);
#pragma unroll
                        for (int n = 0; n < NT; ++n) acc[g][n] = f32x4{b4.x, b4.y, b4.z, b4.w};
                    }
                    if (s == NE - 1) {
                        stream_mfma<8, true>(acc[0], acc[1], acc[2], wf, Hcur(s), H_ROW, lane, gate_frags(W + U_HH, cA, 8, lane));  // next: the update cell
                        load_frags<2>(wi, gate_frags(W + U_IH, cA, 2, lane));
                    } else {
                        stream_mfma<8, false, true>(acc[0], acc[1], acc[2], wf, Hcur(s), H_ROW, lane, gate_frags(W + U_HH, cA, 8, lane));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    store_qkv(s, wq, acc[0]);
                    store_qkv(s, wq + 8, acc[1]);
                    store_qkv(s, wq + 16, acc[2]);
                    if (s != NE - 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
#else
            for (int j = w; j < 24 && !(MAPF_RECUR_ABLATE & 2); j += NWV) {  // (waves 0..7: two tiles, 8..15: one -- two of each kind per SIMD)
                const int tile = (j + rot) % 24;
                f32x4 acc[NT];
                const float4 b4 = *reinterpret_cast<const float4 *>(bias + B_QKV + 16 * tile + 4 * lh);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = f32x4{b4.x, b4.y, b4.z, b4.w};
                gemm16<8>(acc, W + W_QKV, tile, Hcur(0), H_ROW, lane);
                store_qkv(0, tile, acc);
            }
#endif
            phase_sync(3);
            // Attention of one (head, agent tile) per wave, entirely in registers.  S^T = k q^T: an accumulator lane then holds, for ITS
            // agent i = 16 ti + lr, the scores of the partners j = 16 tj + 4 lh + r -- which is, tile pair by tile pair, the B operand
            // (32 k-slots x 16 agents) of ctx^T = v^T P^T once the k-slots are numbered  slot 8 lh + u  <->  j = 16 (2 kk + (u >> 2)) +
            // 4 lh + (u & 3); v^T's A operand reads its columns in the same numbering (two 8-byte reads).  Rounds 1-4 went through a
            // score image and a softmax image in LDS with a barrier behind each: 4.1 k of a step's ~30 k cycles per round at 40 agents.
            // (at one or two agent tiles there are only 2 / 4 (head, tile) pairs for 8 waves: two waves then share a pair, each repeating the
            //  scores and the softmax and taking two of the four d-tiles of ctx -- SPLIT; not with two environments, whose ctx replaces q)
            constexpr int SPLIT = (NT <= 2 && !PAIR) ? 2 : 1;
            constexpr int JOBS1 = 2 * NT * SPLIT;  // per environment
            for (int job = w; job < NE * JOBS1 && !(MAPF_RECUR_ABLATE & 4); job += NTHR / 64) {
                const int s = NE == 1 ? 0 : job / JOBS1, job1 = job - s * JOBS1;
                const int half = job1 / (2 * NT), pair = job1 - 2 * NT * half;
                const int hd = pair / NT, ti = pair - NT * hd, i = 16 * ti + lr;
                const int Ns = agents(s);
                const unsigned char *QK = QKp(s), *VT = VTp(s);
                unsigned char *CTX = CTXp(s);
                const uint32_t *mb = mbp(s);
                f32x4 sc[NT];
                {
                    bf16x8 qf[2];
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) qf[kk] = *reinterpret_cast<const bf16x8 *>(QK + i * QK_ROW + (hd * HD + 32 * kk + 8 * lh) * 2);
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        sc[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(QK + (16 * tj + lr) * QK_ROW + (128 + hd * HD + 32 * kk + 8 * lh) * 2);
                            sc[tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], sc[tj], 0, 0, 0);
                        }
                    }
                }
                // masked softmax over the row (model.py:75-78): columns without a mask bit (incl. j >= N) read -1e9; rows i >= N come out 0
                const uint64_t bits = i < Ns ? ((uint64_t)mb[2 * i] | ((uint64_t)mb[2 * i + 1] << 32)) : 0ull;  // bits >= N are 0
                float mx = -3.0e38f;
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    const uint32_t nib = (uint32_t)(bits >> (16 * tj + 4 * lh));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // as a bit select, not a predicate: compile-time lane masks would live in scalar registers and spill
                        const uint32_t m = 0u - ((nib >> r) & 1u);
                        sc[tj][r] = __uint_as_float((__float_as_uint(sc[tj][r] * scale) & m) | (__float_as_uint(-1e9f) & ~m));
                        mx = fmaxf(mx, sc[tj][r]);
                    }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float sum = 0.f;
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sc[tj][r] = __expf(sc[tj][r] - mx);  // masked and padded columns: exp(-1e9 - mx) == 0 (a row always holds its own agent)
                        sum += sc[tj][r];
                    }
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                const float inv = i < Ns ? 1.f / sum : 0.f;
                uint2 pw[4];
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
                    pw[tj] = tj < NT ? make_uint2(pack2_bf16(sc[tj < NT ? tj : 0][0] * inv, sc[tj < NT ? tj : 0][1] * inv),
                                                  pack2_bf16(sc[tj < NT ? tj : 0][2] * inv, sc[tj < NT ? tj : 0][3] * inv))
                                     : make_uint2(0u, 0u);
                if (hd == 0 && lh == 0 && half == 0) updp(s)[i] = (i < Ns && __popcll(bits) > 1) ? 1 : 0;  // model.py:103
                if (SAVE && half == 0) {  // P rows (2 heads x NA agents x 64 slots = 128 B each; slots >= NA zero) -> global
                    uint16_t *pd = sv.P + ((((long long)round * T + t) * E + e) * 2 + hd) * (NA * 64) + i * 64 + 4 * lh;
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj) *reinterpret_cast<uint2 *>(pd + 16 * tj) = pw[tj];
                }
                // ctx^T[d][i] = sum_j vT[d][j] P[i][j]: 4 d-tiles of this head -> CTX[agent][head*64 + d]
#pragma unroll
                for (int tq = 0; tq < 4 / SPLIT; ++tq) {
                    const int td = (4 / SPLIT) * half + tq;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    const unsigned char *vr = VT + (hd * HD + 16 * td + lr) * VT_ROW + 8 * lh;
#pragma unroll
                    for (int kk = 0; kk < (NT + 1) / 2; ++kk) {
                        const uint2 alo = *reinterpret_cast<const uint2 *>(vr + 64 * kk), ahi = *reinterpret_cast<const uint2 *>(vr + 64 * kk + 32);
                        const bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(alo.x, alo.y, ahi.x, ahi.y));
                        const bf16x8 b = __builtin_bit_cast(bf16x8, make_uint4(pw[2 * kk].x, pw[2 * kk].y, pw[2 * kk + 1].x, pw[2 * kk + 1].y));
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
                    }
                    const uint2 v = make_uint2(pack2_bf16(acc[0], acc[1]), pack2_bf16(acc[2], acc[3]));
                    *reinterpret_cast<uint2 *>(CTX + i * CTX_ROW_ + (hd * HD + 16 * td + 4 * lh) * 2) = v;
                    if (SAVE && ridxp(0)[i] >= 0)
                        *reinterpret_cast<uint2 *>(sv.ctx + ((long long)round * RTOT + ridxp(0)[i]) * 128 + hd * HD + 16 * td + 4 * lh) = v;
                }
            }
            phase_sync(6);
            // info = W_O ctx (no bias): 4 output tiles per environment, K = 128
            for (int j = w; j < 4 * NE && !(MAPF_RECUR_ABLATE & 8); j += NTHR / 64) {
                const int s = NE == 1 ? 0 : j >> 2, ot = NE == 1 ? j : j & 3;
                f32x4 acc[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm16<4>(acc, W + W_O, ot, CTXp(s), CTX_ROW_, lane);
                unsigned char *INFO = INFOp(s);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const uint2 v = make_uint2(pack2_bf16(acc[n][0], acc[n][1]), pack2_bf16(acc[n][2], acc[n][3]));
                    *reinterpret_cast<uint2 *>(INFO + (16 * n + lr) * INFO_ROW + (16 * ot + 4 * lh) * 2) = v;
                    if (SAVE && ridxp(0)[16 * n + lr] >= 0)
                        *reinterpret_cast<uint2 *>(sv.info + ((long long)round * RTOT + ridxp(0)[16 * n + lr]) * 64 + 16 * ot + 4 * lh) = v;
                }
            }
            phase_sync(7);
            // update cell: current -> next state where the agent has a partner
#pragma unroll
            for (int s = 0; s < NE; ++s) {
                Hin[s] = Hcur(s);
                Hout[s] = Hnxt(s);
                Xi[s] = INFOp(s);
            }
#if MAPF_RECUR_WAVES == 8
            if (!(MAPF_RECUR_ABLATE & 16))  // behind it: the second round's q|k|v, or the next step's recurrent cell (a wasted fetch at the last step of the last environment)
                gru_pair<false, 2, NE>(cA, cB, sA, sB, wf, wi, W + U_IH, Xi, INFO_ROW, W + U_HH, bsum + 1024, Hin, Hout, updv, ridxv, lr, lh,
                                       SAVE ? sv.g2 + (long long)round * RTOT * 1024 : nullptr,
                                       round == 0 ? frag3(W + W_QKV, wq, wq + 8, wq + 16, 8, lane) : gate_frags(W + W_HH, cA, 8, lane));
#else
            if (!(MAPF_RECUR_ABLATE & 16))
                gru_one<false, 2>(cA, sA[0], W + U_IH, Xi[0], INFO_ROW, W + U_HH, bsum + 1024, Hin[0], Hout[0], updv[0], ridxv[0], lr, lh,
                                  SAVE ? sv.g2 + (long long)round * RTOT * 1024 : nullptr);
#endif
            phase_sync(8);
            swap_states();
        }
        // agent 0's state after this step (model.py:248).  a0s > 0: the environment is a TILE of several windows of a0s agent rows each
        // (block-diagonal masks: include/mapf_dqn.h, mapf_recurrent_infer_packed) -- rows 0, a0s, 2 a0s, ... are their agents 0
        {
            const int K = a0s > 0 ? (N + a0s - 1) / a0s : 1;
#pragma unroll
            for (int s = 0; s < NE; ++s)
                if (agent0_out != nullptr && tid < 32 * K && agents(s) > 0) {
                    const int k = tid >> 5, c = tid & 31;
                    *reinterpret_cast<uint4 *>(agent0_out + (((long long)t * E + e + s) * K + k) * D + c * 8) =
                        *reinterpret_cast<const uint4 *>(Hcur(s) + (k * a0s) * H_ROW + c * 16);
                }
        }
        if (++t >= T) break;
        fetch_inputs(t);  // the next step's (step 0's were requested in front of the prologue)
    }
#pragma unroll
    for (int s = 0; s < NE; ++s)
        for (int i = tid; i < agents(s) * 32; i += NTHR) {
            const int a = i >> 5, ch = i & 31;
            *reinterpret_cast<uint4 *>(h_out + (hrow0 + (long long)s * N + a) * D + ch * 8) = *reinterpret_cast<const uint4 *>(Hcur(s) + a * H_ROW + ch * 16);
        }
    if (envtab != nullptr || e + NE * G >= E) break;
    e += NE * G;
    hrow0 = (long long)e * N_arg;
    fetch_inputs(0);
    fetch_h0();
    __syncthreads();  // every row of the finished environment is on its way out before the next one's rows replace it
    store_h0();       // (the first barrier of the step publishes them)
    }
    TRACE_POINT(32);
    TRACE_END();
}

// One workgroup per CU (its LDS image and 8 waves of 256 registers fill one), each walking its share of the environments.
// MAPF_RECUR_PERSIST=0 (A/B runs): one workgroup per environment, as in rounds 1-4.
inline int persistent_grid(int E) {
    static const int cus = [] {
        const char *v = std::getenv("MAPF_RECUR_PERSIST");
        if (v != nullptr && v[0] == '0') return 0;
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return n;
    }();
    return (cus > 0 && E > cus) ? cus : E;
}
// Two environments per workgroup (recurrent_infer_kernel<false, true>): builds for one and two agent tiles only, and only when there are
// more environments than CUs -- below that a launch lasts as long as ONE workgroup's steps, which a second environment lengthens (a pair's
// step takes ~1.4 x an environment's).  At three agent tiles the pair kernel is slower than two single steps (0.585 against 0.535 ms at
// 4096 x 40: 128 spilled registers, and its cells are bound by the LDS reads of the activations, which pairing does not share; see
// HISTORY.md).  MAPF_RECUR_PAIR=0 (A/B runs): never.
constexpr bool PAIR_BUILT = NT <= 2 && NWV == 8;
inline int recur_cus() { return persistent_grid(1 << 30); }  // CU count, or 0 with MAPF_RECUR_PERSIST=0 / unknown
inline bool pair_launch(int E) {
    static const int on = [] {
        const char *v = std::getenv("MAPF_RECUR_PAIR");
        return (v != nullptr && v[0] == '0') ? 0 : 1;
    }();
    const int cus = recur_cus();
    return PAIR_BUILT && on && cus > 0 && cus < (1 << 30) && E > cus;
}
inline int pair_grid(int E) {
    const int cus = recur_cus(), pairs = (E + 1) / 2;
    return pairs < cus ? pairs : cus;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            std::fprintf(stderr, "mapf_recur: %s failed: %s\n", #expr, hipGetErrorString(_e)); \
            return MAPF_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

}  // namespace

#if MAPF_RECUR_NT == 3
extern "C" {
#define RECUR_ENTRY(name) name
#else  // the <= 16- / <= 32-agent builds: internal symbols, reached from the entry points of the 48-agent build (mapf_recur_internal.h)
#define RECUR_PASTE2(a, b) a##b
#define RECUR_PASTE(a, b) RECUR_PASTE2(a, b)
#define RECUR_ENTRY(name) __attribute__((visibility("hidden"))) RECUR_PASTE(name, MAPF_RECUR_SUFFIX)
#endif

int RECUR_ENTRY(mapf_recurrent_infer_packed)(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev, const uint16_t *weights_dev,
                         const float *bias_dev, int T, int E, int N, uint16_t *h_out_dev, uint16_t *agent0_out_dev, const int32_t *row_index_dev,
                         int64_t num_rows, int agent0_stride, void *stream) {
    if (T < 1 || E < 0 || N < 1 || N > MAPF_RECUR_MAX_AGENTS || !gi_dev || !comm_dev || !weights_dev || !bias_dev || !h_out_dev)
        return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(gi_dev) & 7) || (reinterpret_cast<uintptr_t>(h0_dev) & 15) || (reinterpret_cast<uintptr_t>(weights_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(bias_dev) & 15) || (reinterpret_cast<uintptr_t>(h_out_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(agent0_out_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    if (agent0_stride < 0 || agent0_stride > N) return MAPF_ERR_INVALID_ARG;
    if (E == 0) return MAPF_OK;
    if (row_index_dev && (N > NA || num_rows < 1)) return MAPF_ERR_UNSUPPORTED;  // compact rows: the <= 48-agent kernels only
    if (agent0_stride > 0 && N > MAPF_RECUR_NARROW_AGENTS) return MAPF_ERR_UNSUPPORTED;  // tiles of several windows: the <= 48-agent kernels only
#if MAPF_RECUR_NT == 3
    if (N <= MAPF_RECUR_SMALL_AGENTS)  // one agent tile: the same kernel built for 16 agents (a third of the MFMA / LDS work per step)
        return mapf_recurrent_infer_packed_nt1(gi_dev, h0_dev, comm_dev, weights_dev, bias_dev, T, E, N, h_out_dev, agent0_out_dev, row_index_dev, num_rows,
                                               agent0_stride, stream);
    if (N <= 2 * MAPF_RECUR_SMALL_AGENTS)  // two tiles
        return mapf_recurrent_infer_packed_nt2(gi_dev, h0_dev, comm_dev, weights_dev, bias_dev, T, E, N, h_out_dev, agent0_out_dev, row_index_dev, num_rows,
                                               agent0_stride, stream);
#endif
    if (N > NA)  // 49..128 agents: csrc/mapf_recur_wide.hip
        return mapf_recur_wide_forward(gi_dev, h0_dev, comm_dev, weights_dev, bias_dev, T, E, N, h_out_dev, agent0_out_dev, nullptr,
                                       static_cast<hipStream_t>(stream));
#if MAPF_RECUR_WAVES == 8 && MAPF_RECUR_NT <= 2
    if (pair_launch(E))
        hipLaunchKernelGGL((recurrent_infer_kernel<false, true>), dim3(pair_grid(E)), dim3(NTHR), 0, static_cast<hipStream_t>(stream), gi_dev, h0_dev,
                           comm_dev, weights_dev, bias_dev, T, E, N, h_out_dev, agent0_out_dev, RecurSave{}, row_index_dev, (long long)num_rows,
                           (const int4 *)nullptr, agent0_stride);
    else
#endif
        hipLaunchKernelGGL((recurrent_infer_kernel<false, false>), dim3(persistent_grid(E)), dim3(NTHR), 0, static_cast<hipStream_t>(stream), gi_dev, h0_dev,
                           comm_dev, weights_dev, bias_dev, T, E, N, h_out_dev, agent0_out_dev, RecurSave{}, row_index_dev, (long long)num_rows,
                           (const int4 *)nullptr, agent0_stride);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int RECUR_ENTRY(mapf_recurrent_forward_save_packed)(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev, const uint16_t *weights_dev,
                                const float *bias_dev, int T, int E, int N, uint16_t *h_out_dev, uint16_t *agent0_out_dev,
                                uint16_t *const *save_dev, const int32_t *row_index_dev, int64_t num_rows, int agent0_stride, void *stream) {
    if (T < 1 || E < 0 || N < 1 || N > MAPF_RECUR_MAX_AGENTS || !gi_dev || !comm_dev || !weights_dev || !bias_dev || !h_out_dev ||
        !agent0_out_dev || !save_dev)
        return MAPF_ERR_INVALID_ARG;
    for (int i = 0; i < 8; ++i)
        if (!save_dev[i] || (reinterpret_cast<uintptr_t>(save_dev[i]) & 15)) return MAPF_ERR_INVALID_ARG;
    if (agent0_stride < 0 || agent0_stride > N) return MAPF_ERR_INVALID_ARG;
    if (E == 0) return MAPF_OK;
    const RecurSave sv{save_dev[0], save_dev[1], save_dev[2], save_dev[3], save_dev[4], save_dev[5], save_dev[6], save_dev[7]};
    if (row_index_dev && (N > NA || num_rows < 1)) return MAPF_ERR_UNSUPPORTED;
    if (agent0_stride > 0 && N > MAPF_RECUR_NARROW_AGENTS) return MAPF_ERR_UNSUPPORTED;
#if MAPF_RECUR_NT == 3
    if (N <= MAPF_RECUR_SMALL_AGENTS)
        return mapf_recurrent_forward_save_packed_nt1(gi_dev, h0_dev, comm_dev, weights_dev, bias_dev, T, E, N, h_out_dev, agent0_out_dev, save_dev,
                                                      row_index_dev, num_rows, agent0_stride, stream);
    if (N <= 2 * MAPF_RECUR_SMALL_AGENTS)
        return mapf_recurrent_forward_save_packed_nt2(gi_dev, h0_dev, comm_dev, weights_dev, bias_dev, T, E, N, h_out_dev, agent0_out_dev, save_dev,
                                                      row_index_dev, num_rows, agent0_stride, stream);
#endif
    if (N > NA)
        return mapf_recur_wide_forward(gi_dev, h0_dev, comm_dev, weights_dev, bias_dev, T, E, N, h_out_dev, agent0_out_dev, &sv,
                                       static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL((recurrent_infer_kernel<true, false>), dim3(persistent_grid(E)), dim3(NTHR), 0, static_cast<hipStream_t>(stream), gi_dev, h0_dev, comm_dev,
                       weights_dev, bias_dev, T, E, N, h_out_dev, agent0_out_dev, sv, row_index_dev, (long long)num_rows, (const int4 *)nullptr, agent0_stride);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

#if MAPF_RECUR_NT == 3
int mapf_recurrent_infer(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev, const uint16_t *weights_dev, const float *bias_dev,
                         int T, int E, int N, uint16_t *h_out_dev, uint16_t *agent0_out_dev, const int32_t *row_index_dev, int64_t num_rows,
                         void *stream) {
    return mapf_recurrent_infer_packed(gi_dev, h0_dev, comm_dev, weights_dev, bias_dev, T, E, N, h_out_dev, agent0_out_dev, row_index_dev, num_rows, 0, stream);
}

int mapf_recurrent_forward_save(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev, const uint16_t *weights_dev,
                                const float *bias_dev, int T, int E, int N, uint16_t *h_out_dev, uint16_t *agent0_out_dev,
                                uint16_t *const *save_dev, const int32_t *row_index_dev, int64_t num_rows, void *stream) {
    return mapf_recurrent_forward_save_packed(gi_dev, h0_dev, comm_dev, weights_dev, bias_dev, T, E, N, h_out_dev, agent0_out_dev, save_dev, row_index_dev,
                                              num_rows, 0, stream);
}
#endif

#if MAPF_RECUR_NT == 1
// One step of E environments of DIFFERENT agent counts (each <= 16) in one launch: the policy recurrence of all active curriculum
// levels (reference worker.py:378 runs model.step per environment; the levels' (num_agents, map) differ, environment.py:148-151).
extern "C" int mapf_recurrent_infer_multi(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev, const uint16_t *weights_dev,
                                          const float *bias_dev, int E, const int32_t *envtab_dev, uint16_t *h_out_dev, void *stream) {
    if (E < 0 || !gi_dev || !comm_dev || !weights_dev || !bias_dev || !h_out_dev || !envtab_dev) return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(gi_dev) & 7) || (reinterpret_cast<uintptr_t>(h0_dev) & 15) || (reinterpret_cast<uintptr_t>(weights_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(bias_dev) & 15) || (reinterpret_cast<uintptr_t>(h_out_dev) & 15) || (reinterpret_cast<uintptr_t>(envtab_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    if (E == 0) return MAPF_OK;
    hipLaunchKernelGGL((recurrent_infer_kernel<false, false>), dim3(E), dim3(NTHR), 0, static_cast<hipStream_t>(stream), gi_dev, h0_dev, comm_dev,
                       weights_dev, bias_dev, 1, E, 1, h_out_dev, (uint16_t *)nullptr, RecurSave{}, (const int32_t *)nullptr, 0ll,
                       reinterpret_cast<const int4 *>(envtab_dev), 0);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}
#endif

#if MAPF_RECUR_NT == 3
}  // extern "C"
#endif
#ifdef MAPF_RECUR_TRACE
#define RECUR_TRACE_NAME2(a, b) a##b
#define RECUR_TRACE_NAME(a, b) RECUR_TRACE_NAME2(a, b)
extern "C" int RECUR_TRACE_NAME(mapf_recur_trace_read_nt, MAPF_RECUR_NT)(unsigned long long *out, int reset) {
    int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_trace_n), sizeof(int)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 128) != hipSuccess) return -1;
    if (reset) {
        const int zero = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace_n), &zero, sizeof(int)) != hipSuccess) return -1;
    }
    return n;
}
#endif
