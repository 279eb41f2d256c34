// mapf_update.hip -- the glue of one DQN batch update as a handful of kernels (reference worker.py:287-338 `Learner.train` body,
// model.py:242-262 `bootstrap`'s head; see include/mapf_dqn.h).  None of this is heavy work -- the encoder and recurrence kernels
// are --, but as PyTorch operations it was ~200 of the ~300 launches of an update and most of its host time:
//   * plan_mark / plan_rows   which (step, agent) entries of a sampled window can reach agent 0's Q-value (the backward closure of
//                             mapf_window_relevance), the window's agents renumbered so that the entries needed at step t are a
//                             PREFIX of the agents (agent 0 first), and everything the encoder / recurrence launches need in that
//                             compact numbering: observation rows, row indices, communication masks, initial hidden states;
//   * head_fwd / head_grad    dueling head of the online and the target network, TD error, priorities, Huber loss (worker.py:296-310,
//                             341-344) and their gradients down to agent 0's hidden states and the head's parameters;
//   * recur_pack              fp32 parameters -> the MFMA-fragment images of mapf_recurrent_infer / _backward;
//   * recur_bias_grads        the recurrence's bias gradients from the backward kernel's per-window column sums;
//   * sumsq / adam            global gradient norm, clip (worker.py:319) and the Adam step (worker.py:260,322) over the flat
//                             parameter / gradient buffers, also refreshing the bf16 copy of the parameters.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>

#include "mapf_dqn.h"
#include "mapf_env.h"

namespace {

__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16(float f) {  // round to nearest even (finite inputs)
    uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

// ---------------------------------------------------------------------------------------------------------------------------
// plan_mark: one workgroup per window, thread j = agent j (N <= 128).  The closure walk is mapf_window_relevance's (csrc/mapf_dqn.hip);
// on top of it every agent remembers the LAST step at which it is in the set.  The set only shrinks going forward in time, so with
// the agents ordered by that step (descending; ties by agent id; agent 0 -- needed until the window's last step -- first) the agents
// needed at step t are exactly the first nact[t] of the order.
// ---------------------------------------------------------------------------------------------------------------------------
struct PlanMarkArgs {
    const uint8_t *comm;  // masks; window b, step t at comm + b * sB + t * sT, [N][N] contiguous
    long long sB, sT;
    const long long *steps;  // [B] 1-based index of the step whose agent-0 state is learned from ...
    const float *extra;      // ... plus extra[b] (optional: the target window ends `forward steps` later, worker.py:296)
    int T, B, N;
    int mark_all;    // every agent at every step up to the window's last one (no pruning: the reference encodes them all)
    uint8_t *rel;    // optional [T][B][N] (the relevance mask itself)
    int16_t *slot;   // [B][N]  position of agent j in the window's order, -1 = never needed
    int16_t *order;  // [B][N]  agent at position i (i < nag[b]), -1 behind
    int32_t *nact;   // [T][B]  agents needed at step t = a prefix of the order
    int32_t *cnt;    // [B]     sum over t of nact = observations of the window to encode
    int32_t *nag;    // [B]     agents needed anywhere in the window (= nact[0][b])
    int32_t *ucnt;   // [B]     optional: := 0 (obs_dup_kernel counts the window's DISTINCT observations into it)
};

__global__ void __launch_bounds__(128) plan_mark_kernel(PlanMarkArgs p) {
    __shared__ unsigned long long s_ballot[2];
    __shared__ int s_key[128], s_last[128], s_ord[128], s_cnt[128];
    __shared__ __attribute__((aligned(16))) uint8_t s_m[128 * 128];
    const int b = blockIdx.x, j = threadIdx.x, wv = j >> 6, T = p.T, N = p.N;
    const long long last = p.steps[b] + (p.extra ? (long long)p.extra[b] : 0) - 1;
    unsigned long long r0 = 0, r1 = 0;
    int my_last = -1;
    for (int t = T - 1; t >= 0; --t) {
        if (t == last) r0 |= 1ull;
        if (p.mark_all && t == last) {
            r0 = N >= 64 ? ~0ull : ((1ull << N) - 1ull);
            r1 = N > 64 ? (N >= 128 ? ~0ull : ((1ull << (N - 64)) - 1ull)) : 0ull;
        } else if (t <= last) {
            // the step's mask goes to LDS first: the closure below reads ~2 x |needed| of its bytes per thread, one after the other --
            // from global memory that was a load latency each (round 5: 170 -> ~25 us per launch at 40 agents)
            const uint8_t *mg = p.comm + (long long)b * p.sB + (long long)t * p.sT;
            const int NN = N * N;
            if ((reinterpret_cast<uintptr_t>(mg) & 3) == 0 && (NN & 3) == 0) {
                for (int i = j; i < NN / 4; i += 128) reinterpret_cast<uint32_t *>(s_m)[i] = reinterpret_cast<const uint32_t *>(mg)[i];
            } else {
                for (int i = j; i < NN; i += 128) s_m[i] = mg[i];
            }
            __syncthreads();
            const uint8_t *m = s_m;
            for (int round = 0; round < 2; ++round) {  // i needed and i reads j  =>  j needed (two attention rounds per step)
                bool v = false;
                for (unsigned long long w = r0; w != 0 && j < N; w &= w - 1) v |= m[(__ffsll((long long)w) - 1) * N + j] != 0;
                for (unsigned long long w = r1; w != 0 && j < N; w &= w - 1) v |= m[(63 + __ffsll((long long)w)) * N + j] != 0;
                const unsigned long long bal = __ballot(v);
                __syncthreads();
                if ((j & 63) == 0) s_ballot[wv] = bal;
                __syncthreads();
                r0 |= s_ballot[0];
                r1 |= s_ballot[1];
            }
        }
        const bool in = j < N && (((j < 64 ? r0 >> j : r1 >> (j - 64)) & 1ull) != 0);
        if (in && my_last < 0) my_last = t;
        if (p.rel && j < N) p.rel[((size_t)t * p.B + b) * N + j] = (uint8_t)in;
    }
    const int key = (j < N && my_last >= 0) ? (T - 1 - my_last) * 128 + j : 0x7FFFFFFF;
    s_key[j] = key;
    s_last[j] = j < N ? my_last : -1;
    s_ord[j] = -1;
    __syncthreads();
    int rank = 0;
    for (int k = 0; k < N; ++k) rank += s_key[k] < key;
    const bool marked = key != 0x7FFFFFFF;
    if (marked) s_ord[rank] = j;
    if (j < T) {
        int c = 0;
        for (int k = 0; k < N; ++k) c += s_last[k] >= j;
        s_cnt[j] = c;
        p.nact[(size_t)j * p.B + b] = c;
    }
    __syncthreads();
    if (j < N) {
        p.slot[(size_t)b * N + j] = (int16_t)(marked ? rank : -1);
        p.order[(size_t)b * N + j] = (int16_t)s_ord[j];
    }
    if (j == 0) {
        int tot = 0;
        for (int t = 0; t < T; ++t) tot += s_cnt[t];
        p.cnt[b] = tot;
        p.nag[b] = s_cnt[0];
        if (p.ucnt) p.ucnt[b] = 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// obs_dup: an agent that stands still in an unchanged neighbourhood -- or comes back to a cell whose neighbourhood looks the same -- has
// the SAME observation at several steps of a window, and the encoder is a deterministic per-observation function: 37 % of the rows a
// batch would encode repeat an EARLIER row of the same agent in the same window (26 % the row one step earlier, which is all round 3
// reused; tools/dup_probe.py).  One wavefront per (window, agent): a 64-bit hash of every needed step's 486 values (one read of the
// row), then per step the first earlier step with the same hash, confirmed value by value (exact, not probabilistic):
// first[t][b][j] = that step, or t itself.  The distinct observations of every window are counted for both closures.
// ---------------------------------------------------------------------------------------------------------------------------
struct ObsDupArgs {
    int T, To, B, N;  // target window steps, online window steps
    const uint16_t *obs;
    long long o_sB, o_sT;
    const int16_t *slot_o, *slot_t;  // [B][N]
    const int32_t *nact_o, *nact_t;  // [To][B], [T][B]
    uint8_t *dup;                    // [T][B][N]: the first step of the window at which this agent saw the same observation (<= t)
    int32_t *ucnt_o, *ucnt_t;        // [B]
};

__device__ __forceinline__ unsigned long long mix64u(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void __launch_bounds__(256) obs_dup_kernel(ObsDupArgs p) {
    __shared__ unsigned long long s_hash[4][MAPF_PLAN_MAX_STEPS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long total = (long long)p.B * p.N;
    for (long long w = (long long)blockIdx.x * 4 + wv; w < total; w += (long long)gridDim.x * 4) {
        const int j = (int)(w % p.N), b = (int)(w / p.N);
        const int st = p.slot_t[(size_t)b * p.N + j], so = p.slot_o[(size_t)b * p.N + j];
        // the steps at which this agent is needed form a prefix 0 .. nt - 1 (the needed set only shrinks going forward): lane t looks at
        // step t (one load latency for the whole window; a loop over the steps paid one per step -- round 5)
        const bool need_t = lane < p.T && st >= 0 && st < p.nact_t[(size_t)lane * p.B + b];
        const bool need_o = lane < p.To && so >= 0 && so < p.nact_o[(size_t)lane * p.B + b];
        const int nt = __popcll(__ballot(need_t)), no = __popcll(__ballot(need_o));
        const uint16_t *base = p.obs + (long long)b * p.o_sB + (long long)j * 486;
        for (int t0 = 0; t0 < nt; t0 += 4) {  // (wave-uniform trip count; the rows of four steps are requested together)
            uint32_t v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t *row = reinterpret_cast<const uint32_t *>(base + (long long)(t0 + u < nt ? t0 + u : t0) * p.o_sT);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[u][q] = lane + 64 * q < 243 ? row[lane + 64 * q] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned long long h = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (lane + 64 * q < 243) h += mix64u(((unsigned long long)v[u][q] << 8) ^ (unsigned long long)(lane + 64 * q));
#pragma unroll
                for (int s2 = 32; s2 > 0; s2 >>= 1) h += __shfl_xor(h, s2, 64);
                if (lane == 0 && t0 + u < nt) s_hash[wv][t0 + u] = h;
            }
        }
        int distinct_t = 0, distinct_o = 0;
        for (int t = 0; t < p.T; ++t) {
            int first = t;
            if (t < nt) {
                const unsigned long long h = s_hash[wv][t];
                for (int t0 = 0; t0 < t; ++t0) {
                    if (s_hash[wv][t0] != h) continue;
                    const uint32_t *cur = reinterpret_cast<const uint32_t *>(base + (long long)t * p.o_sT);
                    const uint32_t *old = reinterpret_cast<const uint32_t *>(base + (long long)t0 * p.o_sT);
                    bool diff = false;
                    for (int d = lane; d < 243; d += 64) diff |= cur[d] != old[d];
                    if (__ballot(diff) == 0ull) {
                        first = t0;
                        break;
                    }
                }
                distinct_t += first == t;
                distinct_o += (first == t && t < no);
            }
            if (lane == 0) p.dup[((size_t)t * p.B + b) * p.N + j] = (uint8_t)first;
        }
        if (lane == 0) {
            if (distinct_t) atomicAdd(&p.ucnt_t[b], distinct_t);
            if (distinct_o) atomicAdd(&p.ucnt_o[b], distinct_o);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// plan_rows: one workgroup per window.  Rows (= observations to encode) are numbered window by window, step by step, position by
// position: row(b, t, i) = sum_{b' < b} cnt[b'] + sum_{t' < t} nact[t'][b] + i  for i < nact[t][b].
// ---------------------------------------------------------------------------------------------------------------------------
struct PlanRowsArgs {
    int T, B, N, Nc;  // Nc: agents per window in the compact layout (a multiple of 16, >= max nag)
    const int16_t *order;
    const int32_t *nact, *cnt, *nag;
    const uint8_t *comm;
    long long c_sB, c_sT;
    const uint16_t *hidden;  // f16 (or bf16) [B * N][256] (quirk Q4: every agent row of a window holds agent 0's state)
    int hidden_bf16;
    const uint16_t *obs;     // bf16 observations, window b / step t at obs + b * o_sB + t * o_sT (elements), [N][486] contiguous
    long long o_sB, o_sT;
    int32_t *gidx;       // [T][B][Nc] row of (t, b, position) or -1
    uint8_t *comm_c;     // [T][B][Nc][Nc] masks in the compact numbering; positions >= nact[t][b] read only themselves
    uint16_t *h0_c;      // [B][Nc][256] bf16
    long long *row_src;  // [rows] ([urows] with dup) element offset of a row's observation in `obs` (the gather: obs_gather_kernel)
    const uint8_t *dup;  // optional [>= T][B][N] from obs_dup_kernel: the entry repeats the same agent's observation of the step before
    const int32_t *ucnt; // [B] distinct observations per window (with dup)
    int32_t *umap;       // [rows] with dup: the row of DISTINCT observations (numbered like the rows, duplicates skipped) an entry uses
    int32_t *row_tbp;    // [rows] optional: (t << 24) | (position << 16) | window of every row
    // bucket-sized launches (mapf_plan_rows_padded): plan_masks_kernel first initialises row_src[0 .. fill_urows) = 0, umap[0 .. fill_rows) = 0 and
    // row_tbp[0 .. fill_rows) = -1, so that every padding row behind the real ones is harmless; plan_rows_kernel then writes the real rows
    long long fill_rows, fill_urows;
};

constexpr int OBS_DWORDS = 243;  // 486 bf16

// masks and initial hidden states in the compact numbering: workgroup (window b, step t) gathers comm_c[t][b], workgroup (b, T) h0_c[b]
// (apart from plan_rows_kernel, whose per-window work is a short serial chain: 20 x more, small, independent workgroups)
__global__ void __launch_bounds__(256) plan_masks_kernel(PlanRowsArgs p) {
    __shared__ short s_ord[128];
    const int b = blockIdx.x, t = blockIdx.y, tid = threadIdx.x, T = p.T, B = p.B, N = p.N, Nc = p.Nc;
    if (p.fill_rows > 0 || p.fill_urows > 0) {  // (grid-stride over all workgroups of this launch; plan_rows_kernel runs behind it)
        const long long nth = (long long)gridDim.x * gridDim.y * 256, me = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid;
        if (p.row_src)
            for (long long i = me; i < p.fill_urows; i += nth) p.row_src[i] = 0;
        if (p.umap)
            for (long long i = me; i < p.fill_rows; i += nth) p.umap[i] = 0;
        if (p.row_tbp)
            for (long long i = me; i < p.fill_rows; i += nth) p.row_tbp[i] = -1;
    }
    if (tid < N) s_ord[tid] = p.order[(size_t)b * N + tid];
    __syncthreads();
    if (t < T) {
        const int na = p.nact[(size_t)t * B + b];
        const uint8_t *src = p.comm + (long long)b * p.c_sB + (long long)t * p.c_sT;
        if (Nc < 16) {
            // Nc = 4 or 8: 16 / Nc windows share a 16 x 16 mask tile, block-diagonally (mapf_recurrent_*_packed): this window writes its Nc
            // rows of tile b / K over all 16 columns -- zeros outside its own block
            const int K = 16 / Nc, k = b % K;
            uint8_t *dst = p.comm_c + ((size_t)t * (B / K) + b / K) * 256 + (size_t)k * Nc * 16;
            for (int idx = tid; idx < Nc * 16; idx += 256) {
                const int i = idx >> 4, jj = (idx & 15) - k * Nc;  // jj: the column as a position of this window
                uint8_t v = (uint8_t)(i == jj);
                if (jj >= 0 && jj < Nc && i < na && jj < na) v = src[(long long)s_ord[i] * N + s_ord[jj]] != 0;
                dst[idx] = v;
            }
            return;
        }
        uint8_t *dst = p.comm_c + ((size_t)t * B + b) * Nc * Nc;
        for (int idx = tid; idx < Nc * Nc; idx += 256) {
            const int i = idx / Nc, j = idx - i * Nc;
            uint8_t v = (uint8_t)(i == j);
            if (i < na && j < na) v = src[(long long)s_ord[i] * N + s_ord[j]] != 0;
            dst[idx] = v;
        }
        return;
    }
    const int nag = p.nag[b];
    for (int idx = tid; idx < Nc * 32; idx += 256) {  // 8 channels per task
        const int i = idx >> 5, ch = idx & 31;
        uint4 o = make_uint4(0, 0, 0, 0);
        if (i < nag) {
            const uint4 h = *reinterpret_cast<const uint4 *>(p.hidden + ((size_t)b * N + s_ord[i]) * 256 + ch * 8);
            const uint32_t w[4] = {h.x, h.y, h.z, h.w};
            uint32_t r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                r[k] = p.hidden_bf16 ? w[k]
                                     : (f32_to_bf16(f16_to_f32((uint16_t)(w[k] & 0xFFFFu))) | (f32_to_bf16(f16_to_f32((uint16_t)(w[k] >> 16))) << 16));
            o = make_uint4(r[0], r[1], r[2], r[3]);
        }
        *reinterpret_cast<uint4 *>(p.h0_c + ((size_t)b * Nc + i) * 256 + ch * 8) = o;
    }
}

__global__ void __launch_bounds__(256) plan_rows_kernel(PlanRowsArgs p) {
    __shared__ int s_red[256];
    __shared__ int s_nact[128], s_base[129];
    __shared__ short s_ord[128];
    __shared__ unsigned short s_map[128 * 20];  // (t << 8 | position) per row of this window; T <= 20 steps... checked on the host
    const int b = blockIdx.x, tid = threadIdx.x, T = p.T, B = p.B, N = p.N, Nc = p.Nc;
    int part = 0;
    for (int k = tid; k < b; k += 256) part += p.cnt[k];
    s_red[tid] = part;
    if (tid < N) s_ord[tid] = p.order[(size_t)b * N + tid];
    if (tid < T) s_nact[tid] = p.nact[(size_t)tid * B + b];
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (tid < d) s_red[tid] += s_red[tid + d];
        __syncthreads();
    }
    const int offset = s_red[0];
    if (tid == 0) {
        int acc = 0;
        for (int t = 0; t < T; ++t) {
            s_base[t] = acc;
            acc += s_nact[t];
        }
        s_base[T] = acc;
    }
    __syncthreads();
    const int rows = s_base[T];
    for (int idx = tid; idx < T * Nc; idx += 256) {
        const int t = idx / Nc, i = idx - t * Nc;
        const bool on = i < s_nact[t];
        p.gidx[((size_t)t * B + b) * Nc + i] = on ? offset + s_base[t] + i : -1;
        if (on) {
            s_map[s_base[t] + i] = (unsigned short)((t << 8) | i);
            if (p.row_tbp) p.row_tbp[offset + s_base[t] + i] = (t << 24) | (i << 16) | b;
        }
    }
    __syncthreads();
    if (p.dup == nullptr) {
        if (p.row_src)
            for (int r = tid; r < rows; r += 256) {
                const int t = s_map[r] >> 8, i = s_map[r] & 255;
                p.row_src[offset + r] = (long long)b * p.o_sB + (long long)t * p.o_sT + (long long)s_ord[i] * 486;
            }
        return;
    }
    // distinct observations: position i remembers the id its observation got at every step; an entry whose observation first appeared at
    // an earlier step t0 (dup[t] = t0 < t) takes the id of (t0, i)
    __shared__ int s_uid[128][MAPF_PLAN_MAX_STEPS + 1], s_flag[128];
    int upart = 0;
    for (int k = tid; k < b; k += 256) upart += p.ucnt[k];
    s_red[tid] = upart;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (tid < d) s_red[tid] += s_red[tid + d];
        __syncthreads();
    }
    int next = s_red[0];  // first distinct row of this window (identical in every thread)
    // this position's duplicate flags of all steps, requested at once instead of one dependent global load per step of the loop below
    // (one workgroup per window: nothing else hides the latency)
    __shared__ short s_t0[MAPF_PLAN_MAX_STEPS][128];
    if (tid < 128) {
        int t0s[MAPF_PLAN_MAX_STEPS];
#pragma unroll
        for (int t = 0; t < MAPF_PLAN_MAX_STEPS; ++t) {
            t0s[t] = t;
            if (t < T && tid < s_nact[t]) t0s[t] = p.dup[((size_t)t * B + b) * N + s_ord[tid]];
        }
#pragma unroll
        for (int t = 0; t < MAPF_PLAN_MAX_STEPS; ++t) s_t0[t][tid] = (short)t0s[t];
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int na = s_nact[t];
        const int t0 = tid < 128 ? s_t0[t][tid] : t;
        if (tid < 128) s_flag[tid] = (tid < na) && t0 >= t;
        __syncthreads();
        if (tid < na) {
            if (s_flag[tid]) {
                int before = 0;
                for (int k = 0; k < tid; ++k) before += s_flag[k];
                s_uid[tid][t] = next + before;
                p.row_src[next + before] = (long long)b * p.o_sB + (long long)t * p.o_sT + (long long)s_ord[tid] * 486;
            } else {
                s_uid[tid][t] = s_uid[tid][t0];
            }
            p.umap[offset + s_base[t] + tid] = s_uid[tid][t];
            // (row_tbp was written above without the flag: an entry that reuses an earlier row is not the head of its id)
            if (p.row_tbp && !s_flag[tid]) p.row_tbp[offset + s_base[t] + tid] |= 1 << 30;
        }
        int tot = 0;
        for (int k = 0; k < na; ++k) tot += s_flag[k];
        next += tot;
        __syncthreads();
    }
}

// d_u[u][:] = sum over the entries r with umap[r] == u of d_rows[r][:] (fp32 sum, one bf16 rounding): the gradient of a distinct
// observation's row is the sum over the steps of the same (window, agent) that share it.  One wavefront per row; the id's FIRST entry
// (row_tbp bit 30 clear) does the sum, walking forward through gidx (the same position at later steps), the others return.
struct DedupSumArgs {
    int T, B, Nc, W8;  // W8: row width in 8-element (16-byte) chunks
    long long rows;
    const int32_t *gidx, *umap;
    const int32_t *row_tbp;  // [rows] (t << 24) | (position << 16) | window
    const uint4 *d_rows;
    uint4 *d_u;
};

__global__ void __launch_bounds__(256) dedup_sum_kernel(DedupSumArgs p) {
    const int lane = threadIdx.x & 63;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < p.rows; r += (long long)gridDim.x * 4) {
        const int tbp = p.row_tbp[r];
        if (tbp < 0) continue;  // (a padding entry of a bucket-sized launch: the caller filled row_tbp with -1 behind the real rows)
        if (tbp & (1 << 30)) continue;  // an entry that reuses the row of an earlier step: the head of the id does the sum
        const int t = tbp >> 24, pos = (tbp >> 16) & 255, b = tbp & 0xFFFF, u = p.umap[r];
        // The later steps of the same position that carry the same id, found by the lanes in parallel (lane k looks at step t + 1 + k;
        // T <= 64): one round trip through gidx / umap instead of a dependent chain per step and per chunk (round 4: 65 -> ~20 us).
        const int nlater = p.T - 1 - t;
        int r2 = -1;
        if (lane < nlater) r2 = p.gidx[((size_t)(t + 1 + lane) * p.B + b) * p.Nc + pos];
        const unsigned long long neg = __ballot(lane < nlater && r2 < 0);
        const int stop = neg ? __ffsll((long long)neg) - 1 : nlater;  // (the needed set only shrinks going forward: nothing behind the first gap)
        const bool same = lane < stop && p.umap[r2] == u;
        unsigned long long mm = __ballot(same);  // bit k: step t + 1 + k shares the row; summed in step order: fixed sum order
        for (int c = lane; c < p.W8; c += 64) {
            float acc[8];
            {
                const uint4 v = p.d_rows[r * p.W8 + c];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[2 * k] = bf16_to_f32(w[k] & 0xFFFFu);
                    acc[2 * k + 1] = bf16_to_f32(w[k] >> 16);
                }
            }
            for (unsigned long long x = mm; x != 0ull;) {  // four rows per round: their loads are in flight together
                uint4 v[4];
                int n = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = make_uint4(0, 0, 0, 0);
                    if (x != 0ull) {
                        const int l = __ffsll((long long)x) - 1;
                        x &= x - 1ull;
                        const int rr = __builtin_amdgcn_readlane(r2, l);  // (any lane, whatever EXEC: the second chunk pass runs on lanes 0..31)
                        v[k] = p.d_rows[(long long)rr * p.W8 + c];
                        n = k + 1;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < n) {
                        const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[2 * q] += bf16_to_f32(w[q] & 0xFFFFu);
                            acc[2 * q + 1] += bf16_to_f32(w[q] >> 16);
                        }
                    }
                }
            }
            p.d_u[(long long)u * p.W8 + c] = make_uint4(f32_to_bf16(acc[0]) | (f32_to_bf16(acc[1]) << 16), f32_to_bf16(acc[2]) | (f32_to_bf16(acc[3]) << 16),
                                                      f32_to_bf16(acc[4]) | (f32_to_bf16(acc[5]) << 16), f32_to_bf16(acc[6]) | (f32_to_bf16(acc[7]) << 16));
        }
    }
}

// obs_rows[r][:] = obs[row_src[r] ..+486] (bf16): one wavefront per row, 243 dwords
__global__ void __launch_bounds__(256) obs_gather_kernel(const uint16_t *__restrict__ obs, const long long *__restrict__ row_src, long long rows,
                                                         uint16_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * 4;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += stride) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(obs + row_src[r]);
        uint32_t *dst = reinterpret_cast<uint32_t *>(out + r * 486);
        for (int d = lane; d < OBS_DWORDS; d += 64) dst[d] = src[d];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Dueling head + TD error + Huber loss, forward and backward (model.py:259-262, worker.py:296-310,341-344).
// head_fwd: one workgroup per sample, thread c = hidden channel.  head_grad: one workgroup, sums over the samples in a fixed order.
// ---------------------------------------------------------------------------------------------------------------------------
struct HeadArgs {
    int B, To, Tt;
    const uint16_t *a0_on;   // bf16 [To][B][256] agent 0's state after every step of the online window (online network)
    const uint16_t *a0_tg;   // bf16 [Tt][B][256] target network on the target window
    const uint16_t *a0_on2;  // bf16 [Tt][B][256] or NULL: online network on the target window (double-DQN: it picks the action)
    const long long *bt;     // [B] 1-based step of the online window
    const float *steps;      // [B] forward steps (1 or 2)
    const long long *action; // [B]
    const float *reward, *done, *weights;  // [B]
    const float *w_adv, *b_adv, *w_st, *b_st;          // online head, fp32: [5][256], [5], [256], [1]
    const float *tw_adv, *tb_adv, *tw_st, *tb_st;      // target head
    float gamma;
    // outputs
    float *q, *q_next, *td, *lossterm;  // [B]
    double *prio;                       // [B] |td| clamped at 1e-6
    float *dA;                          // [B][8]: dA[0..4], dV
    uint16_t *d_a0;                     // bf16 [To][B][256]: gradient w.r.t. a0_on (zero except at step bt - 1)
    // head_grad
    float *g_w_adv, *g_b_adv, *g_w_st, *g_b_st;  // accumulated into (+=)
    float *loss;                                 // [1]
};

__device__ __forceinline__ void dueling(const float (&d)[6], const float *b_adv, const float *b_st, float (&q)[5]) {
    float a[5], mean = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        a[k] = d[k] + b_adv[k];
        mean += a[k];
    }
    mean *= 0.2f;
    const float v = d[5] + b_st[0];
#pragma unroll
    for (int k = 0; k < 5; ++k) q[k] = v + a[k] - mean;  // model.py:262
}

__global__ void __launch_bounds__(256) head_fwd_kernel(HeadArgs p) {
    __shared__ float s_part[4][18];
    __shared__ float s_dA[6];
    const int b = blockIdx.x, c = threadIdx.x, lane = c & 63, w = c >> 6, B = p.B;
    const int t_on = (int)p.bt[b] - 1;
    const int t_tg = (int)p.bt[b] + (int)p.steps[b] - 1;
    const float x = bf16_to_f32(p.a0_on[((size_t)t_on * B + b) * 256 + c]);
    const float y = bf16_to_f32(p.a0_tg[((size_t)t_tg * B + b) * 256 + c]);
    const float z = p.a0_on2 ? bf16_to_f32(p.a0_on2[((size_t)t_tg * B + b) * 256 + c]) : 0.f;
    float v[18];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float wa = p.w_adv[k * 256 + c];
        v[k] = x * wa;
        v[6 + k] = y * p.tw_adv[k * 256 + c];
        v[12 + k] = z * wa;
    }
    v[5] = x * p.w_st[c];
    v[11] = y * p.tw_st[c];
    v[17] = z * p.w_st[c];
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        float s = v[k];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
        if (lane == 0) s_part[w][k] = s;
    }
    __syncthreads();
    if (c == 0) {
        float d_on[6], d_tg[6], d_o2[6], q_on[5], q_tg[5], q_o2[5];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            d_on[k] = s_part[0][k] + s_part[1][k] + s_part[2][k] + s_part[3][k];
            d_tg[k] = s_part[0][6 + k] + s_part[1][6 + k] + s_part[2][6 + k] + s_part[3][6 + k];
            d_o2[k] = s_part[0][12 + k] + s_part[1][12 + k] + s_part[2][12 + k] + s_part[3][12 + k];
        }
        dueling(d_on, p.b_adv, p.b_st, q_on);
        dueling(d_tg, p.tb_adv, p.tb_st, q_tg);
        dueling(d_o2, p.b_adv, p.b_st, q_o2);
        // worker.py:300-303: max_a Q_target; double-DQN: the ONLINE network picks (first maximum, as torch.argmax)
        int pick = 0;
        float best = p.a0_on2 ? q_o2[0] : q_tg[0];
#pragma unroll
        for (int k = 1; k < 5; ++k) {
            const float cur = p.a0_on2 ? q_o2[k] : q_tg[k];
            if (cur > best) {
                pick = k;
                best = cur;
            }
        }
        // (selects, not q[pick] / q[act]: a dynamically indexed local array lives in scratch memory)
        float q_pick = q_tg[0];
#pragma unroll
        for (int k = 1; k < 5; ++k) q_pick = pick == k ? q_tg[k] : q_pick;
        const float qn = (1.f - p.done[b]) * q_pick;
        const int act = (int)p.action[b];
        float qa = q_on[0];
#pragma unroll
        for (int k = 1; k < 5; ++k) qa = act == k ? q_on[k] : qa;
        const float td = qa - (p.reward[b] + powf(p.gamma, p.steps[b]) * qn);  // worker.py:306
        const float a = fabsf(td), wgt = p.weights[b];
        const float hub = a < 1.f ? 0.5f * a * a : a - 0.5f;  // worker.py:341-344, kappa = 1
        p.q[b] = qa;
        p.q_next[b] = qn;
        p.td[b] = td;
        p.prio[b] = (double)fmaxf(a, 1e-6f);  // worker.py:308
        p.lossterm[b] = wgt * hub;
        const float g = wgt * (a < 1.f ? td : (td > 0.f ? 1.f : -1.f)) / (float)B;  // d mean(w huber(td)) / d q[act]
        // q_k = V + A_k - mean(A): dV = sum_k dq_k = g, dA_k = dq_k - g / 5
#pragma unroll
        for (int k = 0; k < 5; ++k) s_dA[k] = (k == act ? g : 0.f) - 0.2f * g;
        s_dA[5] = g;
#pragma unroll
        for (int k = 0; k < 6; ++k) p.dA[(size_t)b * 8 + k] = s_dA[k];
    }
    __syncthreads();
    float ds = s_dA[5] * p.w_st[c];
#pragma unroll
    for (int k = 0; k < 5; ++k) ds += s_dA[k] * p.w_adv[k * 256 + c];
    const uint16_t dsb = (uint16_t)f32_to_bf16(ds);
    for (int t = 0; t < p.To; ++t) p.d_a0[((size_t)t * B + b) * 256 + c] = t == t_on ? dsb : (uint16_t)0;
}

__global__ void __launch_bounds__(256) head_grad_kernel(HeadArgs p) {
    // dA / step indices of all samples into LDS with coalesced loads first; the loop over the samples then carries no dependent
    // global load (unrolled: 8 independent state loads in flight)
    extern __shared__ float s_head[];  // [B][8] dA | [B] row offset (as int)
    const int c = threadIdx.x, B = p.B;
    float *s_dA = s_head;
    int *s_row = reinterpret_cast<int *>(s_head + (size_t)B * 8);
    for (int i = c; i < B * 8; i += 256) s_dA[i] = p.dA[i];
    for (int b = c; b < B; b += 256) s_row[b] = ((int)p.bt[b] - 1) * B + b;
    __syncthreads();
    float gw[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int b = 0; b < B; ++b) {
        const float x = bf16_to_f32(p.a0_on[(size_t)s_row[b] * 256 + c]);
#pragma unroll
        for (int k = 0; k < 6; ++k) gw[k] += s_dA[b * 8 + k] * x;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) p.g_w_adv[k * 256 + c] += gw[k];
    p.g_w_st[c] += gw[5];
    if (c < 6) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += s_dA[b * 8 + c];
        if (c < 5) p.g_b_adv[c] += s;
        else p.g_b_st[0] += s;
    } else if (c == 6) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += p.lossterm[b];
        p.loss[0] = s / (float)B;  // worker.py:310
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// fp32 parameter matrices -> bf16 MFMA A-fragment images ([O/16][K/32][lane = 16 ((k % 32) / 8) + o % 16][k % 8]).
// A segment's source is a virtual matrix V [R][C] = up to 3 equally tall matrices stacked along the rows;
// plain: element (o, k) = V[o][k];  transposed: element (o, k) = V[r0 + k][o].
// ---------------------------------------------------------------------------------------------------------------------------
struct PackSeg {
    const float *src[3];
    int rows_per_src, C;  // rows of each stacked matrix, columns
    int O, K, r0, transposed;
    int dst;  // element offset of the segment in the image
};
struct PackArgs2 {
    PackSeg seg[12];
    int nseg, total;
    const float *bias_src[8];
    int bias_len[8], nbias;
    uint16_t *out;
    float *bias_out;
};

__global__ void __launch_bounds__(256) recur_pack_kernel(PackArgs2 p) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < p.total) {
        int s = 0;
        while (s + 1 < p.nseg && idx >= p.seg[s + 1].dst) ++s;
        const PackSeg &g = p.seg[s];
        const int f = idx - g.dst, slot = f & 7, lane = (f >> 3) & 63, tk = f >> 9;
        const int ksteps = g.K / 32, tile = tk / ksteps, ks = tk - tile * ksteps;
        const int o = 16 * tile + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + slot;
        const int r = g.transposed ? g.r0 + k : o, cc = g.transposed ? o : k;
        const float v = g.src[r / g.rows_per_src][(size_t)(r % g.rows_per_src) * g.C + cc];
        p.out[idx] = (uint16_t)f32_to_bf16(v);
    }
    if (p.bias_out && blockIdx.x == 0) {
        int off = 0;
        for (int q = 0; q < p.nbias; ++q) {
            for (int i = threadIdx.x; i < p.bias_len[q]; i += 256) p.bias_out[off + i] = p.bias_src[q][i];
            off += p.bias_len[q];
        }
    }
}

// bias gradients of the recurrence from the backward kernel's per-window column sums (include/mapf_dqn.h: bsum [E][2432] =
// [update cell: dr | dz | dn | dn r][recurrent cell: dr | dz | dn | dn r][d_qkv: 384]); b_ih: (dr, dz, dn), b_hh: (dr, dz, dn r)
struct BiasGradArgs {
    const float *bsum;
    int E;
    float *rc_bih, *rc_bhh, *bq, *bk, *bv, *uc_bih, *uc_bhh;  // accumulated into (+=)
};

__global__ void __launch_bounds__(256) recur_bias_grads_kernel(BiasGradArgs p) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= MAPF_RECUR_BSUM_ELEMS) return;
    float s = 0.f;
#pragma unroll 8
    for (int e = 0; e < p.E; ++e) s += p.bsum[(size_t)e * MAPF_RECUR_BSUM_ELEMS + col];
    if (col < 2048) {
        const bool upd = col < 1024;
        const int k = upd ? col : col - 1024;
        float *bih = upd ? p.uc_bih : p.rc_bih, *bhh = upd ? p.uc_bhh : p.rc_bhh;
        if (k < 512) {
            bih[k] += s;
            bhh[k] += s;
        } else if (k < 768) {
            bih[k] += s;
        } else {
            bhh[k - 256] += s;
        }
    } else {
        const int k = col - 2048;
        float *dst = k < 128 ? p.bq : (k < 256 ? p.bk : p.bv);
        dst[k & 127] += s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Global gradient norm, clip and Adam over the flat buffers (torch.nn.utils.clip_grad_norm_ + torch.optim.Adam, worker.py:319-322)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int SUMSQ_BLOCKS = 256;

__global__ void __launch_bounds__(256) sumsq_kernel(const float *__restrict__ g, long long n, float *__restrict__ partial, long long *step_dev) {
    __shared__ float s_red[256];
    if (step_dev && blockIdx.x == 0 && threadIdx.x == 0) step_dev[0] += 1;  // (the step adam_kernel, the next launch, takes its bias corrections from)
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)SUMSQ_BLOCKS * 256) s += g[i] * g[i];
    s_red[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) s_red[threadIdx.x] += s_red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0];
}

struct AdamArgs {
    long long n;
    float *p, *g, *m, *v;
    uint16_t *p_bf16;  // optional bf16 copy of the parameters
    const float *partial;  // [SUMSQ_BLOCKS] from sumsq_kernel
    float *norm_out;       // [1] the gradient norm before clipping
    float lr, b1, b2, eps, bc1, bc2, max_norm;
    const long long *step_dev;  // optional: 1-based step count in device memory (bias corrections computed here instead of on the host)
};

__global__ void __launch_bounds__(256) adam_kernel(AdamArgs a) {
    __shared__ float s_red[256];
    s_red[threadIdx.x] = a.partial[threadIdx.x];  // SUMSQ_BLOCKS == blockDim: every block re-adds the partials in the same order
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) s_red[threadIdx.x] += s_red[threadIdx.x + d];
        __syncthreads();
    }
    const float norm = sqrtf(s_red[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.norm_out[0] = norm;
    const float coef = fminf(a.max_norm / (norm + 1e-6f), 1.f);  // clip_grad_norm_: clamp(max_norm / (total_norm + 1e-6), max = 1)
    float bc1 = a.bc1, bc2 = a.bc2;
    if (a.step_dev) {
        const double st = (double)a.step_dev[0];
        bc1 = (float)(1.0 - pow((double)a.b1, st));
        bc2 = (float)(1.0 - pow((double)a.b2, st));
    }
    const float step_size = a.lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
        const float g = a.g[i] * coef;
        const float m = a.m[i] + (g - a.m[i]) * (1.f - a.b1);      // exp_avg.lerp_(grad, 1 - beta1)
        const float v = a.v[i] * a.b2 + (1.f - a.b2) * g * g;      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(v) * inv_sqrt_bc2 + a.eps;
        const float w = a.p[i] - step_size * (m / denom);
        a.g[i] = g;
        a.m[i] = m;
        a.v[i] = v;
        a.p[i] = w;
        if (a.p_bf16) a.p_bf16[i] = (uint16_t)f32_to_bf16(w);
    }
}

// dense[r][:] = idx[r] >= 0 ? rows[idx[r]][:] : 0   /   rows[idx[r]][:] = dense[r][:] where idx[r] >= 0; `chunks` 16-byte pieces per row
__global__ void __launch_bounds__(256) rows_to_dense_kernel(const uint4 *__restrict__ rows, const int32_t *__restrict__ idx, uint4 *__restrict__ dense,
                                                            long long R, int chunks) {
    const long long total = R * chunks, stride = (long long)gridDim.x * 256;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += stride) {
        const long long r = q / chunks;
        const int c = (int)(q - r * chunks), src = idx[r];
        dense[q] = src >= 0 ? rows[(long long)src * chunks + c] : make_uint4(0, 0, 0, 0);
    }
}
__global__ void __launch_bounds__(256) dense_to_rows_kernel(const uint4 *__restrict__ dense, const int32_t *__restrict__ idx, uint4 *__restrict__ rows,
                                                            long long R, int chunks) {
    const long long total = R * chunks, stride = (long long)gridDim.x * 256;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += stride) {
        const long long r = q / chunks;
        const int c = (int)(q - r * chunks), dst = idx[r];
        if (dst >= 0) rows[(long long)dst * chunks + c] = dense[q];
    }
}

// rows [first, last) of up to 16 row-major buffers := 0 (the padding rows behind the compact rows of the recurrence's saved tensors /
// gradient outputs: they enter the weight-gradient GEMMs, whose K is padded to a multiple of the split size)
struct ZeroRowsArgs {
    void *ptr[24];
    int row_bytes[24];  // multiples of 16
    int n;
    long long first, last;
    const int32_t *first_dev;  // optional: the first row to clear, read from device memory (clamped to 0..last)
};
__global__ void __launch_bounds__(256) zero_rows_kernel(ZeroRowsArgs a) {
    const int k = blockIdx.y;
    if (k >= a.n) return;
    if (a.first_dev != nullptr) {
        const long long f = (long long)*a.first_dev;
        a.first = f < 0 ? 0 : (f > a.last ? a.last : f);
    }
    const long long chunks = (a.last - a.first) * (a.row_bytes[k] / 16);
    uint4 *dst = reinterpret_cast<uint4 *>(static_cast<unsigned char *>(a.ptr[k]) + a.first * a.row_bytes[k]);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) dst[i] = make_uint4(0, 0, 0, 0);
}

__global__ void __launch_bounds__(256) to_bf16_kernel(const float *__restrict__ src, uint16_t *__restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = (uint16_t)f32_to_bf16(src[i]);
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            std::fprintf(stderr, "mapf_update: %s failed: %s\n", #expr, hipGetErrorString(_e)); \
            return MAPF_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

}  // namespace

extern "C" {

int mapf_plan_mark(const uint8_t *comm_dev, int64_t stride_b, int64_t stride_t, const int64_t *steps_dev, const float *extra_steps_dev, int T, int B,
                   int N, int mark_all, uint8_t *rel_dev, int16_t *slot_dev, int16_t *order_dev, int32_t *nact_dev, int32_t *cnt_dev, int32_t *nag_dev,
                   int32_t *ucnt_dev, void *stream) {
    if (T < 1 || T > MAPF_PLAN_MAX_STEPS || B < 0 || N < 1 || N > 128 || !comm_dev || !steps_dev || !slot_dev || !order_dev || !nact_dev || !cnt_dev ||
        !nag_dev || stride_b < 0 || stride_t < 0)
        return MAPF_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(steps_dev) & 7) return MAPF_ERR_INVALID_ARG;
    if (B == 0) return MAPF_OK;
    PlanMarkArgs p{comm_dev, stride_b, stride_t, reinterpret_cast<const long long *>(steps_dev), extra_steps_dev, T, B, N, mark_all != 0, rel_dev, slot_dev, order_dev, nact_dev,
                   cnt_dev, nag_dev, ucnt_dev};
    hipLaunchKernelGGL(plan_mark_kernel, dim3(B), dim3(128), 0, static_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

__global__ void __launch_bounds__(256) plan_totals_kernel(const int32_t *__restrict__ counts, int B, int32_t *__restrict__ totals) {
    __shared__ int s_red[256];
    const int k = blockIdx.x, tid = threadIdx.x;
    int part = 0;
    for (int b = tid; b < B; b += 256) part += counts[(size_t)k * B + b];
    s_red[tid] = part;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (tid < d) s_red[tid] += s_red[tid + d];
        __syncthreads();
    }
    if (tid == 0) totals[k] = s_red[0];
}

int mapf_plan_totals(const int32_t *counts_dev, int rows, int B, int32_t *totals_dev, void *stream) {
    if (!counts_dev || !totals_dev || rows < 1 || rows > 16 || B < 0) return MAPF_ERR_INVALID_ARG;
    hipLaunchKernelGGL(plan_totals_kernel, dim3(rows), dim3(256), 0, static_cast<hipStream_t>(stream), counts_dev, B, totals_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

static int plan_rows_launch(int T, int B, int N, int Nc, const int16_t *order_dev, const int32_t *nact_dev, const int32_t *cnt_dev, const int32_t *nag_dev,
                            const uint8_t *comm_dev, int64_t comm_stride_b, int64_t comm_stride_t, const uint16_t *hidden_dev, int hidden_is_bf16,
                            const uint16_t *obs_bf16_dev, int64_t obs_stride_b, int64_t obs_stride_t, int32_t *gidx_dev, uint8_t *comm_c_dev, uint16_t *h0_c_dev,
                            int64_t num_rows, int64_t *row_src_dev, uint16_t *obs_rows_dev, const uint8_t *dup_dev, const int32_t *ucnt_dev, int32_t *umap_dev,
                            int32_t *row_tbp_dev, int64_t fill_rows, int64_t fill_urows, void *stream) {
    if (fill_rows < 0 || fill_urows < 0) return MAPF_ERR_INVALID_ARG;
    const bool tiled = (Nc == 4 || Nc == 8) && B % (16 / Nc) == 0;  // several windows per 16-row tile (see plan_masks_kernel)
    if (T < 1 || T > MAPF_PLAN_MAX_STEPS || B < 0 || N < 1 || N > 128 || (!tiled && (Nc < 16 || Nc > 128 || (Nc & 15))) || !order_dev || !nact_dev || !cnt_dev ||
        !nag_dev || !comm_dev || !hidden_dev || !gidx_dev || !comm_c_dev || !h0_c_dev)
        return MAPF_ERR_INVALID_ARG;
    if (obs_rows_dev && (!obs_bf16_dev || !row_src_dev || num_rows < 0 || (obs_stride_b % 2) || (obs_stride_t % 2) ||
                         (reinterpret_cast<uintptr_t>(obs_bf16_dev) & 3) || (reinterpret_cast<uintptr_t>(obs_rows_dev) & 3)))
        return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(hidden_dev) & 15) || (reinterpret_cast<uintptr_t>(h0_c_dev) & 15)) return MAPF_ERR_INVALID_ARG;
    if (dup_dev && (!ucnt_dev || !umap_dev || !row_src_dev)) return MAPF_ERR_INVALID_ARG;
    if (B > 65535) return MAPF_ERR_UNSUPPORTED;
    if (B == 0) return MAPF_OK;
    PlanRowsArgs p{T, B, N, Nc, order_dev, nact_dev, cnt_dev, nag_dev, comm_dev, comm_stride_b, comm_stride_t, hidden_dev, hidden_is_bf16 != 0, obs_bf16_dev,
                   obs_stride_b, obs_stride_t, gidx_dev, comm_c_dev, h0_c_dev, reinterpret_cast<long long *>(row_src_dev), dup_dev, ucnt_dev, umap_dev,
                   row_tbp_dev, (long long)fill_rows, (long long)fill_urows};
    hipLaunchKernelGGL(plan_masks_kernel, dim3(B, T + 1), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    hipLaunchKernelGGL(plan_rows_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    if (obs_rows_dev && num_rows > 0) {
        long long blocks = (num_rows + 3) / 4;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(obs_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), obs_bf16_dev,
                           reinterpret_cast<const long long *>(row_src_dev), (long long)num_rows, obs_rows_dev);
    }
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_plan_rows(int T, int B, int N, int Nc, const int16_t *order_dev, const int32_t *nact_dev, const int32_t *cnt_dev, const int32_t *nag_dev,
                   const uint8_t *comm_dev, int64_t comm_stride_b, int64_t comm_stride_t, const uint16_t *hidden_dev, int hidden_is_bf16,
                   const uint16_t *obs_bf16_dev, int64_t obs_stride_b, int64_t obs_stride_t, int32_t *gidx_dev, uint8_t *comm_c_dev, uint16_t *h0_c_dev,
                   int64_t num_rows, int64_t *row_src_dev, uint16_t *obs_rows_dev, const uint8_t *dup_dev, const int32_t *ucnt_dev, int32_t *umap_dev,
                   int32_t *row_tbp_dev, void *stream) {
    return plan_rows_launch(T, B, N, Nc, order_dev, nact_dev, cnt_dev, nag_dev, comm_dev, comm_stride_b, comm_stride_t, hidden_dev, hidden_is_bf16, obs_bf16_dev,
                            obs_stride_b, obs_stride_t, gidx_dev, comm_c_dev, h0_c_dev, num_rows, row_src_dev, obs_rows_dev, dup_dev, ucnt_dev, umap_dev,
                            row_tbp_dev, 0, 0, stream);
}

int mapf_plan_rows_padded(int T, int B, int N, int Nc, const int16_t *order_dev, const int32_t *nact_dev, const int32_t *cnt_dev, const int32_t *nag_dev,
                          const uint8_t *comm_dev, int64_t comm_stride_b, int64_t comm_stride_t, const uint16_t *hidden_dev, int hidden_is_bf16,
                          const uint16_t *obs_bf16_dev, int64_t obs_stride_b, int64_t obs_stride_t, int32_t *gidx_dev, uint8_t *comm_c_dev, uint16_t *h0_c_dev,
                          int64_t num_rows, int64_t *row_src_dev, uint16_t *obs_rows_dev, const uint8_t *dup_dev, const int32_t *ucnt_dev, int32_t *umap_dev,
                          int32_t *row_tbp_dev, int64_t fill_rows, int64_t fill_urows, void *stream) {
    return plan_rows_launch(T, B, N, Nc, order_dev, nact_dev, cnt_dev, nag_dev, comm_dev, comm_stride_b, comm_stride_t, hidden_dev, hidden_is_bf16, obs_bf16_dev,
                            obs_stride_b, obs_stride_t, gidx_dev, comm_c_dev, h0_c_dev, num_rows, row_src_dev, obs_rows_dev, dup_dev, ucnt_dev, umap_dev,
                            row_tbp_dev, fill_rows, fill_urows, stream);
}

int mapf_dqn_head_loss(int B, int To, int Tt, const uint16_t *a0_online_dev, const uint16_t *a0_target_dev, const uint16_t *a0_online_next_dev,
                       const int64_t *bt_steps_dev, const float *steps_dev, const int64_t *action_dev, const float *reward_dev, const float *done_dev,
                       const float *weights_dev, const float *const *head_online, const float *const *head_target, float gamma, float *q_dev,
                       float *q_next_dev, float *td_dev, double *prio_dev, float *loss_dev, float *scratch_dev, uint16_t *d_a0_dev,
                       float *const *head_grads, void *stream) {
    if (B < 1 || To < 1 || Tt < To || !a0_online_dev || !a0_target_dev || !bt_steps_dev || !steps_dev || !action_dev || !reward_dev || !done_dev ||
        !weights_dev || !head_online || !head_target || !q_dev || !q_next_dev || !td_dev || !prio_dev || !loss_dev || !scratch_dev || !d_a0_dev ||
        !head_grads)
        return MAPF_ERR_INVALID_ARG;
    for (int i = 0; i < 4; ++i)
        if (!head_online[i] || !head_target[i] || !head_grads[i]) return MAPF_ERR_INVALID_ARG;
    if (B > 1536) return MAPF_ERR_UNSUPPORTED;  // head_grad keeps [B][9] floats in LDS (checked before anything is launched or written)
    HeadArgs p{};
    p.B = B, p.To = To, p.Tt = Tt;
    p.a0_on = a0_online_dev, p.a0_tg = a0_target_dev, p.a0_on2 = a0_online_next_dev;
    p.bt = reinterpret_cast<const long long *>(bt_steps_dev), p.steps = steps_dev, p.action = reinterpret_cast<const long long *>(action_dev);
    p.reward = reward_dev, p.done = done_dev, p.weights = weights_dev;
    p.w_adv = head_online[0], p.b_adv = head_online[1], p.w_st = head_online[2], p.b_st = head_online[3];
    p.tw_adv = head_target[0], p.tb_adv = head_target[1], p.tw_st = head_target[2], p.tb_st = head_target[3];
    p.gamma = gamma;
    p.q = q_dev, p.q_next = q_next_dev, p.td = td_dev, p.prio = prio_dev, p.loss = loss_dev;
    p.lossterm = scratch_dev;      // [B]
    p.dA = scratch_dev + B;        // [B][8]
    p.d_a0 = d_a0_dev;
    p.g_w_adv = head_grads[0], p.g_b_adv = head_grads[1], p.g_w_st = head_grads[2], p.g_b_st = head_grads[3];
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(head_fwd_kernel, dim3(B), dim3(256), 0, s, p);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(head_grad_kernel, dim3(1), dim3(256), (size_t)B * 9 * 4, s, p);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

static int recur_pack(const PackSeg *segs, int nseg, const float *const *bias, const int *bias_len, int nbias, uint16_t *out, float *bias_out,
                      hipStream_t s) {
    PackArgs2 p{};
    int total = 0;
    for (int i = 0; i < nseg; ++i) {
        p.seg[i] = segs[i];
        p.seg[i].dst = total;
        total += segs[i].O * segs[i].K;
    }
    if (total != MAPF_RECUR_WEIGHT_ELEMS) return MAPF_ERR_INVALID_ARG;
    p.nseg = nseg;
    p.total = total;
    p.nbias = nbias;
    for (int i = 0; i < nbias; ++i) {
        p.bias_src[i] = bias[i];
        p.bias_len[i] = bias_len[i];
    }
    p.out = out;
    p.bias_out = bias_out;
    hipLaunchKernelGGL(recur_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, p);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

/* params_dev (fp32, contiguous row-major): 0 recurrent.weight_hh [768][256] | 1 bias_ih | 2 bias_hh | 3 W_Q.weight [128][256] | 4 W_K.weight |
 * 5 W_V.weight | 6 W_Q.bias | 7 W_K.bias | 8 W_V.bias | 9 W_O.weight [64][128] | 10 update_cell.weight_ih [768][64] | 11 weight_hh [768][256] |
 * 12 bias_ih | 13 bias_hh */
int mapf_recurrent_pack(const float *const *params_dev, uint16_t *weights_dev, float *bias_dev, uint16_t *weights_t_dev, void *stream) {
    if (!params_dev || (!weights_dev && !weights_t_dev) || (weights_dev && !bias_dev)) return MAPF_ERR_INVALID_ARG;
    for (int i = 0; i < 14; ++i)
        if (!params_dev[i]) return MAPF_ERR_INVALID_ARG;
    const float *const *P = params_dev;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (weights_dev) {
        const PackSeg segs[5] = {{{P[0], nullptr, nullptr}, 768, 256, 768, 256, 0, 0, 0},
                                 {{P[3], P[4], P[5]}, 128, 256, 384, 256, 0, 0, 0},
                                 {{P[9], nullptr, nullptr}, 64, 128, 64, 128, 0, 0, 0},
                                 {{P[10], nullptr, nullptr}, 768, 64, 768, 64, 0, 0, 0},
                                 {{P[11], nullptr, nullptr}, 768, 256, 768, 256, 0, 0, 0}};
        const float *bias[7] = {P[1], P[2], P[6], P[7], P[8], P[12], P[13]};
        const int blen[7] = {768, 768, 128, 128, 128, 768, 768};
        const int rc = recur_pack(segs, 5, bias, blen, 7, weights_dev, bias_dev, s);
        if (rc != MAPF_OK) return rc;
    }
    if (weights_t_dev) {  // include/mapf_dqn.h: per gate U_ih[g]^T [64][256] | per gate U_hh[g]^T | per gate W_hh[g]^T | W_O^T [128][64] | W_qkv^T [256][384]
        PackSeg segs[11];
        int n = 0;
        for (int g = 0; g < 3; ++g) segs[n++] = PackSeg{{P[10], nullptr, nullptr}, 768, 64, 64, 256, 256 * g, 1, 0};
        for (int g = 0; g < 3; ++g) segs[n++] = PackSeg{{P[11], nullptr, nullptr}, 768, 256, 256, 256, 256 * g, 1, 0};
        for (int g = 0; g < 3; ++g) segs[n++] = PackSeg{{P[0], nullptr, nullptr}, 768, 256, 256, 256, 256 * g, 1, 0};
        segs[n++] = PackSeg{{P[9], nullptr, nullptr}, 64, 128, 128, 64, 0, 1, 0};
        segs[n++] = PackSeg{{P[3], P[4], P[5]}, 128, 256, 256, 384, 0, 1, 0};
        const int rc = recur_pack(segs, n, nullptr, nullptr, 0, weights_t_dev, nullptr, s);
        if (rc != MAPF_OK) return rc;
    }
    return MAPF_OK;
}

int mapf_recurrent_bias_grads(const float *bsum_dev, int E, float *const *grads_dev, void *stream) {
    if (!bsum_dev || E < 1 || !grads_dev) return MAPF_ERR_INVALID_ARG;
    for (int i = 0; i < 7; ++i)
        if (!grads_dev[i]) return MAPF_ERR_INVALID_ARG;
    BiasGradArgs p{bsum_dev, E, grads_dev[0], grads_dev[1], grads_dev[2], grads_dev[3], grads_dev[4], grads_dev[5], grads_dev[6]};
    hipLaunchKernelGGL(recur_bias_grads_kernel, dim3((MAPF_RECUR_BSUM_ELEMS + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

static int adam_step_impl(int64_t n, float *params_dev, float *grads_dev, float *exp_avg_dev, float *exp_avg_sq_dev, uint16_t *params_bf16_dev,
                          float *scratch_dev, float *norm_out_dev, float lr, float beta1, float beta2, float eps, int64_t step, int64_t *step_dev,
                          float max_norm, void *stream) {
    if (n < 1 || !params_dev || !grads_dev || !exp_avg_dev || !exp_avg_sq_dev || !scratch_dev || !norm_out_dev || (!step_dev && step < 1) || !(max_norm > 0.f))
        return MAPF_ERR_INVALID_ARG;
    if (step_dev && (reinterpret_cast<uintptr_t>(step_dev) & 7)) return MAPF_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, s, grads_dev, (long long)n, scratch_dev, reinterpret_cast<long long *>(step_dev));
    AdamArgs a{};
    a.n = n;
    a.p = params_dev, a.g = grads_dev, a.m = exp_avg_dev, a.v = exp_avg_sq_dev, a.p_bf16 = params_bf16_dev;
    a.partial = scratch_dev;
    a.norm_out = norm_out_dev;
    a.lr = lr, a.b1 = beta1, a.b2 = beta2, a.eps = eps, a.max_norm = max_norm;
    a.step_dev = reinterpret_cast<const long long *>(step_dev);
    if (!step_dev) {
        a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
        a.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
    }
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_adam_step(int64_t n, float *params_dev, float *grads_dev, float *exp_avg_dev, float *exp_avg_sq_dev, uint16_t *params_bf16_dev,
                   float *scratch_dev, float *norm_out_dev, float lr, float beta1, float beta2, float eps, int64_t step, float max_norm, void *stream) {
    return adam_step_impl(n, params_dev, grads_dev, exp_avg_dev, exp_avg_sq_dev, params_bf16_dev, scratch_dev, norm_out_dev, lr, beta1, beta2, eps, step,
                          nullptr, max_norm, stream);
}

int mapf_adam_step_dev(int64_t n, float *params_dev, float *grads_dev, float *exp_avg_dev, float *exp_avg_sq_dev, uint16_t *params_bf16_dev,
                       float *scratch_dev, float *norm_out_dev, float lr, float beta1, float beta2, float eps, int64_t *step_dev, float max_norm,
                       void *stream) {
    if (!step_dev) return MAPF_ERR_INVALID_ARG;
    return adam_step_impl(n, params_dev, grads_dev, exp_avg_dev, exp_avg_sq_dev, params_bf16_dev, scratch_dev, norm_out_dev, lr, beta1, beta2, eps, 0,
                          step_dev, max_norm, stream);
}

int mapf_obs_dup(int T, int To, int B, int N, const uint16_t *obs_bf16_dev, int64_t obs_stride_b, int64_t obs_stride_t, const int16_t *slot_online_dev,
                 const int16_t *slot_target_dev, const int32_t *nact_online_dev, const int32_t *nact_target_dev, uint8_t *dup_dev, int32_t *ucnt_online_dev,
                 int32_t *ucnt_target_dev, void *stream) {
    if (T < 1 || To < 1 || To > T || B < 0 || N < 1 || N > 128 || !obs_bf16_dev || !slot_online_dev || !slot_target_dev || !nact_online_dev ||
        !nact_target_dev || !dup_dev || !ucnt_online_dev || !ucnt_target_dev || (obs_stride_b % 2) || (obs_stride_t % 2) ||
        (reinterpret_cast<uintptr_t>(obs_bf16_dev) & 3))
        return MAPF_ERR_INVALID_ARG;
    if (B == 0) return MAPF_OK;
    ObsDupArgs p{T, To, B, N, obs_bf16_dev, obs_stride_b, obs_stride_t, slot_online_dev, slot_target_dev, nact_online_dev, nact_target_dev, dup_dev,
                 ucnt_online_dev, ucnt_target_dev};
    long long blocks = ((long long)B * N + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(obs_dup_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_dedup_sum(int T, int B, int Nc, int64_t rows, int row_bytes, const int32_t *gidx_dev, const int32_t *umap_dev, const int32_t *row_tbp_dev,
                   const void *d_rows_dev, void *d_unique_dev, void *stream) {
    if (T < 1 || T > 64 || B < 1 || Nc < 1 || rows < 0 || row_bytes < 16 || (row_bytes & 15) || !gidx_dev || !umap_dev || !row_tbp_dev || !d_rows_dev || !d_unique_dev)
        return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(d_rows_dev) & 15) || (reinterpret_cast<uintptr_t>(d_unique_dev) & 15)) return MAPF_ERR_INVALID_ARG;
    if (rows == 0) return MAPF_OK;
    DedupSumArgs p{T, B, Nc, row_bytes / 16, (long long)rows, gidx_dev, umap_dev, row_tbp_dev, static_cast<const uint4 *>(d_rows_dev), static_cast<uint4 *>(d_unique_dev)};
    long long blocks = (rows + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dedup_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_rows_scatter(void *rows_dev, const int32_t *idx_dev, void *dense_dev, int64_t R, int row_bytes, int to_dense, void *stream) {
    if (!rows_dev || !idx_dev || !dense_dev || R < 0 || row_bytes < 16 || (row_bytes & 15)) return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(rows_dev) & 15) || (reinterpret_cast<uintptr_t>(dense_dev) & 15)) return MAPF_ERR_INVALID_ARG;
    if (R == 0) return MAPF_OK;
    const int chunks = row_bytes / 16;
    long long blocks = (R * chunks + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (to_dense)
        hipLaunchKernelGGL(rows_to_dense_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint4 *>(rows_dev), idx_dev,
                           static_cast<uint4 *>(dense_dev), (long long)R, chunks);
    else
        hipLaunchKernelGGL(dense_to_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint4 *>(dense_dev), idx_dev,
                           static_cast<uint4 *>(rows_dev), (long long)R, chunks);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

static int zero_rows_launch(void *const *bufs_dev, const int *row_bytes, int n, int64_t first_row, const int32_t *first_row_dev, int64_t last_row,
                            void *stream) {
    if (!bufs_dev || !row_bytes || n < 0 || n > 24 || first_row < 0 || last_row < first_row) return MAPF_ERR_INVALID_ARG;
    if (n == 0 || (!first_row_dev && last_row == first_row)) return MAPF_OK;
    ZeroRowsArgs a{};
    a.first_dev = first_row_dev;
    for (int i = 0; i < n; ++i) {
        if (!bufs_dev[i] || row_bytes[i] < 16 || (row_bytes[i] & 15) || (reinterpret_cast<uintptr_t>(bufs_dev[i]) & 15)) return MAPF_ERR_INVALID_ARG;
        a.ptr[i] = bufs_dev[i];
        a.row_bytes[i] = row_bytes[i];
    }
    a.n = n;
    a.first = first_row;
    a.last = last_row;
    hipLaunchKernelGGL(zero_rows_kernel, dim3(16, n), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_zero_rows(void *const *bufs_dev, const int *row_bytes, int n, int64_t first_row, int64_t last_row, void *stream) {
    return zero_rows_launch(bufs_dev, row_bytes, n, first_row, nullptr, last_row, stream);
}

int mapf_zero_rows_from(void *const *bufs_dev, const int *row_bytes, int n, const int32_t *first_row_dev, int64_t last_row, void *stream) {
    if (!first_row_dev || (reinterpret_cast<uintptr_t>(first_row_dev) & 3) || last_row < 0) return MAPF_ERR_INVALID_ARG;
    return zero_rows_launch(bufs_dev, row_bytes, n, 0, first_row_dev, last_row, stream);
}

int mapf_to_bf16(const float *src_dev, uint16_t *dst_dev, int64_t n, void *stream) {
    if (n < 0 || !src_dev || !dst_dev) return MAPF_ERR_INVALID_ARG;
    if (n == 0) return MAPF_OK;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src_dev, dst_dev, (long long)n);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

}  // extern "C"
