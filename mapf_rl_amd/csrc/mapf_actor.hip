// mapf_actor.hip -- per-step bookkeeping of the vectorised actor loop (reference worker.py:376-414 for E lock-step environments):
// LocalBuffer.add (buffer.py:140-151), the episode-end test (worker.py:390), LocalBuffer.finish's last comm row (buffer.py:153-160,
// quirk Q8) and, behind the replay append + scenario reset, Actor.reset's rewind (worker.py:422-428).  As separate PyTorch
// index / select operations this was ~25 tiny launches per step (0.5 ms of an 11 ms iteration at config 2, most of the iteration for
// the curriculum's first levels); here it is two launches, one block per environment.  See include/mapf_replay.h.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "mapf_dqn.h"
#include "mapf_env.h"
#include "mapf_replay.h"

namespace {

struct RecordParams {
    int E, N, S, RD, RDA, A, CW;
    const float *q;           // [E][N][5]   Q-values of the step (agent 0's row is recorded: quirk Q7)
    const int64_t *actions;   // [E][N]      executed joint action
    const float *reward;      // [E][N]
    const uint16_t *hidden;   // [E][N][256] bf16 post-communication hidden state
    const int32_t *comm;      // [E][N][CW]  packed comm mask of the step
    const int32_t *bits;      // [E][RD]     bit-packed observation AFTER the step
    const uint8_t *done;      // [E]
    int64_t *t;               // [E] in: transitions recorded so far; out: + 1
    float *lb_q;              // [E][S][5]
    uint8_t *lb_act;          // [E][S]
    uint16_t *lb_rew;         // [E][S] f16
    uint16_t *lb_hid;         // [E][S][256] f16
    int32_t *lb_comm;         // [E][S+1][A][CW]
    int32_t *lb_obs;          // [E][S+1][RDA]
    uint8_t *finished;        // [E] out
    const int4 *envtab;       // optional (several levels in one launch): per environment {agents, first agent row, -, dword offset of its bit row};
                              // q / actions / reward / hidden / comm are then indexed by agent row, bits by the dword offset
};

__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }

__global__ void __launch_bounds__(256) actor_record_kernel(RecordParams p) {
    const int e = blockIdx.x, tid = threadIdx.x;
    // one read of the step counter for the whole block: thread 0 overwrites it below, and a wave that started late must not
    // see the incremented value (it would record into the next row)
    __shared__ long long s_t;
    if (tid == 0) s_t = p.t[e];
    __syncthreads();
    const long long t = s_t;
    const size_t tr = (size_t)e * p.S + t;
    int N = p.N, RD = p.RD;
    size_t row0 = (size_t)e * p.N, boff = (size_t)e * p.RD;
    if (p.envtab) {
        const int4 d = p.envtab[e];
        N = d.x, row0 = (size_t)d.y, boff = (size_t)d.w;
        RD = ((N * 486 + 31) / 32 + 3) & ~3;
    }
    // worker.py:388 -> buffer.py:140-151 (agent 0's q / action / reward / hidden; the joint comm mask; the next observation)
    p.lb_hid[tr * 256 + tid] = f32_to_f16_bits(bf16_to_f32(p.hidden[row0 * 256 + tid]));
    if (tid < 5) p.lb_q[tr * 5 + tid] = p.q[row0 * 5 + tid];
    const bool dn = p.done[e] != 0;
    const bool fin = dn || (t + 1 >= p.S);  // worker.py:390
    if (tid == 0) {
        p.lb_act[tr] = (uint8_t)p.actions[row0];
        p.lb_rew[tr] = f32_to_f16_bits(p.reward[row0]);
        p.t[e] = t + 1;
        p.finished[e] = fin ? 1 : 0;
    }
    const int crow = p.A * p.CW;
    int32_t *c_now = p.lb_comm + ((size_t)e * (p.S + 1) + t) * crow, *c_last = c_now + crow;
    const int32_t *cs = p.comm + row0 * p.CW;
    for (int i = tid; i < N * p.CW; i += 256) {
        const int32_t v = cs[i];
        c_now[i] = v;
        // the row behind a finished episode's last transition: zeros after `done` (buffer.py:124), the mask of the stale
        // observation on a time-out (quirk Q8, worker.py:399: same positions as this step's, hence this row again)
        if (fin) c_last[i] = dn ? 0 : v;
    }
    int32_t *o = p.lb_obs + ((size_t)e * (p.S + 1) + t + 1) * p.RDA;
    const int32_t *b = p.bits + boff;
    for (int i = tid; i < RD; i += 256) o[i] = b[i];
}

struct RewindParams {
    int E, N, S, RD, RDA;
    const uint8_t *finished;  // [E]
    const int32_t *bits;      // [E][RD] observation of the freshly reset scenario
    int64_t *t;               // [E]
    int32_t *lb_obs;          // [E][S+1][RDA]
    uint16_t *hidden;         // [E][N][256] bf16
    const int4 *envtab;       // optional, as in RecordParams
    uint16_t *hidden_out;     // optional: every environment's rows of `hidden` go there (zeros for the finished ones) instead of zeroing in place
    unsigned long long *tick; // optional: incremented by one (the iteration counter a replayed graph reads, mapf_actor_explore_multi)
};

__global__ void __launch_bounds__(256) actor_rewind_kernel(RewindParams p) {
    const int e = blockIdx.x, tid = threadIdx.x;
    if (p.tick != nullptr && e == 0 && tid == 0) *p.tick += 1ull;
    if (p.finished[e] == 0 && p.hidden_out == nullptr) return;
    int N = p.N, RD = p.RD;
    size_t row0 = (size_t)e * p.N, boff = (size_t)e * p.RD;
    if (p.envtab) {
        const int4 d = p.envtab[e];
        N = d.x, row0 = (size_t)d.y, boff = (size_t)d.w;
        RD = ((N * 486 + 31) / 32 + 3) & ~3;
    }
    if (p.finished[e] == 0) {  // a running episode: its new hidden states become the next iteration's input
        const uint4 *src = reinterpret_cast<const uint4 *>(p.hidden + row0 * 256);
        uint4 *dst = reinterpret_cast<uint4 *>(p.hidden_out + row0 * 256);
        for (int i = tid; i < N * 32; i += 256) dst[i] = src[i];
        return;
    }
    if (tid == 0) p.t[e] = 0;
    int32_t *o = p.lb_obs + (size_t)e * (p.S + 1) * p.RDA;
    const int32_t *b = p.bits + boff;
    for (int i = tid; i < RD; i += 256) o[i] = b[i];
    uint4 *h = reinterpret_cast<uint4 *>((p.hidden_out ? p.hidden_out : p.hidden) + row0 * 256);  // model.reset(): GRUCell(x, None) == zero state
    for (int i = tid; i < N * 32; i += 256) h[i] = make_uint4(0, 0, 0, 0);
}

// episode counter + outcomes of the statistics-bearing environments (actor id >= 10, worker.py:74) in environment order into a
// ring log of `log_size` entries (+ 1 dump slot); one block
__global__ void __launch_bounds__(1024) actor_log_kernel(int E, const uint8_t *finished, const uint8_t *done, const uint8_t *stat_mask,
                                                        uint8_t *log, int log_size, int64_t *counters /* {episodes, logged} */) {
    __shared__ int s_cnt[1024], s_fin[1024];
    const int tid = threadIdx.x, nth = blockDim.x;
    const int per = (E + nth - 1) / nth, e0 = tid * per, e1 = min(E, e0 + per);
    int cnt = 0, fin = 0;
    for (int e = e0; e < e1; ++e) {
        fin += finished[e] != 0;
        cnt += (finished[e] != 0 && stat_mask[e] != 0);
    }
    s_cnt[tid] = cnt;
    s_fin[tid] = fin;
    __syncthreads();
    for (int d = 1; d < nth; d <<= 1) {
        const int v = tid >= d ? s_cnt[tid - d] : 0, f = tid >= d ? s_fin[tid - d] : 0;
        __syncthreads();
        s_cnt[tid] += v;
        s_fin[tid] += f;
        __syncthreads();
    }
    const long long base = counters[1];
    long long pos = base + s_cnt[tid] - cnt;
    for (int e = e0; e < e1; ++e)
        if (finished[e] != 0 && stat_mask[e] != 0) log[(pos++) % log_size] = done[e] != 0;
    __syncthreads();
    if (tid == 0) {
        counters[0] += s_fin[nth - 1];
        counters[1] = base + s_cnt[nth - 1];
    }
}

// Which agents see something new: a row (486-byte observation) that differs from the copy of the step before is appended to the
// list (order does not matter: every row is encoded independently), copied to the packed buffer and the copy is refreshed.
// One wavefront per CHUNK of 8 row pairs: two rows are 972 bytes = 243 dwords starting on a 4-byte boundary (a single row of odd
// index does not), dword 121 straddles them.  The chunk's changed rows get their list slots with ONE atomic (first version: one per
// pair -- 80 k atomics on one address per call when most agents move, ~0.8 ms), and the packed rows have a 488-byte stride
// (MAPF_ENC_PACKED_OBS_STRIDE) so that they are written with dword stores whatever the slot's parity (first version: 486-byte stride,
// 2-byte stores).
__global__ void zero_words_kernel(uint32_t *p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}

constexpr int CH_PAIRS = 8;
__global__ void __launch_bounds__(256) obs_changed_kernel(const uint32_t *__restrict__ obs, uint32_t *__restrict__ prev, long long rows,
                                                          int32_t *__restrict__ list, int32_t *__restrict__ count, uint32_t *__restrict__ packed) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long pairs = (rows + 1) >> 1;
    const long long chunks = (pairs + CH_PAIRS - 1) / CH_PAIRS;
    // The list slots of a WORKGROUP's four chunks come from one atomic (round 4; one per chunk was 10 k returning atomics on one address
    // per call at 163,840 rows when most agents move -- most of the kernel's time): the loop is uniform over the workgroup.
    __shared__ int s_cnt[4], s_base;
    for (long long cb = (long long)blockIdx.x * 4; cb < chunks; cb += (long long)gridDim.x * 4) {
        const long long c = cb + w;
        uint32_t cm = 0;  // bit 2 i: first row of pair i changed, bit 2 i + 1: its second row
#pragma unroll
        for (int i = 0; i < CH_PAIRS; ++i) {
            const long long q = c * CH_PAIRS + i;
            if (c >= chunks || q >= pairs) break;  // (wave-uniform)
            const bool two = 2 * q + 1 < rows;
            const int nd = two ? 243 : 122;  // a last single row: 121.5 dwords (its buffer ends on a 2-byte boundary: handled below)
            const uint32_t *cur = obs + q * 243;
            uint32_t *old = prev + q * 243;
            uint32_t v[4], o[4];
            bool d0 = false, d1 = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int d = lane + 64 * k;
                v[k] = o[k] = 0u;
                if (d < nd) {
                    if (!two && d == 121) {  // only the low half exists
                        v[k] = reinterpret_cast<const uint16_t *>(cur)[242];
                        o[k] = reinterpret_cast<const uint16_t *>(old)[242];
                    } else {
                        v[k] = cur[d];
                        o[k] = old[d];
                    }
                    const uint32_t x = v[k] ^ o[k];
                    d0 |= d < 121 ? x != 0u : (d == 121 && (x & 0xFFFFu) != 0u);
                    d1 |= d > 121 ? x != 0u : (d == 121 && (x >> 16) != 0u);
                }
            }
            const bool c0 = __ballot(d0) != 0ull, c1 = two && __ballot(d1) != 0ull;
            cm |= ((uint32_t)c0 | ((uint32_t)c1 << 1)) << (2 * i);
            if (c0 | c1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int d = lane + 64 * k;
                    if (d < nd && v[k] != o[k]) {
                        if (!two && d == 121) reinterpret_cast<uint16_t *>(old)[242] = (uint16_t)v[k];
                        else old[d] = v[k];
                    }
                }
            }
        }
        if (lane == 0) s_cnt[w] = __popc(cm);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            s_base = total ? atomicAdd(count, total) : 0;
        }
        __syncthreads();
        int base = s_base;
        for (int k = 0; k < w; ++k) base += s_cnt[k];
        __syncthreads();  // (s_cnt / s_base are rewritten by the next pass)
        if (cm == 0u) continue;
        // list entries: lane r (< 16) owns row r of the chunk
        if (lane < 2 * CH_PAIRS && ((cm >> lane) & 1u)) list[base + __popc(cm & ((1u << lane) - 1u))] = (int32_t)(2 * c * CH_PAIRS + lane);
        if (packed) {  // the changed rows, in list order, 122 dwords each (the last 2 bytes are padding): what mapf_encoder_forward_rows reads
            for (uint32_t m = cm; m != 0u; m &= m - 1u) {
                const int r = __ffs((int)m) - 1;
                const long long row = 2 * c * CH_PAIRS + r;
                uint32_t *dst = packed + (long long)(base + __popc(cm & ((1u << r) - 1u))) * (MAPF_ENC_PACKED_OBS_STRIDE / 4);
                const uint16_t *c16 = reinterpret_cast<const uint16_t *>(obs) + row * 243;  // (the row was just read: cache hits)
                for (int j = lane; j < 122; j += 64) {
                    uint32_t w;
                    if (!(r & 1)) {
                        w = j < 121 ? reinterpret_cast<const uint32_t *>(c16)[j] : (uint32_t)c16[242];
                    } else {
                        w = (uint32_t)c16[2 * j];
                        if (j < 121) w |= (uint32_t)c16[2 * j + 1] << 16;
                    }
                    dst[j] = w;
                }
            }
        }
    }
}

// Exploration of one actor iteration (reference worker.py:380-382: with probability epsilon, agent 0 of an environment takes a uniform
// random action; the other agents act greedily) + the int8 copy of the joint action the environment step reads + a copy of the greedy
// actions: one launch for what was rand / randint / compare / where / index-assign / cast / clone.  Own counter-based generator
// (splitmix64 of seed, iteration and environment): the reference's numpy stream is not reproducible here anyway, epsilon is.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void __launch_bounds__(256) actor_explore_kernel(long long total, int N, int64_t *__restrict__ actions, int64_t *__restrict__ policy,
                                                            int8_t *__restrict__ act8, const double *__restrict__ eps, uint64_t seed, uint64_t counter,
                                                            const unsigned long long *__restrict__ tick) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    if (tick) counter += (uint64_t)tick[0];  // (the iteration counter in device memory: nothing changes on the host from launch to launch)
    const long long e = idx / N;
    int64_t a = actions[idx];
    if (policy) policy[idx] = a;
    if (idx - e * N == 0) {
        const uint64_t r0 = splitmix64(seed ^ splitmix64(counter * 0xD6E8FEB86659FD93ull + (uint64_t)e));
        const double u = (double)(r0 >> 11) * (1.0 / 9007199254740992.0);  // [0, 1)
        if (u < eps[e]) {
            a = (int64_t)(splitmix64(r0) % 5ull);
            actions[idx] = a;
        }
    }
    act8[idx] = (int8_t)a;
}

// the same log for several levels in one launch: block l owns the environments [start[l], start[l + 1]) and its own log / counters
struct LogLevels {
    int n;
    int start[17];
    uint8_t *log[16];
    int64_t *counters[16];
    int log_size;
};
__global__ void __launch_bounds__(1024) actor_log_multi_kernel(LogLevels lv, const uint8_t *finished, const uint8_t *done, const uint8_t *stat_mask) {
    __shared__ int s_cnt[1024], s_fin[1024];
    const int l = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const int E0 = lv.start[l], E = lv.start[l + 1] - E0;
    uint8_t *log = lv.log[l];
    int64_t *counters = lv.counters[l];
    finished += E0, done += E0, stat_mask += E0;
    const int per = (E + nth - 1) / nth, e0 = tid * per, e1 = min(E, e0 + per);
    int cnt = 0, fin = 0;
    for (int e = e0; e < e1; ++e) {
        fin += finished[e] != 0;
        cnt += (finished[e] != 0 && stat_mask[e] != 0);
    }
    s_cnt[tid] = cnt;
    s_fin[tid] = fin;
    __syncthreads();
    for (int d = 1; d < nth; d <<= 1) {
        const int v = tid >= d ? s_cnt[tid - d] : 0, f = tid >= d ? s_fin[tid - d] : 0;
        __syncthreads();
        s_cnt[tid] += v;
        s_fin[tid] += f;
        __syncthreads();
    }
    const long long base = counters[1];
    long long pos = base + s_cnt[tid] - cnt;
    for (int e = e0; e < e1; ++e)
        if (finished[e] != 0 && stat_mask[e] != 0) log[(pos++) % lv.log_size] = done[e] != 0;
    __syncthreads();
    if (tid == 0) {
        counters[0] += s_fin[nth - 1];
        counters[1] = base + s_cnt[nth - 1];
    }
}

// exploration for environments of different agent counts: one thread per environment (agent 0 explores, worker.py:380-382; the other
// agents' greedy actions are copied).  aux int64 [E][3] = {seed of the environment's level, base counter, index within the level}: the
// draws of (seed, base + tick, index) are the ones actor_explore_kernel makes for that level alone.
__global__ void __launch_bounds__(256) actor_explore_multi_kernel(int E, const int4 *__restrict__ envtab, const long long *__restrict__ aux,
                                                                  int64_t *__restrict__ actions, int64_t *__restrict__ policy, int8_t *__restrict__ act8,
                                                                  const double *__restrict__ eps, const unsigned long long *__restrict__ tick) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int4 d = envtab[e];
    const long long row0 = d.y;
    const uint64_t seed = (uint64_t)aux[3 * e], counter = (uint64_t)aux[3 * e + 1] + (tick ? (uint64_t)tick[0] : 0ull), el = (uint64_t)aux[3 * e + 2];
    for (int a = 0; a < d.x; ++a) {
        int64_t act = actions[row0 + a];
        if (policy) policy[row0 + a] = act;
        if (a == 0) {
            const uint64_t r0 = splitmix64(seed ^ splitmix64(counter * 0xD6E8FEB86659FD93ull + el));
            const double u = (double)(r0 >> 11) * (1.0 / 9007199254740992.0);  // [0, 1)
            if (u < eps[e]) {
                act = (int64_t)(splitmix64(r0) % 5ull);
                actions[row0] = act;
            }
        }
        act8[row0 + a] = (int8_t)act;
    }
}

}  // namespace

extern "C" {

// ---- several levels (environments of different agent counts, their agent rows back to back) in one launch each: the per-environment
// table envtab int32 [E][4] = {agents, first agent row, byte offset of the comm mask (mapf_comm_mask_multi / mapf_recurrent_infer_multi),
// dword offset of the bit-packed observation row}; see include/mapf_replay.h ----
int mapf_actor_explore_multi(int num_envs, const int32_t *envtab_dev, const int64_t *aux_dev, int64_t *actions_dev, int64_t *policy_dev, int8_t *act8_dev,
                             const double *eps_dev, const uint64_t *tick_dev, void *stream) {
    if (num_envs < 1 || !envtab_dev || !aux_dev || !actions_dev || !act8_dev || !eps_dev || (reinterpret_cast<uintptr_t>(envtab_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(tick_dev) & 7))
        return MAPF_ERR_INVALID_ARG;
    hipLaunchKernelGGL(actor_explore_multi_kernel, dim3((num_envs + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), num_envs,
                       reinterpret_cast<const int4 *>(envtab_dev), reinterpret_cast<const long long *>(aux_dev), actions_dev, policy_dev, act8_dev, eps_dev,
                       reinterpret_cast<const unsigned long long *>(tick_dev));
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_actor_record_multi(int num_envs, int local_steps, int row_dwords, int max_agents, const int32_t *envtab_dev, const float *q_dev,
                            const int64_t *actions_dev, const float *reward_dev, const uint16_t *hidden_dev, const int32_t *comm_dev,
                            const int32_t *obs_bits_dev, const uint8_t *done_dev, int64_t *t_dev, float *lb_q_dev, uint8_t *lb_act_dev, uint16_t *lb_rew_dev,
                            uint16_t *lb_hid_dev, int32_t *lb_comm_dev, int32_t *lb_obs_dev, uint8_t *finished_dev, void *stream) {
    if (num_envs < 1 || max_agents < 1 || local_steps < 1 || local_steps > MAPF_REPLAY_MAX_STEPS || row_dwords < 1 || !envtab_dev ||
        (reinterpret_cast<uintptr_t>(envtab_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    if (!q_dev || !actions_dev || !reward_dev || !hidden_dev || !comm_dev || !obs_bits_dev || !done_dev || !t_dev || !lb_q_dev || !lb_act_dev ||
        !lb_rew_dev || !lb_hid_dev || !lb_comm_dev || !lb_obs_dev || !finished_dev)
        return MAPF_ERR_INVALID_ARG;
    RecordParams p{num_envs, 0, local_steps, 0, row_dwords, max_agents, (max_agents + 31) / 32, q_dev, actions_dev, reward_dev,
                   hidden_dev, comm_dev, obs_bits_dev, done_dev, t_dev, lb_q_dev, lb_act_dev, lb_rew_dev, lb_hid_dev, lb_comm_dev, lb_obs_dev,
                   finished_dev, reinterpret_cast<const int4 *>(envtab_dev)};
    hipLaunchKernelGGL(actor_record_kernel, dim3(num_envs), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_actor_rewind_multi(int num_envs, int local_steps, int row_dwords, const int32_t *envtab_dev, const uint8_t *finished_dev,
                            const int32_t *obs_bits_dev, int64_t *t_dev, int32_t *lb_obs_dev, uint16_t *hidden_dev, uint16_t *hidden_out_dev,
                            uint64_t *tick_dev, void *stream) {
    if (num_envs < 1 || local_steps < 1 || row_dwords < 1 || !envtab_dev || !finished_dev || !obs_bits_dev || !t_dev || !lb_obs_dev || !hidden_dev ||
        (reinterpret_cast<uintptr_t>(hidden_dev) & 15) || (reinterpret_cast<uintptr_t>(envtab_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(hidden_out_dev) & 15) || (reinterpret_cast<uintptr_t>(tick_dev) & 7) || hidden_out_dev == hidden_dev)
        return MAPF_ERR_INVALID_ARG;
    RewindParams p{num_envs, 0, local_steps, 0, row_dwords, finished_dev, obs_bits_dev, t_dev, lb_obs_dev, hidden_dev,
                   reinterpret_cast<const int4 *>(envtab_dev), hidden_out_dev, reinterpret_cast<unsigned long long *>(tick_dev)};
    hipLaunchKernelGGL(actor_rewind_kernel, dim3(num_envs), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_actor_log_multi(int num_levels, const int32_t *level_start, uint8_t *const *log_dev, int64_t *const *counters_dev, int log_size,
                         const uint8_t *finished_dev, const uint8_t *done_dev, const uint8_t *stat_mask_dev, void *stream) {
    if (num_levels < 1 || num_levels > 16 || !level_start || !log_dev || !counters_dev || log_size < 1 || !finished_dev || !done_dev || !stat_mask_dev)
        return MAPF_ERR_INVALID_ARG;
    LogLevels lv{};
    lv.n = num_levels;
    lv.log_size = log_size;
    for (int l = 0; l <= num_levels; ++l) lv.start[l] = level_start[l];
    for (int l = 0; l < num_levels; ++l) {
        if (!log_dev[l] || !counters_dev[l] || lv.start[l + 1] <= lv.start[l]) return MAPF_ERR_INVALID_ARG;
        lv.log[l] = log_dev[l];
        lv.counters[l] = counters_dev[l];
    }
    hipLaunchKernelGGL(actor_log_multi_kernel, dim3(num_levels), dim3(1024), 0, static_cast<hipStream_t>(stream), lv, finished_dev, done_dev, stat_mask_dev);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_actor_explore(int num_envs, int num_agents, int64_t *actions_dev, int64_t *policy_dev, int8_t *act8_dev, const double *eps_dev,
                       uint64_t seed, uint64_t counter, void *stream) {
    if (num_envs < 1 || num_agents < 1 || !actions_dev || !act8_dev || !eps_dev) return MAPF_ERR_INVALID_ARG;
    const long long total = (long long)num_envs * num_agents;
    hipLaunchKernelGGL(actor_explore_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), total, num_agents,
                       actions_dev, policy_dev, act8_dev, eps_dev, (uint64_t)seed, (uint64_t)counter, (const unsigned long long *)nullptr);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_actor_explore_dev(int num_envs, int num_agents, int64_t *actions_dev, int64_t *policy_dev, int8_t *act8_dev, const double *eps_dev,
                           uint64_t seed, uint64_t counter, const uint64_t *tick_dev, void *stream) {
    if (num_envs < 1 || num_agents < 1 || !actions_dev || !act8_dev || !eps_dev || !tick_dev || (reinterpret_cast<uintptr_t>(tick_dev) & 7))
        return MAPF_ERR_INVALID_ARG;
    const long long total = (long long)num_envs * num_agents;
    hipLaunchKernelGGL(actor_explore_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), total, num_agents,
                       actions_dev, policy_dev, act8_dev, eps_dev, (uint64_t)seed, (uint64_t)counter,
                       reinterpret_cast<const unsigned long long *>(tick_dev));
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_actor_record(int num_envs, int num_agents, int local_steps, int env_row_dwords, int row_dwords, int max_agents, const float *q_dev,
                      const int64_t *actions_dev, const float *reward_dev, const uint16_t *hidden_dev, const int32_t *comm_dev,
                      const int32_t *obs_bits_dev, const uint8_t *done_dev, int64_t *t_dev, float *lb_q_dev, uint8_t *lb_act_dev,
                      uint16_t *lb_rew_dev, uint16_t *lb_hid_dev, int32_t *lb_comm_dev, int32_t *lb_obs_dev, uint8_t *finished_dev, void *stream) {
    if (num_envs < 1 || num_agents < 1 || max_agents < num_agents || local_steps < 1 || local_steps > MAPF_REPLAY_MAX_STEPS ||
        env_row_dwords < 1 || row_dwords < env_row_dwords)
        return MAPF_ERR_INVALID_ARG;
    if (!q_dev || !actions_dev || !reward_dev || !hidden_dev || !comm_dev || !obs_bits_dev || !done_dev || !t_dev || !lb_q_dev || !lb_act_dev ||
        !lb_rew_dev || !lb_hid_dev || !lb_comm_dev || !lb_obs_dev || !finished_dev)
        return MAPF_ERR_INVALID_ARG;
    RecordParams p{num_envs, num_agents, local_steps, env_row_dwords, row_dwords, max_agents, (max_agents + 31) / 32, q_dev, actions_dev, reward_dev,
                   hidden_dev, comm_dev, obs_bits_dev, done_dev, t_dev, lb_q_dev, lb_act_dev, lb_rew_dev, lb_hid_dev, lb_comm_dev, lb_obs_dev,
                   finished_dev, nullptr};
    hipLaunchKernelGGL(actor_record_kernel, dim3(num_envs), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_actor_rewind(int num_envs, int num_agents, int local_steps, int env_row_dwords, int row_dwords, const uint8_t *finished_dev,
                      const int32_t *obs_bits_dev, int64_t *t_dev, int32_t *lb_obs_dev, uint16_t *hidden_dev, void *stream) {
    if (num_envs < 1 || num_agents < 1 || local_steps < 1 || env_row_dwords < 1 || row_dwords < env_row_dwords || !finished_dev || !obs_bits_dev ||
        !t_dev || !lb_obs_dev || !hidden_dev || (reinterpret_cast<uintptr_t>(hidden_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    RewindParams p{num_envs, num_agents, local_steps, env_row_dwords, row_dwords, finished_dev, obs_bits_dev, t_dev, lb_obs_dev, hidden_dev, nullptr};
    hipLaunchKernelGGL(actor_rewind_kernel, dim3(num_envs), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_obs_changed(const uint8_t *obs_dev, uint8_t *prev_dev, int64_t rows, int32_t *list_dev, int32_t *count_dev, uint8_t *packed_dev, void *stream) {
    if (rows < 0 || !obs_dev || !prev_dev || !list_dev || !count_dev || (reinterpret_cast<uintptr_t>(obs_dev) & 3) || (reinterpret_cast<uintptr_t>(prev_dev) & 3) ||
        (reinterpret_cast<uintptr_t>(packed_dev) & 3))
        return MAPF_ERR_INVALID_ARG;
    if (rows == 0) return MAPF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // (a kernel, not hipMemsetAsync: this call is captured into the actors' HIP graph, and on this runtime a graph that holds a memset
    // node faulted at a later replay once the learner's own launches had run in between -- tools/micro/graph_gemm_probe.py)
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<uint32_t *>(count_dev), 1);
    long long blocks = (((rows + 1) / 2 + CH_PAIRS - 1) / CH_PAIRS + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(obs_changed_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint32_t *>(obs_dev),
                       reinterpret_cast<uint32_t *>(prev_dev), (long long)rows, list_dev, count_dev, reinterpret_cast<uint32_t *>(packed_dev));
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_actor_log(int num_envs, const uint8_t *finished_dev, const uint8_t *done_dev, const uint8_t *stat_mask_dev, uint8_t *log_dev,
                   int log_size, int64_t *counters_dev, void *stream) {
    if (num_envs < 1 || log_size < 1 || !finished_dev || !done_dev || !stat_mask_dev || !log_dev || !counters_dev) return MAPF_ERR_INVALID_ARG;
    hipLaunchKernelGGL(actor_log_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), num_envs, finished_dev, done_dev, stat_mask_dev,
                       log_dev, log_size, counters_dev);
    return hipGetLastError() == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_actor_iteration_tail(const mapf_actor_state *a, void *env_v, mapf_replay_t *replay, int64_t *actions_dev, const float *q_dev,
                              uint16_t *hidden_dev, const int32_t *comm_dev, uint64_t explore_seed, uint64_t explore_counter, float density,
                              uint64_t scenario_seed, void *stream) {
    if (!a || !env_v || !actions_dev || !q_dev || !hidden_dev || !comm_dev) return MAPF_ERR_INVALID_ARG;
    mapf_env_t *env = static_cast<mapf_env_t *>(env_v);
    const int E = a->num_envs, N = a->num_agents, S = a->local_steps;
    int rc = mapf_actor_explore(E, N, actions_dev, a->policy_actions, a->act8, a->eps, explore_seed, explore_counter, stream);
    if (rc != MAPF_OK) return rc;
    rc = mapf_step(env, a->act8, a->obs, reinterpret_cast<uint32_t *>(a->obs_bits), a->pos, a->reward_class, a->reward, a->done, stream);
    if (rc != MAPF_OK) return rc;
    rc = mapf_actor_record(E, N, S, a->env_row_dwords, a->row_dwords, a->max_agents, q_dev, actions_dev, a->reward, hidden_dev, comm_dev, a->obs_bits,
                           a->done, a->t, a->lb_q, a->lb_act, a->lb_rew, a->lb_hid, a->lb_comm, a->lb_obs, a->finished, stream);
    if (rc != MAPF_OK) return rc;
    if (replay) {
        rc = mapf_replay_add_many(replay, E, N, S, a->finished, a->t, a->done, reinterpret_cast<const uint32_t *>(a->lb_obs),
                                  reinterpret_cast<const uint32_t *>(a->lb_comm), a->lb_act, a->lb_rew, a->lb_hid, a->lb_q, stream);
        if (rc != MAPF_OK) return rc;
    }
    rc = mapf_actor_log(E, a->finished, a->done, a->stat_mask, a->stat_log, a->log_size, a->counters, stream);
    if (rc != MAPF_OK) return rc;
    rc = mapf_reset_envs(env, a->finished, density, scenario_seed, stream);
    if (rc != MAPF_OK) return rc;
    rc = mapf_observe_masked(env, a->finished, a->obs, reinterpret_cast<uint32_t *>(a->obs_bits), a->pos, stream);
    if (rc != MAPF_OK) return rc;
    return mapf_actor_rewind(E, N, S, a->env_row_dwords, a->row_dwords, a->finished, a->obs_bits, a->t, a->lb_obs, hidden_dev, stream);
}

}  // extern "C"
