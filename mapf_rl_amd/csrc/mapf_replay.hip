// mapf_replay.hip -- on-device prioritized episode replay (see include/mapf_replay.h).
// Replaces reference buffer.py:16-105 (SumTree) and worker.py:71-203 (GlobalBuffer.add / sample_batch /
// update_priorities).  The tree is f64 like the reference; this file must be compiled with
// -ffp-contract=off so that k*interval + u and left + right round exactly like numpy.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <new>

#include "mapf_env.h"
#include "mapf_replay.h"

namespace {

constexpr int kMaxSteps = MAPF_REPLAY_MAX_STEPS;
constexpr int kRows = kMaxSteps + 1;  // observation / comm rows per slot
constexpr int kWindow = MAPF_REPLAY_WINDOW;
constexpr int kObsBitsPerAgent = 6 * 81;
constexpr double kAlpha = 0.6;  // config.prioritized_replay_alpha (config.py:42)

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            std::fprintf(stderr, "mapf_replay: %s failed: %s\n", #expr, hipGetErrorString(_e)); \
            return MAPF_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
    }
    ~DeviceGuard() {
        if (ok && prev >= 0) (void)hipSetDevice(prev);
    }
};

// One block.  Entry i is dropped when `stale` says its slot was overwritten (worker.py:192-201) or when a
// later entry carries the same index (numpy's `tree[idx] = p` keeps the last one).
__global__ void __launch_bounds__(1024) tree_update_kernel(double *tree, long long capacity, int layers,
                                                           const int64_t *idx, const double *pri, int n, double alpha,
                                                           const int64_t *state, const int64_t *old_ptr_dev) {
    const int tid = threadIdx.x;
    // worker.py:190-201: entries whose slot was overwritten between the sample (ring pointer old_ptr) and now (ring pointer ptr)
    // are dropped.  Both pointers live on the device (the ring advances without the host: mapf_replay_add_many)
    int stale_mode = 0;
    long long lo = 0, hi = 0;
    if (old_ptr_dev != nullptr) {
        const long long ptr = state[0], old = *old_ptr_dev;
        lo = old * kMaxSteps;
        hi = ptr * kMaxSteps;
        stale_mode = ptr > old ? 1 : (ptr < old ? 2 : 0);
    }
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + tid;
        bool active = i < n;
        long long node = 0;
        if (active) {
            const long long id = idx[i];
            if (id < 0 || id >= capacity) active = false;
            // stale_mode 1: keep (id < lo) | (id >= hi);  2: keep (id < lo) & (id >= hi)
            if (stale_mode == 1 && !(id < lo || id >= hi)) active = false;
            if (stale_mode == 2 && !(id < lo && id >= hi)) active = false;
            if (active) {
                for (int j = i + 1; j < n; ++j)
                    if (idx[j] == id) {
                        active = false;
                        break;
                    }
            }
            if (active) {
                node = id + capacity - 1;
                const double p = pri[i];
                tree[node] = alpha > 0.0 ? pow(p, alpha) : p;
            }
        }
        __syncthreads();
        for (int l = 1; l < layers; ++l) {  // buffer.py:99-102
            if (active) {
                node = (node - 1) / 2;
                tree[node] = tree[2 * node + 1] + tree[2 * node + 2];
            }
            __syncthreads();
        }
    }
}

__global__ void tree_sample_kernel(const double *tree, long long capacity, int layers, const double *uniforms, int n,
                                   int unit_uniforms, int64_t *idx_out, double *pri_out, int64_t *state, int64_t *old_ptr_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0 && old_ptr_out != nullptr) *old_ptr_out = state[0];  // worker.py:182: the ring pointer at sample time
    if (k >= n) return;
    const double sum = tree[0];
    if (!(sum > 0.0)) {
        // an empty ring (the reference's batch_sample would assert, buffer.py:75-76): no NaN weights downstream -- leaf 0 with
        // priority 1 -- and a sticky flag that the next mapf_replay_state() turns into MAPF_ERR_NOT_READY
        if (k == 0) state[4] = 1;
        idx_out[k] = 0;
        pri_out[k] = 1.0;
        return;
    }
    const double interval = sum / (double)n;              // buffer.py:58
    // buffer.py:60: np.random.uniform(0, interval) = interval * U(0,1); unit_uniforms: the caller passes U(0,1) and the scaling
    // happens here (same f64 product), so that it needs neither the tree's root nor a host round trip
    const double u = unit_uniforms ? uniforms[k] * interval : uniforms[k];
    double prefix = (double)k * interval + u;             // np.arange(0, sum, interval)[k] = k*interval
    if (k == 0 && prefix == 0.0) prefix = 1e-5;           // buffer.py:61-62
    long long node = 0;
    for (int l = 1; l < layers; ++l) {                    // buffer.py:66-70
        const double p = tree[2 * node + 1];
        if (prefix <= p) {
            node = 2 * node + 1;
        } else {
            node = 2 * node + 2;
            prefix = prefix - tree[node - 1];
        }
        if (prefix == 0.0) prefix = 1e-5;
    }
    idx_out[k] = node - (capacity - 1);
    pri_out[k] = tree[node];
}

struct GatherParams {
    int A, RD, CW, n;
    const uint32_t *obs_bits;   // [cap*257][RD]
    const uint32_t *comm_bits;  // [cap*257][A][CW]
    const uint8_t *act;         // [cap*256]
    const uint16_t *rew;        // [cap*256] f16
    const uint16_t *hid;        // [cap*256][256] f16
    const uint8_t *done_buf;    // [cap]
    const int32_t *size_buf;    // [cap]
    const int32_t *nag_buf;     // [cap]
    const int64_t *idx;         // [n]
    uint16_t *obs;              // [18][n][A][486] bf16
    uint8_t *comm;              // [18][n][A][A]
    uint16_t *hidden;           // [n*A][256] f16
    int64_t *action;
    float *reward, *done, *steps;
    int64_t *bt_steps;
};

__device__ __forceinline__ float half_bits_to_float(uint16_t h) {
    // exact f16 -> f32 (normal, subnormal, zero, inf/nan)
    const uint32_t s = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t out;
    if (e == 0) {
        if (m == 0) out = s;
        else {
            e = 127 - 15 + 1;
            while (!(m & 0x400u)) {
                m <<= 1;
                --e;
            }
            out = s | (e << 23) | ((m & 0x3FFu) << 13);
        }
    } else if (e == 31) out = s | 0x7F800000u | (m << 13);
    else out = s | ((e + 127 - 15) << 23) | (m << 13);
    return __uint_as_float(out);
}

// grid (18, n): block (t, b) expands window row t of sample b (worker.py:118-162).
__global__ void __launch_bounds__(256) gather_kernel(GatherParams p) {
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const long long idx = p.idx[b];
    const long long g = idx / kMaxSteps, l = idx % kMaxSteps;
    const int size = p.size_buf[g];
    // worker.py:122; a leaf past the episode end has priority 0 and is never drawn (the reference asserts it,
    // worker.py:120) -- clamp anyway so that a corrupted tree cannot produce negative window lengths
    const int steps = (int)max(0ll, min((long long)MAPF_REPLAY_FWD_STEPS, (long long)size - l));
    const long long first_row = g * kRows;
    const long long lo = (l < MAPF_REPLAY_BT_STEPS - 1) ? first_row : idx + g + 1 - MAPF_REPLAY_BT_STEPS;  // :124-136
    const long long hi = idx + g + 1 + steps;
    const long long row = lo + t;
    const bool valid = row < hi;                                                            // :139-142 zero padding
    const int A = p.A;
    // observation bits -> bf16 (0x3F80 = 1.0)
    const int nbits = A * kObsBitsPerAgent;
    uint16_t *out = p.obs + ((size_t)t * p.n + b) * nbits;  // time-major: the recurrence walks steps, see mapf_replay.h
    const uint32_t *src = p.obs_bits + (size_t)row * p.RD;
    for (int q = tid; q < (nbits >> 3); q += blockDim.x) {  // 8 bits -> 8 bf16 (16 bytes)
        uint32_t bits = 0;
        if (valid) bits = (src[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
        uint4 v;
        v.x = ((bits & 1u) ? 0x3F80u : 0u) | ((bits & 2u) ? 0x3F800000u : 0u);
        v.y = ((bits & 4u) ? 0x3F80u : 0u) | ((bits & 8u) ? 0x3F800000u : 0u);
        v.z = ((bits & 16u) ? 0x3F80u : 0u) | ((bits & 32u) ? 0x3F800000u : 0u);
        v.w = ((bits & 64u) ? 0x3F80u : 0u) | ((bits & 128u) ? 0x3F800000u : 0u);
        *reinterpret_cast<uint4 *>(out + (size_t)q * 8) = v;
    }
    for (int q = (nbits & ~7) + tid; q < nbits; q += blockDim.x) {  // tail (A*486 not a multiple of 8)
        uint32_t bit = valid ? (src[q >> 5] >> (q & 31)) & 1u : 0u;
        out[q] = bit ? 0x3F80u : 0u;
    }
    // comm mask row
    uint8_t *cm = p.comm + ((size_t)t * p.n + b) * A * A;
    const uint32_t *cs = p.comm_bits + (size_t)row * A * p.CW;
    for (int q = tid; q < A * A; q += blockDim.x) {
        const int a = q / A, j = q - a * A;
        cm[q] = valid ? (uint8_t)((cs[a * p.CW + (j >> 5)] >> (j & 31)) & 1u) : 0;
    }
    if (t == 0) {
        // initial hidden: zeros when l <= 15, else the state stored 16 transitions earlier (:127,132,137),
        // broadcast to the episode's agents (quirk Q4), zero for padded agents
        const bool has_h = l > MAPF_REPLAY_BT_STEPS - 1;
        const int nag = p.nag_buf[g];
        const uint16_t *hs = p.hid + (size_t)(idx - MAPF_REPLAY_BT_STEPS) * 256;
        uint16_t *ho = p.hidden + (size_t)b * A * 256;
        for (int q = tid; q < A * 256; q += blockDim.x) {
            const int a = q >> 8;
            ho[q] = (has_h && a < nag) ? hs[q & 255] : (uint16_t)0;
        }
        if (tid == 0) {
            p.action[b] = p.act[idx];
            p.reward[b] = half_bits_to_float(p.rew[idx]);
            p.done[b] = (l == size - 1 && p.done_buf[g]) ? 1.0f : 0.0f;  // :146-149
            p.steps[b] = (float)steps;
            p.bt_steps[b] = min(l + 1, (long long)MAPF_REPLAY_BT_STEPS);  // :151
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// GlobalBuffer.add for every finished environment of a vectorised actor, on the device (worker.py:71-104 per episode, in
// ascending environment order; LocalBuffer.finish priorities buffer.py:170-177 when `td` is not supplied).
//   flush_scan_kernel   one block: ranks the finished environments, assigns ring slots, advances ptr / size / counter and the
//                       slots' (size, done, num_agents);
//   flush_copy_kernel   one block per environment: priorities td^0.6 into the slot's 256 leaves, the episode's rows into the slot,
//                       and the slot's own sub-tree (8 levels) re-summed;
//   flush_top_kernel    one block: the tree above the slot roots re-summed (every node = left + right of its final children, so
//                       the result is bit-identical to the reference's per-episode batch_update, buffer.py:95-105).
// state: int64 {ptr, size, counter, episodes added by the last flush}.
// ---------------------------------------------------------------------------------------------------------------------
struct FlushParams {
    int E, S, A, RD, CW, capacity, num_agents;  // S = transitions per local buffer (rows = S + 1)
    const uint8_t *finished;   // [E] non-zero = flush this environment (nullptr: all)
    const int64_t *sizes;      // [E] episode lengths
    const uint8_t *done;       // [E]
    const uint32_t *obs_bits;  // [E][S+1][RD]
    const uint32_t *comm_bits; // [E][S+1][A][CW]
    const uint8_t *act;        // [E][S]
    const uint16_t *rew;       // [E][S] f16
    const uint16_t *hid;       // [E][S][256] f16
    const float *q;            // [E][S][5] or nullptr when td is given
    const double *td;          // [E][256] or nullptr
    const int32_t *nag_env;    // [E] or nullptr: agents of environment e (several levels in one flush) instead of num_agents
    int32_t *slot_of;          // [E] scratch: ring slot or -1
    int64_t *state;
    uint32_t *dst_obs, *dst_comm;
    uint8_t *dst_act, *dst_done;
    uint16_t *dst_rew, *dst_hid;
    int32_t *dst_size, *dst_nag;
    double *tree;
    long long leaves;
    int cap_log2;
};

__global__ void __launch_bounds__(1024) flush_scan_kernel(FlushParams p) {
    __shared__ int s_cnt[1024];
    __shared__ long long s_sum[1024];
    const int tid = threadIdx.x, nth = blockDim.x;
    const int per = (p.E + nth - 1) / nth, e0 = tid * per, e1 = min(p.E, e0 + per);
    int cnt = 0;
    for (int e = e0; e < e1; ++e) cnt += (p.finished == nullptr || p.finished[e] != 0) ? 1 : 0;
    s_cnt[tid] = cnt;
    __syncthreads();
    for (int d = 1; d < nth; d <<= 1) {  // inclusive scan
        const int v = tid >= d ? s_cnt[tid - d] : 0;
        __syncthreads();
        s_cnt[tid] += v;
        __syncthreads();
    }
    const int n = s_cnt[nth - 1];
    int rank = s_cnt[tid] - cnt;
    const long long ptr = p.state[0];
    long long dsize = 0, dcount = 0;
    for (int e = e0; e < e1; ++e) {
        int slot = -1;
        if (p.finished == nullptr || p.finished[e] != 0) {
            const int size = (int)p.sizes[e];
            dcount += size;
            if (rank >= n - p.capacity) {  // with more episodes than slots only the last `capacity` survive (later adds overwrite)
                slot = (int)((ptr + rank) % p.capacity);
                dsize += size - p.dst_size[slot];      // worker.py:90-91
                p.dst_size[slot] = size;
                p.dst_done[slot] = p.done[e] != 0;
                p.dst_nag[slot] = p.nag_env ? p.nag_env[e] : p.num_agents;
            }
            ++rank;
        }
        p.slot_of[e] = slot;
    }
    s_sum[tid] = dsize;
    __syncthreads();
    for (int d = nth >> 1; d > 0; d >>= 1) {
        if (tid < d) s_sum[tid] += s_sum[tid + d];
        __syncthreads();
    }
    const long long tsize = s_sum[0];
    __syncthreads();
    s_sum[tid] = dcount;
    __syncthreads();
    for (int d = nth >> 1; d > 0; d >>= 1) {
        if (tid < d) s_sum[tid] += s_sum[tid + d];
        __syncthreads();
    }
    if (tid == 0) {
        p.state[0] = (ptr + n) % p.capacity;  // worker.py:104
        p.state[1] += tsize;
        p.state[2] += s_sum[0];               // worker.py:92
        p.state[3] = n;
    }
}

__global__ void __launch_bounds__(256) flush_copy_kernel(FlushParams p) {
    const int e = blockIdx.x, tid = threadIdx.x;
    const int slot = p.slot_of[e];
    if (slot < 0) return;
    const int size = (int)p.sizes[e], S = p.S;
    // ---- priorities (buffer.py:170-177 in f64: |r_t + 0.99 r_{t+1} + max_a Q(s_t) - Q(s_t, a_t)|, zeros past the episode end) ----
    double td = 0.0;
    if (p.td != nullptr) {
        td = p.td[(size_t)e * kMaxSteps + tid];
    } else if (tid < size && tid < S) {
        const float *q = p.q + ((size_t)e * S + tid) * 5;
        float qmax = q[0];
        for (int a = 1; a < 5; ++a) qmax = fmaxf(qmax, q[a]);
        const double r0 = (double)half_bits_to_float(p.rew[(size_t)e * S + tid]);
        const double r1 = (tid + 1 < size) ? (double)half_bits_to_float(p.rew[(size_t)e * S + tid + 1]) : 0.0;
        const double ret = (r0 + 0.99 * r1) + (double)qmax;   // np.convolve(ret, [0.99, 1], 'valid') + q_max
        td = fabs(ret - (double)q[p.act[(size_t)e * S + tid]]);
    }
    const long long leaf0 = p.leaves - 1 + (long long)slot * kMaxSteps;
    p.tree[leaf0 + tid] = pow(td, kAlpha);  // worker.py:94
    // ---- the episode's rows (worker.py:96-102) ----
    const size_t row0 = (size_t)slot * kRows, tr0 = (size_t)slot * kMaxSteps;
    {
        const uint32_t *src = p.obs_bits + (size_t)e * (S + 1) * p.RD;
        uint32_t *dst = p.dst_obs + row0 * p.RD;
        const int n16 = ((size + 1) * p.RD) >> 2;  // RD is a multiple of 4
        for (int i = tid; i < n16; i += 256) reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
    }
    {
        const int rw = p.A * p.CW;
        const uint32_t *src = p.comm_bits + (size_t)e * (S + 1) * rw;
        uint32_t *dst = p.dst_comm + row0 * rw;
        for (int i = tid; i < (size + 1) * rw; i += 256) dst[i] = src[i];
    }
    for (int i = tid; i < size; i += 256) {
        p.dst_act[tr0 + i] = p.act[(size_t)e * S + i];
        p.dst_rew[tr0 + i] = p.rew[(size_t)e * S + i];
    }
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.hid + (size_t)e * S * 256);
        uint4 *dst = reinterpret_cast<uint4 *>(p.dst_hid + tr0 * 256);
        for (int i = tid; i < size * 32; i += 256) dst[i] = src[i];
    }
    // ---- the slot's sub-tree: 256 aligned leaves -> 8 levels up to the slot root ----
    __syncthreads();
    for (int k = 1; k <= 8; ++k) {
        if (tid < (kMaxSteps >> k)) {
            const long long node = ((leaf0 + 1) >> k) - 1 + tid;
            p.tree[node] = p.tree[2 * node + 1] + p.tree[2 * node + 2];
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(1024) flush_top_kernel(double *tree, int cap_log2, const int64_t *state) {
    if (state[3] == 0) return;  // nothing was added
    for (int lvl = cap_log2 - 1; lvl >= 0; --lvl) {
        const long long first = (1ll << lvl) - 1, count = 1ll << lvl;
        for (long long i = threadIdx.x; i < count; i += blockDim.x) {
            const long long node = first + i;
            tree[node] = tree[2 * node + 1] + tree[2 * node + 2];
        }
        __syncthreads();
    }
}

__global__ void state_reset_counter_kernel(int64_t *state) { state[2] = 0; }

// importance-sampling weights of a sampled batch (reference worker.py:165-166): w = (p / min p) ^ -beta, f64 like numpy, one workgroup
__global__ void __launch_bounds__(256) is_weights_kernel(const double *__restrict__ pri, int n, double beta, float *__restrict__ w) {
    __shared__ double red[256];
    double m = 1.0e300;
    for (int i = threadIdx.x; i < n; i += 256) m = fmin(m, pri[i]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmin(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    m = red[0];
    for (int i = threadIdx.x; i < n; i += 256) w[i] = (float)pow(pri[i] / m, -beta);
}

}  // namespace

struct mapf_replay {
    int capacity, A, RD, CW, device, layers, cap_log2;
    long long leaves;
    hipStream_t last_stream;  // stream of the most recent call: the state getters order their read behind it
    uint32_t *obs_bits;
    uint32_t *comm_bits;
    uint8_t *act;
    uint16_t *rew;
    uint16_t *hid;
    uint8_t *done_buf;
    int32_t *size_buf;
    int32_t *nag_buf;
    double *tree;
    int64_t *state;     // device: {ptr, size, counter, episodes added by the last flush} (GlobalBuffer.ptr / .size / .counter)
    int32_t *slot_of;   // scratch [slot_cap]
    int slot_cap;
    uint8_t *one;       // device constants for the single-episode add: finished = 1
    int64_t *tmp_size;  // [1]
    uint8_t *tmp_done;  // [1]
};

extern "C" {

int mapf_replay_create(int capacity_episodes, int max_agents, int device, mapf_replay_t **out) {
    if (!out) return MAPF_ERR_INVALID_ARG;
    *out = nullptr;
    if (capacity_episodes < 1 || max_agents < 1 || max_agents > 255 || device < 0) return MAPF_ERR_INVALID_ARG;
    long long leaves = (long long)capacity_episodes * kMaxSteps;
    int layers = 1;
    while ((1ll << (layers - 1)) < leaves) ++layers;
    if ((1ll << (layers - 1)) != leaves) return MAPF_ERR_INVALID_ARG;  // buffer.py:23
    if (device >= mapf_device_count()) return MAPF_ERR_HIP;
    mapf_replay *r = new (std::nothrow) mapf_replay();
    if (!r) return MAPF_ERR_HIP;
    r->capacity = capacity_episodes;
    r->A = max_agents;
    r->RD = ((max_agents * kObsBitsPerAgent + 31) / 32 + 3) & ~3;
    r->CW = (max_agents + 31) / 32;
    r->device = device;
    r->layers = layers;
    r->cap_log2 = layers - 1 - 8;  // capacity_episodes = 2^cap_log2
    r->leaves = leaves;
    r->last_stream = nullptr;
    r->slot_cap = 0;
    DeviceGuard guard(device);
    hipError_t err = guard.ok ? hipSuccess : hipErrorInvalidDevice;
    auto alloc = [&err](void **p, size_t bytes) {
        if (err == hipSuccess) err = hipMalloc(p, bytes);
        if (err == hipSuccess) err = hipMemset(*p, 0, bytes);
    };
    const size_t cap = capacity_episodes;
    alloc(reinterpret_cast<void **>(&r->obs_bits), cap * kRows * r->RD * 4);
    alloc(reinterpret_cast<void **>(&r->comm_bits), cap * kRows * r->A * r->CW * 4);
    alloc(reinterpret_cast<void **>(&r->act), cap * kMaxSteps);
    alloc(reinterpret_cast<void **>(&r->rew), cap * kMaxSteps * 2);
    alloc(reinterpret_cast<void **>(&r->hid), cap * kMaxSteps * 256 * 2);
    alloc(reinterpret_cast<void **>(&r->done_buf), cap);
    alloc(reinterpret_cast<void **>(&r->size_buf), cap * 4);
    alloc(reinterpret_cast<void **>(&r->nag_buf), cap * 4);
    alloc(reinterpret_cast<void **>(&r->tree), (size_t)(2 * leaves - 1) * 8);
    alloc(reinterpret_cast<void **>(&r->state), 5 * 8);  // {ptr, size, counter, last flush count, sampled-while-empty flag}
    alloc(reinterpret_cast<void **>(&r->slot_of), 4);
    alloc(reinterpret_cast<void **>(&r->one), 16);
    alloc(reinterpret_cast<void **>(&r->tmp_size), 8);
    alloc(reinterpret_cast<void **>(&r->tmp_done), 16);
    if (err == hipSuccess) err = hipMemset(r->one, 1, 1);
    if (err != hipSuccess) {
        std::fprintf(stderr, "mapf_replay_create: %s\n", hipGetErrorString(err));
        mapf_replay_destroy(r);
        return MAPF_ERR_HIP;
    }
    r->slot_cap = 1;
    *out = r;
    return MAPF_OK;
}

int mapf_replay_destroy(mapf_replay_t *r) {
    if (!r) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(r->device);
    (void)hipFree(r->obs_bits);
    (void)hipFree(r->comm_bits);
    (void)hipFree(r->act);
    (void)hipFree(r->rew);
    (void)hipFree(r->hid);
    (void)hipFree(r->done_buf);
    (void)hipFree(r->size_buf);
    (void)hipFree(r->nag_buf);
    (void)hipFree(r->tree);
    (void)hipFree(r->state);
    (void)hipFree(r->slot_of);
    (void)hipFree(r->one);
    (void)hipFree(r->tmp_size);
    (void)hipFree(r->tmp_done);
    delete r;
    return MAPF_OK;
}

int mapf_replay_row_dwords(const mapf_replay_t *r) { return r ? r->RD : MAPF_ERR_INVALID_ARG; }
int mapf_replay_capacity(const mapf_replay_t *r) { return r ? r->capacity : MAPF_ERR_INVALID_ARG; }

// the stream reads of the ring state are ordered behind (mapf_replay_size / _counter); a launch that is being CAPTURED into a graph
// executes later, on whatever stream replays the graph: a capturing stream is not recorded
static void note_stream(mapf_replay_t *r, hipStream_t s) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs == hipStreamCaptureStatusNone) r->last_stream = s;
}

// {ptr, size, counter, last flush count, sampled-while-empty flag}: a 40-byte read ordered behind the handle's most recent stream
// (blocks the host)
static int read_state(mapf_replay_t *r, int64_t st[5], void *stream) {
    DeviceGuard guard(r->device);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : r->last_stream;
    // behind everything the stream holds, INCLUDING replayed graphs: on this runtime a pageable hipMemcpyAsync + hipStreamSynchronize did
    // not wait for a preceding hipGraphLaunch on the stream (measured: the ring state of 18 replayed actor iterations read as empty),
    // an event does
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e1 = hipEventRecord(ev, s);
    if (e1 == hipSuccess) e1 = hipEventSynchronize(ev);
    (void)hipEventDestroy(ev);
    HIP_TRY(e1);
    HIP_TRY(hipMemcpy(st, r->state, 40, hipMemcpyDeviceToHost));
    return MAPF_OK;
}
int mapf_replay_state(mapf_replay_t *r, int64_t out[4], void *stream) {
    if (!r || !out) return MAPF_ERR_INVALID_ARG;
    int64_t st[5];
    const int rc = read_state(r, st, stream);
    if (rc != MAPF_OK) return rc;
    for (int i = 0; i < 4; ++i) out[i] = st[i];
    if (st[4] != 0) {  // a sample was drawn while the sum tree was empty (flagged by tree_sample_kernel); reported once
        DeviceGuard guard(r->device);
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : r->last_stream;
        HIP_TRY(hipMemsetAsync(r->state + 4, 0, 8, s));
        HIP_TRY(hipStreamSynchronize(s));
        return MAPF_ERR_NOT_READY;
    }
    return MAPF_OK;
}
int mapf_replay_ptr(const mapf_replay_t *r) {
    int64_t st[5];
    return r && read_state(const_cast<mapf_replay_t *>(r), st, nullptr) == MAPF_OK ? (int)st[0] : MAPF_ERR_INVALID_ARG;
}
int64_t mapf_replay_size(const mapf_replay_t *r) {
    int64_t st[5];
    return r && read_state(const_cast<mapf_replay_t *>(r), st, nullptr) == MAPF_OK ? st[1] : MAPF_ERR_INVALID_ARG;
}
int64_t mapf_replay_counter(const mapf_replay_t *r, int reset) {
    int64_t st[5];
    mapf_replay_t *rr = const_cast<mapf_replay_t *>(r);
    if (!rr || read_state(rr, st, nullptr) != MAPF_OK) return MAPF_ERR_INVALID_ARG;
    if (reset) {
        DeviceGuard guard(rr->device);
        hipLaunchKernelGGL(state_reset_counter_kernel, dim3(1), dim3(1), 0, rr->last_stream, rr->state);
    }
    return st[2];
}

static int tree_update(mapf_replay_t *r, const int64_t *idx, const double *pri, int n, double alpha, const int64_t *old_ptr_dev,
                       hipStream_t s) {
    if (n <= 0) return MAPF_OK;
    int threads = n < 64 ? 64 : (n > 1024 ? 1024 : ((n + 63) / 64) * 64);
    hipLaunchKernelGGL(tree_update_kernel, dim3(1), dim3(threads), 0, s, r->tree, r->leaves, r->layers, idx, pri, n, alpha, r->state,
                       old_ptr_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_replay_tree_update(mapf_replay_t *r, const int64_t *idx_dev, const double *pri_dev, int n, double alpha,
                            void *stream) {
    if (!r || !idx_dev || !pri_dev || n < 0) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(r->device);
    note_stream(r, static_cast<hipStream_t>(stream));
    return tree_update(r, idx_dev, pri_dev, n, alpha, nullptr, static_cast<hipStream_t>(stream));
}

int mapf_replay_tree_sample(mapf_replay_t *r, const double *uniforms_dev, int n, int unit_uniforms, int64_t *idx_dev, double *pri_dev,
                            int64_t *old_ptr_dev, void *stream) {
    if (!r || !uniforms_dev || !idx_dev || !pri_dev || n < 1) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(r->device);
    note_stream(r, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(tree_sample_kernel, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), r->tree,
                       r->leaves, r->layers, uniforms_dev, n, unit_uniforms, idx_dev, pri_dev, r->state, old_ptr_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_replay_tree_read(mapf_replay_t *r, double *tree_dev, void *stream) {
    if (!r || !tree_dev) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(r->device);
    note_stream(r, static_cast<hipStream_t>(stream));
    HIP_TRY(hipMemcpyAsync(tree_dev, r->tree, (size_t)(2 * r->leaves - 1) * 8, hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(stream)));
    return MAPF_OK;
}

static int flush(mapf_replay_t *r, FlushParams &p, hipStream_t s) {
    if (p.E > r->slot_cap) {  // scratch grows with the widest actor seen (first call with a new width only)
        HIP_TRY(hipStreamSynchronize(s));
        (void)hipFree(r->slot_of);
        r->slot_of = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&r->slot_of), (size_t)p.E * 4));
        r->slot_cap = p.E;
    }
    p.A = r->A;
    p.RD = r->RD;
    p.CW = r->CW;
    p.capacity = r->capacity;
    p.slot_of = r->slot_of;
    p.state = r->state;
    p.dst_obs = r->obs_bits;
    p.dst_comm = r->comm_bits;
    p.dst_act = r->act;
    p.dst_done = r->done_buf;
    p.dst_rew = r->rew;
    p.dst_hid = r->hid;
    p.dst_size = r->size_buf;
    p.dst_nag = r->nag_buf;
    p.tree = r->tree;
    p.leaves = r->leaves;
    p.cap_log2 = r->cap_log2;
    note_stream(r, s);
    hipLaunchKernelGGL(flush_scan_kernel, dim3(1), dim3(1024), 0, s, p);
    hipLaunchKernelGGL(flush_copy_kernel, dim3(p.E), dim3(256), 0, s, p);
    hipLaunchKernelGGL(flush_top_kernel, dim3(1), dim3(1024), 0, s, r->tree, r->cap_log2, r->state);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_replay_add(mapf_replay_t *r, int num_agents, int size, int done, const uint32_t *obs_bits_dev,
                    const uint32_t *comm_bits_dev, const uint8_t *act_dev, const uint16_t *rew_dev,
                    const uint16_t *hid_dev, const double *td_dev, void *stream) {
    if (!r || !obs_bits_dev || !comm_bits_dev || !act_dev || !rew_dev || !hid_dev || !td_dev) return MAPF_ERR_INVALID_ARG;
    if (num_agents < 1 || num_agents > r->A || size < 1 || size > kMaxSteps) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(r->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t sz = size;
    const uint8_t dn = done ? 1 : 0;
    HIP_TRY(hipMemcpyAsync(r->tmp_size, &sz, 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(r->tmp_done, &dn, 1, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // (pageable host sources)
    FlushParams p{};
    p.E = 1;
    p.S = size;  // the caller's arrays hold exactly size (+ 1) rows
    p.num_agents = num_agents;
    p.finished = r->one;
    p.sizes = r->tmp_size;
    p.done = r->tmp_done;
    p.obs_bits = obs_bits_dev;
    p.comm_bits = comm_bits_dev;
    p.act = act_dev;
    p.rew = rew_dev;
    p.hid = hid_dev;
    p.q = nullptr;
    p.td = td_dev;
    return flush(r, p, s);
}

int mapf_replay_add_many(mapf_replay_t *r, int num_envs, int num_agents, int local_steps, const uint8_t *finished_dev,
                         const int64_t *sizes_dev, const uint8_t *done_dev, const uint32_t *obs_bits_dev,
                         const uint32_t *comm_bits_dev, const uint8_t *act_dev, const uint16_t *rew_dev, const uint16_t *hid_dev,
                         const float *q_dev, void *stream) {
    if (!r || !sizes_dev || !done_dev || !obs_bits_dev || !comm_bits_dev || !act_dev || !rew_dev || !hid_dev || !q_dev)
        return MAPF_ERR_INVALID_ARG;
    if (num_envs < 1 || num_agents < 1 || num_agents > r->A || local_steps < 1 || local_steps > kMaxSteps) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(r->device);
    FlushParams p{};
    p.E = num_envs;
    p.S = local_steps;
    p.num_agents = num_agents;
    p.finished = finished_dev;
    p.sizes = sizes_dev;
    p.done = done_dev;
    p.obs_bits = obs_bits_dev;
    p.comm_bits = comm_bits_dev;
    p.act = act_dev;
    p.rew = rew_dev;
    p.hid = hid_dev;
    p.q = q_dev;
    p.td = nullptr;
    return flush(r, p, static_cast<hipStream_t>(stream));
}

int mapf_replay_add_many_env(mapf_replay_t *r, int num_envs, const int32_t *num_agents_dev, int local_steps, const uint8_t *finished_dev,
                             const int64_t *sizes_dev, const uint8_t *done_dev, const uint32_t *obs_bits_dev, const uint32_t *comm_bits_dev,
                             const uint8_t *act_dev, const uint16_t *rew_dev, const uint16_t *hid_dev, const float *q_dev, void *stream) {
    if (!r || !num_agents_dev || !sizes_dev || !done_dev || !obs_bits_dev || !comm_bits_dev || !act_dev || !rew_dev || !hid_dev || !q_dev)
        return MAPF_ERR_INVALID_ARG;
    if (num_envs < 1 || local_steps < 1 || local_steps > kMaxSteps) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(r->device);
    FlushParams p{};
    p.E = num_envs;
    p.S = local_steps;
    p.num_agents = 0;
    p.nag_env = num_agents_dev;
    p.finished = finished_dev;
    p.sizes = sizes_dev;
    p.done = done_dev;
    p.obs_bits = obs_bits_dev;
    p.comm_bits = comm_bits_dev;
    p.act = act_dev;
    p.rew = rew_dev;
    p.hid = hid_dev;
    p.q = q_dev;
    p.td = nullptr;
    return flush(r, p, static_cast<hipStream_t>(stream));
}

int mapf_replay_is_weights(const double *pri_dev, int n, double beta, float *weights_dev, void *stream) {
    if (!pri_dev || !weights_dev || n < 1) return MAPF_ERR_INVALID_ARG;
    hipLaunchKernelGGL(is_weights_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), pri_dev, n, beta, weights_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_replay_sample(mapf_replay_t *r, const double *uniforms_dev, int n, int unit_uniforms, int64_t *idx_dev, double *pri_dev,
                       uint16_t *obs_dev, uint8_t *comm_dev, uint16_t *hidden_dev, int64_t *action_dev,
                       float *reward_dev, float *done_dev, float *steps_dev, int64_t *bt_steps_dev, int64_t *old_ptr_dev,
                       void *stream) {
    if (!r || !uniforms_dev || !idx_dev || !pri_dev || !obs_dev || !comm_dev || !hidden_dev || !action_dev ||
        !reward_dev || !done_dev || !steps_dev || !bt_steps_dev || n < 1)
        return MAPF_ERR_INVALID_ARG;
    int st = mapf_replay_tree_sample(r, uniforms_dev, n, unit_uniforms, idx_dev, pri_dev, old_ptr_dev, stream);
    if (st != MAPF_OK) return st;
    DeviceGuard guard(r->device);
    GatherParams g{};
    g.A = r->A;
    g.RD = r->RD;
    g.CW = r->CW;
    g.n = n;
    g.obs_bits = r->obs_bits;
    g.comm_bits = r->comm_bits;
    g.act = r->act;
    g.rew = r->rew;
    g.hid = r->hid;
    g.done_buf = r->done_buf;
    g.size_buf = r->size_buf;
    g.nag_buf = r->nag_buf;
    g.idx = idx_dev;
    g.obs = obs_dev;
    g.comm = comm_dev;
    g.hidden = hidden_dev;
    g.action = action_dev;
    g.reward = reward_dev;
    g.done = done_dev;
    g.steps = steps_dev;
    g.bt_steps = bt_steps_dev;
    hipLaunchKernelGGL(gather_kernel, dim3(kWindow, n), dim3(256), 0, static_cast<hipStream_t>(stream), g);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_replay_update_priorities(mapf_replay_t *r, const int64_t *idx_dev, const double *pri_dev, int n, const int64_t *old_ptr_dev,
                                  void *stream) {
    if (!r || !idx_dev || !pri_dev || n < 0 || !old_ptr_dev) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(r->device);
    note_stream(r, static_cast<hipStream_t>(stream));
    return tree_update(r, idx_dev, pri_dev, n, kAlpha, old_ptr_dev, static_cast<hipStream_t>(stream));
}

}  // extern "C"
