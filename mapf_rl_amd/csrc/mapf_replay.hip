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
                                                           int stale_mode, long long lo, long long hi) {
    const int tid = threadIdx.x;
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
                                   int64_t *idx_out, double *pri_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double sum = tree[0];
    const double interval = sum / (double)n;              // buffer.py:58
    double prefix = (double)k * interval + uniforms[k];   // buffer.py:60 (np.arange(0, sum, interval)[k] = k*interval)
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

__global__ void set_slot_meta_kernel(uint8_t *done_buf, int32_t *size_buf, int32_t *nag_buf, int slot, int done, int size,
                                     int nag) {
    done_buf[slot] = (uint8_t)done;
    size_buf[slot] = size;
    nag_buf[slot] = nag;
}

__global__ void iota_kernel(int64_t *out, long long start, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = start + i;
}

}  // namespace

struct mapf_replay {
    int capacity, A, RD, CW, device, layers;
    long long leaves;
    int ptr;
    long long size, counter;
    int32_t *host_sizes;  // mirror of size_buf for the host-side `size` accounting (worker.py:90-91)
    uint32_t *obs_bits;
    uint32_t *comm_bits;
    uint8_t *act;
    uint16_t *rew;
    uint16_t *hid;
    uint8_t *done_buf;
    int32_t *size_buf;
    int32_t *nag_buf;
    double *tree;
    int64_t *slot_idx;  // scratch [256]
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
    r->leaves = leaves;
    r->ptr = 0;
    r->size = r->counter = 0;
    r->host_sizes = new (std::nothrow) int32_t[capacity_episodes]();
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
    alloc(reinterpret_cast<void **>(&r->slot_idx), kMaxSteps * 8);
    if (err != hipSuccess || !r->host_sizes) {
        std::fprintf(stderr, "mapf_replay_create: %s\n", hipGetErrorString(err));
        mapf_replay_destroy(r);
        return MAPF_ERR_HIP;
    }
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
    (void)hipFree(r->slot_idx);
    delete[] r->host_sizes;
    delete r;
    return MAPF_OK;
}

int mapf_replay_row_dwords(const mapf_replay_t *r) { return r ? r->RD : MAPF_ERR_INVALID_ARG; }
int mapf_replay_capacity(const mapf_replay_t *r) { return r ? r->capacity : MAPF_ERR_INVALID_ARG; }
int mapf_replay_ptr(const mapf_replay_t *r) { return r ? r->ptr : MAPF_ERR_INVALID_ARG; }
int64_t mapf_replay_size(const mapf_replay_t *r) { return r ? r->size : MAPF_ERR_INVALID_ARG; }
int64_t mapf_replay_counter(const mapf_replay_t *r, int reset) {
    if (!r) return MAPF_ERR_INVALID_ARG;
    int64_t c = r->counter;
    if (reset) const_cast<mapf_replay_t *>(r)->counter = 0;
    return c;
}

static int tree_update(mapf_replay_t *r, const int64_t *idx, const double *pri, int n, double alpha, int stale_mode,
                       long long lo, long long hi, hipStream_t s) {
    if (n <= 0) return MAPF_OK;
    int threads = n < 64 ? 64 : (n > 1024 ? 1024 : ((n + 63) / 64) * 64);
    hipLaunchKernelGGL(tree_update_kernel, dim3(1), dim3(threads), 0, s, r->tree, r->leaves, r->layers, idx, pri, n, alpha,
                       stale_mode, lo, hi);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_replay_tree_update(mapf_replay_t *r, const int64_t *idx_dev, const double *pri_dev, int n, double alpha,
                            void *stream) {
    if (!r || !idx_dev || !pri_dev || n < 0) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(r->device);
    return tree_update(r, idx_dev, pri_dev, n, alpha, 0, 0, 0, static_cast<hipStream_t>(stream));
}

int mapf_replay_tree_sample(mapf_replay_t *r, const double *uniforms_dev, int n, int64_t *idx_dev, double *pri_dev,
                            void *stream) {
    if (!r || !uniforms_dev || !idx_dev || !pri_dev || n < 1) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(r->device);
    hipLaunchKernelGGL(tree_sample_kernel, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), r->tree,
                       r->leaves, r->layers, uniforms_dev, n, idx_dev, pri_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_replay_tree_read(mapf_replay_t *r, double *tree_dev, void *stream) {
    if (!r || !tree_dev) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(r->device);
    HIP_TRY(hipMemcpyAsync(tree_dev, r->tree, (size_t)(2 * r->leaves - 1) * 8, hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(stream)));
    return MAPF_OK;
}

int mapf_replay_add(mapf_replay_t *r, int num_agents, int size, int done, const uint32_t *obs_bits_dev,
                    const uint32_t *comm_bits_dev, const uint8_t *act_dev, const uint16_t *rew_dev,
                    const uint16_t *hid_dev, const double *td_dev, void *stream) {
    if (!r || !obs_bits_dev || !comm_bits_dev || !act_dev || !rew_dev || !hid_dev || !td_dev) return MAPF_ERR_INVALID_ARG;
    if (num_agents < 1 || num_agents > r->A || size < 1 || size > kMaxSteps) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(r->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int p = r->ptr;
    const size_t row0 = (size_t)p * kRows, tr0 = (size_t)p * kMaxSteps;
    // worker.py:90-92
    r->size -= r->host_sizes[p];
    r->size += size;
    r->counter += size;
    r->host_sizes[p] = size;
    // worker.py:94: priorities td^alpha for all 256 leaves of the slot
    hipLaunchKernelGGL(iota_kernel, dim3(1), dim3(kMaxSteps), 0, s, r->slot_idx, (long long)tr0, kMaxSteps);
    HIP_TRY(hipGetLastError());
    int st = tree_update(r, r->slot_idx, td_dev, kMaxSteps, kAlpha, 0, 0, 0, s);
    if (st != MAPF_OK) return st;
    // worker.py:96-102
    HIP_TRY(hipMemcpyAsync(r->obs_bits + row0 * r->RD, obs_bits_dev, (size_t)(size + 1) * r->RD * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(r->comm_bits + row0 * r->A * r->CW, comm_bits_dev, (size_t)(size + 1) * r->A * r->CW * 4,
                           hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(r->act + tr0, act_dev, (size_t)size, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(r->rew + tr0, rew_dev, (size_t)size * 2, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(r->hid + tr0 * 256, hid_dev, (size_t)size * 256 * 2, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(set_slot_meta_kernel, dim3(1), dim3(1), 0, s, r->done_buf, r->size_buf, r->nag_buf, p, done, size,
                       num_agents);
    HIP_TRY(hipGetLastError());
    r->ptr = (p + 1) % r->capacity;  // worker.py:104
    return MAPF_OK;
}

int mapf_replay_sample(mapf_replay_t *r, const double *uniforms_dev, int n, int64_t *idx_dev, double *pri_dev,
                       uint16_t *obs_dev, uint8_t *comm_dev, uint16_t *hidden_dev, int64_t *action_dev,
                       float *reward_dev, float *done_dev, float *steps_dev, int64_t *bt_steps_dev, void *stream) {
    if (!r || !uniforms_dev || !idx_dev || !pri_dev || !obs_dev || !comm_dev || !hidden_dev || !action_dev ||
        !reward_dev || !done_dev || !steps_dev || !bt_steps_dev || n < 1)
        return MAPF_ERR_INVALID_ARG;
    if (r->size <= 0) return MAPF_ERR_NOT_READY;
    int st = mapf_replay_tree_sample(r, uniforms_dev, n, idx_dev, pri_dev, stream);
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

int mapf_replay_update_priorities(mapf_replay_t *r, const int64_t *idx_dev, const double *pri_dev, int n, int old_ptr,
                                  void *stream) {
    if (!r || !idx_dev || !pri_dev || n < 0 || old_ptr < 0 || old_ptr >= r->capacity) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(r->device);
    int mode = 0;
    long long lo = (long long)old_ptr * kMaxSteps, hi = (long long)r->ptr * kMaxSteps;
    if (r->ptr > old_ptr) mode = 1;       // worker.py:192-196
    else if (r->ptr < old_ptr) mode = 2;  // worker.py:197-201
    return tree_update(r, idx_dev, pri_dev, n, kAlpha, mode, lo, hi, static_cast<hipStream_t>(stream));
}

}  // extern "C"
