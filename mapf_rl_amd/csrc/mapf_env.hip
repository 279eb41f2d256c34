// mapf_env.hip -- MI355X (gfx950) kernels + C ABI of the vectorised MAPF environment.
//
// Hot path of the reference being replaced (ZiyuanMa/MAPF_RL): Environment.load / get_navi_map /
// step / observe, reference environment.py:198-467.  See include/mapf_env.h for the boundary and
// DESIGN.md for the data layout and the per-kernel roofline.
//
// Data layout in HBM (per handle; W = uint32_t when L <= 32, uint64_t when L <= 64):
//   map_rows W[E][L]          bit y of row x = obstacle at (x, y)
//   agents   int16[E][N][2]   current (row, col)
//   goals    int16[E][N][2]
//   navi     W[E][N][L][4]    per (agent, row) one 16/32-byte record {up, down, left, right}: bit y of
//                             word k = "neighbour k of (row, y) is strictly closer to the agent's goal"
//   steps    int32[E]
// All integer/bit work: the kernels are HBM-bound (obs write 486 B/agent dominates), not MFMA work.
//
// Why the parallel conflict resolution equals the reference's sequential one (environment.py:368-406).
// After S0-S2 every agent is a *mover* (wants cell next != cur) or *settled* (next == cur).  The reference
// repeatedly takes the first mover (id order) whose target cell is also claimed by somebody else and resolves
// that cell: if a settled agent is among the claimants every mover into the cell reverts, otherwise all but
// the lowest-id mover revert (stable sort on identical keys, :392); a reverted agent becomes settled on its
// own cell, which may evict movers into that cell in a later pass.  Two facts make the outcome independent
// of the processing order:
//   (1) a mover only ever reverts because of its OWN target cell, so all original movers into a cell are
//       still movers when that cell is first resolved; hence every mover that is not the lowest id into its
//       cell reverts no matter what (rule b), and the lowest-id mover M(c) of a cell c can only revert through
//       rule (a): the agent standing on c ends up settled there;
//   (2) rule (a) is monotone -- the set of settled agents only grows -- so "M(c) reverts iff the occupant of c
//       is settled in the end" has a unique least fixed point, reached by iterating rule (a) from the state
//       left by rule (b) until nothing changes (chains revert back to front, rotations with no outside
//       intruder never start reverting and go through).
// The kernel computes exactly that fixed point; tests compare it with the sequential oracle on >100k steps
// (tests/test_env_gpu.py) including the hand-built cascade cases K5, K8, K8b, K8c, K13.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "mapf_env.h"

namespace {

constexpr int kStatusAction = 1;   // action outside [0,5)
constexpr int kStatusOverlap = 2;  // two agents on one cell after a step
constexpr int kStatusRange = 4;    // position outside the map in a device-side load
constexpr int kStatusStage = 8;    // mapf_reset_envs in staged mode found no staged scenario for an environment's next epoch

// LDS atomic OR on a bit row (uint64_t is `unsigned long` here; HIP's overload wants unsigned long long)
__device__ __forceinline__ uint32_t lds_or(uint32_t *p, uint32_t v) { return atomicOr(p, v); }
__device__ __forceinline__ uint64_t lds_or(uint64_t *p, uint64_t v) {
    return (uint64_t)atomicOr(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v);
}

template <typename W>
struct alignas(16) NaviRec {
    W w[4];
};

struct StepParams {
    int E, L, N;
    const void *map_rows;
    int16_t *agents;
    const int16_t *goals;
    const void *navi;
    int32_t *steps;
    int32_t *status;
    const int8_t *actions;
    uint8_t *obs;
    uint32_t *obs_bits;  // optional bit-packed observation rows [E][RD]
    int obs_bits_rd;
    int16_t *pos_out;
    int8_t *rclass;
    float *reward;
    uint8_t *done;
    float rtab[5];
    const uint8_t *mask;  // observe only (optional): a workgroup none of whose environments is flagged returns at once
    int ablate;  // tuning only: bit0 skip navi loads, bit1 skip obs stores
    int plane;   // observation fields assembled per (agent, channel) plane from navi records staged in LDS (see the kernel)
    int nt_store;  // observation bytes written with non-temporal stores
    unsigned long long *dbg;  // diagnostic builds only: per-block phase stamps [E][8]
};

// columns y-R .. y+R of a bit row -> bits 0 .. 2R (out-of-range columns read 0)
template <typename W, int R>
__device__ __forceinline__ unsigned window_bits(W row, int y) {
    constexpr unsigned M = (1u << (2 * R + 1)) - 1u;
    return (y >= R) ? ((unsigned)(row >> (y - R)) & M) : (((unsigned)row << (R - y)) & M);
}

// ---------------------------------------------------------------------------------------------
// env_step_kernel: one workgroup per environment (Environment.step + observe, reference environment.py:278-467).
//   * every global load is issued up front: the navi records an agent can need after the step are rows
//     min(x, x+dx)-R .. max(x, x+dx)+R (<= 2R+2 rows, dx from its action), fetched speculatively into
//     registers while the step logic runs -- no dependent HBM round trip after the step;
//   * occupant lookup through a padded id grid in LDS (O(1)); the rivals for a target cell can only be the
//     occupants of its 4 neighbours, so rule (b) is 4 grid lookups instead of an O(N) scan;
//   * the observation is assembled as a bit string in LDS (bit b of the string = output byte b of this
//     env's obs block, because every channel row is a (2R+1)-bit field at bit offset (2R+1)*field), then
//     expanded 16 bits -> 16 bytes per lane per store.
// ---------------------------------------------------------------------------------------------
template <int WW>
__device__ __forceinline__ void deposit_field(unsigned *bits, unsigned off, unsigned v) {
    if (v) {
        unsigned d = off >> 5, sh = off & 31u;
        atomicOr(&bits[d], v << sh);
        if (sh > 32u - WW) atomicOr(&bits[d + 1], v >> (32u - sh));
    }
}

__device__ __forceinline__ unsigned expand4m(unsigned nib) { return __umul24(nib, 0x00204081u) & 0x01010101u; }

template <int NT>
__device__ __forceinline__ void block_sync() {
    __syncthreads();  // NT == 64: one wavefront per block, the compiler drops the s_barrier
}
template <int NT>
__device__ __forceinline__ int block_or(int pred) {
    if constexpr (NT == 64) return __any(pred);
    else return __syncthreads_or(pred);
}
template <int NT>
__device__ __forceinline__ int block_and(int pred) {
    if constexpr (NT == 64) return __all(pred);
    else return __syncthreads_and(pred);
}

// G > 1 (few agents per environment, L <= 32): one workgroup steps G CONSECUTIVE environments.  Every per-agent array of the
// handle is contiguous over environments ([E][N]...), so the block simply sees G * N "virtual agents" -- agent a belongs to
// environment a / N of the block -- and ONE observation bit string / byte block of G * N * 486 cells; only the maps, the
// occupancy rows, the id grids and the all-on-goal flag exist once per environment.  What it buys: the fixed instruction stream
// of a block (load rounds, LDS init, barriers, the unrolled field loop) is paid once per G environments, and the observation
// block is a multiple of 16 bytes whenever G * N is a multiple of 8 (with 6 agents a lone environment's 2,916 bytes are not even
// 16-byte aligned: 4-byte stores; with 1 agent byte stores).
template <typename W, int R, bool DO_STEP, bool DO_OBS, int VEC, int ITERS, int NT, int G>
__global__ void __launch_bounds__(NT) env_step_kernel(StepParams p) {
    const unsigned blk = blockIdx.x;  // the workgroup's index among the workgroups of the handle
#include "mapf_env_step_body.inc"
}

// the same body for a workgroup of a launch that covers several handles (env_step_multi_kernel): `p` = the handle's parameters in the
// device-side table, `blk` = the workgroup's index among the workgroups of THAT handle
template <typename W, int R, bool DO_STEP, bool DO_OBS, int VEC, int ITERS, int NT, int G>
__device__ __forceinline__ void env_step_body(const StepParams &p, const unsigned blk) {
#include "mapf_env_step_body.inc"
}


// ---------------------------------------------------------------------------------------------
// Several handles in ONE launch (mapf_multi_*).  The reference draws a (num_agents, map side) level per episode inside one actor
// (environment.py:148-151, worker.py:422-428); here every active level of the curriculum is a handle of a few hundred environments,
// and one launch per level and step is a latency-bound launch of a few hundred one-wavefront workgroups each (0.24-0.6 of the
// roofline, profiles/r03_shape_sweep.md).  The table below lives in DEVICE memory (built once per set of handles): a workgroup finds
// its segment from the prefix of workgroup counts, loads that handle's parameters through the scalar unit and runs the body that was
// instantiated for the handle's shape.  One-wavefront workgroups only (every shape the single-handle rules give 64 threads:
// N <= 24, or 64-bit rows with N <= 64 -- here N <= 25 so that four load rounds suffice).
// ---------------------------------------------------------------------------------------------
constexpr int kMultiMax = 16;
struct MultiTable {
    int n;
    int block_end[kMultiMax];  // workgroups of segments 0..i
    int variant[kMultiMax];
    StepParams seg[kMultiMax];
    int env_end[kMultiMax];    // environments of segments 0..i (the reset launch: one wavefront per environment)
    int32_t *epochs[kMultiMax];
    unsigned long long reset_seed[kMultiMax];  // scenario stream of segment i at iteration k: reset_seed[i] + k
};

// variant = instantiation of the body: (W, G, VEC, ITERS)
enum : int {
    MV_U32_G8_I2, MV_U32_G8_I4, MV_U32_G8_I7, MV_U32_G4_I4,
    MV_U32_G1_V16_I2, MV_U32_G1_V16_I4, MV_U32_G1_V4_I2, MV_U32_G1_V4_I4, MV_U32_G1_V1_I2, MV_U32_G1_V1_I4,
    MV_U64_G1_V16_I2, MV_U64_G1_V16_I4, MV_U64_G1_V4_I2, MV_U64_G1_V4_I4, MV_U64_G1_V1_I2, MV_U64_G1_V1_I4,
    MV_COUNT
};

template <bool DO_STEP>
__global__ void __launch_bounds__(64) env_step_multi_kernel(const MultiTable *__restrict__ tab) {
    int seg = 0;
    const int n = tab->n;
    while (seg < n - 1 && (int)blockIdx.x >= tab->block_end[seg]) ++seg;  // (wave-uniform: scalar loads)
    const int blk = (int)blockIdx.x - (seg ? tab->block_end[seg - 1] : 0);
    const StepParams &p = tab->seg[seg];
    switch (tab->variant[seg]) {
#define MV_CASE(id, W, G, VEC, ITERS) \
    case id: env_step_body<W, 4, DO_STEP, true, VEC, ITERS, 64, G>(p, (unsigned)blk); break;
        MV_CASE(MV_U32_G8_I2, uint32_t, 8, 16, 2)
        MV_CASE(MV_U32_G8_I4, uint32_t, 8, 16, 4)
        MV_CASE(MV_U32_G8_I7, uint32_t, 8, 16, 7)
        MV_CASE(MV_U32_G4_I4, uint32_t, 4, 16, 4)
        MV_CASE(MV_U32_G1_V16_I2, uint32_t, 1, 16, 2)
        MV_CASE(MV_U32_G1_V16_I4, uint32_t, 1, 16, 4)
        MV_CASE(MV_U32_G1_V4_I2, uint32_t, 1, 4, 2)
        MV_CASE(MV_U32_G1_V4_I4, uint32_t, 1, 4, 4)
        MV_CASE(MV_U32_G1_V1_I2, uint32_t, 1, 1, 2)
        MV_CASE(MV_U32_G1_V1_I4, uint32_t, 1, 1, 4)
        MV_CASE(MV_U64_G1_V16_I2, uint64_t, 1, 16, 2)
        MV_CASE(MV_U64_G1_V16_I4, uint64_t, 1, 16, 4)
        MV_CASE(MV_U64_G1_V4_I2, uint64_t, 1, 4, 2)
        MV_CASE(MV_U64_G1_V4_I4, uint64_t, 1, 4, 4)
        MV_CASE(MV_U64_G1_V1_I2, uint64_t, 1, 1, 2)
        MV_CASE(MV_U64_G1_V1_I4, uint64_t, 1, 1, 4)
#undef MV_CASE
        default: break;
    }
}

// ---------------------------------------------------------------------------------------------
// navi_bfs_kernel: Environment.get_navi_map (environment.py:217-276) as a bit-parallel BFS.
// lane = map row, the frontier / visited sets are one W-bit word per row; one BFS level is
// (up | down | left | right neighbours of the frontier) & free & ~visited, and a cell first reached at
// level d from a frontier cell in direction k has exactly the reference's "neighbour k is strictly
// closer" flag.  L <= 32: two fields per wavefront; L <= 64: one.
// ---------------------------------------------------------------------------------------------
template <typename W>
__device__ __forceinline__ W shfl_up_w(W v, int width) {
    if constexpr (sizeof(W) == 4) {
        return (W)__shfl_up((unsigned)v, 1, width);
    } else {
        unsigned lo = __shfl_up((unsigned)v, 1, width);
        unsigned hi = __shfl_up((unsigned)(v >> 32), 1, width);
        return ((W)hi << 32) | lo;
    }
}
template <typename W>
__device__ __forceinline__ W shfl_down_w(W v, int width) {
    if constexpr (sizeof(W) == 4) {
        return (W)__shfl_down((unsigned)v, 1, width);
    } else {
        unsigned lo = __shfl_down((unsigned)v, 1, width);
        unsigned hi = __shfl_down((unsigned)(v >> 32), 1, width);
        return ((W)hi << 32) | lo;
    }
}

template <typename W>
__global__ void __launch_bounds__(256) navi_bfs_kernel(int E, int L, int N, const W *map_rows,
                                                      const int16_t *goals, NaviRec<W> *navi, int32_t *status,
                                                      const int32_t *env_ids, const uint8_t *env_mask = nullptr) {
    constexpr int LPF = sizeof(W) == 4 ? 32 : 64;  // lanes per field
    constexpr int FPW = 64 / LPF;                  // fields per wavefront
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long slot = wave * FPW + lane / LPF;  // (listed env, agent) pair
    const int row = lane % LPF;
    const bool valid = slot < (long long)E * N;
    int e = valid ? (int)(slot / N) : 0;
    if (env_ids) e = env_ids[e];                      // partial rebuild: E listed environments
    const long long field = (long long)e * N + (valid ? slot % N : 0);
    // partial rebuild by flags (mapf_reset_envs: the environments whose scenario was just re-drawn): the other fields' lanes idle along
    const bool in_map = valid && row < L && (env_mask == nullptr || env_mask[e] != 0);
    if (env_mask != nullptr && __ballot(in_map) == 0ull) return;
    const W lmask = (L == (int)(8 * sizeof(W))) ? ~(W)0 : (((W)1 << L) - 1);
    W freec = 0;
    if (in_map) freec = ~map_rows[(size_t)e * L + row] & lmask;
    int gx = -1, gy = 0;
    if (valid && (env_mask == nullptr || env_mask[e] != 0)) {
        gx = goals[field * 2];
        gy = goals[field * 2 + 1];
        if (gx < 0 || gx >= L || gy < 0 || gy >= L) {
            if (row == 0) atomicOr(status, kStatusRange);
            gx = -1;
            gy = 0;
        }
    }
    W vis = (in_map && row == gx) ? ((W)1 << gy) : (W)0;  // dist[goal] = 0 unconditionally (:223)
    W fr = vis;
    W up = 0, down = 0, left = 0, right = 0;
    const int max_iter = L * L;
    for (int it = 0; it < max_iter; ++it) {
        W fu = shfl_up_w<W>(fr, LPF);
        if (row == 0) fu = 0;
        W fd = shfl_down_w<W>(fr, LPF);
        if (row == LPF - 1) fd = 0;
        W fl = fr << 1, frr = fr >> 1;
        W nw = (fu | fd | fl | frr) & freec & ~vis;
        up |= nw & fu;      // neighbour (x-1, y) strictly closer  (:260-262)
        down |= nw & fd;    // (x+1, y)                           (:264-266)
        left |= nw & fl;    // (x, y-1)                           (:268-270)
        right |= nw & frr;  // (x, y+1)                           (:272-274)
        vis |= nw;
        fr = nw;
        if (__ballot(nw != 0) == 0ull) break;
    }
    if (in_map) {
        NaviRec<W> rec;
        rec.w[0] = up;
        rec.w[1] = down;
        rec.w[2] = left;
        rec.w[3] = right;
        navi[field * L + row] = rec;
    }
}

// ---------------------------------------------------------------------------------------------
// reset_kernel: on-device scenario generation + navi build for the environments flagged in `mask`
// (Environment.reset, reference environment.py:146-196 / __init__ :100-143, statistical parity only: own
// counter-based RNG).  One wavefront per environment, lane = map row.
//   map      Bernoulli(rho) per cell, rho = density or ~ triangular(0, 0.33, 0.5)               (:156-157)
//   placement  the reference picks a partition with probability proportional to its remaining cells, then two
//            distinct cells of it uniformly (:118-138)  ==  the first cell uniform over all remaining cells of
//            partitions that still hold >= 2 cells, the second uniform over the rest of the same partition.
//            Here the first cell drawn is the GOAL: the navi BFS from it (needed anyway) yields the partition
//            as its visited set, so one BFS per agent does both jobs; cells of partitions found too small are
//            dropped from the candidate set (rejection keeps the draw uniform).
//   a map that cannot host N agents is re-drawn (the reference raises ValueError there, :120).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float u01(uint64_t h) { return (float)(h >> 40) * (1.0f / 16777216.0f); }

template <typename W>
__device__ __forceinline__ int popc_w(W v) {
    if constexpr (sizeof(W) == 4) return __popc((unsigned)v);
    else return __popcll((unsigned long long)v);
}
template <typename W>
__device__ __forceinline__ int ctz_w(W v) {
    if constexpr (sizeof(W) == 4) return __ffs((unsigned)v) - 1;
    else return __ffsll((unsigned long long)v) - 1;
}

// wave-wide: picks the r-th (0-based) set bit of the per-lane words `bits` in (row, column) order; every lane returns (x, y)
template <typename W>
__device__ __forceinline__ void select_bit(W bits, int r, int lane, int &x, int &y) {
    int cnt = popc_w<W>(bits), incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    const int excl = incl - cnt;
    const bool mine = r >= excl && r < incl;
    int yy = 0;
    if (mine) {
        W w = bits;
        for (int k = r - excl; k > 0; --k) w &= w - 1;
        yy = ctz_w<W>(w);
    }
    const unsigned long long who = __ballot(mine);
    const int src = who ? (int)__ffsll(who) - 1 : 0;
    x = src;
    y = __shfl(yy, src, 64);
}

template <typename W>
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// NAVI: the BFS from every goal also leaves the goal's navigation field (one wavefront does placement AND all N fields: 0.25-0.5 ms per
// call at 40 agents, on the actor's critical path whenever an episode ended).  !NAVI (mapf_reset_envs, round 4): placement only --
// the partition of a drawn goal cell is looked up among the partitions already seen on this map (the first few draws find the large
// ones; a BFS only for a cell of a new partition), and the N fields are built afterwards by navi_bfs_kernel on the flagged
// environments, one wavefront per field, in parallel.  Same draws, same scenario, same fields either way.
constexpr int RESET_CACHED_PARTS = 16;
// STAGE (mapf_stage_next, round 5): the scenario of (environment, epoch + 1) is drawn AHEAD into staging arrays -- map_rows / agents /
// goals are then the staging ones, `st_epoch[e]` names the epoch the staged scenario belongs to (nothing is drawn while it is still the
// next one), `st_need[e]` tells the field kernel behind this one whether the environment was re-drawn; the live epoch counter and the
// step counter are not touched.  Same draws as the direct reset of that (seed, environment, epoch).
template <typename W, bool NAVI, bool STAGE = false>
__device__ __forceinline__ void reset_body(const int e, int E, int L, int N, const uint8_t *mask, float density, uint64_t seed,
                                           W *map_rows, int16_t *agents, int16_t *goals, NaviRec<W> *navi,
                                           int32_t *steps, int32_t *epochs, int32_t *status, int32_t *st_epoch = nullptr,
                                           uint8_t *st_need = nullptr) {
    const int lane = threadIdx.x;
    if (e >= E || (mask && !mask[e])) return;
    __shared__ W s_part[NAVI ? 1 : RESET_CACHED_PARTS][64];  // !NAVI: row masks of the partitions seen on the current map
    const int row = lane;
    const bool in_map = row < L;
    const W lmask = (L == (int)(8 * sizeof(W))) ? ~(W)0 : (((W)1 << L) - 1);
    int epoch = 0;
    if (STAGE) {
        epoch = epochs[e] + 1;
        const bool have = st_epoch[e] == epoch;
        if (lane == 0) st_need[e] = have ? 0 : 1;
        if (have) return;
    } else {
        if (lane == 0) epoch = atomicAdd(&epochs[e], 1) + 1;
        epoch = __shfl(epoch, 0, 64);
    }
    const uint64_t base = mix64(seed ^ mix64(((uint64_t)e << 32) | (uint32_t)epoch));
    bool ok = false;
    for (int attempt = 0; attempt < 64 && !ok; ++attempt) {
        const uint64_t akey = mix64(base + (uint64_t)attempt);
        float rho = density;
        if (density < 0.f) {  // np.random.triangular(0, 0.33, 0.5)
            const float u = u01(mix64(akey ^ 0x7269616E67ull));
            const float a = 0.f, c = 0.33f, b = 0.5f;
            rho = (u < (c - a) / (b - a)) ? a + sqrtf(u * (b - a) * (c - a)) : b - sqrtf((1.f - u) * (b - a) * (b - c));
        }
        W obst = 0;
        if (in_map)
            for (int y = 0; y < L; ++y) obst |= (W)(u01(mix64(akey + 0x100 + (uint64_t)(row * L + y))) < rho) << y;
        const W freec = in_map ? (~obst & lmask) : (W)0;
        W avail = freec, elig = freec;
        ok = true;
        int nparts = 0;  // (wave-uniform) partitions cached in s_part for this map
        uint64_t draw = akey ^ 0x64726177ull;
        for (int i = 0; i < N && ok; ++i) {
            int gx = 0, gy = 0, cnt = 0;
            W vis = 0, up = 0, down = 0, left = 0, right = 0;
            for (;;) {  // rejection over candidate goal cells; every rejection removes >= 1 candidate
                const int total = wave_sum<W>(popc_w<W>(elig));
                if (total == 0) {
                    ok = false;
                    break;
                }
                draw = mix64(draw);
                select_bit<W>(elig, (int)(draw % (uint64_t)total), lane, gx, gy);
                if (!NAVI) {
                    int hit = -1;
                    for (int k = 0; k < nparts && hit < 0; ++k)
                        if ((s_part[k][gx] >> gy) & (W)1) hit = k;  // (one address for the whole wave)
                    if (hit >= 0) {
                        vis = s_part[hit][row];
                        cnt = wave_sum<W>(popc_w<W>(vis & avail));
                        if (cnt >= 2) break;
                        elig &= ~vis;
                        continue;
                    }
                }
                vis = (row == gx) ? ((W)1 << gy) : (W)0;
                W fr = vis;
                up = down = left = right = 0;
                for (int it = 0; it < L * L; ++it) {  // same level-synchronous BFS as navi_bfs_kernel
                    W fu = shfl_up_w<W>(fr, 64);
                    if (row == 0) fu = 0;
                    W fd = shfl_down_w<W>(fr, 64);
                    if (row == 63) fd = 0;
                    const W fl = fr << 1, frr = fr >> 1;
                    const W nw = (fu | fd | fl | frr) & freec & ~vis;
                    up |= nw & fu;
                    down |= nw & fd;
                    left |= nw & fl;
                    right |= nw & frr;
                    vis |= nw;
                    fr = nw;
                    if (__ballot(nw != 0) == 0ull) break;
                }
                if (!NAVI && nparts < RESET_CACHED_PARTS) {
                    s_part[nparts][row] = vis;
                    ++nparts;
                    __builtin_amdgcn_wave_barrier();
                }
                cnt = wave_sum<W>(popc_w<W>(vis & avail));
                if (cnt >= 2) break;
                elig &= ~vis;  // this partition holds < 2 remaining cells: never a candidate again
            }
            if (!ok) break;
            const W gbit = (row == gx) ? ((W)1 << gy) : (W)0;
            avail &= ~gbit;
            elig &= ~gbit;
            int sx, sy;
            draw = mix64(draw);
            select_bit<W>(vis & avail, (int)(draw % (uint64_t)(cnt - 1)), lane, sx, sy);
            const W sbit = (row == sx) ? ((W)1 << sy) : (W)0;
            avail &= ~sbit;
            elig &= ~sbit;
            if (cnt - 2 < 2) elig &= ~vis;
            const size_t o = (size_t)e * N + i;
            if (lane == 0) {
                goals[o * 2] = (int16_t)gx;
                goals[o * 2 + 1] = (int16_t)gy;
                agents[o * 2] = (int16_t)sx;
                agents[o * 2 + 1] = (int16_t)sy;
            }
            if (NAVI && in_map) {
                NaviRec<W> rec;
                rec.w[0] = up;
                rec.w[1] = down;
                rec.w[2] = left;
                rec.w[3] = right;
                navi[o * L + row] = rec;
            }
        }
        if (ok && in_map) map_rows[(size_t)e * L + row] = obst;
    }
    if (lane == 0) {
        if (STAGE) {
            if (ok) st_epoch[e] = epoch;
            else st_need[e] = 0, atomicOr(status, kStatusRange);
        } else {
            if (ok) steps[e] = 0;
            else atomicOr(status, kStatusRange);
        }
    }
}

// the staged scenario of every environment that has none for its next epoch (see reset_body<.., STAGE>)
template <typename W>
__global__ void __launch_bounds__(64) stage_fill_kernel(int E, int L, int N, float density, uint64_t seed, W *st_map_rows, int16_t *st_agents,
                                                        int16_t *st_goals, const int32_t *epochs, int32_t *status, int32_t *st_epoch,
                                                        uint8_t *st_need) {
    reset_body<W, false, true>((int)blockIdx.x, E, L, N, nullptr, density, seed, st_map_rows, st_agents, st_goals, nullptr, nullptr,
                               const_cast<int32_t *>(epochs), status, st_epoch, st_need);
}

// mapf_reset_envs with a staged scenario: the flagged environments take theirs over -- map rows, positions, goals, navigation records
// (16-byte chunks) -- and move on to the next epoch.  An environment without a staged scenario for its next epoch raises kStatusStage.
template <typename W>
__global__ void __launch_bounds__(256) stage_swap_kernel(int E, int L, int N, const uint8_t *mask, W *map_rows, int16_t *agents, int16_t *goals,
                                                         NaviRec<W> *navi, int32_t *steps, int32_t *epochs, int32_t *status, const W *st_map_rows,
                                                         const int16_t *st_agents, const int16_t *st_goals, const NaviRec<W> *st_navi,
                                                         const int32_t *st_epoch) {
    const int e = blockIdx.x, tid = threadIdx.x;
    if (e >= E || (mask && !mask[e])) return;
    const int epoch = epochs[e] + 1;
    if (st_epoch[e] != epoch) {
        if (tid == 0) atomicOr(status, kStatusStage);
        return;
    }
    const size_t nrec = (size_t)N * L * sizeof(NaviRec<W>) / 16;
    const uint4 *src = reinterpret_cast<const uint4 *>(st_navi + (size_t)e * N * L);
    uint4 *dst = reinterpret_cast<uint4 *>(navi + (size_t)e * N * L);
    for (size_t i = tid; i < nrec; i += 256) dst[i] = src[i];
    for (int i = tid; i < L; i += 256) map_rows[(size_t)e * L + i] = st_map_rows[(size_t)e * L + i];
    for (int i = tid; i < 2 * N; i += 256) {
        agents[(size_t)e * N * 2 + i] = st_agents[(size_t)e * N * 2 + i];
        goals[(size_t)e * N * 2 + i] = st_goals[(size_t)e * N * 2 + i];
    }
    __syncthreads();  // (every thread has read epochs[e])
    if (tid == 0) {
        epochs[e] = epoch;
        steps[e] = 0;
    }
}

template <typename W>
__global__ void __launch_bounds__(64) reset_kernel(int E, int L, int N, const uint8_t *mask, float density, uint64_t seed,
                                                   W *map_rows, int16_t *agents, int16_t *goals, NaviRec<W> *navi,
                                                   int32_t *steps, int32_t *epochs, int32_t *status) {
    reset_body<W, false>((int)blockIdx.x, E, L, N, mask, density, seed, map_rows, agents, goals, navi, steps, epochs, status);
}

// mapf_reset_envs(mask) of every handle of a set in one launch: workgroup -> (segment, environment); the scenario stream of an
// iteration is seed + the caller's iteration counter in DEVICE memory (the launch can be replayed from a captured graph)
__global__ void __launch_bounds__(64) reset_multi_kernel(const MultiTable *tab, float density, const unsigned long long *tick) {
    int seg = 0;
    const int n = tab->n;
    while (seg < n - 1 && (int)blockIdx.x >= tab->env_end[seg]) ++seg;
    const int e = (int)blockIdx.x - (seg ? tab->env_end[seg - 1] : 0);
    const StepParams &p = tab->seg[seg];
    const uint64_t sd = (uint64_t)tab->reset_seed[seg] + (tick ? (uint64_t)tick[0] : 0ull);
    if (p.L > 32)
        reset_body<uint64_t, true>(e, p.E, p.L, p.N, p.mask, density, sd, const_cast<uint64_t *>(static_cast<const uint64_t *>(p.map_rows)), p.agents,
                             const_cast<int16_t *>(p.goals), const_cast<NaviRec<uint64_t> *>(static_cast<const NaviRec<uint64_t> *>(p.navi)), p.steps,
                             tab->epochs[seg], p.status);
    else
        reset_body<uint32_t, true>(e, p.E, p.L, p.N, p.mask, density, sd, const_cast<uint32_t *>(static_cast<const uint32_t *>(p.map_rows)), p.agents,
                             const_cast<int16_t *>(p.goals), const_cast<NaviRec<uint32_t> *>(static_cast<const NaviRec<uint32_t> *>(p.navi)), p.steps,
                             tab->epochs[seg], p.status);
}

// int8 map [E][L][L] -> bit rows; also range-checks agent/goal positions of a device-side load
template <typename W>
__global__ void pack_map_kernel(int E, int L, const int8_t *maps, W *rows) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)E * L) return;
    const int8_t *m = maps + idx * L;
    W w = 0;
    for (int y = 0; y < L; ++y) w |= (W)(m[y] != 0) << y;
    rows[idx] = w;
}

template <typename W>
__global__ void unpack_map_kernel(int E, int L, const W *rows, int8_t *maps) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)E * L * L) return;
    long long r = idx / L;
    int y = (int)(idx - r * L);
    maps[idx] = (int8_t)((rows[r] >> y) & 1);
}

// partial load: environment ids[k] <- k-th staged scenario (map rows packed on the fly)
template <typename W>
__global__ void scatter_load_kernel(int n, int L, int N, const int32_t *ids, const int8_t *maps, const int16_t *agents,
                                    const int16_t *goals, W *rows, int16_t *dst_agents, int16_t *dst_goals, int32_t *steps,
                                    int E, int32_t *status) {
    const int k = blockIdx.x;
    const int e = ids[k];
    if (e < 0 || e >= E) {
        if (threadIdx.x == 0) atomicOr(status, kStatusRange);
        return;
    }
    for (int r = threadIdx.x; r < L; r += blockDim.x) {
        const int8_t *m = maps + ((size_t)k * L + r) * L;
        W w = 0;
        for (int y = 0; y < L; ++y) w |= (W)(m[y] != 0) << y;
        rows[(size_t)e * L + r] = w;
    }
    for (int q = threadIdx.x; q < N * 2; q += blockDim.x) {
        const int16_t a = agents[(size_t)k * N * 2 + q], g = goals[(size_t)k * N * 2 + q];
        if (a < 0 || a >= L || g < 0 || g >= L) atomicOr(status, kStatusRange);
        dst_agents[(size_t)e * N * 2 + q] = a;
        dst_goals[(size_t)e * N * 2 + q] = g;
    }
    if (threadIdx.x == 0) steps[e] = 0;
}

__global__ void check_positions_kernel(long long n, int L, const int16_t *a, const int16_t *g, int32_t *status) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int v = a[idx], u = g[idx];
    if (v < 0 || v >= L || u < 0 || u >= L) atomicOr(status, kStatusRange);
}

// mapf_set_agents in one launch: copy + range check of the positions, step counters := 0
__global__ void set_agents_kernel(long long n, int L, int E, const int16_t *src, int16_t *dst, int32_t *steps, int32_t *status) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over E*N (row, col) pairs
    if (idx < E) steps[idx] = 0;
    if (idx >= n) return;
    const short2 v = reinterpret_cast<const short2 *>(src)[idx];
    if (v.x < 0 || v.x >= L || v.y < 0 || v.y >= L) atomicOr(status, kStatusRange);
    reinterpret_cast<short2 *>(dst)[idx] = v;
}

// navi records -> uint8 [E][N][4][L][L]
template <typename W>
__global__ void unpack_navi_kernel(long long fields, int L, const NaviRec<W> *navi, uint8_t *out) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over fields*4*L*L
    long long total = fields * 4 * L * L;
    if (idx >= total) return;
    int y = (int)(idx % L);
    long long q = idx / L;
    int x = (int)(q % L);
    q /= L;
    int k = (int)(q % 4);
    long long f = q / 4;
    out[idx] = (uint8_t)((navi[f * L + x].w[k] >> y) & 1);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct mapf_env {
    int E, L, N, R, device;
    bool wide;  // 64-bit rows
    void *map_rows;
    int16_t *agents;
    int16_t *goals;
    void *navi;
    int32_t *steps;
    int32_t *status;
    int32_t *epochs;  // per-environment reset counter (RNG stream of mapf_reset_envs)
    // staged next scenarios (mapf_stage_next): allocated at its first call
    void *st_map_rows = nullptr, *st_navi = nullptr;
    int16_t *st_agents = nullptr, *st_goals = nullptr;
    int32_t *st_epoch = nullptr;
    uint8_t *st_need = nullptr;
    bool staged = false;
    float st_density = 0.f;
    uint64_t st_seed = 0;
    bool loaded, navi_ready;
    float rtab[5];
    int tune_threads;  // 0 = default; MAPF_STEP_THREADS (tuning experiments only)
    int tune_lds_pad;  // extra dynamic LDS bytes per block to cap residency; MAPF_STEP_LDS_PAD
    int tune_ablate;   // MAPF_STEP_ABLATE (timing-only builds; results are wrong)
    int tune_group;    // MAPF_STEP_GROUP: cap on environments per workgroup (1 = never pack)
    int tune_plane;    // MAPF_STEP_PLANE: -1 = automatic (step_use_plane), 0 / 1 = force the field phase's version
    int tune_nt;       // MAPF_STEP_NT: -1 = automatic (step_nt_store), 0 / 1 = force regular / non-temporal observation stores
    unsigned long long *dbg;  // phase-stamp buffer (diagnostics)
};

namespace {

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            std::fprintf(stderr, "mapf_env: %s failed: %s\n", #expr, hipGetErrorString(_e));            \
            return MAPF_ERR_HIP;                                                                        \
        }                                                                                               \
    } while (0)

size_t word_bytes(const mapf_env *h) { return h->wide ? 8 : 4; }

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

size_t step_smem_bytes(const mapf_env *h, int G, bool plane = false) {
    const int WW = 2 * h->R + 1;
    size_t LP = h->L + 2 * h->R, GP = h->L + 2, NP = ((size_t)G * h->N + 7) & ~7;
    size_t rows = (2 * G * LP * word_bytes(h) + 15) & ~(size_t)15;
    size_t grid_q = (GP * GP + 15) >> 4;
    size_t bits_q = (((size_t)G * h->N * 6 * WW * WW + 31) / 32 + 1 + 3) >> 2;
    // the kernel's layout: bit string, three per-agent arrays, flags, then max(step-phase arrays, navi records of the plane field phase)
    const size_t step_part = G * grid_q * 16 + 2 * NP * 2;
    const size_t navi_part = plane ? (size_t)G * h->N * (2 * h->R + 2) * 4 * word_bytes(h) : 0;
    return rows + bits_q * 16 + 3 * NP * 2 + ((4 * (size_t)G + 15) & ~(size_t)15) + (step_part > navi_part ? step_part : navi_part) + 16;
}

// Non-temporal stores for the observation bytes: when one launch writes more observations than the 256 MiB Infinity Cache can keep
// until somebody reads them, writing them THROUGH the cache only evicts the navi records the next launch wants.  Measured on MI355X
// (rocprofv3, us per launch, regular -> non-temporal; observation MB per launch in front): 318 MB (16,384 x 32x32/40) 95.2 -> 78.7;
// 637 MB 184.0 -> 159.3; 239 MB 67.8 -> 60.2; 191 MB (65,536 x 20x20/6) 59.3 -> 54.3; 159 MB 46.2 -> 46.6; 127 MB (262,144 x 10x10/1)
// 35.7 -> 34.9, (2,048 x 64x64/128) 36.9 -> 39.0; 119 MB 31.2 -> 35.1; 96 MB 28.9 -> 32.2; 80 MB (config 2) 19.5 -> 27.1, (64x64/40)
// 24.0 -> 27.9.  Hence from 176 MB of observations per launch on.  (Non-temporal LOADS of the navi records were slower at every size:
// 16,384 x 32x32/40 78.9 -> 88.3, 262,144 x 10x10/1 37.0 -> 44.6 -- neighbouring records share cache lines.)  MAPF_STEP_NT=0/1 overrides
// for tuning runs.
bool step_nt_store(const mapf_env *h) {
    if (h->tune_nt >= 0) return h->tune_nt != 0;
    return (size_t)h->E * h->N * 6 * (2 * h->R + 1) * (2 * h->R + 1) >= ((size_t)176 << 20);
}

// Field phase by (agent, channel) planes (StepParams::plane) costs LDS for the staged navi records (G * N * 10 records of 16 / 32
// bytes, overlaid on the step phase's arrays).  It is used when that still lets as many workgroups onto a CU as the launch can put
// there (wave slots: 8 per SIMD; the launch: ceil(workgroups / 256)); otherwise the first version's per-(agent, row) deposits.
// Measured on MI355X (rocprofv3, us per launch, first version -> planes): 4096 x 32x32/40 21.0 -> 19.5; 4096 x 16x16/40 21.2 -> 19.7;
// 262144 x 10x10/1 39.5 -> 34.7; 4096 x 40x40/16 13.0 -> 12.4; 8192 x 20x20/6 12.3 -> 11.6; forced on where the rule says no:
// 4096 x 64x64/40 23.8 -> 30.3, 2048 x 64x64/128 37.1 -> 55.3 (LDS residency).  A variant for 64-bit rows that stages the four
// 9-bit windows of a record (8 bytes) instead of its four words (32) fits those shapes but was slower everywhere (64x64/128: 53.9;
// 40x40/16: 13.2): the 64-bit shifts move to the (agent, row) lanes, ten rounds of them.
// MAPF_STEP_PLANE=0/1 overrides for tuning runs.
bool step_use_plane(const mapf_env *h, int G, int threads) {
    if (h->tune_plane >= 0) return h->tune_plane != 0;
    if (h->R != 4) return false;  // (the plane packing is written for the 9 x 9 field of view)
    const size_t smem = (step_smem_bytes(h, G, true) + (size_t)h->tune_lds_pad + 1023) & ~(size_t)1023;
    if (smem > 60 * 1024) return false;  // (beyond the default dynamic-LDS limit of a launch)
    const long long by_lds = (160 * 1024) / (long long)smem;
    const long long by_waves = 32 / (threads / 64);
    const long long blocks = (h->E / G + 255) / 256;
    return by_lds >= (blocks < by_waves ? blocks : by_waves);
}

// Environments per workgroup (env_step_kernel's G): with few agents one environment leaves most of a wavefront idle behind a
// fixed instruction stream, and its observation block is not a multiple of 16 bytes.  Packed blocks are one wavefront, need
// 32-bit map rows, G * N agents <= 64, E divisible by G and a 16-byte-granular observation block (G * N a multiple of 8).
// Measured on MI355X (tools/shape_sweep.py under MAPF_STEP_GROUP caps, profiles/r02_shape_sweep.md), us per launch G = 1 / 2 / 4 / 8:
//   10x10, 1 agent, 65,536 envs   51.3 /  -   /  -   / 16.0      (byte stores without packing: 486 B per environment)
//   15x15, 3 agents, 65,536 envs  60.5 /  -   /  -   / 28.9
//   20x20, 6 agents, 32,768 envs  31.9 / 32.4 / 25.6 / 25.8;  8,192 envs 13.5 / 14.0 / 12.5 / 16.6
//   16x16, 8 agents, 4,096 envs   10.0 / 10.5 / 14.0 / 17.8;  32,768 envs 34.2 / 33.0 / 34.2 / 32.1
//   24x24, 12 agents, 4,096 envs  11.5 / 12.4 / 16.4 / 17.2
// i.e. packing pays below 8 agents (4 environments per block, 8 where the agent count is odd or <= 2: the 16-byte rule) and
// costs latency from 8 agents on.  MAPF_STEP_GROUP caps G for tuning runs (1 = never pack).
int step_group(const mapf_env *h, const void *obs) {
    if (h->wide || h->N >= 8 || (reinterpret_cast<uintptr_t>(obs) & 15)) return 1;
    const int cap = h->tune_group > 0 ? h->tune_group : 8;
    const int pref = (h->N <= 2 || (h->N & 1)) ? 8 : 4;
    for (int G = pref; G >= 2; G >>= 1) {
        if (G > cap || G * h->N > 64 || h->E % G != 0 || (G * h->N) % 8 != 0) continue;
        if (h->N > 2 && h->E / G < 1024) continue;  // too few one-wavefront blocks to cover the load latency
        return G;
    }
    return 1;
}

// threads per block (one lane per agent in the step phase), measured on MI355X (tools/shape_sweep.py):
// 32x32 / 40 agents 20.5 us @128 vs 22.0 us @64; 64x64 / 40 agents 23.5 us @64 vs 26.4 us @128; few agents
// (N <= 24) prefer one wavefront per environment.  MAPF_STEP_THREADS overrides for tuning runs.
int step_block_threads(const mapf_env *h) {
    if (h->tune_threads == 64 || h->tune_threads == 128 || h->tune_threads == 256) {
        if (h->tune_threads >= h->N) return h->tune_threads;
    }
    if (h->N <= 64 && (h->N <= 24 || h->L > 32)) return 64;
    // launches of several residency rounds (round 3, 32x32 / 40 agents, us per launch @128 / @256: 8,192 envs 47.0 / 48.4;
    // 12,288 76.0 / 71.9; 16,384 102.8 / 99.0; 32,768 198.6 / 195.2): four waves per environment keep more loads in flight
    if (h->N > 24 && h->N <= 128 && h->L <= 32 && h->E >= 12288) return 256;
    return h->N <= 128 ? 128 : 256;
}

template <typename W, bool DO_STEP, bool DO_OBS, int VEC, int NT>
int launch_step_nt(const mapf_env *h, const StepParams &p, hipStream_t s, size_t smem) {
    const int need = (h->N * 10 + NT - 1) / NT;
    dim3 g(h->E), b(NT);
    if constexpr (!DO_OBS) {
        hipLaunchKernelGGL((env_step_kernel<W, 4, DO_STEP, false, 16, 1, NT, 1>), g, b, smem, s, p);
    } else if constexpr (NT == 64) {
        // (the unrolled rounds all issue their loads and field deposits, predicated off when a round has no task: a round too many
        // costs the one-wavefront blocks of the small shapes ~8 % -- 16 agents need 3 rounds)
        if (need <= 2)
            hipLaunchKernelGGL((env_step_kernel<W, 4, DO_STEP, true, VEC, 2, NT, 1>), g, b, smem, s, p);
        else if (need <= 3)
            hipLaunchKernelGGL((env_step_kernel<W, 4, DO_STEP, true, VEC, 3, NT, 1>), g, b, smem, s, p);
        else if (need <= 4)
            hipLaunchKernelGGL((env_step_kernel<W, 4, DO_STEP, true, VEC, 4, NT, 1>), g, b, smem, s, p);
        else if (need <= 7)
            hipLaunchKernelGGL((env_step_kernel<W, 4, DO_STEP, true, VEC, 7, NT, 1>), g, b, smem, s, p);
        else
            hipLaunchKernelGGL((env_step_kernel<W, 4, DO_STEP, true, VEC, 10, NT, 1>), g, b, smem, s, p);
    } else {
        if (need <= 4)
            hipLaunchKernelGGL((env_step_kernel<W, 4, DO_STEP, true, VEC, 4, NT, 1>), g, b, smem, s, p);
        else
            hipLaunchKernelGGL((env_step_kernel<W, 4, DO_STEP, true, VEC, 10, NT, 1>), g, b, smem, s, p);
    }
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

// G environments per one-wavefront workgroup (32-bit rows, 16-byte stores; see step_group)
template <bool DO_STEP, bool DO_OBS, int G>
int launch_step_packed(const mapf_env *h, const StepParams &p, hipStream_t s, size_t smem) {
    const int need = (G * h->N * 10 + 63) / 64;
    dim3 g(h->E / G), b(64);
    if constexpr (!DO_OBS) {
        hipLaunchKernelGGL((env_step_kernel<uint32_t, 4, DO_STEP, false, 16, 1, 64, G>), g, b, smem, s, p);
    } else {
        if (need <= 2)
            hipLaunchKernelGGL((env_step_kernel<uint32_t, 4, DO_STEP, true, 16, 2, 64, G>), g, b, smem, s, p);
        else if (need <= 3)
            hipLaunchKernelGGL((env_step_kernel<uint32_t, 4, DO_STEP, true, 16, 3, 64, G>), g, b, smem, s, p);
        else if (need <= 4)
            hipLaunchKernelGGL((env_step_kernel<uint32_t, 4, DO_STEP, true, 16, 4, 64, G>), g, b, smem, s, p);
        else if (need <= 7)
            hipLaunchKernelGGL((env_step_kernel<uint32_t, 4, DO_STEP, true, 16, 7, 64, G>), g, b, smem, s, p);
        else
            hipLaunchKernelGGL((env_step_kernel<uint32_t, 4, DO_STEP, true, 16, 10, 64, G>), g, b, smem, s, p);
    }
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

template <typename W, bool DO_STEP, bool DO_OBS, int VEC>
int launch_step_iters(const mapf_env *h, const StepParams &p, hipStream_t s, int threads, size_t smem) {
    if (threads == 64) return launch_step_nt<W, DO_STEP, DO_OBS, VEC, 64>(h, p, s, smem);
    if (threads == 128) return launch_step_nt<W, DO_STEP, DO_OBS, VEC, 128>(h, p, s, smem);
    return launch_step_nt<W, DO_STEP, DO_OBS, VEC, 256>(h, p, s, smem);
}

template <typename W, bool DO_STEP>
int launch_step_vec(const mapf_env *h, const StepParams &p_in, hipStream_t s) {
    StepParams p = p_in;
    const int total = h->N * 6 * 81;
    const int threads = step_block_threads(h);
    const bool a16 = (total % 16 == 0) && ((reinterpret_cast<uintptr_t>(p.obs) & 15) == 0);
    const bool a4 = (total % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.obs) & 3) == 0);
    const bool obs = p.obs != nullptr || p.obs_bits != nullptr;
    p.nt_store = p.obs != nullptr && step_nt_store(h);
    if constexpr (sizeof(W) == 4) {
        const int G = step_group(h, p.obs);
        if (G > 1) {
            p.plane = obs && step_use_plane(h, G, 64);
            const size_t smem_g = step_smem_bytes(h, G, p.plane != 0) + (size_t)h->tune_lds_pad;
            if (G == 8) return obs ? launch_step_packed<DO_STEP, true, 8>(h, p, s, smem_g) : launch_step_packed<DO_STEP, false, 8>(h, p, s, smem_g);
            if (G == 4) return obs ? launch_step_packed<DO_STEP, true, 4>(h, p, s, smem_g) : launch_step_packed<DO_STEP, false, 4>(h, p, s, smem_g);
            return obs ? launch_step_packed<DO_STEP, true, 2>(h, p, s, smem_g) : launch_step_packed<DO_STEP, false, 2>(h, p, s, smem_g);
        }
    }
    p.plane = obs && step_use_plane(h, 1, threads);
    const size_t smem = step_smem_bytes(h, 1, p.plane != 0) + (size_t)h->tune_lds_pad;
    if (p.obs == nullptr && p.obs_bits == nullptr) return launch_step_iters<W, DO_STEP, false, 16>(h, p, s, threads, smem);
    if (p.obs == nullptr) return launch_step_iters<W, DO_STEP, true, 16>(h, p, s, threads, smem);
    if (a16) return launch_step_iters<W, DO_STEP, true, 16>(h, p, s, threads, smem);
    if (a4) return launch_step_iters<W, DO_STEP, true, 4>(h, p, s, threads, smem);
    return launch_step_iters<W, DO_STEP, true, 1>(h, p, s, threads, smem);
}

template <bool DO_STEP>
int launch_step(const mapf_env *h, const StepParams &p, hipStream_t s) {
    return h->wide ? launch_step_vec<uint64_t, DO_STEP>(h, p, s) : launch_step_vec<uint32_t, DO_STEP>(h, p, s);
}

StepParams make_params(mapf_env *h) {
    StepParams p{};
    p.E = h->E;
    p.L = h->L;
    p.N = h->N;
    p.map_rows = h->map_rows;
    p.agents = h->agents;
    p.goals = h->goals;
    p.navi = h->navi;
    p.steps = h->steps;
    p.status = h->status;
    std::memcpy(p.rtab, h->rtab, sizeof(p.rtab));
    p.ablate = h->tune_ablate;
    p.dbg = h->dbg;
    return p;
}

inline unsigned blocks_for(long long n, int threads) { return (unsigned)((n + threads - 1) / threads); }

// ---- host scenario generator (Rng, generate_one, generate_scenarios): host-only C++ kept in its own file so that the CPU
// sanitizer target (tests/test_sanitize_cpu.py) compiles exactly these lines with g++ -fsanitize=address,undefined ----
#include "mapf_generate_host.inc"

}  // namespace

extern "C" {

int mapf_abi_version(void) { return MAPF_ABI_VERSION; }

const char *mapf_strerror(int status) {
    switch (status) {
        case MAPF_OK: return "ok";
        case MAPF_ERR_INVALID_ARG: return "invalid argument";
        case MAPF_ERR_ACTION: return "action index out of range";
        case MAPF_ERR_OVERLAP: return "unique: two agents on one cell";
        case MAPF_ERR_HIP: return "HIP runtime error";
        case MAPF_ERR_UNSUPPORTED: return "unsupported shape";
        case MAPF_ERR_NO_SPACE: return "no empty position";
        case MAPF_ERR_NOT_READY: return "environment not loaded / navi not built";
        default: return "unknown status";
    }
}

int mapf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mapf_create(int num_envs, int map_len, int num_agents, int obs_radius, int device, mapf_env_t **out) {
    if (!out) return MAPF_ERR_INVALID_ARG;
    *out = nullptr;
    if (num_envs < 1 || map_len < 2 || num_agents < 1 || device < 0) return MAPF_ERR_INVALID_ARG;
    if (map_len > 64 || num_agents > 255 || obs_radius != 4) return MAPF_ERR_UNSUPPORTED;
    if (device >= mapf_device_count()) return MAPF_ERR_HIP;
    mapf_env *h = new (std::nothrow) mapf_env();
    if (!h) return MAPF_ERR_HIP;
    h->E = num_envs;
    h->L = map_len;
    h->N = num_agents;
    h->R = obs_radius;
    h->device = device;
    h->wide = map_len > 32;
    h->loaded = h->navi_ready = false;
    h->dbg = nullptr;
    const char *tv = std::getenv("MAPF_STEP_THREADS");
    h->tune_threads = tv ? std::atoi(tv) : 0;
    tv = std::getenv("MAPF_STEP_LDS_PAD");
    h->tune_lds_pad = tv ? std::atoi(tv) : 0;
    tv = std::getenv("MAPF_STEP_ABLATE");
    h->tune_ablate = tv ? std::atoi(tv) : 0;
    tv = std::getenv("MAPF_STEP_GROUP");
    h->tune_group = tv ? std::atoi(tv) : 0;
    tv = std::getenv("MAPF_STEP_PLANE");
    h->tune_plane = tv ? std::atoi(tv) : -1;
    tv = std::getenv("MAPF_STEP_NT");
    h->tune_nt = tv ? std::atoi(tv) : -1;
    const float def[5] = {-0.075f, 0.0f, -0.075f, -0.5f, 3.0f};
    std::memcpy(h->rtab, def, sizeof(def));
    DeviceGuard guard(device);
    if (!guard.ok) {
        delete h;
        return MAPF_ERR_HIP;
    }
    const size_t E = h->E, L = h->L, N = h->N, wb = word_bytes(h);
    hipError_t err = hipSuccess;
    auto alloc = [&err](void **p, size_t bytes) {
        if (err == hipSuccess) err = hipMalloc(p, bytes);
    };
    alloc(&h->map_rows, E * L * wb);
    alloc(reinterpret_cast<void **>(&h->agents), E * N * 2 * sizeof(int16_t));
    alloc(reinterpret_cast<void **>(&h->goals), E * N * 2 * sizeof(int16_t));
    alloc(&h->navi, E * N * L * 4 * wb);
    alloc(reinterpret_cast<void **>(&h->steps), E * sizeof(int32_t));
    alloc(reinterpret_cast<void **>(&h->status), sizeof(int32_t));
    alloc(reinterpret_cast<void **>(&h->epochs), E * sizeof(int32_t));
    if (err == hipSuccess) err = hipMemset(h->steps, 0, E * sizeof(int32_t));
    if (err == hipSuccess) err = hipMemset(h->status, 0, sizeof(int32_t));
    if (err == hipSuccess) err = hipMemset(h->epochs, 0, E * sizeof(int32_t));
    if (err != hipSuccess) {
        std::fprintf(stderr, "mapf_create: %s\n", hipGetErrorString(err));
        mapf_destroy(h);
        return MAPF_ERR_HIP;
    }
    *out = h;
    return MAPF_OK;
}

int mapf_destroy(mapf_env_t *h) {
    if (!h) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(h->device);
    (void)hipFree(h->map_rows);
    (void)hipFree(h->agents);
    (void)hipFree(h->goals);
    (void)hipFree(h->navi);
    (void)hipFree(h->steps);
    (void)hipFree(h->status);
    (void)hipFree(h->epochs);
    (void)hipFree(h->st_map_rows);
    (void)hipFree(h->st_navi);
    (void)hipFree(h->st_agents);
    (void)hipFree(h->st_goals);
    (void)hipFree(h->st_epoch);
    (void)hipFree(h->st_need);
    delete h;
    return MAPF_OK;
}

int mapf_set_reward_table(mapf_env_t *h, const float table[5]) {
    if (!h || !table) return MAPF_ERR_INVALID_ARG;
    std::memcpy(h->rtab, table, sizeof(h->rtab));
    return MAPF_OK;
}

int mapf_load(mapf_env_t *h, const int8_t *maps, const int16_t *agents, const int16_t *goals,
              int src_on_device, void *stream) {
    if (!h || !maps || !agents || !goals) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t E = h->E, L = h->L, N = h->N;
    const size_t pos_bytes = E * N * 2 * sizeof(int16_t);
    const int8_t *maps_dev = maps;
    int8_t *tmp = nullptr;
    if (!src_on_device) {
        for (size_t k = 0; k < E * N * 2; ++k)
            if (agents[k] < 0 || agents[k] >= (int)L || goals[k] < 0 || goals[k] >= (int)L) return MAPF_ERR_INVALID_ARG;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), E * L * L));
        HIP_TRY(hipMemcpyAsync(tmp, maps, E * L * L, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(h->agents, agents, pos_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(h->goals, goals, pos_bytes, hipMemcpyHostToDevice, s));
        maps_dev = tmp;
    } else {
        HIP_TRY(hipMemcpyAsync(h->agents, agents, pos_bytes, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(h->goals, goals, pos_bytes, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(check_positions_kernel, dim3(blocks_for(E * N * 2, 256)), dim3(256), 0, s,
                           (long long)(E * N * 2), h->L, h->agents, h->goals, h->status);
    }
    if (h->wide)
        hipLaunchKernelGGL(pack_map_kernel<uint64_t>, dim3(blocks_for(E * L, 256)), dim3(256), 0, s, h->E, h->L,
                           maps_dev, static_cast<uint64_t *>(h->map_rows));
    else
        hipLaunchKernelGGL(pack_map_kernel<uint32_t>, dim3(blocks_for(E * L, 256)), dim3(256), 0, s, h->E, h->L,
                           maps_dev, static_cast<uint32_t *>(h->map_rows));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(h->steps, 0, E * sizeof(int32_t), s));
    if (tmp) {
        HIP_TRY(hipStreamSynchronize(s));  // host source buffers and the staging copy may be released on return
        HIP_TRY(hipFree(tmp));
    }
    h->loaded = true;
    h->navi_ready = false;
    return MAPF_OK;
}

int mapf_load_envs(mapf_env_t *h, const int32_t *env_ids, int n, const int8_t *maps, const int16_t *agents,
                   const int16_t *goals, void *stream) {
    if (!h || !env_ids || !maps || !agents || !goals || n < 0) return MAPF_ERR_INVALID_ARG;
    if (!h->loaded || !h->navi_ready) return MAPF_ERR_NOT_READY;
    if (n == 0) return MAPF_OK;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t L = h->L, N = h->N;
    const size_t b_ids = ((size_t)n * 4 + 15) & ~(size_t)15, b_maps = ((size_t)n * L * L + 15) & ~(size_t)15,
                 b_pos = ((size_t)n * N * 4 + 15) & ~(size_t)15;
    unsigned char *tmp = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), b_ids + b_maps + 2 * b_pos));
    int32_t *d_ids = reinterpret_cast<int32_t *>(tmp);
    int8_t *d_maps = reinterpret_cast<int8_t *>(tmp + b_ids);
    int16_t *d_ag = reinterpret_cast<int16_t *>(tmp + b_ids + b_maps), *d_go = reinterpret_cast<int16_t *>(tmp + b_ids + b_maps + b_pos);
    HIP_TRY(hipMemcpyAsync(d_ids, env_ids, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_maps, maps, (size_t)n * L * L, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_ag, agents, (size_t)n * N * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_go, goals, (size_t)n * N * 4, hipMemcpyHostToDevice, s));
    const long long slots = (long long)n * h->N;
    if (h->wide) {
        hipLaunchKernelGGL(scatter_load_kernel<uint64_t>, dim3(n), dim3(128), 0, s, n, h->L, h->N, d_ids, d_maps, d_ag, d_go,
                           static_cast<uint64_t *>(h->map_rows), h->agents, h->goals, h->steps, h->E, h->status);
        hipLaunchKernelGGL(navi_bfs_kernel<uint64_t>, dim3(blocks_for(slots * 64, 256)), dim3(256), 0, s, n, h->L, h->N,
                           static_cast<const uint64_t *>(h->map_rows), h->goals, static_cast<NaviRec<uint64_t> *>(h->navi),
                           h->status, d_ids);
    } else {
        hipLaunchKernelGGL(scatter_load_kernel<uint32_t>, dim3(n), dim3(128), 0, s, n, h->L, h->N, d_ids, d_maps, d_ag, d_go,
                           static_cast<uint32_t *>(h->map_rows), h->agents, h->goals, h->steps, h->E, h->status);
        hipLaunchKernelGGL(navi_bfs_kernel<uint32_t>, dim3(blocks_for(((slots + 1) / 2) * 64, 256)), dim3(256), 0, s, n, h->L,
                           h->N, static_cast<const uint32_t *>(h->map_rows), h->goals,
                           static_cast<NaviRec<uint32_t> *>(h->navi), h->status, d_ids);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipFree(tmp));
    return MAPF_OK;
}

int mapf_stage_next(mapf_env_t *h, float density, uint64_t seed, void *stream) {
    if (!h || density >= 1.0f) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t E = h->E, L = h->L, N = h->N, wb = word_bytes(h);
    if (!h->st_epoch) {  // first call: the staging arrays (a second copy of the scenario state) + "nothing staged"
        hipError_t err = hipSuccess;
        auto alloc = [&err](void **p, size_t bytes) {
            if (err == hipSuccess) err = hipMalloc(p, bytes);
        };
        alloc(&h->st_map_rows, E * L * wb);
        alloc(reinterpret_cast<void **>(&h->st_agents), E * N * 2 * sizeof(int16_t));
        alloc(reinterpret_cast<void **>(&h->st_goals), E * N * 2 * sizeof(int16_t));
        alloc(&h->st_navi, E * N * L * 4 * wb);
        alloc(reinterpret_cast<void **>(&h->st_epoch), E * sizeof(int32_t));
        alloc(reinterpret_cast<void **>(&h->st_need), E);
        if (err == hipSuccess) err = hipMemsetAsync(h->st_epoch, 0, E * sizeof(int32_t), s);
        if (err != hipSuccess) {
            // nothing half-allocated survives: a later call starts over (st_epoch is the "staging arrays exist" test above), and the
            // handle stays in direct-draw mode (h->staged untouched)
            std::fprintf(stderr, "mapf_stage_next: %s\n", hipGetErrorString(err));
            (void)hipFree(h->st_map_rows);
            (void)hipFree(h->st_navi);
            (void)hipFree(h->st_agents);
            (void)hipFree(h->st_goals);
            (void)hipFree(h->st_epoch);
            (void)hipFree(h->st_need);
            h->st_map_rows = h->st_navi = nullptr;
            h->st_agents = h->st_goals = nullptr;
            h->st_epoch = nullptr;
            h->st_need = nullptr;
            (void)hipGetLastError();
            return MAPF_ERR_HIP;
        }
    } else if (h->staged && (h->st_seed != seed || h->st_density != density)) {
        HIP_TRY(hipMemsetAsync(h->st_epoch, 0, E * sizeof(int32_t), s));  // another scenario stream: what is staged belongs to the old one
    }
    h->staged = true;
    h->st_seed = seed;
    h->st_density = density;
    const long long fields = (long long)h->E * h->N;
    if (h->wide) {
        hipLaunchKernelGGL(stage_fill_kernel<uint64_t>, dim3(h->E), dim3(64), 0, s, h->E, h->L, h->N, density, seed, static_cast<uint64_t *>(h->st_map_rows),
                           h->st_agents, h->st_goals, h->epochs, h->status, h->st_epoch, h->st_need);
        hipLaunchKernelGGL(navi_bfs_kernel<uint64_t>, dim3(blocks_for(fields * 64, 256)), dim3(256), 0, s, h->E, h->L, h->N,
                           static_cast<const uint64_t *>(h->st_map_rows), h->st_goals, static_cast<NaviRec<uint64_t> *>(h->st_navi), h->status,
                           (const int32_t *)nullptr, h->st_need);
    } else {
        hipLaunchKernelGGL(stage_fill_kernel<uint32_t>, dim3(h->E), dim3(64), 0, s, h->E, h->L, h->N, density, seed, static_cast<uint32_t *>(h->st_map_rows),
                           h->st_agents, h->st_goals, h->epochs, h->status, h->st_epoch, h->st_need);
        hipLaunchKernelGGL(navi_bfs_kernel<uint32_t>, dim3(blocks_for(((fields + 1) / 2) * 64, 256)), dim3(256), 0, s, h->E, h->L, h->N,
                           static_cast<const uint32_t *>(h->st_map_rows), h->st_goals, static_cast<NaviRec<uint32_t> *>(h->st_navi), h->status,
                           (const int32_t *)nullptr, h->st_need);
    }
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_reset_envs(mapf_env_t *h, const uint8_t *mask_dev, float density, uint64_t seed, void *stream) {
    if (!h || density >= 1.0f) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h->staged && h->st_seed == seed && h->st_density == density) {
        // the flagged environments take over the scenario mapf_stage_next drew for their next epoch: one copy launch where the direct
        // path below draws, places and floods N fields (~0.1 ms on the actor's critical path whenever an episode ended)
        if (h->wide)
            hipLaunchKernelGGL(stage_swap_kernel<uint64_t>, dim3(h->E), dim3(256), 0, s, h->E, h->L, h->N, mask_dev, static_cast<uint64_t *>(h->map_rows),
                               h->agents, h->goals, static_cast<NaviRec<uint64_t> *>(h->navi), h->steps, h->epochs, h->status,
                               static_cast<const uint64_t *>(h->st_map_rows), h->st_agents, h->st_goals,
                               static_cast<const NaviRec<uint64_t> *>(h->st_navi), h->st_epoch);
        else
            hipLaunchKernelGGL(stage_swap_kernel<uint32_t>, dim3(h->E), dim3(256), 0, s, h->E, h->L, h->N, mask_dev, static_cast<uint32_t *>(h->map_rows),
                               h->agents, h->goals, static_cast<NaviRec<uint32_t> *>(h->navi), h->steps, h->epochs, h->status,
                               static_cast<const uint32_t *>(h->st_map_rows), h->st_agents, h->st_goals,
                               static_cast<const NaviRec<uint32_t> *>(h->st_navi), h->st_epoch);
        HIP_TRY(hipGetLastError());
        if (!mask_dev) h->loaded = h->navi_ready = true;
        return MAPF_OK;
    }
    if (h->wide)
        hipLaunchKernelGGL(reset_kernel<uint64_t>, dim3(h->E), dim3(64), 0, s, h->E, h->L, h->N, mask_dev, density, seed,
                           static_cast<uint64_t *>(h->map_rows), h->agents, h->goals, static_cast<NaviRec<uint64_t> *>(h->navi),
                           h->steps, h->epochs, h->status);
    else
        hipLaunchKernelGGL(reset_kernel<uint32_t>, dim3(h->E), dim3(64), 0, s, h->E, h->L, h->N, mask_dev, density, seed,
                           static_cast<uint32_t *>(h->map_rows), h->agents, h->goals, static_cast<NaviRec<uint32_t> *>(h->navi),
                           h->steps, h->epochs, h->status);
    HIP_TRY(hipGetLastError());
    // the navigation fields of the re-drawn environments: one wavefront per field (two per wavefront on <= 32 x 32 maps)
    const long long fields = (long long)h->E * h->N;
    if (h->wide)
        hipLaunchKernelGGL(navi_bfs_kernel<uint64_t>, dim3(blocks_for(fields * 64, 256)), dim3(256), 0, s, h->E, h->L, h->N,
                           static_cast<const uint64_t *>(h->map_rows), h->goals, static_cast<NaviRec<uint64_t> *>(h->navi), h->status,
                           (const int32_t *)nullptr, mask_dev);
    else
        hipLaunchKernelGGL(navi_bfs_kernel<uint32_t>, dim3(blocks_for(((fields + 1) / 2) * 64, 256)), dim3(256), 0, s, h->E, h->L, h->N,
                           static_cast<const uint32_t *>(h->map_rows), h->goals, static_cast<NaviRec<uint32_t> *>(h->navi), h->status,
                           (const int32_t *)nullptr, mask_dev);
    HIP_TRY(hipGetLastError());
    if (!mask_dev) h->loaded = h->navi_ready = true;  // every environment now holds a complete scenario
    return MAPF_OK;
}

int mapf_set_agents(mapf_env_t *h, const int16_t *agents_dev, void *stream) {
    if (!h || !agents_dev) return MAPF_ERR_INVALID_ARG;
    if (!h->loaded) return MAPF_ERR_NOT_READY;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n = (long long)h->E * h->N;  // one launch, nothing else on the stream: callers rewind inside timed regions
    hipLaunchKernelGGL(set_agents_kernel, dim3(blocks_for(n > h->E ? n : h->E, 256)), dim3(256), 0, s, n, h->L, h->E, agents_dev, h->agents,
                       h->steps, h->status);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_build_navi(mapf_env_t *h, void *stream) {
    if (!h) return MAPF_ERR_INVALID_ARG;
    if (!h->loaded) return MAPF_ERR_NOT_READY;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long fields = (long long)h->E * h->N;
    if (h->wide) {
        const long long waves = fields;
        hipLaunchKernelGGL(navi_bfs_kernel<uint64_t>, dim3(blocks_for(waves * 64, 256)), dim3(256), 0, s, h->E, h->L,
                           h->N, static_cast<const uint64_t *>(h->map_rows), h->goals,
                           static_cast<NaviRec<uint64_t> *>(h->navi), h->status, (const int32_t *)nullptr);
    } else {
        const long long waves = (fields + 1) / 2;
        hipLaunchKernelGGL(navi_bfs_kernel<uint32_t>, dim3(blocks_for(waves * 64, 256)), dim3(256), 0, s, h->E, h->L,
                           h->N, static_cast<const uint32_t *>(h->map_rows), h->goals,
                           static_cast<NaviRec<uint32_t> *>(h->navi), h->status, (const int32_t *)nullptr);
    }
    HIP_TRY(hipGetLastError());
    h->navi_ready = true;
    return MAPF_OK;
}

int mapf_obs_bits_row_dwords(const mapf_env_t *h) {
    return h ? (((h->N * 486 + 31) / 32 + 3) & ~3) : MAPF_ERR_INVALID_ARG;
}

int mapf_step(mapf_env_t *h, const int8_t *actions_dev, uint8_t *obs_dev, uint32_t *obs_bits_dev, int16_t *pos_dev,
              int8_t *reward_class_dev, float *reward_dev, uint8_t *done_dev, void *stream) {
    if (!h || !actions_dev) return MAPF_ERR_INVALID_ARG;
    if (!h->loaded || ((obs_dev || obs_bits_dev) && !h->navi_ready)) return MAPF_ERR_NOT_READY;
    DeviceGuard guard(h->device);
    StepParams p = make_params(h);
    p.actions = actions_dev;
    p.obs = obs_dev;
    p.obs_bits = obs_bits_dev;
    p.obs_bits_rd = mapf_obs_bits_row_dwords(h);
    p.pos_out = pos_dev;
    p.rclass = reward_class_dev;
    p.reward = reward_dev;
    p.done = done_dev;
    return launch_step<true>(h, p, static_cast<hipStream_t>(stream));
}

int mapf_observe(mapf_env_t *h, uint8_t *obs_dev, uint32_t *obs_bits_dev, int16_t *pos_dev, void *stream) {
    if (!h || (!obs_dev && !obs_bits_dev)) return MAPF_ERR_INVALID_ARG;
    if (!h->loaded || !h->navi_ready) return MAPF_ERR_NOT_READY;
    DeviceGuard guard(h->device);
    StepParams p = make_params(h);
    p.obs = obs_dev;
    p.obs_bits = obs_bits_dev;
    p.obs_bits_rd = mapf_obs_bits_row_dwords(h);
    p.pos_out = pos_dev;
    return launch_step<false>(h, p, static_cast<hipStream_t>(stream));
}

int mapf_observe_masked(mapf_env_t *h, const uint8_t *mask_dev, uint8_t *obs_dev, uint32_t *obs_bits_dev, int16_t *pos_dev, void *stream) {
    if (!h || (!obs_dev && !obs_bits_dev)) return MAPF_ERR_INVALID_ARG;
    if (!h->loaded || !h->navi_ready) return MAPF_ERR_NOT_READY;
    DeviceGuard guard(h->device);
    StepParams p = make_params(h);
    p.mask = mask_dev;
    p.obs = obs_dev;
    p.obs_bits = obs_bits_dev;
    p.obs_bits_rd = mapf_obs_bits_row_dwords(h);
    p.pos_out = pos_dev;
    return launch_step<false>(h, p, static_cast<hipStream_t>(stream));
}

// ---- several handles in one launch -------------------------------------------------------------------------------------------
struct mapf_multi {
    int n, device;
    int blocks, envs;
    size_t smem;
    MultiTable *table_dev;
    mapf_env *env[kMultiMax];
};

namespace {
// the body instantiation for one handle in a multi-handle launch (the single-handle rules of launch_step_vec, one-wavefront shapes)
int multi_variant(const mapf_env *h, const void *obs, int *G_out) {
    const int total = h->N * 6 * 81;
    const int need1 = (h->N * 10 + 63) / 64;
    if (h->R != 4 || step_block_threads(h) != 64 || h->N > 25) return -1;
    if (!h->wide) {
        const int G = step_group(h, obs);
        if (G == 8 || G == 4) {
            const int need = (G * h->N * 10 + 63) / 64;
            *G_out = G;
            if (G == 8) return need <= 2 ? MV_U32_G8_I2 : (need <= 4 ? MV_U32_G8_I4 : (need <= 7 ? MV_U32_G8_I7 : -1));
            return need <= 4 ? MV_U32_G4_I4 : -1;
        }
    }
    *G_out = 1;
    const bool a16 = (total % 16 == 0) && ((reinterpret_cast<uintptr_t>(obs) & 15) == 0);
    const bool a4 = (total % 4 == 0) && ((reinterpret_cast<uintptr_t>(obs) & 3) == 0);
    const int base = h->wide ? MV_U64_G1_V16_I2 : MV_U32_G1_V16_I2;
    const int vec = (obs == nullptr || a16) ? 0 : (a4 ? 2 : 4);
    if (need1 > 4) return -1;
    return base + vec + (need1 <= 2 ? 0 : 1);
}
}  // namespace

int mapf_multi_create(int n, mapf_env_t *const *envs, const int8_t *const *actions_dev, uint8_t *const *obs_dev, uint32_t *const *obs_bits_dev,
                      int16_t *const *pos_dev, int8_t *const *reward_class_dev, float *const *reward_dev, uint8_t *const *done_dev,
                      const uint8_t *const *mask_dev, const uint64_t *reset_seeds, mapf_multi_t **out) {
    if (!out) return MAPF_ERR_INVALID_ARG;
    *out = nullptr;
    if (n < 1 || !envs || !actions_dev || !obs_dev || !obs_bits_dev || !pos_dev || !reward_class_dev || !reward_dev || !done_dev || !mask_dev)
        return MAPF_ERR_INVALID_ARG;
    if (n > kMultiMax) return MAPF_ERR_UNSUPPORTED;
    MultiTable tab{};
    mapf_multi *m = new (std::nothrow) mapf_multi();
    if (!m) return MAPF_ERR_HIP;
    tab.n = m->n = n;
    m->device = envs[0] ? envs[0]->device : 0;
    int blocks = 0, total_envs = 0;
    size_t smem = 0;
    for (int i = 0; i < n; ++i) {
        mapf_env *h = envs[i];
        if (!h || !actions_dev[i] || (!obs_dev[i] && !obs_bits_dev[i]) || h->device != m->device) {
            delete m;
            return MAPF_ERR_INVALID_ARG;
        }
        if (!h->loaded || !h->navi_ready) {
            delete m;
            return MAPF_ERR_NOT_READY;
        }
        int G = 1;
        const int v = multi_variant(h, obs_dev[i], &G);
        if (v < 0) {
            delete m;
            return MAPF_ERR_UNSUPPORTED;
        }
        StepParams p = make_params(h);
        p.actions = actions_dev[i];
        p.obs = obs_dev[i];
        p.obs_bits = obs_bits_dev[i];
        p.obs_bits_rd = mapf_obs_bits_row_dwords(h);
        p.pos_out = pos_dev[i];
        p.rclass = reward_class_dev[i];
        p.reward = reward_dev[i];
        p.done = done_dev[i];
        p.mask = mask_dev[i];  // (read by the masked re-observation and the reset only: the step body ignores it)
        p.nt_store = 0;
        blocks += h->E / G;
        total_envs += h->E;
        m->env[i] = h;
        tab.seg[i] = p;
        tab.variant[i] = v;
        tab.block_end[i] = blocks;
        tab.env_end[i] = total_envs;
        tab.epochs[i] = h->epochs;
        tab.reset_seed[i] = reset_seeds ? reset_seeds[i] : (uint64_t)i * 0xD1B54A32D192ED03ull;
    }
    // the field phase by planes where every segment could take it next to the others (one dynamic-LDS size per launch)
    for (int i = 0; i < n; ++i) {
        const mapf_env *h = m->env[i];
        const int G = (h->E + (tab.block_end[i] - (i ? tab.block_end[i - 1] : 0)) - 1) / (tab.block_end[i] - (i ? tab.block_end[i - 1] : 0));
        bool plane = h->tune_plane >= 0 ? h->tune_plane != 0 : (h->R == 4);
        if (plane) {
            const size_t sm = (step_smem_bytes(h, G, true) + 1023) & ~(size_t)1023;
            const long long by_lds = (160 * 1024) / (long long)sm, rounds = (blocks + 255) / 256;
            if (sm > 60 * 1024 || by_lds < (rounds < 32 ? rounds : 32)) plane = false;
        }
        tab.seg[i].plane = plane;
        const size_t sm = step_smem_bytes(h, G, plane);
        if (sm > smem) smem = sm;
    }
    // non-temporal observation stores when ONE launch of the set writes more observation bytes than the Infinity Cache holds on to
    // (the single-handle rule, step_nt_store, on the set's total)
    size_t obs_bytes = 0;
    for (int i = 0; i < n; ++i)
        if (tab.seg[i].obs) obs_bytes += (size_t)m->env[i]->E * m->env[i]->N * 486;
    const bool nt = m->env[0]->tune_nt >= 0 ? m->env[0]->tune_nt != 0 : obs_bytes >= ((size_t)176 << 20);
    for (int i = 0; i < n; ++i) tab.seg[i].nt_store = (nt && tab.seg[i].obs) ? 1 : 0;
    m->blocks = blocks;
    m->envs = total_envs;
    m->smem = smem;
    DeviceGuard guard(m->device);
    if (hipMalloc(reinterpret_cast<void **>(&m->table_dev), sizeof(MultiTable)) != hipSuccess) {
        delete m;
        return MAPF_ERR_HIP;
    }
    if (hipMemcpy(m->table_dev, &tab, sizeof(MultiTable), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(m->table_dev);
        delete m;
        return MAPF_ERR_HIP;
    }
    *out = m;
    return MAPF_OK;
}

int mapf_multi_destroy(mapf_multi_t *m) {
    if (!m) return MAPF_OK;
    DeviceGuard guard(m->device);
    (void)hipFree(m->table_dev);
    delete m;
    return MAPF_OK;
}

int mapf_multi_num_workgroups(const mapf_multi_t *m) { return m ? m->blocks : MAPF_ERR_INVALID_ARG; }

int mapf_multi_step(mapf_multi_t *m, void *stream) {
    if (!m) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(m->device);
    hipLaunchKernelGGL(env_step_multi_kernel<true>, dim3(m->blocks), dim3(64), m->smem, static_cast<hipStream_t>(stream), m->table_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_multi_observe_masked(mapf_multi_t *m, void *stream) {
    if (!m) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(m->device);
    hipLaunchKernelGGL(env_step_multi_kernel<false>, dim3(m->blocks), dim3(64), m->smem, static_cast<hipStream_t>(stream), m->table_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_multi_reset(mapf_multi_t *m, float density, const uint64_t *tick_dev, void *stream) {
    if (!m || density >= 1.0f || (reinterpret_cast<uintptr_t>(tick_dev) & 7)) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(m->device);
    hipLaunchKernelGGL(reset_multi_kernel, dim3(m->envs), dim3(64), 0, static_cast<hipStream_t>(stream), m->table_dev, density,
                       reinterpret_cast<const unsigned long long *>(tick_dev));
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_get_navi(mapf_env_t *h, uint8_t *navi_dev, void *stream) {
    if (!h || !navi_dev) return MAPF_ERR_INVALID_ARG;
    if (!h->navi_ready) return MAPF_ERR_NOT_READY;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long fields = (long long)h->E * h->N;
    const long long total = fields * 4 * h->L * h->L;
    if (h->wide)
        hipLaunchKernelGGL(unpack_navi_kernel<uint64_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s, fields, h->L,
                           static_cast<const NaviRec<uint64_t> *>(h->navi), navi_dev);
    else
        hipLaunchKernelGGL(unpack_navi_kernel<uint32_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s, fields, h->L,
                           static_cast<const NaviRec<uint32_t> *>(h->navi), navi_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_get_agents(mapf_env_t *h, int16_t *agents_dev, void *stream) {
    if (!h || !agents_dev) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(h->device);
    HIP_TRY(hipMemcpyAsync(agents_dev, h->agents, (size_t)h->E * h->N * 2 * sizeof(int16_t), hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(stream)));
    return MAPF_OK;
}

int mapf_get_goals(mapf_env_t *h, int16_t *goals_dev, void *stream) {
    if (!h || !goals_dev) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(h->device);
    HIP_TRY(hipMemcpyAsync(goals_dev, h->goals, (size_t)h->E * h->N * 2 * sizeof(int16_t), hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(stream)));
    return MAPF_OK;
}

int mapf_get_maps(mapf_env_t *h, int8_t *maps_dev, void *stream) {
    if (!h || !maps_dev) return MAPF_ERR_INVALID_ARG;
    if (!h->loaded) return MAPF_ERR_NOT_READY;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long total = (long long)h->E * h->L * h->L;
    if (h->wide)
        hipLaunchKernelGGL(unpack_map_kernel<uint64_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s, h->E, h->L,
                           static_cast<const uint64_t *>(h->map_rows), maps_dev);
    else
        hipLaunchKernelGGL(unpack_map_kernel<uint32_t>, dim3(blocks_for(total, 256)), dim3(256), 0, s, h->E, h->L,
                           static_cast<const uint32_t *>(h->map_rows), maps_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_get_steps(mapf_env_t *h, int32_t *steps_dev, void *stream) {
    if (!h || !steps_dev) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(h->device);
    HIP_TRY(hipMemcpyAsync(steps_dev, h->steps, (size_t)h->E * sizeof(int32_t), hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(stream)));
    return MAPF_OK;
}

int mapf_check_status(mapf_env_t *h, void *stream) {
    if (!h) return MAPF_ERR_INVALID_ARG;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int32_t st = 0;
    // behind everything the stream holds, INCLUDING replayed graphs: on this runtime a pageable hipMemcpyAsync + hipStreamSynchronize did
    // not wait for a preceding hipGraphLaunch on the stream (measured: stale reads right after graph replays), an event does
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e1 = hipEventRecord(ev, s);
    if (e1 == hipSuccess) e1 = hipEventSynchronize(ev);
    (void)hipEventDestroy(ev);
    HIP_TRY(e1);
    HIP_TRY(hipMemcpy(&st, h->status, sizeof(st), hipMemcpyDeviceToHost));
    if (st != 0) {
        HIP_TRY(hipMemsetAsync(h->status, 0, sizeof(int32_t), s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (st & kStatusRange) return MAPF_ERR_INVALID_ARG;
    if (st & kStatusAction) return MAPF_ERR_ACTION;
    if (st & kStatusOverlap) return MAPF_ERR_OVERLAP;
    if (st & kStatusStage) return MAPF_ERR_NOT_READY;
    return MAPF_OK;
}

// diagnostics only (not part of include/mapf_env.h): device buffer uint64[E][8] receiving s_memtime phase stamps
int mapf_debug_set_stamps(mapf_env_t *h, unsigned long long *buf_dev) {
    if (!h) return MAPF_ERR_INVALID_ARG;
    h->dbg = buf_dev;
    return MAPF_OK;
}

int mapf_num_envs(const mapf_env_t *h) { return h ? h->E : MAPF_ERR_INVALID_ARG; }
int mapf_map_len(const mapf_env_t *h) { return h ? h->L : MAPF_ERR_INVALID_ARG; }
int mapf_num_agents(const mapf_env_t *h) { return h ? h->N : MAPF_ERR_INVALID_ARG; }
int mapf_obs_radius(const mapf_env_t *h) { return h ? h->R : MAPF_ERR_INVALID_ARG; }

int mapf_generate(int num_envs, int map_len, int num_agents, float density, uint64_t seed, int8_t *maps,
                  int16_t *agents, int16_t *goals, int32_t *redraws) {
    return generate_scenarios(num_envs, map_len, num_agents, density, seed, maps, agents, goals, redraws);
}

}  // extern "C"
