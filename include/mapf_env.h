/*
 * mapf_env.h -- C ABI of the MI355X-native vectorised MAPF environment (libmapf_env.so).
 *
 * This is the drop-in boundary for the reference's environment hot path (ZiyuanMa/MAPF_RL):
 * the reference has no FFI, the path sits behind the Python class `Environment`
 * (reference environment.py:74-467).  Each entry point names the reference code it replaces.
 * A handle owns E independent lock-step environments of identical shape (L x L map, N agents,
 * FOV radius r) resident in HBM; all heavy entry points are asynchronous launches on the
 * caller's HIP stream.
 *
 * Conventions: plain C types only, no exceptions cross the ABI, every function returns an int
 * status (0 = MAPF_OK, negative = error).  Pointers named *_dev are device pointers owned by the
 * caller; `stream` is a hipStream_t passed as void* (NULL = the default stream).  A handle is not
 * thread-safe; distinct handles are independent.
 *
 * Layouts (row-major, C order):
 *   maps     int8  [E][L][L]      0 free, 1 obstacle                (reference environment.py:82-84)
 *   agents   int16 [E][N][2]      (row, col)                        (reference environment.py:112)
 *   goals    int16 [E][N][2]
 *   actions  int8  [E][N]         0 stay 1 up 2 down 3 left 4 right (reference environment.py:12)
 *   obs      uint8 [E][N][6][2r+1][2r+1]  0/1 per flag              (reference environment.py:444-465)
 *   reward_class int8 [E][N]      index into the reward table below
 *   reward   float [E][N]
 *   done     uint8 [E]
 */
#ifndef MAPF_ENV_H
#define MAPF_ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAPF_ABI_VERSION 2

/* status codes; the reference's exception each one stands for is in the comment */
#define MAPF_OK 0
#define MAPF_ERR_INVALID_ARG (-1)   /* bad pointer / shape / position outside the map */
#define MAPF_ERR_ACTION (-2)        /* AssertionError 'action index out of range', environment.py:289-290 */
#define MAPF_ERR_OVERLAP (-3)       /* RuntimeError('unique'): two agents on one cell, environment.py:424-428 */
#define MAPF_ERR_HIP (-4)           /* a HIP runtime call failed (no GPU, OOM, launch failure) */
#define MAPF_ERR_UNSUPPORTED (-5)   /* shape outside what the kernels are built for (L > 64, N > 255, r != 4) */
#define MAPF_ERR_NO_SPACE (-6)      /* ValueError from placement exhaustion, environment.py:120 */
#define MAPF_ERR_NOT_READY (-7)     /* step/observe before load + build_navi */

/* reward classes = keys of the reference's config.reward_fn (config.py:8-12), in this order */
#define MAPF_RC_MOVE 0          /* -0.075 */
#define MAPF_RC_STAY_ON_GOAL 1  /*  0     */
#define MAPF_RC_STAY_OFF_GOAL 2 /* -0.075 */
#define MAPF_RC_COLLISION 3     /* -0.5   */
#define MAPF_RC_FINISH 4        /*  3     */

typedef struct mapf_env mapf_env_t;

int mapf_abi_version(void);
const char *mapf_strerror(int status);

/* Number of HIP devices visible (0 without a GPU); never fails. */
int mapf_device_count(void);

/*
 * Replaces Environment.__init__'s shape arguments (environment.py:75-76): allocates device state for
 * E environments of map side L (2..64), N agents (1..255), FOV radius r (must be 4: the model's
 * obs_shape (6,9,9) is hard-wired, config.py:14, model.py:148,164,235).
 */
int mapf_create(int num_envs, int map_len, int num_agents, int obs_radius, int device, mapf_env_t **out);
int mapf_destroy(mapf_env_t *env);

/* config.reward_fn (config.py:8-12) in MAPF_RC_* order; default {-0.075, 0, -0.075, -0.5, 3}. */
int mapf_set_reward_table(mapf_env_t *env, const float table[5]);

/*
 * Replaces Environment.load (environment.py:198-215), batched: ingest E scenarios, zero the step
 * counters.  Source buffers are host memory (src_on_device = 0; positions are range-checked) or
 * device memory (src_on_device = 1; an out-of-range position raises the sticky device status).
 * Does NOT build the navi field: call mapf_build_navi next (load = this + build_navi).
 */
int mapf_load(mapf_env_t *env, const int8_t *maps, const int16_t *agents, const int16_t *goals,
              int src_on_device, void *stream);

/*
 * Partial Environment.load + get_navi_map: environment env_ids[k] (HOST int32 [n]) receives the k-th scenario
 * of the HOST buffers maps [n][L][L] / agents [n][N][2] / goals [n][N][2]; its step counter is zeroed and only
 * its N distance fields are rebuilt.  This is the per-environment reset of the actor loop
 * (reference worker.py:422-428 -> environment.py:146-196).  Synchronises `stream`.
 */
int mapf_load_envs(mapf_env_t *env, const int32_t *env_ids, int n, const int8_t *maps, const int16_t *agents,
                   const int16_t *goals, void *stream);

/*
 * On-device Environment.reset (environment.py:146-196) for every environment e with mask_dev[e] != 0
 * (mask_dev == NULL: all): new Bernoulli map, placement by the reference's rule, navi fields, step counter 0 --
 * one launch, no host round trip (the actor loop's auto-reset).  Own counter-based RNG keyed by
 * (seed, environment, per-environment reset count): statistical, not bitwise, parity with the reference.
 * density < 0 draws rho ~ triangular(0, 0.33, 0.5) per reset.  A map that cannot host the agents is re-drawn.
 */
int mapf_reset_envs(mapf_env_t *env, const uint8_t *mask_dev, float density, uint64_t seed, void *stream);

/*
 * Scenario generation off the caller's critical path (reference worker.py:422-428: an actor draws its next scenario when an episode
 * ends -- but the scenario of (seed, environment, reset count + 1) does not depend on when it is drawn).  mapf_stage_next draws it
 * AHEAD into a second copy of the scenario state, for every environment that has none staged for its next reset (on any stream,
 * e.g. beside the policy's forward pass); from the first call on, mapf_reset_envs with the SAME (density, seed) no longer draws:
 * the flagged environments take their staged scenario over in one copy launch -- bit for bit the scenario the direct reset would have
 * drawn.  The caller orders the two (stage before the reset that consumes it, the next stage behind that reset); a reset that finds
 * nothing staged raises a sticky error (mapf_check_status: MAPF_ERR_NOT_READY).  A different (density, seed) takes the direct path.
 */
int mapf_stage_next(mapf_env_t *env, float density, uint64_t seed, void *stream);

/* Overwrite agent positions only (e.g. rewind to the start of an action tape); steps := 0. */
int mapf_set_agents(mapf_env_t *env, const int16_t *agents_dev, void *stream);

/*
 * Replaces Environment.get_navi_map (environment.py:217-276) == search.compute_heuristics
 * (search.py:24-55) for all E x N goals: bit-parallel BFS, one wavefront per distance field.
 */
int mapf_build_navi(mapf_env_t *env, void *stream);

/*
 * Replaces Environment.step (environment.py:278-430) incl. the observe() it returns (:430 -> :433-467),
 * for all E environments in one launch.  Any output pointer may be NULL (obs_dev == NULL and
 * obs_bits_dev == NULL skips the observation build).  obs_bits_dev, uint32 [E][mapf_obs_bits_row_dwords()],
 * receives the same observation bit-packed (bit a*486 + c*81 + cell of row e = obs[e][a][c][cell]) -- the
 * storage format of the replay (mapf_replay.h), 8x smaller than obs.  Errors detected on the device (action outside [0,5), overlap) are sticky and
 * reported by mapf_check_status.
 */
int mapf_step(mapf_env_t *env, const int8_t *actions_dev, uint8_t *obs_dev, uint32_t *obs_bits_dev, int16_t *pos_dev,
              int8_t *reward_class_dev, float *reward_dev, uint8_t *done_dev, void *stream);

/* Replaces Environment.observe (environment.py:433-467) for all E environments. */
int mapf_observe(mapf_env_t *env, uint8_t *obs_dev, uint32_t *obs_bits_dev, int16_t *pos_dev, void *stream);

/*
 * The same for the environments e with mask_dev[e] != 0 only (mask_dev == NULL: all): rows of the other environments are
 * left as they are.  For callers whose output buffers already hold the current observation of the unflagged environments --
 * the actor loop after mapf_step + mapf_reset_envs(mask): only the reset environments changed (reference worker.py:422-428
 * -> environment.py:196 returns observe() of the fresh scenario).  Granularity is a workgroup: with several small environments
 * per workgroup the unflagged neighbours of a flagged one are rewritten with identical values.
 */
int mapf_observe_masked(mapf_env_t *env, const uint8_t *mask_dev, uint8_t *obs_dev, uint32_t *obs_bits_dev, int16_t *pos_dev,
                        void *stream);

/*
 * Several handles of DIFFERENT shapes stepped by one launch.  The reference draws a (num_agents, map side) level per episode inside
 * one actor (environment.py:148-151, worker.py:422-428: Environment.reset(level)); here every active level is a handle of a few
 * hundred lock-step environments, and a set binds up to 16 of them (one device; shapes whose workgroup is one wavefront: at most 25
 * agents) together with the buffers their step reads and writes -- per handle i the arguments of mapf_step (actions, obs, obs_bits,
 * pos, reward_class, reward, done) and the u8 [E_i] mask of mapf_observe_masked / mapf_reset_envs (the actor's end-of-episode flags).
 * All pointers are device pointers that must stay valid while the set lives; the arrays of pointers themselves are host arrays read
 * at creation.  MAPF_ERR_UNSUPPORTED for a shape outside the limits (callers then step the handles one by one).
 *   mapf_multi_step             = mapf_step of every handle, ONE launch (same outputs, bit for bit);
 *   mapf_multi_reset            = mapf_reset_envs(mask_i, seed = reset_seeds[i] + *tick_dev) of every handle, ONE launch
 *                                 (reset_seeds: host array given at creation, NULL = i * a constant; tick_dev: uint64 in device
 *                                 memory, optional -- the caller's iteration counter);
 *   mapf_multi_observe_masked   = mapf_observe_masked(mask_i) of every handle, ONE launch.
 * No host scalar has to change from iteration to iteration: the three launches can be replayed from a captured HIP graph.
 */
typedef struct mapf_multi mapf_multi_t;
int mapf_multi_create(int n, mapf_env_t *const *envs, const int8_t *const *actions_dev, uint8_t *const *obs_dev,
                      uint32_t *const *obs_bits_dev, int16_t *const *pos_dev, int8_t *const *reward_class_dev,
                      float *const *reward_dev, uint8_t *const *done_dev, const uint8_t *const *mask_dev,
                      const uint64_t *reset_seeds, mapf_multi_t **out);
int mapf_multi_destroy(mapf_multi_t *set);
int mapf_multi_num_workgroups(const mapf_multi_t *set);
int mapf_multi_step(mapf_multi_t *set, void *stream);
int mapf_multi_reset(mapf_multi_t *set, float density, const uint64_t *tick_dev, void *stream);
int mapf_multi_observe_masked(mapf_multi_t *set, void *stream);

/* Dwords per bit-packed observation row: ceil(N*486/32) rounded up to a multiple of 4. */
int mapf_obs_bits_row_dwords(const mapf_env_t *env);

/* State read-back (device -> caller's device buffers). navi: uint8 [E][N][4][L][L], unpadded. */
int mapf_get_navi(mapf_env_t *env, uint8_t *navi_dev, void *stream);
int mapf_get_agents(mapf_env_t *env, int16_t *agents_dev, void *stream);
int mapf_get_goals(mapf_env_t *env, int16_t *goals_dev, void *stream);
int mapf_get_maps(mapf_env_t *env, int8_t *maps_dev, void *stream);
int mapf_get_steps(mapf_env_t *env, int32_t *steps_dev, void *stream);

/* Synchronises `stream`, returns and clears the sticky device status (MAPF_OK, MAPF_ERR_ACTION, ...). */
int mapf_check_status(mapf_env_t *env, void *stream);

/* Shape queries. */
int mapf_num_envs(const mapf_env_t *env);
int mapf_map_len(const mapf_env_t *env);
int mapf_num_agents(const mapf_env_t *env);
int mapf_obs_radius(const mapf_env_t *env);

/*
 * Host-side scenario generator following the reference's rule (environment.py:100-138,
 * map_partition :21-70): Bernoulli(density) map, keep connected components of >= 2 free cells,
 * per agent pick a component with probability proportional to its size, then start and goal
 * uniformly without replacement from it.  density < 0 draws rho ~ triangular(0, 0.33, 0.5) per
 * environment like the reference.  Own counter-based RNG (statistical, not bitwise, parity with
 * the reference's global Python/numpy RNG).  An environment whose map cannot host N agents is
 * re-drawn (the reference raises ValueError there); `redraws` (may be NULL) receives the total.
 * Outputs are HOST buffers.
 */
int mapf_generate(int num_envs, int map_len, int num_agents, float density, uint64_t seed,
                  int8_t *maps, int16_t *agents, int16_t *goals, int32_t *redraws);

#ifdef __cplusplus
}
#endif
#endif /* MAPF_ENV_H */
