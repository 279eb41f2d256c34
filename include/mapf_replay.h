/*
 * mapf_replay.h -- C ABI of the on-device prioritized episode replay (libmapf_env.so).
 *
 * Replaces, for the DQN hot path of ZiyuanMa/MAPF_RL: buffer.SumTree (reference buffer.py:16-105) and
 * worker.GlobalBuffer.add / sample_batch / update_priorities (reference worker.py:71-203).  The ring of
 * `capacity` episode slots, the f64 sum tree over capacity*256 transition leaves and all episode data live
 * in HBM; sampling is a tree descent + one gather kernel that expands bit-packed observations into the
 * learner's [B, 18, A, 6, 9, 9] bf16 window tensor.
 *
 * Same conventions as mapf_env.h: plain C, int status returns (MAPF_* codes), *_dev = caller-owned device
 * memory, `stream` = hipStream_t as void*.  The ring pointer / size / counter (GlobalBuffer.ptr / .size / .counter)
 * live on the DEVICE: a vectorised actor appends the episodes of all environments that finished in a step with
 * one call (mapf_replay_add_many) and no host round trip; the getters below read them back (and block the host).
 *
 * Storage per slot p (A = max_agents, RD = mapf_replay_row_dwords, CW = (A+31)/32):
 *   obs_bits  uint32 [257][RD]     bit (a*486 + c*81 + cell) of row t = obs[t][a][c][cell]   (worker.py:36)
 *   comm_bits uint32 [257][A][CW]  bit j of word [a][j/32] = comm_mask[t][a][j]               (worker.py:42)
 *   act u8 [256], rew f16 [256], hid f16 [256][256] (agent 0's hidden state; the reference stores the same
 *   vector in every agent row -- quirk Q4, worker.py:388 + buffer.py:148), done u8, size i32, num_agents i32.
 * Declared deviation: agent rows >= num_agents and obs/comm bits of padded agents are zero; the reference
 * leaves whatever the slot's previous episode wrote there (worker.py:96,99,102 assign only [:num_agents]).
 */
#ifndef MAPF_REPLAY_H
#define MAPF_REPLAY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAPF_REPLAY_MAX_STEPS 256   /* config.max_steps = local_buffer_size (config.py:29,33) */
#define MAPF_REPLAY_BT_STEPS 16     /* config.bt_steps (config.py:30) */
#define MAPF_REPLAY_FWD_STEPS 2     /* config.forward_steps (config.py:65) */
#define MAPF_REPLAY_WINDOW 18

typedef struct mapf_replay mapf_replay_t;

/* GlobalBuffer.__init__ (worker.py:23-42). capacity_episodes * 256 must be a power of two (buffer.py:23). */
int mapf_replay_create(int capacity_episodes, int max_agents, int device, mapf_replay_t **out);
int mapf_replay_destroy(mapf_replay_t *r);

int mapf_replay_row_dwords(const mapf_replay_t *r);  /* RD: dwords per bit-packed observation row */
int mapf_replay_capacity(const mapf_replay_t *r);
/* Host reads of the device-side ring state; each waits for the stream of the handle's most recent call. */
int mapf_replay_ptr(const mapf_replay_t *r);         /* GlobalBuffer.ptr */
int64_t mapf_replay_size(const mapf_replay_t *r);    /* len(GlobalBuffer): stored transitions */
int64_t mapf_replay_counter(const mapf_replay_t *r, int reset); /* GlobalBuffer.counter (worker.py:92,226) */
/* out = {ptr, size, counter, episodes added by the last add call}, read behind `stream` (NULL: the most recent one).
 * Returns MAPF_ERR_NOT_READY (once; `out` is still filled) if a batch was sampled while the ring held no transition since the
 * previous call: the reference's batch_sample asserts there (buffer.py:75-76); here the sample itself cannot fail without a
 * host round trip, so it returns leaf 0 with priority 1 and raises this flag. */
int mapf_replay_state(mapf_replay_t *r, int64_t out[4], void *stream);

/*
 * SumTree.batch_update (buffer.py:95-105): leaves[idx[i]] = alpha > 0 ? pri[i]^alpha : pri[i], then every
 * ancestor is re-summed as left + right in f64 (bit-identical to the reference given equal leaves).
 * Duplicate indices: the entry with the largest i wins (numpy assignment order).
 */
int mapf_replay_tree_update(mapf_replay_t *r, const int64_t *idx_dev, const double *pri_dev, int n, double alpha,
                            void *stream);
/*
 * SumTree.batch_sample (buffer.py:56-78) with the uniform draws supplied by the caller:
 * prefix[k] = k*(sum/n) + u[k] (0 -> 1e-5), descent "left iff prefix <= tree[left]"; u[k] = uniforms[k] when
 * unit_uniforms == 0 (draws already scaled to [0, sum/n), as np.random.uniform(0, interval) returns them) and
 * uniforms[k] * (sum/n) when unit_uniforms != 0 (draws in [0, 1): numpy's own scaling, done on the device so that the
 * caller needs no copy of the root).  old_ptr_dev (optional int64[1]) receives the ring pointer at sample time.
 */
int mapf_replay_tree_sample(mapf_replay_t *r, const double *uniforms_dev, int n, int unit_uniforms, int64_t *idx_dev,
                            double *pri_dev, int64_t *old_ptr_dev, void *stream);
/* Copies the whole tree (2*leaves-1 doubles, root first, leaves last) to a device buffer. */
int mapf_replay_tree_read(mapf_replay_t *r, double *tree_dev, void *stream);

/*
 * GlobalBuffer.add for ONE episode (worker.py:86-104) into slot `ptr`, then ptr = (ptr+1) % capacity.
 *   obs_bits_dev  uint32 [size+1][RD]   (already in the A-agent layout, padded agents zero)
 *   comm_bits_dev uint32 [size+1][A][CW]
 *   act_dev u8 [size], rew_dev f16 [size], hid_dev f16 [size][256], td_dev f64 [256] (zeros past `size`)
 * Priorities written: td^0.6 for all 256 leaves of the slot (worker.py:94).
 */
int mapf_replay_add(mapf_replay_t *r, int num_agents, int size, int done, const uint32_t *obs_bits_dev,
                    const uint32_t *comm_bits_dev, const uint8_t *act_dev, const uint16_t *rew_dev,
                    const uint16_t *hid_dev, const double *td_dev, void *stream);

/*
 * GlobalBuffer.add for EVERY finished environment of a vectorised actor (worker.py:71-104 per episode, in ascending
 * environment order) including LocalBuffer.finish's initial priorities (buffer.py:170-177), entirely on the device: three
 * launches, no host read.  Local buffers of `num_envs` environments with `local_steps` (<= 256) transitions each:
 *   finished_dev u8 [E] (non-zero = append this environment's episode; NULL = all), sizes_dev int64 [E] episode lengths,
 *   done_dev u8 [E], obs_bits_dev uint32 [E][local_steps+1][RD], comm_bits_dev uint32 [E][local_steps+1][A][CW],
 *   act_dev u8 [E][local_steps], rew_dev f16 [E][local_steps], hid_dev f16 [E][local_steps][256] (agent 0's state, quirk Q4),
 *   q_dev f32 [E][local_steps][5] agent 0's Q-values (priorities |r_t + 0.99 r_{t+1} + max Q(s_t) - Q(s_t, a_t)| in f64).
 * With more finished episodes than slots the last `capacity` ones survive, as if added one by one.  A call with nothing
 * finished changes nothing (the kernels return at once).
 */
int mapf_replay_add_many(mapf_replay_t *r, int num_envs, int num_agents, int local_steps, const uint8_t *finished_dev,
                         const int64_t *sizes_dev, const uint8_t *done_dev, const uint32_t *obs_bits_dev,
                         const uint32_t *comm_bits_dev, const uint8_t *act_dev, const uint16_t *rew_dev,
                         const uint16_t *hid_dev, const float *q_dev, void *stream);
/* The same for environments of DIFFERENT agent counts (several curriculum levels whose local buffers lie back to back, all at the
 * replay's row width): num_agents_dev int32 [E] instead of one count.  Episodes are appended in environment order, as by per-level
 * calls one after the other. */
int mapf_replay_add_many_env(mapf_replay_t *r, int num_envs, const int32_t *num_agents_dev, int local_steps, const uint8_t *finished_dev,
                             const int64_t *sizes_dev, const uint8_t *done_dev, const uint32_t *obs_bits_dev,
                             const uint32_t *comm_bits_dev, const uint8_t *act_dev, const uint16_t *rew_dev, const uint16_t *hid_dev,
                             const float *q_dev, void *stream);

/*
 * Per-step bookkeeping of a vectorised actor (reference worker.py:376-414 for E lock-step environments; csrc/mapf_actor.hip).
 * Local buffers as in mapf_replay_add_many (S = local_steps transitions per environment, A = max_agents, CW = (A+31)/32,
 * RDA = row_dwords of the replay, RD = env_row_dwords of the environment handle, RD <= RDA).
 *
 * mapf_actor_record = LocalBuffer.add (buffer.py:140-151; agent 0's q / action / reward / hidden: quirk Q7) at row t_dev[e], then
 *   t_dev[e] += 1, finished[e] = done[e] | (t_dev[e] >= S) (worker.py:390), and for finished environments the comm row behind
 *   the last transition: zeros after `done`, this step's mask on a time-out (quirk Q8, worker.py:399).
 *   q f32 [E][N][5], actions int64 [E][N], reward f32 [E][N], hidden bf16 [E][N][256], comm int32 [E][N][CW] (packed, as
 *   mapf_comm_mask writes it with cw = CW), obs_bits int32 [E][RD] (observation after the step), done u8 [E].
 * mapf_actor_rewind = Actor.reset's local part (worker.py:422-428) for finished environments, called behind the replay append, the
 *   scenario reset and the observation of the fresh scenario: row 0 <- that observation, t <- 0, hidden rows <- 0.
 * mapf_actor_log: counters_dev int64 {episodes, logged}: episodes += finished; the `done` flags of finished environments with
 *   stat_mask[e] != 0 (actor id >= 10, worker.py:74) are appended in environment order to the ring log_dev[log_size].
 */
int mapf_actor_record(int num_envs, int num_agents, int local_steps, int env_row_dwords, int row_dwords, int max_agents,
                      const float *q_dev, const int64_t *actions_dev, const float *reward_dev, const uint16_t *hidden_dev,
                      const int32_t *comm_dev, const int32_t *obs_bits_dev, const uint8_t *done_dev, int64_t *t_dev,
                      float *lb_q_dev, uint8_t *lb_act_dev, uint16_t *lb_rew_dev, uint16_t *lb_hid_dev, int32_t *lb_comm_dev,
                      int32_t *lb_obs_dev, uint8_t *finished_dev, void *stream);
int mapf_actor_rewind(int num_envs, int num_agents, int local_steps, int env_row_dwords, int row_dwords,
                      const uint8_t *finished_dev, const int32_t *obs_bits_dev, int64_t *t_dev, int32_t *lb_obs_dev,
                      uint16_t *hidden_dev, void *stream);
/* Rows (486-byte observations) of obs_dev that differ from prev_dev: their indices are appended to list_dev (any order), their number
 * goes to count_dev [1] (zeroed here), the rows themselves go, in list order, to packed_dev u8 [rows][488] (optional; 486 bytes + 2 of
 * padding per row = MAPF_ENC_PACKED_OBS_STRIDE of mapf_dqn.h: dword stores, and what mapf_encoder_forward_rows reads)
 * and prev_dev is refreshed -- reference worker.py:378 calls the network on every agent every step; an unchanged observation has an
 * unchanged encoding. */
int mapf_obs_changed(const uint8_t *obs_dev, uint8_t *prev_dev, int64_t rows, int32_t *list_dev, int32_t *count_dev, uint8_t *packed_dev,
                     void *stream);
/* Exploration of one actor iteration (worker.py:380-382): actions_dev i64 [E][N] holds the policy's greedy actions; with probability
 * eps_dev[e] (f64 [E]) agent 0's is replaced by a uniform action in [0, 5) (counter-based generator: same (seed, counter) -> same
 * draws); act8_dev i8 [E][N] receives the joint action as the environment step reads it, policy_dev (optional) the greedy actions. */
int mapf_actor_explore(int num_envs, int num_agents, int64_t *actions_dev, int64_t *policy_dev, int8_t *act8_dev,
                       const double *eps_dev, uint64_t seed, uint64_t counter, void *stream);
/* The same with the iteration counter in device memory: draws as for counter + *tick_dev (uint64 [1], 8-byte aligned) -- the launch can
 * be replayed from a captured HIP graph while the caller advances the counter on the device. */
int mapf_actor_explore_dev(int num_envs, int num_agents, int64_t *actions_dev, int64_t *policy_dev, int8_t *act8_dev,
                           const double *eps_dev, uint64_t seed, uint64_t counter, const uint64_t *tick_dev, void *stream);
/*
 * Several levels -- environments of different agent counts whose per-agent arrays (actions, q, reward, hidden, packed comm rows) and
 * per-environment arrays (t, finished, done, eps, local buffers) lie back to back -- by ONE launch each.
 * envtab_dev int32 [E][4] (16-byte aligned) per environment: {agents, first agent row, byte offset of its comm mask
 * (mapf_comm_mask_multi / mapf_recurrent_infer_multi of mapf_dqn.h), dword offset of its bit-packed observation row in obs_bits_dev}.
 *   mapf_actor_explore_multi: aux_dev int64 [E][3] = {exploration seed of the environment's level, base counter, index of the
 *     environment within its level}: the draws are those of mapf_actor_explore_dev(seed, base, tick) for that level alone;
 *   mapf_actor_record_multi / mapf_actor_rewind_multi: mapf_actor_record / mapf_actor_rewind with the per-environment table;
 *     rewind_multi optionally (hidden_out_dev != NULL, a different buffer) writes EVERY environment's hidden rows to hidden_out_dev --
 *     hidden_dev's for the running episodes, zeros for the finished ones -- i.e. the policy's next input, and (tick_dev != NULL)
 *     increments the iteration counter: the two element-wise launches that otherwise end a graph-replayed iteration;
 *   mapf_actor_log_multi: mapf_actor_log per level l over the environments [level_start[l], level_start[l + 1]) (host array of
 *     num_levels + 1 entries, num_levels <= 16) into that level's log / counters (host arrays of device pointers).
 */
int mapf_actor_explore_multi(int num_envs, const int32_t *envtab_dev, const int64_t *aux_dev, int64_t *actions_dev, int64_t *policy_dev,
                             int8_t *act8_dev, const double *eps_dev, const uint64_t *tick_dev, void *stream);
int mapf_actor_record_multi(int num_envs, int local_steps, int row_dwords, int max_agents, const int32_t *envtab_dev, const float *q_dev,
                            const int64_t *actions_dev, const float *reward_dev, const uint16_t *hidden_dev, const int32_t *comm_dev,
                            const int32_t *obs_bits_dev, const uint8_t *done_dev, int64_t *t_dev, float *lb_q_dev, uint8_t *lb_act_dev,
                            uint16_t *lb_rew_dev, uint16_t *lb_hid_dev, int32_t *lb_comm_dev, int32_t *lb_obs_dev, uint8_t *finished_dev,
                            void *stream);
int mapf_actor_rewind_multi(int num_envs, int local_steps, int row_dwords, const int32_t *envtab_dev, const uint8_t *finished_dev,
                            const int32_t *obs_bits_dev, int64_t *t_dev, int32_t *lb_obs_dev, uint16_t *hidden_dev, uint16_t *hidden_out_dev,
                            uint64_t *tick_dev, void *stream);
int mapf_actor_log_multi(int num_levels, const int32_t *level_start, uint8_t *const *log_dev, int64_t *const *counters_dev, int log_size,
                         const uint8_t *finished_dev, const uint8_t *done_dev, const uint8_t *stat_mask_dev, void *stream);
/*
 * Everything of one actor iteration behind the policy's forward (worker.py:380-428), as ONE call: mapf_actor_explore, mapf_step
 * (include/mapf_env.h), mapf_actor_record, mapf_replay_add_many (replay may be NULL), mapf_actor_log, mapf_reset_envs +
 * mapf_observe_masked of the environments whose episode ended, mapf_actor_rewind -- the same launches in the same order; what it
 * saves is the caller's side of eight calls (the vectorised actor of mapf_rl_amd/actor.py is host-bound at curriculum shapes).
 * `a`: the actor's persistent device buffers (shapes as documented at the calls above); per iteration: the policy's greedy actions
 * (in: greedy, out: executed), its Q-values, the new hidden states and the packed comm rows.
 */
typedef struct mapf_actor_state {
    int32_t num_envs, num_agents, local_steps, env_row_dwords, row_dwords, max_agents, log_size, reserved;
    float *lb_q;
    uint8_t *lb_act;
    uint16_t *lb_rew, *lb_hid;
    int32_t *lb_comm, *lb_obs;
    int64_t *t;
    uint8_t *finished;
    int32_t *obs_bits;
    const uint8_t *stat_mask;
    uint8_t *stat_log;
    int64_t *counters;
    const double *eps;
    int64_t *policy_actions;
    int8_t *act8;
    uint8_t *obs;      /* the environment handle's output buffers */
    int16_t *pos;
    int8_t *reward_class;
    float *reward;
    uint8_t *done;
} mapf_actor_state;
int mapf_actor_iteration_tail(const mapf_actor_state *a, void *env, mapf_replay_t *replay, int64_t *actions_dev, const float *q_dev,
                              uint16_t *hidden_dev, const int32_t *comm_dev, uint64_t explore_seed, uint64_t explore_counter,
                              float density, uint64_t scenario_seed, void *stream);
int mapf_actor_log(int num_envs, const uint8_t *finished_dev, const uint8_t *done_dev, const uint8_t *stat_mask_dev,
                   uint8_t *log_dev, int log_size, int64_t *counters_dev, void *stream);

/* Importance-sampling weights of a sampled batch (reference worker.py:165-166): weights[i] = (pri[i] / min pri) ^ -beta, computed in
 * f64 like the reference's numpy, written as f32 [n]. */
int mapf_replay_is_weights(const double *pri_dev, int n, double beta, float *weights_dev, void *stream);

/*
 * GlobalBuffer.sample_batch (worker.py:106-184) minus the IS weights (a reduction the caller does on
 * pri_dev): tree sample + window gather.  Outputs (device):
 *   idx int64 [n], pri f64 [n], obs bf16 [18][n][A][6][9][9], comm u8 [18][n][A][A] (TIME-major: the consumer is a
 *   recurrence over the 18 window steps; the host wrapper hands them out as [n][18]... views, the reference's shape),
 *   hidden f16 [n*A][256], action int64 [n], reward f32 [n], done f32 [n], steps f32 [n], bt_steps int64 [n],
 *   old_ptr int64 [1] (optional) the ring pointer at sample time (worker.py:182).  unit_uniforms: see _tree_sample.
 */
int mapf_replay_sample(mapf_replay_t *r, const double *uniforms_dev, int n, int unit_uniforms, int64_t *idx_dev,
                       double *pri_dev, uint16_t *obs_dev, uint8_t *comm_dev, uint16_t *hidden_dev, int64_t *action_dev,
                       float *reward_dev, float *done_dev, float *steps_dev, int64_t *bt_steps_dev, int64_t *old_ptr_dev,
                       void *stream);

/*
 * GlobalBuffer.update_priorities (worker.py:186-203): entries whose slot was overwritten between *old_ptr_dev
 * (the ring pointer written by mapf_replay_sample / _tree_sample) and now are dropped, the rest get pri^0.6.
 */
int mapf_replay_update_priorities(mapf_replay_t *r, const int64_t *idx_dev, const double *pri_dev, int n,
                                  const int64_t *old_ptr_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MAPF_REPLAY_H */
