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
 * memory, `stream` = hipStream_t as void*.  The ring pointer / size counters are host-side state exactly as
 * in the reference (GlobalBuffer.ptr / .size / .counter).
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
int mapf_replay_ptr(const mapf_replay_t *r);         /* GlobalBuffer.ptr */
int64_t mapf_replay_size(const mapf_replay_t *r);    /* len(GlobalBuffer): stored transitions */
int64_t mapf_replay_counter(const mapf_replay_t *r, int reset); /* GlobalBuffer.counter (worker.py:92,226) */

/*
 * SumTree.batch_update (buffer.py:95-105): leaves[idx[i]] = alpha > 0 ? pri[i]^alpha : pri[i], then every
 * ancestor is re-summed as left + right in f64 (bit-identical to the reference given equal leaves).
 * Duplicate indices: the entry with the largest i wins (numpy assignment order).
 */
int mapf_replay_tree_update(mapf_replay_t *r, const int64_t *idx_dev, const double *pri_dev, int n, double alpha,
                            void *stream);
/*
 * SumTree.batch_sample (buffer.py:56-78) with the uniform draws supplied by the caller:
 * prefix[k] = k*(sum/n) + uniforms[k] (0 -> 1e-5), descent "left iff prefix <= tree[left]".
 */
int mapf_replay_tree_sample(mapf_replay_t *r, const double *uniforms_dev, int n, int64_t *idx_dev, double *pri_dev,
                            void *stream);
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
 * GlobalBuffer.sample_batch (worker.py:106-184) minus the IS weights (a reduction the caller does on
 * pri_dev): tree sample + window gather.  Outputs (device):
 *   idx int64 [n], pri f64 [n], obs bf16 [18][n][A][6][9][9], comm u8 [18][n][A][A] (TIME-major: the consumer is a
 *   recurrence over the 18 window steps; the host wrapper hands them out as [n][18]... views, the reference's shape),
 *   hidden f16 [n*A][256], action int64 [n], reward f32 [n], done f32 [n], steps f32 [n], bt_steps int64 [n].
 */
int mapf_replay_sample(mapf_replay_t *r, const double *uniforms_dev, int n, int64_t *idx_dev, double *pri_dev,
                       uint16_t *obs_dev, uint8_t *comm_dev, uint16_t *hidden_dev, int64_t *action_dev,
                       float *reward_dev, float *done_dev, float *steps_dev, int64_t *bt_steps_dev, void *stream);

/*
 * GlobalBuffer.update_priorities (worker.py:186-203): entries whose slot was overwritten between `old_ptr`
 * (the ring pointer returned with the sample) and now are dropped, the rest get pri^0.6.
 */
int mapf_replay_update_priorities(mapf_replay_t *r, const int64_t *idx_dev, const double *pri_dev, int n, int old_ptr,
                                  void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MAPF_REPLAY_H */
