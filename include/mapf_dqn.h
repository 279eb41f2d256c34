/*
 * mapf_dqn.h -- C ABI of the hand-written HIP pieces of the DQN forward/backward (libmapf_env.so).
 *
 * The dense math of the reference's Network (reference model.py:139-263) runs through MIOpen / hipBLASLt;
 * what surrounds each convolution of the observation encoder (model.py:147-162, ResBlock :30-42) -- bias
 * add, residual add, ReLU -- is memory-bound elementwise work that PyTorch issues as 3-4 separate passes over
 * a [M,128,7,7] activation (35 % of the GPU time of an update in profiles/r01_c_*).  These entry points fuse
 * it into ONE pass behind each convolution (and one pass + a bias-gradient reduction in backward).
 *
 * Tensors are bf16 in NHWC (channels_last) memory order: element i belongs to channel i % C; C % 8 == 0.
 */
#ifndef MAPF_DQN_H
#define MAPF_DQN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* y[i] = relu(y[i] + bias[i % C] + (res ? res[i] : 0)), in place.  y, res: bf16 [n]; bias: f32 [C]. */
int mapf_bias_res_relu_fwd(uint16_t *y_dev, const float *bias_dev, const uint16_t *res_dev, int64_t n, int C,
                           void *stream);
/* gx[i] = y[i] > 0 ? g[i] : 0;  gbias[c] += sum_i gx[i] (f32, gbias must be zeroed by the caller). */
int mapf_bias_res_relu_bwd(const uint16_t *g_dev, const uint16_t *y_dev, uint16_t *gx_dev, float *gbias_dev,
                           int64_t n, int C, void *stream);

/*
 * Fused observation encoder, inference only (no autograd): replaces the reference's
 * `self.obs_encoder(obs)` in `Network.step` (model.py:184) and in the target network's `bootstrap`
 * (model.py:237, called without gradients at worker.py:300-303) by ONE kernel that keeps the activations of
 * 4 observations per workgroup in LDS across all 8 convolutions (csrc/mapf_encoder.hip).  f16 MFMA, fp32
 * accumulation, one rounding to f16 per layer output (clamped to +-65504) -- the arithmetic of the reference's fp16
 * autocast (worker.py:283); the OUTPUT is bf16 like the rest of the network's activations.  ("Element" below = IEEE half,
 * 16 bits, carried in uint16_t.)
 *
 * Packed weights (produced by mapf_encoder_pack from the 8 convolutions of `Network.obs_encoder`, in module
 * order: [0], [2].block1, [2].block2, [3].block1, [3].block2, [4].block1, [4].block2, [5]; each weight fp32
 * contiguous [co][ci][kh][kw], each bias fp32 [co]): f16 A-operand fragments of v_mfma_f32_16x16x32_f16,
 *   conv0   [s=2 ][c=8][lane=64][j=8]   k = 32 s + 8 (lane>>4) + j = ci*9 + kh*3 + kw  (k >= 54 -> 0)
 *   3x3 x6  [s=36][c=8][lane=64][j=8]   s = (kh*3+kw)*4 + chunk, ci = 32 chunk + 8 (lane>>4) + j
 *   1x1     [s=4 ]     [lane=64][j=8]   ci = 32 s + 8 (lane>>4) + j
 * with co = 16 c + (lane & 15); MAPF_ENC_PACKED_ELEMS f16 in total.  Biases: MAPF_ENC_BIAS_ELEMS fp32,
 * concatenated in the same order.  Both buffers must be 16-byte aligned.
 */
#define MAPF_ENC_OBS_PER_BLOCK 4
#define MAPF_ENC_PACKED_ELEMS 894976 /* 8192 + 6*147456 + 2048 */
#define MAPF_ENC_BIAS_ELEMS 912      /* 7*128 + 16 */
#define MAPF_ENC_OBS_U8 0            /* obs elements are bytes (0/1, any 0..255 is exact) */
#define MAPF_ENC_OBS_BF16 1          /* obs elements are bf16 */

/* w_dev / b_dev: HOST arrays of 8 DEVICE pointers (see above).  weights_nhwc != 0: every weight is stored [co][kh][kw][ci]
 * (PyTorch's channels_last memory of a [co][ci][kh][kw] tensor) instead. */
int mapf_encoder_pack(const float *const *w_dev, const float *const *b_dev, int weights_nhwc, uint16_t *packed_dev, float *bias_dev,
                      void *stream);
/* obs [M][6][9][9] (u8 or bf16, 4-byte aligned) -> latent bf16 [M][784] (= Flatten of [16][7][7]). */
int mapf_encoder_forward(const void *obs_dev, int obs_dtype, int64_t M, const uint16_t *packed_dev,
                         const float *bias_dev, uint16_t *latent_dev, void *stream);

/*
 * The same for LISTED rows only: observation i (i < *row_count_dev <= max_rows; the count is read on the device) of obs_dev
 * u8 [*row_count][MAPF_ENC_PACKED_OBS_STRIDE] (486 bytes + 2 of padding: rows start on 4-byte boundaries, what mapf_obs_changed
 * writes) is encoded into row row_index[i] of latent_dev; every other row of latent_dev is left as it is.  For the
 * actor loop: an agent whose observation did not change since the previous step keeps its latent (the encoder is a deterministic
 * per-observation function), and the host never learns how many changed (include/mapf_replay.h: mapf_obs_changed).
 */
#define MAPF_ENC_PACKED_OBS_STRIDE 488
int mapf_encoder_forward_rows(const uint8_t *obs_dev, int64_t max_rows, const int32_t *row_index_dev, const int32_t *row_count_dev,
                              const uint16_t *packed_dev, const float *bias_dev, uint16_t *latent_dev, void *stream);

/* Training forward: the same kernel, additionally storing what the backward pass needs:
 *   acts_dev f16 [7][M][7][7][128] (NHWC) the 7 post-ReLU layer outputs, in order conv0, res1.block1, res1,
 *            res2.block1, res2, res3.block1, res3 -- the inputs of the weight gradients (16-byte aligned);
 *   relu_bits_dev uint32 [7][M][49][4] their sign bits -- the ReLU masks of the backward-data chain (16 bytes per
 *            position instead of 256): channel c = 32 w + 16 a + 4 h + r (w < 4, a < 2, h < 4, r < 4) of a position is > 0
 *            iff bit 8 h + 4 a + r of its word w is set, i.e. byte h of the word belongs to the lane group that holds
 *            those channels in both kernels' MFMA accumulators (each lane stores / loads one byte, no cross-lane traffic). */
int mapf_encoder_forward_save(const void *obs_dev, int obs_dtype, int64_t M, const uint16_t *packed_dev,
                              const float *bias_dev, uint16_t *latent_dev, uint16_t *acts_dev,
                              uint32_t *relu_bits_dev, void *stream);

/*
 * Backward-data chain of the encoder in one kernel (the mirror image of the forward: the transposed convolutions are
 * the same LDS-resident implicit GEMMs on weights packed by mapf_encoder_pack_bwd -- channels swapped, taps
 * flipped; MAPF_ENC_PACKED_BWD_ELEMS f16).
 *   gz7_dev  f16 [M][7][7][16]    gradient w.r.t. the 1x1 convolution's pre-activation (ReLU mask already applied), in whatever
 *            units the caller chose (f16 has 5 exponent bits: scale small gradients up); every output is in the same units
 *   relu_bits_dev uint32 [7][M][49][4] the ReLU sign bits written by mapf_encoder_forward_save
 *   gz_dev   f16 [7][M][7][7][128] OUT: gradient w.r.t. each 128-channel layer's pre-activation (ReLU-masked), same
 *            layer order; conv_k's weight gradient is the correlation of gz_dev[k] with the layer's input
 *            (acts_dev[k-1] of the forward, or the observation for k = 0).
 *   gbias_partial_dev f32 [7][ceil(M / MAPF_ENC_OBS_PER_BLOCK)][128] OUT: per-workgroup sums of gz over its
 *            positions; the bias gradient of layer k is the sum of slab k over its middle axis (no atomics).
 */
#define MAPF_ENC_PACKED_BWD_ELEMS 888832 /* 6*147456 + 4096 */
int mapf_encoder_pack_bwd(const float *const *w_dev, int weights_nhwc, uint16_t *packed_bwd_dev, void *stream);
int mapf_encoder_backward_data(const uint16_t *gz7_dev, int64_t M, const uint32_t *relu_bits_dev,
                               const uint16_t *packed_bwd_dev, uint16_t *gz_dev, float *gbias_partial_dev,
                               void *stream);
/*
 * Weight gradient of conv0 (6 -> 128, 3x3 valid on the 9x9 observation; csrc/mapf_wgrad0.hip):
 *   partial_dev f32 [MAPF_ENC_WGRAD0_PARTS][128][64]: per-partition partial sums of
 *   dW0[co][j = ci*9 + ky*3 + kx] = sum_{m,y,x} gz0[m][y][x][co] * obs[m][ci][y+ky][x+kx]   (columns 54..63 are zero);
 *   gz0_dev f16 [M][49][128] = layer 0 of mapf_encoder_backward's gz; obs_dev / obs_dtype as in mapf_encoder_forward;
 *   grad_scale_dev = mapf_encoder_backward's (the partial sums are multiplied by grad_scale_dev[1]), or NULL: gz0 as it is.
 * The caller adds the partitions (deterministic) and keeps the first 54 columns = the weight's own [co][ci][ky][kx] order.
 */
#define MAPF_ENC_WGRAD0_PARTS 512
int mapf_encoder_wgrad0(const uint16_t *gz0_dev, const void *obs_dev, int obs_dtype, int64_t M, const uint32_t *grad_scale_dev,
                        float *partial_dev, void *stream);

/*
 * The same chain, starting one step earlier: `g_latent_dev` is the gradient w.r.t. the encoder's OUTPUT (bf16 [M][784],
 * the forward's flattened NCHW order: channel * 49 + position) and `latent_dev` that output; the kernel applies the 1x1
 * layer's ReLU mask itself while staging (three elementwise passes and a reduction less for the caller) and also writes
 * the masked gradient position-major, gz7_dev f16 [M][49][16] (operand of the caller's 1x1 weight-gradient GEMM), and
 * gb7_partial_dev f32 [4 * ceil(M/4)][16], per-wave partial bias gradients of the 1x1 layer (the caller adds the rows).
 *
 * Loss scale.  The chain's gradients are f16, so it works on S * gradient with S the power of two that puts max |g_latent| into
 * (8, 16] (a reduction over g_latent launched by this call; the reference's GradScaler, worker.py:283,316-323, does the same job
 * with a dynamic S and skipped steps).  grad_scale_dev, uint32 [2], OUT: [0] = max |g_latent| as bf16 bits, [1] = the bits of the
 * float 1 / S.  gz_dev and gz7_dev are S * gradient; the bias partials are already divided by S; the weight-gradient kernels take
 * grad_scale_dev and divide theirs; a caller that forms a weight gradient itself (the 1x1 layer's GEMM) multiplies by [1].
 */
int mapf_encoder_backward(const uint16_t *g_latent_dev, const uint16_t *latent_dev, int64_t M, const uint32_t *relu_bits_dev,
                          const uint16_t *packed_bwd_dev, uint16_t *gz_dev, float *gbias_partial_dev, uint16_t *gz7_dev,
                          float *gb7_partial_dev, uint32_t *grad_scale_dev, void *stream);

/*
 * Weight gradient of one 3x3 128->128 convolution of the encoder (csrc/mapf_wgrad.hip):
 *   dW[co][ky][kx][ci] = sum_m sum_(y,x) gz[m][y][x][co] * in[m][y+ky-1][x+kx-1][ci]
 * gz_dev = the layer's slice of mapf_encoder_backward_data's output, in_dev = the layer's input (the matching slice
 * of mapf_encoder_forward_save's acts), both f16 [M][7][7][128]; grad_scale_dev = mapf_encoder_backward's (the partial sums are
 * multiplied by grad_scale_dev[1]) or NULL (gz as it is).  The kernel writes MAPF_ENC_WGRAD_PARTS partial
 * sums, fp32 [MAPF_ENC_WGRAD_PARTS][128][3][3][128] (co, ky, kx, ci -- the channels_last order of a [co][ci][3][3]
 * weight); the weight gradient is their sum over the first axis (deterministic, no atomics).
 */
#define MAPF_ENC_WGRAD_PARTS 128
#define MAPF_ENC_WGRAD_SLABS 2 /* internal: input-channel halves of the output, one workgroup each per partition */
int mapf_encoder_wgrad(const uint16_t *gz_dev, const uint16_t *in_dev, int64_t M, const uint32_t *grad_scale_dev, float *partial_dev,
                       void *stream);
/*
 * The same for `layers` (<= 8) layers in ONE launch -- the encoder's six 3x3 layers of an update (reference model.py:30-42 x 3 ResBlocks,
 * worker.py:316): layer l reads gz_dev + l * gz_layer_stride and in_dev + l * in_layer_stride (elements; the slices of
 * mapf_encoder_backward's gz and mapf_encoder_forward_save's acts lie that far apart), the observations of every layer are split into
 * `parts` (<= MAPF_ENC_WGRAD_PARTS; 128 / layers keeps the launch at one workgroup per CU) partitions, and partial_dev is
 * fp32 [layers][parts][128][3][3][128]: layer l's weight gradient is the sum of its `parts` slabs (mapf_sum_parts takes the six sums in
 * one launch).  Same arithmetic per partition as mapf_encoder_wgrad; the partition boundaries -- and with them the fp32 summation
 * order -- follow `parts`.  valid_rows_dev: see the `_bounded` entry points below (NULL = all M observations).
 */
int mapf_encoder_wgrad_multi(const uint16_t *gz_dev, int64_t gz_layer_stride, const uint16_t *in_dev, int64_t in_layer_stride, int layers,
                             int parts, int64_t M, const int32_t *valid_rows_dev, const uint32_t *grad_scale_dev, float *partial_dev,
                             void *stream);
/*
 * Row counts from DEVICE memory (`_bounded` entry points; `valid_rows_dev` of mapf_encoder_wgrad_multi may be NULL = M).  The learner's
 * graph-replayed update (reference worker.py:287-338 replayed from captured launches) runs its kernels on bucket-sized buffers -- the
 * number of distinct observations of a batch rounded up to 1,024 -- and knows the true count only on the device: these variants take it
 * from `valid_rows_dev` (int32 [1], clamped to 0..M) and compute nothing for the rows behind it.  M stays the ALLOCATED row count (the
 * launch grid, the stride between the saved layers).  What the skipped rows' owners leave behind where other kernels read all M rows
 * or all workgroups' partials: zeros in `latent_dev`, in layer 6 of `acts_dev`, in `gz7_dev`, in the bias partials; everything else of
 * those rows (layers 0-5 of acts / gz, the ReLU words) is left untouched and must only be read through the same bound
 * (mapf_encoder_wgrad_multi / mapf_encoder_wgrad0_bounded partition the valid rows only).  `latent_dev` must be 16-byte aligned here.
 */
int mapf_encoder_forward_bounded(const void *obs_dev, int obs_dtype, int64_t M, const int32_t *valid_rows_dev, const uint16_t *packed_dev,
                                 const float *bias_dev, uint16_t *latent_dev, void *stream);
int mapf_encoder_forward_save_bounded(const void *obs_dev, int obs_dtype, int64_t M, const int32_t *valid_rows_dev, const uint16_t *packed_dev,
                                      const float *bias_dev, uint16_t *latent_dev, uint16_t *acts_dev, uint32_t *relu_bits_dev, void *stream);
int mapf_encoder_backward_bounded(const uint16_t *g_latent_dev, const uint16_t *latent_dev, int64_t M, const int32_t *valid_rows_dev,
                                  const uint32_t *relu_bits_dev, const uint16_t *packed_bwd_dev, uint16_t *gz_dev, float *gbias_partial_dev,
                                  uint16_t *gz7_dev, float *gb7_partial_dev, uint32_t *grad_scale_dev, void *stream);
int mapf_encoder_wgrad0_bounded(const uint16_t *gz0_dev, const void *obs_dev, int obs_dtype, int64_t M, const int32_t *valid_rows_dev,
                                const uint32_t *grad_scale_dev, float *partial_dev, void *stream);

/*
 * Inference recurrence behind the encoder (csrc/mapf_recur.hip): for T steps and E environments of N <= 48 agents
 *   hidden = GRUCell(latent_t, hidden)                                   (reference model.py:186-189 / :244)
 *   2 x: info = MultiHeadAttention(hidden, comm_mask_t); hidden = where(partners > 1, GRUCell(info, hidden), hidden)
 *                                                                        (CommBlock, model.py:99-135)
 * with a workgroup keeping an environment's hidden states in LDS across all steps (one workgroup per CU walks its share of the
 * environments; with more environments than CUs and N <= 32, two environments side by side per workgroup).  No autograd.
 *   gi_dev      bf16 [T][E][N][768]  the GRU's input projection W_ih latent (no bias), one GEMM done by the caller
 *   h0_dev      bf16 [E][N][256] or NULL (zeros: episode start)
 *   comm_dev    u8   [T][E][N][N]    communication masks (non-zero = j talks to i)
 *   weights_dev bf16 MAPF_RECUR_WEIGHT_ELEMS: recurrent.weight_hh [768][256] | W_Q;W_K;W_V [384][256] | W_O [64][128] |
 *               update_cell.weight_ih [768][64] | update_cell.weight_hh [768][256], each matrix [O][K] stored in MFMA
 *               A-fragment order [O/16][K/32][lane = 64][8]: element (o, k) at tile o/16, k-step k/32,
 *               lane 16*((k%32)/8) + o%16, slot k%8
 *   bias_dev    f32  MAPF_RECUR_BIAS_ELEMS: recurrent.bias_ih | recurrent.bias_hh | b_Q;b_K;b_V | update_cell.bias_ih |
 *               update_cell.bias_hh
 *   h_out_dev   bf16 [E][N][256] hidden state after the last step;  agent0_out_dev bf16 [T][E][256] or NULL: agent 0's
 *               state after every step (what `bootstrap` feeds the Q head, model.py:248).
 *
 * Compact rows (row_index_dev != NULL; N <= 48 only, else MAPF_ERR_UNSUPPORTED): gi_dev holds num_rows rows [num_rows][768] and
 * row_index_dev int32 [T][E][N] names the row of every (step, environment, agent) entry that has one, -1 for the others (an agent
 * without a row gets no input projection); this is mapf_plan_rows' gidx -- only the entries of a training window that can reach
 * agent 0's Q-value are projected at all.  row_index_dev == NULL: gi_dev is dense [T][E][N][768].
 */
#define MAPF_RECUR_BSUM_ELEMS 2432
#define MAPF_RECUR_WEIGHT_ELEMS 548864 /* 196608 + 98304 + 8192 + 49152 + 196608 */
#define MAPF_RECUR_BIAS_ELEMS 3456     /* 768 + 768 + 384 + 768 + 768 */
int mapf_recurrent_infer(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev,
                         const uint16_t *weights_dev, const float *bias_dev, int T, int E, int N, uint16_t *h_out_dev,
                         uint16_t *agent0_out_dev, const int32_t *row_index_dev, int64_t num_rows, void *stream);
/*
 * One step (T = 1) of E environments of DIFFERENT agent counts (each <= 16) in ONE launch -- the policy recurrence of all active
 * curriculum levels of an actor iteration (the reference draws a (num_agents, map) level per episode, environment.py:148-151, and
 * runs model.step per environment, worker.py:378).  The environments' agent rows lie back to back: gi bf16 [rows][768], h0 / h_out bf16
 * [rows][256] (h0 may be NULL = zero state), comm_dev u8: environment e's [N_e][N_e] mask at byte offset envtab[e].z.
 * envtab_dev int32 [E][4] (16-byte aligned): {N_e, first row of environment e, byte offset of its mask, n_e}.  n_e = 0 or N_e: one
 * environment.  0 < n_e < N_e: entry e stands for N_e / n_e consecutive environments of n_e agents each (rows and masks back to back)
 * stepped by ONE workgroup under the block-diagonal mask -- a step streams the 1.1 MB of weights once per workgroup, so few-agent
 * environments are packed up to the 16 rows of an agent tile (same results up to the summation order of the softmax).
 */
int mapf_recurrent_infer_multi(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev, const uint16_t *weights_dev,
                               const float *bias_dev, int E, const int32_t *envtab_dev, uint16_t *h_out_dev, void *stream);

/*
 * Training pair of mapf_recurrent_infer (csrc/mapf_recur.hip, csrc/mapf_recur_bwd.hip): the same forward, additionally
 * storing what backward-through-time needs, and the backward kernel (one workgroup per environment, steps in reverse).
 *   save_dev[8] (all bf16, R = T*E*N rows, row = (t*E + e)*N + agent):
 *     0 hin0 [R][256] state entering the step | 1 g1 [R][4][256] r, z, n, W_hn h + b_hn of the recurrent cell |
 *     2 hr [2][R][256] state entering round 0/1 | 3 qkv [2][R][384] | 4 ctx [2][R][128] | 5 info [2][R][64] |
 *     6 g2 [2][R][4][256] gates of the update cell | 7 P [2][T*E][2][48][64] attention weights (zero padded)
 *   mapf_recurrent_backward: d_agent0_dev bf16 [T][E][256] = gradient w.r.t. agent 0's state after every step;
 *     weights_t_dev = the TRANSPOSED matrices in fragment order: per gate g (r, z, n) U_ih[g]^T [64][256] | per gate
 *     U_hh[g]^T [256][256] | per gate W_hh[g]^T [256][256] | W_O^T [128][64] | W_qkv^T [256][384]
 *     (MAPF_RECUR_WEIGHT_ELEMS bf16);
 *   out_dev[7]: (bf16) 0 d_gi1 [R][768] gradient w.r.t. the GRU input projection gi | 1 d_gh1 [R][768] |
 *     2 d_gi2 [2][R][768] | 3 d_gh2 [2][R][768] | 4 d_info [2][R][64] | 5 d_qkv [2][R][384]
 *   -- the gradients of the pre-activations of every linear map; weight gradient = (that)^T (its saved input), formed by
 *   the caller (tall GEMMs, once per update) --, and
 *     (f32) 6 bsum [E][MAPF_RECUR_BSUM_ELEMS]: per-environment column sums over steps and agents of
 *     [update cell: dr | dz | dn | dn r] (both rounds) [recurrent cell: dr | dz | dn | dn r] [d_qkv: 384]; summed over E
 *     they are the bias gradients (b_ih: dr, dz, dn; b_hh: dr, dz, dn r).
 * Compact rows (row_index_dev != NULL, N <= 48; see mapf_recurrent_infer): R = num_rows instead of T*E*N in every saved tensor and
 * every output above, row = row_index[t][e][agent]; nothing is saved / written for entries without a row (their gradient is zero;
 * d_agent0_dev must be zero at the steps where agent 0 has no row, i.e. behind the window's last step).
 */
int mapf_recurrent_forward_save(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev,
                                const uint16_t *weights_dev, const float *bias_dev, int T, int E, int N,
                                uint16_t *h_out_dev, uint16_t *agent0_out_dev, uint16_t *const *save_dev,
                                const int32_t *row_index_dev, int64_t num_rows, void *stream);
int mapf_recurrent_backward(const uint16_t *const *saved_dev, const uint8_t *comm_dev, const uint16_t *d_agent0_dev,
                            const uint16_t *weights_t_dev, int T, int E, int N, void *const *out_dev,
                            const int32_t *row_index_dev, int64_t num_rows, void *stream);
/*
 * The same three entry points over TILES of several training windows (round 5): with agent0_stride = S > 0 (N <= 48, else
 * MAPF_ERR_UNSUPPORTED) an "environment" of N agent rows stands for ceil(N / S) windows of S rows each, lying back to back -- rows
 * k S .. k S + S - 1 are window k's agents, its agent 0 in row k S -- whose masks in comm_dev [T][E][N][N] are block-diagonal (no entry
 * between two windows; mapf_plan_rows writes them so for compact widths 4 and 8).  agent0_out_dev / d_agent0_dev are then
 * [T][E][ceil(N / S)][256]: one agent-0 row per window.  Everything else as above, row by row the same bits as one window per
 * environment: a workgroup streams the 1.1 MB of weights once per step whatever the rows of its tile hold, so a batch of 192 windows
 * of <= 8 agents becomes 96 workgroups -- and the online and the target network's recurrences of an update, 192 workgroups each on 256
 * CUs before, run side by side (reference: the same Network.forward per window, model.py:220-249).  agent0_stride = 0: the plain
 * entry points above.  bsum: one row per tile ([E][MAPF_RECUR_BSUM_ELEMS] with the E passed here).
 */
int mapf_recurrent_infer_packed(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev, const uint16_t *weights_dev,
                                const float *bias_dev, int T, int E, int N, uint16_t *h_out_dev, uint16_t *agent0_out_dev,
                                const int32_t *row_index_dev, int64_t num_rows, int agent0_stride, void *stream);
int mapf_recurrent_forward_save_packed(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev, const uint16_t *weights_dev,
                                       const float *bias_dev, int T, int E, int N, uint16_t *h_out_dev, uint16_t *agent0_out_dev,
                                       uint16_t *const *save_dev, const int32_t *row_index_dev, int64_t num_rows, int agent0_stride,
                                       void *stream);
int mapf_recurrent_backward_packed(const uint16_t *const *saved_dev, const uint8_t *comm_dev, const uint16_t *d_agent0_dev,
                                   const uint16_t *weights_t_dev, int T, int E, int N, void *const *out_dev, const int32_t *row_index_dev,
                                   int64_t num_rows, int agent0_stride, void *stream);

/*
 * Input projection of the recurrent cell for the ACTOR's step (reference model.py:191: the W_ih x half of `self.recurrent(latent, hidden)`;
 * mapf_recurrent_infer adds the bias): gi[row] = W_ih latent[row] for the rows of a LIST with a device-side count -- the rows whose
 * observation changed since the previous step (include/mapf_replay.h: mapf_obs_changed): an agent that keeps its latent keeps its gi
 * row -- or, row_list_dev == row_count_dev == NULL, for all num_rows rows.  csrc/mapf_inproj.hip.
 *   mapf_input_proj_pack: W_ih f32 [768][784] -> packed_dev bf16 [MAPF_INPROJ_PACKED_ELEMS] (MFMA fragment order, K padded to 800);
 *   mapf_input_proj_rows: latent_dev bf16 [num_rows][784], gi_dev bf16 [num_rows][768] (rows not listed are left as they are);
 *     with a list, min(*row_count_dev, num_rows) entries of row_list_dev are read.
 */
#define MAPF_INPROJ_PACKED_ELEMS 614400
int mapf_input_proj_pack(const float *w_ih_dev, uint16_t *packed_dev, void *stream);
int mapf_input_proj_rows(const uint16_t *latent_dev, int64_t num_rows, const int32_t *row_list_dev, const int32_t *row_count_dev,
                         const uint16_t *packed_dev, uint16_t *gi_dev, void *stream);

/*
 * The learner's remaining dense products and reductions (csrc/mapf_gemm.hip; reference worker.py:312-324: what `loss.backward()`
 * derives for the Linear / GRUCell weights, rounds 1-4 through hipBLASLt and torch reductions).
 *
 *   mapf_tall_tn: out[m][n] (f32, row-major, ld n) = [accumulate ? out : 0] + scale * sum_k a[k][m] b[k][n]; a [K][lda], b [K][ldb]
 *     16-bit (bf16, or f16 when f16 != 0), m, n, lda, ldb multiples of 8, 16-byte aligned; scale = the float at scale_dev[1]
 *     (the encoder chain's 1 / loss scale, include above) or 1 when NULL.  K is split into mapf_tall_tn_plan's `parts` partitions
 *     whose partial slabs go through ws_dev (>= mapf_tall_tn_plan's ws_elems floats) and are summed in partition order by a second,
 *     small launch: deterministic.  (An agent-scope fence + last-arriver sum inside the one launch cost 100 us per product.)
 *   mapf_sum_parts: out_dev[g][i] = scale * sum_p parts_dev[g][p * n + i] for `groups` <= 8 pointer pairs (HOST arrays of device
 *     pointers), n a multiple of 4: the per-partition slabs of mapf_encoder_wgrad for up to six layers in one launch.
 *   mapf_encoder_small_grads: the encoder's bias gradients from mapf_encoder_backward's per-workgroup partials (gb_part [7][nblk][128]
 *     -> bias7_out [7][128]; gb7_part [rows7][16] -> bias1_out [16]) and conv0's weight gradient from mapf_encoder_wgrad0's slabs
 *     (ws0 [parts0][128][64], column ci*9 + ky*3 + kx -> w0_out [128][3][3][6], the weight's channels_last memory); scratch_dev:
 *     MAPF_SMALL_GRADS_SCRATCH_ELEMS floats between the two stages of the bias sums.
 *   mapf_latent_grad_pack / _rows: g_lat[row] = W_ih^T d_gi[row] (bf16 [num_rows][768] -> bf16 [num_rows][784]): the gradient of the
 *     GRU's input projection w.r.t. the encoder's latents; W_ih f32 [768][784] packed once per weight change into
 *     MAPF_LATGRAD_PACKED_ELEMS bf16 (fragments of W_ih^T).  The forward projection is mapf_input_proj_rows without a list.
 */
#define MAPF_TALL_TN_MAX_PARTS 256
#define MAPF_LATGRAD_PACKED_ELEMS 602112
int mapf_tall_tn_plan(int64_t K, int m, int n, int *slabs_out, int *parts_out, int64_t *ws_elems_out);
int mapf_tall_tn(const uint16_t *a_dev, int64_t lda, const uint16_t *b_dev, int64_t ldb, int64_t K, int m, int n, int f16, float *out_dev,
                 const uint32_t *scale_dev, int accumulate, float *ws_dev, int64_t ws_elems, void *stream);
int mapf_sum_parts(const float *const *parts_dev, float *const *out_dev, int groups, int P, int64_t n, const uint32_t *scale_dev, void *stream);
#define MAPF_SMALL_GRADS_SCRATCH_ELEMS 65536 /* 8 x 64 x 128 floats */
int mapf_encoder_small_grads(const float *gb_part_dev, int64_t nblk, float *bias7_out_dev, const float *gb7_part_dev, int64_t rows7,
                             float *bias1_out_dev, const float *ws0_dev, int parts0, float *w0_out_dev, float *scratch_dev, void *stream);
int mapf_latent_grad_pack(const float *w_ih_dev, uint16_t *packed_dev, void *stream);
int mapf_latent_grad_rows(const uint16_t *d_gi_dev, int64_t num_rows, const uint16_t *packed_dev, uint16_t *g_lat_dev, void *stream);

/*
 * Communication mask of `Network.step` (reference model.py:195-208): mask[e][i][j] = j lies inside i's FOV square
 * (|drow| <= r and |dcol| <= r) AND j is among i's `max_comm` nearest agents by Euclidean distance, i itself
 * included; distance ties go to the LOWEST agent index (the reference's CPU topk leaves ties unspecified).
 * pos int16 [E][N][2]; mask_dev u8 [E][N][N] and/or packed_dev int32 [E][N][cw] (bit j of word j/32, the
 * replay's comm-row format, include/mapf_replay.h); either output may be NULL.  N <= 128, max_comm <= 8.
 */
int mapf_comm_mask(const int16_t *pos_dev, int E, int N, int obs_radius, int max_comm, uint8_t *mask_dev,
                   int32_t *packed_dev, int cw, void *stream);
/* The same for E environments of different agent counts whose agents' positions lie back to back (pos_dev int16 [rows][2]):
 * envtab_dev int32 [E][4] = {agents, first agent row, byte offset of the environment's [N][N] mask in mask_dev, unused}; packed rows go
 * to packed_dev [rows][cw]. */
int mapf_comm_mask_multi(const int16_t *pos_dev, int E, const int32_t *envtab_dev, int obs_radius, int max_comm, uint8_t *mask_dev,
                         int32_t *packed_dev, int cw, void *stream);
/*
 * Dueling Q head of the policy's forward + arg-max (reference model.py:216-220: adv = adv(h), state = state(h),
 * q = state + adv - adv.mean(-1); actions = argmax(q)): hidden bf16 [rows][256], the head's fp32 parameters (adv.weight [5][256],
 * adv.bias [5], state.weight [1][256], state.bias [1]) -> q f32 [rows][5], action int64 [rows] (optional; first maximum).
 */
int mapf_q_head(const uint16_t *hidden_dev, int64_t rows, const float *adv_weight_dev, const float *adv_bias_dev,
                const float *state_weight_dev, const float *state_bias_dev, float *q_dev, int64_t *action_dev, void *stream);

/*
 * Which (step, window, agent) entries of a training batch can influence `Network.bootstrap`'s output at all.  The reference
 * learns from agent 0's hidden state at step steps[b] - 1 only (model.py:248,255), and within a step an agent's state depends on
 * another's only through the communication mask (two attention rounds, model.py:116-130: agent i reads the agents j with
 * comm[b][t][i][j]); encoder, GRU cells and Q head are per agent.  rel = the backward closure of {agent 0 at step steps[b] - 1}:
 * per step two hops along the mask, carried to the step before through the agent's own recurrent state.  Entries outside it
 * contribute nothing to q and receive an exactly zero gradient, so the caller need not encode their observations.
 * comm_dev u8 [B][T][N][N] (non-zero = allowed); steps_dev int64 [B], 1-based; rel_dev u8 [T][B][N] (time-major).  N <= 128.
 */
int mapf_window_relevance(const uint8_t *comm_dev, const int64_t *steps_dev, int T, int B, int N, uint8_t *rel_dev, void *stream);

/*
 * ---- the glue of one batch update (reference worker.py:287-338) as a handful of launches (csrc/mapf_update.hip) ----
 *
 * mapf_plan_mark: mapf_window_relevance's closure plus a renumbering of every window's agents such that the agents needed at step t
 * are a PREFIX of the order (the set only shrinks going forward in time; agent 0, needed until the window's last step, stays agent 0):
 *   comm: window b, step t at comm_dev + b * stride_b + t * stride_t (bytes = elements), [N][N] contiguous; steps int64 [B] 1-based
 *   (+ extra_steps f32 [B] if given: the target window ends `steps` later, worker.py:296); mark_all != 0: no pruning -- every agent
 *   at every step up to the window's last one (what the reference encodes);
 *   rel_dev u8 [T][B][N] optional; slot int16 [B][N] position of agent j (-1: never needed); order int16 [B][N] agent at position i;
 *   nact int32 [T][B] agents needed at step t; cnt int32 [B] = sum_t nact (observations of the window to encode); nag int32 [B] = nact[0].
 * mapf_plan_rows: everything the encoder / recurrence launches need in the compact numbering, Nc (multiple of 16, >= max nag) agent
 * positions per window; rows are numbered window by window, step by step, position by position:
 *   gidx int32 [T][B][Nc] row of (t, b, position) or -1; comm_c u8 [T][B][Nc][Nc] (positions >= nact[t][b] read only themselves);
 *   h0_c bf16 [B][Nc][256] from hidden f16 (or bf16) [B*N][256]; row_src int64 [num_rows] element offset of every row's observation
 *   in the bf16 observations (window b, step t at obs + b * obs_stride_b + t * obs_stride_t elements, [N][486] contiguous) and
 *   obs_rows bf16 [num_rows][486] the rows themselves (both optional; num_rows = sum of cnt, known to the caller).
 * Nc = 4 or 8 (every nag <= Nc, B a multiple of K = 16 / Nc): K consecutive windows share a 16-row tile of the recurrence kernels
 * (mapf_recurrent_*_packed with agent0_stride = Nc and E = B / K): gidx and h0_c as above -- [T][B][Nc] IS [T][B / K][16] --, and
 * comm_c u8 [T][B / K][16][16] block-diagonal, window b in rows and columns (b % K) Nc .. of tile b / K.
 */
#define MAPF_PLAN_MAX_STEPS 20
int mapf_plan_mark(const uint8_t *comm_dev, int64_t stride_b, int64_t stride_t, const int64_t *steps_dev, const float *extra_steps_dev,
                   int T, int B, int N, int mark_all, uint8_t *rel_dev, int16_t *slot_dev, int16_t *order_dev, int32_t *nact_dev, int32_t *cnt_dev, int32_t *nag_dev,
                   int32_t *ucnt_dev, void *stream);
/* totals_dev[k] = sum over the B windows of counts_dev[k][b] for k < rows: the batch totals of mapf_plan_mark / mapf_obs_dup's per-window
 * counts (entries online, agents online, entries target, agents target, distinct online, distinct target) -- what the `_bounded` encoder
 * entry points read as their row count.  counts_dev int32 [rows][B], totals_dev int32 [rows]; rows <= 16. */
int mapf_plan_totals(const int32_t *counts_dev, int rows, int B, int32_t *totals_dev, void *stream);
int mapf_plan_rows(int T, int B, int N, int Nc, const int16_t *order_dev, const int32_t *nact_dev, const int32_t *cnt_dev,
                   const int32_t *nag_dev, const uint8_t *comm_dev, int64_t comm_stride_b, int64_t comm_stride_t,
                   const uint16_t *hidden_dev, int hidden_is_bf16, const uint16_t *obs_bf16_dev, int64_t obs_stride_b, int64_t obs_stride_t,
                   int32_t *gidx_dev, uint8_t *comm_c_dev, uint16_t *h0_c_dev, int64_t num_rows, int64_t *row_src_dev,
                   uint16_t *obs_rows_dev, const uint8_t *dup_dev, const int32_t *ucnt_dev, int32_t *umap_dev, int32_t *row_tbp_dev,
                   void *stream);
/* The same for bucket-sized buffers (the graph-replayed update: launch sizes rounded up, true counts on the device only): first
 * row_src_dev[0 .. fill_urows) = 0 (a padding row gathers observation 0 of the batch), umap_dev[0 .. fill_rows) = 0 and
 * row_tbp_dev[0 .. fill_rows) = -1 (a padding entry uses distinct row 0 and is skipped by mapf_dedup_sum), inside the same launches. */
int mapf_plan_rows_padded(int T, int B, int N, int Nc, const int16_t *order_dev, const int32_t *nact_dev, const int32_t *cnt_dev,
                          const int32_t *nag_dev, const uint8_t *comm_dev, int64_t comm_stride_b, int64_t comm_stride_t,
                          const uint16_t *hidden_dev, int hidden_is_bf16, const uint16_t *obs_bf16_dev, int64_t obs_stride_b, int64_t obs_stride_t,
                          int32_t *gidx_dev, uint8_t *comm_c_dev, uint16_t *h0_c_dev, int64_t num_rows, int64_t *row_src_dev,
                          uint16_t *obs_rows_dev, const uint8_t *dup_dev, const int32_t *ucnt_dev, int32_t *umap_dev, int32_t *row_tbp_dev,
                          int64_t fill_rows, int64_t fill_urows, void *stream);
/*
 * Repeated observations (exact reuse: the encoder is a deterministic per-observation function).  mapf_obs_dup: dup u8 [T][B][N] = the
 * FIRST step of the window at which the same agent carried the same 486 values (<= t; t itself for a new observation and for entries
 * outside the target window's closure; compared value by value, a hash only preselects); ucnt_online / ucnt_target
 * int32 [B] (zeroed by mapf_plan_mark when given there) += the DISTINCT observations of either closure per window.  With dup_dev,
 * mapf_plan_rows numbers the distinct observations like the rows (duplicates skipped): umap int32 [rows] = the distinct row an entry
 * uses, row_src / obs_rows then hold the num_rows = sum(ucnt) DISTINCT observations only; row_tbp int32 [rows] (optional) =
 * (t << 24) | (position << 16) | window of every row.  mapf_dedup_sum: d_unique[u] = sum of d_rows[r] over the entries r with
 * umap[r] == u (fp32 sum, fixed order), the gradient of a shared row; entries with row_tbp[r] < 0 are skipped (padding of a
 * bucket-sized launch).
 */
int mapf_obs_dup(int T, int To, int B, int N, const uint16_t *obs_bf16_dev, int64_t obs_stride_b, int64_t obs_stride_t,
                 const int16_t *slot_online_dev, const int16_t *slot_target_dev, const int32_t *nact_online_dev,
                 const int32_t *nact_target_dev, uint8_t *dup_dev, int32_t *ucnt_online_dev, int32_t *ucnt_target_dev, void *stream);
int mapf_dedup_sum(int T, int B, int Nc, int64_t rows, int row_bytes, const int32_t *gidx_dev, const int32_t *umap_dev,
                   const int32_t *row_tbp_dev, const void *d_rows_dev, void *d_unique_dev, void *stream);
/* dense[r][:] = idx[r] >= 0 ? rows[idx[r]][:] : 0 (to_dense != 0)  /  rows[idx[r]][:] = dense[r][:] where idx[r] >= 0 (to_dense == 0);
 * R dense rows of row_bytes (multiple of 16) bytes. */
int mapf_rows_scatter(void *rows_dev, const int32_t *idx_dev, void *dense_dev, int64_t R, int row_bytes, int to_dense, void *stream);

/*
 * Dueling head of both networks, TD error, priorities and Huber loss of a batch, forward AND backward (model.py:259-262,
 * worker.py:296-310,341-344), fp32 from the bf16 states:
 *   a0_online bf16 [To][B][256] / a0_target bf16 [Tt][B][256]: agent 0's state after every step (mapf_recurrent_*'s agent0_out) of
 *   the online window under the online network / the target window under the target network; a0_online_next bf16 [Tt][B][256] or
 *   NULL: the online network on the target window -- double-DQN, it picks the action the target network values (NULL: the
 *   reference's max_a Q_target, worker.py:300-303); bt_steps int64 [B], steps f32 [B], action int64 [B], reward / done / weights f32 [B];
 *   head_online / head_target: HOST arrays of 4 DEVICE pointers {adv.weight [5][256], adv.bias [5], state.weight [256], state.bias [1]} fp32.
 * Outputs: q (of the taken action), q_next, td f32 [B]; prio f64 [B] = max(|td|, 1e-6); loss f32 [1] = mean(w huber(td));
 *   d_a0 bf16 [To][B][256] gradient of the loss w.r.t. a0_online; head_grads: HOST array of 4 DEVICE pointers, ACCUMULATED into.
 *   scratch f32 [9 B].
 */
int mapf_dqn_head_loss(int B, int To, int Tt, const uint16_t *a0_online_dev, const uint16_t *a0_target_dev,
                       const uint16_t *a0_online_next_dev, const int64_t *bt_steps_dev, const float *steps_dev,
                       const int64_t *action_dev, const float *reward_dev, const float *done_dev, const float *weights_dev,
                       const float *const *head_online, const float *const *head_target, float gamma, float *q_dev,
                       float *q_next_dev, float *td_dev, double *prio_dev, float *loss_dev, float *scratch_dev, uint16_t *d_a0_dev,
                       float *const *head_grads, void *stream);

/*
 * fp32 parameters -> the fragment images of the recurrence kernels, one launch each: weights_dev / bias_dev as mapf_recurrent_infer
 * takes them (either both or neither), weights_t_dev as mapf_recurrent_backward takes it (optional).  params_dev: HOST array of 14
 * DEVICE pointers (fp32, row-major contiguous): recurrent.weight_hh, .bias_ih, .bias_hh, W_Q.weight, W_K.weight, W_V.weight, W_Q.bias,
 * W_K.bias, W_V.bias, W_O.weight, update_cell.weight_ih, .weight_hh, .bias_ih, .bias_hh.
 */
int mapf_recurrent_pack(const float *const *params_dev, uint16_t *weights_dev, float *bias_dev, uint16_t *weights_t_dev, void *stream);
/* Bias gradients of the recurrence from mapf_recurrent_backward's bsum [E][MAPF_RECUR_BSUM_ELEMS], ACCUMULATED into grads_dev: HOST
 * array of 7 DEVICE pointers {recurrent.bias_ih, .bias_hh, W_Q.bias, W_K.bias, W_V.bias, update_cell.bias_ih, .bias_hh}. */
int mapf_recurrent_bias_grads(const float *bsum_dev, int E, float *const *grads_dev, void *stream);

/*
 * torch.nn.utils.clip_grad_norm_(max_norm) + one torch.optim.Adam step (worker.py:260,319-322) over flat fp32 buffers of n elements:
 * the gradients are scaled in place by min(1, max_norm / (||g|| + 1e-6)); norm_out f32 [1] receives ||g|| before clipping;
 * params_bf16_dev (optional) receives the bf16 copy of the new parameters; scratch f32 [256]; step = 1-based count of this step.
 */
int mapf_adam_step(int64_t n, float *params_dev, float *grads_dev, float *exp_avg_dev, float *exp_avg_sq_dev,
                   uint16_t *params_bf16_dev, float *scratch_dev, float *norm_out_dev, float lr, float beta1, float beta2, float eps,
                   int64_t step, float max_norm, void *stream);
/* The same step with the 1-based step count in DEVICE memory (int64 [1], 8-byte aligned): the first of the two launches increments it, the
 * second takes Adam's bias corrections 1 - beta^step from it -- no host scalar changes from call to call, so the launches can be
 * replayed from a captured HIP graph (mapf_rl_amd/update.py). */
int mapf_adam_step_dev(int64_t n, float *params_dev, float *grads_dev, float *exp_avg_dev, float *exp_avg_sq_dev,
                       uint16_t *params_bf16_dev, float *scratch_dev, float *norm_out_dev, float lr, float beta1, float beta2, float eps,
                       int64_t *step_dev, float max_norm, void *stream);
int mapf_to_bf16(const float *src_dev, uint16_t *dst_dev, int64_t n, void *stream);
/* rows [first_row, last_row) of n <= 24 row-major device buffers := 0; bufs_dev: HOST array of DEVICE pointers (16-byte aligned),
 * row_bytes: HOST array of row sizes (multiples of 16). */
int mapf_zero_rows(void *const *bufs_dev, const int *row_bytes, int n, int64_t first_row, int64_t last_row, void *stream);
/* The same with the first row read from device memory (int32 [1], clamped to 0..last_row): the graph-replayed update clears the padding
 * rows of its bucket-sized GEMM operands behind the batch's true entry count (mapf_plan_totals) instead of the whole buffers. n <= 24. */
int mapf_zero_rows_from(void *const *bufs_dev, const int *row_bytes, int n, const int32_t *first_row_dev, int64_t last_row, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MAPF_DQN_H */
