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

#ifdef __cplusplus
}
#endif
#endif /* MAPF_DQN_H */
