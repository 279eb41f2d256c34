// mapf_recur_internal.h -- shared between csrc/mapf_recur.hip / mapf_recur_bwd.hip (environments of up to 48 agents) and
// csrc/mapf_recur_wide.hip / mapf_recur_wide_bwd.hip (49..128 agents): the saved-tensor / argument structs and the launchers the
// C ABI entry points (include/mapf_dqn.h: mapf_recurrent_infer / _forward_save / _backward) dispatch to.  Not part of the ABI.
#ifndef MAPF_RECUR_INTERNAL_H
#define MAPF_RECUR_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstdint>

#define MAPF_RECUR_NARROW_AGENTS 48  /* one workgroup keeps every image of <= 48 agents in LDS (mapf_recur.hip) */
#define MAPF_RECUR_SMALL_AGENTS 16   /* up to here the same kernels built for ONE agent tile run (mapf_recur_nt1.hip, mapf_recur_bwd_nt1.hip) */
#define MAPF_RECUR_MAX_AGENTS 128    /* widest environment the fused recurrence kernels accept */

// what the training forward stores for the backward kernel; R = T*E*N rows, row = (t*E + e)*N + agent
struct RecurSave {
    uint16_t *hin0;  // [R][256]       state entering the step
    uint16_t *g1;    // [R][4][256]    r, z, n, W_hn h + b_hn of the recurrent cell
    uint16_t *hr;    // [2][R][256]    state entering communication round 0 / 1
    uint16_t *qkv;   // [2][R][384]
    uint16_t *ctx;   // [2][R][128]
    uint16_t *info;  // [2][R][64]
    uint16_t *g2;    // [2][R][4][256] gate terms of the update cell
    uint16_t *P;     // [2][T*E][2][48][64] attention weights (N <= 48 only; wider environments recompute them in the backward)
};

struct RecurBwdArgs {
    // saved by the forward (RecurSave)
    const uint16_t *hin0, *g1, *hr, *qkv, *ctx_unused, *info_unused, *g2, *P;
    const uint8_t *comm;   // [T][E][N][N]
    const uint16_t *dA0;   // [T][E][256] gradient w.r.t. agent 0's state after every step
    const uint16_t *WT;    // transposed weights, fragment order
    // outputs, rows as in the forward's saved tensors
    uint16_t *d_gi1, *d_gh1;  // [R][768]
    uint16_t *d_gi2, *d_gh2;  // [2][R][768]
    uint16_t *d_info;         // [2][R][64]
    uint16_t *d_qkv;          // [2][R][384]
    float *bsum;              // [E][MAPF_RECUR_BSUM_ELEMS] per-environment column sums (bias gradients)
    int T, E, N;
};

// csrc/mapf_recur_nt1.hip / mapf_recur_bwd_nt1.hip (and _nt2): csrc/mapf_recur.hip / mapf_recur_bwd.hip compiled for one agent tile (N <= 16:
// the reference's own training shapes and every curriculum level) and for two (N <= 32).  Same arguments as the C ABI entry points that dispatch to them; the
// attention weights P of the saved state are laid out for 16 (32) agent rows per (step, environment) by these pairs.
int mapf_recurrent_infer_packed_nt1(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev, const uint16_t *weights_dev,
                                    const float *bias_dev, int T, int E, int N, uint16_t *h_out_dev, uint16_t *agent0_out_dev, const int32_t *row_index_dev,
                                    int64_t num_rows, int agent0_stride, void *stream);
int mapf_recurrent_forward_save_packed_nt1(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev, const uint16_t *weights_dev,
                                           const float *bias_dev, int T, int E, int N, uint16_t *h_out_dev, uint16_t *agent0_out_dev,
                                           uint16_t *const *save_dev, const int32_t *row_index_dev, int64_t num_rows, int agent0_stride, void *stream);
int mapf_recurrent_backward_packed_nt1(const uint16_t *const *saved_dev, const uint8_t *comm_dev, const uint16_t *d_agent0_dev,
                                       const uint16_t *weights_t_dev, int T, int E, int N, void *const *out_dev, const int32_t *row_index_dev,
                                       int64_t num_rows, int agent0_stride, void *stream);

int mapf_recurrent_infer_packed_nt2(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev, const uint16_t *weights_dev,
                                    const float *bias_dev, int T, int E, int N, uint16_t *h_out_dev, uint16_t *agent0_out_dev, const int32_t *row_index_dev,
                                    int64_t num_rows, int agent0_stride, void *stream);
int mapf_recurrent_forward_save_packed_nt2(const uint16_t *gi_dev, const uint16_t *h0_dev, const uint8_t *comm_dev, const uint16_t *weights_dev,
                                           const float *bias_dev, int T, int E, int N, uint16_t *h_out_dev, uint16_t *agent0_out_dev,
                                           uint16_t *const *save_dev, const int32_t *row_index_dev, int64_t num_rows, int agent0_stride, void *stream);
int mapf_recurrent_backward_packed_nt2(const uint16_t *const *saved_dev, const uint8_t *comm_dev, const uint16_t *d_agent0_dev,
                                       const uint16_t *weights_t_dev, int T, int E, int N, void *const *out_dev, const int32_t *row_index_dev,
                                       int64_t num_rows, int agent0_stride, void *stream);

// csrc/mapf_recur_wide.hip: forward for 48 < N <= 128 (sv == nullptr: inference, nothing saved).  Returns a MAPF_* status.
int mapf_recur_wide_forward(const uint16_t *gi, const uint16_t *h0, const uint8_t *comm, const uint16_t *W, const float *bias, int T, int E,
                            int N, uint16_t *h_out, uint16_t *agent0_out, const RecurSave *sv, hipStream_t stream);
// csrc/mapf_recur_wide_bwd.hip: backward through time for 48 < N <= 128
int mapf_recur_wide_backward(const RecurBwdArgs &a, hipStream_t stream);

#endif
