// mapf_encoder.hip -- the observation encoder of the DQN (reference model.py:147-162: Conv(6->128,3x3,valid)+ReLU,
// 3 x ResBlock(128) (model.py:30-42, two 3x3 pad-1 convolutions + identity skip, no normalisation),
// Conv(128->16,1x1)+ReLU, Flatten) as ONE inference kernel for gfx950 (see include/mapf_dqn.h).
//
// Why one kernel: per observation the encoder is 87.6 MFLOP over activations of only 7x7x128 f16 = 12.5 KB.
// Layer-by-layer (MIOpen implicit GEMM + an epilogue pass per layer) every layer round-trips that activation
// through HBM and pays a launch; here a workgroup keeps the activations of G = 4 observations resident in LDS for
// all 8 layers and only the packed weights (1.8 MB, L2-resident) stream in.  MFMA-bound.
//
// Mapping (256 threads = 4 waves, <= 256 VGPR each; 76 KB LDS -> TWO independent workgroups per CU, 2 waves per
// SIMD: while one workgroup is in a layer epilogue / at a barrier the other one keeps the MFMA pipe busy):
//  * Each 3x3 layer is the GEMM  out[co][p] = sum_{tap,ci} W[co][ci][tap] * act[p + tap][ci]  with M = 128 output
//    channels, N = 49*G positions, K = 9*128, on v_mfma_f32_16x16x32_f16: A = weights, B = activations, so a lane's
//    4 accumulator registers are 4 consecutive output channels of one position -> one 8-byte LDS store.
//  * Wave w owns output channels [32*w, +32) (two 16-row A tiles) and all 13 position tiles (16 positions each,
//    208 >= 196): 104 accumulator registers.
//  * Activations live in LDS as rows of 128 channels (256 B + 16 B pad = 272 B: conflict-free ds_read_b128 over
//    16 consecutive rows).  Row index = 64*obs + 8*(y+1) + (x+1): an 8-wide zero-bordered image in which the right
//    border of one image row IS the left border of the next and the bottom border of one observation IS the top
//    border of the next, so a 3x3 tap is a constant row offset (8*ky + kx) -- an immediate in the ds_read -- and
//    the zero padding costs no selects.  Border rows are zeroed once and never written.
//  * The A operand is pre-packed in exact fragment order (mapf_encoder_pack): one k-step of one 16-channel tile is a
//    contiguous 1 KiB, read straight into registers with global_load_dwordx4 (no LDS round trip; each weight byte
//    is used by exactly one wave of the workgroup).
//  * ResBlock: the skip input of this wave's own (co, position) elements is read from LDS as packed f16 before
//    conv1's output overwrites it and becomes the initial accumulator of conv2 -> a single activation buffer.
//  * conv0 (K = 54, padded to 64) builds its B fragments directly from the raw observation bytes staged in LDS.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "mapf_dqn.h"
#include "mapf_env.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

#ifndef MAPF_ENC_NT_SAVE  // diagnostic builds only (tools/micro/enc_ablate.py): 1 = the saved-activation copies leave as non-temporal stores
#define MAPF_ENC_NT_SAVE 0
#endif
#ifndef MAPF_ENC_ABLATE  // diagnostic builds only (tools/micro/enc_ablate.py): 1 = no saved-activation copies, 2 = no ReLU sign words, 4 = 16 consecutive LDS rows per position tile, 8 = conv0 without its input gather (zeros), 64 = conv0 gathering one tile's window for all tiles, 16 = no LDS zero fill, 32 = no 1x1 head
#define MAPF_ENC_ABLATE 0
#endif
constexpr int G = MAPF_ENC_OBS_PER_BLOCK;  // observations per workgroup
constexpr int ROWB = 272;                  // bytes per activation row (128 f16 + 16 B pad)
constexpr int ACT_ROWS = 64 * G + 9;
constexpr int ACT_BYTES = ACT_ROWS * ROWB;  // 141,712
constexpr int OBS_ELEMS = 6 * 9 * 9;        // 486
constexpr int RAW_BYTES = ((G * OBS_ELEMS * 2 + 15) / 16) * 16;  // sized for 2-byte inputs (>= G * MAPF_ENC_PACKED_OBS_STRIDE bytes too)
constexpr int NT = 13;                      // position tiles (16 positions each)
constexpr int NTHREADS = 256;
constexpr int NPOS = 49 * G;
static_assert(NT * 16 >= NPOS, "tiles must cover all positions");
static_assert(ACT_BYTES % 16 == 0, "");
static_assert(2 * (ACT_BYTES + RAW_BYTES) <= 160 * 1024, "LDS budget: two workgroups per CU");

// packed-weight element offsets (f16 elements), see mapf_dqn.h
constexpr int WP_L0 = 0;
constexpr int WP_L0_SIZE = 2 * 8 * 512;
constexpr int WP_RES = WP_L0 + WP_L0_SIZE;
constexpr int WP_RES_SIZE = 36 * 8 * 512;
constexpr int WP_L7 = WP_RES + 6 * WP_RES_SIZE;
constexpr int WP_L7_SIZE = 4 * 512;
constexpr int WP_TOTAL = WP_L7 + WP_L7_SIZE;
static_assert(WP_TOTAL == MAPF_ENC_PACKED_ELEMS, "header constant out of date");
constexpr int BIAS_TOTAL = 128 * 7 + 16;
static_assert(BIAS_TOTAL == MAPF_ENC_BIAS_ELEMS, "header constant out of date");

__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {  // round to nearest even (finite inputs)
    uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }

// ---- the element type of everything the encoder kernels keep between layers: IEEE half (f16) ----
// Activations (LDS and the saved copies), packed weights and pre-activation gradients are f16, accumulated in fp32 by
// v_mfma_f32_16x16x32_f16 -- the arithmetic the reference trains in (fp16 autocast, worker.py:283,316-323).  Rounds 1-2 kept
// them in bf16: same speed, but 3 fewer mantissa bits on every layer output put the weight gradients 7-10 % off the fp32
// direction (profiles/r03_encoder_grad_error_fp16.txt: f16 activations + weights bring that to 2-3 %, the format of the
// gradients themselves does not matter).  f16's range is the price: conversions clamp to +-65504 (no inf, hence no NaN from
// inf * 0), and the backward chain works on gradients multiplied by a power-of-two loss scale (GradScale below).  The encoder's
// OUTPUT (the latent the projection GEMM reads) and the gradient arriving for it stay bf16 like the rest of the network.
typedef __attribute__((ext_vector_type(8))) _Float16 el8;  // one MFMA A/B fragment
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
constexpr float EL_MAX = 65504.f;
__device__ __forceinline__ f32x4 el_mfma(const el8 &a, const el8 &b, const f32x4 &c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t el_pack2(float lo, float hi) {  // round to nearest even; callers clamp
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2));
}
__device__ __forceinline__ float el_to_f32(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }
__device__ __forceinline__ float el_lo(uint32_t w) { return el_to_f32(w & 0xFFFFu); }
__device__ __forceinline__ float el_hi(uint32_t w) { return el_to_f32(w >> 16); }
__device__ __forceinline__ uint16_t f32_to_el_bits(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)__builtin_amdgcn_fmed3f(f, -EL_MAX, EL_MAX));
}
__device__ __forceinline__ uint16_t raw_to_el(uint8_t v) { return __builtin_bit_cast(uint16_t, (_Float16)(float)v); }  // exact for 0..255
__device__ __forceinline__ uint16_t raw_to_el(uint16_t v) { return f32_to_el_bits(bf16_bits_to_f32(v)); }            // bf16 input

// one 16-byte chunk of a saved activation tensor (written once here, read ~2 ms later by the weight-gradient kernels)
__device__ __forceinline__ void save_store(uint4 *p, const uint4 &v) {
#if MAPF_ENC_NT_SAVE
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4 *>(p));
#else
    *p = v;
#endif
}

// relu(acc + bias) for the 4 consecutive channels of one lane, packed to 4 elements (v_med3_f32 = ReLU and range clamp in one)
__device__ __forceinline__ uint2 pack_relu(const f32x4 &a, const float4 &b) {
    return make_uint2(el_pack2(__builtin_amdgcn_fmed3f(a[0] + b.x, 0.f, EL_MAX), __builtin_amdgcn_fmed3f(a[1] + b.y, 0.f, EL_MAX)),
                      el_pack2(__builtin_amdgcn_fmed3f(a[2] + b.z, 0.f, EL_MAX), __builtin_amdgcn_fmed3f(a[3] + b.w, 0.f, EL_MAX)));
}

// bit r of the result: the r-th of the 4 f16 values in `y` (a ReLU output) is > 0
__device__ __forceinline__ uint32_t positive4(const uint2 y) {
    // y holds max(., 0) results: the sign bit can only belong to a -0, so "> 0" is "magnitude bits non-zero"
    return ((y.x & 0x7FFFu) ? 1u : 0u) | ((y.x & 0x7FFF0000u) ? 2u : 0u) | ((y.y & 0x7FFFu) ? 4u : 0u) | ((y.y & 0x7FFF0000u) ? 8u : 0u);
}

// Training forward: ReLU sign bits of a layer output, one uint32 per (position, 32-channel block cb = wave), of which every
// lane owns one BYTE: byte lh holds the lane's two nibbles (tile a = 0 in bits 0-3, a = 1 in bits 4-7), i.e. channel
// 32 cb + 16 a + 4 lh + r is bit 8 lh + 4 a + r of the word.  The backward kernel has the same lane <-> channel mapping, so a
// lane stores and later loads just its own byte: no cross-lane traffic (the first version ORed the four lanes' nibbles into
// one word with two shuffles per tile -- ds_bpermute, i.e. LDS-pipe instructions in a kernel the LDS pipe co-limits --
// and cost the training forward 1.0 ms of its 9.0).  16 bytes per position instead of the 256-byte activation row.
__device__ __forceinline__ void store_relu_bits(uint32_t nib[NT], uint32_t vmask, uint32_t *__restrict__ dst, int cb, int lr, int lh) {
    if (MAPF_ENC_ABLATE & 2) return;
    uint8_t *d8 = reinterpret_cast<uint8_t *>(dst) + cb * 4 + lh;
#pragma unroll
    for (int n = 0; n < NT; ++n)
        if ((vmask >> n) & 1u) d8[(n * 16 + lr) * 16] = (uint8_t)nib[n];
}

// One 3x3 pad-1 128->128 convolution over the LDS-resident activations: acc[a][n] += W(a) * act(n).
// `wv` points at this lane's element of this wave's first co tile of the layer: index (s*8 + a)*64.
// The 36 k-steps x 13 position tiles are one flat software-pipelined sequence: the B fragment of tile-step t + PB is
// read from LDS while tile-step t's two MFMAs issue, and the A fragments of k-step s + PA are loaded from L2 at the
// start of k-step s.  hipcc on its own emits each load right before its use (s_waitcnt 0 per tile); the
// sched_barrier pins the order written here, and the wait-count pass then inserts counted lgkmcnt / vmcnt.
constexpr int PB = 4, RB = PB + 1;  // B prefetch distance in tile-steps / ring size
constexpr int PA = 2, RA = PA + 1;  // A prefetch distance in k-steps / ring size

__device__ __forceinline__ int tap_off(int s) {
    const int tap = s >> 2, chunk = s & 3;
    return (8 * (tap / 3) + (tap % 3)) * ROWB + chunk * 64;
}

//
// COPY: the layer the convolution READS is also a tensor the training path saves ([M][49][128], valid rows only).  Its LDS ->
// global copy (13 16-byte chunks per thread) rides inside the MFMA stream, one chunk per 36 tile-steps: the LDS read at
// step 4, the store at step 22 of each group.  As a separate pass between the barrier and the convolution the same copy
// cost 3.3 us per layer and workgroup (1.3 of the training forward's 8.5 ms).
template <bool COPY = false>
__device__ __forceinline__ void conv3x3(const unsigned char *act, const el8 *__restrict__ wv, const int (&addr)[NT],
                                        f32x4 (&acc)[2][NT], uint16_t *__restrict__ cdst = nullptr, int ctotal = 0, int tid = 0) {
    el8 ar[RA][2], br[RB];
    uint4 cv = make_uint4(0, 0, 0, 0);
    int cc = 0;
#pragma unroll
    for (int s = 0; s < PA; ++s) {
        ar[s][0] = wv[(s * 8 + 0) * 64];
        ar[s][1] = wv[(s * 8 + 1) * 64];
    }
#pragma unroll
    for (int t = 0; t < PB; ++t) br[t] = *reinterpret_cast<const el8 *>(act + addr[t % NT] + tap_off(t / NT));
#pragma unroll
    for (int s = 0; s < 36; ++s) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int t = s * NT + n, tn = t + PB;
            if (tn < 36 * NT) br[tn % RB] = *reinterpret_cast<const el8 *>(act + addr[tn % NT] + tap_off(tn / NT));
            if (n == 0 && s + PA < 36) {
                ar[(s + PA) % RA][0] = wv[((s + PA) * 8 + 0) * 64];
                ar[(s + PA) % RA][1] = wv[((s + PA) * 8 + 1) * 64];
            }
            // (branch-free: a chunk index past the end is clamped to the last chunk, which is then written twice with the same
            // bytes -- a branch here splits the pinned basic block and the register allocator spills 50-150 registers;
            // laundered: otherwise the 13 chunk addresses, which depend on nothing but tid, are all computed before the loop)
            if (COPY && !(MAPF_ENC_ABLATE & 1) && t % 36 == 4) {
                cc = tid;
                asm volatile("" : "+v"(cc));
                cc = min(cc + NTHREADS * (t / 36), ctotal - 1);
                const int rowi = cc >> 4, ch = cc & 15;
                const int o = rowi / 49, q = rowi - 49 * o, y = q / 7, x = q - 7 * y;
                cv = *reinterpret_cast<const uint4 *>(act + (64 * o + 8 * y + x + 9) * ROWB + ch * 16);
            }
            if (COPY && !(MAPF_ENC_ABLATE & 1) && t % 36 == 22) save_store(reinterpret_cast<uint4 *>(cdst) + cc, cv);
            acc[0][n] = el_mfma(ar[s % RA][0], br[t % RB], acc[0][n]);
            acc[1][n] = el_mfma(ar[s % RA][1], br[t % RB], acc[1][n]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Training forward: copy the layer output that is now resident in LDS (valid rows only) to its saved-activation
// tensor [M][49][128] (NHWC) -- 49*nobs rows of 256 B, contiguous in global memory for the block's observations.
__device__ __forceinline__ void save_rows(const unsigned char *act, uint16_t *__restrict__ dst, int nobs, int tid) {
    if (MAPF_ENC_ABLATE & 1) return;
    const int total = nobs * 49 * 16;  // 16-byte chunks
    for (int c = tid; c < total; c += NTHREADS) {
        const int rowi = c >> 4, ch = c & 15;
        const int o = rowi / 49, q = rowi - 49 * o, y = q / 7, x = q - 7 * y;
        save_store(reinterpret_cast<uint4 *>(dst) + c, *reinterpret_cast<const uint4 *>(act + (64 * o + 8 * y + x + 9) * ROWB + ch * 16));
    }
}

#ifdef MAPF_ENC_CLOCK  // diagnostic build only (libmapf_enc_clock.so, bench.py's encoder_roofline.clock_ghz): per workgroup the shader-clock
__device__ unsigned long long g_enc_clock[8192][2];  // cycles (s_memtime) and the 100 MHz reference ticks (s_memrealtime) it ran for
#endif

template <typename InT, bool SAVE, bool IDX = false>
__global__ void __launch_bounds__(NTHREADS, 2) encoder_fwd_kernel(const InT *__restrict__ obs, long long M,
                                                          const uint16_t *__restrict__ wp, const float *__restrict__ bias,
                                                          uint16_t *__restrict__ out, uint16_t *__restrict__ save,
                                                          uint32_t *__restrict__ relu_bits, const int32_t *__restrict__ row_index,
                                                          const int32_t *__restrict__ row_count) {
    // IDX (inference only): `obs` holds *row_count <= M observations back to back (the count is read from device memory) and the
    // latent of observation i goes to row row_index[i] of `out` -- the actor loop re-encodes only the agents whose observation
    // changed since the previous step (mapf_obs_changed packs them), without the host knowing how many.
    // The grid is a fixed number of workgroups that WALK the list (a grid of max_rows / 4 workgroups that mostly return at once cost
    // 2 ms at 163,840 rows: every one of them is a 76-KB-LDS workgroup to place).
    if constexpr (IDX) M = min(M, (long long)*row_count);
    if constexpr (IDX)
        if ((long long)blockIdx.x * G >= M) return;
    // not IDX, row_count given (mapf_encoder_forward[_save]_bounded): only the first *row_count <= M observations are computed -- the
    // learner's graph-replayed update launches on bucket-sized buffers and hands the true count over in device memory.  M stays the
    // allocated row count (the stride between the saved layers).  The rows this workgroup owns but does not compute read as zeros where
    // OTHER kernels read all M rows (the latents; the last saved layer, an operand of the 1x1 layer's weight-gradient product).
    long long Mv = M;
    if constexpr (!IDX) {
        if (row_count != nullptr) {
            const long long cnt = (long long)*row_count;
            Mv = cnt < M ? (cnt < 0 ? 0 : cnt) : M;
            const long long o0 = (long long)blockIdx.x * G;
            const long long z0 = o0 > Mv ? o0 : Mv, z1 = o0 + G < M ? o0 + G : M;
            if (z1 > z0) {
                uint4 *lz = reinterpret_cast<uint4 *>(out + z0 * 784);
                for (long long i = threadIdx.x; i < (z1 - z0) * 98; i += NTHREADS) lz[i] = make_uint4(0, 0, 0, 0);
                if constexpr (SAVE) {
                    uint4 *sz = reinterpret_cast<uint4 *>(save + (6 * M + z0) * 6272);
                    for (long long i = threadIdx.x; i < (z1 - z0) * 784; i += NTHREADS) sz[i] = make_uint4(0, 0, 0, 0);
                }
            }
            if (o0 >= Mv) return;
        }
    }
#ifdef MAPF_ENC_CLOCK
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), ref0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) alone: the stamps are back before the first LDS wait is counted
#endif
    __shared__ __attribute__((aligned(16))) unsigned char smem[ACT_BYTES + RAW_BYTES];
    unsigned char *const act = smem;
    const InT *const raw = reinterpret_cast<const InT *>(smem + ACT_BYTES);

    const int tid_o = threadIdx.x;
  for (long long wg = blockIdx.x;; wg += gridDim.x) {  // (one pass unless IDX)
    // IDX: the thread index goes through an opaque move at the top of every walk step: everything per-lane below (tile addresses,
    // masks, staging offsets) is otherwise loop-invariant, gets hoisted in front of the walk and spilled (80 bytes of scratch per
    // lane until round 4)
    int tid = tid_o;
    if constexpr (IDX) asm volatile("" : "+v"(tid));
    const int lane = tid & 63, w = tid >> 6;
    const int cb = w;
    const int lr = lane & 15, lh = lane >> 4;
    // IDX: an opaque zero (the row count is never negative) added to the weight / bias bases -- otherwise every layer's per-lane
    // weight address is loop-invariant, gets hoisted in front of the walk and spilled (44 registers in scratch)
    int z = 0;
    if constexpr (IDX) z = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int32_t *>(row_count) >> 31);
    const uint16_t *const wpz = wp + z;
    const float *const biasz = bias + z;
    const long long obs0 = wg * G;
    const long long left = Mv - obs0;
    const int nobs = left < G ? (int)left : G;  // >= 1 by the grid size (and the early return above)

    // ---- zero the activation image (its border rows must be zero; they are never written afterwards) ----
    if (!(MAPF_ENC_ABLATE & 16))
        for (int i = tid; i < ACT_BYTES / 16; i += NTHREADS) reinterpret_cast<uint4 *>(act)[i] = make_uint4(0, 0, 0, 0);
    // ---- stage the raw observations of this block (contiguous in global memory); missing ones read as zero ----
    // (IDX: packed rows of RAW_STRIDE = 488 bytes, dword-aligned each; otherwise 486 * sizeof(InT) back to back)
    constexpr int RAW_STRIDE = IDX ? MAPF_ENC_PACKED_OBS_STRIDE : OBS_ELEMS;  // elements (IDX is u8 only)
    static_assert(!IDX || sizeof(InT) == 1, "");
    {
        constexpr int DW = G * RAW_STRIDE * (int)sizeof(InT) / 4;
        static_assert((G * RAW_STRIDE * sizeof(InT)) % 4 == 0, "block input must stay 4-byte aligned");
        const uint32_t *src = reinterpret_cast<const uint32_t *>(obs + obs0 * RAW_STRIDE);
        uint32_t *dst = reinterpret_cast<uint32_t *>(smem + ACT_BYTES);
        const int have = nobs * RAW_STRIDE * (int)sizeof(InT) / 4;  // 486*sizeof(InT) is a multiple of 4 only for even sizes...
        const int have_bytes = nobs * RAW_STRIDE * (int)sizeof(InT);
        for (int i = tid; i < DW; i += NTHREADS) {
            uint32_t v = 0;
            if (i < have) {
                v = src[i];
            } else if (i * 4 < have_bytes) {  // ragged last dword (odd number of 486-byte observations)
                const unsigned char *sb = reinterpret_cast<const unsigned char *>(src);
                for (int k = 0; k < 4; ++k)
                    if (i * 4 + k < have_bytes) v |= (uint32_t)sb[i * 4 + k] << (8 * k);
            }
            dst[i] = v;
        }
    }

    // ---- per-tile geometry of this lane: byte address of (row(p) - 9) + this lane's 16-byte k slice ----
    int addr[NT];
    uint32_t vmask = 0;  // bit n: position of tile n exists
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int p = n * 16 + lr;
        const bool v = p < 49 * nobs;
        const int o = p / 49, q = p - 49 * o, y = q / 7, x = q - 7 * y;
        addr[n] = (v ? (64 * o + 8 * y + x) * ROWB : 0) + lh * 16;
        if (MAPF_ENC_ABLATE & 4) addr[n] = (n * 16 + lr) * ROWB + lh * 16;  // diagnostic: 16 CONSECUTIVE rows per tile (wrong results, conflict-free reads)
        vmask |= (v ? 1u : 0u) << n;
    }
    __syncthreads();

    f32x4 acc[2][NT];
    uint32_t nib[NT];  // training forward: this lane's ReLU sign nibbles of the layer being finished
    const float *bl = biasz;
    const int co_lane = cb * 32 + 4 * lh;  // + 16*a: first of this lane's 4 output channels

    // =========================== conv0: 6 -> 128, 3x3 valid on 9x9 (K = 54 -> 64) ===========================
    {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[a][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const el8 *wv = reinterpret_cast<const el8 *>(wpz + WP_L0) + (2 * cb) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const el8 a0 = wv[(s * 8 + 0) * 64];
            const el8 a1 = wv[(s * 8 + 1) * 64];
            int koff[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = s * 32 + lh * 8 + j;
                const int ci = k / 9, t = k - 9 * ci, ky = t / 3, kx = t - 3 * ky;
                koff[j] = k < 54 ? ci * 81 + ky * 9 + kx : -1;
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int p = n * 16 + lr;
                const bool v = (vmask >> n) & 1u;
                const int o = p / 49, q = p - 49 * o, y = q / 7, x = q - 7 * y;
                int rb = v ? o * RAW_STRIDE + y * 9 + x : 0;
                if (MAPF_ENC_ABLATE & 64) rb = lr;  // diagnostic: every tile gathers tile 0's window (the compiler keeps one gather per k-step)
                union {
                    el8 v8;
                    uint16_t u[8];
                } b;
#pragma unroll
                for (int j = 0; j < 8; ++j) b.u[j] = (koff[j] >= 0 && !(MAPF_ENC_ABLATE & 8)) ? raw_to_el(raw[rb + koff[j]]) : (uint16_t)0;
                acc[0][n] = el_mfma(a0, b.v8, acc[0][n]);
                acc[1][n] = el_mfma(a1, b.v8, acc[1][n]);
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float4 b4 = *reinterpret_cast<const float4 *>(bl + co_lane + 16 * a);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const uint2 v = pack_relu(acc[a][n], b4);
                if (SAVE) nib[n] = (a ? nib[n] : 0u) | (positive4(v) << (4 * a));
                if ((vmask >> n) & 1u) *reinterpret_cast<uint2 *>(act + addr[n] - lh * 16 + 9 * ROWB + (co_lane + 16 * a) * 2) = v;
            }
        }
        if (SAVE) store_relu_bits(nib, vmask, relu_bits + obs0 * 196, cb, lr, lh);
        bl += 128;
        __syncthreads();
    }
    const int ctotal = nobs * 49 * 16;  // 16-byte chunks of one saved layer of this block

    // =========================== 3 residual blocks ===========================
    for (int blk = 0; blk < 3; ++blk) {
        const el8 *wv1 = reinterpret_cast<const el8 *>(wpz + WP_RES + (2 * blk) * WP_RES_SIZE) + (2 * cb) * 64 + lane;
        const el8 *wv2 = wv1 + WP_RES_SIZE / 8;
        // ---- block1: t = relu(conv(x) + b1) ----
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[a][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // x (conv0's output or the previous block's) is saved layer 2 blk: copied out while this convolution reads it
        conv3x3<SAVE>(act, wv1, addr, acc, save + ((2 * blk) * M + obs0) * 6272, ctotal, tid);
        __syncthreads();  // every wave has finished reading x
        uint2 xres[2][NT];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float4 b4 = *reinterpret_cast<const float4 *>(bl + co_lane + 16 * a);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                uint2 *cell = reinterpret_cast<uint2 *>(act + addr[n] - lh * 16 + 9 * ROWB + (co_lane + 16 * a) * 2);
                xres[a][n] = *cell;  // this lane's skip input (lanes without a position read row 9 and drop it)
                const uint2 v = pack_relu(acc[a][n], b4);
                if (SAVE) nib[n] = (a ? nib[n] : 0u) | (positive4(v) << (4 * a));
                if ((vmask >> n) & 1u) *cell = v;
            }
        }
        if (SAVE) store_relu_bits(nib, vmask, relu_bits + ((1 + 2 * blk) * M + obs0) * 196, cb, lr, lh);
        __syncthreads();
        // ---- block2: x' = relu(conv(t) + b2 + x): the skip input is the initial accumulator ----
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[a][n] = f32x4{el_lo(xres[a][n].x), el_hi(xres[a][n].x), el_lo(xres[a][n].y), el_hi(xres[a][n].y)};
        conv3x3<SAVE>(act, wv2, addr, acc, save + ((1 + 2 * blk) * M + obs0) * 6272, ctotal, tid);  // t = saved layer 1 + 2 blk
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float4 b4 = *reinterpret_cast<const float4 *>(bl + 128 + co_lane + 16 * a);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const uint2 v = pack_relu(acc[a][n], b4);
                if (SAVE) nib[n] = (a ? nib[n] : 0u) | (positive4(v) << (4 * a));
                if ((vmask >> n) & 1u) *reinterpret_cast<uint2 *>(act + addr[n] - lh * 16 + 9 * ROWB + (co_lane + 16 * a) * 2) = v;
            }
        }
        if (SAVE) store_relu_bits(nib, vmask, relu_bits + ((2 + 2 * blk) * M + obs0) * 196, cb, lr, lh);
        bl += 256;
        __syncthreads();
    }
    if (SAVE) save_rows(act, save + (6 * M + obs0) * 6272, nobs, tid);  // the last block's output: no convolution left to hide in

    // =========================== conv 1x1: 128 -> 16, ReLU, NCHW flatten ===========================
    {
        const el8 *wv = reinterpret_cast<const el8 *>(wpz + WP_L7) + lane;
        el8 a7[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) a7[s] = wv[s * 64];
        const float4 b4 = *reinterpret_cast<const float4 *>(bl + 4 * lh);
        for (int n = w; n < NT && !(MAPF_ENC_ABLATE & 32); n += NTHREADS / 64) {
            const int p = n * 16 + lr;
            const bool v = p < 49 * nobs;
            const int o = p / 49, q = p - 49 * o, y = q / 7, x = q - 7 * y;
            const unsigned char *rowp = act + (v ? (64 * o + 8 * y + x + 9) * ROWB : 0) + lh * 16;
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
                c = el_mfma(a7[s], *reinterpret_cast<const el8 *>(rowp + s * 64), c);
            if (v) {
                const long long orow = IDX ? (long long)row_index[obs0 + o] : obs0 + o;
                uint16_t *dst = out + orow * 784 + (4 * lh) * 49 + q;  // latent[obs][co*49 + y*7 + x]
                dst[0] = (uint16_t)f32_to_bf16_bits(fmaxf(c[0] + b4.x, 0.f));
                dst[49] = (uint16_t)f32_to_bf16_bits(fmaxf(c[1] + b4.y, 0.f));
                dst[98] = (uint16_t)f32_to_bf16_bits(fmaxf(c[2] + b4.z, 0.f));
                dst[147] = (uint16_t)f32_to_bf16_bits(fmaxf(c[3] + b4.w, 0.f));
            }
        }
    }
    if constexpr (!IDX) break;
    if ((wg + gridDim.x) * G >= M) break;
    __syncthreads();  // every wave is done with the activation image before the next pass zeroes it
  }
#ifdef MAPF_ENC_CLOCK
    if (threadIdx.x == 0) {
        const unsigned long long clk1 = __builtin_amdgcn_s_memtime(), ref1 = __builtin_amdgcn_s_memrealtime();
        g_enc_clock[blockIdx.x & 8191][0] = clk1 - clk0;
        g_enc_clock[blockIdx.x & 8191][1] = ref1 - ref0;
    }
#endif
}

#ifdef MAPF_ENC_CLOCK
extern "C" int mapf_enc_clock_read(unsigned long long *out_host, int n) {  // [n][2] of the last launch's workgroups (n <= 8192)
    if (!out_host || n < 1 || n > 8192) return -1;
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_enc_clock), sizeof(unsigned long long) * 2 * n) == hipSuccess ? 0 : -1;
}
#endif

// ---- weight packing: fp32 [co][ci][kh][kw] (contiguous) -> f16 MFMA A fragments, biases concatenated ----
struct PackArgs {
    const float *w[8];
    const float *b[8];
    int nhwc;  // weights stored [co][kh][kw][ci] (PyTorch channels_last memory) instead of [co][ci][kh][kw]
};

// element (co, ci, tap) of a [co][CI][3][3] weight in either memory order
__device__ __forceinline__ float wread(const float *w, int nhwc, int CI, int co, int ci, int tap) {
    return nhwc ? w[(co * 9 + tap) * CI + ci] : w[(co * CI + ci) * 9 + tap];
}

__global__ void __launch_bounds__(256) encoder_pack_kernel(PackArgs pa, uint16_t *__restrict__ wp, float *__restrict__ bias) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < BIAS_TOTAL) {
        const int layer = i < 896 ? i / 128 : 7;
        bias[i] = pa.b[layer][i - layer * 128];
    }
    if (i >= WP_TOTAL) return;
    const int j = i & 7, l = (i >> 3) & 63;
    const int r = l & 15, h = l >> 4;
    float v = 0.f;
    if (i < WP_RES) {  // conv0: [s=2][c=8][l][j], k = 32 s + 8 h + j = ci*9 + tap
        const int c = (i >> 9) & 7, s = i >> 12;
        const int k = 32 * s + 8 * h + j, co = 16 * c + r;
        if (k < 54) v = wread(pa.w[0], pa.nhwc, 6, co, k / 9, k % 9);
    } else if (i < WP_L7) {  // 3x3 layers: [s=36][c=8][l][j], s = tap*4 + chunk, ci = 32 chunk + 8 h + j
        const int e = i - WP_RES, layer = e / WP_RES_SIZE, f = e - layer * WP_RES_SIZE;
        const int c = (f >> 9) & 7, s = f >> 12;
        const int tap = s >> 2, chunk = s & 3;
        const int ci = 32 * chunk + 8 * h + j, co = 16 * c + r;
        v = wread(pa.w[1 + layer], pa.nhwc, 128, co, ci, tap);
    } else {  // 1x1: [s=4][l][j], ci = 32 s + 8 h + j, co = r
        const int f = i - WP_L7, s = f >> 9;
        v = pa.w[7][r * 128 + 32 * s + 8 * h + j];
    }
    wp[i] = f32_to_el_bits(v);
}

// =====================================================================================================
// Backward-data chain of the encoder in ONE kernel (the mirror image of encoder_fwd_kernel).
//
// In: gz7 = gradient w.r.t. the 1x1 convolution's pre-activation (already ReLU-masked), bf16 [M][49][16];
// the ReLU sign bits of the 7 layer outputs written by the training forward (uint32 [7][M][49][4]); weights packed for the TRANSPOSED
// convolutions (mapf_encoder_pack_bwd: channels swapped, taps flipped), so that every step
//     g_in[ci][p] = sum_{co,tap} W[co][ci][tap] * gz[co][p - tap]
// is the same LDS-resident implicit GEMM as the forward (conv3x3 above).
// Out: gz [7][M][49][128] = the gradient w.r.t. every 128-channel layer's PRE-activation (ReLU-masked with those
// sign bits) -- exactly what the weight-/bias-gradient reductions consume.
// Residual block backward (y = relu(x + conv2(t) + b2), t = relu(conv1(x) + b1)):
//     gz2 = g_y * (y > 0);  g_t = conv2^T(gz2);  gz1 = g_t * (t > 0);  g_x = conv1^T(gz1) + gz2
// -- the skip term gz2 is what LDS holds when gz1 overwrites it, so, as in the forward, each lane reads its own
// elements back as the initial accumulator of conv1^T.
// =====================================================================================================
constexpr int GZ7_BYTES = G * 49 * 16 * 2;  // 6272
static_assert(2 * (ACT_BYTES + GZ7_BYTES) <= 160 * 1024, "LDS budget: two workgroups per CU");
constexpr int WPT_L7 = 6 * WP_RES_SIZE;
constexpr int WPT_TOTAL = WPT_L7 + 8 * 512;
static_assert(WPT_TOTAL == MAPF_ENC_PACKED_BWD_ELEMS, "header constant out of date");

// Loss scale of the backward chain: the power of two that puts the largest magnitude of the gradient arriving at the encoder's
// output into (8, 16] -- 12 binary orders of headroom below f16's 65504 for growth inside the chain (conversions clamp beyond
// that), the small end of the distribution as far above f16's subnormal range as it can be.  `max_bits`: that largest magnitude
// as bf16 bits (grad_absmax_kernel).  Exact: scaling by 2^k and back changes no mantissa.
__device__ __forceinline__ float grad_scale_from_max(uint32_t max_bits) {
    int k = 130 - (int)((max_bits >> 7) & 0xFFu);  // the maximum lies in [2^(e-127), 2^(e-126)): times 2^(130-e) -> [8, 16)
    k = k < -24 ? -24 : (k > 60 ? 60 : k);
    return __uint_as_float((uint32_t)(k + 127) << 23);
}
__global__ void zero_scale_kernel(uint32_t *p) {
    if (threadIdx.x < 2) p[threadIdx.x] = 0u;
}

__global__ void __launch_bounds__(256) grad_absmax_kernel(const uint4 *__restrict__ g, long long chunks, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) {
        const uint4 v = g[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = w[k] & 0x7FFFu, hi = (w[k] >> 16) & 0x7FFFu;
            m = max(m, max(lo < 0x7F80u ? lo : 0u, hi < 0x7F80u ? hi : 0u));  // (inf / NaN do not set the scale)
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    // one atomic per WORKGROUP: one per wave of a 1024-block grid was 4096 atomics on one address, ~40 of this kernel's 48 us
    __shared__ uint32_t s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        if (m) atomicMax(out, m);
    }
}
__device__ __forceinline__ float clamp_el(float v) { return __builtin_amdgcn_fmed3f(v, -EL_MAX, EL_MAX); }

// masked gradient of 4 channels, packed to f16 (still multiplied by the loss scale); `bs` accumulates this lane's share of the bias gradient
__device__ __forceinline__ uint2 pack_masked(const f32x4 &a, uint32_t m, float (&bs)[4]) {
    const float v0 = (m & 1u) ? a[0] : 0.f, v1 = (m & 2u) ? a[1] : 0.f, v2 = (m & 4u) ? a[2] : 0.f, v3 = (m & 8u) ? a[3] : 0.f;
    bs[0] += v0;
    bs[1] += v1;
    bs[2] += v2;
    bs[3] += v3;
    return make_uint2(el_pack2(clamp_el(v0), clamp_el(v1)), el_pack2(clamp_el(v2), clamp_el(v3)));
}
// Sum over the 16 lanes of a DPP row (every lane ends up with the total): row rotations by 8 and 4, then the two quad
// permutations.  VALU-only: __shfl_xor compiles to ds_bpermute_b32, an LDS-pipe instruction with LDS latency.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row_sum16(float v) {
    v = dpp_add<0x128>(v);  // row_ror:8
    v = dpp_add<0x124>(v);  // row_ror:4
    v = dpp_add<0x4E>(v);   // quad_perm:[2,3,0,1]
    return dpp_add<0xB1>(v);  // quad_perm:[1,0,3,2]
}
// Sum the per-lane bias shares over the 16 lanes that hold the same channels (different positions) and store this
// workgroup's partial bias gradient of one layer: dst[co] for co = co_lane + 16 a + r (each written by one lane).
__device__ __forceinline__ void store_bias_partial(float (&bs)[2][4], float *__restrict__ dst, int co_lane, int lr, float inv_scale) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = row_sum16(bs[a][r]) * inv_scale;
            if (lr == 0) dst[co_lane + 16 * a + r] = v;
            bs[a][r] = 0.f;
        }
}

// HEAD = true: the ReLU mask of the 1x1 layer is applied here as well -- `gz7` is then the gradient w.r.t. the encoder's
// OUTPUT, bf16 [M][784] in the forward's flattened NCHW order (channel * 49 + position), `latent` that output; the masked
// gradient is transposed into position-major order while it is staged, and also written to `gz7_out` [M][49][16] (the
// caller's weight-gradient GEMM wants it) together with this workgroup's partial bias gradient `gb7_part` [blocks][16].
template <bool HEAD>
__global__ void __launch_bounds__(NTHREADS, 2) encoder_bwd_kernel(const uint16_t *__restrict__ gz7, long long M,
                                                                  const uint32_t *__restrict__ relu_bits,
                                                                  const uint16_t *__restrict__ wpt, uint16_t *__restrict__ gz,
                                                                  float *__restrict__ gb_part, const uint16_t *__restrict__ latent,
                                                                  uint16_t *__restrict__ gz7_out, float *__restrict__ gb7_part,
                                                                  uint32_t *__restrict__ grad_scale, const int32_t *__restrict__ valid_rows) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[ACT_BYTES + GZ7_BYTES];
    unsigned char *const act = smem;
    const unsigned char *const raw = smem + ACT_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, cb = tid >> 6;
    const int lr = lane & 15, lh = lane >> 4;
    const long long obs0 = (long long)blockIdx.x * G;
    // valid_rows (mapf_encoder_backward_bounded): only the first *valid_rows <= M observations carry a gradient (see encoder_fwd_kernel);
    // M stays the allocated row count (layer stride).  What this workgroup owns but skips reads as zeros where other kernels read all M
    // rows or all workgroups' partials: gz7_out, the bias partials.
    long long Mv = M;
    if (valid_rows != nullptr) {
        const long long cnt = (long long)*valid_rows;
        Mv = cnt < M ? (cnt < 0 ? 0 : cnt) : M;
        if (HEAD) {
            const long long z0 = obs0 > Mv ? obs0 : Mv, z1 = obs0 + G < M ? obs0 + G : M;
            if (z1 > z0) {
                uint4 *gzz = reinterpret_cast<uint4 *>(gz7_out + z0 * (49 * 16));
                for (long long i = tid; i < (z1 - z0) * 98; i += NTHREADS) gzz[i] = make_uint4(0, 0, 0, 0);
            }
        }
        if (obs0 >= Mv) {
            if (HEAD && blockIdx.x == 0 && tid == 0) grad_scale[1] = __float_as_uint(1.f / grad_scale_from_max(grad_scale[0]));
            for (int i = tid; i < 7 * 128; i += NTHREADS) gb_part[((long long)(i >> 7) * gridDim.x + blockIdx.x) * 128 + (i & 127)] = 0.f;
            if (HEAD && lane < 16) gb7_part[((long long)blockIdx.x * 4 + cb) * 16 + lane] = 0.f;
            return;
        }
    }
    const long long left = Mv - obs0;
    const int nobs = left < G ? (int)left : G;

    // HEAD: the chain runs on gradients times `scale`; bias partials are divided by it again here, the weight-gradient kernels
    // divide theirs (they read 1/scale from grad_scale[1]).  Without HEAD the caller's gz7 is taken as it is.
    const float scale = HEAD ? grad_scale_from_max(grad_scale[0]) : 1.f, inv_scale = 1.f / scale;
    if (HEAD && blockIdx.x == 0 && tid == 0) grad_scale[1] = __float_as_uint(inv_scale);
    for (int i = tid; i < ACT_BYTES / 16; i += NTHREADS) reinterpret_cast<uint4 *>(act)[i] = make_uint4(0, 0, 0, 0);
    if (HEAD) {  // mask with (latent > 0), scale, convert bf16 -> f16 and transpose [obs][c][p] -> [obs][p][c] on the way into LDS
        // (16-byte loads of both tensors issued together; one observation is 98 chunks, so a chunk is all-valid or all-padding)
        const uint4 *gsrc = reinterpret_cast<const uint4 *>(gz7 + obs0 * 784), *lsrc = reinterpret_cast<const uint4 *>(latent + obs0 * 784);
        uint16_t *dst = reinterpret_cast<uint16_t *>(smem + ACT_BYTES);
        const int have = nobs * 98;
        uint4 gv[2], lv[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ch = tid + r * NTHREADS;
            gv[r] = lv[r] = make_uint4(0, 0, 0, 0);
            if (ch < have) {
                gv[r] = gsrc[ch];
                lv[r] = lsrc[ch];
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ch = tid + r * NTHREADS;
            if (ch >= G * 98) break;
            const uint32_t gw[4] = {gv[r].x, gv[r].y, gv[r].z, gv[r].w}, lw[4] = {lv[r].x, lv[r].y, lv[r].z, lv[r].w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = ch * 8 + k, o = i / 784, rem = i - 784 * o, c = rem / 49, p = rem - 49 * c;
                const uint32_t y = (lw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu, gq = (gw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                dst[(o * 49 + p) * 16 + c] = ((y & 0x7FFFu) != 0u && !(y & 0x8000u)) ? f32_to_el_bits(bf16_bits_to_f32(gq) * scale) : (uint16_t)0;
            }
        }
    } else {  // gz7 rows of this block: contiguous, 32 B per position
        const uint4 *src = reinterpret_cast<const uint4 *>(gz7 + obs0 * (49 * 16));
        uint4 *dst = reinterpret_cast<uint4 *>(smem + ACT_BYTES);
        const int have = nobs * 49 * 2;
        for (int i = tid; i < GZ7_BYTES / 16; i += NTHREADS) dst[i] = i < have ? src[i] : make_uint4(0, 0, 0, 0);
    }
    int addr[NT];
    uint32_t vmask = 0;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int p = n * 16 + lr;
        const bool v = p < 49 * nobs;
        const int o = p / 49, q = p - 49 * o, y = q / 7, x = q - 7 * y;
        addr[n] = (v ? (64 * o + 8 * y + x) * ROWB : 0) + lh * 16;
        vmask |= (v ? 1u : 0u) << n;
    }
    __syncthreads();
    if (HEAD) {
        const uint4 *src = reinterpret_cast<const uint4 *>(smem + ACT_BYTES);
        uint4 *dst = reinterpret_cast<uint4 *>(gz7_out + obs0 * (49 * 16));
        for (int i = tid; i < nobs * 49 * 2; i += NTHREADS) dst[i] = src[i];
        {   // bias gradient of the 1x1 layer: wave cb sums the 49 rows of observation cb (padding rows are zero), 4 rows at a time
            const uint16_t *rows = reinterpret_cast<const uint16_t *>(smem + ACT_BYTES) + (49 * cb) * 16 + lr;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 13; ++j)
                if (lh + 4 * j < 49) v += el_to_f32(rows[(lh + 4 * j) * 16]);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) gb7_part[((long long)blockIdx.x * 4 + cb) * 16 + lane] = v * inv_scale;
        }
    }

    const int co_lane = cb * 32 + 4 * lh;
    // this lane's element (a, n) of a saved activation / gz tensor of layer k: 4 channels = 8 bytes
    auto cell_off = [&](int a, int n) -> long long { return ((obs0 * 49 + n * 16 + lr) * 128 + co_lane + 16 * a); };
    const long long LSTRIDE = M * 6272;
    float bs[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // bias-gradient partials: gb_part[layer][block][128]
    auto gb_dst = [&](int layer) -> float * { return gb_part + ((long long)layer * gridDim.x + blockIdx.x) * 128; };

    // ReLU sign bits of the layer whose mask comes next: this lane's own byte per tile (see store_relu_bits: nibble a in
    // bits 4 a .. 4 a + 3); loaded one convolution ahead, so the reads are long finished when needed
    uint32_t mk[NT];
    auto load_masks = [&](int layer) {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(relu_bits + ((long long)layer * M + obs0) * 196) + cb * 4 + lh;
#pragma unroll
        for (int n = 0; n < NT; ++n) mk[n] = ((vmask >> n) & 1u) ? (uint32_t)src[(n * 16 + lr) * 16] : 0u;
    };
    load_masks(6);

    f32x4 acc[2][NT];
    // ---- 1x1^T: g_y3[ci][p] = sum_co W7[co][ci] gz7[co][p]  (K = 16, zero-padded to 32) ----
    {
        const el8 *wv = reinterpret_cast<const el8 *>(wpt + WPT_L7) + (2 * cb) * 64 + lane;
        const el8 a0 = wv[0], a1 = wv[64];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int p = n * 16 + lr;
            el8 b = {0, 0, 0, 0, 0, 0, 0, 0};
            if (((vmask >> n) & 1u) && lh < 2) b = *reinterpret_cast<const el8 *>(raw + p * 32 + lh * 16);
            acc[0][n] = el_mfma(a0, b, f32x4{0.f, 0.f, 0.f, 0.f});
            acc[1][n] = el_mfma(a1, b, f32x4{0.f, 0.f, 0.f, 0.f});
        }
    }

    for (int blk = 2; blk >= 0; --blk) {
        const el8 *wv1 = reinterpret_cast<const el8 *>(wpt + (2 * blk) * WP_RES_SIZE) + (2 * cb) * 64 + lane;
        const el8 *wv2 = wv1 + WP_RES_SIZE / 8;
        // ---- gz2 = g_y * (y > 0) -> LDS (input of conv2^T) ----
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                if ((vmask >> n) & 1u)
                    *reinterpret_cast<uint2 *>(act + addr[n] - lh * 16 + 9 * ROWB + (co_lane + 16 * a) * 2) =
                        pack_masked(acc[a][n], (mk[n] >> (4 * a)) & 0xFu, bs[a]);
        store_bias_partial(bs, gb_dst(2 + 2 * blk), co_lane, lr, inv_scale);
        load_masks(1 + 2 * blk);
        __syncthreads();
        // ---- g_t = conv2^T(gz2) ----
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[a][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        conv3x3<true>(act, wv2, addr, acc, gz + (2 + 2 * blk) * LSTRIDE + obs0 * 6272, nobs * 49 * 16, tid);  // gz2 out while it is read
        __syncthreads();
        // ---- gz1 = g_t * (t > 0) -> LDS; the gz2 it overwrites is the skip term of g_x ----
        uint2 skip[2][NT];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                uint2 *cell = reinterpret_cast<uint2 *>(act + addr[n] - lh * 16 + 9 * ROWB + (co_lane + 16 * a) * 2);
                skip[a][n] = *cell;
                if ((vmask >> n) & 1u) *cell = pack_masked(acc[a][n], (mk[n] >> (4 * a)) & 0xFu, bs[a]);
            }
        store_bias_partial(bs, gb_dst(1 + 2 * blk), co_lane, lr, inv_scale);
        load_masks(2 * blk);  // y of the previous block, or conv0's output for blk = 0
        __syncthreads();
        // ---- g_x = conv1^T(gz1) + gz2 ----
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[a][n] = f32x4{el_lo(skip[a][n].x), el_hi(skip[a][n].x), el_lo(skip[a][n].y), el_hi(skip[a][n].y)};
        conv3x3<true>(act, wv1, addr, acc, gz + (1 + 2 * blk) * LSTRIDE + obs0 * 6272, nobs * 49 * 16, tid);
        __syncthreads();
    }
    // ---- gz0 = g_y0 * (y0 > 0): straight to global memory (conv0 has no data gradient) ----
    {
        uint16_t *g0 = gz;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                if ((vmask >> n) & 1u)
                    *reinterpret_cast<uint2 *>(g0 + cell_off(a, n)) = pack_masked(acc[a][n], (mk[n] >> (4 * a)) & 0xFu, bs[a]);
        store_bias_partial(bs, gb_dst(0), co_lane, lr, inv_scale);
    }
}

// transposed-convolution weight image: [6 layers][s=36][c=8][lane][j] with o = 16c + (lane&15) the INPUT channel of
// the forward layer, i = 32 chunk + 8 (lane>>4) + j its OUTPUT channel, tap flipped; then W7^T [c=8][lane][j]
__global__ void __launch_bounds__(256) encoder_pack_bwd_kernel(PackArgs pa, uint16_t *__restrict__ wpt) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= WPT_TOTAL) return;
    const int j = idx & 7, l = (idx >> 3) & 63, r = l & 15, h = l >> 4;
    float v = 0.f;
    if (idx < WPT_L7) {
        const int layer = idx / WP_RES_SIZE, f = idx - layer * WP_RES_SIZE;
        const int c = (f >> 9) & 7, s = f >> 12;
        const int tap = s >> 2, chunk = s & 3;
        const int i = 32 * chunk + 8 * h + j, o = 16 * c + r;
        v = wread(pa.w[1 + layer], pa.nhwc, 128, i, o, 8 - tap);
    } else {
        const int c = ((idx - WPT_L7) >> 9) & 7, k = 8 * h + j, o = 16 * c + r;
        if (k < 16) v = pa.w[7][k * 128 + o];
    }
    wpt[idx] = f32_to_el_bits(v);
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            std::fprintf(stderr, "mapf_encoder: %s failed: %s\n", #expr, hipGetErrorString(_e)); \
            return MAPF_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)

}  // namespace

extern "C" {

int mapf_encoder_pack(const float *const *w_dev, const float *const *b_dev, int weights_nhwc, uint16_t *packed_dev, float *bias_dev, void *stream) {
    if (!w_dev || !b_dev || !packed_dev || !bias_dev) return MAPF_ERR_INVALID_ARG;
    PackArgs pa;
    pa.nhwc = weights_nhwc != 0;
    for (int i = 0; i < 8; ++i) {
        if (!w_dev[i] || !b_dev[i]) return MAPF_ERR_INVALID_ARG;
        pa.w[i] = w_dev[i];
        pa.b[i] = b_dev[i];
    }
    hipLaunchKernelGGL(encoder_pack_kernel, dim3((WP_TOTAL + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), pa,
                       packed_dev, bias_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

static int encoder_launch(const void *obs_dev, int obs_dtype, int64_t M, const uint16_t *packed_dev, const float *bias_dev,
                          uint16_t *latent_dev, uint16_t *save_dev, uint32_t *bits_dev, bool save, void *stream,
                          const int32_t *valid_rows_dev = nullptr) {
    if (M < 0 || !packed_dev || !bias_dev || (M > 0 && (!obs_dev || !latent_dev || (save && (!save_dev || !bits_dev))))) return MAPF_ERR_INVALID_ARG;
    if (obs_dtype != MAPF_ENC_OBS_U8 && obs_dtype != MAPF_ENC_OBS_BF16) return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(obs_dev) & 3) || (reinterpret_cast<uintptr_t>(packed_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(bias_dev) & 15) || (reinterpret_cast<uintptr_t>(latent_dev) & 1) ||
        (reinterpret_cast<uintptr_t>(save_dev) & 15) || (reinterpret_cast<uintptr_t>(bits_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    if (valid_rows_dev && ((reinterpret_cast<uintptr_t>(latent_dev) & 15) || (reinterpret_cast<uintptr_t>(valid_rows_dev) & 3))) return MAPF_ERR_INVALID_ARG;
    if (M == 0) return MAPF_OK;
    const long long blocks = (M + G - 1) / G;
    if (blocks > 0x7FFFFFFFLL) return MAPF_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)blocks), block(NTHREADS);
    const uint8_t *o8 = static_cast<const uint8_t *>(obs_dev);
    const uint16_t *o16 = static_cast<const uint16_t *>(obs_dev);
    if (obs_dtype == MAPF_ENC_OBS_U8 && !save)
        hipLaunchKernelGGL((encoder_fwd_kernel<uint8_t, false>), grid, block, 0, st, o8, (long long)M, packed_dev, bias_dev, latent_dev, save_dev, bits_dev, (const int32_t *)nullptr, valid_rows_dev);
    else if (obs_dtype == MAPF_ENC_OBS_U8)
        hipLaunchKernelGGL((encoder_fwd_kernel<uint8_t, true>), grid, block, 0, st, o8, (long long)M, packed_dev, bias_dev, latent_dev, save_dev, bits_dev, (const int32_t *)nullptr, valid_rows_dev);
    else if (!save)
        hipLaunchKernelGGL((encoder_fwd_kernel<uint16_t, false>), grid, block, 0, st, o16, (long long)M, packed_dev, bias_dev, latent_dev, save_dev, bits_dev, (const int32_t *)nullptr, valid_rows_dev);
    else
        hipLaunchKernelGGL((encoder_fwd_kernel<uint16_t, true>), grid, block, 0, st, o16, (long long)M, packed_dev, bias_dev, latent_dev, save_dev, bits_dev, (const int32_t *)nullptr, valid_rows_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_encoder_forward(const void *obs_dev, int obs_dtype, int64_t M, const uint16_t *packed_dev, const float *bias_dev,
                         uint16_t *latent_dev, void *stream) {
    return encoder_launch(obs_dev, obs_dtype, M, packed_dev, bias_dev, latent_dev, nullptr, nullptr, false, stream);
}

int mapf_encoder_forward_rows(const uint8_t *obs_dev, int64_t max_rows, const int32_t *row_index_dev, const int32_t *row_count_dev,
                              const uint16_t *packed_dev, const float *bias_dev, uint16_t *latent_dev, void *stream) {
    if (max_rows < 0 || !packed_dev || !bias_dev || !row_index_dev || !row_count_dev || (max_rows > 0 && (!obs_dev || !latent_dev)))
        return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(obs_dev) & 3) || (reinterpret_cast<uintptr_t>(packed_dev) & 15) || (reinterpret_cast<uintptr_t>(bias_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(latent_dev) & 1))
        return MAPF_ERR_INVALID_ARG;
    if (max_rows == 0) return MAPF_OK;
    long long blocks = (max_rows + G - 1) / G;
    if (blocks > 2048) blocks = 2048;  // 2 workgroups per CU x 256 CUs x 4: they walk the list (see the kernel)
    hipLaunchKernelGGL((encoder_fwd_kernel<uint8_t, false, true>), dim3((unsigned)blocks), dim3(NTHREADS), 0, static_cast<hipStream_t>(stream), obs_dev,
                       (long long)max_rows, packed_dev, bias_dev, latent_dev, (uint16_t *)nullptr, (uint32_t *)nullptr, row_index_dev, row_count_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_encoder_forward_save(const void *obs_dev, int obs_dtype, int64_t M, const uint16_t *packed_dev, const float *bias_dev,
                              uint16_t *latent_dev, uint16_t *acts_dev, uint32_t *relu_bits_dev, void *stream) {
    return encoder_launch(obs_dev, obs_dtype, M, packed_dev, bias_dev, latent_dev, acts_dev, relu_bits_dev, true, stream);
}

int mapf_encoder_forward_bounded(const void *obs_dev, int obs_dtype, int64_t M, const int32_t *valid_rows_dev, const uint16_t *packed_dev,
                                 const float *bias_dev, uint16_t *latent_dev, void *stream) {
    if (!valid_rows_dev) return MAPF_ERR_INVALID_ARG;
    return encoder_launch(obs_dev, obs_dtype, M, packed_dev, bias_dev, latent_dev, nullptr, nullptr, false, stream, valid_rows_dev);
}

int mapf_encoder_forward_save_bounded(const void *obs_dev, int obs_dtype, int64_t M, const int32_t *valid_rows_dev, const uint16_t *packed_dev,
                                      const float *bias_dev, uint16_t *latent_dev, uint16_t *acts_dev, uint32_t *relu_bits_dev, void *stream) {
    if (!valid_rows_dev) return MAPF_ERR_INVALID_ARG;
    return encoder_launch(obs_dev, obs_dtype, M, packed_dev, bias_dev, latent_dev, acts_dev, relu_bits_dev, true, stream, valid_rows_dev);
}

int mapf_encoder_pack_bwd(const float *const *w_dev, int weights_nhwc, uint16_t *packed_bwd_dev, void *stream) {
    if (!w_dev || !packed_bwd_dev || (reinterpret_cast<uintptr_t>(packed_bwd_dev) & 15)) return MAPF_ERR_INVALID_ARG;
    PackArgs pa;
    pa.nhwc = weights_nhwc != 0;
    for (int i = 0; i < 8; ++i) {
        if (!w_dev[i]) return MAPF_ERR_INVALID_ARG;
        pa.w[i] = w_dev[i];
        pa.b[i] = nullptr;
    }
    hipLaunchKernelGGL(encoder_pack_bwd_kernel, dim3((WPT_TOTAL + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), pa,
                       packed_bwd_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_encoder_backward_data(const uint16_t *gz7_dev, int64_t M, const uint32_t *relu_bits_dev, const uint16_t *packed_bwd_dev,
                               uint16_t *gz_dev, float *gbias_partial_dev, void *stream) {
    if (M < 0 || !packed_bwd_dev || (M > 0 && (!gz7_dev || !relu_bits_dev || !gz_dev || !gbias_partial_dev))) return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(gz7_dev) & 15) || (reinterpret_cast<uintptr_t>(relu_bits_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(packed_bwd_dev) & 15) || (reinterpret_cast<uintptr_t>(gz_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    if (M == 0) return MAPF_OK;
    const long long blocks = (M + G - 1) / G;
    if (blocks > 0x7FFFFFFFLL) return MAPF_ERR_INVALID_ARG;
    hipLaunchKernelGGL(encoder_bwd_kernel<false>, dim3((unsigned)blocks), dim3(NTHREADS), 0, static_cast<hipStream_t>(stream), gz7_dev,
                       (long long)M, relu_bits_dev, packed_bwd_dev, gz_dev, gbias_partial_dev, nullptr, nullptr, nullptr, nullptr, (const int32_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

static int encoder_backward_launch(const uint16_t *g_latent_dev, const uint16_t *latent_dev, int64_t M, const int32_t *valid_rows_dev,
                                   const uint32_t *relu_bits_dev, const uint16_t *packed_bwd_dev, uint16_t *gz_dev, float *gbias_partial_dev,
                                   uint16_t *gz7_dev, float *gb7_partial_dev, uint32_t *grad_scale_dev, void *stream) {
    if (M < 0 || !packed_bwd_dev || !grad_scale_dev || (reinterpret_cast<uintptr_t>(grad_scale_dev) & 3) ||
        (M > 0 && (!g_latent_dev || !latent_dev || !relu_bits_dev || !gz_dev || !gbias_partial_dev || !gz7_dev || !gb7_partial_dev)))
        return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(g_latent_dev) & 15) || (reinterpret_cast<uintptr_t>(latent_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(relu_bits_dev) & 15) || (reinterpret_cast<uintptr_t>(packed_bwd_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(gz_dev) & 15) || (reinterpret_cast<uintptr_t>(gz7_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    if (M == 0) return MAPF_OK;
    const long long blocks = (M + G - 1) / G;
    if (blocks > 0x7FFFFFFFLL) return MAPF_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the loss scale: largest |g_latent| -> grad_scale_dev[0]; the chain below derives the power of two from it and leaves its
    // inverse in grad_scale_dev[1] for the weight-gradient kernels
    // (a kernel, not hipMemsetAsync: this call is captured into the update's HIP graphs; see mapf_obs_changed in csrc/mapf_actor.hip)
    hipLaunchKernelGGL(zero_scale_kernel, dim3(1), dim3(64), 0, st, grad_scale_dev);
    const long long chunks = M * 98;  // 784 bf16 = 98 x 16 bytes per observation
    hipLaunchKernelGGL(grad_absmax_kernel, dim3((unsigned)(chunks < 256 * 512 ? (chunks + 255) / 256 : 512)), dim3(256), 0, st,
                       reinterpret_cast<const uint4 *>(g_latent_dev), chunks, grad_scale_dev);
    hipLaunchKernelGGL(encoder_bwd_kernel<true>, dim3((unsigned)blocks), dim3(NTHREADS), 0, st, g_latent_dev,
                       (long long)M, relu_bits_dev, packed_bwd_dev, gz_dev, gbias_partial_dev, latent_dev, gz7_dev, gb7_partial_dev, grad_scale_dev,
                       valid_rows_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_encoder_backward(const uint16_t *g_latent_dev, const uint16_t *latent_dev, int64_t M, const uint32_t *relu_bits_dev,
                          const uint16_t *packed_bwd_dev, uint16_t *gz_dev, float *gbias_partial_dev, uint16_t *gz7_dev,
                          float *gb7_partial_dev, uint32_t *grad_scale_dev, void *stream) {
    return encoder_backward_launch(g_latent_dev, latent_dev, M, nullptr, relu_bits_dev, packed_bwd_dev, gz_dev, gbias_partial_dev, gz7_dev,
                                   gb7_partial_dev, grad_scale_dev, stream);
}

int mapf_encoder_backward_bounded(const uint16_t *g_latent_dev, const uint16_t *latent_dev, int64_t M, const int32_t *valid_rows_dev,
                                  const uint32_t *relu_bits_dev, const uint16_t *packed_bwd_dev, uint16_t *gz_dev, float *gbias_partial_dev,
                                  uint16_t *gz7_dev, float *gb7_partial_dev, uint32_t *grad_scale_dev, void *stream) {
    if (!valid_rows_dev || (reinterpret_cast<uintptr_t>(valid_rows_dev) & 3)) return MAPF_ERR_INVALID_ARG;
    return encoder_backward_launch(g_latent_dev, latent_dev, M, valid_rows_dev, relu_bits_dev, packed_bwd_dev, gz_dev, gbias_partial_dev, gz7_dev,
                                   gb7_partial_dev, grad_scale_dev, stream);
}

}  // extern "C"
