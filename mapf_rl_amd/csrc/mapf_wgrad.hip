// mapf_wgrad.hip -- weight gradient of the encoder's 3x3 128->128 convolutions (reference model.py:30-42, the
// backward of ResBlock.block1/block2 that autograd derives for `Learner.train`, worker.py:316) as one streaming
// MFMA kernel per layer (see include/mapf_dqn.h: mapf_encoder_wgrad).
//
//     dW[co][ky][kx][ci] = sum over observations m and positions (y,x) of
//                          gz[m][y][x][co] * a[m][y+ky-1][x+kx-1][ci]            (zero outside the 7x7 image)
//
// gz = ReLU-masked pre-activation gradient of the layer (mapf_encoder_backward_data), a = the layer's input
// (mapf_encoder_forward_save), both f16 [M][49][128] (gz times the chain's loss scale).  As a GEMM: 128 (co) x 1152 (tap, ci) outputs, K = all
// positions of all observations -- tiny output, enormous K, so the output is held in registers and the
// operands stream through LDS:
//  * A workgroup (512 threads = 8 waves, two per SIMD) owns one of TWO slabs of the output -- all 128 co x 9 taps x
//    64 of the input channels -- for a partition of the observations; wave w holds 64 co x (9 taps x 16 ci) = 4 x 9
//    tiles of 16x16 (144 accumulator registers).  Splitting by INPUT CHANNEL means a workgroup stages only half of
//    every input row.
//  * K RUNS OVER IMAGE ROWS: the partition's 7 * observations image rows form one stream; a K slot is (image row,
//    x' = 0..7) with x' = 7 a padding slot (its gz row is zero), a block is 8 image rows = 64 slots = two MFMA k-steps,
//    regardless of observation boundaries: 1/8 of the MFMAs multiply zeros.  (Round 1 gave every observation 64 slots,
//    49 real: 23 % zeros.  A fully dense position stream was built too, 0 % zeros -- and 38 % of its LDS cycles were
//    bank conflicts: 8 consecutive positions of a 7-wide image span 9 bordered rows minus a border row that differs
//    per kx tap, and no row -> bank map separates all three; here a half-wave reads the 8 slots of ONE image row,
//    8 consecutive rows for every tap.)
//  * gz blocks (64 slot rows of 256 B; the x' = 7 rows are zeroed once and never written) go through a ring of four
//    16-KiB LDS buffers.
//  * The INPUT lives as the zero-bordered 8-wide image of the forward kernel: bordered image row (8 cells of 128 B =
//    one 1-KiB load chunk, cell 0 the left border) number 8 * obs + y + 1, in a circular buffer of 7 observations = 56
//    chunks; the top border row of an observation is the bottom border row of the one before, tap (ky,kx) of slot
//    (row, x') is cell 8 ky + kx further on.  Border cells are zeroed once and never written.  A block needs exactly 8
//    new image rows (rows 8 j + 1 .. 8 j + 8; its taps reach one row back and one ahead), one per wave.
//  * Both operands need K along the fragment's register axis while memory has channels contiguous:
//    ds_read_b64_tr_b16 (hardware transpose read) delivers a [4 slots x 16 channels] block column-major.
//  * No row padding: rows are 256 B (gz) / 128 B (input) and bank conflicts are avoided by XOR-swizzling the 32-byte
//    column groups with the row number -- free on the way in, because a lane of global_load_lds_dwordx4 (HBM -> LDS
//    without passing through registers; lane i writes LDS bytes [16 i, 16 i + 16) of a 1-KiB chunk from ANY global
//    address, masked lanes write nothing: tools/micro/lds_direct_load.hip) simply fetches the 16 bytes that belong in
//    its slot; on the way out the swizzle of a slot's taps depends only on the lane (x' + kx), so it is folded into
//    three per-lane constants.  Everything that varies per block (which observation / row an image row is) is
//    wave-uniform and computed on the scalar unit.
//  * Every wave issues the same 2 gz + 1 input loads per block, spread over the block's tile-steps (issued together
//    behind the barrier they queue up in the CU's one vector-memory pipeline).  The loads are inline asm (hipcc's
//    wait-count pass would make every later LDS read wait for a builtin LDS-DMA load it cannot disambiguate) and counted
//    by hand: `s_waitcnt vmcnt(3)` at the end of a block means "all but the newest block have landed".  While block j
//    is multiplied, j+1 and j+2 are resident and j+3 is in flight.
//  * The MFMA is inline asm with the destination TIED to the accumulator: hipcc does not tie the builtin's, and a loop
//    that carries its accumulators then needs register-shuffle space (v_accvgpr_mov storms).
//  * The two slabs of one observation partition run on the same XCD (ids i, i+8 share an L2), so gz comes from HBM once.
//  * Output: per-partition partial sums fp32 [P][128][3][3][128]; the caller adds the P slabs (deterministic).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <type_traits>

#include "mapf_dqn.h"
#include "mapf_env.h"

__device__ __attribute__((aligned(16))) unsigned int g_wgrad_zero[4];  // what the loads of image rows behind the partition's end read

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 el8;  // the encoder kernels' element type is f16 (mapf_encoder.hip)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
template <int V>
using I = std::integral_constant<int, V>;

#ifndef MAPF_WGRAD_ABLATE  // diagnostic builds only (tools/micro/wgrad_ab.py): 1 = no staging, 2 = no fragment reads, 4 = no barrier
#define MAPF_WGRAD_ABLATE 0
#endif
constexpr int SLABS = MAPF_ENC_WGRAD_SLABS;  // input-channel halves of the output: one workgroup each per partition (2)
constexpr int NW = 8;                        // waves = 2 co halves x 4 groups of 16 input channels
constexpr int CT = 4;                        // 16-row co tiles per wave (64 co)
constexpr int NTHR = 64 * NW;
constexpr int NTAP = 9;                      // 16-column tiles per wave: one per tap
static_assert(SLABS == 2, "");
constexpr int RING_OBS = 7;                  // input ring: observations (8 bordered image rows of 1 KiB each)
constexpr int RING_CHUNKS = 8 * RING_OBS;    // 56
constexpr int NBUF = 4;                      // gz ring: computing block j, j+1 and j+2 resident, j+3 in flight
constexpr int GZ_BLK = 64 * 256;
constexpr int IN_BASE = NBUF * GZ_BLK;       // 65,536: gz buffers first, so that their fragment reads fit the 16-bit DS offset
constexpr int IN_BYTES = (8 * RING_CHUNKS + 8 + 18) * 128;  // + what the taps of the last ring row reach: border, always zero
constexpr int LDS_BYTES = IN_BASE + IN_BYTES;
constexpr int RB = 4;                        // ring of input (B) fragments: tile t's MFMAs run while tile t + RB - 1 is being read
constexpr int LPW = 3;                       // loads per wave and block: gz chunks w and w + 8, input image row 8 j + 1 + w
static_assert((2 * NTAP * NBUF) % RB == 0 && LDS_BYTES <= 160 * 1024, "");
static_assert(MAPF_ENC_WGRAD_PARTS % 8 == 0, "");

__device__ __forceinline__ el8 tr_read2(const unsigned char *p0, const unsigned char *p1) {
    // two transposed 4-row blocks -> the 8 k-elements of one MFMA fragment
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p1));
    union {
        s16x4 h[2];
        el8 v;
    } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
}

// Workgroup id -> (layer, partition, slab): ids 16 g + r, r < 8 (slab 0) and 16 g + 8 + r (slab 1) are the two slabs of pair 8 g + r
// and land on the same XCD (id mod 8); pair = layer * parts + partition.  One layer with MAPF_ENC_WGRAD_PARTS partitions is
// mapf_encoder_wgrad; `layers` layers whose gz / input arrays lie `gz_stride` / `in_stride` elements apart, with parts = 128 / layers,
// is mapf_encoder_wgrad_multi: ONE round of workgroups (<= 256, one per CU -- the kernel holds 104 KB of LDS) does the six 3x3 layers of
// the encoder, and each workgroup writes its 295 KB partial slab once per launch instead of once per layer (at the few thousand
// observations of a few-agent update that write, 75 MB per layer with 128 partitions, was most of a 54 us launch).
__global__ void __launch_bounds__(NTHR) encoder_wgrad_kernel(const uint16_t *__restrict__ gz, const uint16_t *__restrict__ ain, long long M,
                                                           float *__restrict__ ws, const uint32_t *__restrict__ grad_scale, int parts,
                                                           int layers, long long gz_stride, long long in_stride,
                                                           const int32_t *__restrict__ valid_rows) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = blockIdx.x;
    const int slab = (id >> 3) & 1, pair = (id >> 4) * 8 + (id & 7);
    const int layer = pair / parts, part = pair - layer * parts;
    if (layer >= layers) return;  // (the grid is rounded up to whole groups of 16)
    const int chalf = w & 1, nq = w >> 1;  // this wave's 64 output channels / 16 input channels of the slab
    gz += layer * gz_stride;
    ain += layer * in_stride;
    if (valid_rows != nullptr) {  // only the first *valid_rows <= M observations carry a gradient: they are what is partitioned
        const long long cnt = (long long)*valid_rows;
        M = cnt < M ? (cnt < 0 ? 0 : cnt) : M;
    }

    // observations of this partition -> a stream of IR image rows in nblk blocks of 8
    const long long per = (M + parts - 1) / parts;
    const long long ob0 = per * part;
    long long nobl = M - ob0;
    nobl = nobl < 0 ? 0 : (nobl > per ? per : nobl);
    const int IR = 7 * (int)nobl, nblk = (IR + 7) >> 3;
    const uint16_t *gpart = gz + ob0 * 6272, *apart = ain + ob0 * 6272 + 64 * slab;

    for (int i = tid; i < LDS_BYTES / 16; i += NTHR) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);

    // ---- staging ----
    // gz chunk c = w + 8 k (k = 0, 1) = slot rows 4 c .. 4 c + 3 of the block = image row c >> 1, x' = 4 (c & 1) + (lane >> 4);
    // lane l holds 16-byte column (l & 15) ^ (2 x') of its row (x' = row & 7: the bank swizzle); x' = 7 is the padding slot
    const int g_xp = 4 * (w & 1) + (lane >> 4);
    const int g_off = g_xp * 256 + 16 * ((lane & 15) ^ (g_xp << 1));
    // input chunk = one bordered image row, cell Xb = lane >> 3 (0 = border, pixel x = Xb - 1); lane l holds 16-byte column
    // (l & 7) ^ (2 * ((row >> 1) & 3)) of its cell, and (row >> 1) & 3 = (l >> 4) & 3 because chunks start at multiples of 8 rows
    const int i_off = ((lane >> 3) - 1) * 256 + 16 * ((lane & 7) ^ (((lane >> 4) & 3) << 1));
    typedef __attribute__((address_space(3))) unsigned char *lds_byte_ptr;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte_ptr)smem;
    const unsigned long long zsrc = (unsigned long long)(uintptr_t)g_wgrad_zero;
    // load k of this wave for block jb: k = 0, 1 gz chunks w, w + 8; k = 2 image row 8 jb + 1 + w of the input (-1: row 0, prologue)
    auto issue_one = [&](int jb, int buf, int k) __attribute__((always_inline)) {
        if (k < 2) {
            const int c = w + 8 * k, Ir = 8 * jb + (c >> 1), o = Ir / 7, y = Ir - 7 * o;  // wave-uniform
            const uint32_t dst = lds0 + buf * GZ_BLK + 1024 * c;
            if (Ir < IR) {
                const uint16_t *src = gpart + (49 * o + 7 * y) * 128;
                if (g_xp < 7) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(g_off), "s"(src) : "memory");
            } else {  // behind the end of the stream: zeros (also over the padding rows, which are zero anyway)
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(zsrc) : "memory");
            }
        } else {
            const int Ir = k == 2 ? 8 * jb + 1 + w : 0, o = Ir / 7, y = Ir - 7 * o;  // wave-uniform
            const uint32_t dst = lds0 + IN_BASE + 1024 * (8 * (o % RING_OBS) + y + 1);
            if (Ir < IR) {
                const uint16_t *src = apart + (49 * o + 7 * y) * 128;
                if (lane >= 8) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(i_off), "s"(src) : "memory");
            } else {  // nobody multiplies these rows by anything but zero gz rows; the load keeps every wave's count equal
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(zsrc) : "memory");
            }
        }
    };
    auto wait_all_but_newest = [&]() __attribute__((always_inline)) {
        static_assert(LPW == 3, "the count below is the loads per wave and block");
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    };

    // ---- fragment addresses of this lane (ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies row q, columns 4p..) ----
    const int li = lane & 15, lh = lane >> 4, q4 = li >> 2, p4 = li & 3;
    // K slot (lh, j) of a 32-slot k-step: slot 4 lh + j for j < 4 (first read, "blk 0"), 16 + 4 lh + (j - 4) for the second;
    // slot s of a k-step = image row 4 ks + (s >> 3) of the block, x' = s & 7: this lane has x' = (4 lh + q4) & 7 in both reads
    // and image rows 4 ks + 2 blk + (lh >> 1)
    const int xp = (4 * lh + q4) & 7;
    // gz: row = slot of the block; the 32 bytes of co tile ct sit in column group ct ^ (row & 7), and row & 7 = x'
    int a_addr[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) a_addr[c] = (4 * lh + q4) * 256 + (((CT * chalf + c) ^ xp) << 5) + 8 * p4;
    // input: tap (ky,kx) of slot (image row, x') is bordered cell row 8 * (chunk of the row above) + x' + kx + 8 ky; the 32 bytes of
    // this wave's 16 channels sit in column group nq ^ ((row >> 1) & 3) = nq ^ (((x' + kx) & 7) >> 1): per lane and kx
    const int b_kx0 = IN_BASE + (xp + 0) * 128 + ((nq ^ (((xp + 0) & 7) >> 1)) << 5) + 8 * p4;
    const int b_kx1 = IN_BASE + (xp + 1) * 128 + ((nq ^ (((xp + 1) & 7) >> 1)) << 5) + 8 * p4;
    const int b_kx2 = IN_BASE + (xp + 2) * 128 + ((nq ^ (((xp + 2) & 7) >> 1)) << 5) + 8 * p4;
    // scalar walk over the image rows: (observation mod 7, y) of the first image row of the NEXT k-step to be addressed
    int s_om = 0, s_y = 0;
    struct KAddr {  // byte offsets of the taps' cell rows (ky = 0) of one k-step: blk 0 / 1, kx 0..2 (named members: an array
        int a0, a1, a2, b0, b1, b2;  // handed through a lambda lands in LDS)
    };
    const bool upper = (lh >> 1) != 0;
    auto next_chunk = [&]() __attribute__((always_inline)) {  // chunk (bordered image row) ABOVE the walk's image row: 8 om + y
        const int c = 8 * s_om + s_y;
        const bool wrap = s_y == 6;
        s_y = wrap ? 0 : s_y + 1;
        s_om = wrap ? (s_om == RING_OBS - 1 ? 0 : s_om + 1) : s_om;
        return c;
    };
    auto next_addresses = [&]() __attribute__((always_inline)) {
        const int c0 = next_chunk(), c1 = next_chunk(), c2 = next_chunk(), c3 = next_chunk();
        const int base0 = (upper ? c1 : c0) << 10, base1 = (upper ? c3 : c2) << 10;  // 8 cells of 128 B per chunk
        return KAddr{base0 + b_kx0, base0 + b_kx1, base0 + b_kx2, base1 + b_kx0, base1 + b_kx1, base1 + b_kx2};
    };
    KAddr bc, bn;  // current / next k-step

    f32x4 acc[CT][NTAP];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();  // zero fill done
    if (!(MAPF_WGRAD_ABLATE & 1) && nblk > 0) {
        issue_one(0, 0, 3);  // image row 0 (every wave loads it: same data, equal load counts)
        for (int jb = 0; jb < 3 && jb < nblk; ++jb)
            for (int k = 0; k < LPW; ++k) issue_one(jb, jb, k);
        if (nblk > 2)
            wait_all_but_newest();  // blocks 0 and 1 have landed
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // fragments: the CT gz (A) tiles of the current k-step and of the next one; the input (B) tiles pass through a ring of RB
    el8 af[2][CT], br[RB];
    auto read_a = [&](const unsigned char *gbuf, int ks, int c) {
        return tr_read2(gbuf + a_addr[c] + (32 * ks) * 256, gbuf + a_addr[c] + (32 * ks + 16) * 256);
    };
    auto read_b = [&](const KAddr &q, int t) {
        const int q0 = t % 3 == 0 ? q.a0 : (t % 3 == 1 ? q.a1 : q.a2), q1 = t % 3 == 0 ? q.b0 : (t % 3 == 1 ? q.b1 : q.b2);
        return tr_read2(smem + q0 + (8 * (t / 3)) * 128, smem + q1 + (8 * (t / 3)) * 128);
    };
    bc = next_addresses();
    if (nblk > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) af[0][c] = read_a(smem, 0, c);
#pragma unroll
        for (int t = 0; t < RB - 1; ++t) br[t] = read_b(bc, t);
    }

    // One block (step j): 2 k-steps x NTAP x CT MFMAs on gz buffer `b`; the fragments of its second k-step, then of the next
    // block's first k-step (buffer b1, complete since the last barrier), are read behind the MFMAs.
    // Staging: the loads of block j + 3 go into the gz buffer that step j - 1 read (free since its barrier) and into the
    // input ring (image rows at least 17 behind the oldest one block j still reads); the step ends by waiting for block
    // j + 2 (issued one step earlier) and a barrier.
    auto one_blk = [&](int j, auto B, auto B1, auto B3) __attribute__((always_inline)) {
        constexpr int b = decltype(B)::value, b1 = decltype(B1)::value, b3 = decltype(B3)::value;
        const unsigned char *ga = smem + b * GZ_BLK, *ga1 = smem + b1 * GZ_BLK;
        const bool more = j + 3 < nblk && !(MAPF_WGRAD_ABLATE & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int cur = ks, nxt = ks ^ 1;
            // the k-step after this one: (this block, ks 1) or (next block, ks 0; past the last block ga1 holds stale
            // but valid LDS: harmless)
            const unsigned char *nbuf = ks == 0 ? ga : ga1;
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int g = ks * NTAP + t, gr = b * 2 * NTAP + g;  // tile-step of the block / of the 4-block turn (72: a multiple of RB)
                if (t == 1) bn = next_addresses();
                if (more && g % 5 == 1 && g / 5 < LPW) issue_one(j + 3, b3, g / 5);  // tile-steps 1, 6, 11
                if (!(MAPF_WGRAD_ABLATE & 2)) {
                    if (t + RB - 1 < NTAP)
                        br[(gr + RB - 1) % RB] = read_b(bc, t + RB - 1);
                    else
                        br[(gr + RB - 1) % RB] = read_b(bn, t + RB - 1 - NTAP);
                    if (t < CT) af[nxt][t] = read_a(nbuf, nxt, t);
                }
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    // tied destination: hipcc does not tie the builtin's, and shuffles 4 registers per MFMA on the loop back-edge
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[c][t]) : "v"(af[cur][c]), "v"(br[gr % RB]));
                __builtin_amdgcn_sched_barrier(0);
            }
            bc = bn;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more)
            wait_all_but_newest();
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(MAPF_WGRAD_ABLATE & 4)) __syncthreads();
    };
    // Four blocks per loop iteration (one turn of the gz ring): every gz buffer offset is a constant, and whatever hipcc
    // emits on the loop back-edge is paid once per 288 MFMAs.
    int j = 0;
    for (; j + 4 <= nblk; j += 4) {
        one_blk(j, I<0>{}, I<1>{}, I<3>{});
        one_blk(j + 1, I<1>{}, I<2>{}, I<0>{});
        one_blk(j + 2, I<2>{}, I<3>{}, I<1>{});
        one_blk(j + 3, I<3>{}, I<0>{}, I<2>{});
    }
    // tail (j % 4 == 0 here, so the ring is in its initial phase)
    if (j < nblk) one_blk(j, I<0>{}, I<1>{}, I<3>{});
    if (j + 1 < nblk) one_blk(j + 1, I<1>{}, I<2>{}, I<0>{});
    if (j + 2 < nblk) one_blk(j + 2, I<2>{}, I<3>{}, I<1>{});

    // ---- partial sums of this partition: ws[part][co][tap = ky*3 + kx][ci] ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the asm MFMAs are opaque to hipcc's hazard padding: let the last ones retire
    // (gz carries the backward chain's loss scale: taken out here, in fp32)
    const float inv_scale = grad_scale ? __uint_as_float(grad_scale[1]) : 1.f;
    float *out = ws + (long long)pair * (128 * 9 * 128);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int ci = 64 * slab + 16 * nq + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * (CT * chalf + c) + 4 * lh + r;
                out[(co * 9 + t) * 128 + ci] = acc[c][t][r] * inv_scale;
            }
        }
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            std::fprintf(stderr, "mapf_wgrad: %s failed: %s\n", #expr, hipGetErrorString(_e)); \
            return MAPF_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

}  // namespace

extern "C" {

int mapf_encoder_wgrad(const uint16_t *gz_dev, const uint16_t *in_dev, int64_t M, const uint32_t *grad_scale_dev, float *partial_dev,
                       void *stream) {
    if (M < 0 || !partial_dev || (M > 0 && (!gz_dev || !in_dev))) return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(gz_dev) & 15) || (reinterpret_cast<uintptr_t>(in_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(partial_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    if ((M + MAPF_ENC_WGRAD_PARTS - 1) / MAPF_ENC_WGRAD_PARTS > (1 << 18)) return MAPF_ERR_INVALID_ARG;  // int element offsets inside a partition
    // every partition writes its slab (zeros when it has no observations), so the caller's sum is always defined
    hipLaunchKernelGGL(encoder_wgrad_kernel, dim3(SLABS * MAPF_ENC_WGRAD_PARTS), dim3(NTHR), 0, static_cast<hipStream_t>(stream), gz_dev,
                       in_dev, (long long)M, partial_dev, grad_scale_dev, MAPF_ENC_WGRAD_PARTS, 1, 0LL, 0LL, (const int32_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

int mapf_encoder_wgrad_multi(const uint16_t *gz_dev, int64_t gz_layer_stride, const uint16_t *in_dev, int64_t in_layer_stride, int layers,
                             int parts, int64_t M, const int32_t *valid_rows_dev, const uint32_t *grad_scale_dev, float *partial_dev,
                             void *stream) {
    if (reinterpret_cast<uintptr_t>(valid_rows_dev) & 3) return MAPF_ERR_INVALID_ARG;
    if (M < 0 || !partial_dev || layers < 1 || layers > 8 || parts < 1 || parts > MAPF_ENC_WGRAD_PARTS || (M > 0 && (!gz_dev || !in_dev)))
        return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(gz_dev) & 15) || (reinterpret_cast<uintptr_t>(in_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(partial_dev) & 15) || (gz_layer_stride & 7) || (in_layer_stride & 7))
        return MAPF_ERR_INVALID_ARG;
    if (layers > 1 && (gz_layer_stride < M * 6272 || in_layer_stride < M * 6272)) return MAPF_ERR_INVALID_ARG;  // layers may not overlap
    if ((M + parts - 1) / parts > (1 << 18)) return MAPF_ERR_INVALID_ARG;  // int element offsets inside a partition
    const int pairs = layers * parts, groups = (pairs + 7) / 8;
    // every (layer, partition) writes its slab (zeros when it has no observations), so the caller's sum is always defined
    hipLaunchKernelGGL(encoder_wgrad_kernel, dim3(16 * groups), dim3(NTHR), 0, static_cast<hipStream_t>(stream), gz_dev, in_dev,
                       (long long)M, partial_dev, grad_scale_dev, parts, layers, (long long)gz_layer_stride, (long long)in_layer_stride,
                       valid_rows_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

}  // extern "C"
