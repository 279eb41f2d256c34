// mapf_wgrad.hip -- weight gradient of the encoder's 3x3 128->128 convolutions (reference model.py:30-42, the
// backward of ResBlock.block1/block2 that autograd derives for `Learner.train`, worker.py:316) as one streaming
// MFMA kernel per layer (see include/mapf_dqn.h: mapf_encoder_wgrad).
//
//     dW[co][ky][kx][ci] = sum over observations m and positions (y,x) of
//                          gz[m][y][x][co] * a[m][y+ky-1][x+kx-1][ci]            (zero outside the 7x7 image)
//
// gz = ReLU-masked pre-activation gradient of the layer (mapf_encoder_backward_data), a = the layer's input
// (mapf_encoder_forward_save), both bf16 [M][49][128].  As a GEMM: 128 (co) x 1152 (tap, ci) outputs, K = all
// positions of all observations -- tiny output, enormous K, so the output is held in registers and the
// operands stream through LDS:
//  * A workgroup (512 threads = 8 waves, two per SIMD) owns one of TWO slabs of the output -- all 128 co x 9 taps x
//    64 of the input channels -- for a partition of the observations; wave w holds 64 co x (9 taps x 16 ci) = 4 x 9
//    tiles of 16x16 (144 accumulator registers).  Splitting by INPUT CHANNEL means a workgroup stages only half of
//    every input row.
//  * K IS DENSE: the partition's positions form one stream s = 49 * obs + 7 y + x (exactly the row order of gz in
//    memory) that is cut into blocks of 64 = two MFMA k-steps, regardless of observation boundaries.  (The first
//    version gave every observation 64 K slots, 49 of them real: 23 % of the MFMAs and of the LDS fragment reads
//    multiplied zeros.)
//  * gz blocks (64 rows of 256 B, contiguous in global memory) go through a ring of four 16-KiB LDS buffers.
//  * The INPUT needs, for stream position s and tap (ky,kx), the row of the zero-bordered 8-wide image
//        beta(s) - 9 + 8 ky + kx,      beta = 64 obs + 8 (y+1) + (x+1)
//    (the same image as in the forward kernel: one image row's right border is the next one's left border, one
//    observation's bottom border the next one's top border).  These rows live in ONE circular buffer of 448 rows
//    (7 observations): bordered row rb sits at LDS row rb mod 448, so border rows are always the same LDS rows, zeroed
//    once and never overwritten with anything but zeros.  Every block tops the buffer up with a window of 16 chunks of 8
//    rows starting at the chunk that holds the first bordered row its positions need and nobody loaded yet (a block
//    needs at most 90 new rows; three blocks + a window stay below 448 rows, so nothing that is still needed is
//    overwritten: tools/micro/wgrad_ring_check.py replays the schedule).
//  * Both operands need K (= position) along the fragment's register axis while memory has channels contiguous:
//    ds_read_b64_tr_b16 (hardware transpose read) delivers a [4 positions x 16 channels] block column-major and takes
//    a row address per lane -- so a k-step's 32 positions may sit in any rows.  A lane keeps the ring position
//    (49 * slot + 7 y + x) of its K slots, advances it by 32 per k-step and looks the row addresses up in a small LDS table.
//  * No row padding: rows are 256 B (gz) / 128 B (input) and bank conflicts are avoided by XOR-swizzling the 32-byte
//    column groups with the row number -- free on the way in, because an LDS-DMA lane may fetch ANY 16 global bytes
//    for its fixed LDS slot, and one v_xad_u32 per fragment address on the way out.  A load instruction is then a
//    whole number of rows, every lane of every load is active (lanes without data of this partition -- border rows,
//    rows behind the partition's end -- read 16 zero bytes), and all waves issue the same loads per block: 2 gz
//    chunks + 2 input chunks.  Earlier versions masked lanes and gave waves different numbers of loads; the exec
//    juggling and the wave-dependent scalar branches around every load cost more issue time than the MFMAs left over.
//  * Staging is global_load_lds_dwordx4 (HBM -> LDS without passing through registers); while block j is multiplied, j+1
//    and j+2 are resident and j+3 is in flight.  The loads are inline asm (hipcc's wait-count pass would make every later
//    LDS read wait for a builtin LDS-DMA load it cannot disambiguate), spread over the block's tile-steps (issued together
//    behind the barrier they queue up in the CU's one vector-memory pipeline), and counted by hand:
//    `s_waitcnt vmcnt(4)` at the end of a block means "all but the newest block have landed".
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

__device__ __attribute__((aligned(16))) unsigned int g_wgrad_zero[4];  // what lanes without data of the partition read

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
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
constexpr int RING = 448;                    // input ring: bordered rows (7 observations)
constexpr int RING_OBS = RING / 64;
constexpr int CH_ROWS = 8;                   // input load chunk: 8 ring rows of 128 B
constexpr int RING_CHUNKS = RING / CH_ROWS;  // 56
constexpr int WIN_CHUNKS = 16;             // input window of a block, in chunks
constexpr int GZ_CHUNKS = 16;              // 64 rows of 256 B in 1-KiB chunks
constexpr int NBUF = 4;                      // gz ring: computing block j, j+1 and j+2 resident, j+3 in flight
constexpr int GZ_BLK = 64 * 256;
constexpr int IN_BASE = NBUF * GZ_BLK;       // 65,536: gz buffers first, so that their fragment reads fit the 16-bit DS offset
constexpr int IN_BYTES = (RING + 9) * 128;   // + the 9 rows the taps of the last ring rows reach: a top border, always zero
constexpr int TAB_BASE = IN_BASE + ((IN_BYTES + 1023) / 1024) * 1024;
constexpr int TAB_N = RING_OBS * 49;         // u16 [343][4]: swizzled byte offset of the top-left tap row + kx of ring position u, kx = 0..2
constexpr int LTAB_BASE = TAB_BASE + ((TAB_N * 8 + 63) / 64) * 64;  // u16 [56 chunks][8 rows]: ring position of a ring row (0xFFFF: border)
constexpr int LDS_BYTES = LTAB_BASE + RING_CHUNKS * CH_ROWS * 2;
constexpr int RB = 4;                        // ring of input (B) fragments: tile t's MFMAs run while tile t + RB - 1 is being read
constexpr int LPW = (GZ_CHUNKS + WIN_CHUNKS) / NW, GPW = GZ_CHUNKS / NW;  // loads per wave and block (4); the first GPW (2) are gz
static_assert(RING % 64 == 0 && (2 * NTAP * NBUF) % RB == 0 && LDS_BYTES <= 160 * 1024 && (RING + 11) * 128 < 65536, "");
static_assert(MAPF_ENC_WGRAD_PARTS % 8 == 0, "");

__device__ __forceinline__ bf16x8 tr_read2(const unsigned char *p0, const unsigned char *p1) {
    // two transposed 4-row blocks -> the 8 k-elements of one MFMA fragment
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p1));
    union {
        s16x4 h[2];
        bf16x8 v;
    } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
}

// first bordered row that the blocks before block jb have not asked for: beta(64 jb - 1) + 10
__device__ __forceinline__ int frontier(int jb) {
    if (jb == 0) return 0;
    const int s = 64 * jb - 1, o = s / 49, p = s - 49 * o, y = p / 7;
    return 64 * o + p + y + 19;
}

__global__ void __launch_bounds__(NTHR) encoder_wgrad_kernel(const uint16_t *__restrict__ gz, const uint16_t *__restrict__ ain, long long M,
                                                           float *__restrict__ ws) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int slab = slot % SLABS, part = (slot / SLABS) * 8 + xcd;
    const int chalf = w & 1, nq = w >> 1;  // this wave's 64 output channels / 16 input channels of the slab

    // observations of this partition -> a stream of S positions in nblk blocks of 64
    const long long per = (M + MAPF_ENC_WGRAD_PARTS - 1) / MAPF_ENC_WGRAD_PARTS;
    const long long ob0 = per * part;
    long long nobl = M - ob0;
    nobl = nobl < 0 ? 0 : (nobl > per ? per : nobl);
    const int nob = (int)nobl, S = 49 * nob, nblk = (S + 63) >> 6;
    const uint16_t *gpart = gz + ob0 * 6272, *apart = ain + ob0 * 6272 + 64 * slab;

    for (int i = tid; i < TAB_BASE / 16; i += NTHR) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < TAB_N * 4; i += NTHR) {  // fragment table: ring position u, kx -> (row * 128) | (swizzle * 32), row = 64 o + 8 y + x + kx
        const int u = i >> 2, kx = i & 3, o = u / 49, p = u - 49 * o, row = 64 * o + p + p / 7 + kx;
        reinterpret_cast<uint16_t *>(smem + TAB_BASE)[i] = (uint16_t)(kx < 3 ? row * 128 + ((row >> 1) & 3) * 32 : 0);
    }
    for (int i = tid; i < RING_CHUNKS * CH_ROWS; i += NTHR) {  // load table: ring row -> ring position 49 * slot + 7 y + x, or 0xFFFF for a border row
        const int rr = (i & 63) - 9;
        reinterpret_cast<uint16_t *>(smem + LTAB_BASE)[i] = (uint16_t)((rr >= 0 && rr < 56 && (rr & 7) != 7) ? 49 * (i >> 6) + 7 * (rr >> 3) + (rr & 7) : 0xFFFF);
    }

    // ---- staging: every wave issues LPW chunk-loads per block: gz chunks w + NW k (k < GPW), then input window chunks w + NW (k - GPW) ----
    // gz chunk = 4 rows of 256 B: lane l holds 16-byte column (l & 15) ^ (2 * (row & 7)) of row 4 chunk + (l >> 4)
    int goff[GPW];
#pragma unroll
    for (int k = 0; k < GPW; ++k) {
        const int row = 4 * (w + NW * k) + (lane >> 4);
        goff[k] = row * 256 + 16 * ((lane & 15) ^ ((row & 7) << 1));
    }
    // input chunk = 8 rows of 128 B: lane l holds 16-byte column (l & 7) ^ (2 * ((row >> 1) & 3)) of chunk row l >> 3, and
    // (row >> 1) & 3 = (l >> 4) & 3 because chunks start at multiples of 8 rows
    const int in_col = 16 * ((lane & 7) ^ (((lane >> 4) & 3) << 1)), in_row = lane >> 3;
    typedef __attribute__((address_space(3))) unsigned char *lds_byte_ptr;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte_ptr)smem;
    const unsigned long long zsrc = (unsigned long long)(uintptr_t)g_wgrad_zero, abase = (unsigned long long)(uintptr_t)apart;
    struct LoadCtx {  // wave-uniform description of one block's loads
        const uint16_t *gsrc;
        int rows_left, c0, posA, thr;
    };
    auto load_ctx = [&](int jb) __attribute__((always_inline)) {
        // input window: 16 chunks of 8 ring rows from the chunk that holds bordered row F.  A ring row's position q (table)
        // is stream position q + posA, + 343 if its ring slot is before the slot of the window's first row
        const int F = frontier(jb), Fm = F % RING, c0 = Fm / CH_ROWS, rb0 = F - (Fm - c0 * CH_ROWS), ow = rb0 >> 6, omw = ow % RING_OBS;
        return LoadCtx{gpart + (long long)jb * (64 * 128), S - 64 * jb, c0, 49 * (ow - omw), 49 * omw};
    };
    int qv[LPW - GPW];  // ring positions of this lane's rows in the block's input chunks, read from the table well before the loads need them
    auto fetch_q = [&](const LoadCtx &c) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < LPW - GPW; ++k) {
            int ch = c.c0 + w + NW * k;
            ch = ch >= RING_CHUNKS ? ch - RING_CHUNKS : ch;
            qv[k] = reinterpret_cast<const uint16_t *>(smem + LTAB_BASE)[ch * CH_ROWS + in_row];
        }
    };
    auto issue_one = [&](const LoadCtx &c, int buf, int k) __attribute__((always_inline)) {
        if (k < GPW) {
            const uint32_t dst = lds0 + buf * GZ_BLK + 1024 * (w + NW * k);
            if (c.rows_left >= 64) {
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(goff[k]), "s"(c.gsrc) : "memory");
            } else {  // last block: rows behind the end of the stream read zeros
                const int row = 4 * (w + NW * k) + (lane >> 4);
                const unsigned long long src = row < c.rows_left ? (unsigned long long)(uintptr_t)c.gsrc + (unsigned)goff[k] : zsrc;
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
            }
        } else {
            int ch = c.c0 + w + NW * (k - GPW);
            ch = ch >= RING_CHUNKS ? ch - RING_CHUNKS : ch;
            const uint32_t dst = lds0 + IN_BASE + 1024 * ch;
            const int q = qv[k - GPW];
            const int pos = q + c.posA + (q < c.thr ? TAB_N : 0);
            // every lane loads: the ones without data of this partition read 16 zero bytes (border rows are zero anyway)
            const unsigned long long src = (q != 0xFFFF && pos < S) ? abase + (unsigned)(pos * 256 + in_col) : zsrc;
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
        }
    };
    auto issue_loads = [&](int jb, int buf) __attribute__((always_inline)) {
        const LoadCtx c = load_ctx(jb);
        fetch_q(c);
#pragma unroll
        for (int k = 0; k < LPW; ++k) issue_one(c, buf, k);
    };
    auto wait_all_but_newest = [&]() __attribute__((always_inline)) {
        static_assert(LPW == 4, "the count below is the loads per wave and block");
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    };

    // ---- fragment addresses of this lane (ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies row q, columns 4p..) ----
    const int li = lane & 15, lh = lane >> 4, q4 = li >> 2, p4 = li & 3;
    // K slot (lh, j) of a 32-slot k-step: slot 4 lh + j for j < 4 (first read, "blk 0"), 16 + 4 lh + (j - 4) for the second
    // gz: row = slot of the block; the 32 bytes of co tile ct sit in column group ct ^ (row & 7), and row & 7 = (4 lh + q4) & 7
    int a_addr[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) a_addr[c] = (4 * lh + q4) * 256 + (((CT * chalf + c) ^ ((4 * lh + q4) & 7)) << 5) + 8 * p4;
    // input: ring position u = (49 * observation + 7 y + x) mod 343 of this lane's two slots in the NEXT k-step to be addressed;
    // table -> (row << 7) | (swizzle << 5) per kx; this lane's 8 bytes of its wave's 16 channels: ^ ((nq << 5) | (p4 << 3)); tap row
    // ky is the immediate 8 ky * 128 (8 rows further the swizzle is the same)
    int un[2] = {4 * lh + q4, 16 + 4 * lh + q4};
    const unsigned b_lane = (unsigned)((nq << 5) | (p4 << 3));
    unsigned bc[2][3], bn[2][3];  // current / next k-step: [blk][kx]
    auto next_addresses = [&](unsigned (&dst)[2][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const u32x2 e = *reinterpret_cast<const u32x2 *>(smem + TAB_BASE + 8 * un[blk]);
            dst[blk][0] = ((e[0] & 0xFFFFu) ^ b_lane) + (unsigned)IN_BASE;
            dst[blk][1] = ((e[0] >> 16) ^ b_lane) + (unsigned)IN_BASE;
            dst[blk][2] = (e[1] ^ b_lane) + (unsigned)IN_BASE;
            const unsigned u = (unsigned)un[blk] + 32u;  // advance by one k-step (32 positions)
            un[blk] = (int)min(u, u - (unsigned)TAB_N);
        }
    };

    f32x4 acc[CT][NTAP];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();  // zero fill and tables done
    if (!(MAPF_WGRAD_ABLATE & 1)) {
        if (nblk > 0) issue_loads(0, 0);
        if (nblk > 1) issue_loads(1, 1);
        if (nblk > 2) {
            issue_loads(2, 2);
            wait_all_but_newest();  // blocks 0 and 1 have landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();

    // fragments: the CT gz (A) tiles of the current k-step and of the next one; the input (B) tiles pass through a ring of RB
    bf16x8 af[2][CT], br[RB];
    auto read_a = [&](const unsigned char *gbuf, int ks, int c) {
        return tr_read2(gbuf + a_addr[c] + (32 * ks) * 256, gbuf + a_addr[c] + (32 * ks + 16) * 256);
    };
    auto read_b = [&](const unsigned (&q)[2][3], int t) {
        return tr_read2(smem + q[0][t % 3] + (8 * (t / 3)) * 128, smem + q[1][t % 3] + (8 * (t / 3)) * 128);
    };
    next_addresses(bc);
    if (nblk > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) af[0][c] = read_a(smem, 0, c);
#pragma unroll
        for (int t = 0; t < RB - 1; ++t) br[t] = read_b(bc, t);
    }

    // One block (step j): 2 k-steps x NTAP x CT MFMAs on gz buffer `b`; the fragments of its second k-step, then of the next
    // block's first k-step (buffer b1, complete since the last barrier), are read behind the MFMAs.
    // Staging: the loads of block j + 3 go into the gz buffer that step j - 1 read (free since its barrier) and into the
    // input ring; the step ends by waiting for block j + 2 (issued one step earlier) and a barrier.
    auto one_blk = [&](int j, auto B, auto B1, auto B3) __attribute__((always_inline)) {
        constexpr int b = decltype(B)::value, b1 = decltype(B1)::value, b3 = decltype(B3)::value;
        const unsigned char *ga = smem + b * GZ_BLK, *ga1 = smem + b1 * GZ_BLK;
        const bool more = j + 3 < nblk && !(MAPF_WGRAD_ABLATE & 1);
        LoadCtx lc = {gpart, 64, 0, 0, 0};
        if (more) {
            lc = load_ctx(j + 3);
            fetch_q(lc);
        }
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
                if (t == 1) next_addresses(bn);
                if (more && g % 4 == 1 && g / 4 < LPW) issue_one(lc, b3, g / 4);  // tile-steps 1, 5, 9, 13
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
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[c][t]) : "v"(af[cur][c]), "v"(br[gr % RB]));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) bc[i / 3][i % 3] = bn[i / 3][i % 3];
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
    float *out = ws + (long long)part * (128 * 9 * 128);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int ci = 64 * slab + 16 * nq + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * (CT * chalf + c) + 4 * lh + r;
                out[(co * 9 + t) * 128 + ci] = acc[c][t][r];
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

int mapf_encoder_wgrad(const uint16_t *gz_dev, const uint16_t *in_dev, int64_t M, float *partial_dev, void *stream) {
    if (M < 0 || !partial_dev || (M > 0 && (!gz_dev || !in_dev))) return MAPF_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(gz_dev) & 15) || (reinterpret_cast<uintptr_t>(in_dev) & 15) ||
        (reinterpret_cast<uintptr_t>(partial_dev) & 15))
        return MAPF_ERR_INVALID_ARG;
    if ((M + MAPF_ENC_WGRAD_PARTS - 1) / MAPF_ENC_WGRAD_PARTS > (1 << 17)) return MAPF_ERR_INVALID_ARG;  // 32-bit byte offsets inside a partition
    // every partition writes its slab (zeros when it has no observations), so the caller's sum is always defined
    hipLaunchKernelGGL(encoder_wgrad_kernel, dim3(SLABS * MAPF_ENC_WGRAD_PARTS), dim3(NTHR), 0, static_cast<hipStream_t>(stream), gz_dev,
                       in_dev, (long long)M, partial_dev);
    HIP_TRY(hipGetLastError());
    return MAPF_OK;
}

}  // extern "C"
