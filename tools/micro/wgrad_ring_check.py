#!/usr/bin/env python3
"""Replays the input-ring schedule of encoder_wgrad_kernel (csrc/mapf_wgrad.hip) on the host: for partitions of several
sizes, every bordered row a resident block reads through its 9 taps must hold exactly that row (or be a never-written
border row) at the time it is read, although blocks keep topping the 448-row ring up three blocks ahead."""
R, CH_ROWS, WIN = 448, 8, 16


def beta(s):  # bordered row of stream position s; its top-left tap row is beta - 9
    o, p = divmod(s, 49)
    return 64 * o + p + p // 7 + 9


def frontier(j):  # first bordered row the blocks before j have not asked for
    return 0 if j == 0 else beta(64 * j - 1) + 10


def interior(rb, nob):
    o, rr = rb >> 6, (rb & 63) - 9
    return rb >= 0 and o < nob and 0 <= rr < 56 and (rr & 7) != 7


def window(j):
    F = frontier(j)
    Fm = F % R
    c0 = Fm // CH_ROWS
    row_s = c0 * CH_ROWS
    rb0 = F - (Fm - row_s)
    for i in range(WIN):
        ch = (c0 + i) % (R // CH_ROWS)
        for lr in range(CH_ROWS):
            lrow = ch * CH_ROWS + lr
            yield rb0 + (lrow - row_s) % R, lrow


def check(nob):
    S = 49 * nob
    nblk = (S + 63) // 64
    ring = {}

    def load(j):
        for rb, lrow in window(j):
            if interior(rb, nob):
                ring[lrow] = rb

    for j in range(min(3, nblk)):
        load(j)
    for j in range(nblk):
        if j + 3 < nblk:
            load(j + 3)  # issued at the start of block j
        for jj in range(j, min(j + 3, nblk)):
            for s in range(64 * jj, min(64 * jj + 64, S)):
                for tap in range(9):
                    rb = beta(s) - 9 + 8 * (tap // 3) + tap % 3
                    lrow = (beta(s) - 9) % R + 8 * (tap // 3) + tap % 3  # no wrap: rows 448..456 are permanent zeros
                    if interior(rb, nob):
                        assert lrow < R and ring.get(lrow) == rb, (nob, j, jj, s, tap)
                    else:
                        assert lrow >= R or lrow not in ring, (nob, j, s, tap)  # a border row: never written, still zero
    for lrow in ring:  # loads only ever touch interior ring rows
        rr = (lrow & 63) - 9
        assert 0 <= rr < 56 and (rr & 7) != 7
    return nblk


if __name__ == "__main__":
    for nob in (1, 2, 3, 5, 6, 7, 13, 40, 162, 1080):
        print("observations per partition %5d: %4d blocks ok" % (nob, check(nob)))
