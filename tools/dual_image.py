"""CPU model of the dual-use LDS image of csrc/fa_bwd_w64.hpp (`DualImg`): one swizzled row-major copy of a [rows][E] 16-bit
tile that serves BOTH MFMA operand reads of the backward kernels --

  * the row read  (ds_read_b128):        lane (r, h) takes embedding elements 16 ks + 8 h + (0..7) of tile row 32 zb + r
  * the column read (ds_read_b64_tr_b16): lane (r, h) takes, for embedding column 32 eb + r, the tile rows
                                          16 kk + 8 (j >> 2) + 4 h + (j & 3), j = 0..7 (two transposed reads s = 0, 1)

filled by LDS-DMA (lane-linear destination: the swizzle is applied to each lane's SOURCE offset).  This file restates the
address arithmetic of the kernel (same formulas, same names) and checks on the CPU, for E = 64 and 128,

  1. that every read delivers the element the MFMA operand map asks for, and
  2. that every read is bank-conflict free under the LDS rules of MI355X_MICROARCH.md (64 banks of 4 bytes; ds_read_b128 in
     four fixed 16-lane groups, ds_read_b64_tr_b16 per 32-lane half).

Run: python tools/dual_image.py        (also imported by tests/test_dual_image.py)
"""
import numpy as np

B128_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]


def row_bytes(E):
    return 2 * E


def xor_of(E, row):
    """chunk swizzle of tile row `row` (DualImg::xor_of)"""
    if E >= 128:                                   # 256-byte rows and longer: the low 4 chunk bits
        return ((row & 3) << 2) | ((row >> 2) & 3)
    if E == 64:
        return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1)
    raise ValueError(E)


def off(E, row, ch):
    return row * row_bytes(E) + ((ch ^ xor_of(E, row)) << 4)


def dma_source(E, o):
    """source byte (inside the dense [rows][E] tile) of the 16 bytes that land at image byte `o` (DualImg::src_of)"""
    rb = row_bytes(E)
    row, phys = o // rb, (o % rb) >> 4
    return row * rb + ((phys ^ xor_of(E, row)) << 4)


def fill_image(E, rows):
    """image as an array of element ids (id = row * E + e), filled the way the DMA fills it"""
    img = np.full(rows * E, -1, dtype=np.int64)
    for o in range(0, rows * row_bytes(E), 16):
        s = dma_source(E, o)
        for b in range(8):
            img[o // 2 + b] = s // 2 + b
    return img


# ---- the kernel's read addresses -------------------------------------------------------------------------------------
def row_lane_base(E, lane):
    r, h = lane & 31, lane >> 5
    return r * row_bytes(E) + ((xor_of(E, r) ^ h) << 4)


def row_read_addr(E, lane, zb, ks):
    """(A ^ (ks << 5)) + zb * 32 * row bytes,  A = row_lane_base"""
    return (row_lane_base(E, lane) ^ (ks << 5)) + zb * 32 * row_bytes(E)


def col_lane_base(E, lane):
    h, g1, q, p = lane >> 5, (lane >> 4) & 1, (lane >> 2) & 3, lane & 3
    rb = row_bytes(E)
    qx = q if E >= 128 else (q >> 1)
    return (4 * h + q) * rb + 16 * (4 * qx + ((2 * g1 + (p >> 1)) ^ h)) + 8 * (p & 1)


def col_read_addr(E, lane, kk, eb, s):
    """(B ^ (eb << 6) ^ (s << 5)) + (16 kk + 8 s) * row bytes,  B = col_lane_base"""
    return (col_lane_base(E, lane) ^ (eb << 6) ^ (s << 5)) + (16 * kk + 8 * s) * row_bytes(E)


# ---- bank model ------------------------------------------------------------------------------------------------------
def conflicts(addrs, nbytes, groups):
    """extra LDS cycles of one wave-instruction: per lane group, max over banks of (distinct dwords on the bank) - 1"""
    extra = 0
    for g in groups:
        per_bank = {}
        for lane in g:
            for d in range(nbytes // 4):
                dw = addrs[lane] // 4 + d
                per_bank.setdefault(dw % 64, set()).add(dw)
        extra += max(len(v) for v in per_bank.values()) - 1
    return extra


def check(E, rows=64):
    img = fill_image(E, rows)
    assert (img >= 0).all()
    KS, EB = E // 16, E // 32
    worst = {"row": 0, "col": 0}
    # row reads
    for zb in range(rows // 32):
        for ks in range(KS):
            addrs = [row_read_addr(E, l, zb, ks) for l in range(64)]
            for l in range(64):
                r, h = l & 31, l >> 5
                assert addrs[l] % 16 == 0
                got = img[addrs[l] // 2: addrs[l] // 2 + 8]
                want = (32 * zb + r) * E + 16 * ks + 8 * h + np.arange(8)
                assert (got == want).all(), ("row", E, zb, ks, l)
            worst["row"] = max(worst["row"], conflicts(addrs, 16, B128_GROUPS))
    # column (transposed) reads: the hardware gathers per 16-lane group; lane 4q+p supplies row q, columns 4p..4p+3 of the block,
    # lane i receives column i of the four rows
    for kk in range(rows // 16):
        for eb in range(EB):
            for s in range(2):
                addrs = [col_read_addr(E, l, kk, eb, s) for l in range(64)]
                for grp in range(4):
                    lanes = list(range(16 * grp, 16 * grp + 16))
                    h, g1 = grp >> 1, grp & 1
                    block = np.zeros((4, 16), dtype=np.int64)
                    for l in lanes:
                        q, p = (l >> 2) & 3, l & 3
                        assert addrs[l] % 8 == 0
                        block[q, 4 * p: 4 * p + 4] = img[addrs[l] // 2: addrs[l] // 2 + 4]
                    for i, l in enumerate(lanes):
                        got = block[:, i]
                        e = 32 * eb + (l & 31)
                        rws = 16 * kk + 8 * s + 4 * h + np.arange(4)
                        assert (got == rws * E + e).all(), ("col", E, kk, eb, s, l)
                worst["col"] = max(worst["col"], conflicts(addrs, 8, [list(range(32)), list(range(32, 64))]))
    return worst


def col_conflicts_with(E, xor_fn, rows=64):
    """extra LDS cycles of the transposed column read if the image used `xor_fn(row)` as its chunk swizzle instead (addresses from
    the operand map directly: lane 4q+p of a 16-lane group asks for row 16 kk + 8 s + 4 h + q, columns 32 eb + 16 g1 + 4 p ..)"""
    worst = 0
    for kk in range(rows // 16):
        for eb in range(E // 32):
            for s in range(2):
                addrs = []
                for l in range(64):
                    h, g1, q, p = l >> 5, (l >> 4) & 1, (l >> 2) & 3, l & 3
                    row, ch = 16 * kk + 8 * s + 4 * h + q, 4 * eb + 2 * g1 + (p >> 1)
                    addrs.append(row * row_bytes(E) + ((ch ^ xor_fn(row)) << 4) + 8 * (p & 1))
                worst = max(worst, conflicts(addrs, 8, [list(range(32)), list(range(32, 64))]))
    return worst


if __name__ == "__main__":
    for E in (64, 128, 256):
        print(E, check(E))
