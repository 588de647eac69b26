#!/usr/bin/env python3
"""Model of the refinement kernel's tree-shaped accumulation (me_kernels.hpp me_frac_tree_add): which lane of a wave adds which partial sum
to which slot.  The kernel's closed forms are restated here and checked against the slot table (hmme_slot_rect, i.e. the reference's
getIndexBlock layout): every slot must be tiled exactly once by the pieces the four waves add to it.  tests/test_tree_sim.py runs this.

A workgroup of 256 lanes: lane tid holds quadrant `role` = tid & 3 (rx = role & 1, ry = role >> 1) of the 8x8 position (x, y) = ((tid >> 2) & 7,
tid >> 5); wave w = tid >> 6 holds the two position rows y = 2w, 2w + 1.  Lane bits inside the wave: 0 rx, 1 ry, 2 x0, 3 x1, 4 x2, 5 y0.

Kind-8 slots (multiples of 8: sets of 8x8 Hadamard blocks) take the quad's 8x8 value v; the sums over aligned groups of positions are
    A = v             one position          B = A + A^x0      16x8           C = A + A^y0     8x16          D = B + B^y0   16x16
    E = B + B^x1      32x8 (a row half)     F = D + D^x1      32x16          G = F + F^x2     64x16 (the wave's whole strip)
Kind-4 slots (the AMP shapes at 16, 8x4, 4x8: sets of 4x4 blocks) take the lane's own 4x4 value u:
    a = u + u^rx      8x4      b = u + u^ry    4x8      c = a + a^x0   16x4      d = b + b^y0   4x16      e = c + c^ry   16x8      f = d + d^rx   8x16
`duties(tid)` lists (sum name, slot) for the lane; a sum is the same in every lane of its group, ONE designated lane of the group adds it.
"""
AMP_ROW = {0: 0, 1: 3, 2: 2, 3: 1}    # row (of four) of a CU -> the AMP part that is exactly / starts with that single row: 2NxnU.p0, 2NxnU.p1, 2NxnD.p0, 2NxnD.p1
AMP_COL = {0: 4, 1: 7, 2: 6, 3: 5}    # column (of four) of a CU -> likewise: nLx2N.p0, nLx2N.p1, nRx2N.p0, nRx2N.p1


def duties(tid):
    role, p8 = tid & 3, tid >> 2
    rx, ry, x, y = role & 1, role >> 1, p8 & 7, p8 >> 3
    w, x0, y0 = y >> 1, x & 1, y & 1
    r16, r32 = (y >> 1) * 4 + (x >> 1), (y >> 2) * 2 + (x >> 2)
    out = []
    # ---- kind 8 ----
    if role == 0:
        out.append(("A", 384 + y * 8 + x))                                     # 8x8 2Nx2N
    if x0 == 0 and role == 1:
        out.append(("B", 448 + (y >> 1) * 8 + (y & 1) * 4 + (x >> 1)))         # 16: 2NxN part y & 1
    if y0 == 0 and role == 2:
        out.append(("C", 480 + (y >> 1) * 8 + x))                              # 16: Nx2N part x & 1
    if y0 == 0 and role == 3:
        out.append(("C", 512 + AMP_COL[x & 3] * 4 + r32))                      # 32: the AMP part this column (of the CU's four) belongs to, alone or as a piece
    if x0 == 1 and y0 == 1:
        cx = x >> 1
        out.append(("D", [544 + r16,                                           # 16: 2Nx2N
                          568 + (y >> 2) * 4 + 2 * (x >> 2) + (cx & 1),        # 32: Nx2N part
                          512 + (7 if cx & 1 else 6) * 4 + r32,                # 32: the two-column piece of nLx2N.p1 / nRx2N.p0
                          576 + AMP_COL[cx]][role]))                           # 64: the AMP part this 16-column belongs to
    if (x & 3) == 1 and role == 1:
        out.append(("E", 512 + AMP_ROW[y & 3] * 4 + r32))                      # 32: the AMP part this position row belongs to, alone or as a piece
    if (x & 3) == 2 and y0 == 0:
        out.append(("F", [584 + r32,                                           # 32: 2Nx2N
                          560 + (y >> 2) * 4 + ((y >> 1) & 1) * 2 + (x >> 2),  # 32: 2NxN part
                          512 + (3 if w & 1 else 2) * 4 + r32,                 # 32: the two-row piece of 2NxnU.p1 (lower CU half) / 2NxnD.p0 (upper half)
                          590 + (x >> 2)][role]))                              # 64: Nx2N part
    if (x & 3) == 3 and y0 == 0 and role == 0:
        out.append(("F", 576 + (7 if x >> 2 else 6)))                          # 64: the 32-column piece of nLx2N.p1 / nRx2N.p0
    if x == 7 and y0 == 1:
        out.append(("G", [592, 588 + (w >> 1), 576 + (0 if w == 0 else 3), 576 + (1 if w == 3 else 2)][role]))
    # ---- kind 4 ----
    if rx == ry:
        out.append(("a", y * 16 + ry * 8 + x))                                 # 8x4: 2NxN part ry of the 8x8 CU
    else:
        out.append(("b", 128 + y * 16 + 2 * x + rx))                           # 4x8: Nx2N part rx
    r, c = 2 * y0 + ry, 2 * x0 + rx                                            # 4x4 row / column inside the 16x16 CU
    if rx == 0 and x0 == 1:
        out.append(("c", 256 + AMP_ROW[r] * 16 + r16))
    if ry == 1 and y0 == 1:
        out.append(("d", 256 + AMP_COL[c] * 16 + r16))
    if rx == 1 and ry == 1 and x0 == 0:
        out.append(("e", 256 + (3 if y0 else 2) * 16 + r16))                   # rows 2,3 of 2NxnU.p1 / rows 0,1 of 2NxnD.p0
    if rx == 1 and ry == 0 and y0 == 0:
        out.append(("f", 256 + (7 if x0 else 6) * 16 + r16))                   # columns 2,3 of nLx2N.p1 / columns 0,1 of nRx2N.p0
    return out


# lane bits each sum runs over (the group of lanes that hold the same value of it)
GROUP_BITS = {"A": (), "B": (2,), "C": (5,), "D": (2, 5), "E": (2, 3), "F": (2, 3, 5), "G": (2, 3, 4, 5),
              "a": (0,), "b": (1,), "c": (0, 2), "d": (1, 5), "e": (0, 1, 2), "f": (0, 1, 5)}


def region(tid, name):
    """pixel rectangle (x, y, w, h) inside the CTU that sum `name` of lane tid covers"""
    kind8 = name.isupper()
    lanes = [tid]
    for b in GROUP_BITS[name]:
        lanes += [l ^ (1 << b) for l in lanes]
    xs, ys = [], []
    for l in lanes:
        t = (tid & ~63) | (l & 63)
        role, p8 = t & 3, t >> 2
        px, py = (p8 & 7) * 8, (p8 >> 3) * 8
        if kind8:
            xs += [px, px + 8]; ys += [py, py + 8]
        else:
            xs += [px + 4 * (role & 1), px + 4 * (role & 1) + 4]; ys += [py + 4 * (role >> 1), py + 4 * (role >> 1) + 4]
    x0, y0, x1, y1 = min(xs), min(ys), max(xs), max(ys)
    # the group's lanes tile a rectangle: each stands for its 8x8 position (kind 8: the quad's four lanes hold the same value, the group
    # runs over position bits only) or for its own 4x4 block (kind 4)
    assert (x1 - x0) * (y1 - y0) == len(lanes) * (64 if kind8 else 16), (tid, name)
    return x0, y0, x1 - x0, y1 - y0


def check(slot_rect):
    """slot_rect(slot) -> (x, y, w, h).  Every slot is tiled exactly once by the pieces added to it; returns the number of (slot, piece) adds."""
    import numpy as np
    cover = {s: np.zeros((64, 64), np.int32) for s in range(593)}
    n = 0
    for tid in range(256):
        for name, slot in duties(tid):
            x, y, w, h = region(tid, name)
            sx, sy, sw, sh = slot_rect(slot)
            kind8 = sw % 8 == 0 and sh % 8 == 0
            assert kind8 == name.isupper(), (tid, name, slot)
            assert sx <= x and sy <= y and x + w <= sx + sw and y + h <= sy + sh, (tid, name, slot, (x, y, w, h), (sx, sy, sw, sh))
            cover[slot][y:y + h, x:x + w] += 1
            n += 1
    for s in range(593):
        sx, sy, sw, sh = slot_rect(s)
        want = np.zeros((64, 64), np.int32)
        want[sy:sy + sh, sx:sx + sw] = 1
        assert (cover[s] == want).all(), ("slot", s, (sx, sy, sw, sh))
    return n


if __name__ == "__main__":
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hm-opencl_amd"))
    from hmme import api
    print("pieces added per CTU and stage:", check(api.slot_rect), "(the per-entry walk: 64 x 18 + 256 x 6 = 2688 lane adds)")
