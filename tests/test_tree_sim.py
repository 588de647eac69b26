"""CPU check of the generated reduction tree (tools/gen_me_tree.py): the numpy interpreter runs
the very op list that is emitted as HIP source, wrapped in a Python model of the kernel's task
loop / key packing / raster-order merge, and must reproduce the reference goldens bit for bit."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_me_tree as G  # noqa: E402

INV = 3146751
IDX = G.IDX_BITS


def cbits(v):
    t = (-v * 2 + 1) if v <= 0 else 2 * v
    return 2 * (int(t).bit_length() - 1) + 1


def model_search(tree, cur, ref, origin, lt, rb, pred, lq, iters_per_task=2):
    """python model of me_search_kernel: returns (593,3) mvx, mvy, sad"""
    wx, wy = rb[0] - lt[0] + 1, rb[1] - lt[1] + 1
    ox, oy = origin[0] + lt[0], origin[1] + lt[1]
    win = np.zeros((wy + 63, G.PDW * 4), np.uint8)
    w = ref[oy:oy + wy + 63, ox:ox + min(G.PDW * 4, ref.shape[1] - ox)]
    win[:w.shape[0], :w.shape[1]] = w
    slot_of = tree.slot_of_lane()
    best64 = np.full(593, (1 << 64) - 1, dtype=np.uint64)
    lanes = np.arange(64)
    quads = (wx + 3) >> 2
    fold = bool(quads & 32) and bool(quads & 31) and bool(wy & 1)   # me_fold(): idle lanes of the 32-quad part's last iteration
    x0 = 0
    for k in range(5, -1, -1):
        if not quads & (1 << k):
            continue
        ty = 64 >> k
        iters = ((wy if k == 5 or not fold else wy - 1) + ty - 1) // ty
        for it0 in range(0, iters, iters_per_task):
            best = np.full((G.N_GROUPS, 64), 0xFFFFFFFF, np.uint32)
            n_it = min(iters_per_task, iters - it0)
            lx, ly = lanes & ((1 << k) - 1), lanes >> k
            for it in range(n_it):
                cx = x0 + 4 * lx
                cy = (it0 + it) * ty + ly
                if fold and k == 5:
                    m_ = cy == wy
                    cx = np.where(m_, cx + 128, cx)
                    cy = np.where(m_, wy - 1, cy)
                c = np.zeros((4, 64), np.uint32)
                for l in range(64):
                    by = cbits(((lt[1] + int(cy[l])) << 2) - pred[1])
                    for j in range(4):
                        cost = ((lq * (cbits(((lt[0] + int(cx[l]) + j) << 2) - pred[0]) + by)) & 0xFFFFFFFF) >> 16
                        valid = cy[l] < wy and cx[l] + j < wx
                        c[j, l] = ((cost if valid else INV) << IDX) | (it << 8) | (l << 2) | j
                lane_off = (np.minimum(cy, wy - 1) * G.PDW + (np.minimum(cx, wx - 1) >> 2)) * 4
                G.simulate(tree, win, cur, lane_off, c, best)
            for g in range(G.N_GROUPS):
                for l in range(64):
                    s, key = slot_of[g, l], int(best[g, l])
                    cost = key >> IDX
                    if s < 0 or cost >= INV:
                        continue
                    kit, kl, kj = (key >> 8) & 3, (key >> 2) & 63, key & 3
                    bx = x0 + 4 * (kl & ((1 << k) - 1)) + kj
                    byy = (it0 + kit) * ty + (kl >> k)
                    if fold and k == 5 and byy == wy:
                        bx, byy = bx + 128, wy - 1
                    v = np.uint64((cost << 32) | (byy << 16) | bx)
                    if v < best64[s]:
                        best64[s] = v
        x0 += 4 << k
    out = np.zeros((593, 3), np.int64)
    for s in range(593):
        v = int(best64[s])
        mvx, mvy = lt[0] + (v & 0xffff), lt[1] + ((v >> 16) & 0xffff)
        mvc = ((lq * (cbits((mvx << 2) - pred[0]) + cbits((mvy << 2) - pred[1]))) & 0xFFFFFFFF) >> 16
        out[s] = (mvx, mvy, (v >> 32) - mvc)
    return out


def test_slot_map_is_a_permutation():
    t = G.Tree(1).build().slot_of_lane()
    assert sorted(int(v) for v in t.reshape(-1) if v >= 0) == list(range(593))
    assert np.array_equal(t, G.Tree(0).build().slot_of_lane())


@pytest.mark.parametrize("case", [0, 1, 2, 5, 6, 8, 9, 10, 12, 13])
def test_generated_tree_reproduces_reference_goldens(case):
    d = np.load(os.path.join(GOLDEN, "search_sr8.npz"))
    m = dict(zip(d["meta_columns"].tolist(), (int(v) for v in d["meta"][case])))
    if m["bit_depth"] != 8:
        pytest.skip("8-bit tree")
    tree = G.Tree(m["fen"]).build()
    cur = d["cur"][case].astype(np.uint8)
    ref = d["ref"][case].astype(np.uint8)
    got = model_search(tree, cur, ref, (m["origin_x"], m["origin_y"]), (m["lt_x"], m["lt_y"]), (m["rb_x"], m["rb_y"]),
                       (m["pred_x"], m["pred_y"]), m["lambda_q16"])
    assert np.array_equal(got, d["out"][case])


def test_generated_tree_full_window_with_folded_last_row():
    """the 129 x 129 window of SearchRange 64 (33 quads per row, odd row count): the leftover quad of the last row rides in
    the idle lanes of the 32-quad part's last iteration (me_fold)"""
    d = np.load(os.path.join(GOLDEN, "search_sr64.npz"))
    m = dict(zip(d["meta_columns"].tolist(), (int(v) for v in d["meta"][0])))
    assert (m["rb_x"] - m["lt_x"] + 1, m["rb_y"] - m["lt_y"] + 1) == (129, 129) and m["bit_depth"] == 8
    tree = G.Tree(m["fen"]).build()
    got = model_search(tree, d["cur"][0].astype(np.uint8), d["ref"][0].astype(np.uint8), (m["origin_x"], m["origin_y"]),
                       (m["lt_x"], m["lt_y"]), (m["rb_x"], m["rb_y"]), (m["pred_x"], m["pred_y"]), m["lambda_q16"])
    assert np.array_equal(got, d["out"][0])


INV16, IDX16 = 8000000, G.IDX_BITS16


def model_search16(tree, cur, ref, origin, lt, rb, pred, lq, bit_depth):
    """python model of me_search16_kernel (16-bit samples; two passes, even and odd window columns, the LDS window loaded with a
    shift of `par` samples; a lane owns candidates (x, x+2, x+4); linear lane packing; one lane-iteration per task)"""
    sh = bit_depth - 8
    wx, wy = rb[0] - lt[0] + 1, rb[1] - lt[1] + 1
    ox, oy = origin[0] + lt[0], origin[1] + lt[1]
    pitch = 2 * ((wx + 63 + 2 + 1) // 2) + 8
    slot_of = tree.slot_of_lane()
    best64 = np.full(593, (1 << 64) - 1, dtype=np.uint64)
    lanes = np.arange(64)
    for par in range(2):
        win = np.zeros((wy + 63, pitch), np.uint16)
        w = ref[oy:oy + wy + 63, ox + par:ox + par + min(pitch, ref.shape[1] - ox - par)]
        win[:w.shape[0], :w.shape[1]] = w
        n_par = (wx + 1 - par) // 2            # candidates of this column parity per window row
        P = (n_par + 2) // 3                   # lanes per window row
        iters = (wy * P + 63) // 64
        for it0 in range(iters):
            best = np.full((G.N_GROUPS, 64), 0xFFFFFFFF, np.uint32)
            q = it0 * 64 + lanes
            cy, cx = q // P, par + 6 * (q % P)
            c = np.zeros((3, 64), np.uint32)
            for l in range(64):
                by = cbits(((lt[1] + int(cy[l])) << 2) - pred[1])
                for j in range(3):
                    x = int(cx[l]) + 2 * j
                    cost = ((lq * (cbits(((lt[0] + x) << 2) - pred[0]) + by)) & 0xFFFFFFFF) >> 16
                    valid = cy[l] < wy and x < wx
                    c[j, l] = ((cost if valid else INV16) << IDX16) | (l << 2) | j
            lane_off = np.minimum(cy, wy - 1) * pitch + (cx - par)
            G.simulate16(tree, win, cur, lane_off, c, best, sh)
            for g in range(G.N_GROUPS):
                for l in range(64):
                    s, key = slot_of[g, l], int(best[g, l])
                    cost = key >> IDX16
                    if s < 0 or cost >= INV16:
                        continue
                    kl, kj = (key >> 2) & 63, key & 3
                    q1 = it0 * 64 + kl
                    v = np.uint64((cost << 32) | ((q1 // P) << 16) | (par + 6 * (q1 % P) + 2 * kj))
                    if v < best64[s]:
                        best64[s] = v
    out = np.zeros((593, 3), np.int64)
    for s in range(593):
        v = int(best64[s])
        mvx, mvy = lt[0] + (v & 0xffff), lt[1] + ((v >> 16) & 0xffff)
        mvc = ((lq * (cbits((mvx << 2) - pred[0]) + cbits((mvy << 2) - pred[1]))) & 0xFFFFFFFF) >> 16
        out[s] = (mvx, mvy, (v >> 32) - mvc)
    return out


@pytest.mark.parametrize("case", [3, 4, 11, 15, 0, 1, 5, 10])
def test_generated_16bit_tree_reproduces_reference_goldens(case):
    """10-bit goldens through the 16-bit tree; 8-bit goldens too (shift 0 must also be exact)"""
    d = np.load(os.path.join(GOLDEN, "search_sr8.npz"))
    m = dict(zip(d["meta_columns"].tolist(), (int(v) for v in d["meta"][case])))
    tree = G.Tree16(m["fen"]).build()
    got = model_search16(tree, d["cur"][case].astype(np.uint16), d["ref"][case].astype(np.uint16),
                         (m["origin_x"], m["origin_y"]), (m["lt_x"], m["lt_y"]), (m["rb_x"], m["rb_y"]),
                         (m["pred_x"], m["pred_y"]), m["lambda_q16"], m["bit_depth"])
    assert np.array_equal(got, d["out"][case])


def test_refinement_tree_accumulation_tiles_every_slot_exactly_once():
    """tools/frac_tree_model.py restates the designated lanes and slot numbers of me_frac_tree_add (the refinement kernel's ME_FRAC_TREE build):
    the pieces the four waves add to a slot must tile its rectangle exactly once, for all 593 slots of the reference's layout"""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import frac_tree_model
    from hmme import api
    api.build()
    n = frac_tree_model.check(api.slot_rect)
    assert n == 744          # 296 sums over 8x8 positions + 448 over 4x4 blocks, against 2 688 lane adds of the entry-by-entry walk
