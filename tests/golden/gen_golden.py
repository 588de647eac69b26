#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the COMPILED REFERENCE
(oracle/_ref/libhmref.so, built by oracle/Makefile from /root/reference).

Run here (the reference does not travel to the GPU box):   python tests/golden/gen_golden.py
Every .npz holds inputs and the reference's outputs -- data only.

  slots.npz        G4  TComDataCU::getIndexBlock for every tabulated (partSize, depth, partIdx, absZIdx)
  cost.npz         G3  xGetComponentBits / getCost / lambda scaling
  sad.npz          G2  xGetSAD{4,8,12,16,24,32,48,64} known answers, iSubShift 0/1, bit depth 8/10
  range.npz            xSetSearchRange + clipMv
  search_sr8.npz   G1  xPatternSearch on all 593 PU rectangles of a CTU, SR 8, many parameter mixes
  search_sr64.npz  G1  same at SR 64 (two cases)
  tz.npz           G5  xTZSearch for a set of PUs
  frac.npz             xPatternSearchFracDIF (half + quarter-pel refinement, HAD or SAD) for a set of PUs
  border.npz           TComPicYuv::create + extendPicBorder: small pictures in, the whole padded luma buffer out
  frac_bipred.npz      xPatternSearchFracDIF(..., biPred = true) on bi-prediction origins 2*org - pred_other (samples outside the
                       sample range, TEncSearch.cpp:3702-3712): `python tests/golden/gen_golden.py frac_bipred` makes this file alone
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))
import oracle_py as O  # noqa: E402
from hmme import synth  # noqa: E402

R = None


def zidx(bx, by):
    z = 0
    for b in range(4):
        z |= ((bx >> b) & 1) << (2 * b)
        z |= ((by >> b) & 1) << (2 * b + 1)
    return z


def pu_rect(ps, pi, s):
    x = y = 0
    w = h = s
    if ps == 1:
        h = s // 2; y = pi * s // 2
    elif ps == 2:
        w = s // 2; x = pi * s // 2
    elif ps == 4:
        h = (3 * s // 4) if pi else s // 4; y = s // 4 if pi else 0
    elif ps == 5:
        h = s // 4 if pi else 3 * s // 4; y = 3 * s // 4 if pi else 0
    elif ps == 6:
        w = (3 * s // 4) if pi else s // 4; x = s // 4 if pi else 0
    elif ps == 7:
        w = s // 4 if pi else 3 * s // 4; x = 3 * s // 4 if pi else 0
    return x, y, w, h


def gen_slots():
    rows = []
    for depth in range(4):
        s = 64 >> depth
        n = 64 // s
        for cy in range(n):
            for cx in range(n):
                z = zidx(cx * s // 4, cy * s // 4)
                for ps in (0, 1, 2, 3, 4, 5, 6, 7):
                    for pi in range(4 if ps == 3 else 2):
                        if ps == 0 and pi:
                            continue
                        slot = R.ref_index_block(ps, depth, pi, z, s, s)
                        if slot < 0:
                            continue
                        x, y, w, h = pu_rect(ps, pi, s)
                        rows.append((slot, ps, depth, pi, z, s, cx * s + x, cy * s + y, w, h))
    t = np.array(rows, np.int32)
    assert sorted(t[:, 0].tolist()) == list(range(593)), "reference table is not a bijection onto 0..592"
    np.savez_compressed(os.path.join(HERE, "slots.npz"), table=t,
                        columns=np.array("slot part_size depth part_idx abs_z cu_size x y w h".split()))
    print("slots:", t.shape)
    return t


def gen_cost():
    vals = np.arange(-2200, 2201, dtype=np.int32)
    bits = np.array([R.ref_component_bits(int(v)) for v in vals], np.uint32)
    lambdas = np.array([0.0, 0.25, 1.0, 4.7, 57.9, 238.5, 4670.3, 1.0e6, 4.0e6, 3.0e7], np.float64)
    lq16 = np.array([R.ref_lambda_q16(float(l)) for l in lambdas], np.uint32)
    rng = np.random.default_rng(7)
    pts = rng.integers(-140, 141, size=(400, 2)).astype(np.int32)
    preds = rng.integers(-600, 601, size=(400, 2)).astype(np.int32)
    preds[:50] = 0
    costs = np.zeros((len(lambdas), len(pts)), np.uint32)
    for i, l in enumerate(lambdas):
        for j in range(len(pts)):
            costs[i, j] = R.ref_mv_cost(float(l), int(pts[j, 0]), int(pts[j, 1]), int(preds[j, 0]), int(preds[j, 1]))
    np.savez_compressed(os.path.join(HERE, "cost.npz"), vals=vals, bits=bits, lambdas=lambdas, lambda_q16=lq16,
                        pts=pts, preds=preds, costs=costs)
    print("cost: bits", bits.shape, "costs", costs.shape)


def gen_sad():
    rng = np.random.default_rng(11)
    cases, pairs_a, pairs_b, outs = [], [], [], []
    for bd in (8, 10):
        maxv = (1 << bd) - 1
        for kind in range(3):
            if kind == 0:
                a = rng.integers(0, maxv + 1, size=(64, 64)).astype(np.int16)
                b = rng.integers(0, maxv + 1, size=(64, 64)).astype(np.int16)
            elif kind == 1:
                a = np.full((64, 64), maxv, np.int16); b = np.zeros((64, 64), np.int16)
            else:  # bi-pred style origin outside the sample range (TComYuv.cpp:426-438)
                a = rng.integers(-maxv, 2 * maxv + 1, size=(64, 64)).astype(np.int16)
                b = rng.integers(0, maxv + 1, size=(64, 64)).astype(np.int16)
            pair = len(pairs_a)
            pairs_a.append(a); pairs_b.append(b)
            for w in (4, 8, 12, 16, 24, 32, 48, 64):
                for h in (4, 8, 12, 16, 24, 32, 48, 64):
                    for sub in (0, 1):
                        ox, oy = int(rng.integers(0, 64 - w + 1)), int(rng.integers(0, 64 - h + 1))
                        v = R.ref_sad(O._addr(a, oy * 64 + ox), 64, O._addr(b, oy * 64 + ox), 64, w, h, sub, bd)
                        cases.append((w, h, sub, bd, pair, ox, oy)); outs.append(v)
    np.savez_compressed(os.path.join(HERE, "sad.npz"), cases=np.array(cases, np.int32), a=np.stack(pairs_a),
                        b=np.stack(pairs_b), sad=np.array(outs, np.uint32),
                        columns=np.array("w h sub_shift bit_depth pair x y".split()))
    print("sad:", len(outs))


def gen_range():
    rng = np.random.default_rng(13)
    rows = []
    for (pw, ph) in ((64, 64), (192, 128), (416, 240), (1920, 1080), (3840, 2160)):
        ctx, cty = (pw + 63) // 64, (ph + 63) // 64
        for _ in range(60):
            cu_x, cu_y = int(rng.integers(0, ctx)) * 64, int(rng.integers(0, cty)) * 64
            sr = int(rng.choice([4, 8, 64, 128]))
            px, py = (int(v) for v in rng.integers(-1200, 1201, size=2))
            if rng.random() < 0.3:
                px = py = 0
            lt_x, lt_y, rb_x, rb_y = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            R.ref_set_search_range(px, py, sr, cu_x, cu_y, pw, ph, 64, C.byref(lt_x), C.byref(lt_y), C.byref(rb_x), C.byref(rb_y))
            rows.append((px, py, sr, cu_x, cu_y, pw, ph, 64, lt_x.value, lt_y.value, rb_x.value, rb_y.value))
    np.savez_compressed(os.path.join(HERE, "range.npz"), rows=np.array(rows, np.int32))
    print("range:", len(rows))


def ref_search_all_slots(cur, ref_plane, origin, table, lt, rb, pred, lam, fen, bd):
    """593 x (mvx, mvy, sad) from the reference's xPatternSearch, one call per PU rectangle.
    cur: (64,64) int16 CTU; ref_plane with the CTU origin at `origin` (x,y)."""
    out = np.zeros((593, 3), np.int64)
    rs = ref_plane.shape[1]
    for row in table:
        slot, x, y, w, h = int(row[0]), int(row[6]), int(row[7]), int(row[8]), int(row[9])
        mx, my, sad = C.c_int(), C.c_int(), C.c_uint32()
        R.ref_pattern_search(O._addr(cur, y * 64 + x), 64, w, h,
                             O._addr(ref_plane, (origin[1] + y) * rs + origin[0] + x), rs,
                             lt[0], lt[1], rb[0], rb[1], pred[0], pred[1], float(lam), fen, bd,
                             C.byref(mx), C.byref(my), C.byref(sad))
        out[slot] = (mx.value, my.value, sad.value)
    return out


def make_case(seed, sr, bd, kind="motion"):
    """-> cur CTU (64,64), ref window (64+2*sr+2*g)^2 with the CTU origin at (sr+g, sr+g)"""
    g = 8  # slack so clipped / shifted windows stay inside the stored array
    side = 64 + 2 * (sr + g)
    rng = np.random.default_rng(seed)
    maxv = (1 << bd) - 1
    if kind == "motion":
        cur_p, ref_p, _ = synth.make_pair(side, side, seed=seed, bit_depth=bd, max_mv=min(sr - 1, 12), region=48, margin=0)
        o = sr + g
        return cur_p[o:o + 64, o:o + 64].copy(), ref_p.copy(), (o, o)
    if kind == "noise":
        return (rng.integers(0, maxv + 1, size=(64, 64)).astype(np.int16),
                rng.integers(0, maxv + 1, size=(side, side)).astype(np.int16), (sr + g, sr + g))
    if kind == "flat":
        return np.full((64, 64), 100, np.int16), np.full((side, side), 100, np.int16), (sr + g, sr + g)
    if kind == "extreme":
        return np.full((64, 64), maxv, np.int16), np.zeros((side, side), np.int16), (sr + g, sr + g)
    if kind == "zeros":  # many zero samples on both sides (catches "masked SAD" instruction misuse)
        c = rng.integers(0, 4, size=(64, 64)).astype(np.int16) * (rng.random((64, 64)) < 0.5)
        r = (rng.integers(0, maxv + 1, size=(side, side)) * (rng.random((side, side)) < 0.5)).astype(np.int16)
        return c.astype(np.int16), r, (sr + g, sr + g)
    raise ValueError(kind)


def gen_search(table, name, specs):
    curs, refs, metas, outs = [], [], [], []
    for (seed, sr, bd, kind, fen, lam, pred, lt, rb) in specs:
        cur, ref_plane, origin = make_case(seed, sr, bd, kind)
        if lt is None:
            lt, rb = (-sr, -sr), (sr, sr)
        out = ref_search_all_slots(cur, ref_plane, origin, table, lt, rb, pred, lam, fen, bd)
        curs.append(cur); refs.append(ref_plane); outs.append(out)
        metas.append((seed, sr, bd, fen, pred[0], pred[1], lt[0], lt[1], rb[0], rb[1], origin[0], origin[1],
                      R.ref_lambda_q16(float(lam))))
        print(f"  {name} case seed={seed} sr={sr} bd={bd} {kind} fen={fen} lam={lam} pred={pred} lt={lt} rb={rb}"
              f" -> slot592 {out[592].tolist()}")
    np.savez_compressed(os.path.join(HERE, name), cur=np.stack(curs), ref=np.stack(refs),
                        meta=np.array(metas, np.int64), lambdas=np.array([s[5] for s in specs], np.float64),
                        kinds=np.array([s[3] for s in specs]), out=np.stack(outs),
                        meta_columns=np.array("seed sr bit_depth fen pred_x pred_y lt_x lt_y rb_x rb_y origin_x origin_y lambda_q16".split()))


def gen_tz(table):
    rng = np.random.default_rng(17)
    sr, bd = 64, 8
    pic = 64 * 5
    cur_p, ref_p, _ = synth.make_pair(pic, pic, seed=99, bit_depth=bd, max_mv=20, region=64)
    m = synth.MARGIN
    rows, outs = [], []
    slots = [592, 588, 591, 584, 585, 560, 576, 512, 544, 545, 448, 480, 256, 300, 384, 400, 0, 128, 200]
    for it in range(120):
        slot = slots[it % len(slots)]
        x, y, w, h = (int(v) for v in table[table[:, 0] == slot][0, 6:10])
        ctu_x, ctu_y = int(rng.integers(0, 5)) * 64, int(rng.integers(0, 5)) * 64
        fen = int(rng.integers(0, 2))
        lam = float(rng.choice([4.7, 57.9, 238.5]))
        pred = [int(v) for v in rng.integers(-100, 101, size=2)] if rng.random() < 0.7 else [0, 0]
        has_int = int(rng.random() < 0.5)
        imv = [int(v) for v in rng.integers(-30, 31, size=2)]
        lt_x, lt_y, rb_x, rb_y = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        R.ref_set_search_range(pred[0], pred[1], sr, ctu_x, ctu_y, pic, pic, 64, C.byref(lt_x), C.byref(lt_y),
                               C.byref(rb_x), C.byref(rb_y))
        cs = cur_p.shape[1]
        off = (m + ctu_y + y) * cs + m + ctu_x + x
        mx, my, sad = C.c_int(), C.c_int(), C.c_uint32()
        R.ref_tz_search(O._addr(cur_p, off), cs, w, h, O._addr(ref_p, off), cs, lt_x.value, lt_y.value, rb_x.value,
                        rb_y.value, pred[0], pred[1], lam, fen, bd, sr, ctu_x, ctu_y, pic, pic, 64, has_int, imv[0],
                        imv[1], C.byref(mx), C.byref(my), C.byref(sad))
        rows.append((slot, x, y, w, h, ctu_x, ctu_y, fen, pred[0], pred[1], has_int, imv[0], imv[1], lt_x.value,
                     lt_y.value, rb_x.value, rb_y.value, R.ref_lambda_q16(lam)))
        outs.append((mx.value, my.value, sad.value))
    np.savez_compressed(os.path.join(HERE, "tz.npz"), rows=np.array(rows, np.int64), out=np.array(outs, np.int64),
                        pic=np.array([pic, pic, sr, bd, 99, 20, 64]), cur=cur_p, ref=ref_p,
                        columns=np.array("slot x y w h ctu_x ctu_y fen pred_x pred_y has_int imv_x imv_y lt_x lt_y rb_x rb_y lambda_q16".split()))
    print("tz:", len(rows))


def gen_border():
    """TComPicYuv::extendPicBorder (TComPicYuv.cpp:214-262) on pictures of awkward sizes: margin = maxCU + 16, stride = W + 2 * margin"""
    rng = np.random.default_rng(31)
    R.ref_extend_border.restype = C.c_int
    d = {}
    for i, (w, h, bd) in enumerate([(40, 24, 8), (8, 8, 8), (72, 8, 10), (8, 136, 8), (200, 136, 10)]):
        img = rng.integers(0, 1 << bd, size=(h, w)).astype(np.int16)
        out = np.zeros((w + 400) * (h + 400), np.int16)
        st = C.c_int()
        m = R.ref_extend_border(img.ctypes.data_as(C.POINTER(C.c_int16)), w, w, h, 64, out.ctypes.data_as(C.POINTER(C.c_int16)), out.size, C.byref(st))
        assert m == 80 and st.value == w + 160
        d[f"img{i}"] = img
        d[f"out{i}"] = out[:st.value * (h + 2 * m)].reshape(h + 2 * m, st.value).copy()
    np.savez_compressed(os.path.join(HERE, "border.npz"), n=np.array(5), margin=np.array(80), **d)
    print("border: 5")


def gen_frac(table):
    """xPatternSearchFracDIF: (PU, integer MV, predictor, lambda, HAD on/off, bit depth) -> (half, quarter, cost)"""
    rng = np.random.default_rng(23)
    planes = {}
    for bd in (8, 10):
        cur_p, ref_p, _ = synth.make_pair(192, 192, seed=50 + bd, bit_depth=bd, max_mv=3, region=64, margin=16, noise_sigma=2.0)
        planes[bd] = (cur_p, ref_p)
    rows, outs = [], []
    for it in range(160):
        bd = 10 if it % 4 == 3 else 8
        cur_p, ref_p = planes[bd]
        slot = int(rng.integers(0, 593)) if it >= 20 else [592, 588, 590, 576, 512, 544, 448, 256, 300, 340, 384, 0, 128, 130, 584, 560, 568, 520, 530, 480][it]
        x, y, w, h = (int(v) for v in table[table[:, 0] == slot][0, 6:10])
        mv = [int(v) for v in rng.integers(-6, 7, size=2)]
        pred = [int(v) for v in rng.integers(-40, 41, size=2)]
        lam = float(rng.choice([0.0, 4.7, 57.9, 900.0]))
        had = int(it % 5 != 4)
        o = 16 + 64
        cs = cur_p.shape[1]
        off = (o + y) * cs + o + x
        h_ = [C.c_int() for _ in range(4)]
        cost = C.c_uint32()
        R.ref_frac_refine(O._addr(cur_p, off), cs, w, h, O._addr(ref_p, off), cs, mv[0], mv[1], pred[0], pred[1], lam, had, bd,
                          *[C.byref(v) for v in h_], C.byref(cost))
        rows.append((slot, x, y, w, h, mv[0], mv[1], pred[0], pred[1], had, bd, R.ref_lambda_q16(lam), o))
        outs.append(tuple(v.value for v in h_) + (cost.value,))
    np.savez_compressed(os.path.join(HERE, "frac.npz"), rows=np.array(rows, np.int64), out=np.array(outs, np.int64),
                        cur8=planes[8][0], ref8=planes[8][1], cur10=planes[10][0], ref10=planes[10][1],
                        columns=np.array("slot x y w h int_x int_y pred_x pred_y had bit_depth lambda_q16 origin".split()))
    print("frac:", len(rows))


def gen_frac_wp(table):
    """xPatternSearchFracDIF in a slice with explicit weighted prediction (m_cDistParam.bApplyWeight: xGetHADsw / xGetSADw on the weighted
    interpolated prediction): faded current pictures and the weights that undo them, mismatched weights, a negative weight, shift 0"""
    rng = np.random.default_rng(29)
    wps = [(52, 17, 6, 32), (80, -30, 6, 32), (-64, 255, 6, 32), (71, 5, 6, 32), (2, -100, 0, 0), (100, -64, 7, 64), (64, 0, 6, 32)]
    planes = {}
    for bd in (8, 10):
        cur_p, ref_p, _ = synth.make_pair(192, 192, seed=70 + bd, bit_depth=bd, max_mv=3, region=64, margin=16, noise_sigma=2.0)
        planes[bd] = (cur_p, ref_p)
    rows, outs, curs = [], [], []
    for it in range(112):
        bd = 10 if it % 4 == 3 else 8
        cur_p, ref_p = planes[bd]
        maxv = (1 << bd) - 1
        wp = wps[it % len(wps)]
        wp = (wp[0], wp[1] << (bd - 8), wp[2], wp[3])
        # the current block as the weighted reference would predict it (a fade), so that the weights matter and the search has a clear minimum
        denom = float(1 << wp[2])
        faded = np.clip(np.rint(cur_p.astype(np.float64) * (wp[0] / denom) + wp[1]), 0, maxv).astype(np.int16) if it % 3 else cur_p
        slot = int(rng.integers(0, 593)) if it >= 16 else [592, 588, 590, 576, 512, 544, 448, 256, 300, 340, 384, 0, 128, 130, 584, 560][it]
        x, y, w, h = (int(v) for v in table[table[:, 0] == slot][0, 6:10])
        mv = [int(v) for v in rng.integers(-6, 7, size=2)]
        pred = [int(v) for v in rng.integers(-40, 41, size=2)]
        lam = float(rng.choice([0.0, 4.7, 57.9, 900.0]))
        had = int(it % 5 != 4)
        o = 16 + 64
        cs = cur_p.shape[1]
        off = (o + y) * cs + o + x
        h_ = [C.c_int() for _ in range(4)]
        cost = C.c_uint32()
        R.ref_frac_refine_w(O._addr(faded, off), cs, w, h, O._addr(ref_p, off), cs, mv[0], mv[1], pred[0], pred[1], lam, had, bd, *wp,
                            *[C.byref(v) for v in h_], C.byref(cost))
        rows.append((slot, x, y, w, h, mv[0], mv[1], pred[0], pred[1], had, bd, R.ref_lambda_q16(lam), o) + wp + (len(curs),))
        curs.append(faded)
        outs.append(tuple(v.value for v in h_) + (cost.value,))
    # the faded planes repeat: keep the distinct ones
    uniq, index = [], []
    for c in curs:
        for i, u in enumerate(uniq):
            if u.shape == c.shape and np.array_equal(u, c):
                index.append(i); break
        else:
            index.append(len(uniq)); uniq.append(c)
    rows = [r[:-1] + (index[r[-1]],) for r in rows]
    np.savez_compressed(os.path.join(HERE, "frac_wp.npz"), rows=np.array(rows, np.int64), out=np.array(outs, np.int64),
                        cur=np.stack(uniq), ref8=planes[8][1], ref10=planes[10][1],
                        columns=np.array("slot x y w h int_x int_y pred_x pred_y had bit_depth lambda_q16 origin wp_w wp_offset wp_shift wp_round cur_index".split()))
    print("frac_wp:", len(rows), "cases,", len(uniq), "current planes")


def gen_frac_bipred(table):
    """the bBi call: (PU of the origin plane 2*org - pred_other, integer MV, predictor, lambda, HAD on/off, bit depth) -> (half, quarter, cost)"""
    rng = np.random.default_rng(29)
    planes = {}
    for bd in (8, 10):
        cur_p, ref_p, _ = synth.make_pair(192, 192, seed=70 + bd, bit_depth=bd, max_mv=3, region=64, margin=16, noise_sigma=2.0)
        _, other, _ = synth.make_pair(192, 192, seed=90 + bd, bit_depth=bd, max_mv=3, region=64, margin=16, noise_sigma=2.0)
        # TComYuv::removeHighFreq with DISABLING_CLIP_FOR_BIPREDME: 2 * org - pred, unclipped.  `other` is an unrelated picture for the
        # first plane (origins spread over the whole [-maxv, 2*maxv]) and a close prediction for the second (what an encoder sees)
        near = np.clip(cur_p.astype(np.int32) + rng.integers(-12, 13, size=cur_p.shape) * (1 << (bd - 8)), 0, (1 << bd) - 1)
        maxv = (1 << bd) - 1
        stretch = lambda a: np.clip((a.astype(np.int32) - maxv // 2) * 4 + maxv // 2, 0, maxv)   # saturates: origins reach -maxv and 2*maxv
        planes[bd] = ((2 * stretch(cur_p) - stretch(other)).astype(np.int16), (2 * cur_p.astype(np.int32) - near).astype(np.int16), ref_p)
    rows, outs = [], []
    for it in range(96):
        bd = 10 if it % 3 == 2 else 8
        which = it % 2
        org_p, ref_p = planes[bd][which], planes[bd][2]
        slot = int(rng.integers(0, 593)) if it >= 12 else [592, 588, 590, 576, 512, 544, 448, 256, 384, 0, 128, 584][it]
        x, y, w, h = (int(v) for v in table[table[:, 0] == slot][0, 6:10])
        mv = [int(v) for v in rng.integers(-4, 5, size=2)]
        pred = [int(v) for v in rng.integers(-40, 41, size=2)]
        lam = float(rng.choice([0.0, 4.7, 57.9, 900.0]))
        had = int(it % 5 != 4)
        o = 16 + 64
        cs = org_p.shape[1]
        off = (o + y) * cs + o + x
        h_ = [C.c_int() for _ in range(4)]
        cost = C.c_uint32()
        R.ref_frac_refine_bi(O._addr(org_p, off), cs, w, h, O._addr(ref_p, off), cs, mv[0], mv[1], pred[0], pred[1], lam, had, bd,
                             *[C.byref(v) for v in h_], C.byref(cost))
        rows.append((slot, x, y, w, h, mv[0], mv[1], pred[0], pred[1], had, bd, R.ref_lambda_q16(lam), o, which))
        outs.append(tuple(v.value for v in h_) + (cost.value,))
    for bd in (8, 10):
        maxv = (1 << bd) - 1
        assert planes[bd][0].min() == -maxv and planes[bd][0].max() == 2 * maxv, (bd, planes[bd][0].min(), planes[bd][0].max())
    np.savez_compressed(os.path.join(HERE, "frac_bipred.npz"), rows=np.array(rows, np.int64), out=np.array(outs, np.int64),
                        org8_0=planes[8][0], org8_1=planes[8][1], ref8=planes[8][2], org10_0=planes[10][0], org10_1=planes[10][1],
                        ref10=planes[10][2],
                        columns=np.array("slot x y w h int_x int_y pred_x pred_y had bit_depth lambda_q16 origin which_plane".split()))
    print("frac_bipred:", len(rows))


def gen_wp(table):
    """explicit weighted prediction (TEncSearch::setWpScalingDistParam, TEncSearch.cpp:5594-5635 -> TComRdCostWeightPrediction::xGetSADw):
    weighted SAD known answers for every width function, and whole 593-slot searches with bApplyWeight from the reference's own
    xPatternSearch.  wp = (w, offset, shift, round) as WPScalingParam carries them for luma."""
    rng = np.random.default_rng(23)
    # ---- SAD known answers: all widths x heights, sub_shift 0 / 1 (must make no difference), 8 / 10 bit, weights incl. negative and shift 0
    wps = [(64, 0, 6, 32), (80, -12, 6, 32), (45, 21, 6, 32), (3, -100, 0, 0), (-17, 300, 4, 8), (127, -128, 7, 64), (1, 0, 0, 0)]
    cases, outs, pa, pb = [], [], [], []
    for bd in (8, 10):
        maxv = (1 << bd) - 1
        a = rng.integers(0, maxv + 1, size=(64, 64)).astype(np.int16)
        b = rng.integers(0, maxv + 1, size=(64, 64)).astype(np.int16)
        pa.append(a); pb.append(b)
        for wi, wp in enumerate(wps):
            for w in (4, 8, 12, 16, 24, 32, 48, 64):
                for h in (4, 8, 16, 24, 64):
                    for sub in (0, 1):
                        ox, oy = int(rng.integers(0, 64 - w + 1)), int(rng.integers(0, 64 - h + 1))
                        v = R.ref_sad_w(O._addr(a, oy * 64 + ox), 64, O._addr(b, oy * 64 + ox), 64, w, h, sub, bd, *wp)
                        cases.append((w, h, sub, bd, len(pa) - 1, ox, oy, wi)); outs.append(v)
    # ---- searches: a fade (cur = gain * displaced ref + offset, the weights that undo it), the same content with a NEGATIVE weight, a
    # weight table that does not match the content, shift 0, 10-bit with the offset scaled to the bit depth, a clipped window
    specs = [
        # seed sr bd  fen lam     pred      lt        rb        wp (w, offset, shift, round)   fade (gain, offset) applied to cur
        (31, 8, 8, 1, 57.9, (0, 0), None, None, (52, 17, 6, 32), (52 / 64, 17)),
        (32, 8, 8, 0, 57.9, (9, -6), None, None, (80, -30, 6, 32), (80 / 64, -30)),
        (33, 8, 8, 1, 57.9, (0, 0), None, None, (-64, 255, 6, 32), (-1.0, 255)),
        (34, 8, 8, 1, 238.5, (-14, 3), None, None, (71, 5, 6, 32), (1.0, 0)),
        (35, 8, 8, 1, 57.9, (0, 0), None, None, (2, -100, 0, 0), (1.0, 0)),
        (36, 8, 10, 1, 57.9, (4, 4), None, None, (100, -64, 7, 64), (100 / 128, -64)),
        (37, 8, 10, 0, 4.0e6, (0, 0), (-8, -3), (5, 8), (140, 80, 7, 64), (140 / 128, 80)),
        (38, 8, 8, 1, 0.0, (0, 0), None, None, (64, 0, 6, 32), (1.0, 0)),
    ]
    curs, refs, metas, results = [], [], [], []
    for (seed, sr, bd, fen, lam, pred, lt, rb, wp, fade) in specs:
        cur, ref_plane, origin = make_case(seed, sr, bd, "motion")
        maxv = (1 << bd) - 1
        cur = np.clip(np.rint(cur.astype(np.float64) * fade[0] + fade[1]), 0, maxv).astype(np.int16)
        if lt is None:
            lt, rb = (-sr, -sr), (sr, sr)
        out = np.zeros((593, 3), np.int64)
        rs = ref_plane.shape[1]
        for row in table:
            slot, x, y, w, h = int(row[0]), int(row[6]), int(row[7]), int(row[8]), int(row[9])
            mx, my, sad = C.c_int(), C.c_int(), C.c_uint32()
            R.ref_pattern_search_w(O._addr(cur, y * 64 + x), 64, w, h, O._addr(ref_plane, (origin[1] + y) * rs + origin[0] + x), rs,
                                   lt[0], lt[1], rb[0], rb[1], pred[0], pred[1], float(lam), fen, bd, *wp, C.byref(mx), C.byref(my), C.byref(sad))
            out[slot] = (mx.value, my.value, sad.value)
        curs.append(cur); refs.append(ref_plane); results.append(out)
        metas.append((seed, sr, bd, fen, pred[0], pred[1], lt[0], lt[1], rb[0], rb[1], origin[0], origin[1], R.ref_lambda_q16(float(lam))) + tuple(wp))
        print(f"  wp case seed={seed} bd={bd} fen={fen} wp={wp} -> slot592 {out[592].tolist()}")
    np.savez_compressed(os.path.join(HERE, "wp.npz"), sad_cases=np.array(cases, np.int32), sad_a=np.stack(pa), sad_b=np.stack(pb),
                        sad_wp=np.array(wps, np.int32), sad=np.array(outs, np.uint32),
                        sad_columns=np.array("w h sub_shift bit_depth pair x y wp_index".split()),
                        cur=np.stack(curs), ref=np.stack(refs), meta=np.array(metas, np.int64), out=np.stack(results),
                        meta_columns=np.array("seed sr bit_depth fen pred_x pred_y lt_x lt_y rb_x rb_y origin_x origin_y lambda_q16 wp_w wp_offset wp_shift wp_round".split()))
    print("wp:", len(outs), "weighted SADs,", len(specs), "weighted searches")


def gen_slots_amp_off():
    """The 425-entry table of a reference build with AMP_ENC_SPEEDUP (TComDataCU.cpp:3393-4675, compiled out in the tree as shipped, so it
    cannot be called): the (key, index) pairs of its switch, read from the source text -- data, like slots.npz.
    key = ((((partSize + 10*depth + 100*partIdx)*1000 + absZIdx)*100 + height)*100 + width (TComDataCU.cpp:3379-3391)."""
    import re
    src = open("/root/reference/source/Lib/TLibCommon/TComDataCU.cpp").read().split("\n")
    seg = "\n".join(src[3392:4676])
    pairs = re.findall(r"case\s+(\d+):\s*\n\s*index\s*=\s*(\d+);", seg)
    rows = []
    for k, i in pairs:
        k = int(k)
        w, h, z, t = k % 100, (k // 100) % 100, (k // 10000) % 1000, k // 10000000
        rows.append((int(i), t % 10, (t // 10) % 10, t // 100, z, h, w))
    rows.sort()
    assert len(rows) == 425 and [r[0] for r in rows] == list(range(425))
    np.savez_compressed(os.path.join(HERE, "slots_amp_off.npz"), table=np.array(rows, np.int32),
                        columns=np.array("index part_size depth part_idx abs_z_idx height width".split()))
    print("slots_amp_off:", len(rows))


def main():
    global R
    O.build(ref=True)
    R = O.ref()
    if sys.argv[1:] == ["frac_bipred"]:   # this file alone; the slot table comes from the committed slots.npz
        gen_frac_bipred(np.load(os.path.join(HERE, "slots.npz"))["table"])
        return
    if sys.argv[1:] == ["slots_amp_off"]:
        gen_slots_amp_off()
        return
    if sys.argv[1:] == ["wp"]:
        gen_wp(np.load(os.path.join(HERE, "slots.npz"))["table"])
        return
    if sys.argv[1:] == ["frac_wp"]:
        gen_frac_wp(np.load(os.path.join(HERE, "slots.npz"))["table"])
        return
    table = gen_slots()
    gen_cost()
    gen_sad()
    gen_range()
    sr8 = [
        # seed sr bd kind     fen lam     pred        lt        rb
        (1, 8, 8, "motion", 1, 57.9, (0, 0), None, None),
        (2, 8, 8, "motion", 0, 57.9, (0, 0), None, None),
        (3, 8, 8, "motion", 1, 57.9, (13, -7), None, None),
        (4, 8, 10, "motion", 1, 238.5, (-21, 30), None, None),
        (5, 8, 10, "motion", 0, 238.5, (0, 0), None, None),
        (6, 8, 8, "motion", 1, 57.9, (5, 9), (-3, -8), (8, 2)),
        (7, 8, 8, "flat", 1, 0.0, (0, 0), None, None),
        (8, 8, 8, "flat", 1, 57.9, (6, -10), None, None),
        (9, 8, 8, "noise", 1, 4.0e6, (3, 3), None, None),
        (10, 8, 8, "zeros", 1, 4.7, (0, 0), None, None),
        (11, 8, 8, "extreme", 1, 57.9, (0, 0), None, None),
        (12, 8, 10, "extreme", 0, 57.9, (0, 0), None, None),
        (13, 8, 8, "noise", 0, 1.0, (-40, 40), (-8, -8), (-8, 8)),   # one-column window
        (14, 8, 8, "motion", 1, 57.9, (0, 0), (2, 3), (2, 3)),       # single candidate
        (15, 8, 8, "noise", 1, 57.9, (1, -2), (-8, -5), (7, 8)),
        (16, 8, 10, "noise", 1, 3.0e7, (0, 0), None, None),
    ]
    gen_search(table, "search_sr8.npz", sr8)
    sr64 = [
        (21, 64, 8, "motion", 1, 57.9, (0, 0), None, None),
        (22, 64, 8, "motion", 1, 57.9, (-37, 22), (-64, -40), (50, 64)),
    ]
    gen_search(table, "search_sr64.npz", sr64)
    gen_tz(table)
    gen_frac(table)
    gen_frac_bipred(table)
    gen_wp(table)
    gen_frac_wp(table)
    gen_slots_amp_off()
    gen_border()


if __name__ == "__main__":
    main()
