"""The oracle (oracle/hm_oracle.c) against the golden vectors generated from the compiled
reference (tests/golden/gen_golden.py) -- runs on CPU, here and on the GPU box."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN


def g(name):
    return np.load(os.path.join(GOLDEN, name))


def test_slot_table_matches_reference_getIndexBlock(oracle_lib, slots):
    # reference: TComDataCU::getIndexBlock, TComDataCU.cpp:3379-3391 + :4676-6461
    L = oracle_lib.oracle()
    mine = oracle_lib.slot_table()
    assert len(slots) == 593
    for row in slots:
        slot, ps, depth, pi, z, s, x, y, w, h = (int(v) for v in row)
        assert L.hmo_index_block(ps, depth, pi, z, s) == slot
        assert tuple(mine[slot]) == (x, y, w, h)
    # keys that the reference does not tabulate return -1 (NxN, wrong depth)
    assert L.hmo_index_block(3, 3, 0, 0, 8) == -1
    assert L.hmo_index_block(4, 3, 0, 0, 8) == -1
    assert L.hmo_index_block(0, 0, 0, 0, 32) == -1
    assert L.hmo_index_key(0, 0, 0, 0, 64, 64) == 6464
    assert L.hmo_index_key(2, 0, 1, 0, 64, 64) == 1020006464


def test_component_bits_and_mv_cost(oracle_lib):
    # reference: TComRdCost::xGetComponentBits (TComRdCost.cpp:278-292), getCost (TComRdCost.h:172-189)
    L = oracle_lib.oracle()
    d = g("cost.npz")
    for v, b in zip(d["vals"], d["bits"]):
        assert L.hmo_component_bits(int(v)) == int(b)
    for lam, q in zip(d["lambdas"], d["lambda_q16"]):
        assert L.hmo_lambda_q16(float(lam)) == int(q)
    for i, q in enumerate(d["lambda_q16"]):
        for j in range(len(d["pts"])):
            got = L.hmo_mv_cost(int(q), int(d["pts"][j, 0]), int(d["pts"][j, 1]), int(d["preds"][j, 0]),
                                int(d["preds"][j, 1]), 2)
            assert got == int(d["costs"][i, j]), (i, j)
    assert d["costs"].max() < 65536  # (u32 product) >> 16: the bound the GPU key packing relies on


def test_sad_known_answers(oracle_lib):
    # reference: TComRdCost::xGetSAD4..64 / 12 / 24 / 48 (TComRdCost.cpp:493-964)
    L = oracle_lib.oracle()
    d = g("sad.npz")
    for (w, h, sub, bd, pair, x, y), want in zip(d["cases"], d["sad"]):
        a, b = d["a"][pair], d["b"][pair]
        off = int(y) * 64 + int(x)
        got = L.hmo_sad(oracle_lib._addr(a, off), 64, oracle_lib._addr(b, off), 64, int(w), int(h), int(sub), int(bd))
        assert got == int(want), (w, h, sub, bd, pair)


def test_search_range_and_clip(oracle_lib):
    # reference: TEncSearch::xSetSearchRange (TEncSearch.cpp:3814-3830), TComDataCU::clipMv (:2907-2920)
    L = oracle_lib.oracle()
    for r in g("range.npz")["rows"]:
        out = [C.c_int() for _ in range(4)]
        L.hmo_set_search_range(*[int(v) for v in r[:8]], *[C.byref(o) for o in out])
        assert [o.value for o in out] == [int(v) for v in r[8:12]], r


def _check_search(oracle_lib, fname, literal_slots):
    d = g(fname)
    table = oracle_lib.slot_table()
    for i in range(len(d["cur"])):
        m = dict(zip(d["meta_columns"].tolist(), (int(v) for v in d["meta"][i])))
        cur, ref = np.ascontiguousarray(d["cur"][i]), np.ascontiguousarray(d["ref"][i])
        p = oracle_lib.make_params((m["lt_x"], m["lt_y"]), (m["rb_x"], m["rb_y"]), (m["pred_x"], m["pred_y"]),
                                   m["lambda_q16"], m["fen"], m["bit_depth"])
        ox, oy, osad = oracle_lib.search_ctu(cur, (0, 0), ref, (m["origin_x"], m["origin_y"]), p)
        want = d["out"][i]
        assert np.array_equal(ox, want[:, 0]) and np.array_equal(oy, want[:, 1]), f"case {i}: MV mismatch"
        assert np.array_equal(osad.astype(np.int64), want[:, 2]), f"case {i}: SAD mismatch"
        # the literal xPatternSearch restatement, PU by PU
        for s in literal_slots:
            x, y, w, h = (int(v) for v in table[s])
            got = oracle_lib.pattern_search(cur, (x, y), ref, (m["origin_x"] + x, m["origin_y"] + y), w, h, p)
            assert got == tuple(int(v) for v in want[s]), (i, s)


def test_full_search_all_slots_sr8(oracle_lib):
    # reference: TEncSearch::xPatternSearch (TEncSearch.cpp:3835-3897) on every PU rectangle
    _check_search(oracle_lib, "search_sr8.npz", range(0, 593, 7))


def test_weighted_prediction_sads_and_searches(oracle_lib):
    """explicit weighted prediction: TComRdCostWeightPrediction::xGetSADw (TComRdCostWeightPrediction.cpp:55-90) for every width function,
    and the reference's xPatternSearch with bApplyWeight on all 593 rectangles (tests/golden/wp.npz, gen_golden.py wp)"""
    L = oracle_lib.oracle()
    d = g("wp.npz")
    for (w, h, sub, bd, pair, x, y, wi), want in zip(d["sad_cases"], d["sad"]):
        a, b = d["sad_a"][pair], d["sad_b"][pair]
        off = int(y) * 64 + int(x)
        wp = oracle_lib.Wp(*[int(v) for v in d["sad_wp"][wi]])
        got = L.hmo_sad_w(oracle_lib._addr(a, off), 64, oracle_lib._addr(b, off), 64, int(w), int(h), int(bd), C.byref(wp))
        assert got == int(want), (w, h, sub, bd, pair, wi)      # sub_shift 0 and 1 give the same value: the FEN rows are never skipped
    table = oracle_lib.slot_table()
    for i in range(len(d["cur"])):
        m = dict(zip(d["meta_columns"].tolist(), (int(v) for v in d["meta"][i])))
        cur, ref = np.ascontiguousarray(d["cur"][i]), np.ascontiguousarray(d["ref"][i])
        p = oracle_lib.make_params((m["lt_x"], m["lt_y"]), (m["rb_x"], m["rb_y"]), (m["pred_x"], m["pred_y"]), m["lambda_q16"], m["fen"], m["bit_depth"])
        wp = (m["wp_w"], m["wp_offset"], m["wp_shift"], m["wp_round"])
        ox, oy, osad = oracle_lib.search_ctu_w(cur, (0, 0), ref, (m["origin_x"], m["origin_y"]), p, wp)
        want = d["out"][i]
        assert np.array_equal(ox, want[:, 0]) and np.array_equal(oy, want[:, 1]), f"case {i}: MV mismatch"
        assert np.array_equal(osad.astype(np.int64), want[:, 2]), f"case {i}: SAD mismatch"
        for s in range(0, 593, 37):
            x, y, w, h = (int(v) for v in table[s])
            got = oracle_lib.pattern_search_w(cur, (x, y), ref, (m["origin_x"] + x, m["origin_y"] + y), w, h, p, wp)
            assert got == tuple(int(v) for v in want[s]), (i, s)


def test_weighted_fractional_refinement(oracle_lib):
    """xPatternSearchFracDIF with m_cDistParam.bApplyWeight (xGetHADsw / xGetSADw on the weighted interpolated prediction): 112 goldens
    from the compiled reference (tests/golden/frac_wp.npz, gen_golden.py frac_wp)"""
    d = g("frac_wp.npz")
    cols = d["columns"].tolist()
    differs_from_unweighted = 0
    for row, want in zip(d["rows"], d["out"]):
        m = dict(zip(cols, (int(v) for v in row)))
        cur = np.ascontiguousarray(d["cur"][m["cur_index"]])
        ref = np.ascontiguousarray(d["ref8"] if m["bit_depth"] == 8 else d["ref10"])
        o = m["origin"]
        xy = (o + m["x"], o + m["y"])
        wp = (m["wp_w"], m["wp_offset"], m["wp_shift"], m["wp_round"])
        got = oracle_lib.frac_refine_w(cur, xy, ref, xy, m["w"], m["h"], (m["int_x"], m["int_y"]), (m["pred_x"], m["pred_y"]), m["lambda_q16"],
                                       m["had"], m["bit_depth"], wp)
        assert got == tuple(int(v) for v in want), (m, got, want)
        plain = oracle_lib.frac_refine(cur, xy, ref, xy, m["w"], m["h"], (m["int_x"], m["int_y"]), (m["pred_x"], m["pred_y"]), m["lambda_q16"],
                                       m["had"], m["bit_depth"])
        differs_from_unweighted += plain != got
    assert differs_from_unweighted > 60      # the weights matter in these cases (the identity weight (64, 0, 6, 32) is among them)


def test_full_search_all_slots_sr64(oracle_lib):
    _check_search(oracle_lib, "search_sr64.npz", (592, 588, 576, 300, 5))


def test_tz_search(oracle_lib):
    # reference: TEncSearch::xTZSearch (TEncSearch.cpp:3935-4136)
    L = oracle_lib.oracle()
    d = g("tz.npz")
    cur, ref = np.ascontiguousarray(d["cur"]), np.ascontiguousarray(d["ref"])
    pic_w, pic_h, sr, bd = (int(v) for v in d["pic"][:4])
    m = 80
    cs = cur.shape[1]
    for row, want in zip(d["rows"], d["out"]):
        (slot, x, y, w, h, ctu_x, ctu_y, fen, px, py, has_int, ix, iy, ltx, lty, rbx, rby, lq) = (int(v) for v in row)
        p = oracle_lib.make_params((ltx, lty), (rbx, rby), (px, py), lq, fen, bd)
        tz = oracle_lib.TzCtx(sr, ctu_x, ctu_y, pic_w, pic_h, 64)
        off = (m + ctu_y + y) * cs + m + ctu_x + x
        imv = (C.c_int * 2)(ix, iy)
        mx, my, sad = C.c_int(), C.c_int(), C.c_uint32()
        n = L.hmo_tz_search(oracle_lib._addr(cur, off), cs, w, h, oracle_lib._addr(ref, off), cs, C.byref(p), C.byref(tz),
                            imv if has_int else None, px, py, C.byref(mx), C.byref(my), C.byref(sad))
        assert n > 0
        assert (mx.value, my.value, sad.value) == tuple(int(v) for v in want), row


def test_extend_border(oracle_lib):
    # reference: TComPicYuv::extendPicBorder (TComPicYuv.cpp:214-262) == numpy edge padding
    L = oracle_lib.oracle()
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(24, 40)).astype(np.int16)
    mx, my = 16, 9
    buf = np.zeros((24 + 2 * my, 40 + 2 * mx), np.int16)
    buf[my:my + 24, mx:mx + 40] = img
    L.hmo_extend_border(oracle_lib._addr(buf, my * buf.shape[1] + mx), buf.shape[1], 40, 24, mx, my)
    assert np.array_equal(buf, np.pad(img, ((my, my), (mx, mx)), mode="edge"))


def test_extend_border_matches_reference_goldens(oracle_lib):
    # reference: TComPicYuv::create + extendPicBorder (TComPicYuv.cpp:80-133, :214-262) -- margin maxCU + 16 = 80, stride W + 160;
    # the oracle's restatement and synth.pad_plane (what every frame test feeds the oracle with) must both reproduce the buffer
    from hmme import synth
    d = g("border.npz")
    L = oracle_lib.oracle()
    m = int(d["margin"])
    assert m == synth.MARGIN
    for i in range(int(d["n"])):
        img, want = d[f"img{i}"], d[f"out{i}"]
        h, w = img.shape
        assert want.shape == (h + 2 * m, w + 2 * m)
        buf = np.zeros_like(want)
        buf[m:m + h, m:m + w] = img
        L.hmo_extend_border(oracle_lib._addr(buf, m * buf.shape[1] + m), buf.shape[1], w, h, m, m)
        assert np.array_equal(buf, want), i
        assert np.array_equal(synth.pad_plane(img), want), i


def test_search_frame_matches_per_ctu(oracle_lib):
    from hmme import synth
    w, h, sr = 160, 136, 8   # partial CTUs on both edges
    cur, ref, _ = synth.make_pair(w, h, seed=5, max_mv=6, region=64)
    n_ctu = 3 * 3
    pred = synth.random_predictors(n_ctu, 5)
    lq = oracle_lib.oracle().hmo_lambda_q16(57.9)
    ox, oy, osad = oracle_lib.search_frame(cur, ref, (80, 80), w, h, sr, pred, lq, 1, 8, n_threads=3)
    L = oracle_lib.oracle()
    for ctu in (0, 4, 8):
        cx, cy = (ctu % 3) * 64, (ctu // 3) * 64
        out = [C.c_int() for _ in range(4)]
        L.hmo_set_search_range(int(pred[ctu, 0]), int(pred[ctu, 1]), sr, cx, cy, w, h, 64, *[C.byref(o) for o in out])
        p = oracle_lib.make_params((out[0].value, out[1].value), (out[2].value, out[3].value),
                                   (int(pred[ctu, 0]), int(pred[ctu, 1])), lq, 1, 8)
        x1, y1, s1 = oracle_lib.search_ctu(cur, (80 + cx, 80 + cy), ref, (80 + cx, 80 + cy), p)
        assert np.array_equal(x1, ox[ctu]) and np.array_equal(y1, oy[ctu]) and np.array_equal(s1, osad[ctu])


def test_fractional_refinement(oracle_lib):
    # reference: TEncSearch::xPatternSearchFracDIF (TEncSearch.cpp:4294-4331) incl. interpolation and Hadamard cost
    d = g("frac.npz")
    planes = {8: (np.ascontiguousarray(d["cur8"]), np.ascontiguousarray(d["ref8"])),
              10: (np.ascontiguousarray(d["cur10"]), np.ascontiguousarray(d["ref10"]))}
    for row, want in zip(d["rows"], d["out"]):
        slot, x, y, w, h, ix, iy, px, py, had, bd, lq, o = (int(v) for v in row)
        cur, ref = planes[bd]
        got = oracle_lib.frac_refine(cur, (o + x, o + y), ref, (o + x, o + y), w, h, (ix, iy), (px, py), lq, had, bd)
        assert got == tuple(int(v) for v in want), row


def test_fractional_refinement_of_biprediction_origins(oracle_lib):
    # the bBi call xPatternSearchFracDIF(..., biPred = true) (TEncSearch.cpp:3798) on 2*org - pred_other (TEncSearch.cpp:3702-3712):
    # origins outside the sample range, interpolated reference clipped to it
    d = g("frac_bipred.npz")
    planes = {(bd, k): np.ascontiguousarray(d[f"org{bd}_{k}"]) for bd in (8, 10) for k in (0, 1)}
    refs = {bd: np.ascontiguousarray(d[f"ref{bd}"]) for bd in (8, 10)}
    assert planes[(8, 0)].min() == -255 and planes[(8, 0)].max() == 510 and planes[(10, 0)].min() == -1023 and planes[(10, 0)].max() == 2046
    for row, want in zip(d["rows"], d["out"]):
        slot, x, y, w, h, ix, iy, px, py, had, bd, lq, o, which = (int(v) for v in row)
        got = oracle_lib.frac_refine(planes[(bd, which)], (o + x, o + y), refs[bd], (o + x, o + y), w, h, (ix, iy), (px, py), lq, had, bd)
        assert got == tuple(int(v) for v in want), row
