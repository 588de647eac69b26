"""Live cross-check of the oracle against the reference's own compiled code
(oracle/_ref/libhmref.so).  Skipped where the library was never built (it is prebuilt here
and travels with the gpurun snapshot; /root/reference itself never does)."""
import ctypes as C

import numpy as np
import pytest

import oracle_py

pytestmark = pytest.mark.skipif(not oracle_py.ref_available(), reason="oracle/_ref/libhmref.so not built")


def test_random_pu_searches_match_reference(oracle_lib):
    R = oracle_lib.ref()
    rng = np.random.default_rng(2024)
    table = oracle_lib.slot_table()
    for it in range(40):
        bd = int(rng.choice([8, 10]))
        sr = int(rng.choice([3, 8, 12]))
        side = 64 + 2 * sr + 8
        cur = rng.integers(0, 1 << bd, size=(64, 64)).astype(np.int16)
        ref = rng.integers(0, 1 << bd, size=(side, side)).astype(np.int16)
        o = sr + 4
        lam = float(rng.choice([0.0, 3.3, 57.9, 900.0, 5.0e6]))
        pred = (int(rng.integers(-60, 61)), int(rng.integers(-60, 61)))
        lt = (-int(rng.integers(0, sr + 1)), -int(rng.integers(0, sr + 1)))
        rb = (int(rng.integers(0, sr + 1)), int(rng.integers(0, sr + 1)))
        fen = int(rng.integers(0, 2))
        lq = R.ref_lambda_q16(lam)
        p = oracle_lib.make_params(lt, rb, pred, lq, fen, bd)
        ox, oy, osad = oracle_lib.search_ctu(cur, (0, 0), ref, (o, o), p)
        for s in rng.choice(593, size=25, replace=False):
            x, y, w, h = (int(v) for v in table[s])
            want = oracle_lib.pattern_search(cur, (x, y), ref, (o + x, o + y), w, h, p, use_ref=True, lam=lam)
            assert (int(ox[s]), int(oy[s]), int(osad[s])) == want, (it, s)
            assert oracle_lib.pattern_search(cur, (x, y), ref, (o + x, o + y), w, h, p) == want


def test_random_weighted_pu_searches_match_reference(oracle_lib):
    """live: the reference's xPatternSearch with bApplyWeight (explicit weighted prediction) against the oracle's all-slot and
    per-PU weighted searches, random weights incl. negative ones and shift 0"""
    R = oracle_lib.ref()
    if not hasattr(R, "ref_pattern_search_w"):
        pytest.skip("libhmref.so predates the weighted-prediction harness")
    rng = np.random.default_rng(404)
    table = oracle_lib.slot_table()
    for it in range(24):
        bd = int(rng.choice([8, 10]))
        sr = int(rng.choice([3, 8]))
        side = 64 + 2 * sr + 8
        cur = rng.integers(0, 1 << bd, size=(64, 64)).astype(np.int16)
        ref = rng.integers(0, 1 << bd, size=(side, side)).astype(np.int16)
        o = sr + 4
        lam = float(rng.choice([0.0, 57.9, 900.0]))
        pred = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
        shift = int(rng.integers(0, 8))
        wp = (int(rng.integers(-40, 128)), int(rng.integers(-128, 128)) << (bd - 8), shift, (1 << (shift - 1)) if shift else 0)
        fen = int(rng.integers(0, 2))
        p = oracle_lib.make_params((-sr, -sr), (sr, sr), pred, R.ref_lambda_q16(lam), fen, bd)
        ox, oy, osad = oracle_lib.search_ctu_w(cur, (0, 0), ref, (o, o), p, wp)
        for s in rng.choice(593, size=16, replace=False):
            x, y, w, h = (int(v) for v in table[s])
            want = oracle_lib.pattern_search_w(cur, (x, y), ref, (o + x, o + y), w, h, p, wp, use_ref=True, lam=lam)
            assert (int(ox[s]), int(oy[s]), int(osad[s])) == want, (it, s, wp)
            assert oracle_lib.pattern_search_w(cur, (x, y), ref, (o + x, o + y), w, h, p, wp) == want


def test_cost_and_bits_match_reference(oracle_lib):
    L, R = oracle_lib.oracle(), oracle_lib.ref()
    rng = np.random.default_rng(5)
    for v in rng.integers(-40000, 40001, size=500):
        assert L.hmo_component_bits(int(v)) == R.ref_component_bits(int(v))
    for lam in (0.1, 33.0, 2.0e5, 9.9e6):
        lq = R.ref_lambda_q16(lam)
        assert lq == L.hmo_lambda_q16(lam)
        for _ in range(100):
            x, y, px, py = (int(v) for v in rng.integers(-300, 301, size=4))
            assert L.hmo_mv_cost(lq, x, y, px, py, 2) == R.ref_mv_cost(lam, x, y, px, py)


def test_fractional_refinement_matches_reference(oracle_lib):
    from hmme import synth
    rng = np.random.default_rng(99)
    table = oracle_lib.slot_table()
    for it in range(40):
        bd = 8 if it % 3 else 10
        cur, ref, _ = synth.make_pair(160, 160, seed=200 + it, bit_depth=bd, max_mv=3, region=64, margin=16, noise_sigma=3.0)
        x, y, w, h = (int(v) for v in table[int(rng.integers(0, 593))])
        mv = (int(rng.integers(-5, 6)), int(rng.integers(-5, 6)))
        pred = (int(rng.integers(-50, 51)), int(rng.integers(-50, 51)))
        lam = float(rng.choice([0.0, 12.0, 57.9, 3000.0]))
        had = int(rng.integers(0, 2))
        lq = oracle_lib.oracle().hmo_lambda_q16(lam)
        o = 16 + 48
        a = oracle_lib.frac_refine(cur, (o + x, o + y), ref, (o + x, o + y), w, h, mv, pred, lq, had, bd)
        b = oracle_lib.frac_refine(cur, (o + x, o + y), ref, (o + x, o + y), w, h, mv, pred, lam, had, bd, use_ref=True)
        assert a == b, (it, w, h, mv, pred, lam, had, bd)
        # the bBi call: origin 2*org - pred_other (unclipped, TComYuv.cpp:409-440), biPred = true
        other = np.roll(cur, (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))), axis=(0, 1)).astype(np.int32) + rng.integers(-30, 31, size=cur.shape)
        org = (2 * cur.astype(np.int32) - np.clip(other, 0, (1 << bd) - 1)).astype(np.int16)
        a = oracle_lib.frac_refine(org, (o + x, o + y), ref, (o + x, o + y), w, h, mv, pred, lq, had, bd)
        b = oracle_lib.frac_refine(org, (o + x, o + y), ref, (o + x, o + y), w, h, mv, pred, lam, had, bd, use_ref=True, bi=True)
        assert a == b, ("bi", it, w, h, mv, pred, lam, had, bd)


def test_tz_search_over_whole_ctus_matches_reference(oracle_lib):
    """bench.py's CPU baseline legs: the reference's own xTZSearch over all 593 PU shapes of a CTU range (ref_tz_frame: 64x64 PU first,
    the others seeded with its integer MV, TEncSearch.cpp:3780-3789) == the oracle's threaded restatement (hmo_tz_frame), 8 and 10 bit"""
    from hmme import synth
    for bd, sr in ((8, 64), (10, 24)):
        w, h = 448, 320
        cur, ref, _ = synth.make_pair(w, h, seed=60 + bd, bit_depth=bd)
        m = synth.MARGIN
        lq = oracle_lib.oracle().hmo_lambda_q16(57.9)
        rx, ry, rs = oracle_lib.ref_tz_frame(cur, ref, (m, m), w, h, sr, 57.9, 1, bd, 8, 9)
        _, _, ox, oy, os_ = oracle_lib.tz_frame(cur, ref, (m, m), w, h, sr, None, lq, 1, bd, 8, 9, 4, True, True)
        assert np.array_equal(rx, ox) and np.array_equal(ry, oy) and np.array_equal(rs, os_)


def test_full_search_over_whole_ctus_matches_reference(oracle_lib):
    """bench.py's `cpu_baseline.reference_full_search` leg: the reference's own xPatternSearch for each of the 593 PUs of a CTU range (one
    window per CTU, predictor (0,0)) == the oracle's whole-frame restatement, incl. a picture-edge CTU whose window clipMv cuts, 8 and 10 bit"""
    from hmme import synth
    for bd, sr, first in ((8, 12, 0), (10, 7, 8)):
        w, h = 448, 320
        cur, ref, _ = synth.make_pair(w, h, seed=70 + bd, bit_depth=bd)
        m = synth.MARGIN
        lq = oracle_lib.oracle().hmo_lambda_q16(57.9)
        rx, ry, rs, dt = oracle_lib.ref_full_search_ctus(cur, ref, (m, m), w, h, sr, 57.9, 1, bd, first, min_ctus=2, max_ctus=2, budget_s=0.0)
        ox, oy, os_ = oracle_lib.search_frame(cur, ref, (m, m), w, h, sr, None, lq, 1, bd, first, 2, 2)
        assert rx.shape == (2, 593) and dt > 0
        assert np.array_equal(rx, ox) and np.array_equal(ry, oy) and np.array_equal(rs, os_)
