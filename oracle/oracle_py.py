"""ctypes bindings for the CHECKER libraries (test infrastructure, never the product).

  liboracle.so        our CPU restatement (oracle/hm_oracle.c)
  _ref/libhmref.so    the reference's own code compiled from /root/reference (oracle/Makefile)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NUM_PARTS = 593

_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


class Params(C.Structure):
    _fields_ = [("lt_x", C.c_int), ("lt_y", C.c_int), ("rb_x", C.c_int), ("rb_y", C.c_int),
                ("pred_x", C.c_int), ("pred_y", C.c_int), ("lambda_q16", C.c_uint32),
                ("fen", C.c_int), ("bit_depth", C.c_int)]


class Wp(C.Structure):
    """hmo_wp: luma WPScalingParam of the reference picture (w, offset, shift, round)"""
    _fields_ = [("w0", C.c_int), ("offset", C.c_int), ("shift", C.c_int), ("round", C.c_int)]


class Rect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("w", C.c_int), ("h", C.c_int)]


class TzCtx(C.Structure):
    _fields_ = [("sr", C.c_int), ("cu_x", C.c_int), ("cu_y", C.c_int), ("pic_w", C.c_int),
                ("pic_h", C.c_int), ("max_cu", C.c_int)]


def build(ref=True):
    """(re)build liboracle.so and, when /root/reference is present, _ref/libhmref.so"""
    subprocess.run(["make", "-s", "-C", HERE, "-j8", "all" if ref else "oracle"], check=True)


_oracle = None
_ref = None


def _addr(a, off=0):
    """pointer to element `off` (may be negative w.r.t. a view) of a contiguous int16 array"""
    return C.cast(a.ctypes.data + 2 * off, C.POINTER(C.c_int16))


def oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.hmo_component_bits.restype = C.c_uint32
        L.hmo_component_bits.argtypes = [C.c_int]
        L.hmo_lambda_q16.restype = C.c_uint32
        L.hmo_lambda_q16.argtypes = [C.c_double]
        L.hmo_mv_cost.restype = C.c_uint32
        L.hmo_mv_cost.argtypes = [C.c_uint32] + [C.c_int] * 5
        L.hmo_sad.restype = C.c_uint32
        L.hmo_sad.argtypes = [C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int16), C.c_int] + [C.c_int] * 4
        L.hmo_slot_rect.restype = C.c_int
        L.hmo_slot_rect.argtypes = [C.c_int, C.POINTER(Rect)]
        L.hmo_index_key.restype = C.c_int32
        L.hmo_index_key.argtypes = [C.c_int] * 6
        L.hmo_index_block.restype = C.c_int
        L.hmo_index_block.argtypes = [C.c_int] * 5
        L.hmo_set_search_range.restype = None
        L.hmo_set_search_range.argtypes = [C.c_int] * 8 + [C.POINTER(C.c_int)] * 4
        L.hmo_pattern_search.restype = None
        L.hmo_pattern_search.argtypes = [C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int16),
                                         C.c_int, C.POINTER(Params), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.POINTER(C.c_uint32)]
        L.hmo_search_ctu.restype = None
        L.hmo_search_ctu.argtypes = [C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int16), C.c_int,
                                     C.POINTER(Params), _i32p, _i32p, _u32p, C.c_void_p]
        L.hmo_sad_w.restype = C.c_uint32
        L.hmo_sad_w.argtypes = [C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Wp)]
        L.hmo_pattern_search_w.restype = None
        L.hmo_pattern_search_w.argtypes = [C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int16), C.c_int, C.POINTER(Params),
                                           C.POINTER(Wp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
        L.hmo_search_ctu_w.restype = None
        L.hmo_search_ctu_w.argtypes = [C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int16), C.c_int, C.POINTER(Params), C.POINTER(Wp),
                                       _i32p, _i32p, _u32p, C.c_void_p]
        L.hmo_frac_refine_w.restype = None
        L.hmo_frac_refine_w.argtypes = ([C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int16), C.c_int]
                                        + [C.c_int] * 4 + [C.c_uint32, C.c_int, C.c_int, C.POINTER(Wp)] + [C.POINTER(C.c_int)] * 4
                                        + [C.POINTER(C.c_uint32)])
        L.hmo_ocl_compat_params.restype = None
        L.hmo_ocl_compat_params.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_uint32]
        L.hmo_tz_search.restype = C.c_long
        L.hmo_tz_search.argtypes = [C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int16), C.c_int,
                                    C.POINTER(Params), C.POINTER(TzCtx), C.POINTER(C.c_int), C.c_int, C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
        L.hmo_extend_border.restype = None
        L.hmo_extend_border.argtypes = [C.POINTER(C.c_int16)] + [C.c_int] * 5
        L.hmo_search_frame.restype = C.c_int
        L.hmo_search_frame.argtypes = [C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int16), C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, _i32p, _i32p, _u32p]
        L.hmo_frac_refine.restype = None
        L.hmo_frac_refine.argtypes = ([C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int16), C.c_int]
                                      + [C.c_int] * 4 + [C.c_uint32, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4
                                      + [C.POINTER(C.c_uint32)])
        L.hmo_refine_frame.restype = C.c_int
        L.hmo_refine_frame.argtypes = [C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hmo_had.restype = C.c_uint32
        L.hmo_had.argtypes = [C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.c_int]
        L.hmo_tz_frame.restype = C.c_int
        L.hmo_tz_frame.argtypes = [C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_long),
                                   C.POINTER(C.c_double)]
        _oracle = L
    return _oracle


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libhmref.so"))


def ref():
    """the compiled reference (None-safe: raises if it was never built)"""
    global _ref
    if _ref is None:
        L = C.CDLL(os.path.join(HERE, "_ref", "libhmref.so"))
        L.ref_component_bits.restype = C.c_uint32
        L.ref_component_bits.argtypes = [C.c_int]
        L.ref_lambda_q16.restype = C.c_uint32
        L.ref_lambda_q16.argtypes = [C.c_double]
        L.ref_mv_cost.restype = C.c_uint32
        L.ref_mv_cost.argtypes = [C.c_double] + [C.c_int] * 4
        L.ref_sad.restype = C.c_uint32
        L.ref_sad.argtypes = [C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int16), C.c_int] + [C.c_int] * 4
        L.ref_index_block.restype = C.c_int
        L.ref_index_block.argtypes = [C.c_int] * 6
        L.ref_set_search_range.restype = None
        L.ref_set_search_range.argtypes = [C.c_int] * 8 + [C.POINTER(C.c_int)] * 4
        L.ref_pattern_search.restype = None
        L.ref_pattern_search.argtypes = ([C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int16), C.c_int]
                                         + [C.c_int] * 6 + [C.c_double, C.c_int, C.c_int]
                                         + [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint32)])
        L.ref_tz_search.restype = None
        L.ref_tz_search.argtypes = ([C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int16), C.c_int]
                                    + [C.c_int] * 6 + [C.c_double, C.c_int, C.c_int] + [C.c_int] * 9
                                    + [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint32)])
        L.ref_frac_refine.restype = None
        L.ref_frac_refine.argtypes = ([C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int16), C.c_int]
                                      + [C.c_int] * 4 + [C.c_double, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4
                                      + [C.POINTER(C.c_uint32)])
        if hasattr(L, "ref_tz_frame"):   # harness of round 3 and later
            L.ref_tz_frame.restype = C.c_long
            L.ref_tz_frame.argtypes = [C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                       C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, _u32p]
        if hasattr(L, "ref_pattern_search_w"):   # harness of round 4 and later: explicit weighted prediction
            L.ref_pattern_search_w.restype = None
            L.ref_pattern_search_w.argtypes = ([C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int16), C.c_int]
                                               + [C.c_int] * 6 + [C.c_double, C.c_int, C.c_int] + [C.c_int] * 4
                                               + [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint32)])
            L.ref_frac_refine_w.restype = None
            L.ref_frac_refine_w.argtypes = ([C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int16), C.c_int]
                                            + [C.c_int] * 4 + [C.c_double, C.c_int, C.c_int] + [C.c_int] * 4 + [C.POINTER(C.c_int)] * 4
                                            + [C.POINTER(C.c_uint32)])
            L.ref_sad_w.restype = C.c_uint32
            L.ref_sad_w.argtypes = [C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int16), C.c_int] + [C.c_int] * 8
        L.ref_frac_refine_bi.restype = None
        L.ref_frac_refine_bi.argtypes = L.ref_frac_refine.argtypes
        _ref = L
    return _ref


# ---------------------------------------------------------------------------------------------
# convenience wrappers
# ---------------------------------------------------------------------------------------------

def slot_table():
    """(593, 4) int array of x, y, w, h from the oracle's closed form"""
    L = oracle()
    out = np.zeros((NUM_PARTS, 4), np.int32)
    r = Rect()
    for s in range(NUM_PARTS):
        assert L.hmo_slot_rect(s, C.byref(r)) == 0
        out[s] = (r.x, r.y, r.w, r.h)
    return out


def make_params(lt, rb, pred, lambda_q16, fen, bit_depth):
    return Params(lt[0], lt[1], rb[0], rb[1], pred[0], pred[1], int(lambda_q16), int(fen), int(bit_depth))


def search_ctu(plane_cur, cur_xy, plane_ref, ref_xy, p):
    """oracle exhaustive search of all 593 slots.  plane_* are 2-D C-contiguous int16 arrays,
    *_xy the (x, y) of the CTU origin inside them."""
    L = oracle()
    ox = np.zeros(NUM_PARTS, np.int32)
    oy = np.zeros(NUM_PARTS, np.int32)
    osad = np.zeros(NUM_PARTS, np.uint32)
    cs, rs = plane_cur.shape[1], plane_ref.shape[1]
    L.hmo_search_ctu(_addr(plane_cur, cur_xy[1] * cs + cur_xy[0]), cs,
                     _addr(plane_ref, ref_xy[1] * rs + ref_xy[0]), rs, C.byref(p), ox, oy, osad, None)
    return ox, oy, osad


def search_ctu_w(plane_cur, cur_xy, plane_ref, ref_xy, p, wp):
    """all 593 slots in a slice with explicit weighted prediction; wp = (w0, offset, shift, round)"""
    L = oracle()
    ox = np.zeros(NUM_PARTS, np.int32)
    oy = np.zeros(NUM_PARTS, np.int32)
    osad = np.zeros(NUM_PARTS, np.uint32)
    cs, rs = plane_cur.shape[1], plane_ref.shape[1]
    w = Wp(*[int(v) for v in wp])
    L.hmo_search_ctu_w(_addr(plane_cur, cur_xy[1] * cs + cur_xy[0]), cs, _addr(plane_ref, ref_xy[1] * rs + ref_xy[0]), rs, C.byref(p), C.byref(w),
                       ox, oy, osad, None)
    return ox, oy, osad


def pattern_search_w(plane_cur, cur_xy, plane_ref, ref_xy, w, h, p, wp, use_ref=False, lam=None):
    """exhaustive search of ONE PU with explicit weighted prediction wp = (w0, offset, shift, round): the oracle, or the reference's
    own xPatternSearch with bApplyWeight (use_ref; `lam` the double-precision lambda)"""
    cs, rs = plane_cur.shape[1], plane_ref.shape[1]
    org = _addr(plane_cur, cur_xy[1] * cs + cur_xy[0])
    rf = _addr(plane_ref, ref_xy[1] * rs + ref_xy[0])
    mx, my, sad = C.c_int(), C.c_int(), C.c_uint32()
    if use_ref:
        ref().ref_pattern_search_w(org, cs, w, h, rf, rs, p.lt_x, p.lt_y, p.rb_x, p.rb_y, p.pred_x, p.pred_y, float(lam), p.fen, p.bit_depth,
                                   *[int(v) for v in wp], C.byref(mx), C.byref(my), C.byref(sad))
    else:
        ww = Wp(*[int(v) for v in wp])
        oracle().hmo_pattern_search_w(org, cs, w, h, rf, rs, C.byref(p), C.byref(ww), C.byref(mx), C.byref(my), C.byref(sad))
    return mx.value, my.value, sad.value


def pattern_search(plane_cur, cur_xy, plane_ref, ref_xy, w, h, p, use_ref=False, lam=None):
    """exhaustive search of ONE PU: the oracle's literal restatement, or the reference's own
    xPatternSearch when use_ref (then `lam` is the double-precision lambda)."""
    cs, rs = plane_cur.shape[1], plane_ref.shape[1]
    org = _addr(plane_cur, cur_xy[1] * cs + cur_xy[0])
    rf = _addr(plane_ref, ref_xy[1] * rs + ref_xy[0])
    mx, my, sad = C.c_int(), C.c_int(), C.c_uint32()
    if use_ref:
        ref().ref_pattern_search(org, cs, w, h, rf, rs, p.lt_x, p.lt_y, p.rb_x, p.rb_y, p.pred_x, p.pred_y,
                                 float(lam), p.fen, p.bit_depth, C.byref(mx), C.byref(my), C.byref(sad))
    else:
        oracle().hmo_pattern_search(org, cs, w, h, rf, rs, C.byref(p), C.byref(mx), C.byref(my), C.byref(sad))
    return mx.value, my.value, sad.value


def search_frame(cur, ref_plane, origin, pic_w, pic_h, sr, pred_q, lambda_q16, fen, bit_depth,
                 ctu_first=0, ctu_count=-1, n_threads=1):
    """oracle whole-frame search on padded int16 planes; origin = (x, y) of sample (0,0)"""
    L = oracle()
    ctus = ((pic_w + 63) // 64) * ((pic_h + 63) // 64)
    n = ctus - ctu_first if ctu_count < 0 else ctu_count
    ox = np.zeros((n, NUM_PARTS), np.int32)
    oy = np.zeros((n, NUM_PARTS), np.int32)
    osad = np.zeros((n, NUM_PARTS), np.uint32)
    cs, rs = cur.shape[1], ref_plane.shape[1]
    pq = None
    if pred_q is not None:
        pred_q = np.ascontiguousarray(pred_q, dtype=np.int16)
        pq = pred_q.ctypes.data
    L.hmo_search_frame(_addr(cur, origin[1] * cs + origin[0]), cs, _addr(ref_plane, origin[1] * rs + origin[0]), rs,
                       pic_w, pic_h, sr, pq, int(lambda_q16), int(fen), int(bit_depth), ctu_first, n,
                       n_threads, ox.reshape(-1), oy.reshape(-1), osad.reshape(-1))
    return ox, oy, osad


def refine_frame(cur, ref_plane, origin, pic_w, pic_h, int_mv, pred_q, lambda_q16, use_had, bit_depth, ctu_first=0, ctu_count=-1, n_threads=1):
    """oracle xPatternSearchFracDIF for every slot of the CTUs [ctu_first, +ctu_count); int_mv: int16 [count, 593, 2]
    -> (qmv int16 [count, 593, 2] quarter-pel, cost uint32 [count, 593])"""
    L = oracle()
    ctus = ((pic_w + 63) // 64) * ((pic_h + 63) // 64)
    n = ctus - ctu_first if ctu_count < 0 else ctu_count
    imv = np.ascontiguousarray(int_mv, dtype=np.int16)
    assert imv.shape == (n, NUM_PARTS, 2)
    qmv = np.zeros((n, NUM_PARTS, 2), np.int16)
    cost = np.zeros((n, NUM_PARTS), np.uint32)
    cs, rs = cur.shape[1], ref_plane.shape[1]
    pq = None
    if pred_q is not None:
        pred_q = np.ascontiguousarray(pred_q, dtype=np.int16)
        pq = pred_q.ctypes.data
    L.hmo_refine_frame(_addr(cur, origin[1] * cs + origin[0]), cs, _addr(ref_plane, origin[1] * rs + origin[0]), rs, pic_w, pic_h, pq,
                       int(lambda_q16), int(use_had), int(bit_depth), ctu_first, n, n_threads, imv.ctypes.data, qmv.ctypes.data, cost.ctypes.data)
    return qmv, cost


def tz_frame(cur, ref_plane, origin, pic_w, pic_h, sr, pred_q, lambda_q16, fen, bit_depth, ctu_first=0,
             ctu_count=-1, n_threads=1, all_slots=True, want_results=False):
    """oracle xTZSearch over CTUs (CPU baseline).  -> (probes, sad4x4_equiv[, x, y, sad])"""
    L = oracle()
    ctus = ((pic_w + 63) // 64) * ((pic_h + 63) // 64)
    n = ctus - ctu_first if ctu_count < 0 else ctu_count
    cs, rs = cur.shape[1], ref_plane.shape[1]
    pq = None
    if pred_q is not None:
        pred_q = np.ascontiguousarray(pred_q, dtype=np.int16)
        pq = pred_q.ctypes.data
    ox = oy = osad = None
    px = py = ps = None
    if want_results:
        ox = np.zeros((n, NUM_PARTS), np.int32); oy = np.zeros((n, NUM_PARTS), np.int32)
        osad = np.zeros((n, NUM_PARTS), np.uint32)
        px, py, ps = ox.ctypes.data, oy.ctypes.data, osad.ctypes.data
    probes, s4 = C.c_long(), C.c_double()
    L.hmo_tz_frame(_addr(cur, origin[1] * cs + origin[0]), cs, _addr(ref_plane, origin[1] * rs + origin[0]), rs,
                   pic_w, pic_h, sr, pq, int(lambda_q16), int(fen), int(bit_depth), ctu_first, n, n_threads,
                   int(all_slots), px, py, ps, C.byref(probes), C.byref(s4))
    if want_results:
        return probes.value, s4.value, ox, oy, osad
    return probes.value, s4.value


def ref_tz_frame(cur, ref_plane, origin, pic_w, pic_h, sr, lam, fen, bit_depth, ctu_first, ctu_count):
    """the REFERENCE's own xTZSearch over all 593 PU shapes of a CTU range (oracle/_ref/libhmref.so, single-threaded like HM);
    predictor (0,0).  -> (x, y, sad) int arrays [count, 593]"""
    L = ref()
    ox = np.zeros((ctu_count, NUM_PARTS), np.int32)
    oy = np.zeros((ctu_count, NUM_PARTS), np.int32)
    osad = np.zeros((ctu_count, NUM_PARTS), np.uint32)
    cs, rs = cur.shape[1], ref_plane.shape[1]
    rects = np.ascontiguousarray(slot_table().astype(np.int32))
    L.ref_tz_frame(_addr(cur, origin[1] * cs + origin[0]), cs, _addr(ref_plane, origin[1] * rs + origin[0]), rs, pic_w, pic_h, sr, float(lam),
                   int(fen), int(bit_depth), ctu_first, ctu_count, rects.reshape(-1), ox.reshape(-1), oy.reshape(-1), osad.reshape(-1))
    return ox, oy, osad


def frac_refine(plane_cur, cur_xy, plane_ref, ref_xy, w, h, int_mv, pred, lam_or_q16, use_had, bit_depth, use_ref=False, bi=False):
    """xPatternSearchFracDIF for one PU: the oracle (lam_or_q16 = lambda_q16) or the reference (lam_or_q16 = lambda; bi: the call is
    made with biPred = true, as for a bi-prediction origin).  -> (half_x, half_y, qter_x, qter_y, cost)"""
    cs, rs = plane_cur.shape[1], plane_ref.shape[1]
    org = _addr(plane_cur, cur_xy[1] * cs + cur_xy[0])
    rf = _addr(plane_ref, ref_xy[1] * rs + ref_xy[0])
    o = [C.c_int() for _ in range(4)]
    cost = C.c_uint32()
    if use_ref:
        fn = ref().ref_frac_refine_bi if bi else ref().ref_frac_refine
        fn(org, cs, w, h, rf, rs, int_mv[0], int_mv[1], pred[0], pred[1], float(lam_or_q16), int(use_had),
           bit_depth, *[C.byref(v) for v in o], C.byref(cost))
    else:
        oracle().hmo_frac_refine(org, cs, w, h, rf, rs, int_mv[0], int_mv[1], pred[0], pred[1], int(lam_or_q16), int(use_had),
                                 bit_depth, *[C.byref(v) for v in o], C.byref(cost))
    return tuple(v.value for v in o) + (cost.value,)


def frac_refine_w(plane_cur, cur_xy, plane_ref, ref_xy, w, h, int_mv, pred, lam_or_q16, use_had, bit_depth, wp, use_ref=False):
    """xPatternSearchFracDIF for one PU in a slice with explicit weighted prediction wp = (w0, offset, shift, round): the oracle
    (lam_or_q16 = lambda_q16) or the reference (lam_or_q16 = lambda).  -> (half_x, half_y, qter_x, qter_y, cost)"""
    cs, rs = plane_cur.shape[1], plane_ref.shape[1]
    org = _addr(plane_cur, cur_xy[1] * cs + cur_xy[0])
    rf = _addr(plane_ref, ref_xy[1] * rs + ref_xy[0])
    o = [C.c_int() for _ in range(4)]
    cost = C.c_uint32()
    if use_ref:
        ref().ref_frac_refine_w(org, cs, w, h, rf, rs, int_mv[0], int_mv[1], pred[0], pred[1], float(lam_or_q16), int(use_had), bit_depth,
                                *[int(v) for v in wp], *[C.byref(v) for v in o], C.byref(cost))
    else:
        ww = Wp(*[int(v) for v in wp])
        oracle().hmo_frac_refine_w(org, cs, w, h, rf, rs, int_mv[0], int_mv[1], pred[0], pred[1], int(lam_or_q16), int(use_had), bit_depth,
                                   C.byref(ww), *[C.byref(v) for v in o], C.byref(cost))
    return tuple(v.value for v in o) + (cost.value,)


def ref_full_search_ctus(cur, ref_plane, origin, pic_w, pic_h, sr, lam, fen, bit_depth, ctu_first, min_ctus=4, max_ctus=16, budget_s=3.0):
    """the REFERENCE's own exhaustive search (TEncSearch::xPatternSearch, TEncSearch.cpp:3835-3897, in oracle/_ref/libhmref.so) for
    every one of the 593 PU rectangles of consecutive CTUs starting at `ctu_first`, one PU at a time on ONE thread as HM runs it, with the
    CTU's window (xSetSearchRange around predictor (0,0)) shared by all its PUs.  CTUs are added until `budget_s`
    seconds have passed (at least min_ctus, at most max_ctus).  -> (x, y, sad) int arrays [n, 593], seconds"""
    import time
    L = ref()
    table = slot_table()
    cs, rs = cur.shape[1], ref_plane.shape[1]
    ctus_x = (pic_w + 63) // 64
    xs, ys, sads = [], [], []
    lt = [C.c_int() for _ in range(4)]
    mx, my, sad = C.c_int(), C.c_int(), C.c_uint32()
    t0 = time.perf_counter()
    n = 0
    while n < max_ctus and (n < min_ctus or time.perf_counter() - t0 < budget_s):
        ctu = ctu_first + n
        cx, cy = (ctu % ctus_x) * 64, (ctu // ctus_x) * 64
        oracle().hmo_set_search_range(0, 0, sr, cx, cy, pic_w, pic_h, 64, *[C.byref(v) for v in lt])
        ox, oy, os_ = np.zeros(NUM_PARTS, np.int32), np.zeros(NUM_PARTS, np.int32), np.zeros(NUM_PARTS, np.uint32)
        for s in range(NUM_PARTS):
            x, y, bw, bh = (int(v) for v in table[s])
            L.ref_pattern_search(_addr(cur, (origin[1] + cy + y) * cs + origin[0] + cx + x), cs, bw, bh,
                                 _addr(ref_plane, (origin[1] + cy + y) * rs + origin[0] + cx + x), rs,
                                 lt[0].value, lt[1].value, lt[2].value, lt[3].value, 0, 0, float(lam), int(fen), int(bit_depth),
                                 C.byref(mx), C.byref(my), C.byref(sad))
            ox[s], oy[s], os_[s] = mx.value, my.value, sad.value
        xs.append(ox); ys.append(oy); sads.append(os_)
        n += 1
    return np.stack(xs), np.stack(ys), np.stack(sads), time.perf_counter() - t0
