"""ctypes bindings onto the C ABI (include/hmme.h) of libhmme.so -- the HIP engine.

There is no CPU fallback: `load()` raises if the library is missing, and every call that needs
the GPU raises HmmeError carrying the engine's own error text.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIB_PATH = os.environ.get("HMME_LIB", os.path.join(CSRC, "libhmme.so"))   # HMME_LIB: A/B builds of the kernel
NUM_PARTS = 593

# every symbol include/hmme.h declares (tests check the library exports all of them)
SYMBOLS = ["hmme_create", "hmme_destroy", "hmme_last_error", "hmme_device_info", "hmme_set_lambda",
           "hmme_set_lambda_q16", "hmme_get_lambda_q16", "hmme_params_ocl_compat", "hmme_set_search_range",
           "hmme_slot_index", "hmme_slot_rect", "hmme_slot_index_amp_off", "hmme_amp_off_slot", "hmme_compact_amp_off", "hmme_search_ctu", "hmme_search_ctu_w", "hmme_search_refine_ctu_w", "hmme_refine_ctu_w", "hmme_search_refine_ctu", "hmme_refine_ctu", "hmme_plane_create", "hmme_plane_create_ex", "hmme_plane_bit_depth", "hmme_plane_destroy", "hmme_plane_upload_pel",
           "hmme_plane_upload_u8", "hmme_host_register", "hmme_host_unregister", "hmme_plane_set_device_u8", "hmme_plane_width", "hmme_plane_height",
           "hmme_num_ctus", "hmme_search_frame", "hmme_search_frame_device", "hmme_search_frame_multi",
           "hmme_search_frame_multi_device", "hmme_refine_frame", "hmme_refine_frame_multi_device",
           "hmme_search_pairs_device", "hmme_refine_pairs_device", "hmme_plane_upload_async",
           "hmme_upload_status", "hmme_abi_version", "hmme_build_id", "hmme_device_index", "hmme_set_error_printing"]
# test / measurement entry points (include/hmme_test.h): not part of the boundary
TEST_SYMBOLS = ["hmme_test_time_search_kernel", "hmme_test_device_address", "hmme_test_frac_deal", "hmme_test_tail_plan"]
ABI_VERSION = 6   # HMME_ABI_VERSION of the include/hmme.h these bindings were written against


class HmmeError(RuntimeError):
    pass


class SearchParams(C.Structure):
    _fields_ = [("lt_x", C.c_int), ("lt_y", C.c_int), ("rb_x", C.c_int), ("rb_y", C.c_int),
                ("pred_x", C.c_int), ("pred_y", C.c_int), ("fen", C.c_int), ("bit_depth", C.c_int),
                ("shift_free", C.c_int)]


class Weight(C.Structure):
    """hmme_weight: luma WPScalingParam of the reference picture"""
    _fields_ = [("w0", C.c_int), ("offset", C.c_int), ("shift", C.c_int), ("round", C.c_int)]


class FrameParams(C.Structure):
    _fields_ = [("search_range", C.c_int), ("fen", C.c_int), ("bit_depth", C.c_int),
                ("ctu_first", C.c_int), ("ctu_count", C.c_int)]


def build():
    """compile libhmme.so for gfx950 (hipcc cross-compiles without a GPU)"""
    subprocess.run(["make", "-s", "-C", CSRC], check=True)


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HmmeError(f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (python __graft_entry__.py). "
                        "The engine has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i = C.c_void_p, C.c_int
    if not hasattr(L, "hmme_abi_version") or L.hmme_abi_version() != ABI_VERSION:
        raise HmmeError(f"{LIB_PATH}: ABI version {L.hmme_abi_version() if hasattr(L, 'hmme_abi_version') else '< 3'}, "
                        f"these bindings need {ABI_VERSION}: rebuild with `make -C {CSRC}`")
    L.hmme_build_id.restype = C.c_char_p
    L.hmme_create.argtypes = [i, i, C.c_uint, C.POINTER(vp)]
    L.hmme_destroy.argtypes = [vp]
    L.hmme_destroy.restype = None
    L.hmme_last_error.argtypes = [vp]
    L.hmme_last_error.restype = C.c_char_p
    L.hmme_set_error_printing.argtypes = [vp, i]
    L.hmme_set_error_printing.restype = i
    L.hmme_device_info.argtypes = [vp]
    L.hmme_device_info.restype = C.c_char_p
    L.hmme_set_lambda.argtypes = [vp, C.c_double]
    L.hmme_set_lambda_q16.argtypes = [vp, C.c_uint32]
    L.hmme_get_lambda_q16.argtypes = [vp]
    L.hmme_get_lambda_q16.restype = C.c_uint32
    L.hmme_params_ocl_compat.argtypes = [C.POINTER(SearchParams), i, i, i]
    L.hmme_params_ocl_compat.restype = None
    L.hmme_set_search_range.argtypes = [i] * 7 + [C.POINTER(i)] * 4
    L.hmme_set_search_range.restype = None
    L.hmme_slot_index.argtypes = [i, i, i, i]
    L.hmme_slot_rect.argtypes = [i] + [C.POINTER(i)] * 4
    L.hmme_search_ctu.argtypes = [vp, vp, i, vp, i, C.POINTER(SearchParams), vp, vp]
    L.hmme_search_ctu_w.argtypes = [vp, vp, i, vp, i, C.POINTER(SearchParams), C.POINTER(Weight), vp, vp]
    L.hmme_search_refine_ctu_w.argtypes = [vp, vp, i, vp, i, C.POINTER(SearchParams), C.POINTER(Weight), i, vp, vp, vp, vp]
    L.hmme_refine_ctu_w.argtypes = [vp, vp, i, vp, i, C.POINTER(SearchParams), C.POINTER(Weight), vp, i, vp, vp]
    L.hmme_search_refine_ctu.argtypes = [vp, vp, i, vp, i, C.POINTER(SearchParams), i, vp, vp, vp, vp]
    L.hmme_refine_ctu.argtypes = [vp, vp, i, vp, i, C.POINTER(SearchParams), vp, i, vp, vp]
    L.hmme_plane_create.argtypes = [vp, i, i, C.POINTER(vp)]
    L.hmme_plane_create_ex.argtypes = [vp, i, i, i, C.POINTER(vp)]
    L.hmme_plane_bit_depth.argtypes = [vp]
    L.hmme_plane_destroy.argtypes = [vp]
    L.hmme_plane_destroy.restype = None
    L.hmme_plane_upload_pel.argtypes = [vp, vp, i]
    L.hmme_plane_upload_u8.argtypes = [vp, vp, i]
    L.hmme_host_register.argtypes = [vp, vp, C.c_size_t]
    L.hmme_host_unregister.argtypes = [vp, vp]
    L.hmme_plane_set_device_u8.argtypes = [vp, vp, i, vp]
    L.hmme_plane_width.argtypes = [vp]
    L.hmme_plane_height.argtypes = [vp]
    L.hmme_num_ctus.argtypes = [i, i]
    L.hmme_search_frame.argtypes = [vp, vp, vp, C.POINTER(FrameParams), vp, vp, vp]
    L.hmme_search_frame_device.argtypes = [vp, vp, vp, C.POINTER(FrameParams), vp, vp, vp, vp]
    L.hmme_search_frame_multi.argtypes = [vp, vp, C.POINTER(vp), i, C.POINTER(FrameParams), vp, vp, vp]
    L.hmme_search_frame_multi_device.argtypes = [vp, vp, C.POINTER(vp), i, C.POINTER(FrameParams), vp, vp, vp, vp]
    L.hmme_refine_frame.argtypes = [vp, vp, vp, C.POINTER(FrameParams), vp, vp, i, vp, vp]
    L.hmme_refine_frame_multi_device.argtypes = [vp, vp, C.POINTER(vp), i, C.POINTER(FrameParams), vp, vp, i, vp, vp, vp]
    L.hmme_test_time_search_kernel.argtypes = [vp, vp, vp, C.POINTER(FrameParams), vp, vp, vp, vp, i, C.POINTER(C.c_float)]
    L.hmme_search_pairs_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), i, C.POINTER(FrameParams), vp, vp, vp, vp]
    L.hmme_refine_pairs_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), i, C.POINTER(FrameParams), vp, vp, i, vp, vp, vp]
    L.hmme_plane_upload_async.argtypes = [vp, vp, i, i, vp]
    L.hmme_upload_status.argtypes = [vp, vp]
    L.hmme_test_device_address.argtypes = [vp, vp]
    L.hmme_test_device_address.restype = C.c_uint64
    _lib = L
    return L


class Plane:
    def __init__(self, engine, width, height, bit_depth=8):
        self.engine = engine
        self.width, self.height, self.bit_depth = width, height, bit_depth
        h = C.c_void_p()
        engine._check(engine.L.hmme_plane_create_ex(engine.h, width, height, bit_depth, C.byref(h)))
        self.h = h

    def upload_pel(self, padded, origin):
        """padded: 2-D int16 HM plane, origin = (x, y) of sample (0,0) inside it"""
        a = np.ascontiguousarray(padded, dtype=np.int16)
        ptr = a.ctypes.data + 2 * (origin[1] * a.shape[1] + origin[0])
        self.engine._check(self.engine.L.hmme_plane_upload_pel(self.h, ptr, a.shape[1]))

    def upload_u8(self, img):
        a = np.ascontiguousarray(img, dtype=np.uint8)
        assert a.shape == (self.height, self.width)
        self.engine._check(self.engine.L.hmme_plane_upload_u8(self.h, a.ctypes.data, a.shape[1]))

    def upload_async(self, host_ptr, stride, sample_bytes, stream):
        """asynchronous upload from (page-locked) host memory on `stream`; host_ptr: address of sample (0,0)"""
        self.engine._check(self.engine.L.hmme_plane_upload_async(self.h, host_ptr, stride, sample_bytes, stream))

    @property
    def device_address(self):
        return int(self.engine.L.hmme_test_device_address(self.engine.h, self.h))

    def set_device_u8(self, dptr, pitch, stream=0):
        self.engine._check(self.engine.L.hmme_plane_set_device_u8(self.h, dptr, pitch, stream))

    def close(self):
        if self.h:
            self.engine.L.hmme_plane_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Engine:
    """one context = one GPU (reference: one TEncOpenCL object, TEncTop.h:82)"""

    def __init__(self, device=0, sr_max=64):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.hmme_create(device, sr_max, 0, C.byref(h))
        if rc != 0:
            raise HmmeError(f"hmme_create failed ({rc}): {self.L.hmme_last_error(None).decode()}")
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise HmmeError(f"hmme error {rc}: {self.L.hmme_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.L.hmme_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def device_info(self):
        return self.L.hmme_device_info(self.h).decode()

    def set_lambda(self, lam):
        self._check(self.L.hmme_set_lambda(self.h, float(lam)))

    def set_lambda_q16(self, q):
        self._check(self.L.hmme_set_lambda_q16(self.h, int(q)))

    @property
    def lambda_q16(self):
        return int(self.L.hmme_get_lambda_q16(self.h))

    def plane(self, width, height, bit_depth=8):
        return Plane(self, width, height, bit_depth)

    def host_register(self, array):
        """page-lock a long-lived numpy buffer: uploads from it then run at PCIe rate (hmme_host_register)"""
        self._check(self.L.hmme_host_register(self.h, array.ctypes.data, array.nbytes))

    def host_unregister(self, array):
        self._check(self.L.hmme_host_unregister(self.h, array.ctypes.data))

    def search_ctu(self, cur_plane, cur_xy, ref_plane, ref_xy, params):
        """per-CTU drop-in (calcMotionVectors).  planes: 2-D int16; *_xy = CTU origin inside them.
        -> (mv int16[593,2], sad uint32[593])"""
        cur = np.ascontiguousarray(cur_plane, dtype=np.int16)
        ref = np.ascontiguousarray(ref_plane, dtype=np.int16)
        mv = np.zeros((NUM_PARTS, 2), np.int16)
        sad = np.zeros(NUM_PARTS, np.uint32)
        cp = cur.ctypes.data + 2 * (cur_xy[1] * cur.shape[1] + cur_xy[0])
        rp = ref.ctypes.data + 2 * (ref_xy[1] * ref.shape[1] + ref_xy[0])
        self._check(self.L.hmme_search_ctu(self.h, cp, cur.shape[1], rp, ref.shape[1], C.byref(params),
                                           mv.ctypes.data, sad.ctypes.data))
        return mv, sad

    def search_ctu_w(self, cur_plane, cur_xy, ref_plane, ref_xy, params, wp):
        """hmme_search_ctu_w: the per-CTU search of a slice with explicit weighted prediction; wp = (w0, offset, shift, round)"""
        cur = np.ascontiguousarray(cur_plane, dtype=np.int16)
        ref = np.ascontiguousarray(ref_plane, dtype=np.int16)
        mv = np.zeros((NUM_PARTS, 2), np.int16)
        sad = np.zeros(NUM_PARTS, np.uint32)
        cp = cur.ctypes.data + 2 * (cur_xy[1] * cur.shape[1] + cur_xy[0])
        rp = ref.ctypes.data + 2 * (ref_xy[1] * ref.shape[1] + ref_xy[0])
        w = Weight(*[int(v) for v in wp])
        self._check(self.L.hmme_search_ctu_w(self.h, cp, cur.shape[1], rp, ref.shape[1], C.byref(params), C.byref(w), mv.ctypes.data, sad.ctypes.data))
        return mv, sad

    def search_refine_ctu_w(self, cur_plane, cur_xy, ref_plane, ref_xy, params, wp, use_hadamard=True):
        """hmme_search_refine_ctu_w: weighted integer search + weighted xPatternSearchFracDIF in one call -> (mv, sad, qmv, cost)"""
        cur = np.ascontiguousarray(cur_plane, dtype=np.int16)
        ref = np.ascontiguousarray(ref_plane, dtype=np.int16)
        mv, qmv = np.zeros((NUM_PARTS, 2), np.int16), np.zeros((NUM_PARTS, 2), np.int16)
        sad, cost = np.zeros(NUM_PARTS, np.uint32), np.zeros(NUM_PARTS, np.uint32)
        cp = cur.ctypes.data + 2 * (cur_xy[1] * cur.shape[1] + cur_xy[0])
        rp = ref.ctypes.data + 2 * (ref_xy[1] * ref.shape[1] + ref_xy[0])
        w = Weight(*[int(v) for v in wp])
        self._check(self.L.hmme_search_refine_ctu_w(self.h, cp, cur.shape[1], rp, ref.shape[1], C.byref(params), C.byref(w), 1 if use_hadamard else 0,
                                                    mv.ctypes.data, sad.ctypes.data, qmv.ctypes.data, cost.ctypes.data))
        return mv, sad, qmv, cost

    def search_refine_ctu(self, cur_plane, cur_xy, ref_plane, ref_xy, params, use_hadamard=True):
        """hmme_search_ctu + xPatternSearchFracDIF of its winners in one call
        -> (mv int16[593,2], sad uint32[593], qmv int16[593,2] quarter-pel, cost uint32[593])"""
        cur = np.ascontiguousarray(cur_plane, dtype=np.int16)
        ref = np.ascontiguousarray(ref_plane, dtype=np.int16)
        mv, qmv = np.zeros((NUM_PARTS, 2), np.int16), np.zeros((NUM_PARTS, 2), np.int16)
        sad, cost = np.zeros(NUM_PARTS, np.uint32), np.zeros(NUM_PARTS, np.uint32)
        cp = cur.ctypes.data + 2 * (cur_xy[1] * cur.shape[1] + cur_xy[0])
        rp = ref.ctypes.data + 2 * (ref_xy[1] * ref.shape[1] + ref_xy[0])
        self._check(self.L.hmme_search_refine_ctu(self.h, cp, cur.shape[1], rp, ref.shape[1], C.byref(params), int(use_hadamard),
                                                  mv.ctypes.data, sad.ctypes.data, qmv.ctypes.data, cost.ctypes.data))
        return mv, sad, qmv, cost

    def refine_ctu(self, cur_plane, cur_xy, ref_plane, ref_xy, params, int_mv, use_hadamard=True):
        """xPatternSearchFracDIF for the 593 slots of one CTU at the caller's integer MVs -> (qmv int16[593,2], cost uint32[593])"""
        cur = np.ascontiguousarray(cur_plane, dtype=np.int16)
        ref = np.ascontiguousarray(ref_plane, dtype=np.int16)
        imv = np.ascontiguousarray(int_mv, dtype=np.int16)
        assert imv.shape == (NUM_PARTS, 2)
        qmv, cost = np.zeros((NUM_PARTS, 2), np.int16), np.zeros(NUM_PARTS, np.uint32)
        cp = cur.ctypes.data + 2 * (cur_xy[1] * cur.shape[1] + cur_xy[0])
        rp = ref.ctypes.data + 2 * (ref_xy[1] * ref.shape[1] + ref_xy[0])
        self._check(self.L.hmme_refine_ctu(self.h, cp, cur.shape[1], rp, ref.shape[1], C.byref(params), imv.ctypes.data, int(use_hadamard),
                                           qmv.ctypes.data, cost.ctypes.data))
        return qmv, cost

    def refine_ctu_w(self, cur_plane, cur_xy, ref_plane, ref_xy, params, wp, int_mv, use_hadamard=True):
        """hmme_refine_ctu_w: the weighted xPatternSearchFracDIF alone, at the caller's integer MVs -> (qmv, cost)"""
        cur = np.ascontiguousarray(cur_plane, dtype=np.int16)
        ref = np.ascontiguousarray(ref_plane, dtype=np.int16)
        imv = np.ascontiguousarray(int_mv, dtype=np.int16)
        assert imv.shape == (NUM_PARTS, 2)
        qmv, cost = np.zeros((NUM_PARTS, 2), np.int16), np.zeros(NUM_PARTS, np.uint32)
        cp = cur.ctypes.data + 2 * (cur_xy[1] * cur.shape[1] + cur_xy[0])
        rp = ref.ctypes.data + 2 * (ref_xy[1] * ref.shape[1] + ref_xy[0])
        w = Weight(*[int(v) for v in wp])
        self._check(self.L.hmme_refine_ctu_w(self.h, cp, cur.shape[1], rp, ref.shape[1], C.byref(params), C.byref(w), imv.ctypes.data, int(use_hadamard),
                                             qmv.ctypes.data, cost.ctypes.data))
        return qmv, cost

    def search_frame(self, cur, ref, sr, pred_q=None, fen=1, bit_depth=None, ctu_first=0, ctu_count=-1):
        """-> (mv int16[count,593,2], sad uint32[count,593])"""
        bit_depth = cur.bit_depth if bit_depth is None else bit_depth
        n = self.L.hmme_num_ctus(cur.width, cur.height)
        count = n - ctu_first if ctu_count < 0 else ctu_count
        fp = FrameParams(sr, int(fen), bit_depth, ctu_first, count)
        mv = np.zeros((count, NUM_PARTS, 2), np.int16)
        sad = np.zeros((count, NUM_PARTS), np.uint32)
        pq = None
        if pred_q is not None:
            pred_q = np.ascontiguousarray(pred_q, dtype=np.int16)
            assert pred_q.shape == (n, 2)
            pq = pred_q.ctypes.data
        self._check(self.L.hmme_search_frame(self.h, cur.h, ref.h, C.byref(fp), pq, mv.ctypes.data, sad.ctypes.data))
        return mv, sad

    def search_frame_multi(self, cur, refs, sr, pred_q=None, fen=1, ctu_first=0, ctu_count=-1):
        """several reference pictures in one launch -> (mv int16[n_refs,count,593,2], sad uint32[n_refs,count,593])"""
        n = self.L.hmme_num_ctus(cur.width, cur.height)
        count = n - ctu_first if ctu_count < 0 else ctu_count
        fp = FrameParams(sr, int(fen), cur.bit_depth, ctu_first, count)
        mv = np.zeros((len(refs), count, NUM_PARTS, 2), np.int16)
        sad = np.zeros((len(refs), count, NUM_PARTS), np.uint32)
        pq = None
        if pred_q is not None:
            pred_q = np.ascontiguousarray(pred_q, dtype=np.int16)
            assert pred_q.shape == (len(refs), n, 2)
            pq = pred_q.ctypes.data
        arr = (C.c_void_p * len(refs))(*[r.h for r in refs])
        self._check(self.L.hmme_search_frame_multi(self.h, cur.h, arr, len(refs), C.byref(fp), pq, mv.ctypes.data, sad.ctypes.data))
        return mv, sad

    def refine_frame(self, cur, ref, sr, int_mv, pred_q=None, use_hadamard=True, ctu_first=0, ctu_count=-1):
        """fractional-pel refinement of integer winners -> (qmv int16[count,593,2] quarter-pel, cost uint32[count,593])"""
        n = self.L.hmme_num_ctus(cur.width, cur.height)
        count = n - ctu_first if ctu_count < 0 else ctu_count
        fp = FrameParams(sr, 1, cur.bit_depth, ctu_first, count)
        int_mv = np.ascontiguousarray(int_mv, dtype=np.int16)
        assert int_mv.shape == (count, NUM_PARTS, 2)
        qmv = np.zeros((count, NUM_PARTS, 2), np.int16)
        cost = np.zeros((count, NUM_PARTS), np.uint32)
        pq = None
        if pred_q is not None:
            pred_q = np.ascontiguousarray(pred_q, dtype=np.int16)
            pq = pred_q.ctypes.data
        self._check(self.L.hmme_refine_frame(self.h, cur.h, ref.h, C.byref(fp), pq, int_mv.ctypes.data, int(use_hadamard),
                                             qmv.ctypes.data, cost.ctypes.data))
        return qmv, cost

    def refine_frame_multi_device(self, cur, refs, fp, d_pred, d_int_mv, use_hadamard, d_qmv, d_cost, stream=0):
        arr = (C.c_void_p * len(refs))(*[r.h for r in refs])
        self._check(self.L.hmme_refine_frame_multi_device(self.h, cur.h, arr, len(refs), C.byref(fp), d_pred, d_int_mv,
                                                          int(use_hadamard), d_qmv, d_cost, stream))

    def search_frame_multi_device(self, cur, refs, fp, d_pred, d_mv, d_sad, stream=0):
        arr = (C.c_void_p * len(refs))(*[r.h for r in refs])
        self._check(self.L.hmme_search_frame_multi_device(self.h, cur.h, arr, len(refs), C.byref(fp), d_pred, d_mv, d_sad, stream))

    def search_pairs_device(self, curs, refs, fp, d_pred, d_mv, d_sad, stream=0):
        """up to 16 (current, reference) picture pairs of one size in one launch (hmme_search_pairs_device)"""
        assert len(curs) == len(refs)
        ca = (C.c_void_p * len(curs))(*[c.h for c in curs])
        ra = (C.c_void_p * len(refs))(*[r.h for r in refs])
        self._check(self.L.hmme_search_pairs_device(self.h, ca, ra, len(refs), C.byref(fp), d_pred, d_mv, d_sad, stream))

    def refine_pairs_device(self, curs, refs, fp, d_pred, d_int_mv, use_hadamard, d_qmv, d_cost, stream=0):
        assert len(curs) == len(refs)
        ca = (C.c_void_p * len(curs))(*[c.h for c in curs])
        ra = (C.c_void_p * len(refs))(*[r.h for r in refs])
        self._check(self.L.hmme_refine_pairs_device(self.h, ca, ra, len(refs), C.byref(fp), d_pred, d_int_mv, int(use_hadamard),
                                                    d_qmv, d_cost, stream))

    def upload_status(self, stream=0):
        """waits for `stream`; raises if an asynchronous upload carried an out-of-range sample"""
        self._check(self.L.hmme_upload_status(self.h, stream))

    @property
    def call_block_address(self):
        """device address of the per-CTU call's current-block staging area (high-address test)"""
        return int(self.L.hmme_test_device_address(self.h, None))

    def search_frame_device(self, cur, ref, fp, d_pred, d_mv, d_sad, stream=0):
        self._check(self.L.hmme_search_frame_device(self.h, cur.h, ref.h, C.byref(fp), d_pred, d_mv, d_sad, stream))

    def time_search_kernel(self, cur, ref, fp, d_pred, d_mv, d_sad, stream=0, reps=3):
        ms = C.c_float()
        self._check(self.L.hmme_test_time_search_kernel(self.h, cur.h, ref.h, C.byref(fp), d_pred, d_mv, d_sad, stream, reps,
                                                   C.byref(ms)))
        return float(ms.value)


def build_id():
    """identifies the kernel sources + flags the loaded library was built from (hmme_build_id)"""
    return load().hmme_build_id().decode()


def ocl_compat_params(lt_x, lt_y, sr):
    p = SearchParams()
    load().hmme_params_ocl_compat(C.byref(p), lt_x, lt_y, sr)
    return p


def set_search_range(pred_x_q, pred_y_q, sr, cu_x, cu_y, pic_w, pic_h):
    out = [C.c_int() for _ in range(4)]
    load().hmme_set_search_range(pred_x_q, pred_y_q, sr, cu_x, cu_y, pic_w, pic_h, *[C.byref(o) for o in out])
    return tuple(o.value for o in out)


def slot_index(part_size, depth, part_idx, abs_z_idx):
    return load().hmme_slot_index(part_size, depth, part_idx, abs_z_idx)


def slot_rect(slot):
    out = [C.c_int() for _ in range(4)]
    rc = load().hmme_slot_rect(slot, *[C.byref(o) for o in out])
    if rc != 0:
        raise HmmeError(f"slot {slot} out of range")
    return tuple(o.value for o in out)
