"""CPU-side checks of the drop-in boundary: the HIP library builds, loads and exports every
symbol include/hmme.h declares; host-side helpers that need no GPU agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np

from conftest import GOLDEN, ROOT


def test_library_exports_every_declared_symbol():
    from hmme import api
    api.build()
    L = api.load()
    header = open(os.path.join(ROOT, "include", "hmme.h")).read()
    declared = sorted(set(re.findall(r"\b(hmme_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(api.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name


def test_host_search_range_matches_reference_goldens():
    # reference: TEncSearch::xSetSearchRange + TComDataCU::clipMv (golden rows from the compiled reference)
    from hmme import api
    api.build()
    for r in np.load(os.path.join(GOLDEN, "range.npz"))["rows"]:
        px, py, sr, cu_x, cu_y, pw, ph, max_cu = (int(v) for v in r[:8])
        assert max_cu == 64
        assert api.set_search_range(px, py, sr, cu_x, cu_y, pw, ph) == tuple(int(v) for v in r[8:12])


def test_ocl_compat_preset():
    # reference: TEncOpenCL.cpp:312-313 (x,y in [0,2*SR]), cl/sad.cl:374-398 (pred 0, all rows)
    from hmme import api
    api.build()
    p = api.ocl_compat_params(-13, 5, 8)
    assert (p.lt_x, p.lt_y, p.rb_x, p.rb_y, p.pred_x, p.pred_y, p.fen, p.bit_depth) == (-13, 5, 3, 21, 0, 0, 0, 8)


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    from hmme import api
    api.build()
    try:
        api.Engine(0)
    except api.HmmeError as e:
        assert "no HIP device" in str(e) or "hmme_create failed" in str(e)
    else:
        raise AssertionError("Engine() must not succeed without a GPU")


def test_host_module_compiles_inside_the_reference_tree():
    """drop-in check (compile only, needs /root/reference): TEncOpenCL.{h,cpp} build against HM's own
    TypeDef.h / TComMv.h in place of the reference's files, with HM's C++98 dialect"""
    import subprocess
    import pytest
    ref = "/root/reference/source/Lib"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present")
    host = os.path.join(ROOT, "hm-opencl_amd", "host")
    subprocess.run(["g++", "-std=gnu++98", "-fsyntax-only", "-DHMME_IN_HM_TREE", "-DMSYS_LINUX", "-I" + ref,
                    os.path.join(host, "TEncOpenCL.cpp")], check=True)


def test_slot_layout_matches_reference_getIndexBlock(slots):
    # goldens: TComDataCU::getIndexBlock evaluated by the compiled reference for every tabulated PU
    from hmme import api
    api.build()
    for row in slots:
        slot, ps, depth, pi, z, s, x, y, w, h = (int(v) for v in row)
        assert api.slot_index(ps, depth, pi, z) == slot
        assert api.slot_rect(slot) == (x, y, w, h)
    assert api.slot_index(3, 3, 0, 0) == -1 and api.slot_index(4, 3, 0, 0) == -1 and api.slot_index(0, 1, 0, 1) == -1


def test_generated_isa_has_no_dpp_hazard():
    """the unpadded DPP merges of me_search_kernel rely on instruction spacing the generator arranges; the final ISA
    (hipcc cross-compiles gfx950 here) must not have a DPP read within 2 wait states of the write of its source"""
    import shutil
    import subprocess
    from conftest import ROOT
    if not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "hm-opencl_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "check-isa"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 hazard(s)" in r.stdout
