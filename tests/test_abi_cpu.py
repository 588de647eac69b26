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
    # measurement / test entry points are declared in a header of their own, not in the boundary
    test_header = open(os.path.join(ROOT, "include", "hmme_test.h")).read()
    test_declared = sorted(set(re.findall(r"\b(hmme_[a-z0-9_]+)\s*\(", test_header)))
    assert test_declared == sorted(api.TEST_SYMBOLS) and all(n.startswith("hmme_test_") for n in test_declared)
    for name in test_declared:
        assert hasattr(L, name), name
    assert not re.search(r"hmme_(test|debug|time)_[a-z0-9_]*\s*\(", header)      # mentioned in comments at most, never declared


def test_host_search_range_matches_reference_goldens():
    # reference: TEncSearch::xSetSearchRange + TComDataCU::clipMv (golden rows from the compiled reference)
    from hmme import api
    api.build()
    for r in np.load(os.path.join(GOLDEN, "range.npz"))["rows"]:
        px, py, sr, cu_x, cu_y, pw, ph, max_cu = (int(v) for v in r[:8])
        assert max_cu == 64
        assert api.set_search_range(px, py, sr, cu_x, cu_y, pw, ph) == tuple(int(v) for v in r[8:12])


def test_amp_off_table_view_matches_the_reference_table():
    """the 425-entry layout of a reference build with AMP_ENC_SPEEDUP (TComDataCU.cpp:3393-4675; tests/golden/slots_amp_off.npz holds the
    425 (key -> index) pairs of that switch): hmme_slot_index_amp_off reproduces it, incl. the reversed part order of the 64x64 2NxN /
    Nx2N CUs; every entry maps to the 593-layout slot with the same rectangle; compaction moves results accordingly (no GPU needed)"""
    from hmme import api
    api.build()
    L = api.load()
    L.hmme_slot_index_amp_off.argtypes = [C.c_int] * 4
    L.hmme_amp_off_slot.argtypes = [C.c_int]
    t = np.load(os.path.join(GOLDEN, "slots_amp_off.npz"))["table"]
    assert len(t) == 425
    seen = set()
    for idx, ps, depth, pi, z, h, w in (tuple(int(v) for v in r) for r in t):
        assert L.hmme_slot_index_amp_off(ps, depth, pi, z) == idx, (idx, ps, depth, pi, z)
        slot = L.hmme_amp_off_slot(idx)
        assert slot == api.slot_index(ps, depth, pi, z) and 0 <= slot < 593 and slot not in seen
        seen.add(slot)
        x, y, rw, rh = api.slot_rect(slot)
        assert h == w == 64 >> depth and (rw, rh) == ((w, h) if ps == 0 else ((w, h // 2) if ps == 1 else (w // 2, h)))
    assert L.hmme_slot_index_amp_off(1, 0, 1, 0) == 420 and L.hmme_slot_index_amp_off(1, 0, 0, 0) == 421      # part 1 before part 0 at 64x64
    for ps in (3, 4, 5, 6, 7):
        assert L.hmme_slot_index_amp_off(ps, 1, 0, 0) == -1                                                     # no NxN, no AMP shapes
    assert L.hmme_amp_off_slot(425) == -1 and L.hmme_amp_off_slot(-1) == -1
    mv = np.arange(593 * 2, dtype=np.int16).reshape(593, 2)
    sad = np.arange(593, dtype=np.uint32) * 7
    mv4, sad4 = np.zeros((425, 2), np.int16), np.zeros(425, np.uint32)
    L.hmme_compact_amp_off.argtypes = [C.c_void_p] * 4
    assert L.hmme_compact_amp_off(mv.ctypes.data, sad.ctypes.data, mv4.ctypes.data, sad4.ctypes.data) == 0
    for i in range(425):
        s_ = L.hmme_amp_off_slot(i)
        assert tuple(mv4[i]) == tuple(mv[s_]) and sad4[i] == sad[s_]


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
    assert "0 hazard(s)" in r.stdout and "0 SGPR(s) touched" in r.stdout and ", 0 out of range" in r.stdout


def test_dpp_hazard_checker_treats_branch_targets_as_joins(tmp_path):
    """tools/check_dpp_hazard.py: a DPP read right after a local label is a violation whatever the fall-through path did
    (another path into the join may have written the register one branch ago); one padded instruction later it is not"""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "check_dpp_hazard.py")

    def run(body):
        p = tmp_path / "k.s"
        p.write_text("kernel_a:\n" + body)
        return subprocess.run([sys.executable, tool, str(p)], capture_output=True, text=True)
    dpp = "\tv_min_u32_dpp v1, v2, v2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
    ok_linear = "\tv_add_u32 v2, v3, v4\n\ts_nop 1\n" + dpp
    assert run(ok_linear).returncode == 0
    assert run("\tv_add_u32 v2, v3, v4\n\ts_nop 0\n" + dpp).returncode == 1                       # one wait state only
    assert run("\tv_add_u32 v9, v3, v4\n\ts_nop 1\n.LBB0_3:\n" + dpp).returncode == 1             # join: unknown writer + branch
    assert run("\tv_add_u32 v9, v3, v4\n.LBB0_3:\n\ts_nop 0\n" + dpp).returncode == 0             # ... padded once: fine
    assert run("\tv_add_u32 v9, v3, v4\n.LBB0_3:\n\tv_mov_b32 v7, v8\n" + dpp).returncode == 0


def test_isa_checker_flags_sgprs_touched_while_a_scalar_load_is_pending(tmp_path):
    """tools/check_dpp_hazard.py, second check: the inline-asm s_load_dword* are invisible to the compiler; nothing may read or
    write their destination before the next `s_waitcnt lgkmcnt(0)` (scalar loads return out of order: lgkmcnt(N > 0) retires none)"""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "check_dpp_hazard.py")

    def run(body):
        p = tmp_path / "k.s"
        p.write_text("kernel_a:\n" + body)
        return subprocess.run([sys.executable, tool, str(p)], capture_output=True, text=True)
    ld = "\ts_load_dwordx2 s[8:9], s[4:5], s6 offset:16\n"
    assert run(ld + "\ts_waitcnt lgkmcnt(0)\n\tv_qsad_pk_u16_u8 v[0:1], v[2:3], s8, v[0:1]\n").returncode == 0
    assert run(ld + "\tv_qsad_pk_u16_u8 v[0:1], v[2:3], s8, v[0:1]\n\ts_waitcnt lgkmcnt(0)\n").returncode == 1       # stale read
    assert run(ld + "\ts_mul_i32 s9, s3, 7\n\ts_waitcnt lgkmcnt(0)\n").returncode == 1                              # destination reused
    assert run(ld + "\ts_waitcnt lgkmcnt(1)\n\ts_add_u32 s2, s8, 1\n").returncode == 1                              # lgkmcnt(1) retires nothing
    assert run(ld + "\ts_mul_i32 s6, s3, 7\n\ts_load_dwordx2 s[8:9], s[4:5], s6 offset:80\n\ts_waitcnt lgkmcnt(0)\n").returncode == 0   # prefetch pattern
    assert run(ld + "\ts_load_dwordx2 s[10:11], s[4:5], s8 offset:80\n\ts_waitcnt lgkmcnt(0)\n").returncode == 1     # pending register as offset


def test_isa_checker_flags_immediates_the_assembler_accepts_and_the_hardware_does_not(tmp_path):
    """tools/check_dpp_hazard.py, third check: round 5's wrong-result build asked v_lshl_add_u64 for a shift by 6 and 7 -- it assembles, the
    hardware shifts by 0..4 only.  The checker must flag that on the CPU (and the other hand-picked encodings whose fields have a range),
    and must pass the forms the tree uses"""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "check_dpp_hazard.py")

    def run(body):
        p = tmp_path / "k.s"
        p.write_text("kernel_a:\n" + body)
        return subprocess.run([sys.executable, tool, str(p)], capture_output=True, text=True)
    good = ("\tv_lshl_add_u64 v[6:7], v[6:7], 4, s[2:3]\n\tv_lshl_add_u64 v[0:1], v[0:1], 0, v[4:5]\n\tds_read2_b32 v[12:13], v132 offset0:98 offset1:255\n"
            "\tds_read_b32 v1, v2 offset:65532\n\tv_mad_u32_u16 v62, v20, s29, v109 op_sel:[1,0,0,0]\n\ts_setprio 3\n\ts_nop 15\n"
            "\tv_add_u32 v9, v3, v4\n\ts_nop 1\n\tv_min_u32_dpp v69, v27, v27 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0x3\n"
            "\tv_min_u32_dpp v1, v2, v2 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_load_dwordx2 s[8:9], s[4:5], 0xff8\n\ts_waitcnt lgkmcnt(0)\n"
            "\tv_alignbyte_b32 v1, v2, v3, 3\n\tv_alignbyte_b32 v1, v2, v3, v4\n")
    r = run(good)
    assert r.returncode == 0 and "0 out of range" in r.stdout, r.stdout
    for bad, why in (("\tv_lshl_add_u64 v[6:7], v[6:7], 6, s[2:3]\n", "shifts by 0..4 only"),            # round 5's build
                     ("\tv_lshl_add_u64 v[6:7], v[6:7], 7, v[2:3]\n", "shifts by 0..4 only"),
                     ("\tds_read2_b32 v[12:13], v132 offset0:98 offset1:256\n", "outside 0..255"),
                     ("\tds_write2_b64 v4, v[6:7], v[52:53] offset:8\n", "offset0 / offset1"),
                     ("\tv_min_u32_dpp v69, v27, v27 row_ror:16 row_mask:0xf bank_mask:0xf\n", "outside 1..15"),
                     ("\tv_min_u32_dpp v69, v27, v27 quad_perm:[1,0,4,2] row_mask:0xf bank_mask:0xf\n", "four digits 0..3"),
                     ("\tv_min_u32_dpp v69, v27, v27 row_share:3 row_mask:0xf bank_mask:0xf\n", "not a DPP control"),
                     ("\tv_min_u32_dpp v69, v27, v27 row_ror:8 row_mask:0x1f bank_mask:0xf\n", "outside 0..0xf"),
                     ("\tv_mov_b32_dpp v1, v2 row_bcast:16 row_mask:0xf bank_mask:0xf\n", "only 15 and 31"),
                     ("\tv_mad_u32_u16 v62, v20, s29, v109 op_sel:[2,0,0,0]\n", "0 or 1"),
                     ("\ts_setprio 4\n", "outside 0..3"), ("\ts_nop 16\n", "outside 0..15"),
                     ("\ts_load_dwordx2 s[8:9], s[4:5], s6 offset:-8\n\ts_waitcnt lgkmcnt(0)\n", "scalar-load offset"),
                     ("\tv_alignbyte_b32 v1, v2, v3, 4\n", "outside 0..3")):
        r = run(good + bad)
        assert r.returncode == 1 and why in r.stdout and "1 out of range" in r.stdout, (bad, r.stdout)


def test_cpu_side_code_is_clean_under_asan_and_ubsan(tmp_path):
    """SURVEY 5 ("-fsanitize=address on host code"): the oracle and the TEncOpenCL host module + the GPU-free C-ABI entry points,
    compiled with -fsanitize=address,undefined and driven by tests/cpp/asan_driver.cpp.  GPU sanitizers are not available on
    this pool; here (no GPU) the driver also walks the failure paths of the class."""
    import subprocess
    import torch
    from hmme import api
    api.build()
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-build check; on the GPU box the class is covered by tests/cpp/test_tencopencl.cpp")
    exe = str(tmp_path / "asan_driver")
    csrc = os.path.join(ROOT, "hm-opencl_amd", "csrc")
    subprocess.run(["g++", "-O1", "-g", "-std=c++11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                    "-o", exe, os.path.join(ROOT, "tests", "cpp", "asan_driver.cpp"), os.path.join(ROOT, "hm-opencl_amd", "host", "TEncOpenCL.cpp"),
                    "-x", "c", os.path.join(ROOT, "oracle", "hm_oracle.c"), "-x", "none",
                    "-L" + csrc, "-lhmme", "-Wl,-rpath," + csrc, "-lm", "-lpthread"], check=True)
    supp = tmp_path / "lsan.supp"
    supp.write_text("leak:libamdhip64\nleak:libhsa-runtime64\nleak:libamd_comgr\nleak:librccl\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", LSAN_OPTIONS=f"suppressions={supp}:print_suppressions=0",
               UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    assert "asan_driver: PASS" in r.stdout


def test_cpp_sequence_driver_is_clean_under_asan_and_ubsan(tmp_path):
    """hm-opencl_amd/host/SequenceME.cpp + MultiDeviceME.cpp (launch / plane-slot planning, the planner on eight host threads at once as the
    N-device driver runs it, the pair -> device shard rule, and -- without a GPU -- the failure paths of both run() functions) compiled with
    -fsanitize=address,undefined together with tests/cpp/test_sequence_plan.cpp"""
    import shutil
    import subprocess
    import torch
    from hmme import api
    if not shutil.which("hipcc"):
        pytest.skip("ROCm headers / libamdhip64 not available")
    api.build()
    if torch.cuda.is_available():
        pytest.skip("CPU-build check; on the GPU box the class is covered by tests/test_gpu_sequence.py")
    exe = str(tmp_path / "seq_plan_asan")
    csrc = os.path.join(ROOT, "hm-opencl_amd", "csrc")
    subprocess.run(["g++", "-O1", "-g", "-std=c++11", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                    "-D__HIP_PLATFORM_AMD__=1", "-I/opt/rocm/include", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_sequence_plan.cpp"),
                    os.path.join(ROOT, "hm-opencl_amd", "host", "SequenceME.cpp"), os.path.join(ROOT, "hm-opencl_amd", "host", "MultiDeviceME.cpp"),
                    "-L" + csrc, "-lhmme", "-Wl,-rpath," + csrc, "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    supp = tmp_path / "lsan.supp"
    supp.write_text("leak:libamdhip64\nleak:libhsa-runtime64\nleak:libamd_comgr\nleak:librccl\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", LSAN_OPTIONS=f"suppressions={supp}:print_suppressions=0",
               UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]


def test_refinement_jobs_are_dealt_as_a_permutation_with_the_edge_ctus_first():
    """me_frac_deal (host + device function): workgroup k of a whole-picture refinement launch takes job hmme_test_frac_deal(k) -- every job of
    every pair exactly once, the CTUs on the picture's edge (bottom row, top row, side columns) of all pairs before any interior CTU; pictures
    less than three CTUs wide or high are dealt from the end of the table"""
    from hmme import api
    L = api.load()
    for (w, h, pairs) in ((3840, 2160, 1), (1920, 1080, 4), (832, 480, 3), (200, 192, 2), (64, 64, 1), (136, 72, 5), (16384, 128, 1)):
        X, Y = (w + 63) // 64, (h + 63) // 64
        n = X * Y
        deal = [L.hmme_test_frac_deal(k, pairs, w, h) for k in range(pairs * n)]
        assert sorted(deal) == list(range(pairs * n)), (w, h, pairs)
        assert L.hmme_test_frac_deal(pairs * n, pairs, w, h) == -1 and L.hmme_test_frac_deal(-1, pairs, w, h) == -1
        if X >= 3 and Y >= 3:
            edge = lambda j: (j % n) % X in (0, X - 1) or (j % n) // X in (0, Y - 1)
            n_edge = pairs * (2 * X + 2 * (Y - 2))
            assert all(edge(j) for j in deal[:n_edge]) and not any(edge(j) for j in deal[n_edge:]), (w, h, pairs)
            assert [j % n for j in deal[:X]] == list(range((Y - 1) * X, n))      # the (possibly partial) bottom row goes first
        else:
            assert deal == list(range(pairs * n - 1, -1, -1))


def test_bench_counts_the_candidates_actually_searched():
    """bench.py's work count with per-CTU predictors = sum over CTUs of (in-picture 4x4 blocks) x (candidates of the window the oracle's
    xSetSearchRange + clipMv restatement gives for that CTU's predictor)"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from hmme import api, synth
    import ctypes as C
    import oracle_py as O
    w, h, sr = 832, 480, 64
    n = ((w + 63) // 64) * ((h + 63) // 64)
    pred = synth.random_predictors(n, seed=5, max_pel=16)
    total = 0
    for ctu in range(n):
        cx, cy = (ctu % ((w + 63) // 64)) * 64, (ctu // ((w + 63) // 64)) * 64
        lt = [C.c_int() for _ in range(4)]   # (pred_x, pred_y, sr, cu_x, cu_y, pic_w, pic_h, max_cu) -> lt_x, lt_y, rb_x, rb_y
        O.oracle().hmo_set_search_range(int(pred[ctu, 0]), int(pred[ctu, 1]), sr, cx, cy, w, h, 64, *[C.byref(v) for v in lt])
        lt_x, lt_y, rb_x, rb_y = (v.value for v in lt)
        total += (min(64, w - cx) // 4) * (min(64, h - cy) // 4) * (rb_x - lt_x + 1) * (rb_y - lt_y + 1)
    assert bench.work_4x4_sads(api, w, h, sr, pred) == total
    assert bench.work_4x4_sads(api, w, h, sr) >= total * 0.95 and bench.work_4x4_sads(api, w, h, sr) != total


def test_tail_plan_of_whole_picture_searches():
    """hmme.hip prep_jobs, 8-bit: which jobs of a launch run whole and how the rest -- the jobs beyond the last full round of the chip's 512 workgroup
    slots -- are cut into equal segments (hmme_test_tail_plan: host arithmetic).  2160p's 504 and 1080p's 510 left-over jobs stay whole (measured
    slower as segments); 720p is all tail on exactly one round of segments; 1440p's 408 tail jobs share one launch with the 512 head jobs, 1200p's
    58 get a launch of their own; tiny windows never get more workgroups than they have units of four tasks"""
    import ctypes as C
    from hmme import api
    L = api.load()
    L.hmme_test_tail_plan.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_int)]

    def plan(w, h, sr=64, pairs=1, slots=512):
        out = (C.c_int * 4)()
        assert L.hmme_test_tail_plan(w, h, sr, pairs, slots, out) == 0
        return tuple(out)
    assert plan(3840, 2160) == (2040, 2040, 0, 0)
    assert plan(1920, 1080) == (510, 510, 0, 0)
    assert plan(3840, 2160, pairs=4) == (8160, 8160, 0, 0)            # 8 160 = 15 rounds + 480: whole
    assert plan(1280, 720) == (240, 0, 512, 1)                        # all tail: 240 x 17 units on one round of 512 segments
    assert plan(2560, 1440) == (920, 512, 512, 1)                     # 408 x 17 = 6 936 units on 512 segments behind 512 whole jobs, one launch
    assert plan(1920, 1200) == (570, 512, 512, 0)                     # 58 tail jobs: a launch of their own
    assert plan(64, 64, sr=8) == (1, 1, 0, 0)                         # one job of 17 x 17 candidates is 3 tasks = one unit: nothing to cut
    assert plan(64, 64, sr=64) == (1, 0, 17, 1)                       # one job of 129 x 129: 17 units, a workgroup each
    jobs, head, wgs, one = plan(832, 480, sr=16)                      # 104 jobs of 33 x 33 candidates
    assert jobs == 104 and head == 0 and 1 <= wgs <= 512
    for w, h, sr, pairs in ((1280, 720, 64, 3), (3840, 2160, 32, 1), (640, 360, 64, 16), (4096, 2304, 64, 1)):
        jobs, head, wgs, one = plan(w, h, sr, pairs)
        assert head in (jobs, jobs - jobs % 512) and (wgs > 0) == (head < jobs) and (not one or wgs > 0)
    assert L.hmme_test_tail_plan(1280, 720, 65, 1, 512, (C.c_int * 4)()) != 0      # windows beyond 129 x 129 are tiled, not planned here
