#!/usr/bin/env python3
"""ISA check for the hand-written DPP instructions (inline asm is not padded by the compiler's hazard recognizer):
a VALU write of a VGPR must be followed by 2 wait states before a DPP instruction reads that VGPR as its DPP source
(gfx9 data hazard).  Every instruction the wave issues is one wait state, `s_nop N` is N + 1.
Control flow: a local label (.LBB*) is a join -- some other path can reach it through a branch whose source the linear scan
does not see.  Conservatively, every VGPR is taken to have been written by the instruction right before that branch; the
branch itself is the one wait state in between.  So a DPP read within the first instruction after a label is a violation
unless padded, whatever the fall-through path did.
Second check (scalar loads): the inline-asm s_load_dword* of the search kernels (current block -> SGPRs, me_kernels.hpp ME8_CUR /
ME16_CUR / me_prefetch_cur) are invisible to the compiler's waitcnt insertion and liveness tracking.  Scalar loads return out of
order, so only `s_waitcnt lgkmcnt(0)` retires them: between an s_load and the next lgkmcnt(0) NO instruction may read or write a
destination SGPR of a pending load (a read would see stale data, a write -- e.g. the allocator reusing a dead destination for
an offset temporary -- would be overwritten when the load lands).  Another s_load into the same register is allowed (the
prefetch's deliberate pattern).  The scan is linear per kernel; the compiler's own kernarg loads pass trivially.
usage: check_dpp_hazard.py <file.s> [kernel-name-substring ...]   -> exit 1 on a violation"""
import re
import sys


def regs(tok):
    """VGPR numbers named by an operand token: v12 -> {12}, v[10:13] -> {10..13}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


ALL_VGPRS = frozenset(range(512))


def check(path, names):
    bad = total = 0
    kernel = None
    hist = []   # (wait states this instruction provides, set of VGPRs it writes)
    for ln in open(path):
        ln = ln.split(";")[0].strip()
        if not ln:
            continue
        if ln.endswith(":") and not ln.startswith("."):
            kernel = ln[:-1]
            hist = []
            continue
        if re.match(r"\.L[A-Za-z0-9_$.]*:$", ln) and kernel is not None:
            hist.append((1, ALL_VGPRS))   # unknown writer on the other path into this join ...
            hist.append((1, set()))       # ... then the branch that brought the wave here
            hist = hist[-4:]
            continue
        if ln.startswith(".") or kernel is None or (names and not any(n in kernel for n in names)):
            continue
        parts = ln.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        if op.endswith("_dpp"):
            total += 1
            src = regs(args[1]) if len(args) > 1 else set()
            need, i = 2, len(hist) - 1
            while need > 0 and i >= 0:
                ws, written = hist[i]
                if written & src:
                    bad += 1
                    print(f"{kernel}: '{ln}' reads {sorted(written & src)} {2 - need} wait state(s) after its write")
                    break
                need -= ws
                i -= 1
        if op == "s_nop":
            hist.append((int(args[0], 0) + 1, set()))
        elif op.startswith("v_") and not op.startswith("v_cmp"):
            hist.append((1, regs(args[0]) if args else set()))
        else:
            hist.append((1, set()))   # loads write VGPRs too, but their data arrives behind s_waitcnt, not by this hazard
        hist = hist[-4:]
    print(f"{path}: {total} DPP instructions checked, {bad} hazard(s)")
    return bad


def sregs(tok):
    """SGPR numbers named by an operand token: s12 -> {12}, s[10:13] -> {10..13}"""
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check_smem(path, names):
    bad = total = 0
    kernel = None
    pending = {}   # SGPR -> the load that targets it
    for ln in open(path):
        ln = ln.split(";")[0].strip()
        if not ln:
            continue
        if ln.endswith(":") and not ln.startswith("."):
            kernel, pending = ln[:-1], {}
            continue
        if ln.startswith(".") or kernel is None or (names and not any(n in kernel for n in names)):
            continue
        parts = ln.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        if op == "s_waitcnt":
            if "lgkmcnt(0)" in ln or (len(args) == 1 and args[0] in ("0", "0x0")):
                pending = {}
            continue
        touched = set()
        for a in args:
            touched |= sregs(a)
        is_load = op.startswith("s_load_dword") or op.startswith("s_buffer_load_dword")
        dest = sregs(args[0]) if (is_load and args) else set()
        hit = (touched - dest if is_load else touched) & set(pending)
        if hit:
            bad += 1
            print(f"{kernel}: '{ln}' touches s{sorted(hit)} while '{pending[min(hit)]}' is in flight (no s_waitcnt lgkmcnt(0) in between)")
        if is_load:
            total += 1
            for r in dest:
                pending[r] = ln
    print(f"{path}: {total} scalar loads checked, {bad} SGPR(s) touched while a load into them was pending")
    return bad


if __name__ == "__main__":
    rc = check(sys.argv[1], sys.argv[2:])
    rc += check_smem(sys.argv[1], sys.argv[2:])
    sys.exit(1 if rc else 0)
