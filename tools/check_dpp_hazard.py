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
Third check (immediates): operands the assembler accepts but the hardware does not honour, or that only mean what the source intended inside
a range.  Round 5 shipped a build to the GPU box whose `v_lshl_add_u64` asked for a shift by 6 and 7: it assembles (6 is an inline constant
like any other), the hardware shifts by 0..4 only, and the kernel returned garbage that only a golden vector on the GPU caught.  Checked in
every kernel: v_lshl_add_u64 shift 0..4; ds_read2 / ds_write2 offset0 / offset1 0..255 and single ds offsets 0..65535; DPP controls
(quad_perm digits 0..3, row_shl / row_shr / row_ror 1..15, the wave_* shifts :1, row_bcast 15 / 31, row_mask / bank_mask 0..0xf, nothing
gfx950 does not have: row_share, row_xmask, row_newbcast on 32-bit ops, dpp8); op_sel entries 0 / 1; s_setprio 0..3; s_nop 0..15; scalar-load
immediate offsets 0..0xfffff (the field is 21 bits signed; a negative one is never meant here); v_alignbyte_b32 literal shifts 0..3.
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


def _int(tok):
    """an integer literal of the assembler (decimal, 0x.., negative), or None for registers / symbols / expressions"""
    try:
        return int(tok, 0)
    except ValueError:
        return None


def immediate_errors(ln):
    """-> list of complaints about one instruction line (comment already stripped)"""
    errs = []
    parts = ln.replace(",", " ").split()
    if not parts:
        return errs
    op, args = parts[0], parts[1:]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base == "v_lshl_add_u64" and len(args) >= 3:
        sh = _int(args[2])
        if sh is not None and not 0 <= sh <= 4:
            errs.append(f"shift {sh}: v_lshl_add_u64 shifts by 0..4 only")
    if base == "v_alignbyte_b32" and len(args) >= 4:
        sh = _int(args[3])
        if sh is not None and not 0 <= sh <= 3:
            errs.append(f"byte shift {sh} outside 0..3")
    if base.startswith("ds_"):
        two = re.match(r"ds_(read2|write2|wrxchg2)", base) is not None
        for name, val in re.findall(r"\b(offset[01]?):(-?(?:0x[0-9a-fA-F]+|\d+))", ln):
            v = int(val, 0)
            if two and name == "offset":
                errs.append("a two-address LDS instruction takes offset0 / offset1, not offset")
            elif two and not 0 <= v <= 255:
                errs.append(f"{name}:{v} outside 0..255")
            elif not two and (name != "offset" or not 0 <= v <= 65535):
                errs.append(f"{name}:{v} is not a 16-bit offset of a one-address LDS instruction")
    if base in ("s_setprio", "s_nop") and args:
        v, hi = _int(args[0]), (3 if base == "s_setprio" else 15)
        if v is None or not 0 <= v <= hi:
            errs.append(f"{base} {args[0]} outside 0..{hi}")
    if re.match(r"s_(buffer_)?load_dword", base):
        for val in re.findall(r"\boffset:(-?(?:0x[0-9a-fA-F]+|\d+))", ln) + ([args[2]] if len(args) >= 3 and _int(args[2]) is not None else []):
            v = int(val, 0)
            if not 0 <= v <= 0xFFFFF:
                errs.append(f"scalar-load offset {v} outside 0..0xfffff")
    m = re.search(r"op_sel(?:_hi)?:\[([^\]]*)\]", ln)
    if m and any(t.strip() not in ("0", "1") for t in m.group(1).split(",")):
        errs.append(f"op_sel entries must be 0 or 1: [{m.group(1)}]")
    if op.endswith("_dpp") or " quad_perm:" in ln or " row_" in ln or " wave_" in ln or "dpp8:" in ln:
        ctrl = 0
        for m in re.finditer(r"\bquad_perm:\[([^\]]*)\]", ln):
            ctrl += 1
            d = [t.strip() for t in m.group(1).split(",")]
            if len(d) != 4 or any(t not in ("0", "1", "2", "3") for t in d):
                errs.append(f"quad_perm:[{m.group(1)}] needs four digits 0..3")
        for name, val in re.findall(r"\b(row_shl|row_shr|row_ror):(-?(?:0x[0-9a-fA-F]+|\d+))", ln):
            ctrl += 1
            if not 1 <= int(val, 0) <= 15:
                errs.append(f"{name}:{val} outside 1..15")
        for name, val in re.findall(r"\b(wave_shl|wave_shr|wave_rol|wave_ror):(-?(?:0x[0-9a-fA-F]+|\d+))", ln):
            ctrl += 1
            if int(val, 0) != 1:
                errs.append(f"{name}:{val}: only :1 exists")
        for val in re.findall(r"\brow_bcast:(-?(?:0x[0-9a-fA-F]+|\d+))", ln):
            ctrl += 1
            if int(val, 0) not in (15, 31):
                errs.append(f"row_bcast:{val}: only 15 and 31 exist")
        ctrl += len(re.findall(r"\brow_mirror\b|\brow_half_mirror\b", ln))
        for bad in re.findall(r"\b(row_share|row_xmask|row_newbcast|dpp8):", ln):
            ctrl += 1
            errs.append(f"{bad} is not a DPP control of this target's 32-bit operations")
        for name, val in re.findall(r"\b(row_mask|bank_mask):(-?(?:0x[0-9a-fA-F]+|\d+))", ln):
            if not 0 <= int(val, 0) <= 0xF:
                errs.append(f"{name}:{val} outside 0..0xf")
        if op.endswith("_dpp") and ctrl != 1:
            errs.append(f"{ctrl} DPP controls (exactly one is needed)")
    return errs


def check_immediates(path, names):
    bad = total = 0
    kernel = None
    for ln in open(path):
        ln = ln.split(";")[0].strip()
        if not ln:
            continue
        if ln.endswith(":") and not ln.startswith("."):
            kernel = ln[:-1]
            continue
        if ln.startswith(".") or kernel is None or (names and not any(n in kernel for n in names)):
            continue
        total += 1
        for e in immediate_errors(ln):
            bad += 1
            print(f"{kernel}: '{ln}': {e}")
    print(f"{path}: {total} instructions checked for immediate ranges, {bad} out of range")
    return bad


if __name__ == "__main__":
    rc = check(sys.argv[1], sys.argv[2:])
    rc += check_smem(sys.argv[1], sys.argv[2:])
    rc += check_immediates(sys.argv[1], sys.argv[2:])
    sys.exit(1 if rc else 0)
