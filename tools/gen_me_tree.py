#!/usr/bin/env python3
"""Generator for the per-lane reduction tree of the MI355X full-search ME kernel.

One *lane-iteration* of the kernel evaluates FOUR horizontally adjacent candidate MVs
(x, x+1, x+2, x+3 at one y) against the whole 64x64 CTU:

  leaves   256 4x4 blocks x 4 rows of v_qsad_pk_u16_u8 (4 candidates per instruction, packed
           u16 accumulators); per block an "even rows" sum E (rows 0,2 -- what HM's FEN
           sub-sampled SAD reads) and an "all rows" sum A (E chained through rows 1,3)
  tree     packed-u16 sums (v_pk_add_u16) up to 16x16, then 32-bit *keys*
           K = sad * MULT + C_j,  C_j = (mvcost_j << 10) | candidate index   (one v_mad_u32_u16)
           Keys are linear:  K(a U b) = K(a) + K(b) - C  (v_add3_u32),  K(a \\ b) = K(a) - K(b) + C,
           so everything above 16x16 is one VALU op per candidate and slot.
  arg-min  per slot: min over the lane's 4 candidates (v_min3_u32 + v_min_u32), then a
           64-register -> 1-register *butterfly transpose-reduce* across the wave
           (v_permlane32_swap, v_permlane16_swap, DPP row_ror:8 / row_half_mirror / quad_perm):
           afterwards lane l of group register g holds the wave-wide minimum of slot
           SLOT_OF[g][l]; ten running-minimum registers persist across iterations.

The slot numbering is the reference's (TComDataCU::getIndexBlock, TComDataCU.cpp:3379-3391 and
:4676-6461; same offsets in cl/sad.cl:200-365); FEN semantics are TEncSearch.cpp:3853-3859 +
TComRdCost.cpp:509-521 (rows > 8 -> every 2nd row, sum << 1).

The same op list is (a) emitted as straight-line HIP C++ (csrc/me_tree_fen{0,1}.inc) and
(b) interpreted with numpy (`simulate`) so the tree can be checked against the oracle on
the CPU (tests/test_tree_sim.py) before it ever runs on a GPU.

usage: python tools/gen_me_tree.py            (writes the .inc files)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_DIR = os.path.join(ROOT, "hm-opencl_amd", "csrc")

IDX_BITS = 10          # key = cost << 10 | iter(2) | lane(6) | j(2)
IDX_BITS16 = 8         # 16-bit path: cost << 8 | lane(6) | j(2), one lane-iteration per task  (24-bit cost: shift-free 9-bit bi-pred origins reach 6.3 M)
MULT_A = 1 << IDX_BITS
N_GROUPS = 10          # ceil(593 / 64)
PDW = 49               # LDS window pitch in dwords (odd: conflict-free for every lane shape)

# slot bases (SURVEY 8a closed form; verified against the reference table in tests)
BASE_2NxN = {8: 0, 16: 448, 32: 560, 64: 588}
BASE_Nx2N = {8: 128, 16: 480, 32: 568, 64: 590}
BASE_AMP = {16: 256, 32: 512, 64: 576}
BASE_2Nx2N = {8: 384, 16: 544, 32: 584, 64: 592}


def slot_2NxN(s, cx, cy, p):
    n = 64 // s
    return BASE_2NxN[s] + cy * 2 * n + p * n + cx


def slot_Nx2N(s, cx, cy, p):
    n = 64 // s
    return BASE_Nx2N[s] + cy * 2 * n + 2 * cx + p


def slot_AMP(s, cx, cy, k):
    n = 64 // s
    return BASE_AMP[s] + k * n * n + cy * n + cx


def slot_2Nx2N(s, cx, cy):
    n = 64 // s
    return BASE_2Nx2N[s] + cy * n + cx


SHARED_BASES = True   # LDS base registers made once per lane-iteration, not once per CU (A/B: False)
LOAD_OPS = ("LDS", "CURLD", "BASE", "LDS16Q", "ROWBASE", "CURLD16")


class Tree:
    """builds the op list for one lane-iteration"""

    def __init__(self, fen):
        self.fen = fen
        self.ops = []
        self.n = 0
        self.emitted = []          # slot ids in emission order (None = padding)
        self.pending = {}          # butterfly level -> var
        self.group = 0
        self.loaded = {}
        self.bases = {}
        self.cu_loads = []

    def new(self, prefix="v"):
        self.n += 1
        return f"{prefix}{self.n}"

    # ---- leaves -------------------------------------------------------------------------
    def lds(self, row, k):
        """dwords k, k+1 of window row `row`.  ds_read2_b32 reaches only 255 dwords from its address
        register, so each 4-row block row of a CU gets its own (opaque) base: 2 v_add per CU."""
        key = (row, k)
        if key not in self.loaded:
            # SHARED_BASES: one base per 4-row block row for the whole lane-iteration (offsets reach 3 * PDW + 17 = 164 dwords), 16 v_add
            # instead of 2 per CU = 128
            k0 = 0 if SHARED_BASES else self.cu_k0
            bkey = ("base", row // 4, k0)
            where = self.bases if SHARED_BASES else self.loaded
            if bkey not in where:
                b = self.new("a")
                self.ops.append(("BASE", b, (row // 4) * 4, k0))
                where[bkey] = b
            v = self.new("d")
            self.ops.append(("LDS", v, row, k, where[bkey], (row % 4) * PDW + (k - k0)))
            self.loaded[key] = v
        return self.loaded[key]

    def curld(self, row, cx8):
        """the 8 current-block bytes of `row` that belong to 8x8 CU column cx8 (scalar load into an SGPR pair)"""
        key = ("cur", row, cx8)
        if key not in self.loaded:
            v = self.new("w")
            self.ops.append(("CURLD", v, row, cx8))
            self.loaded[key] = v
        return self.loaded[key]

    def block(self, bx, by):
        """-> (E, A) packed-u16 sums of the 4x4 block at block coords (bx, by)"""
        def q(row, acc):
            v = self.new("q")
            self.ops.append(("QSAD", v, self.lds(row, bx), self.curld(row, bx >> 1), bx & 1, row, bx, acc))
            return v
        r0 = by * 4
        if self.fen:
            e = q(r0 + 2, q(r0, None))
            a = q(r0 + 3, q(r0 + 1, e))
            return e, a
        a = q(r0 + 3, q(r0 + 2, q(r0 + 1, q(r0, None))))
        return a, a

    # ---- packed sums / keys ---------------------------------------------------------------
    def pkadd(self, a, b):
        v = self.new("p")
        self.ops.append(("PKADD", v, a, b))
        return v

    def pksub(self, a, b):
        v = self.new("p")
        self.ops.append(("PKSUB", v, a, b))
        return v

    def keys(self, p, fam):
        v = self.new("k")
        self.ops.append(("KEYS", v, p, fam))
        return v

    def lin(self, a, b):
        v = self.new("k")
        self.ops.append(("LIN", v, a, b))
        return v

    def sub(self, a, b):
        v = self.new("k")
        self.ops.append(("SUB", v, a, b))
        return v

    # ---- arg-min ----------------------------------------------------------------------------
    def emit(self, slot, k):
        r = self.new("r")
        self.ops.append(("MIN4", r, k))
        self._push(slot, r)

    def _push(self, slot, r):
        self.emitted.append(slot)
        level = 0
        while level in self.pending:
            a = self.pending.pop(level)
            if a is None and r is None:
                m = None
            else:
                m = self.new("m")
                self.ops.append(("MERGE", level, m, a, r))
            r = m
            level += 1
            if level == 6:
                if r is not None:
                    self.ops.append(("ACC", self.group, r))
                self.group += 1
                return
        self.pending[level] = r

    def flush(self):
        while len(self.emitted) % 64:
            self._push(None, None)
        assert not self.pending and self.group == N_GROUPS

    # ---- the tree -------------------------------------------------------------------------------
    def build(self):
        U = "E" if self.fen else "A"   # family used by PUs taller than 8 rows
        quads = []
        for qy in range(2):
            for qx in range(2):
                regions = []
                for ry in range(2):
                    for rx in range(2):
                        cus = []
                        for cy in range(2):
                            for cx in range(2):
                                start = len(self.ops)
                                cus.append(self.level0(qx * 4 + rx * 2 + cx, qy * 4 + ry * 2 + cy, cx, cy))
                                new = self.ops[start:]
                                self.cu_loads.append([o for o in new if o[0] in LOAD_OPS])
                                self.ops[start:] = [("LOADS_FOR", len(self.cu_loads))] + \
                                    [o for o in new if o[0] not in LOAD_OPS]
                        regions.append(self.level1(qx * 2 + rx, qy * 2 + ry, rx, ry, cus, U))
                quads.append(self.level2(qx, qy, regions))
        self.level3(quads)
        self.flush()
        self.ops = self._assemble()
        assert sorted(s for s in self.emitted if s is not None) == list(range(593))
        return self

    def _assemble(self):
        """software prefetch: the loads of CU n+1 are issued at the top of CU n's arithmetic"""
        def arrived(loads):
            """the loads of the CU about to be consumed (issued one CU earlier); see Tree16._assemble"""
            return ("CURWAIT8", [o[1] for o in loads if o[0] == "CURLD"], [o[1] for o in loads if o[0] == "LDS"])
        ops = list(self.cu_loads[0])
        for o in self.ops:
            if o[0] == "LOADS_FOR":
                ops.append(arrived(self.cu_loads[o[1] - 1]))       # marker n+1 sits at the top of CU n
                if o[1] < len(self.cu_loads):
                    ops.extend(self.cu_loads[o[1]])
            elif o[0] != "MID":
                ops.append(o)
        return ops

    def level0(self, cx8, cy8, cx, cy):
        """8x8 CU at CU coords (cx8, cy8); (cx, cy) = position inside its 16x16 region"""
        self.loaded = {}   # window dwords are re-read per CU: short live ranges beat 30% fewer LDS reads
        self.cu_k0 = 2 * cx8
        (e0, a0), (e1, a1) = self.block(2 * cx8, 2 * cy8), self.block(2 * cx8 + 1, 2 * cy8)
        (e2, a2), (e3, a3) = self.block(2 * cx8, 2 * cy8 + 1), self.block(2 * cx8 + 1, 2 * cy8 + 1)
        at, ab = self.pkadd(a0, a1), self.pkadd(a2, a3)
        al, ar = self.pkadd(a0, a2), self.pkadd(a1, a3)
        k_at, k_ab = self.keys(at, "A"), self.keys(ab, "A")
        self.emit(slot_2NxN(8, cx8, cy8, 0), k_at)
        self.emit(slot_2NxN(8, cx8, cy8, 1), k_ab)
        self.emit(slot_Nx2N(8, cx8, cy8, 0), self.keys(al, "A"))
        self.emit(slot_Nx2N(8, cx8, cy8, 1), self.keys(ar, "A"))
        k_a8 = self.lin(k_at, k_ab)
        self.emit(slot_2Nx2N(8, cx8, cy8), k_a8)
        out = {"k_a8": k_a8, "k_at": k_at, "k_ab": k_ab}
        if self.fen:
            ut, ub = self.pkadd(e0, e1), self.pkadd(e2, e3)
            out["ucol"] = self.pkadd(e0, e2) if cx == 0 else self.pkadd(e1, e3)
        else:
            ut, ub = at, ab
            out["ucol"] = al if cx == 0 else ar
        out["u8"] = self.pkadd(ut, ub)
        out["urow"] = ut if cy == 0 else ub
        return out

    def level1(self, rx16, ry16, rx, ry, c, U):
        """16x16 region at region coords (rx16, ry16); (rx, ry) = position inside its 32x32"""
        S = 16
        # all-rows family (h <= 8)
        k_top = self.lin(c[0]["k_a8"], c[1]["k_a8"])
        k_bot = self.lin(c[2]["k_a8"], c[3]["k_a8"])
        self.emit(slot_2NxN(S, rx16, ry16, 0), k_top)
        self.emit(slot_2NxN(S, rx16, ry16, 1), k_bot)
        self.emit(slot_AMP(S, rx16, ry16, 0), self.lin(c[0]["k_at"], c[1]["k_at"]))   # 16x4 top
        self.emit(slot_AMP(S, rx16, ry16, 1), self.lin(c[2]["k_ab"], c[3]["k_ab"]))   # 16x4 bottom
        # tall family (h > 8): packed sums still fit u16 (<= 16*8*255 even rows / 16*16*255 all rows)
        left, right = self.pkadd(c[0]["u8"], c[2]["u8"]), self.pkadd(c[1]["u8"], c[3]["u8"])
        u16 = self.pkadd(left, right)
        t4, b4 = self.pkadd(c[0]["urow"], c[1]["urow"]), self.pkadd(c[2]["urow"], c[3]["urow"])
        l4, r4 = self.pkadd(c[0]["ucol"], c[2]["ucol"]), self.pkadd(c[1]["ucol"], c[3]["ucol"])
        k_left, k_right, k_u16 = self.keys(left, U), self.keys(right, U), self.keys(u16, U)
        self.emit(slot_Nx2N(S, rx16, ry16, 0), k_left)
        self.emit(slot_Nx2N(S, rx16, ry16, 1), k_right)
        self.emit(slot_2Nx2N(S, rx16, ry16), k_u16)
        self.emit(slot_AMP(S, rx16, ry16, 2), self.keys(self.pksub(u16, b4), U))   # 16x12 top
        self.emit(slot_AMP(S, rx16, ry16, 3), self.keys(self.pksub(u16, t4), U))   # 16x12 bottom
        self.emit(slot_AMP(S, rx16, ry16, 4), self.keys(l4, U))                    # 4x16 left
        self.emit(slot_AMP(S, rx16, ry16, 5), self.keys(r4, U))                    # 4x16 right
        self.emit(slot_AMP(S, rx16, ry16, 6), self.keys(self.pksub(u16, r4), U))   # 12x16 left
        self.emit(slot_AMP(S, rx16, ry16, 7), self.keys(self.pksub(u16, l4), U))   # 12x16 right
        out = {"k_u16": k_u16,
               "k_ucol": k_left if rx == 0 else k_right,                          # 8x16 strip on the 32x32's edge
               "k_a16x8": k_top if ry == 0 else k_bot}                            # all-rows 16x8 strip on the edge
        strip = self.pkadd(c[0]["u8"], c[1]["u8"]) if ry == 0 else self.pkadd(c[2]["u8"], c[3]["u8"])
        out["k_u16x8"] = self.keys(strip, U)                                        # tall-family 16x8 strip
        return out

    def level2(self, qx, qy, r):
        """32x32 CU; every value is a key from here on"""
        S = 32
        k_t, k_b = self.lin(r[0]["k_u16"], r[1]["k_u16"]), self.lin(r[2]["k_u16"], r[3]["k_u16"])
        k_l, k_r = self.lin(r[0]["k_u16"], r[2]["k_u16"]), self.lin(r[1]["k_u16"], r[3]["k_u16"])
        k_u32 = self.lin(k_t, k_b)
        self.emit(slot_2NxN(S, qx, qy, 0), k_t)
        self.emit(slot_2NxN(S, qx, qy, 1), k_b)
        self.emit(slot_Nx2N(S, qx, qy, 0), k_l)
        self.emit(slot_Nx2N(S, qx, qy, 1), k_r)
        self.emit(slot_2Nx2N(S, qx, qy), k_u32)
        self.emit(slot_AMP(S, qx, qy, 0), self.lin(r[0]["k_a16x8"], r[1]["k_a16x8"]))   # 32x8 top   (all rows)
        self.emit(slot_AMP(S, qx, qy, 1), self.lin(r[2]["k_a16x8"], r[3]["k_a16x8"]))   # 32x8 bottom
        k_u32x8t = self.lin(r[0]["k_u16x8"], r[1]["k_u16x8"])
        k_u32x8b = self.lin(r[2]["k_u16x8"], r[3]["k_u16x8"])
        k_u8x32l = self.lin(r[0]["k_ucol"], r[2]["k_ucol"])
        k_u8x32r = self.lin(r[1]["k_ucol"], r[3]["k_ucol"])
        self.emit(slot_AMP(S, qx, qy, 2), self.sub(k_u32, k_u32x8b))   # 32x24 top
        self.emit(slot_AMP(S, qx, qy, 3), self.sub(k_u32, k_u32x8t))   # 32x24 bottom
        self.emit(slot_AMP(S, qx, qy, 4), k_u8x32l)                    # 8x32 left
        self.emit(slot_AMP(S, qx, qy, 5), k_u8x32r)                    # 8x32 right
        self.emit(slot_AMP(S, qx, qy, 6), self.sub(k_u32, k_u8x32r))   # 24x32 left
        self.emit(slot_AMP(S, qx, qy, 7), self.sub(k_u32, k_u8x32l))   # 24x32 right
        return {"k_u32": k_u32, "k_row": k_t if qy == 0 else k_b, "k_col": k_l if qx == 0 else k_r}

    def level3(self, q):
        S = 64
        k_t, k_b = self.lin(q[0]["k_u32"], q[1]["k_u32"]), self.lin(q[2]["k_u32"], q[3]["k_u32"])
        k_l, k_r = self.lin(q[0]["k_u32"], q[2]["k_u32"]), self.lin(q[1]["k_u32"], q[3]["k_u32"])
        k_64 = self.lin(k_t, k_b)
        self.emit(slot_2NxN(S, 0, 0, 0), k_t)
        self.emit(slot_2NxN(S, 0, 0, 1), k_b)
        self.emit(slot_Nx2N(S, 0, 0, 0), k_l)
        self.emit(slot_Nx2N(S, 0, 0, 1), k_r)
        self.emit(slot_2Nx2N(S, 0, 0), k_64)
        k_16t, k_16b = self.lin(q[0]["k_row"], q[1]["k_row"]), self.lin(q[2]["k_row"], q[3]["k_row"])
        k_16l, k_16r = self.lin(q[0]["k_col"], q[2]["k_col"]), self.lin(q[1]["k_col"], q[3]["k_col"])
        self.emit(slot_AMP(S, 0, 0, 0), k_16t)                   # 64x16 top (h=16: tall family)
        self.emit(slot_AMP(S, 0, 0, 1), k_16b)                   # 64x16 bottom
        self.emit(slot_AMP(S, 0, 0, 2), self.sub(k_64, k_16b))   # 64x48 top
        self.emit(slot_AMP(S, 0, 0, 3), self.sub(k_64, k_16t))   # 64x48 bottom
        self.emit(slot_AMP(S, 0, 0, 4), k_16l)                   # 16x64 left
        self.emit(slot_AMP(S, 0, 0, 5), k_16r)                   # 16x64 right
        self.emit(slot_AMP(S, 0, 0, 6), self.sub(k_64, k_16r))   # 48x64 left
        self.emit(slot_AMP(S, 0, 0, 7), self.sub(k_64, k_16l))   # 48x64 right

    # ---- slot map ---------------------------------------------------------------------------------
    def slot_of_lane(self):
        """[group][lane] -> slot id (or -1): lane l ends up with emission index bitrev6(l)"""
        t = np.full((N_GROUPS, 64), -1, np.int32)
        for g in range(N_GROUPS):
            for lane in range(64):
                e = sum(((lane >> LEVEL_ROLE_BIT[lv]) & 1) << lv for lv in range(6))   # emission index bit lv = role at level lv
                s = self.emitted[g * 64 + e]
                t[g, lane] = -1 if s is None else s
        return t


class Tree16(Tree):
    """Reduction tree of the 16-bit sample path (bit depth 9..12, bi-prediction origins of any depth).

    A lane owns NC = 3 candidates of one column parity, (x, x+2, x+4): the kernel runs the even and the odd window columns as two
    passes over an LDS window loaded with a one-sample shift, so every candidate reads dword-aligned u16 pairs, each one dword
    further on than the previous -- together the six dwords that three 64-bit reads of a window row deliver, nothing realigned
    per lane.  Lane bases are 3 dwords apart: the reads are 4-byte-aligned 64-bit loads (ds_read2_b32) through one opaque base
    per two window rows (its offsets reach 255 dwords).
    Every value is a triple of exact 32-bit sums: HM applies `>> (bitDepth-8)` to the whole-PU sum *after* the FEN `<< 1`
    (TComRdCost.cpp:520-521), a floor and therefore not linear -- no key linearity, no u16 packing; each slot's key is formed
    from its own exact sum.  Round 5 cut the INSTRUCTION COUNT of that arithmetic (this kernel does not care what an instruction costs in
    a pure stream -- VOP2 keys instead of VOP3 ones changed nothing -- it cares how many there are):
      * every slot is a sum of at least two 4x4 leaves, and the first add of two leaf sums also shifts the result to its place in
        the key: (a + b) << LSH_f, LSH_f = 8 - sh for the all-rows family, 9 - sh for the even-rows family of FEN (its `<< 1`).
        Every sum above the leaves is S << LSH_f (28 bits at most: adds and subtractions of such sums are exact) with the bits that
        HM's `>> (bitDepth-8)` drops sitting just below the key's cost field:   key_j = (S'_j & ~0xff) + C_j
      * candidates 0 and 1 of every value travel as ONE 64-bit register pair (PAIR64): an add of two sums is one v_lshl_add_u64 for
        both (+ one v_add_u32 for the third candidate), the first-level shift one v_lshlrev_b64, the add of the key constants one
        v_lshl_add_u64 -- neither half ever carries into or borrows from the other, every sum and key stays below 2^32.  (The shift
        cannot ride on v_lshl_add_u64 itself: the instruction only shifts by 0..4 -- a build that asked it for 6 and 7 was 5 % faster
        and wrong.)  Per slot 6 instructions instead of 7, per add 2 instead of 3: 14 268 -> 13 534 VALU instructions per
        lane-iteration, BASELINE config 5 1 811 -> 1 868 GSAD/s (profiles/r05aa_search16_instruction_count.txt).
    Leaves are v_sad_u16 (2 samples per op).  Three 32-bit sums per value leave no room for a whole CU of loads in flight on top
    of the CU being consumed, so the prefetch distance is half a CU (_assemble)."""

    nc = 3

    def block16(self, cx8, cy8):
        """-> [(E, A)] for the 4 blocks TL, TR, BL, BR of the CU; E/A are NC-candidate sum tuples"""
        out = []
        rows = {}
        nc = self.nc
        for r in range(8):
            row = cy8 * 8 + r
            # one opaque base per two window rows: ds_read2_b32 offsets reach 255 dwords, a row is ME16_PDW <= 162 (+ 37 for the last CU column).
            # SHARED_BASES: the base is made once per lane-iteration and serves all eight CU columns (it used to be made again for every
            # CU: 256 v_mov per lane-iteration of values the compiler kept in registers anyway)
            base = f"lrow{row & ~1}" if SHARED_BASES else f"lrow{row & ~1}_{cx8}"
            if r % 2 == 0 and not (SHARED_BASES and base in self.rowbases):
                self.rowbases.add(base)
                self.ops.append(("ROWBASE", base, row))
            d = []
            for i in range(0, 6, 2):
                v0, v1 = self.new("d"), self.new("d")
                self.ops.append(("LDS16Q", v0, v1, base, (row & 1), 4 * cx8 + i))
                d += [v0, v1]
            w = self.new("w")
            self.ops.append(("CURLD16", w, row, cx8))
            rows[r] = (d, w)
        for by in range(2):
            if by == 1:
                self.ops.append(("MID",))   # rows 4..7 are first read from here on (see _assemble)
            for bl in range(2):
                def chain(r, acc):
                    d, w = rows[by * 4 + r]
                    v = self.new("s")
                    # candidate j reads the row j dwords on: samples (2*bl + j) * 2 .. + 3
                    self.ops.append(("SAD16xN", v, [(d[2 * bl + j], d[2 * bl + j + 1]) for j in range(nc)], w, 2 * bl, acc,
                                     cy8 * 8 + by * 4 + r, cx8 * 8 + bl * 4))
                    return v
                if self.fen:
                    e = chain(2, chain(0, None))
                    a = chain(3, chain(1, e))
                    self.leaf_family[e], self.leaf_family[a] = "E", "A"
                    out.append((e, a))
                else:
                    a = chain(3, chain(2, chain(1, chain(0, None))))
                    self.leaf_family[a] = "A"
                    out.append((a, a))
        return out

    def level0(self, cx8, cy8, cx, cy):
        if not hasattr(self, "rowbases"):
            self.rowbases = set()
        if not hasattr(self, "leaf_family"):
            self.leaf_family = {}     # leaf sum -> family ("A" all rows, "E" even rows): unshifted values; everything else is shifted
        self._blocks = self.block16(cx8, cy8)
        self._next_block = 0
        return Tree.level0(self, cx8, cy8, cx, cy)

    def block(self, bx, by):
        b = self._blocks[self._next_block]   # level0 asks for TL, TR, BL, BR in this order
        self._next_block += 1
        return b

    def pkadd(self, a, b):
        v = self.new("p")
        fa, fb = self.leaf_family.get(a), self.leaf_family.get(b)
        assert fa == fb, "a leaf sum only ever meets a leaf sum of its own family"
        if fa:     # two leaves: the add that also shifts the sum to its place in the key
            self.ops.append(("ADDSHLN", v, a, b, fa))
        else:
            self.ops.append(("ADDN", v, a, b))
        return v

    def pksub(self, a, b):
        assert a not in self.leaf_family and b not in self.leaf_family
        v = self.new("p")
        self.ops.append(("SUBN", v, a, b))
        return v

    # "keys" stay exact sums tagged with their family until they are emitted
    def keys(self, p, fam):
        return (p, fam)

    def lin(self, a, b):
        assert a[1] == b[1]
        return (self.pkadd(a[0], b[0]), a[1])

    def sub(self, a, b):
        assert a[1] == b[1]
        return (self.pksub(a[0], b[0]), a[1])

    def emit(self, slot, k):
        r = self.new("r")
        self.ops.append(("KEYMINN", r, k[0], k[1]))
        self._push(slot, r)

    def _assemble(self):
        """Half-CU prefetch: rows 4..7 of CU n are issued at the top of CU n's arithmetic (needed from its middle on), rows 0..3 of
        CU n+1 at the middle of CU n: at most one CU's worth of loads (48 window + 32 current-block registers) is live at any time."""
        def half(loads, h):
            return [o for o in loads if (o[0] == "ROWBASE" and (o[2] & 7) // 4 == h) or (o[0] == "LDS16Q" and self._row_of(o) // 4 == h)
                    or (o[0] == "CURLD16" and (o[2] & 7) // 4 == h)]
        def arrived(loads):
            """the batch about to be consumed: its scalar loads (the current block) complete out of order, so the wave waits for
            the whole batch BEFORE it issues the next one -- the batch has had half a CU of arithmetic to arrive"""
            return ("CURWAIT", [o[1] for o in loads if o[0] == "CURLD16"], [o[1] for o in loads if o[0] == "LDS16Q"])
        ops = list(half(self.cu_loads[0], 0))
        cu = 0
        for o in self.ops:
            if o[0] == "LOADS_FOR":
                cu = o[1] - 1                      # marker n+1 sits at the top of CU n
                ops.append(arrived(half(self.cu_loads[cu], 0)))
                ops.extend(half(self.cu_loads[cu], 1))
            elif o[0] == "MID":
                ops.append(arrived(half(self.cu_loads[cu], 1)))
                if cu + 1 < len(self.cu_loads):
                    ops.extend(half(self.cu_loads[cu + 1], 0))
            else:
                ops.append(o)
        return ops

    def _row_of(self, o):
        """row (0..7 inside its CU) of an LDS16Q op: its base names the even row, op[4] the parity"""
        return (int(o[3][4:].split("_")[0]) & 7) + o[4]


# =====================================================================================================
# C++ emitter
# =====================================================================================================
HEADER = """// GENERATED by tools/gen_me_tree.py -- do not edit.  One lane-iteration of the full-search
// reduction tree (fen=%d): 256 4x4 leaves -> 593 PU keys -> wave butterfly -> best[0..9].
// Expects in scope: lpc (per-lane LDS byte pointer at the candidate's window row), ME8_CUR(row, q) (scalar load of 8 current-block
// bytes) and ME8_CUR_WAIT (the CU's loads have arrived), c0..c3, nc0..nc3, b0..b9 (running minima), lane role
// masks rb3, rb2, rb1, rb0, and the ME_* helpers of me_kernel.hip.
"""


HEADER16 = """// GENERATED by tools/gen_me_tree.py -- do not edit.  One lane-iteration of the 16-bit-sample reduction tree
// (fen=%d): three candidates (x, x+2, x+4) per lane, exact 32-bit sums, v_sad_u16 leaves.
// Expects in scope: lpd (per-lane LDS byte pointer at the first candidate, window row 0; 4-byte aligned), ME16_PDW (window pitch in
// dwords), ME16_CUR(row, q) (scalar load of 8 current-block samples) and ME16_CUR_WAIT (the batch has arrived), c0, c1, c2,
// b0..b9, rb1, rb0, lsh_a / lsh_e (the shift of the first add of two leaf sums, all-rows / even-rows family), the key macro
// ME16_KEYMIN and the me_merge* helpers.
"""


MASKED_MERGE_LEVELS = (0, 1)   # the two levels done with hand-written bank-masked DPP pairs (me_merge0 / me_merge1)
# instructions of the ops that the 8-bit kernel emits as `asm volatile` (their mutual order in the ISA is the order here)
# 16-bit tree: candidates 0 and 1 of every sum above the leaves travel as ONE 64-bit value -- an add of two sums and the add of the
# pair's key constants are one v_lshl_add_u64 each (shift 0: the instruction only shifts by 0..4, which is why the key's shift rides on the
# first add of two leaf sums), 6 instructions per slot where three separate candidates took 7, 2 per add where they took 3
PAIR64 = True
ORDERED_INSTRS = {"KEYS": 4, "LIN": 4, "SUB": 4, "MIN4": 2, "KEYMINN": 1 if PAIR64 else 7}


def space_merges(ops, enable):
    """A DPP instruction must not read a VGPR within 2 wait states of the VALU write that produced it, and the compiler
    does not pad inside inline asm.  The masked merges, the per-slot minima that feed them and the key arithmetic are
    `asm volatile` in both kernels, so their mutual order in the ISA is this list's order; a masked merge is held
    back in a FIFO until >= 2 such instructions (one KEYS / LIN / SUB / MIN4 / other masked merge) separate it from the
    ops that produce its inputs -- then it needs no s_nop (unpadded variant, checked by tools/check_dpp_hazard.py).
    What is still waiting when nothing independent is left gets the padded variant.
    -> [(op, padded)]; `padded` only matters for masked-level MERGE ops.  enable=False: original order, all padded."""
    if not enable:
        return [(op, True) for op in ops]
    out, where = [], {}          # emitted (op, padded); value name -> ordered-instruction count right after its producer
    pending = []
    count = [0]

    def inputs(op):
        return [x for x in (op[3], op[4]) if x is not None] if op[0] == "MERGE" else [op[2]]

    def masked(op):
        return op[0] == "MERGE" and op[1] in MASKED_MERGE_LEVELS

    def try_flush(force=False):
        progress = True
        while progress and pending:
            progress = False
            for op in list(pending):
                ins = inputs(op)
                if any(x not in where for x in ins):
                    continue                                        # its producer is itself waiting
                fresh = masked(op) and any(count[0] - where[x] < 2 for x in ins)
                if fresh and not force:
                    continue
                pending.remove(op)
                if masked(op):
                    count[0] += 2
                if op[0] == "MERGE":
                    # results of the compiler-scheduled levels can land anywhere: treat them as fresh for good
                    where[op[2]] = count[0] if masked(op) else float("inf")
                out.append((op, fresh))
                progress = True

    key_at = {}                  # key name -> ordered-instruction count right after the KEYS / LIN / SUB that made it
    held = []                    # MIN4 ops waiting for one other ordered op to separate them from the op that made their keys:
                                 # hipcc pads an asm block that reads what the asm block right before it wrote (s_nop 0)

    def emit_plain(op):
        count[0] += ORDERED_INSTRS.get(op[0], 0)
        if op[0] in ("KEYS", "LIN", "SUB"):
            key_at[op[1]] = count[0]
        if op[0] in ("MIN4", "KEYMINN"):       # the ops whose result a level-0 merge reads
            where[op[1]] = count[0]
        out.append((op, False))

    def release_held(force=False):
        for op in list(held):
            if force or count[0] > key_at.get(op[2], -1):
                held.remove(op)
                emit_plain(op)

    for op in ops:
        try_flush()
        if op[0] in ("MERGE", "ACC"):
            pending.append(op)
            try_flush()
        elif op[0] == "MIN4" and key_at.get(op[2], -1) == count[0]:
            held.append(op)
        else:
            emit_plain(op)
            release_held()
    release_held(force=True)
    try_flush(force=True)
    assert not pending
    return out


def emit_cpp(tree, path, header=None):
    o = [(header or HEADER) % tree.fen]
    max_declared = False
    for op, padded in space_merges(tree.ops, enable=True):
        t = op[0]
        if t == "BASE":
            _, b, row, k = op
            o.append(f"const lds_char_t* {b} = lpc + {(row * PDW + k) * 4}; asm volatile(\"\" : \"+v\"({b}));")
        elif t == "LDS":
            _, v, row, k, b, off = op
            # dwords k, k+1 of the window row as ONE 4-byte-aligned 64-bit LDS read (ds_read2_b32): lands in an
            # even-aligned VGPR pair, which is what v_qsad_pk_u16_u8's 64-bit operand needs -> no v_mov
            o.append(f"const uint64_t {v} = *(const lds_vu64a4_t*)({b} + {off * 4});")
        elif t == "CURLD":
            _, v, row, cx8 = op
            o.append(f"uint64_t {v} = ME8_CUR({row}, {cx8});")
        elif t == "CURWAIT8":
            o.append(f"ME8_CUR_WAIT({', '.join(op[1])}, {', '.join(op[2])});")
        elif t == "QSAD":
            _, v, pair, cw, half, row, bx, acc = op
            o.append(f"const uint64_t {v} = ME_QSAD({pair}, (uint32_t)({cw}{' >> 32' if half else ''}), {acc if acc else '0ull'});")
        elif t == "PKADD":
            o.append(f"const uint64_t {op[1]} = me_pkadd({op[2]}, {op[3]});")
        elif t == "PKSUB":
            o.append(f"const uint64_t {op[1]} = me_pksub({op[2]}, {op[3]});")
        elif t == "KEYS":
            _, v, p, fam = op
            mult = "mult_e" if fam == "E" else "mult_a"
            o.append(f"ME_KEYS({v}, {p}, {mult});")
        elif t == "LIN":
            o.append(f"ME_LIN({op[1]}, {op[2]}, {op[3]});")
        elif t == "SUB":
            o.append(f"ME_SUB({op[1]}, {op[2]}, {op[3]});")
        elif t == "MIN4":
            o.append(f"const uint32_t {op[1]} = ME_MIN4({op[2]});")
        elif t == "MERGE":
            _, level, m, a, b = op
            a = a if a is not None else "ME_MAXKEY"
            b = b if b is not None else "ME_MAXKEY"
            nn = "_nn" if level in MASKED_MERGE_LEVELS and not padded else ""
            o.append(f"const uint32_t {m} = me_merge{level}{nn}({a}, {b}{'' if level < 4 else f', rb{5 - level}'});")
        elif t == "ACC":
            o.append(f"b{op[1]} = min(b{op[1]}, {op[2]});")
        elif t == "LDS16Q":   # 4-byte-aligned 64-bit read (ds_read2_b32)
            o.append(f"const uint64_t {op[1]}_q = *(const lds_vu64a4_t*)({op[3]} + ({op[4]} * ME16_PDW + {op[5]}) * 4); "
                     f"const uint32_t {op[1]} = (uint32_t){op[1]}_q, {op[2]} = (uint32_t)({op[1]}_q >> 32);")
        elif t == "ROWBASE":
            o.append(f"const lds_char_t* {op[1]} = lpd + {op[2]} * ME16_PDW * 4; asm volatile(\"\" : \"+v\"({op[1]}));")
        elif t == "SAD16xN":
            _, v, dd, w, wi, acc, row, col = op
            parts = []
            for j, (d0, d1) in enumerate(dd):
                a = f"{acc}_{j}" if acc else "0u"
                parts.append(f"{v}_{j} = ME_SAD16({d1}, {w}[{wi + 1}], ME_SAD16({d0}, {w}[{wi}], {a}))")
            o.append("const uint32_t " + ", ".join(parts) + ";")
            if PAIR64 and v in tree.leaf_family:
                o.append(f"const uint64_t {v}_p = me_pair({v}_0, {v}_1);")
        elif t == "ADDN" and PAIR64:   # candidates 0 and 1 as one 64-bit add (v_lshl_add_u64): neither half ever carries, a sum stays below 2^32
            o.append(f"const uint64_t {op[1]}_p = {op[2]}_p + {op[3]}_p; const uint32_t {op[1]}_2 = {op[2]}_2 + {op[3]}_2;")
        elif t == "ADDN":
            o.append("const uint32_t " + ", ".join(f"{op[1]}_{j} = {op[2]}_{j} + {op[3]}_{j}" for j in range(tree.nc)) + ";")
        elif t == "ADDSHLN":
            if PAIR64:   # the two leaves' pairs added as one (v_lshl_add_u64), the pair shifted as one (v_lshlrev_b64: a leaf-pair sum is 17 bits, the low half never reaches the high one)
                o.append(f"const uint64_t {op[1]}_p = ({op[2]}_p + {op[3]}_p) << lsh_{op[4].lower()}; const uint32_t {op[1]}_2 = ({op[2]}_2 + {op[3]}_2) << lsh_{op[4].lower()};")
            else:
                o.append("const uint32_t " + ", ".join(f"{op[1]}_{j} = ({op[2]}_{j} + {op[3]}_{j}) << lsh_{op[4].lower()}" for j in range(tree.nc)) + ";")
        elif t == "SUBN" and PAIR64:   # a region minus a part of it: neither half ever borrows
            o.append(f"const uint64_t {op[1]}_p = {op[2]}_p - {op[3]}_p; const uint32_t {op[1]}_2 = {op[2]}_2 - {op[3]}_2;")
        elif t == "SUBN":
            o.append("const uint32_t " + ", ".join(f"{op[1]}_{j} = {op[2]}_{j} - {op[3]}_{j}" for j in range(tree.nc)) + ";")
        elif t == "KEYMINN" and PAIR64:
            o.append(f"const uint32_t {op[1]} = ME16_KEYMIN_P({op[2]}_p, {op[2]}_2);")
        elif t == "KEYMINN":
            o.append(f"const uint32_t {op[1]} = ME16_KEYMIN({op[2]}_0, {op[2]}_1, {op[2]}_2);")
        elif t == "CURLD16":
            o.append(f"u32x4_t {op[1]} = ME16_CUR({op[2]}, {op[3]});")
        elif t == "CURWAIT":
            o.append(f"ME16_CUR_WAIT({', '.join(op[1])}, {', '.join(v + '_q' for v in op[2])});")
        else:
            raise ValueError(t)
    write_if_changed(path, "\n".join(o) + "\n")


def write_if_changed(path, text):
    """keep mtimes stable so that `make` does not rebuild libhmme.so after a no-op regeneration"""
    try:
        if open(path).read() == text:
            return
    except OSError:
        pass
    with open(path, "w") as f:
        f.write(text)


def emit_slotmap(tree, path):
    t = tree.slot_of_lane()
    o = ["// GENERATED by tools/gen_me_tree.py: slot held by lane l of running-minimum register g",
         "// after the butterfly (-1 = padding).  Same for fen=0 and fen=1.",
         "static __device__ const short ME_SLOT_OF[%d][64] = {" % N_GROUPS]
    for g in range(N_GROUPS):
        o.append("  {" + ", ".join(str(int(v)) for v in t[g]) + "},")
    o.append("};")
    write_if_changed(path, "\n".join(o) + "\n")


# =====================================================================================================
# numpy interpreter (64 lanes at once)
# =====================================================================================================


LANES = np.arange(64)

# butterfly levels, cheapest mechanism for the most numerous merges (measured on gfx950: masked DPP pair
# 2 x 4.3 cycles, v_permlane*_swap 8 + v_min 4.3, cndmask pair + DPP min 3 x 4.3):
#   level 0  row_ror:8 + bank masks        role = lane bit 3      level 3  v_permlane16_swap     role = bit 4
#   level 1  row_half_mirror + bank masks  role = lane bit 2      level 4  quad_perm [2,3,0,1]   role = bit 1
#   level 2  v_permlane32_swap             role = lane bit 5      level 5  quad_perm [1,0,3,2]   role = bit 0
LEVEL_XOR = [8, 7, 32, 16, 2, 1]
LEVEL_ROLE_BIT = [3, 2, 5, 4, 1, 0]


def _perm_level(level):
    """lane permutation and role bit of butterfly level 0..5"""
    return LANES ^ LEVEL_XOR[level], (LANES >> LEVEL_ROLE_BIT[level]) & 1


def simulate16(tree, window, cur, lane_off, c, best, sh):
    """interpret one lane-iteration of a Tree16.  window: (rows, pitch_samples) uint16 LDS image;
    cur: (64,64) uint16; lane_off[l] = sample index of lane l's first candidate at window row 0;
    c: (3, 64) uint32 per-candidate constants; sh = bit_depth - 8."""
    rowbase = {}
    flat = np.concatenate([window.reshape(-1), np.zeros(64, window.dtype)]).astype(np.int64)
    pitch = window.shape[1]
    lsh = {"A": IDX_BITS16 - sh, "E": (IDX_BITS16 + 1 - sh) if tree.fen else IDX_BITS16 - sh}
    keymask = ~((1 << IDX_BITS16) - 1)
    val = {}
    MAXK = np.full(64, 0xFFFFFFFF, np.uint32)
    for op in tree.ops:
        t = op[0]
        if t == "ROWBASE":
            rowbase[op[1]] = op[2]
        elif t == "LDS16Q":    # dwords k, k+1 of a window row = samples 2k .. 2k+3 (relative to the lane's first candidate)
            for v, k in ((op[1], op[5]), (op[2], op[5] + 1)):
                idx = lane_off + (rowbase[op[3]] + op[4]) * pitch + 2 * k
                val[v] = np.stack([flat[idx], flat[idx + 1]], axis=1)
        elif t == "CURWAIT":
            pass
        elif t == "CURLD16":
            val[op[1]] = cur[op[2], op[3] * 8:op[3] * 8 + 8].astype(np.int64).reshape(4, 2)
        elif t == "SAD16xN":
            _, v, dd, w, wi, acc, row, col = op
            cw = val[w]
            res = np.stack([np.abs(val[d0] - cw[wi][None, :]).sum(axis=1) + np.abs(val[d1] - cw[wi + 1][None, :]).sum(axis=1) for d0, d1 in dd], axis=1)
            if acc:
                res = res + val[acc]
            val[v] = res
        elif t == "ADDN":
            val[op[1]] = val[op[2]] + val[op[3]]
        elif t == "ADDSHLN":
            val[op[1]] = (val[op[2]] + val[op[3]]) << lsh[op[4]]
        elif t == "SUBN":
            val[op[1]] = val[op[2]] - val[op[3]]
            assert (val[op[1]] >= 0).all()
        elif t == "KEYMINN":
            assert (val[op[2]] < (1 << 32)).all()
            k = (val[op[2]] & keymask) + c.T.astype(np.int64)
            val[op[1]] = k.min(axis=1).astype(np.uint32)
        elif t == "MERGE":
            _, level, m, a, b = op
            A = val[a] if a is not None else MAXK
            B = val[b] if b is not None else MAXK
            perm, role = _perm_level(level)
            val[m] = np.minimum(np.where(role == 1, B, A), np.where(role == 1, A, B)[perm])
        elif t == "ACC":
            best[op[1]] = np.minimum(best[op[1]], val[op[2]])
        else:
            raise ValueError(t)
    return best


def simulate(tree, window, cur, lane_off, c, best):
    """interpret one lane-iteration.
    window: (rows, PDW*4) uint8 LDS image; cur: (64,64) uint8; lane_off[l] = byte offset of lane l's
    (row 0, dword 0) in the flattened window; c: (4, 64) uint32 per-candidate constants;
    best: (N_GROUPS, 64) uint32, updated in place."""
    mult = {"A": np.uint32(MULT_A), "E": np.uint32(2 * MULT_A)}
    flat = window.reshape(-1)
    pad = np.zeros(8, np.uint8)
    flat = np.concatenate([flat, pad])
    val = {}
    MAXK = np.full(64, 0xFFFFFFFF, np.uint32)

    def bytes_at(row, k):
        idx = lane_off + (row * PDW + k) * 4
        return np.stack([flat[idx + i] for i in range(4)], axis=1).astype(np.int32)   # (64, 4)

    for op in tree.ops:
        t = op[0]
        if t == "LDS":
            val[op[1]] = np.concatenate([bytes_at(op[2], op[3]), bytes_at(op[2], op[3] + 1)], axis=1)   # (64, 8) bytes
        elif t in ("CURLD", "BASE", "CURWAIT8"):
            pass
        elif t == "QSAD":
            _, v, pair, cw, half, row, bx, acc = op
            src = val[pair]
            cb = cur[row, bx * 4:bx * 4 + 4].astype(np.int32)
            res = np.zeros((64, 4), np.uint32)
            for j in range(4):
                res[:, j] = np.abs(src[:, j:j + 4] - cb[None, :]).sum(axis=1)
            if acc:
                res = (res + val[acc]) & 0xFFFF
            val[v] = res                                                               # (64, 4) u16 values
        elif t == "PKADD":
            val[op[1]] = (val[op[2]] + val[op[3]]) & 0xFFFF
        elif t == "PKSUB":
            val[op[1]] = (val[op[2]] - val[op[3]]) & 0xFFFF
        elif t == "KEYS":
            val[op[1]] = (val[op[2]].astype(np.uint32) * mult[op[3]] + c.T).astype(np.uint32)
        elif t == "LIN":
            val[op[1]] = (val[op[2]] + val[op[3]] - c.T).astype(np.uint32)
        elif t == "SUB":
            val[op[1]] = (val[op[2]] - val[op[3]] + c.T).astype(np.uint32)
        elif t == "MIN4":
            val[op[1]] = val[op[2]].min(axis=1).astype(np.uint32)
        elif t == "MERGE":
            _, level, m, a, b = op
            A = val[a] if a is not None else MAXK
            B = val[b] if b is not None else MAXK
            perm, role = _perm_level(level)
            keep = np.where(role == 1, B, A)
            give = np.where(role == 1, A, B)
            val[m] = np.minimum(keep, give[perm])
        elif t == "ACC":
            best[op[1]] = np.minimum(best[op[1]], val[op[2]])
    return best


def main():
    for fen in (0, 1):
        tree = Tree(fen).build()
        emit_cpp(tree, os.path.join(OUT_DIR, f"me_tree_fen{fen}.inc"))
        counts = {}
        for op in tree.ops:
            counts[op[0]] = counts.get(op[0], 0) + 1
        valu = (counts.get("PKADD", 0) + counts.get("PKSUB", 0)) * 2 + (counts.get("KEYS", 0) + counts.get("LIN", 0)) * 4 \
            + counts.get("SUB", 0) * 4 + counts.get("MIN4", 0) * 2 + counts.get("ACC", 0)
        merges = [sum(1 for op in tree.ops if op[0] == "MERGE" and op[1] == lv) for lv in range(6)]
        valu += 2 * (merges[0] + merges[1]) + 4 * sum(merges[2:])
        print(f"fen={fen}: {len(tree.ops)} IR ops {counts}; merges/level {merges}; ~{valu} full-rate VALU + "
              f"{counts['QSAD']} qsad per lane-iteration (4 candidates)")
        if fen == 1:
            emit_slotmap(tree, os.path.join(OUT_DIR, "me_slotmap.inc"))
        else:
            ref_map = tree.slot_of_lane()
    assert np.array_equal(ref_map, Tree(1).build().slot_of_lane()), "slot map must not depend on fen"
    for fen in (0, 1):
        t16 = Tree16(fen).build()
        assert np.array_equal(ref_map, t16.slot_of_lane()), "the 16-bit tree must use the same slot map"
        emit_cpp(t16, os.path.join(OUT_DIR, f"me_tree16_fen{fen}.inc"), HEADER16)
        counts = {}
        for op in t16.ops:
            counts[op[0]] = counts.get(op[0], 0) + 1
        print(f"16-bit fen={fen}: {len(t16.ops)} IR ops {counts}")


if __name__ == "__main__":
    sys.exit(main())
