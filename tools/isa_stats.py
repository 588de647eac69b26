#!/usr/bin/env python3
"""instruction mix of one kernel in the built ISA (hm-opencl_amd/csrc/build/*.s): tools/isa_stats.py <mangled-name prefix> [file]
prints the opcode histogram, the SGPR-spill traffic (v_readlane / v_writelane) and s_nop counts, in total and inside the largest
basic block (the generated straight-line body of a lane-iteration)"""
import collections, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "hm-opencl_amd/csrc/build/hmme-hip-amdgcn-amd-amdhsa-gfx950.s")
s = open(f).read()
m = re.search(r"^(%s[^\n:]*):" % re.escape(sys.argv[1]), s, re.M)
start = m.end(); end = s.index(".Lfunc_end", start)
body = s[start:end].split("\n")
op = lambda l: l.strip().split(" ")[0] if l.strip() and not l.strip().startswith((";", ".")) else ""
idx = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)] + [len(body)]
blocks = sorted(((b - a, a, b) for a, b in zip([0] + idx[:-1], idx)), reverse=True)
n, a, b = blocks[0]
for name, lines in (("whole kernel", body), ("largest basic block", body[a:b])):
    c = collections.Counter(op(l) for l in lines if op(l))
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    nop_cycles = sum(int(l.split()[1]) + 1 for l in lines if op(l) == "s_nop")
    print(f"{name}: {sum(c.values())} instructions, {valu} VALU, {c['s_nop']} s_nop ({nop_cycles} cycles), v_readlane {c['v_readlane_b32']}, v_writelane {c['v_writelane_b32']}, "
          f"s_waitcnt {c['s_waitcnt']}, ds {sum(v for k, v in c.items() if k.startswith('ds_'))}, s_load {sum(v for k, v in c.items() if k.startswith('s_load'))}")
    print("   " + ", ".join(f"{k} {v}" for k, v in c.most_common(28)))
