#!/usr/bin/env python3
"""per-workgroup timeline of me_search_kernel<FEN, 0> from a library built with -DME_SEARCH_T_TIMELINE (tools/build_variant.sh tls
-DME_SEARCH_T_TIMELINE; HMME_LIB=<that library>): every workgroup leaves its start, window-staged, task-counter-dry (per wave) and end
times (100 MHz wall clock) plus its hardware id in the SAD table of its own job -- a timing-only build, the tables are not results.
    HMME_LIB=... python tools/search_timeline.py 1920x1080 [pairs]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))
import numpy as np
import torch
from hmme import api, synth
w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080").split("x"))
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sr, bd = 64, 8
cur, ref, _ = synth.make_pair(w, h, seed=1234, bit_depth=bd)
m = synth.MARGIN
eng = api.Engine(0, 128); eng.set_lambda(57.9)
pc, pr = eng.plane(w, h, bd), eng.plane(w, h, bd)
pc.upload_pel(cur, (m, m)); pr.upload_pel(ref, (m, m))
n = api.load().hmme_num_ctus(w, h)
dev = torch.device("cuda", 0)
buf = torch.zeros((2, pairs, n, 593), dtype=torch.int32, device=dev)
fp = api.FrameParams(sr, 1, bd, 0, n)
st = torch.cuda.current_stream().cuda_stream
for _ in range(30):
    eng.search_pairs_device([pc] * pairs, [pr] * pairs, fp, None, buf[0].data_ptr(), buf[1].data_ptr(), st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    eng.search_pairs_device([pc] * pairs, [pr] * pairs, fp, None, buf[0].data_ptr(), buf[1].data_ptr(), st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
c = buf[1].cpu().numpy().view(np.uint32).reshape(pairs * n, 593)
t0 = c[:, 0].astype(np.uint64) | (c[:, 1].astype(np.uint64) << 32)
base = t0.min()
start = (t0 - base) / 100.0
staged, alld, end = c[:, 2] / 100.0, c[:, 3] / 100.0, c[:, 4] / 100.0
dry = c[:, 8:12] / 100.0
hw, xcc = c[:, 5], c[:, 6] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7     # gfx9 HW_ID: wave 3:0 simd 5:4 pipe 7:6 cu 11:8 sh 12 se 15:13 (gfx950: se 3 bits)
unit = xcc.astype(np.int64) * 1000 + se * 100 + sh * 16 + cu
uniq, cnt = np.unique(unit, return_counts=True)
span = (start + end).max()
pct = lambda a: [round(float(v), 1) for v in (a.min(), np.percentile(a, 50), np.percentile(a, 95), a.max())]
out = {"size": f"{w}x{h}", "pairs": pairs, "workgroups": int(pairs * n), "ms_per_launch_events": round(ms, 4), "kernel_span_us": round(float(span), 1),
       "start_us_min_p50_p95_max": pct(start), "staging_us": pct(staged), "tasks_us(start->all waves dry)": pct(alld), "lifetime_us": pct(end),
       "wave_dry_spread_us(max-min over the 4 waves)": pct(dry.max(1) - dry.min(1)), "end_us": pct(start + end),
       "distinct_cus": int(len(uniq)), "workgroups_per_cu_hist": {int(k): int(v) for k, v in zip(*np.unique(cnt, return_counts=True))},
       "xcc_hist": {int(k): int(v) for k, v in zip(*np.unique(xcc, return_counts=True))},
       "busy_share": round(float(end.sum() / (span * max(1, 2 * len(uniq)))), 3)}
late = start > 50
if late.any():
    out["late_starters(>50us)"] = {"count": int(late.sum()), "lifetime_us": pct(end[late]), "start_us": pct(start[late])}
print(json.dumps(out))
