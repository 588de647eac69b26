#!/usr/bin/env python3
"""per-workgroup timeline of me_search16_kernel from a library built with -DME_SEARCH_T_TIMELINE (tools/build_variant.sh tls
-DME_SEARCH_T_TIMELINE): HMME_LIB=<that library> python tools/search16_timeline.py [WxH] [bit depth] [search range]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))
import numpy as np
import torch
from hmme import api, synth
w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3840x2160").split("x"))
bd = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sr = int(sys.argv[3]) if len(sys.argv) > 3 else 128
cur, ref, _ = synth.make_pair(w, h, seed=1234, bit_depth=bd)
m = synth.MARGIN
eng = api.Engine(0, 128); eng.set_lambda(57.9)
pc, pr = eng.plane(w, h, bd), eng.plane(w, h, bd)
pc.upload_pel(cur, (m, m)); pr.upload_pel(ref, (m, m))
n = api.load().hmme_num_ctus(w, h)
dev = torch.device("cuda", 0)
buf = torch.zeros((2, 1, n, 593), dtype=torch.int32, device=dev)
fp = api.FrameParams(sr, 1, bd, 0, n)
st = torch.cuda.current_stream().cuda_stream
for _ in range(6):
    eng.search_pairs_device([pc], [pr], fp, None, buf[0].data_ptr(), buf[1].data_ptr(), st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4):
    eng.search_pairs_device([pc], [pr], fp, None, buf[0].data_ptr(), buf[1].data_ptr(), st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 4
L = api.load()
raw = np.zeros(16384 * 12, dtype=np.uint32)
L.hmme_test_timeline16.argtypes = [C.c_void_p, C.c_size_t]
assert L.hmme_test_timeline16(raw.ctypes.data, raw.nbytes) == 0
t = raw.reshape(16384, 12)
t = t[t[:, 10] > 0]
t0 = t[:, 0].astype(np.uint64) | (t[:, 1].astype(np.uint64) << 32)
start = (t0 - t0.min()) / 100.0
ph = t[:, 2:9].astype(np.float64) / 100.0     # staged0, wave0 dry0, all dry0, staged1, wave0 dry1, all dry1, end
pct = lambda a: [round(float(v), 1) for v in (a.min(), np.percentile(a, 50), np.percentile(a, 95), a.max())]
out = {"size": f"{w}x{h}", "bit_depth": bd, "sr": sr, "ms_per_launch": round(ms, 3), "workgroups_seen": int(len(t)), "strip_rows": sorted(set(int(v) for v in t[:, 10])),
       "span_us": round(float((start + ph[:, 6]).max()), 1), "lifetime_us": pct(ph[:, 6]),
       "stage0_us": pct(ph[:, 0]), "iter0_us(wave 0)": pct(ph[:, 1] - ph[:, 0]), "barrier_wait0_us": pct(ph[:, 2] - ph[:, 1]),
       "stage1_us": pct(ph[:, 3] - ph[:, 2]), "iter1_us(wave 0)": pct(ph[:, 4] - ph[:, 3]), "barrier_wait1_us": pct(ph[:, 5] - ph[:, 4]), "end_us": pct(ph[:, 6] - ph[:, 5]),
       "last_start_us": round(float(start.max()), 1),
       "busy_share": round(float(ph[:, 6].sum() / ((start + ph[:, 6]).max() * 512)), 3)}
print(json.dumps(out))
