#!/usr/bin/env python3
"""time of the fractional-pel refinement kernel on a 2160p picture pair (device-resident, torch events)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))
import torch
from hmme import api, synth
w, h, sr = 3840, 2160, int(os.environ.get("SR", "64"))
if len(sys.argv) > 1: w, h = (int(v) for v in sys.argv[1].split("x"))
bd = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cur, ref, _ = synth.make_pair(w, h, seed=1234, bit_depth=bd)
content = sys.argv[3] if len(sys.argv) > 3 else "coherent"
if content == "mixed":   # small regions with their own motion and stronger noise: sharing inside a CTU is partial
    cur, ref, _ = synth.make_pair(w, h, seed=1234, bit_depth=bd, max_mv=10, region=24, noise_sigma=6.0)
if content == "noise":   # unrelated pictures: nearly every slot has its own integer MV, nothing to share
    import numpy as np
    rng = np.random.default_rng(5)
    cur = np.ascontiguousarray(np.pad(rng.integers(0, 1 << bd, size=(h, w)), synth.MARGIN, mode="edge").astype(cur.dtype))
    ref = np.ascontiguousarray(np.pad(rng.integers(0, 1 << bd, size=(h, w)), synth.MARGIN, mode="edge").astype(ref.dtype))
m = synth.MARGIN
eng = api.Engine(0, 128); eng.set_lambda(57.9)
pc, pr = eng.plane(w, h, bd), eng.plane(w, h, bd)
pc.upload_pel(cur, (m, m)); pr.upload_pel(ref, (m, m))
n = api.load().hmme_num_ctus(w, h)
dev = torch.device("cuda", 0)
d_mv = torch.zeros((n, 593, 2), dtype=torch.int16, device=dev); d_sad = torch.zeros((n, 593), dtype=torch.int32, device=dev)
d_q = torch.zeros_like(d_mv); d_c = torch.zeros_like(d_sad)
fp = api.FrameParams(sr, 1, bd, 0, n)
st = torch.cuda.current_stream().cuda_stream
eng.search_frame_device(pc, pr, fp, None, d_mv.data_ptr(), d_sad.data_ptr(), st)
out = {}
for had in (1, 0):
    for _ in range(2):
        eng.refine_frame_multi_device(pc, [pr], fp, None, d_mv.data_ptr(), had, d_q.data_ptr(), d_c.data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        eng.refine_frame_multi_device(pc, [pr], fp, None, d_mv.data_ptr(), had, d_q.data_ptr(), d_c.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    out["hadamard" if had else "sad"] = round(e0.elapsed_time(e1) / 5, 3)
frac = (d_q.to(torch.int32) - 4 * d_mv.to(torch.int32)).abs().amax().item()
print(json.dumps({"size": f"{w}x{h}", "bit_depth": bd, "content": content, "refine_ms": out, "slots_per_s_hadamard": round(n * 593 / (out["hadamard"] * 1e-3)), "max_frac_offset_qpel": frac}))
