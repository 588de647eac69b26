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
# WARM=n: n untimed searches first -- the power management takes tens of ms of load to bring an idle GPU to its sustained clock, and the
# handful of short refinement launches timed below would otherwise run (and be timed) on the way up
for _ in range(int(os.environ.get("WARM", "0")) + 1):
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
if os.environ.get("HMME_TIMELINE"):   # a library built with -DME_FRAC_T_TIMELINE left (start, end, workgroup) of every job in the cost table
    import numpy as np
    c = d_c.cpu().numpy().view(np.uint32)
    t0 = c[:, 0].astype(np.uint64) | (c[:, 1].astype(np.uint64) << 32)
    t1 = c[:, 2].astype(np.uint64) | (c[:, 3].astype(np.uint64) << 32)
    wg = c[:, 4]
    base = t0.min()
    s_us, e_us = (t0 - base) / 100.0, (t1 - base) / 100.0
    dur = e_us - s_us
    per_wg = np.bincount(wg, minlength=int(wg.max()) + 1)
    first = np.array([s_us[wg == g].min() for g in np.unique(wg)])
    out["timeline"] = {"jobs": int(n), "kernel_span_us": round(float(e_us.max()), 1), "job_us_mean_min_max_p95": [round(float(v), 1) for v in (dur.mean(), dur.min(), dur.max(), np.percentile(dur, 95))],
                       "first_job_start_us_mean_max_p95": [round(float(v), 1) for v in (first.mean(), first.max(), np.percentile(first, 95))],
                       "jobs_per_workgroup_min_max": [int(per_wg[per_wg > 0].min()), int(per_wg.max())], "workgroups": int((per_wg > 0).sum()),
                       "last_start_us": round(float(s_us.max()), 1), "longest_jobs": [int(v) for v in np.argsort(-dur)[:12]],
                       "mean_us_bottom_ctu_row": round(float(dur[-((w + 63) // 64):].mean()), 1),
                       "shader_clock_ghz_mean_min_max": [round(float(v), 3) for v in ((c[:, 12] / (dur * 1000.0)).mean(), (c[:, 12] / (dur * 1000.0)).min(), (c[:, 12] / (dur * 1000.0)).max())],
                       "latest_ends(job,start_us,dur_us,end_us)": [[int(j), round(float(s_us[j]), 1), round(float(dur[j]), 1), round(float(e_us[j]), 1)] for j in np.argsort(-e_us)[:8]],
                       "starts_histogram_10us": np.histogram(s_us, bins=np.arange(0, float(e_us.max()) + 10, 10))[0].tolist(),
                       "busy_share": round(float(dur.sum() / (e_us.max() * (per_wg > 0).sum())), 3),
                       "phase_us_mean": dict(zip(("setup", "lists0", "items0", "winners0", "lists1", "items1", "winners1"), (round(float(c[:, 5 + i].mean()) / 100.0, 2) for i in range(7))))}
frac = (d_q.to(torch.int32) - 4 * d_mv.to(torch.int32)).abs().amax().item()
import zlib
crc = zlib.crc32(d_c.cpu().numpy().tobytes(), zlib.crc32(d_q.cpu().numpy().tobytes()))   # of the last launch's tables (SAD distortion)
print(json.dumps({"size": f"{w}x{h}", "bit_depth": bd, "content": content, "refine_ms": out, "slots_per_s_hadamard": round(n * 593 / (out["hadamard"] * 1e-3)), "max_frac_offset_qpel": frac,
                  "tables_crc32": crc}))
