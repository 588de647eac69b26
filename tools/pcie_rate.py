#!/usr/bin/env python3
"""PCIe-inclusive rate of the frame path: host `Pel` planes in, host tables out.
Not the benchmark value (bench.py times with inputs resident in HBM); printed for the record."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))
sys.path.insert(0, ROOT)
from hmme import api, synth  # noqa: E402
import bench  # noqa: E402

w, h, sr = 3840, 2160, 64
cur, ref, _ = synth.make_pair(w, h, seed=1234)
m = synth.MARGIN
eng = api.Engine(0, 128)
eng.set_lambda(57.9)
pc, pr = eng.plane(w, h), eng.plane(w, h)
for _ in range(2):   # warm-up (allocations, first-touch)
    pc.upload_pel(cur, (m, m)); pr.upload_pel(ref, (m, m)); eng.search_frame(pc, pr, sr)
n = 5
t0 = time.perf_counter()
for _ in range(n):
    pc.upload_pel(cur, (m, m)); pr.upload_pel(ref, (m, m))
    mv, sad = eng.search_frame(pc, pr, sr)
dt = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    mv, sad = eng.search_frame(pc, pr, sr)
dt_res = (time.perf_counter() - t0) / n
# the same with the source planes page-locked once (hmme_host_register: what an encoder does for its picture buffers)
eng.host_register(cur); eng.host_register(ref)
pc.upload_pel(cur, (m, m)); pr.upload_pel(ref, (m, m))
t0 = time.perf_counter()
for _ in range(n):
    pc.upload_pel(cur, (m, m)); pr.upload_pel(ref, (m, m))
    mv, sad = eng.search_frame(pc, pr, sr)
dt_reg = (time.perf_counter() - t0) / n
eng.host_unregister(cur); eng.host_unregister(ref)
sads = bench.work_4x4_sads(api, w, h, sr)
print(json.dumps({"registered_planes_upload_search_download_ms": round(dt_reg * 1e3, 3), "gsad_per_s_registered": round(sads / dt_reg / 1e9, 1),
                  "upload_both_planes_search_download_ms": round(dt * 1e3, 3), "gsad_per_s_pcie_inclusive": round(sads / dt / 1e9, 1),
                  "search_download_only_ms": round(dt_res * 1e3, 3), "gsad_per_s_planes_resident_results_to_host": round(sads / dt_res / 1e9, 1)}))
