#!/usr/bin/env python3
"""Latency of the per-CTU drop-in call (hmme_search_ctu == TEncOpenCL::calcMotionVectors): what HM pays per
(CTU, reference picture) when it drives the engine the way it drives the reference's OpenCL module."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))
from hmme import api  # noqa: E402

rng = np.random.default_rng(0)
out = {}
eng = api.Engine(0, 128)
eng.set_lambda(57.9)
for name, sr, bd in (("8bit_sr64", 64, 8), ("8bit_sr8", 8, 8), ("8bit_sr4_bipred", 4, 8), ("10bit_sr64", 64, 10)):
    side = 64 + 2 * sr + 8
    maxv = (1 << bd) - 1
    cur = rng.integers(0, maxv + 1, size=(64, 64)).astype(np.int16)
    if "bipred" in name:
        cur = (2 * cur - rng.integers(0, maxv + 1, size=(64, 64))).astype(np.int16)
    ref = rng.integers(0, maxv + 1, size=(side, side)).astype(np.int16)
    p = api.SearchParams(-sr, -sr, sr, sr, 5, -3, 1, bd)
    for _ in range(5):
        eng.search_ctu(cur, (0, 0), ref, (sr + 4, sr + 4), p)
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        eng.search_ctu(cur, (0, 0), ref, (sr + 4, sr + 4), p)
    out[name + "_ms_per_call"] = round((time.perf_counter() - t0) / n * 1e3, 4)
print(json.dumps(out))
