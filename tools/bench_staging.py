#!/usr/bin/env python3
"""Rate of the pageable host <-> device staging (csrc/ctx.hip `stage_copy`) against the plain hipMemcpyAsync path, per
number of copy lanes (SBTV_STAGE_THREADS, read once: child processes).  128 MB arrays, 5 repetitions after a warm-up."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.join(%(root)r, "semi-blind-image-deblurring-problems-with-tv_amd"))
import numpy as np
import sbtv
from sbtv import _lib as L
ctx = sbtv.Context(0)
n = 128 << 20
src = np.random.default_rng(0).integers(0, 256, n, dtype=np.uint8)
dst = np.zeros(n, dtype=np.uint8)
p = C.c_void_p()
ctx.check(ctx.lib.sbtv_malloc(ctx.h, n, C.byref(p)))
def rate(fn):
    fn(); fn()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return n / min(ts) / 1e9, n / (sum(ts) / len(ts)) / 1e9
h2d = rate(lambda: ctx.check(ctx.lib.sbtv_memcpy_h2d(ctx.h, p, L.vptr(src), n)))
d2h = rate(lambda: ctx.check(ctx.lib.sbtv_memcpy_d2h(ctx.h, L.vptr(dst), p, n)))
assert np.array_equal(src, dst)
print(json.dumps(dict(h2d_best=h2d[0], h2d_mean=h2d[1], d2h_best=d2h[0], d2h_mean=d2h[1])))
"""
print("| SBTV_STAGE_THREADS | host -> device GB/s (best / mean) | device -> host GB/s (best / mean) |")
print("|---|---|---|")
for t in ("0", "1", "2", "3", "4"):
    e = dict(os.environ, SBTV_STAGE_THREADS=t)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=e, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        print(t, "FAILED", r.stderr[-500:])
        continue
    v = json.loads(r.stdout.strip().splitlines()[-1])
    label = "0 (plain hipMemcpyAsync)" if t == "0" else t
    print(f"| {label} | {v['h2d_best']:.1f} / {v['h2d_mean']:.1f} | {v['d2h_best']:.1f} / {v['d2h_mean']:.1f} |", flush=True)
