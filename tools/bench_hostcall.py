#!/usr/bin/env python3
"""What a host-pointer call costs beyond its loop (the path a MATLAB host always takes): sbtv_SALSA_v2 on 4 x 2048^2 images
with MAXITERA = 2, host buffers in (y, true) and out (x), (a) the same arrays every call, (b) FRESH arrays every call
(newly allocated and filled, as a MATLAB host passes them), with the plain pageable hipMemcpyAsync (SBTV_STAGE_THREADS=0)
and with the copy lanes.  Prints one markdown table; each setting in a child process."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "semi-blind-image-deblurring-problems-with-tv_amd"))
for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(v, "1")
import numpy as np
import sbtv, bench
x1, y1, s1, _ = bench.make_problem(seed=1)
ctx = sbtv.Context(0)
A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, *bench.W_TRUE), ctx=ctx)
mu, tau = bench.THETA / 10, bench.THETA * s1 ** 2
B = 4
def cm(a):            # column-major images: the mirror passes the buffer on without a layout copy
    return np.stack([np.ascontiguousarray(a.T)] * B).transpose(0, 2, 1)
def run(ys, xs, n):
    return sbtv.SALSA_v2(ys, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xs, "StopCriterion", 1, "ToleranceA", -1.0,
                         "MAXITERA", n, "TVINITIALIZATION", 1, "TViters", 10, ctx=ctx)
ys, xs = cm(y1), cm(x1)
run(ys, xs, 2); run(ys, xs, 2)
def timed(fresh, n):
    ts = []
    for _ in range(5):
        a, b = (cm(y1), cm(x1)) if fresh else (ys, xs)
        t0 = time.perf_counter(); run(a, b, n); ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts), 1e3 * sorted(ts)[len(ts) // 2]
g = sbtv.Group([0, 0])
def run_g(ys, xs, n):
    return sbtv.SALSA_v2(ys, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xs, "StopCriterion", 1, "ToleranceA", -1.0,
                         "MAXITERA", n, "TVINITIALIZATION", 1, "TViters", 10, ctx=g)
run_g(ys, xs, 2); run_g(ys, xs, 2)
tg = []
for _ in range(5):
    t0 = time.perf_counter(); run_g(ys, xs, 2); tg.append(1e3 * (time.perf_counter() - t0))
yd, xd = sbtv.to_device(ys), sbtv.to_device(xs)
run(yd, xd, 2)
import torch
t0 = time.perf_counter(); run(yd, xd, 2); torch.cuda.synchronize(); dev = 1e3 * (time.perf_counter() - t0)
out = dict(same=timed(False, 2), fresh=timed(True, 2), device_ms=dev, stage=ctx.stage_stats(), group=(min(tg), sorted(tg)[2]))
print(json.dumps(out))
"""
print("| SBTV_STAGE_THREADS | same host arrays every call: ms (best / median) | fresh host arrays every call: ms (best / median) | group of two virtual shards, same arrays: ms | device-resident call ms |")
print("|---|---|---|---|---|")
for t in ("0", "2", "4"):
    e = dict(os.environ, SBTV_STAGE_THREADS=t)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=e, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        print(t, "FAILED", r.stderr[-800:])
        continue
    v = json.loads(r.stdout.strip().splitlines()[-1])
    label = "0 (plain hipMemcpyAsync)" if t == "0" else t
    print(f"| {label} | {v['same'][0]:.1f} / {v['same'][1]:.1f} | {v['fresh'][0]:.1f} / {v['fresh'][1]:.1f} | {v['group'][0]:.1f} / {v['group'][1]:.1f} | {v['device_ms']:.1f} |", flush=True)
