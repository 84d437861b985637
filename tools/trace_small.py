#!/usr/bin/env python3
"""SALSA_v2 on one 512 x 512 image for a fixed number of outer iterations (host-side API trace target:
`rocprofv3 --hip-trace --kernel-trace --stats -- python3 tools/trace_small.py`)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd"))
import torch
import sbtv
import bench

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = sbtv.default_context(0)
x, y, sigma, noise = bench.make_problem(1, size)
yd, xd = sbtv.to_device(y), sbtv.to_device(x)
A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3), ctx=ctx)
mu, tau = 0.003, 0.03 * sigma ** 2


def solve(k):
    return sbtv.SALSA_v2(yd, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xd, "ToleranceA", -1.0, "MAXITERA", k,
                         "TVINITIALIZATION", 1, "TViters", 10, ctx=ctx)


solve(30)
torch.cuda.synchronize()
t0 = time.perf_counter()
solve(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{size}^2: {steps / dt:.0f} outer iterations/s ({1e6 * dt / steps:.1f} us per iteration)")
