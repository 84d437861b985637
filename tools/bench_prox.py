#!/usr/bin/env python3
"""Micro-benchmark of the TV prox (chambolle_prox_TV_stop) on device-resident images.

  python tools/bench_prox.py [--size 2048] [--batch 1] [--iters 10] [--reps 50]

Reports ms per prox call, ms per Chambolle iteration and the algorithmic GB/s
((40 K + 32) B/pixel).  SBTV_FUSED_VARIANT="cj,minw" and
SBTV_SINGLE_STEP=1 select kernel variants (read once at library start)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--cols", type=int, default=0, help="columns N (default: size)")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    import numpy as np
    import torch
    import sbtv
    ctx = sbtv.default_context(0)
    rng = np.random.default_rng(0)
    man = np.load(os.path.join(ROOT, "tests", "golden", "man_512.npy")).astype(np.float64)
    r = max(1, a.size // 512)
    ncols = a.cols or a.size
    img = np.tile(man, (r, r))[:a.size, :ncols]
    g = np.stack([img + rng.standard_normal(img.shape) for _ in range(a.batch)])
    gd = sbtv.to_device(g)
    lam = 10.0
    f, px, py = sbtv.chambolle_prox_TV_stop(gd, "lambda", lam, "maxiter", a.iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        f, px, py = sbtv.chambolle_prox_TV_stop(gd, "lambda", lam, "maxiter", a.iters, "dualvars", (px, py))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    P = a.size * ncols * a.batch
    byts = (40 * a.iters + 32) * P
    print(f"variant={os.environ.get('SBTV_FUSED_VARIANT', 'default')} single={os.environ.get('SBTV_SINGLE_STEP', '0')} "
          f"size={a.size}x{ncols} batch={a.batch} K={a.iters}: {dt * 1e3:.3f} ms/prox (incl. python + copies), "
          f"{dt * 1e6 / a.iters:.1f} us/iter, {dt * 1e12 / a.iters / P:.2f} ps/pixel/iter, {byts / dt / 1e9:.0f} GB/s algorithmic")


if __name__ == "__main__":
    main()
