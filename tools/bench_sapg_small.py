#!/usr/bin/env python3
"""SAPG iterations/s at the reference's demo size (512^2) and around it: device-resident parameter loop (default)
against the host-side loop (one synchronisation per iteration), fixed and estimated PSF parameters, eager launches
and hipGraph replay (SBTV_GRAPH is read once per process, so every variant is a child process).
  python tools/bench_sapg_small.py [--sizes 256 512 1024] [--iters 600]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "semi-blind-image-deblurring-problems-with-tv_amd"))
import sbtv
size, iters, fixed, host_loop = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
man = np.load(os.path.join(%(root)r, "tests", "golden", "man_512.npy")).astype(np.float64)
r = max(1, size // 512)
x = np.tile(man, (r, r))[:size, :size]
ctx = sbtv.Context(0)
st = sbtv.demo_setup("gaussian", x, np.random.default_rng(1).standard_normal(x.shape), evMax=0.99, ctx=ctx)
op = dict(samples=iters + 1, warmup=0, burnIn=2, psf_size=7, phi=0.0, gamma=st["gamma"], th_init=0.01, min_th=1e-3,
          max_th=1.0, sigma=st["sigma"], sigma_init=st["sigma_init"], sigma_min=st["sigma_min"], sigma_max=st["sigma_max"],
          d_scale=1.0, d_exp=0.8, fix_sigma=0, seed=7, w1=0.4, w2=0.3, w1_init=0.4 if fixed else 0.5, w2_init=0.3,
          min_w1=0.1, min_w2=0.1, max_w1=1.0, max_w2=1.0, fix_w1=fixed, fix_w2=fixed)
op["lambda"] = st["lambda"]
c = dict(theta=0.01, w1=10.0, w2=10.0, sigma=1000.0, lam=1.0, gam=1.0)
y = sbtv.to_device(st["y"], "cuda:0")
sbtv.SAPG_algorithm_Guassian(y, dict(op, samples=20), c, ctx=ctx, host_loop=bool(host_loop))
best = 0.0
for rep in range(3):
    t0 = time.perf_counter()
    sbtv.SAPG_algorithm_Guassian(y, op, c, ctx=ctx, host_loop=bool(host_loop))
    best = max(best, iters / (time.perf_counter() - t0))
print(json.dumps({"it_per_s": best}))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[256, 512, 1024])
    ap.add_argument("--iters", type=int, default=600)
    a = ap.parse_args()
    rows = []
    for size in a.sizes:
        for fixed in (1, 0):
            row = {"size": size, "psf": "fixed" if fixed else "estimated"}
            for name, host_loop, graph in (("host loop", 1, "0"), ("device loop", 0, "0"), ("device loop + graph", 0, "1")):
                env = dict(os.environ, SBTV_GRAPH=graph)
                r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, str(size), str(a.iters), str(fixed),
                                    str(host_loop)], env=env, capture_output=True, text=True, timeout=900)
                if r.returncode != 0:
                    print(r.stderr[-2000:], file=sys.stderr)
                    row[name] = None
                else:
                    row[name] = round(json.loads(r.stdout.strip().splitlines()[-1])["it_per_s"], 1)
            rows.append(row)
            print(json.dumps(row), flush=True)
    print("\n| size | PSF parameters | host loop it/s | device loop it/s | device loop + hipGraph it/s |\n|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['size']}² | {r['psf']} | {r['host loop']} | {r['device loop']} | {r['device loop + graph']} |")


if __name__ == "__main__":
    main()
