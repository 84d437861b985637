#!/usr/bin/env python3
"""Launch-bound regime (images up to 1024^2): SALSA outer-iterations/s and SAPG iterations/s under the host-side
hooks that matter there - SBTV_SPIN (poll instead of blocking in the per-iteration host wait), SBTV_GRAPH (replay a
captured hipGraph per iteration).  One child process per variant; prints a markdown table (profiles/r02_small_sizes.md).

  python tools/bench_small.py [--sizes 256 512 1024] [--steps 1500]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "semi-blind-image-deblurring-problems-with-tv_amd"))
import numpy as np, torch
import sbtv, bench
ctx = sbtv.default_context(0)
out = {}
for size in %(sizes)r:
    x, y, sigma, noise = bench.make_problem(1, size)
    yd, xd = sbtv.to_device(y), sbtv.to_device(x)
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3), ctx=ctx)
    mu, tau = 0.003, 0.03 * sigma ** 2
    def solve(k):
        return sbtv.SALSA_v2(yd, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xd, "ToleranceA", -1.0,
                             "MAXITERA", k, "TVINITIALIZATION", 1, "TViters", 10, ctx=ctx)
    solve(30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solve(%(steps)d)
    torch.cuda.synchronize()
    out["SALSA it/s %%d^2" %% size] = %(steps)d / (time.perf_counter() - t0)
    if size <= 512:
        st = sbtv.demo_setup("gaussian", x, noise, evMax=0.99, ctx=ctx)
        op = dict(samples=401, warmup=0, burnIn=2, psf_size=7, phi=0.0, gamma=st["gamma"], th_init=0.01, min_th=1e-3,
                  max_th=1.0, sigma=st["sigma"], sigma_init=st["sigma_init"], sigma_min=st["sigma_min"],
                  sigma_max=st["sigma_max"], d_scale=1.0, d_exp=0.8, fix_sigma=0, seed=1,
                  w1=0.4, w1_init=0.4, min_w1=0.1, max_w1=1.0, fix_w1=1, w2=0.3, w2_init=0.3, min_w2=0.1, max_w2=1.0, fix_w2=1)
        op["lambda"] = st["lambda"]
        c = dict(theta=0.01, w1=10.0, w2=10.0, sigma=1000.0, lam=1.0, gam=1.0)
        ysd = sbtv.to_device(st["y"])
        sbtv.SAPG_algorithm_Guassian(ysd, dict(op, samples=20), c, ctx=ctx)
        t0 = time.perf_counter()
        sbtv.SAPG_algorithm_Guassian(ysd, op, c, ctx=ctx)
        out["SAPG it/s %%d^2 (fixed PSF)" %% size] = 400 / (time.perf_counter() - t0)
print("LABJSON" + json.dumps(out))
"""
VARIANTS = {"default (block)": {}, "SBTV_SPIN=1 (poll)": {"SBTV_SPIN": "1"}, "SBTV_GRAPH=1": {"SBTV_GRAPH": "1"},
            "SBTV_GRAPH=1 SBTV_SPIN=1": {"SBTV_GRAPH": "1", "SBTV_SPIN": "1"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[256, 512, 1024])
    ap.add_argument("--steps", type=int, default=1500)
    a = ap.parse_args()
    res = {}
    for name, envv in VARIANTS.items():
        env = {k: v for k, v in os.environ.items() if k not in ("SBTV_SPIN", "SBTV_GRAPH")}
        env.update(envv)
        r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, sizes=a.sizes, steps=a.steps)], env=env,
                           capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("LABJSON")]
        if r.returncode != 0 or not line:
            print(f"variant {name!r} failed:\n{r.stderr[-2000:]}", file=sys.stderr)
            continue
        res[name] = json.loads(line[0][7:])
    keys = sorted({k for v in res.values() for k in v})
    print("| quantity | " + " | ".join(res) + " |")
    print("|---|" + "---|" * len(res))
    for k in keys:
        print(f"| {k} | " + " | ".join(f"{res[n].get(k, float('nan')):.0f}" for n in res) + " |")


if __name__ == "__main__":
    main()
