#!/usr/bin/env python3
"""What the lanes of a context buy (include/sbtv.h `sbtv_ctx_set_lanes`): image-iterations/s of batched calls with one
stream (SBTV_LANES=1), the default two lanes, and 3 / 4 lanes (SBTV_LANE_COUNT, experiment knob), each setting in a child
process (the variables are read once), same box.  Prints one markdown table (profiles/r04_lanes.md).

  python tools/bench_lanes.py [--steps 120]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys, time
sys.path.insert(0, os.path.join(%(root)r, "semi-blind-image-deblurring-problems-with-tv_amd"))
for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(v, "1")
import numpy as np
import sbtv
steps = %(steps)d
man = np.load(os.path.join(%(root)r, "tests", "golden", "man_512.npy")).astype(np.float64)
out = {}
def problem(kind, size, p):
    r = max(1, size // 512)
    x = np.tile(man, (r, r))[:size, :size]
    return x, sbtv.demo_setup(kind, x, np.random.default_rng(1).standard_normal(x.shape), evMax=1.0)
def timed(fn, reps=3):
    fn()                                     # workspaces, clocks
    best = None
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best
# SALSA batches
for size, B in ((2048, 4), (2048, 8), (1024, 8), (512, 16)):
    x, st = problem("gaussian", size, (0.4, 0.3))
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
    yb, xb = sbtv.to_device(np.stack([st["y"]] * B)), sbtv.to_device(np.stack([x] * B))
    tau = 0.03 * st["sigma"] ** 2
    f = lambda: sbtv.SALSA_v2(yb, A, tau, "MU", 0.003, "AT", A.T, "LS", A.LS(0.003), "True_x", xb, "ToleranceA", -1.0,
                              "MAXITERA", steps, "TVINITIALIZATION", 1, "TViters", 10, "VERBOSE", 0)
    out[f"SALSA {B} x {size}^2"] = B * steps / timed(f)
    del yb, xb
# FISTA batch
x, st = problem("moffat", 2048, (0.4, 3.5))
A = sbtv.BlurOperator(sbtv.psf_moffat(7, 0.4, 3.5))
yb, xb = sbtv.to_device(np.stack([st["y"]] * 4)), sbtv.to_device(np.stack([x] * 4))
f = lambda: sbtv.my_fista(yb, A, A.T, 0.03 * st["sigma"] ** 2, 1.0, sbtv.TVnorm, sbtv.Psi_TV(25), 1, 0.0, steps // 2, xb)
out["FISTA 4 x 2048^2"] = 4 * (steps // 2) / timed(f)
del yb, xb
# SAPG Laplace, 8 independent 1024^2 images (configs[3] share)
x, st = problem("laplace", 1024, (0.3,))
op = dict(samples=steps // 2, warmup=5, burnIn=2, chambolleit=25, psf_size=7, phi=0.0, gamma=st["gamma"], th_init=0.01,
          min_th=1e-3, max_th=1.0, sigma=st["sigma"], sigma_init=st["sigma_init"], sigma_min=st["sigma_min"],
          sigma_max=st["sigma_max"], d_scale=1.0, d_exp=0.8, fix_sigma=0, seed=1, b=0.3, b_init=0.1, min_b=1e-3, max_b=1.0, fix_b=0)
op["lambda"] = st["lambda"]
y8 = sbtv.to_device(np.stack([st["y"]] * 8))
f = lambda: sbtv.SAPG_algorithm_laplace(y8, op, dict(theta=0.01, b=100.0, sigma=1e4, lam=1.0, gam=1.0))
out["SAPG Laplace 8 x 1024^2"] = 8 * (op["samples"] + op["warmup"] - 2) / timed(f)
# SAPG Gaussian, 4 chains with shared gradients on one 2048^2 image (configs[4] share)
x, st = problem("gaussian", 2048, (0.4, 0.3))
op = dict(samples=steps // 4, warmup=5, burnIn=2, chambolleit=25, psf_size=7, phi=0.0, gamma=st["gamma"], th_init=0.01,
          min_th=1e-3, max_th=1.0, sigma=st["sigma"], sigma_init=st["sigma_init"], sigma_min=st["sigma_min"],
          sigma_max=st["sigma_max"], d_scale=1.0, d_exp=0.8, fix_sigma=0, seed=1, chains=4)
op["lambda"] = st["lambda"]
for q, nm in enumerate(("w1", "w2")):
    op[nm] = (0.4, 0.3)[q]; op[nm + "_init"] = (0.4, 0.3)[q]; op["min_" + nm] = 0.1; op["max_" + nm] = 1.0; op["fix_" + nm] = 1
yd = sbtv.to_device(st["y"])
f = lambda: sbtv.SAPG_algorithm_Guassian(yd, op, dict(theta=0.01, w1=10.0, w2=10.0, sigma=1000.0, lam=1.0, gam=1.0), share_gradients=True)
out["SAPG Gaussian 4 shared chains 2048^2"] = 4 * (op["samples"] + op["warmup"] - 2) / timed(f)
out["switches"] = sbtv.switches()
print(json.dumps(out))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=120)
    a = ap.parse_args()
    settings = [("one stream (SBTV_LANES=1)", {"SBTV_LANES": "1"}), ("default: 2 lanes", {}),
                ("2 lanes, shared chains too (SBTV_LANES=2)", {"SBTV_LANES": "2"}),
                ("3 lanes (SBTV_LANE_COUNT=3)", {"SBTV_LANE_COUNT": "3"}), ("4 lanes (SBTV_LANE_COUNT=4)", {"SBTV_LANE_COUNT": "4"}),
                ("4 lanes, shared chains too", {"SBTV_LANE_COUNT": "4", "SBTV_LANES": "2"})]
    rows = {}
    for name, env in settings:
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "steps": a.steps}], env=e, capture_output=True,
                           text=True, timeout=900)
        if r.returncode != 0:
            print(name, "FAILED", r.stderr[-800:], file=sys.stderr)
            continue
        rows[name] = json.loads(r.stdout.strip().splitlines()[-1])
        print("#", name, rows[name], file=sys.stderr, flush=True)
    keys = [k for k in next(iter(rows.values())) if k != "switches"]
    print("| setting | " + " | ".join(keys) + " |")
    print("|---|" + "---|" * len(keys))
    for name, v in rows.items():
        print(f"| {name} | " + " | ".join(f"{v[k]:.0f}" for k in keys) + " |")


if __name__ == "__main__":
    main()
