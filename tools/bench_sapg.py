#!/usr/bin/env python3
"""Secondary benchmark: SAPG (MYULA) iterations/s per GPU for the shapes of BASELINE configs 4 and 5.

  python tools/bench_sapg.py --config 4   # 8 independent 1024x1024 images, Laplace PSF (one GPU's share of 64)
  python tools/bench_sapg.py --config 5   # 4 MYULA chains on one 2048x2048 image, Gaussian PSF (share of 32)
  python tools/bench_sapg.py --config 3   # FISTA + TV prox, 2048x2048, Moffat PSF
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd"))
import numpy as np
import sbtv


def image(size, seed=0):
    man = np.load(os.path.join(ROOT, "tests", "golden", "man_512.npy")).astype(np.float64)
    r = max(1, size // 512)
    return np.tile(man, (r, r))[:size, :size]


def op_struct(kind, st, samples, warmup, burnIn):
    d = {"gaussian": dict(names=("w1", "w2"), init=(0.5, 0.3), pmin=(0.1, 0.1), pmax=(1.0, 1.0), fix=(1, 1),
                          c=dict(theta=0.01, w1=10.0, w2=10.0, sigma=1000.0)),
         "laplace": dict(names=("b",), init=(0.1,), pmin=(1e-3,), pmax=(1.0,), fix=(0,),
                         c=dict(theta=0.01, b=100.0, sigma=1e4))}[kind]
    op = dict(samples=samples, warmup=warmup, burnIn=burnIn, psf_size=7, phi=0.0, gamma=st["gamma"],
              th_init=0.01, min_th=1e-3, max_th=1.0, sigma=st["sigma"], sigma_init=st["sigma_init"],
              sigma_min=st["sigma_min"], sigma_max=st["sigma_max"], d_scale=1.0, d_exp=0.8, fix_sigma=0)
    op["lambda"] = st["lambda"]
    for q, nm in enumerate(d["names"]):
        op[nm] = st["p_true"][q]
        op[nm + "_init"] = st["p_true"][q] if d["fix"][q] else d["init"][q]
        op["min_" + nm], op["max_" + nm], op["fix_" + nm] = d["pmin"][q], d["pmax"][q], d["fix"][q]
    c = dict(d["c"], lam=1.0, gam=1.0)
    return op, c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=5)
    ap.add_argument("--iters", type=int, default=40)
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    if a.config == 3:
        x = image(2048)
        st = sbtv.demo_setup("moffat", x, rng.standard_normal(x.shape), evMax=1.0)
        A = sbtv.BlurOperator(sbtv.psf_moffat(7, 0.4, 3.5))
        yd, xd = sbtv.to_device(st["y"]), sbtv.to_device(x)
        tau = 0.03 * st["sigma"] ** 2
        sbtv.my_fista(yd, A, A.T, tau, 1.0, sbtv.TVnorm, sbtv.Psi_TV(25), 1, -1.0, 5, xd)
        t0 = time.perf_counter()
        out = sbtv.my_fista(yd, A, A.T, tau, 1.0, sbtv.TVnorm, sbtv.Psi_TV(25), 1, -1.0, a.iters, xd)
        dt = time.perf_counter() - t0
        print(f"config 3: FISTA + TV prox(25), 2048^2 Moffat: {(a.iters - 1) / dt:.1f} iterations/s "
              f"({1e3 * dt / (a.iters - 1):.3f} ms/iteration), objective {out[1][0]:.4e} -> {out[1][-1]:.4e}")
        return
    if a.config == 4:
        kind, size, nimg, share = "laplace", 1024, 8, False
    else:
        kind, size, nimg, share = "gaussian", 2048, 4, True
    x = image(size)
    st = sbtv.demo_setup(kind, x, rng.standard_normal(x.shape), evMax=0.99)
    samples, warmup = a.iters + 1, 0
    op, c = op_struct(kind, st, samples, warmup, 2)
    fn = sbtv.SAPG_algorithm_laplace if kind == "laplace" else sbtv.SAPG_algorithm_Guassian
    if share:
        op["chains"] = nimg
        y = sbtv.to_device(st["y"])
        kw = dict(share_gradients=True)
    else:
        y = sbtv.to_device(np.stack([st["y"]] * nimg))
        kw = {}
    op2 = dict(op, samples=3)
    fn(y, op2, c, **kw)                      # warm-up (workspaces, twiddles)
    t0 = time.perf_counter()
    out = fn(y, op, c, **kw)
    dt = time.perf_counter() - t0
    it = samples - 1
    print(f"config {a.config}: SAPG {kind}, {nimg} x {size}^2 {'chains on one image' if share else 'images'}: "
          f"{it / dt:.2f} SAPG iterations/s per GPU ({1e3 * dt / it:.2f} ms/iteration, "
          f"{1e3 * dt / it / nimg:.2f} ms per image-iteration)")


if __name__ == "__main__":
    main()
