#!/usr/bin/env python3
"""run_Gaussian_demo.m (and the Moffat / Laplace twins) end to end on the MI355X through the host mirror:
observation model -> evMax -> SAPG (MYULA) estimates of theta, PSF parameters, sigma^2 -> SALSA_v2 MAP image.

  python tools/run_gaussian_demo.py [--kind gaussian|moffat|laplace] [--samples 20000 --warmup 15000]
                                    [--image tests/golden/wheel_512.npy]

The default image is the one the three demos load: images/wheel.png (entry 8 of the directory listing,
run_Gaussian_demo.m:100,117; run_moffat_demo.m:108,116; run_laplace_demo.m:83,90).

Constants follow run_Gaussian_demo.m:34-85 / run_moffat_demo.m:34-73 / run_laplace_demo.m:34-62
(SURVEY.md §9.1).  MATLAB's randn('state',1) stream cannot be reproduced: noise comes from NumPy
(observation) and the device Philox generator (MYULA).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd"))
import numpy as np
import sbtv

DEMO = {
    "gaussian": dict(names=("w1", "w2"), true=(0.4, 0.3), init=(0.5, 0.3), pmin=(0.1, 0.1), pmax=(1.0, 1.0),
                     fix=(1, 1), ev=(1.0, 1.0), c=dict(theta=0.01, w1=10.0, w2=10.0, sigma=1000.0)),
    "moffat": dict(names=("alpha", "beta"), true=(0.4, 3.5), init=(1.0, 10.0), pmin=(1e-2, 0.1), pmax=(1.0, 10.0),
                   fix=(0, 0), ev=(1.0, 5.0), c=dict(theta=0.1, alpha=10.0, beta=1e4, sigma=1e4)),
    "laplace": dict(names=("b",), true=(0.3,), init=(0.1,), pmin=(1e-3,), pmax=(1.0,), fix=(0,), ev=(1.0,),
                    c=dict(theta=0.01, b=100.0, sigma=1e4)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="gaussian", choices=list(DEMO))
    ap.add_argument("--image", default=os.path.join(ROOT, "tests", "golden", "wheel_512.npy"))
    ap.add_argument("--samples", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=15000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="", help="directory for the results .npz, the trace figure (SVG) and the images (PGM)")
    a = ap.parse_args()
    d = DEMO[a.kind]
    x = np.load(a.image).astype(np.float64)
    rng = np.random.default_rng(a.seed)
    ctx = sbtv.default_context(0)

    # evMax: power iteration at the demo's parameters (run_Gaussian_demo.m:142)
    A_ev = sbtv.BlurOperator(sbtv.psf_family(a.kind, 7, d["ev"])[0])
    evMax = sbtv.max_eigenval(A_ev, A_ev.T, d["ev"], x.shape, 1e-4, 1e4, x0=rng.standard_normal(x.shape))
    st = sbtv.demo_setup(a.kind, x, rng.standard_normal(x.shape), evMax=evMax)      # :145-184
    print(f"{a.kind}: image {x.shape}, evMax {evMax:.5f}, sigma {st['sigma']:.4f}, lambda {st['lambda']:.4g}, "
          f"gamma {st['gamma']:.4g}")

    op = dict(samples=a.samples, warmup=a.warmup, burnIn=int(a.samples * 80 / 100), psf_size=7, phi=0.0,
              gamma=st["gamma"], th_init=0.01, min_th=1e-3, max_th=1.0, sigma=st["sigma"],
              sigma_init=st["sigma_init"], sigma_min=st["sigma_min"], sigma_max=st["sigma_max"],
              d_scale=0.01 / 0.01, d_exp=0.8, fix_sigma=0, seed=a.seed)
    op["lambda"] = st["lambda"]
    for q, nm in enumerate(d["names"]):
        op[nm] = d["true"][q]
        op[nm + "_init"] = d["true"][q] if d["fix"][q] else d["init"][q]
        op["min_" + nm], op["max_" + nm], op["fix_" + nm] = d["pmin"][q], d["pmax"][q], d["fix"][q]
    c = dict(d["c"], lam=1.0, gam=1.0)
    fn = {"gaussian": sbtv.SAPG_algorithm_Guassian, "moffat": sbtv.SAPG_algorithm_moffat,
          "laplace": sbtv.SAPG_algorithm_laplace}[a.kind]
    yd = sbtv.to_device(st["y"])
    t0 = time.perf_counter()
    out = fn(yd, op, c)
    t_sapg = time.perf_counter() - t0
    theta_EB, sigma_EB, res = out[0], out[-2], out[-1]
    p_EB = [res[nm + "_EB"] for nm in d["names"]]
    print(f"SAPG: {a.warmup} warm-up + {a.samples} iterations in {t_sapg:.2f} s "
          f"({1e3 * t_sapg / (a.warmup + a.samples):.3f} ms/iteration)")
    print(f"  theta_EB {theta_EB:.5f}   " + "   ".join(f"{n}_EB {v:.4f} (true {t})" for n, v, t in
                                                       zip(d["names"], p_EB, d["true"])) +
          f"   sigma2_EB {sigma_EB:.4f} (true {st['sigma'] ** 2:.4f})")

    # MAP estimate with the EB parameters (run_Gaussian_demo.m:210-242)
    A = sbtv.BlurOperator(sbtv.psf_family(a.kind, 7, p_EB)[0])
    mu = theta_EB / 10
    ctx.reset_calls()
    t0 = time.perf_counter()
    xMAP, numA, numAt, obj, dist, times, mses = sbtv.SALSA_v2(
        yd, A, theta_EB * sigma_EB, "MU", mu, "AT", A.T, "StopCriterion", 1, "True_x", sbtv.to_device(x),
        "ToleranceA", 1e-5, "MAXITERA", 500, "TVINITIALIZATION", 1, "TViters", 10, "LS", A.LS(mu), "VERBOSE", 0)
    t_salsa = time.perf_counter() - t0
    xm = sbtv.to_host(xMAP)
    print(f"SALSA_v2: {len(obj) - 1} outer iterations in {1e3 * t_salsa:.1f} ms, calls = {ctx.calls}, "
          f"mse = {sbtv.MSE(x, xm):.3f} dB, PSNR = {sbtv.PSNR(x, xm):.3f} dB "
          f"(observation: {sbtv.PSNR(x, st['y']):.3f} dB), ssim = {sbtv.ssim(x, xm):.4f}")
    if a.out:
        # results file + the demo's figures (run_Gaussian_demo.m:247-301): parameter traces, x, y, xMAP
        os.makedirs(a.out, exist_ok=True)
        sbtv.save_results(os.path.join(a.out, f"{a.kind}_results.npz"), res, xMAP=xm, objective=obj, mses=mses,
                          psnr=sbtv.PSNR(x, xm), ssim=sbtv.ssim(x, xm))
        true = {nm + "s": t for nm, t in zip(d["names"], d["true"])}
        true["sigmas"] = st["sigma"] ** 2
        sbtv.plot_traces(os.path.join(a.out, f"{a.kind}_traces.svg"), res, true_values=true)
        for nm, img in (("x", x), ("y", st["y"]), ("xMAP", xm)):
            sbtv.save_image(os.path.join(a.out, f"{a.kind}_{nm}.pgm"), img, 0.0, 255.0)
        print("wrote results, traces and images to", a.out)


if __name__ == "__main__":
    main()
