#!/usr/bin/env python3
"""Expected outputs of the full-size cases of `large_cases.py`, produced by the ORACLE (oracle/sbtv_oracle.py) in the
build container -> `large_configs.npz` (plain arrays, loadable without pickle).  These are oracle outputs, not MATLAB
outputs (parity unpinned); they exist so that the GPU box can hold the HIP path against the oracle
at sizes where the oracle itself needs minutes (the whole script: ~15 min on 8 cores).

    python tests/golden/make_golden_large.py [case ...]     cases: salsa2048 salsa512 fista2048 sapg_l sapg_s sapg_l_ref sapg_s_ref
Cases not named on the command line keep their entries of the existing file.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import large_cases as lc  # noqa: E402

o = lc._oracle()


def crop_all(prefix, arr, out):
    for name, (si, sj) in lc.crops(*arr.shape).items():
        out[f"{prefix}.{name}"] = np.ascontiguousarray(arr[si, sj])


def salsa(tag, pr, out):
    st = o.demo_setup("gaussian", pr["x"], np.zeros_like(pr["x"]), evMax=1.0, true_params=pr["w"])
    st["y"] = pr["y"]                      # the observation bench.py builds (its own NumPy blur)
    res = o.salsa_from_estimates(st, pr["theta"], pr["w"], pr["sigma"] ** 2, tol=pr["tol"], outeriters=pr["maxiter"],
                                 TViters=pr["TViters"])
    out[f"{tag}.n_outer"] = np.array(res["n_outer"])
    for k in ("objective", "mses", "distance", "criterion"):
        out[f"{tag}.{k}"] = res[k]
    out[f"{tag}.psnr"] = np.array(o.PSNR(pr["x"], res["x"]))
    out[f"{tag}.numA"] = np.array(res["numA"])
    out[f"{tag}.numAt"] = np.array(res["numAt"])
    out[f"{tag}.sigma"] = np.array(pr["sigma"])
    for k in ("x", "u", "bu"):
        crop_all(f"{tag}.{k}", res[k], out)
    print(tag, "n_outer", res["n_outer"], "psnr", float(out[f"{tag}.psnr"]))


def fista(out):
    pr = lc.fista2048()
    model, p = pr["st"]["model"], pr["p"]
    Psi = lambda v, th: o.chambolle_prox_TV_stop(v, lam=th, maxiter=25)[0]
    res = o.my_fista(pr["y"], lambda v: model.A(v, *p), lambda v: model.AT(v, *p), pr["tau"], pr["L"], o.TVnorm, Psi,
                     1, 0.0, lc.FISTA_ITERS, pr["x"])
    out["fista2048.objective"] = res["objective"]
    out["fista2048.mses"] = res["mses"]
    out["fista2048.sigma"] = np.array(pr["sigma"])
    out["fista2048.psnr"] = np.array(o.PSNR(pr["x"], res["x"]))
    crop_all("fista2048.x", res["x"], out)
    print("fista2048", res["objective"])


def sapg_l(out, tag="sapg_l", S=None):
    S = S or lc.SAPG_L
    pr = lc.sapg_laplace_1024x8(S)
    for b, st in enumerate(pr["sts"]):
        it = iter(pr["noise"][:, b])
        r = o.SAPG_algorithm(st, samples=S["samples"], warmup=S["warmup"], burnIn=S["burnIn"], randn=lambda s: next(it),
                             c=S["c"])
        t = f"{tag}.{b}"
        out[f"{t}.thetas"], out[f"{t}.sigmas"], out[f"{t}.bs"] = r["thetas"], r["sigmas"], r["ps"][0]
        out[f"{t}.grads"], out[f"{t}.logPi"], out[f"{t}.gX"] = r["grads"], r["logPiTraceX"], r["gXTrace"]
        out[f"{t}.logPi_WU"] = r["logPiTrace_WU"]
        out[f"{t}.eb"] = np.array([r["theta_EB"], r["p_EB"][0], r["sigma_EB"]])
        crop_all(f"{t}.X", r["Xlast_sample"], out)
        print(t, r["thetas"], r["ps"][0])


def sapg_s(out, tag="sapg_s", S=None):
    S = S or lc.SAPG_S
    pr = lc.sapg_shared_2048x2(S)
    nz, step = pr["noise"], [0] * S["chains"]

    def randn(shape, k):
        z = nz[step[k], k]
        step[k] += 1
        return z
    r = o.SAPG_algorithm_shared(pr["st"], S["chains"], S["samples"], S["warmup"], S["burnIn"], randn,
                                p_init=S["p_init"], fix=(False, False), c=S["c"])
    out[f"{tag}.thetas"], out[f"{tag}.sigmas"], out[f"{tag}.ps"] = r["thetas"], r["sigmas"], r["ps"]
    out[f"{tag}.grads"], out[f"{tag}.logPi"], out[f"{tag}.gX"] = r["grads"], r["logPiTraceX"], r["gXTrace"]
    for k in range(S["chains"]):
        crop_all(f"{tag}.X{k}", r["Xlast_samples"][k], out)
    print(tag, r["thetas"], r["ps"], r["sigmas"])


if __name__ == "__main__":
    want = sys.argv[1:] or ["salsa2048", "salsa512", "fista2048", "sapg_l", "sapg_s", "sapg_l_ref", "sapg_s_ref"]
    out = {}
    if os.path.exists(lc.FIXTURE):
        with np.load(lc.FIXTURE) as old:
            out = {k: old[k] for k in old.files if k.split(".")[0] not in want}
    for case in want:
        t0 = time.time()
        if case == "salsa2048":
            salsa("salsa2048", lc.salsa2048(), out)
        elif case == "salsa512":
            salsa("salsa512", lc.salsa512(), out)
        elif case == "fista2048":
            fista(out)
        elif case == "sapg_l":
            sapg_l(out)
        elif case == "sapg_s":
            sapg_s(out)
        elif case == "sapg_l_ref":
            sapg_l(out, "sapg_l_ref", lc.SAPG_L_REF)
        elif case == "sapg_s_ref":
            sapg_s(out, "sapg_s_ref", lc.SAPG_S_REF)
        else:
            raise SystemExit(f"unknown case {case}")
        print(f"  [{case}: {time.time() - t0:.0f} s]", flush=True)
        np.savez_compressed(lc.FIXTURE, **out)
    print(lc.FIXTURE, os.path.getsize(lc.FIXTURE), "bytes,", len(out), "arrays")
