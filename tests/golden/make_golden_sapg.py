#!/usr/bin/env python3
"""Expected outputs of the long-chain SAPG cases of `sapg_cases.py`, produced by the ORACLE (oracle/sbtv_oracle.py) in
the build container -> `sapg_long.npz` (plain arrays, loadable without pickle).  Oracle outputs, not MATLAB outputs
(parity unpinned).

    python tests/golden/make_golden_sapg.py [trace] [trace_fs] [stat]

trace.<kind>.*   one chain of 320 samples (60 warm-up) per PSF family at 64^2 with injected noise and the reference's
                 own step scales: every trace, the EB estimates, the running means / tolerances the reference logs
                 (SAPG_algorithm_Guassian.m:217-284) and the last sample.  ~2 s per family.
                 trace.<kind>.sens / .horizon: the oracle's OWN sensitivity.  With these step scales the sigma^2
                 recursion s <- s + c delta (R / 2s^2 - n / 2s) overshoots and jumps (a sawtooth between the bounds
                 whose map has |slope| >> 1), so the parameter iteration amplifies rounding differences by ~2x per
                 sample while it lasts.  `sens` = per-sample relative difference (max over theta, p, sigma^2) between
                 this chain and the SAME oracle on the observation scaled by (1 + 1e-12); `horizon` = the first sample
                 where it exceeds 1e-10.  Per-step parity at rtol 1e-9 is a meaningful statement only before the
                 horizon; beyond it no two floating-point implementations agree.
trace_fs.<kind>.* the same chains with op.fix_sigma = 1 (sigma^2 held at its true value, SAPG_algorithm_Guassian.m:189-194):
                 without the sigma^2 sawtooth the iteration is stable (sens <= ~1e-10 over all 320 samples), so the WHOLE
                 trace is comparable at rtol 1e-9 - with the PSF projections bouncing between their bounds (Laplace: every
                 sample), engaging and releasing (Moffat) or both (Gaussian).
stat.<kind>.eb   EB estimates [theta, p..., sigma^2] of 16 independent oracle chains (NumPy noise) of 1600 samples
                 (300 warm-up, mean over 1280..1600) on ONE observation per family.  ~8 s per chain; the chains run
                 in a process pool.
"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import sapg_cases as sc  # noqa: E402

o = sc._oracle()
o.set_workers(1)             # 64^2 transforms: threads only cost


def running_mean_and_tol(trace, burnIn):
    """mean_x(ii - burnIn) = mean(x(burnIn:ii)) for ii > burnIn; tol_x(ii) = |mean(x(burnIn:ii)) - mean(x(burnIn:ii-1))| /
    mean(x(burnIn:ii-1)) (SAPG_algorithm_Guassian.m:217-244, 1-based, inclusive ranges), as plain loops."""
    n = len(trace)
    means, tol = [], np.zeros(n)
    for ii in range(2, n + 1):                       # 1-based iteration number
        if ii > burnIn:
            means.append(np.mean(trace[burnIn - 1:ii]))
        cur = np.mean(trace[burnIn - 1:ii]) if ii >= burnIn else np.nan
        prev = np.mean(trace[burnIn - 1:ii - 1]) if ii - 1 >= burnIn else np.nan
        tol[ii - 1] = abs(cur - prev) / prev
    return np.array(means), tol


def trace_case(kind, out, tag="trace", fix_sigma=False):
    st = sc.setup(kind)
    T = sc.TRACE
    it = iter(sc.trace_noise(kind))
    fr = sc.FREE[kind]
    r = o.SAPG_algorithm(st, samples=T["samples"], warmup=T["warmup"], burnIn=T["burnIn"], randn=lambda s: next(it),
                         fix=fr["fix"], p_init=fr["p_init"], fix_sigma=fix_sigma)
    t = f"{tag}.{kind}"
    it2 = iter(sc.trace_noise(kind))
    r2 = o.SAPG_algorithm(dict(st, y=st["y"] * (1.0 + 1e-12)), samples=T["samples"], warmup=T["warmup"], burnIn=T["burnIn"],
                          randn=lambda s: next(it2), fix=fr["fix"], p_init=fr["p_init"], fix_sigma=fix_sigma)
    rel = lambda a, b: np.abs(a - b) / np.abs(a)
    sens = np.maximum(np.maximum(rel(r["thetas"], r2["thetas"]), rel(r["sigmas"], r2["sigmas"])),
                      np.max(rel(r["ps"], r2["ps"]), axis=0))
    over = np.nonzero(sens > 1e-10)[0]
    out[f"{t}.sens"] = sens
    out[f"{t}.horizon"] = np.array(int(over[0]) if len(over) else T["samples"])
    print(t, "horizon", int(out[f"{t}.horizon"]), "sens at 50/100/200/319:", sens[[50, 100, 200, 319]])
    out[f"{t}.thetas"], out[f"{t}.sigmas"], out[f"{t}.ps"] = r["thetas"], r["sigmas"], r["ps"]
    out[f"{t}.grads"], out[f"{t}.logPi"], out[f"{t}.gX"] = r["grads"], r["logPiTraceX"], r["gXTrace"]
    out[f"{t}.logPi_WU"] = r["logPiTrace_WU"]
    out[f"{t}.eb"] = np.array([r["theta_EB"]] + list(r["p_EB"]) + [r["sigma_EB"]])
    out[f"{t}.err_psf"] = r["err_psf"]
    out[f"{t}.X"] = r["Xlast_sample"]
    m, tl = running_mean_and_tol(r["thetas"], T["burnIn"])
    out[f"{t}.mean_thetas"], out[f"{t}.tol_thetas"] = m, tl
    m, tl = running_mean_and_tol(r["sigmas"], T["burnIn"])
    out[f"{t}.mean_sigmas"], out[f"{t}.tol_sigma"] = m, tl
    d = o.DEMO[kind]
    lo, hi = min(st["sigma_min"], st["sigma_max"]), max(st["sigma_min"], st["sigma_max"])
    print(t, "theta_EB %.5f" % r["theta_EB"], "p_EB", np.round(r["p_EB"], 4), "sigma_EB %.4f" % r["sigma_EB"],
          "| samples on a bound: theta", int(np.sum((r["thetas"] == 1e-3) | (r["thetas"] == 1.0))),
          "p", [int(np.sum((r["ps"][q] == d["pmin"][q]) | (r["ps"][q] == d["pmax"][q]))) for q in range(len(d["true"]))],
          "sigma", int(np.sum((r["sigmas"] == lo) | (r["sigmas"] == hi))))


def _stat_chain(args):
    kind, chain = args
    o.set_workers(1)
    st = sc.setup(kind)
    S = sc.STAT
    rng = sc.stat_rng(kind, chain)
    fr = sc.FREE[kind]
    r = o.SAPG_algorithm(st, samples=S["samples"], warmup=S["warmup"], burnIn=S["burnIn"],
                         randn=lambda s: rng.standard_normal(s), fix=fr["fix"], p_init=fr["p_init"])
    return kind, chain, np.array([r["theta_EB"]] + list(r["p_EB"]) + [r["sigma_EB"]])


def stat_cases(out):
    jobs = [(kind, c) for kind in sc.KINDS for c in range(sc.STAT["chains"])]
    res = {}
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for kind, chain, eb in ex.map(_stat_chain, jobs):
            res.setdefault(kind, {})[chain] = eb
    for kind in sc.KINDS:
        eb = np.stack([res[kind][c] for c in range(sc.STAT["chains"])])
        out[f"stat.{kind}.eb"] = eb
        print(f"stat.{kind}: mean", eb.mean(0), "sd", eb.std(0, ddof=1))


if __name__ == "__main__":
    want = sys.argv[1:] or ["trace", "trace_fs", "stat"]
    out = {}
    if os.path.exists(sc.FIXTURE):
        with np.load(sc.FIXTURE) as old:
            out = {k: old[k] for k in old.files if k.split(".")[0] not in want}
    t0 = time.time()
    if "trace" in want:
        for kind in sc.KINDS:
            trace_case(kind, out)
    if "trace_fs" in want:
        for kind in sc.KINDS:
            trace_case(kind, out, "trace_fs", fix_sigma=True)
    if "stat" in want:
        stat_cases(out)
    np.savez_compressed(sc.FIXTURE, **out)
    print(sc.FIXTURE, os.path.getsize(sc.FIXTURE), "bytes,", len(out), "arrays, %.0f s" % (time.time() - t0))
