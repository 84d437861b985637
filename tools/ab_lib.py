#!/usr/bin/env python3
"""A/B of library builds on ONE box: for every library given (default: the in-tree libsbtv.so) a child process
  1. holds the TV prox against the oracle (512 x 512, K = 10 cold with f, then K = 10 warm on a changed g): max relative
     deviation of px / f and of err - a variant that is faster but wrong must show here;
  2. times single fused launches of 1..5 iterations and the two prox shapes at 2048^2 (sbtv_diag_time_pass, HIP events);
  3. runs the bench's 2048^2 SALSA solve for --steps iterations: outer iterations/s and us per Chambolle iteration.
The variants are alternated --reps times.  Prints a markdown table (stdout) and one JSON line per run (stderr).

    python tools/ab_lib.py --reps 2 default lib/libsbtv_diet0.so default@SBTV_TILE_ORDER=0 ...
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd")

CHILD = r'''
import json, os, sys, time
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1"); os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, %(root)r); sys.path.insert(0, %(pkg)r); sys.path.insert(0, os.path.join(%(root)r, "oracle")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, sbtv, bench
import sbtv_oracle as o
from conftest import synth_image
steps = int(sys.argv[1])
ctx = sbtv.Context(0)
res = {"lib": os.path.basename(sbtv.LIB_PATH), "switches": sbtv.switches()}
# 1. parity
g = synth_image(512, 512, 5) + np.random.default_rng(0).standard_normal((512, 512))
f, px, py, k, err = sbtv.chambolle_prox_TV_stop(g, "lambda", 9.0, "maxiter", 10, return_info=True, ctx=ctx)
rf, rpx, rpy, rk, rerr = o.chambolle_prox_TV_stop(g, lam=9.0, maxiter=10, return_info=True)
g2 = g + 0.5 * np.random.default_rng(1).standard_normal(g.shape)
f2, px2, py2, k2, err2 = sbtv.chambolle_prox_TV_stop(g2, "lambda", 9.0, "maxiter", 10, "dualvars", np.hstack([px, py]), return_info=True, ctx=ctx)
wf2, wpx2, wpy2, wk2, werr2 = o.chambolle_prox_TV_stop(g2, lam=9.0, maxiter=10, dualvars=np.hstack([rpx, rpy]), return_info=True)
res["parity"] = {"px_max_abs": float(max(np.max(np.abs(px - rpx)), np.max(np.abs(px2 - wpx2)))),
                 "f_max_abs": float(max(np.max(np.abs(f - rf)), np.max(np.abs(f2 - wf2)))),
                 "err_rel": float(max(abs(float(np.ravel(err)[0]) - rerr) / rerr, abs(float(np.ravel(err2)[0]) - werr2) / werr2)),
                 "k": [int(np.ravel(k)[0]), int(np.ravel(k2)[0])]}
# 2. passes
res["passes_us"] = {}
for nm in ("fused_steps1", "fused_steps3", "fused_steps5", "prox10_warm", "prox25_cold"):
    try:
        res["passes_us"][nm] = 1e3 * ctx.time_pass(nm, 2048, 2048, 1, 40)["ms"]
    except Exception as e:
        res["passes_us"][nm] = None
# 3. the loop
x, y, sigma, _ = bench.make_problem(1, 2048)
yd, xd = sbtv.to_device(y), sbtv.to_device(x)
A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, *bench.W_TRUE), ctx=ctx)
mu, tau = bench.THETA / 10, bench.THETA * sigma ** 2
def solve(n):
    return sbtv.SALSA_v2(yd, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xd, "ToleranceA", -1.0, "MAXITERA", n,
                         "TVINITIALIZATION", 1, "TViters", 10, ctx=ctx)
# the converged solve of the bench problem against its committed oracle fixture (stopping iteration, PSNR)
xg, _, _, obj, _, _, _ = sbtv.SALSA_v2(yd, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xd, "ToleranceA", 1e-5, "MAXITERA", 500,
                                      "TVINITIALIZATION", 1, "TViters", 10, ctx=ctx)
res["salsa_fixture"] = bench.fixture_check("salsa2048", bench.psnr(x, sbtv.to_host(xg)), len(obj) - 1)["matches"]
solve(300)
vals = []
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); solve(steps); torch.cuda.synchronize()
    vals.append(steps / (time.perf_counter() - t0))
tm = ctx.last_timing()
res["salsa_it_per_s"] = sorted(vals)[1]
res["us_per_chambolle_iteration"] = 1e3 * tm["chambolle_ms"] / max(tm["chambolle_launches"], 1)
print(json.dumps(res))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*", default=["default"])
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    rows = []
    for rep in range(a.reps):
        for lib in a.libs:
            env = dict(os.environ)
            parts = lib.split("@")                    # "lib/libsbtv_x.so@SBTV_TILE_ORDER=0": library + environment settings
            for kv in parts[1:]:
                k, v = kv.split("=", 1)
                env[k] = v
            if parts[0] != "default":
                env["SBTV_LIBRARY"] = parts[0] if os.path.isabs(parts[0]) else os.path.join(PKG, parts[0])
            r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "pkg": PKG}, str(a.steps)], env=env,
                               capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                print(f"{lib}: FAILED\n{r.stderr[-1500:]}", file=sys.stderr)
                continue
            d = json.loads(r.stdout.strip().splitlines()[-1])
            d["variant"] = lib
            print(json.dumps(d), file=sys.stderr, flush=True)
            rows.append(d)
    print("(every run: converged 2048² solve vs its oracle fixture: " + ", ".join(str(d.get("salsa_fixture")) for d in rows) + ")\n")
    print("| library | SALSA it/s @2048² | µs / Chambolle it (loop) | launch of 1 / 3 / 5 its (µs) | prox10 warm / prox25 cold (µs) | "
          "parity vs oracle: max abs px, f; rel err |")
    print("|---|---|---|---|---|---|")
    for d in rows:
        p, q = d["passes_us"], d["parity"]
        fmt = lambda v: "—" if v is None else f"{v:.1f}"
        print(f"| {d['variant']} | {d['salsa_it_per_s']:.0f} | {d['us_per_chambolle_iteration']:.2f} | "
              f"{fmt(p['fused_steps1'])} / {fmt(p['fused_steps3'])} / {fmt(p['fused_steps5'])} | "
              f"{fmt(p['prox10_warm'])} / {fmt(p['prox25_cold'])} | {q['px_max_abs']:.1e}, {q['f_max_abs']:.1e}; {q['err_rel']:.1e} |")


if __name__ == "__main__":
    main()
