#!/usr/bin/env python3
"""Where the fixed cost of one 20-step `SALSA_v2` call goes (the driver's bench run times ONE such call): Python mirror,
C-ABI call, device loop (HIP events inside the library), final synchronisation.  Medians over --calls calls."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd"))
import bench  # sets the thread-pool environment first
import numpy as np, torch, sbtv
import ctypes as C
from sbtv import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--calls", type=int, default=40)
a = ap.parse_args()
ctx = sbtv.Context(0)
x, y, sigma, noise = bench.make_problem(1)
yd, xd = sbtv.to_device(y, "cuda:0"), sbtv.to_device(x, "cuda:0")
A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, *bench.W_TRUE), ctx=ctx)
mu, tau = bench.THETA / 10, bench.THETA * sigma ** 2
def solve(K):
    return sbtv.SALSA_v2(yd, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xd, "StopCriterion", 1, "ToleranceA", -1.0,
                         "MAXITERA", K, "TVINITIALIZATION", 1, "TViters", 10, ctx=ctx)
solve(300)
# raw C-ABI call with everything prepared
so = L.sbtv_salsa_opts(); ctx.lib.sbtv_salsa_opts_default(C.byref(so))
so.stopcriterion, so.maxiter, so.TViters, so.tolA, so.compute_mse = 1, a.steps, 10, -1.0, 1
yi, ti = L.Images(yd), L.Images(xd)
xo = L.empty_like_images(yi)
K = a.steps
obj, dist, tim, mse = np.zeros((1, K + 1)), np.zeros((1, K)), np.zeros((1, K + 1)), np.zeros((1, K + 1))
nA, nAt, nout = (C.c_int * 1)(), (C.c_int * 1)(), (C.c_int * 1)()
taps = A._cm(1); tau_a, tau_p = L.dvec(tau, 1); mu_a, mu_p = L.dvec(mu, 1)
def raw():
    rc = ctx.lib.sbtv_SALSA_v2(ctx.h, yi.ptr, yi.M, yi.N, 1, L.vptr(taps), A.taille, tau_p, mu_p, C.byref(so), ti.ptr, None, xo.ptr,
                               L.vptr(obj), L.vptr(dist), L.vptr(tim), L.vptr(mse), nA, nAt, nout, yi.flags)
    assert rc == 0
res = {"mirror_total": [], "raw_total": [], "loop_dev": [], "sync_after": []}
for i in range(a.calls):
    torch.cuda.synchronize(); ctx.sync()
    t0 = time.perf_counter(); solve(a.steps); t1 = time.perf_counter(); torch.cuda.synchronize(); ctx.sync(); t2 = time.perf_counter()
    res["mirror_total"].append(t2 - t0); res["sync_after"].append(t2 - t1)
    res["loop_dev"].append(ctx.last_timing()["loop_ms"] * 1e-3)
    t0 = time.perf_counter(); raw(); torch.cuda.synchronize(); ctx.sync(); t1 = time.perf_counter()
    res["raw_total"].append(t1 - t0)
med = {k: 1e3 * float(np.median(v)) for k, v in res.items()}
p10 = {k: 1e3 * float(np.percentile(v, 10)) for k, v in res.items()}
p90 = {k: 1e3 * float(np.percentile(v, 90)) for k, v in res.items()}
print("ms (median / p10 / p90) of one %d-step call:" % a.steps)
for k in res: print("  %-14s %.3f / %.3f / %.3f" % (k, med[k], p10[k], p90[k]))
print("  fixed cost: mirror %.3f ms, raw C-ABI %.3f ms over the device loop" % (med["mirror_total"] - med["loop_dev"], med["raw_total"] - med["loop_dev"]))
hs = ctx.last_host_stats() if hasattr(ctx, "last_host_stats") else None
print("  host stats of the last call:", hs)
