#!/usr/bin/env python3
"""Soak of the lanes: `python tools/soak_lanes.py SECONDS` keeps solving random batches (1..6 images, 64 x 48 .. 1024^2, host and
device images, random MAXITERA / TViters) on two contexts - one with the lanes off - and requires bit-equal images and objective
traces every time; then as long again on SAPG: independent chains (bit-equal) and shared-gradient chains split over the lanes
with the in-stream exchange (1e-9), the three PSF families, random sizes / chain counts / lengths.
Round 4, final tree: 9 359 paired SALSA calls in 240 s; 5 888 SALSA + 9 308 SAPG paired calls in 2 x 150 s: no difference, no hang."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("semi-blind-image-deblurring-problems-with-tv_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np, sbtv
from conftest import synth_image
from test_gpu_group import _salsa_problem
rng = np.random.default_rng(2024)
A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
c1, c2 = sbtv.Context(0), sbtv.Context(0)
c1.set_lanes(1)
t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[1]):
    nb = int(rng.integers(1, 7)); M, N = [(64, 48), (128, 128), (256, 256), (512, 512), (96, 80), (1024, 1024)][int(rng.integers(0, 6))]
    if M == 1024: nb = min(nb, 3)
    xs, ys, taus = _salsa_problem(nb, M, N)
    args = ("MU", 0.003, "AT", A.T, "LS", A.LS(0.003), "True_x", xs, "ToleranceA", 1e-4, "MAXITERA", int(rng.integers(5, 120)),
            "TVINITIALIZATION", 1, "TViters", int(rng.integers(2, 11)))
    dev = bool(rng.integers(0, 2))
    yy, xx = (sbtv.to_device(ys), sbtv.to_device(xs)) if dev else (ys, xs)
    a2 = list(args); a2[7] = xx
    one = sbtv.SALSA_v2(yy, A, taus, *a2, ctx=c1); two = sbtv.SALSA_v2(yy, A, taus, *a2, ctx=c2)
    x1, x2 = (sbtv.to_host(one[0]), sbtv.to_host(two[0])) if dev else (one[0], two[0])
    assert np.array_equal(x1, x2), (n, nb, M, N, dev)
    ob1 = one[3] if nb > 1 else [one[3]]; ob2 = two[3] if nb > 1 else [two[3]]
    assert all(np.array_equal(a, b) for a, b in zip(ob1, ob2))
    n += 1
    if n % 25 == 0: print(n, "calls ok, %.0f s" % (time.time() - t0), flush=True)
print("soak ok:", n, "paired SALSA calls bit-equal (lanes vs one stream, host and device images, sizes 64x48 .. 1024^2)")

# ---- second half of the budget: SAPG - independent chains (bit-equal) and shared-gradient chains split over the lanes with
# the in-stream exchange (equal to the rounding of the sum order), random sizes / chain counts / lengths
import sbtv_oracle as o
from test_gpu_group import _sapg_op
c2.set_lanes(2)
t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[1]):
    M = N = int(rng.choice([32, 64, 128, 256]))
    kind = ["gaussian", "laplace", "moffat"][int(rng.integers(0, 3))]
    nch = int(rng.integers(2, 7))
    S, W = int(rng.integers(4, 40)), int(rng.integers(0, 6))
    shared = bool(rng.integers(0, 2))
    st = o.demo_setup(kind, synth_image(M, N, int(rng.integers(0, 100))), rng.standard_normal((M, N)), evMax=0.99)
    op, c, names = _sapg_op(kind, st, S, W, max(1, S // 2))
    op["seed"] = int(rng.integers(1, 1000))
    for nm in names:
        op["fix_" + nm] = 0
    c = dict(c, sigma=100.0, **{nm: 0.3 for nm in names})
    fn = {"gaussian": sbtv.SAPG_algorithm_Guassian, "laplace": sbtv.SAPG_algorithm_laplace, "moffat": sbtv.SAPG_algorithm_moffat}[kind]
    if shared:
        op["chains"] = nch
        a = fn(st["y"], op, c, share_gradients=True, ctx=c1)[-1]
        b = fn(st["y"], op, c, share_gradients=True, ctx=c2)[-1]
    else:
        y = np.stack([st["y"] * (1 - 0.01 * k) for k in range(nch)])
        a = fn(y, op, c, ctx=c1)[-1]
        b = fn(y, op, c, ctx=c2)[-1]
    for k in range(nch):
        for key in ("thetas", "sigmas", "logPiTraceX") + tuple(nm + "s" for nm in names):
            if shared:
                assert np.allclose(a[k][key], b[k][key], rtol=1e-9, atol=0), (n, kind, M, nch, S, W, key)
            else:
                assert np.array_equal(a[k][key], b[k][key]), (n, kind, M, nch, S, W, key)
    n += 1
    if n % 25 == 0: print(n, "SAPG calls ok, %.0f s" % (time.time() - t0), flush=True)
print("soak ok:", n, "paired SAPG calls (independent chains bit-equal, shared chains split over the lanes to 1e-9)")
