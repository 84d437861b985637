#!/usr/bin/env python3
"""Soak of the lanes: `python tools/soak_lanes.py SECONDS` keeps solving random batches (1..6 images, 64 x 48 .. 1024^2, host and
device images, random MAXITERA / TViters) on two contexts - one with the lanes off - and requires bit-equal images and objective
traces every time.  Round 4, final tree: 9 359 paired calls in 240 s, no difference, no hang."""
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
