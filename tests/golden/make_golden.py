#!/usr/bin/env python3
"""Regenerates the fixtures of this directory from the reference checkout (run in the build container, where
/root/reference exists; the GPU box only ever sees the committed .npy files).

The reference ships no numeric test vectors; what it does ship are its test IMAGES.  They are data, stored here
losslessly as uint8 arrays (imread of the 8-bit greyscale PNGs, run_Gaussian_demo.m:117):
  man_512.npy   <- images/man.png   (512 x 512)   config[1] of BASELINE.json, tiled 4 x 4 for the 2048^2 bench
  cman_256.npy  <- image/cman.png   (256 x 256)   config[0]
  wheel_512.npy <- images/wheel.png (512 x 512)   the image the three demos load (entry 8 of the directory listing,
                                                  run_Gaussian_demo.m:100,117; run_moffat_demo.m:108,116; run_laplace_demo.m:83,90)
"""
import os
import sys

import numpy as np
from PIL import Image

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
for out, src in (("man_512.npy", "images/man.png"), ("cman_256.npy", "image/cman.png"),
                 ("wheel_512.npy", "images/wheel.png")):
    a = np.asarray(Image.open(os.path.join(REF, src)).convert("L"), dtype=np.uint8)
    np.save(os.path.join(HERE, out), a)
    print(out, a.shape, a.dtype, int(a.min()), int(a.max()))

# ---- oracle known answers (regression anchors) -----------------------------------------------------------
# Scalars and a small crop produced by oracle/sbtv_oracle.py on the committed images.  They pin the oracle to
# itself across rounds (a change of the restatement shows up as a diff of this file) and give the GPU tests a
# committed vector to hit besides the live oracle.  They are NOT MATLAB outputs (parity unpinned).
import json

sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import sbtv_oracle as o  # noqa: E402

cman = np.load(os.path.join(HERE, "cman_256.npy")).astype(np.float64)
f, px, py, k, err = o.chambolle_prox_TV_stop(cman, lam=10.0, maxiter=10, return_info=True)
st = o.demo_setup("gaussian", cman, np.random.default_rng(1).standard_normal(cman.shape), evMax=1.0, BSNR=30.0,
                  true_params=(1 / 1.6, 1 / 1.6))
res = o.salsa_from_estimates(st, 0.03, st["p_true"], st["sigma"] ** 2)
kat = dict(
    chambolle_cman256_lambda10_K10=dict(k=int(k), err=float(err), sum_f=float(f.sum()), sum_px=float(px.sum()),
                                        sum_py=float(py.sum()), f_crop_8x8=f[100:108, 100:108].tolist()),
    tvnorm_cman256=float(o.TVnorm(cman)),
    salsa_cman256_gaussian_sigma1p6=dict(sigma=float(st["sigma"]), n_outer=int(res["n_outer"]),
                                          objective_first=float(res["objective"][0]),
                                          objective_last=float(res["objective"][-1]),
                                          psnr_db=float(o.PSNR(cman, res["x"])), mse_last=float(res["mses"][-1]),
                                          x_crop_4x4=res["x"][60:64, 60:64].tolist()),
)
with open(os.path.join(HERE, "oracle_kat.json"), "w") as fjson:
    json.dump(kat, fjson, indent=1)
print("oracle_kat.json:", {k2: (v if not isinstance(v, dict) else "...") for k2, v in kat.items()})
