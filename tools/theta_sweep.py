#!/usr/bin/env python3
"""Parameter sweep over independent SALSA solves: the legacy script's grid over the PSF
parameter `Tau_op` and the regularisation weight `Theta_op` (SALSA/salsa_m.m:234-325: one SALSA_v2 solve per grid
point on the same observation, MSE recorded per point).  Grid points are independent units: they are sharded over
the ranks (point i -> rank i mod world, `sbtv.dist.shard`), each rank solves its points as ONE batched call
(per-image tau / mu / PSF), and the per-point scalars are gathered at the end - no data-path collective.

  python tools/theta_sweep.py                                   # one GPU
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/theta_sweep.py
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd"))
import numpy as np
import sbtv
from sbtv import dist as sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--image", default=os.path.join(ROOT, "tests", "golden", "man_512.npy"))
    ap.add_argument("--thetas", default="0.005,0.01,0.02,0.03,0.05,0.08,0.12,0.2")
    ap.add_argument("--w1", default="0.4", help="comma-separated Gaussian w1 values to sweep (w2 = 0.3)")
    ap.add_argument("--backend", default=None)
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="rehearsal on a one-GPU box (use with --backend gloo)")
    a = ap.parse_args()
    rank, world = sd.init(a.backend)
    local = 0 if a.all_ranks_on_device0 else int(os.environ.get("LOCAL_RANK", "0"))
    ctx = sbtv.Context(local)
    x = np.load(a.image).astype(np.float64)
    st = sbtv.demo_setup("gaussian", x, np.random.default_rng(1).standard_normal(x.shape), evMax=1.0, ctx=ctx)
    grid = [(th, w1) for w1 in map(float, a.w1.split(",")) for th in map(float, a.thetas.split(","))]
    mine = sd.shard(len(grid))
    out = []
    if mine:
        B = len(mine)
        th = np.array([grid[i][0] for i in mine])
        taps = np.stack([sbtv.Gaussian_psf(7, grid[i][1], 0.3) for i in mine])       # per-image PSF
        A = sbtv.BlurOperator(taps, ctx=ctx)
        dev = f"cuda:{local}"
        yb = sbtv.to_device(np.stack([st["y"]] * B), dev)
        xb = sbtv.to_device(np.stack([x] * B), dev)
        mu = th / 10                                                                   # salsa_m.m: mu = theta/10
        res = sbtv.SALSA_v2(yb, A, th * st["sigma"] ** 2, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xb,
                            "StopCriterion", 1, "ToleranceA", 1e-5, "MAXITERA", 2000, "TVINITIALIZATION", 1,
                            "TViters", 10, "VERBOSE", 0, ctx=ctx)
        xs = sbtv.to_host(res[0]).reshape(B, *x.shape)
        objs = res[3] if B > 1 else [res[3]]
        for k in range(B):
            out.append(dict(mse_db=float(sbtv.MSE(x, xs[k], ctx=ctx)), psnr=float(sbtv.PSNR(x, xs[k], ctx=ctx)),
                            n_outer=len(objs[k]) - 1))
    merged = sd.merge_sharded(len(grid), sd.gather_objects(out))
    if rank == 0:
        print(f"{len(grid)} grid points on {world} rank(s), image {x.shape}")
        for (th, w1), r in zip(grid, merged):
            print(f"  theta {th:7.4f}  w1 {w1:5.2f}   MSE {r['mse_db']:7.3f} dB   PSNR {r['psnr']:7.3f} dB   "
                  f"outer iterations {r['n_outer']:4d}")
        best = int(np.argmax([r["psnr"] for r in merged]))
        print(f"best: theta = {grid[best][0]}, w1 = {grid[best][1]} (PSNR {merged[best]['psnr']:.3f} dB)")
    sd.barrier()


if __name__ == "__main__":
    main()
