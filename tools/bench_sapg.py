#!/usr/bin/env python3
"""Secondary benchmark: SAPG (MYULA) iterations/s per GPU for the shapes of BASELINE configs 4 and 5.

  python tools/bench_sapg.py --config 4   # 8 independent 1024x1024 images, Laplace PSF (one GPU's share of 64)
  python tools/bench_sapg.py --config 5   # 4 MYULA chains on one 2048x2048 image, Gaussian PSF (share of 32)
  python tools/bench_sapg.py --config 3   # FISTA + TV prox, 2048x2048, Moffat PSF
  python tools/bench_sapg.py --config 6   # the demo's own SAPG loop: one chain, 512x512 wheel.png, Gaussian PSF fixed

Under torch.distributed.run the same script runs the whole configuration, one rank per GPU:
  config 4: 64 images sharded over the ranks (image i -> rank i mod world), no data-path collective;
  config 5: 32 chains on one image split over the ranks (disjoint Philox streams through chain_offset), ONE
            all-reduce of 5 doubles per SAPG iteration (sbtv.dist.make_allreduce_fn = RCCL with the nccl backend).
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_sapg.py --config 5
  (--backend gloo --all-ranks-on-device0 rehearses the multi-rank path on a one-GPU box)
`--gpus N` outside torch.distributed.run starts the N ranks itself (child `python -m torch.distributed.run`, before
this process imports torch).  Rank 0 prints prose on stderr and ONE JSON line on stdout:
  {"metric", "value" (image-iterations/s over all ranks), "unit", "n_gpus", "world_size_seen" (size of the process group
   the collective ran on), "backend", "ms_per_iteration" (max over ranks), "config": {...}, "scaling": "weak"}
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd"))


def _self_launch(n):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _n = [int(sys.argv[i + 1]) for i, v in enumerate(sys.argv[:-1]) if v == "--gpus"]
    if _n and _n[0] > 1:
        sys.exit(_self_launch(_n[0]))          # nothing here has imported torch or touched a GPU

import numpy as np
import sbtv


def image(size, seed=0, name="man_512.npy"):
    man = np.load(os.path.join(ROOT, "tests", "golden", name)).astype(np.float64)
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
    ap.add_argument("--total", type=int, default=0, help="images (config 4, default 64) / chains (config 5, default 32) "
                                                         "over all ranks; single process default: one GPU's share")
    ap.add_argument("--gpus", type=int, default=1, help="ranks to start when not already under torch.distributed.run")
    ap.add_argument("--backend", default=None)
    ap.add_argument("--host-reduce", action="store_true", help="config 5: host-callback all-reduce (round-1 path)")
    ap.add_argument("--all-ranks-on-device0", action="store_true")
    a = ap.parse_args()
    from sbtv import dist as sd
    rank, world = sd.init(a.backend)
    local = 0 if a.all_ranks_on_device0 else int(os.environ.get("LOCAL_RANK", "0"))
    ctx = sbtv.Context(local)
    dev = f"cuda:{local}"
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
              f"({1e3 * dt / (a.iters - 1):.3f} ms/iteration), objective {out[1][0]:.4e} -> {out[1][-1]:.4e}",
              file=sys.stderr)
        print(json.dumps({"metric": "FISTA + TV prox(25) iterations/s, 2048x2048 Moffat PSF (BASELINE configs[2])",
                          "value": (a.iters - 1) / dt, "unit": "FISTA iterations/s", "n_gpus": 1,
                          "ms_per_iteration": 1e3 * dt / (a.iters - 1), "dtype": "f64", "data": "synthetic",
                          "higher_is_better": True, "config": {"workload": "my_fista + cold Chambolle(25)",
                                                               "image": [2048, 2048]}}), flush=True)
        return
    if a.config == 4:
        kind, size, share = "laplace", 1024, False
        total = a.total or (64 if world > 1 else 8)
        nimg, first = len(sd.shard(total)), 0
    elif a.config == 6:
        # the demo's own shape (run_Gaussian_demo.m:199): ONE chain on the 512 x 512 wheel.png, PSF fixed (fix_w1 = fix_w2 = 1)
        kind, size, share, total, nimg, first = "gaussian", 512, False, 1, 1, 0
    else:
        kind, size, share = "gaussian", 2048, True
        total = a.total or (32 if world > 1 else 4)
        nimg, first = sd.split_chains(total)
    x = image(size, name="wheel_512.npy" if a.config == 6 else "man_512.npy")
    st = sbtv.demo_setup(kind, x, rng.standard_normal(x.shape), evMax=0.99, ctx=ctx)
    samples, warmup = a.iters + 1, 0
    op, c = op_struct(kind, st, samples, warmup, 2)
    fn = sbtv.SAPG_algorithm_laplace if kind == "laplace" else sbtv.SAPG_algorithm_Guassian
    if share:
        op["chains"], op["chain_offset"] = nimg, first
        y = sbtv.to_device(st["y"], dev)
        # in-stream RCCL all-reduce on the library's device buffer (the loop never waits for the host);
        # --host-reduce selects the round-1 host callback (D2H, all-reduce, H2D and a synchronisation per iteration)
        import torch.distributed as tdist0
        if world > 1 and tdist0.get_backend() != "nccl" and not a.all_ranks_on_device0:
            a.host_reduce = True         # gloo stages device tensors through the host anyway: use the host callback
        # (the one-GPU rehearsal --backend gloo --all-ranks-on-device0 keeps the in-stream device hook, so that the
        # production callback - ExternalStream + CUDA-array-interface view - is what the rehearsal exercises)
        if a.host_reduce:
            kw = dict(share_gradients=True, reduce_fn=sd.make_allreduce_fn(), ctx=ctx)
        else:
            kw = dict(share_gradients=True, reduce_dev_fn=sd.make_device_allreduce_fn(local), ctx=ctx)
    else:
        y = sbtv.to_device(np.stack([st["y"]] * nimg), dev)
        kw = dict(ctx=ctx)
    op2 = dict(op, samples=3)
    fn(y, op2, c, **kw)                      # warm-up (workspaces, twiddles)
    sd.barrier()
    t0 = time.perf_counter()
    out = fn(y, op, c, **kw)
    ctx.sync()
    sd.barrier()
    dt = time.perf_counter() - t0
    if world > 1:                                # the slowest rank defines the iteration time
        import torch
        import torch.distributed as tdist
        t = torch.tensor([dt], dtype=torch.float64, device=dev if tdist.get_backend() == "nccl" else "cpu")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())
    it = samples - 1
    thetas = [float(np.ravel(r["thetas"])[-1]) for r in out[-1]] if isinstance(out[-1], list) else [float(out[-1]["thetas"][-1])]
    allth = sd.gather_objects(thetas)
    if rank == 0:
        unit = "chains on one image" if share else "images"
        print(f"config {a.config}: SAPG {kind}, {total} x {size}^2 {unit} on {world} rank(s) ({nimg} per rank): "
              f"{it / dt:.2f} SAPG iterations/s ({1e3 * dt / it:.2f} ms/iteration, "
              f"{1e3 * dt / it / nimg:.2f} ms per local image-iteration), "
              f"{total * it / dt:.0f} image-iterations/s in total", file=sys.stderr)
        same = None
        if share:
            flat = [t for part in allth for t in part]
            same = (max(flat) - min(flat) == 0.0)
            print(f"  last theta of every chain equal across ranks: {same} ({flat[0]:.6g})", file=sys.stderr)
        import torch.distributed as tdist
        live = tdist.is_available() and tdist.is_initialized()
        print(json.dumps({
            "metric": ("SAPG chain-iterations/s, 32 MYULA chains with shared gradients on one 2048x2048 image "
                       "(BASELINE configs[4])" if share else
                       "SAPG iterations/s, one chain on 512x512 wheel.png, Gaussian PSF fixed (run_Gaussian_demo.m:199)"
                       if a.config == 6 else
                       "SAPG image-iterations/s, batch of 64 independent 1024x1024 images, Laplace PSF "
                       "(BASELINE configs[3])"),
            "value": total * it / dt, "unit": ("chain" if share else "image") + "-iterations/s", "n_gpus": world,
            "world_size_seen": tdist.get_world_size() if live else 1,
            "backend": tdist.get_backend() if live else None,
            "steps": it, "ms_per_iteration": 1e3 * dt / it, "ms_per_local_unit_iteration": 1e3 * dt / it / nimg,
            "higher_is_better": True, "scaling": "weak" if a.total == 0 and world == 1 else "strong",
            "dtype": "f64", "data": "synthetic",
            "collective": (("all-reduce of 6 doubles per iteration, " + ("host callback (sbtv.dist.make_allreduce_fn)"
                            if a.host_reduce else "in-stream on the device buffer (sbtv.dist.make_device_allreduce_fn)"))
                           if share and world > 1 else "none"),
            "chains_agree_across_ranks": same,
            "config": {"workload": f"SAPG {kind}, chambolleit=25, {total} x {size}^2 {unit}", "image": [size, size],
                       "units_total": total, "units_per_rank": nimg, "parallelism": f"{unit} x{world}"}}), flush=True)
    sd.barrier()


if __name__ == "__main__":
    main()
