#!/usr/bin/env python3
"""Where does a 512^2 SALSA solve lose its time when it runs "slow"?  (round-2 review: bench.py's extra_512 samples were
bimodal, 4 800 ... 18 800 it/s for the same 400-step call.)

Runs the bench's 512^2 solve `--samples` times (400 steps each), with and without a host-side pause before the call,
and prints per sample: it/s by the host clock, the device-side duration of the call (HIP events; equal to the host's
when the GPU never starved), and `sbtv_last_host_stats` (waits that had to sleep, stream-query fallbacks, longest wait).
SBTV_TAG_SPIN_US (read once per process) selects the spin window of the tag wait: 0 = sleep at once (round 2's
behaviour), 150 = default, 1e9 = never sleep.

    python tools/slowmode_512.py [--samples 12] [--steps 400] [--size 512]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=12)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--heavy", action="store_true", help="first what bench.py does before its 512^2 block: a 2048^2 solve, "
                                                         "the per-pass timings, large host arrays made and dropped")
    a = ap.parse_args()
    import torch
    import bench
    import sbtv
    ctx = sbtv.Context(0)
    x, y, s, _ = bench.make_problem(seed=1, size=a.size)
    yd, xd = sbtv.to_device(y, "cuda:0"), sbtv.to_device(x, "cuda:0")
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, *bench.W_TRUE), ctx=ctx)
    mu, tau = bench.THETA / 10, bench.THETA * s ** 2

    def solve(n):
        return sbtv.SALSA_v2(yd, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xd, "StopCriterion", 1,
                             "ToleranceA", -1.0, "MAXITERA", n, "TVINITIALIZATION", 1, "TViters", 10, ctx=ctx)
    if a.heavy:
        import numpy as np
        xb, yb, sb, _ = bench.make_problem(seed=1, size=2048)
        ybd, xbd = sbtv.to_device(yb, "cuda:0"), sbtv.to_device(xb, "cuda:0")
        sbtv.SALSA_v2(ybd, A, bench.THETA * sb ** 2, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xbd, "StopCriterion", 1,
                      "ToleranceA", -1.0, "MAXITERA", 300, "TVINITIALIZATION", 1, "TViters", 10, ctx=ctx)
        bench.pass_block(ctx, 2048, ("cols_fwd", "rows_salsa", "cols_inv_post", "prox10_warm"), 50)
        big = sbtv.to_device(np.stack([yb] * 4), "cuda:0")
        del big, ybd, xbd, xb, yb
    solve(50)
    rows = []
    for i in range(a.samples):
        pause = (0.0, 0.3, 0.0, 1.0)[i % 4]
        if pause:
            time.sleep(pause)           # the bench does host work (NumPy set-up, PSNR) between its GPU phases
        torch.cuda.synchronize()
        thr0 = bench.cgroup_throttled_us()
        t0 = time.perf_counter()
        solve(a.steps)
        torch.cuda.synchronize()
        e = time.perf_counter() - t0
        hs = ctx.last_host_stats()
        tm = ctx.last_timing()
        rows.append(dict(pause_s=pause, it_per_s=round(a.steps / e), host_ms=round(1e3 * e, 2),
                         device_ms=round(tm["loop_ms"], 2), waits_slept=int(hs["waits_slept"]), sleeps=int(hs["sleeps"]),
                         stream_queries=int(hs["stream_queries"]), ready_at_once=int(hs["ready_at_once"]),
                         wait_max_us=round(1e6 * hs["wait_max_s"]), wait_ms=round(1e3 * hs["wait_s"], 2),
                         enqueue_ms=round(1e3 * hs["enqueue_s"], 2), enqueue_max_us=round(1e6 * hs["enqueue_max_s"]),
                         wait_max_outer=int(hs["wait_max_outer"]), ctxsw=[int(hs["nvcsw"]), int(hs["nivcsw"])],
                         faults=[int(hs["minflt"]), int(hs["majflt"])],
                         throttled_us=(bench.cgroup_throttled_us() or 0) - (thr0 or 0)))
    print(json.dumps({"switches": sbtv.switches(), "size": a.size, "steps": a.steps, "samples": rows}))


if __name__ == "__main__":
    main()
