#!/usr/bin/env python3
"""Kernel tuning harness for the FFT passes and the TV prox (run on an MI355X): times every pass of the hot path with
`sbtv_diag_time_pass` under the environment hooks that select kernel variants, one child process per variant (the
hooks are read once per process), and prints one table.  The committed tables under profiles/ come from here.

  python tools/fft_lab.py [--size 2048] [--batch 1 4] [--reps 50] [--check]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys
sys.path.insert(0, os.path.join(%(root)r, "semi-blind-image-deblurring-problems-with-tv_amd"))
import numpy as np
import sbtv
ctx = sbtv.default_context(0)
out = {}
for size in %(sizes)r:
    for batch in %(batches)r:
        for name in %(passes)r:
            r = ctx.time_pass(name, size, size, batch, %(reps)d)
            out["%%s/%%d/b%%d" %% (name, size, batch)] = [r["ms"] * 1e3 / batch, r["gbs"]]
if %(check)r:
    rng = np.random.default_rng(0)
    for size in %(sizes)r:
        x = rng.uniform(0, 255, (size, size))
        A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
        h = np.zeros_like(x); h[:7, :7] = sbtv.Gaussian_psf(7, 0.4, 0.3)
        H = np.fft.fft2(h)
        for nm, got, want in (("A", A(x), np.real(np.fft.ifft2(H * np.fft.fft2(x)))),
                              ("AT", A.T(x), np.real(np.fft.ifft2(np.conj(H) * np.fft.fft2(x)))),
                              ("LS", A.LS(0.01)(x), np.real(np.fft.ifft2(np.fft.fft2(x) / (np.abs(H) ** 2 + 0.01))))):
            out["err_%%s/%%d" %% (nm, size)] = [float(np.max(np.abs(got - want)) / np.max(np.abs(want))), 0.0]
print("LABJSON" + json.dumps(out))
"""

VARIANTS = {
    "legacy (workgroup kernels)": {"SBTV_FFT_WAVE": "0"},
    "wave, rows V=16": {"SBTV_FFT_WAVE": "1", "SBTV_ROWS_V": "16"},
    "wave, rows V=8": {"SBTV_FFT_WAVE": "1", "SBTV_ROWS_V": "8"},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs="+", default=[2048])
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--passes", nargs="+", default=["cols_fwd", "rows_salsa", "cols_inv_post", "cols_inv", "rows_fwd",
                                                    "rows_grad", "rows_gradf", "prox10_warm", "prox25_cold"])
    ap.add_argument("--variants", nargs="+", default=None)
    ap.add_argument("--env", nargs="+", default=[], help='extra variants as "name:K=V,K2=V2" (e.g. '
                                                         '"tile 6x8:SBTV_FUSED_VARIANT=6,8,4" - commas inside a value are '
                                                         'kept when the next token has no =)')
    a = ap.parse_args()
    for spec in a.env:
        name, _, kv = spec.partition(":")
        d, last = {}, None
        for tok in kv.split(","):
            if "=" in tok:
                last, _, val = tok.partition("=")
                d[last] = val
            elif last:
                d[last] += "," + tok
        VARIANTS[name] = d
    if a.variants is None:
        a.variants = list(VARIANTS) if not a.env else [s.partition(":")[0] for s in a.env]
    res = {}
    for name in a.variants:
        env = dict(os.environ)
        for k in ("SBTV_FFT_WAVE", "SBTV_ROWS_V", "SBTV_FUSED_VARIANT", "SBTV_INLINE_CTRL"):
            env.pop(k, None)
        env.update(VARIANTS[name])
        code = CHILD % dict(root=ROOT, sizes=a.size, batches=a.batch, passes=a.passes, reps=a.reps, check=a.check)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("LABJSON")]
        if r.returncode != 0 or not line:
            print(f"variant {name!r} failed:\n{r.stderr[-3000:]}", file=sys.stderr)
            continue
        res[name] = json.loads(line[0][7:])
    keys = sorted({k for v in res.values() for k in v})
    print("| pass/size/batch (us per image; GB/s) | " + " | ".join(res) + " |")
    print("|---|" + "---|" * len(res))
    for k in keys:
        cells = []
        for name in res:
            v = res[name].get(k)
            cells.append("-" if v is None else (f"{v[0]:.2e}" if k.startswith("err_") else f"{v[0]:.1f} ({v[1]:.0f})"))
        print(f"| {k} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main()
