#!/bin/bash
# per-launch kernel durations / gaps of the last SAPG iterations at a small size (run on the GPU box): bash tools/trace_sapg.sh [size]
set -eo pipefail
export TMPDIR=/tmp
R=$PWD
SIZE=${1:-512}
O=$R/gpurun_out/trace_sapg; rm -rf "$O"
cat > /tmp/sapg_child.py <<PY
import sys, os
sys.path.insert(0, "$R/semi-blind-image-deblurring-problems-with-tv_amd")
import numpy as np, sbtv
size = $SIZE
man = np.load("$R/tests/golden/man_512.npy").astype(np.float64)
r = max(1, size // 512)
x = np.tile(man, (r, r))[:size, :size]
ctx = sbtv.Context(0)
st = sbtv.demo_setup("gaussian", x, np.random.default_rng(1).standard_normal(x.shape), evMax=0.99, ctx=ctx)
op = dict(samples=40, warmup=0, burnIn=2, psf_size=7, phi=0.0, gamma=st["gamma"], th_init=0.01, min_th=1e-3, max_th=1.0,
          sigma=st["sigma"], sigma_init=st["sigma_init"], sigma_min=st["sigma_min"], sigma_max=st["sigma_max"], d_scale=1.0,
          d_exp=0.8, fix_sigma=0, seed=7, w1=0.4, w2=0.3, w1_init=0.5, w2_init=0.3, min_w1=0.1, min_w2=0.1, max_w1=1.0,
          max_w2=1.0, fix_w1=int(os.environ.get("FIXED", "1")), fix_w2=int(os.environ.get("FIXED", "1")))
op["lambda"] = st["lambda"]
c = dict(theta=0.01, w1=10.0, w2=10.0, sigma=1000.0, lam=1.0, gam=1.0)
y = sbtv.to_device(st["y"], "cuda:0")
sbtv.SAPG_algorithm_Guassian(y, op, c, ctx=ctx)
PY
(cd /tmp; rocprofv3 --kernel-trace --output-format csv -d "$O" -- python3 /tmp/sapg_child.py > "$O.log" 2>&1)
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows[-34:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-44s dur %7.2f  gap %6.2f" % (r["Kernel_Name"][:44], (e - s) / 1e3, gap))
    prev_end = e
PY
rm -rf "$O"
