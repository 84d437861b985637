#!/bin/bash
# per-launch kernel durations and the gaps between consecutive launches of the last SALSA outer iterations (run on the
# GPU box): bash tools/trace_seq.sh [size]        size = 2048 (default, bench.py) or a small size (tools/trace_small.py)
# TRACE_STEPS (12) = outer iterations of the traced 2048^2 call, TRACE_TAIL (45) = launches printed (a short call with a
# long tail shows the set-up of a call as well)
set -eo pipefail
export TMPDIR=/tmp
R=$PWD
SIZE=${1:-2048}
O=$R/gpurun_out/trace_seq; rm -rf "$O"
if [ "$SIZE" = "2048" ]; then
  (cd /tmp; rocprofv3 --kernel-trace --output-format csv -d "$O" -- python3 $R/bench.py --steps ${TRACE_STEPS:-12} --warmup 3 --no-cpu-baseline --no-batched --no-extras > "$O.log" 2>&1)
else
  (cd /tmp; rocprofv3 --kernel-trace --output-format csv -d "$O" -- python3 $R/tools/trace_small.py $SIZE 40 > "$O.log" 2>&1)
fi
python3 - "$O" "${TRACE_TAIL:-45}" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append((r["Kernel_Name"][:42], (s - t0) / 1e3, (e - s) / 1e3))
prev_end = None
for name, s, d in out[-int(sys.argv[2]):]:
    gap = (s - prev_end) if prev_end is not None else 0.0
    print("%-42s start %10.1f us  dur %7.2f  gap %6.2f" % (name, s, d, gap))
    prev_end = s + d
PY
rm -rf "$O"
