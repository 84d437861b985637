#!/bin/bash
# In-situ kernel times of the bench loop under several environment settings, same box (run on the GPU box):
#   bash tools/prof_env.sh "base:SBTV_U_TILED=0 SBTV_ROWS_PIPE=0" "pipe:SBTV_ROWS_PIPE=1"
# -> gpurun_out/prof_env_<label>_kernel_stats.csv and the top kernels on stdout
set -eo pipefail
export TMPDIR=/tmp
R=$PWD
for spec in "$@"; do
  label=${spec%%:*}; setting=${spec#*:}
  O=$R/gpurun_out/prof_env_$label; rm -rf "$O"
  (export $setting; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-batched --no-extras > "$O.log" 2>&1)
  python3 tools/summarize_profiles.py "tmp_env_$label" "$O" > /dev/null
  mv "profiles/tmp_env_${label}_kernel_stats.csv" "gpurun_out/prof_env_${label}_kernel_stats.csv"
  rm -rf "$O"
  echo "== $label ($setting)"
  python3 - "gpurun_out/prof_env_${label}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
tot = 0.0
for r in rows[1:8]:
    print("  %-44s calls %4s avg %7s min %7s" % (r[0][:44], r[1], r[2], r[3]))
PY
done
