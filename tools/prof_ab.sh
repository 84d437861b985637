#!/bin/bash
# In-situ kernel times of the bench loop under two settings of an environment hook (run on the GPU box):
#   bash tools/prof_ab.sh SBTV_FFT_WAVE 0 1    -> profiles/r02_ab_SBTV_FFT_WAVE_{0,1}_kernel_stats.csv
set -eo pipefail
VAR=${1:?env var}; shift
export TMPDIR=/tmp
R=$PWD
for val in "$@"; do
  O=$R/gpurun_out/prof_ab_$val; rm -rf "$O"
  (export "$VAR=$val"; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-batched --no-extras > "$O.log" 2>&1)
  python3 tools/summarize_profiles.py "r02_ab_${VAR}_${val}" "$O"
  rm -rf "$O"
  echo "== $VAR=$val"; cut -d, -f1-3 "profiles/r02_ab_${VAR}_${val}_kernel_stats.csv" | head -12
done
