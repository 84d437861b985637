#!/bin/bash
# Reproduce the numbers committed under profiles/ (run on the GPU box from the repo root):
#   bash tools/profile_round.sh r01_v5
# 1. rocprofv3 --kernel-trace --stats      -> profiles/<tag>_kernel_stats.csv
# 2. three separate --pmc passes (SQ activity / FETCH_SIZE / WRITE_SIZE; never combined with tracing)
#                                          -> profiles/<tag>_pmc.json, profiles/pmc_current.json
# 3. plain bench line (reads pmc_current)  -> profiles/<tag>_bench.json
set -eo pipefail
TAG=${1:?tag}
R=$PWD
O=$R/gpurun_out/prof_$TAG
rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
B="$R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-batched --no-extras"   # 2048^2 single-image launches only
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 $B > "$O/stats.log" 2>&1
echo "kernel trace done"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
          --output-format csv -d "$O/pmc_sq" -- python3 $B > "$O/pmc_sq.log" 2>&1
echo "pmc sq done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 $B > "$O/pmc_fetch.log" 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 $B > "$O/pmc_write.log" 2>&1
echo "pmc write done"
cd "$R"
python3 tools/summarize_profiles.py "$TAG" "$O/stats" "$O/pmc_sq" "$O/pmc_fetch" "$O/pmc_write"
# the bench line last: it then finds profiles/pmc_current.json of THIS source revision and carries roofline.traffic
python3 bench.py --steps 500 --warmup 20 > "$O/bench.json" 2> "$O/bench.err"
cp "$O/bench.json" "profiles/${TAG}_bench.json"
echo "bench done"; tail -c 600 "$O/bench.json"; echo
mkdir -p gpurun_out/profiles_$TAG
cp profiles/${TAG}_* profiles/pmc_current.json gpurun_out/profiles_$TAG/
# the raw traces are large: keep only the summaries in gpurun_out
rm -rf "$O/stats" "$O/pmc_sq" "$O/pmc_fetch" "$O/pmc_write"
