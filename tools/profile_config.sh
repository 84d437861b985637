#!/bin/bash
# rocprofv3 evidence for the secondary configurations (BASELINE configs[2..4]) on the GPU box:
#   bash tools/profile_config.sh <tag> <config 3|4|5>
# kernel trace + separate FETCH_SIZE / WRITE_SIZE passes of `tools/bench_sapg.py --config N`, condensed by
# tools/summarize_config.py into profiles/<tag>_config<N>.md (per kernel: launches per iteration, average duration,
# HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE, GB/s; whole iteration: bytes, GB/s, fraction of the 8 TB/s peak)
set -eo pipefail
TAG=${1:?tag}; CFG=${2:?config}
R=$PWD
O=$R/gpurun_out/prof_${TAG}_c$CFG
rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
B="$R/tools/bench_sapg.py --config $CFG --iters 40"
rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -- python3 $B > "$O/trace.log" 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 $B > "$O/pmc_fetch.log" 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 $B > "$O/pmc_write.log" 2>&1
echo "pmc write done"
cd "$R"
python3 tools/summarize_config.py "$TAG" "$CFG" "$O/trace" "$O/pmc_fetch" "$O/pmc_write" "$O/trace.log"
mkdir -p gpurun_out/profiles_$TAG
cp profiles/${TAG}_config${CFG}.md gpurun_out/profiles_$TAG/
rm -rf "$O/trace" "$O/pmc_fetch" "$O/pmc_write"
