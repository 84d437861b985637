#!/bin/bash
# Quick look at one build: kernel-trace averages + an LDS / wait counter pass of the 2048^2 single-image loop
#   bash tools/prof_quick.sh TAG      -> gpurun_out/quick_TAG/{kernels.txt,lds.txt}
set -eo pipefail
TAG=${1:?tag}
R=$PWD
O=$R/gpurun_out/quick_$TAG
rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
B="$R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-batched --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 $B > "$O/stats.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY \
          --output-format csv -d "$O/pmc" -- python3 $B > "$O/pmc.log" 2>&1 || true
cd "$R"
python3 - "$O" <<'PY'
import csv, glob, sys, collections, re
O = sys.argv[1]
f = glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True)
with open(O + "/kernels.txt", "w") as out:
    for p in f:
        for r in csv.DictReader(open(p)):
            out.write("%-70s calls %6s avg %10.2f us  %5s%%\n" % (re.sub(r"\(.*", "", r["Name"])[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for p in glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(O + "/lds.txt", "w") as out:
    for k, d in agg.items():
        out.write(k + " " + " ".join("%s=%.3g" % kv for kv in sorted(d.items())) + "\n")
PY
rm -rf "$O/stats" "$O/pmc"
cat "$O/kernels.txt" | head -12; cat "$O/lds.txt" | grep -i "rows\|cols" | head
