#!/bin/bash
# LDS / wait counters of the FFT kernels in the 2048^2 loop (run on the GPU box): bash tools/pmc_lds.sh
set -eo pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_lds; rm -rf "$O"
B="$R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-batched --no-extras"
(cd /tmp; rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O" -- python3 $B > "$O.log" 2>&1)
python3 - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if not any(t in k for t in ("rows_pipe", "cols_inv_wave", "cols_fwd_wave", "chambolle_fused_kernel")): continue
    agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    m = {c: max(v) for c, v in cs.items()}
    print(k)
    print("   " + "  ".join("%s=%.3g" % (c, m[c]) for c in sorted(m)))
    if m.get("SQ_LDS_IDX_ACTIVE"): print("   bank conflict cycles / LDS active cycles = %.2f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]))
PY
rm -rf "$O"
