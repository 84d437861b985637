#!/bin/bash
# PMC passes for the fused Chambolle kernel alone (run on the GPU box from the repo root):
#   bash tools/pmc_fused.sh   -> gpurun_out/pmc_fused/summary.txt
set -eo pipefail
R=$PWD; O=$R/gpurun_out/pmc_fused; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
B="$R/tools/bench_prox.py --reps 6"
pass() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$O/$n" -- python3 $B > "$O/$n.log" 2>&1; echo "pass $n done"; }
pass a SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU
pass b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pass c SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM
pass d SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
cd "$R"
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
out = open(O + "/summary.txt", "w")
for n in "abcd":
    f = glob.glob(f"{O}/{n}/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "chambolle_fused_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(agg.items()):
        out.write(f"{c:28s} max {max(v):.5g}  n {len(v)}\n")
out.close()
print(open(O + "/summary.txt").read())
PY
rm -rf "$O"/[abcd]
