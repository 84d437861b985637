set -e
export TMPDIR=/tmp
R=$PWD
cd /tmp
python3 $R/tools/trace_small.py 512 2000
SBTV_GRAPH=1 python3 $R/tools/trace_small.py 512 2000
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace512 -- python3 $R/tools/trace_small.py 512 1000 > $R/gpurun_out/trace512.log 2>&1
cd $R
for f in $(find gpurun_out/trace512 -name "*hip_api_stats.csv" -o -name "*kernel_stats.csv"); do echo "== $f"; head -16 $f | cut -c1-140; done
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace512/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows)//2: len(rows)//2 + 400]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
print("kernels %d busy %.1f us span %.1f us -> GPU busy fraction %.2f; median gap %.2f us, max gap %.1f us" % (len(rows), busy/1e3, span/1e3, busy/span, sorted(gaps)[len(gaps)//2]/1e3, max(gaps)/1e3))
big = sorted(gaps)[-10:]
print("10 largest gaps (us):", [round(g/1e3,1) for g in big])
PY
rm -rf gpurun_out/trace512
