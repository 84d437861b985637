#!/usr/bin/env python3
"""Same-box A/B of the two row passes of the wave-granular sizes (SBTV_ROWS_SUB=1: four wave-local sub-transforms per row,
=0: the software-pipelined workgroup kernel) over every loop bench.py and tools/bench_admm.py time.  Prints one line per
variant and round."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for rnd in range(rounds):
    for v in ("1", "0"):
        lab = os.environ.get("AB_ROWS_LIB", os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd", "lib", "libsbtv_lab.so"))
        env = dict(os.environ, SBTV_ROWS_SUB=v, SBTV_LIBRARY=lab)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            print("ROWS_SUB=" + v, "bench failed:", r.stderr[-400:]); continue
        ex = {k: round(x["value"]) for k, x in d.get("extra_configs", {}).items() if isinstance(x, dict) and "value" in x}
        line = {"salsa2048": round(d["value"]), "512": round(d.get("extra_512", {}).get("value", 0)), **ex}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_admm.py")], env=env, capture_output=True, text=True)
        for l in r.stdout.splitlines():
            try:
                e = json.loads(l)
                line[e["metric"].split()[0] + e["metric"].split(",")[-1].strip()[:5]] = round(e["value"])
            except Exception:
                pass
        print("ROWS_SUB=" + v, json.dumps(line), flush=True)
