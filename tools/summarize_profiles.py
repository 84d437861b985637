#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats / kernel trace / PMC passes) of `bench.py` into the small
files committed under profiles/.

  python tools/summarize_profiles.py <tag> <stats_dir> [<pmc_sq_dir> <pmc_fetch_dir> <pmc_write_dir>]

Writes profiles/<tag>_kernel_stats.csv (per-kernel calls / avg / share; fused Chambolle launches split
into real and empty (redo no-op) launches) and, when PMC passes are given, profiles/<tag>_pmc.json
plus profiles/r01_pmc_chambolle.json (HBM bytes per real launch of the dominant kernel, with the
gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md §HBM; bench.py reads it for `roofline.traffic`).
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not f:
        raise SystemExit(f"no {pat} under {d}")
    return f[0]


def short(name):
    n = name.replace("void ", "").replace("sbtv::", "")
    return n.split("(")[0]


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    rows = list(csv.DictReader(open(find(stats_dir, "*_kernel_trace.csv"))))
    per = collections.defaultdict(list)
    for r in rows:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = short(r["Kernel_Name"])
        if k.startswith("chambolle_fused_kernel"):
            k += " [real]" if dur > 15.0 else " [empty redo pass]"
        per[k].append(dur)
    tot = sum(sum(v) for v in per.values())
    out = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")
    with open(out, "w") as f:
        f.write("kernel,calls,avg_us,min_us,max_us,total_ms,share_pct\n")
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            f.write(f"\"{k}\",{len(v)},{sum(v) / len(v):.2f},{min(v):.2f},{max(v):.2f},{sum(v) / 1e3:.3f},"
                    f"{100 * sum(v) / tot:.1f}\n")
    print("wrote", out)
    if len(sys.argv) >= 6:
        res = {}
        for d, names in ((sys.argv[3], None), (sys.argv[4], ["FETCH_SIZE"]), (sys.argv[5], ["WRITE_SIZE"])):
            rr = list(csv.DictReader(open(find(d, "*_counter_collection.csv"))))
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for x in rr:
                agg[short(x["Kernel_Name"])][x["Counter_Name"]].append(float(x["Counter_Value"]))
            for k, cs in agg.items():
                for c, v in cs.items():
                    res.setdefault(k, {})[c] = {"n": len(v), "mean": sum(v) / len(v), "max": max(v)}
        with open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
        ck = [k for k in res if k.startswith("chambolle_fused_kernel")]
        if ck:
            c = res[ck[0]]
            # every third launch is an empty redo pass (moves no data): the max over launches is a real launch
            fetch_kb, write_kb = c["FETCH_SIZE"]["max"], c["WRITE_SIZE"]["max"]
            d = {"kernel": ck[0], "fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb,
                 "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
                 "note": "per real 5-iteration launch at 2048x2048; FETCH_SIZE doubled (gfx950 reports 1/2 of wide "
                         "coalesced reads, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are included in "
                         "these fabric counters, so this is an upper bound on true HBM traffic"}
            with open(os.path.join(ROOT, "profiles", "r01_pmc_chambolle.json"), "w") as f:
                json.dump(d, f, indent=1)
            print(d)


if __name__ == "__main__":
    main()
