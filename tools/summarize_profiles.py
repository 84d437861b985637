#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats / kernel trace / PMC passes) of `bench.py` into the small
files committed under profiles/.

  python tools/summarize_profiles.py <tag> <stats_dir> [<pmc_sq_dir> <pmc_fetch_dir> <pmc_write_dir>]

Writes profiles/<tag>_kernel_stats.csv (per-kernel calls / avg / share; fused Chambolle launches split
into real and empty (redo no-op) launches) and, when PMC passes are given, profiles/<tag>_pmc.json
plus profiles/pmc_current.json: what bench.py needs for its roofline entries - per kernel of the SALSA loop the
VALU wave-instructions and the HBM-side bytes of one real launch (2 x FETCH_SIZE + WRITE_SIZE: the gfx950
correction of MI355X_MICROARCH.md §HBM), the bytes of one outer iteration, and the SHA-256 of the kernel sources
the numbers were measured on (bench.py refuses the file when the sources have changed since).
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
        sys.path.insert(0, ROOT)
        import bench
        loop = ("chambolle_fused_kernel", "chambolle_pipe_kernel", "chambolle_fused_ctrl_kernel", "fft_cols_fwd_kernel",
                "cols_fwd_wave_kernel", "fft_rows_kernel", "rows_wave_kernel", "rows_pipe_kernel", "fft_cols_inv_kernel",
                "cols_inv_wave_kernel", "salsa_collect_kernel")
        # outer iterations = launches of the inverse column pass with the SALSA bookkeeping (the collector has no launch of
        # its own any more: it rides on the next iteration's first Chambolle launch)
        post = [len(v) for k, v in per.items() if k.startswith(("cols_inv_wave_kernel", "fft_cols_inv_kernel"))]
        n_outer = max(post) if post else max(len(per.get(k, [])) for k in per if k.startswith("salsa_collect_kernel"))
        kernels, total = {}, 0.0
        for k, c in res.items():
            if not k.startswith(loop) or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                continue
            # launches that do nothing (the empty redo pass of the prox) move no data: the max is a real launch,
            # mean x n the sum over all launches
            f_kb, w_kb = c["FETCH_SIZE"], c["WRITE_SIZE"]
            real = 2 * f_kb["max"] * 1024 + w_kb["max"] * 1024
            total += (2 * f_kb["mean"] * f_kb["n"] + w_kb["mean"] * w_kb["n"]) * 1024
            kernels[k] = {"hbm_bytes_per_launch": real, "fetch_size_kb_raw": f_kb["max"], "write_size_kb": w_kb["max"],
                          "valu_insts_per_launch": c.get("SQ_INSTS_VALU", {}).get("max"),
                          "active_valu_quadcycles": c.get("SQ_ACTIVE_INST_VALU", {}).get("max"),
                          "wave_cycles": c.get("SQ_WAVE_CYCLES", {}).get("max"), "wait_any": c.get("SQ_WAIT_ANY", {}).get("max"),
                          "launches_profiled": f_kb["n"]}
        ck = [k for k in kernels if k.startswith("chambolle_fused_kernel")]
        if ck:
            kernels[ck[0]]["core_fraction"] = (116.0 * 21.0) / (128.0 * 32.0)   # core / region of the 4x8 tile geometry
        d = {"tag": tag, "source_sha256": bench.source_sha(), "image": [bench.SIZE, bench.SIZE],
             "outer_iterations_profiled": n_outer, "bytes_per_outer_iteration": total / max(n_outer, 1),
             "kernels": kernels,
             "note": "rocprofv3 --pmc passes of `bench.py --steps 30 --warmup 5` (separate SQ / FETCH_SIZE / WRITE_SIZE "
                     "passes); bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 reports 1/2 of wide coalesced reads, "
                     "MI355X_MICROARCH.md HBM section); fabric-side counters: Infinity-Cache hits are included"}
        with open(os.path.join(ROOT, "profiles", "pmc_current.json"), "w") as f:
            json.dump(d, f, indent=1, sort_keys=True)
        print(json.dumps({k: v for k, v in d.items() if k != "kernels"}))


if __name__ == "__main__":
    main()
