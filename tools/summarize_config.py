#!/usr/bin/env python3
"""profiles/<tag>_config<N>.md from the rocprofv3 passes of tools/profile_config.sh (see there)."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8000.0   # GB/s (MI355X_MICROARCH.md)


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not f:
        raise SystemExit(f"no {pat} under {d}")
    return f[0]


def short(name):
    return name.replace("void ", "").replace("sbtv::", "").split("(")[0]


def main():
    tag, cfg, trace, fetch, write, log = sys.argv[1:7]
    line = None
    for l in open(log):
        if l.startswith("{"):
            line = json.loads(l)
    rows = list(csv.DictReader(open(find(trace, "*_kernel_trace.csv"))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the timed loop = the second half of the launches (the first half holds the warm-up call and the set-up)
    names = [short(r["Kernel_Name"]) for r in rows]
    half = rows[len(rows) // 2:]
    per = collections.defaultdict(list)
    for r in half:
        per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    span = (int(half[-1]["End_Timestamp"]) - int(half[0]["Start_Timestamp"])) / 1e3
    busy = sum(sum(v) for v in per.values())

    def counter(d, name):
        agg = collections.defaultdict(list)
        for x in csv.DictReader(open(find(d, "*_counter_collection.csv"))):
            if x["Counter_Name"] == name:
                agg[short(x["Kernel_Name"])].append(float(x["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in agg.items()}
    fe, wr = counter(fetch, "FETCH_SIZE"), counter(write, "WRITE_SIZE")
    # iterations in the half: launches of the kernel that runs once per iteration (the collector)
    once = [k for k in per if k.endswith("collect_kernel")]
    n_it = len(per[once[0]]) if once else 1
    # a batch of independent items runs as two lanes (two streams, one collector launch per lane and iteration): count
    # iterations of the CALL, not of a lane
    qcol = next((c for c in ("Stream_Id", "Queue_Id") if c in half[0]), None)
    lanes = len({r[qcol] for r in half if once and short(r["Kernel_Name"]) == once[0]}) if qcol else 1
    n_it = max(n_it // max(lanes, 1), 1)
    out = os.path.join(ROOT, "profiles", f"{tag}_config{cfg}.md")
    with open(out, "w") as f:
        f.write(f"# {tag}: rocprofv3 profile of `tools/bench_sapg.py --config {cfg}` on one MI355X\n\n")
        if line:
            f.write(f"Bench line of the traced run: {line['value']:.1f} {line['unit']} ({line['ms_per_iteration']:.4f} ms per iteration; "
                    f"{json.dumps(line['config'])}).\n\n")
        f.write("Kernel-trace pass (durations, second half of the launches = the timed loop) and two separate PMC passes\n"
                "(FETCH_SIZE, WRITE_SIZE; never combined with tracing).  Bytes = 2 x FETCH_SIZE + WRITE_SIZE per launch, mean over\n"
                "all launches of the kernel (gfx950 correction of MI355X_MICROARCH.md; fabric-side: Infinity-Cache hits are\n"
                "included).  Peak 8 000 GB/s.\n\n")
        f.write("| kernel | launches / iteration | avg µs | share of GPU time | MB / launch (PMC) | GB/s | of peak |\n|---|---|---|---|---|---|---|\n")
        tot_bytes = 0.0
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            avg = sum(v) / len(v)
            b = 2 * fe.get(k, 0.0) * 1024 + wr.get(k, 0.0) * 1024      # counters are in KiB
            tot_bytes += b * len(v)
            gbs = b / (avg * 1e-6) / 1e9 if avg > 0 else 0.0
            f.write(f"| `{k}` | {len(v) / n_it:.2f} | {avg:.2f} | {100 * sum(v) / busy:.1f} % | {b / 1e6:.1f} | {gbs:.0f} | {gbs / HBM_PEAK:.2f} |\n")
        it_us = span / n_it
        if lanes > 1:
            f.write(f"\n(The call's {lanes} lanes run on {lanes} streams: launches / iteration count both, durations overlap.)\n")
        f.write(f"\nWhole iteration ({n_it} iterations in the window): {it_us:.1f} µs from first launch to last end "
                f"({100 * busy / span:.0f} % of it inside kernels), {tot_bytes / n_it / 1e6:.0f} MB (PMC) = "
                f"**{tot_bytes / n_it / (it_us * 1e-6) / 1e9:.0f} GB/s = {tot_bytes / n_it / (it_us * 1e-6) / 1e9 / HBM_PEAK:.2f} of the HBM peak**.\n")
    print("wrote", out)
    # machine-readable twin for bench.py's `extra_configs[*].roofline`: stamped with the hash of the kernel sources the
    # passes were measured on (bench.py refuses it when the sources have changed since)
    sys.path.insert(0, ROOT)
    import bench
    cur = os.path.join(ROOT, "profiles", "configs_current.json")
    try:
        d = json.load(open(cur))
    except Exception:
        d = {}
    sha = bench.source_sha()
    if d.get("source_sha256") != sha:
        d = {"source_sha256": sha, "configs": {}}
    top = max(per.items(), key=lambda kv: sum(kv[1]))
    d["configs"][str(cfg)] = {
        "tag": tag, "iterations_in_window": n_it, "us_per_iteration": it_us, "hbm_bytes_per_iteration": tot_bytes / n_it,
        "kernel_time_share": busy / span, "launches_per_iteration": sum(len(v) for v in per.values()) / n_it,
        "dominant_kernel": top[0], "dominant_kernel_share": sum(top[1]) / busy, "dominant_kernel_avg_us": sum(top[1]) / len(top[1]),
        "dominant_kernel_hbm_bytes_per_launch": 2 * fe.get(top[0], 0.0) * 1024 + wr.get(top[0], 0.0) * 1024,
        "bench_line": line, "bytes_are": "2 x FETCH_SIZE + WRITE_SIZE per launch (separate PMC passes), summed over the launches of one iteration"}
    json.dump(d, open(cur, "w"), indent=1)
    print("updated", cur)


if __name__ == "__main__":
    main()
