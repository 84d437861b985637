#!/usr/bin/env python3
"""Per-kernel roofline table of one profiled build: python tools/roofline_table.py r02_v2 > profiles/r02_v2_roofline_table.md

Reads profiles/<tag>_kernel_stats.csv (rocprofv3 kernel trace of the 2048^2 SALSA loop), profiles/<tag>_pmc.json /
pmc_current.json (PMC passes of the same build) and profiles/<tag>_bench.json; model bytes as in DESIGN.md section 5."""
import csv
import json
import sys

tag = sys.argv[1]
P = 2048 * 2048
MODEL = [   # kernel prefix, what moves, bytes per pixel
    ("chambolle_fused_kernel<4, 8, 4, true>", "read g,px,py + write px,py once per 5-iteration launch (+ f on every 2nd): 44 B/px", 44),
    # round 3, last step: the optimistic launches of the loop sum their error over a subset of the pixels (template ESUB)
    ("chambolle_fused_kernel<4, 8, 4, true, false, true>", "read g,px,py + write px,py once per 5-iteration launch (+ f on every 2nd): 44 B/px", 44),
    ("cols_fwd_wave_kernel<10, 16>", "read u, bu; write S: 24 B/px", 24),
    ("rows_pipe_kernel<11, 4>", "read S, H, Y; write S: 32 B/px", 32),
    ("cols_inv_wave_kernel<10, 16, 3>", "read S, u, bu, true; write x, bu, g: 56 B/px", 56),
    # round 3: the bookkeeping pass without the x store (SALSA at 1024 / 2048, csrc/salsa.hip NOX)
    ("cols_inv_wave_kernel<10, 16, 35>", "read S, u, bu, true; write bu, g (x is not stored): 48 B/px", 48),
]
stats = {r[0]: r for r in csv.reader(open(f"profiles/{tag}_kernel_stats.csv"))}
pmc = json.load(open("profiles/pmc_current.json"))
assert pmc["tag"] == tag, (pmc["tag"], tag)
bench = json.load(open(f"profiles/{tag}_bench.json"))
print(f"# Per-kernel roofline table of build {tag} (2048 x 2048 SALSA_v2 iteration, in the loop)\n")
print(f"Times = rocprofv3 kernel-trace averages of `bench.py --steps 30 --warmup 5` (`{tag}_kernel_stats.csv`); model bytes = the")
print("byte model of the fused design (DESIGN.md §5); PMC bytes = 2·FETCH_SIZE + WRITE_SIZE of the same kernel")
print("(`pmc_current.json`; gfx950 correction of MI355X_MICROARCH.md; memory-side counters, Infinity-Cache hits included).")
print("Peak 8 TB/s HBM; fp64 vector peak 39.3 T lane-instructions/s.\n")
print("| kernel | what moves (model) | model MB | avg µs | model TB/s (frac of 8) | PMC MB | PMC TB/s (frac of 8) | PMC / model | VALU issue (frac of 39.3 T/s) | SQ_WAIT_ANY / wave-cycles |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, what, bpp in MODEL:
    key = next((k for k in stats if k.startswith("void sbtv::" + name + "(") or k.startswith(name + "(") or k == name), None) or \
        next((k for k in stats if k.startswith(name) and "empty" not in k and not k[len(name):].startswith(",")), None)
    if key is None or name not in pmc["kernels"]:
        continue                      # (this build does not launch that variant)
    us = float(stats[key][2])
    k = pmc["kernels"][name]
    mb, pb = bpp * P / 1e6, k["hbm_bytes_per_launch"] / 1e6
    valu = k["valu_insts_per_launch"] * 64 / (us * 1e-6) / 1e12
    print(f"| `{name}` | {what} | {mb:.0f} | {us:.1f} | {mb / us:.2f} ({mb / us / 8:.2f}) | {pb:.0f} | {pb / us:.2f} ({pb / us / 8:.2f}) | "
          f"{pb / mb:.2f} | {valu:.1f} T/s ({valu / 39.3216:.2f}) | {k['wait_any'] / k['wave_cycles']:.2f} |")
ms = bench["ms_per_step"]
tb, mbm = pmc["bytes_per_outer_iteration"] / 1e6, bench["step_roofline"]["model_bytes_per_step"] / 1e6
print(f"\nWhole outer iteration: {tb:.0f} MB (PMC) / {mbm:.0f} MB (model) in {ms:.4f} ms = {tb / ms / 1e3:.2f} / {mbm / ms / 1e3:.2f} TB/s = "
      f"{tb / ms / 8e3:.2f} / {mbm / ms / 8e3:.2f} of the HBM peak; {bench['value']:.0f} outer iterations/s on the profiled box.")
