#!/usr/bin/env python3
"""Phase timeline of the fused Chambolle kernel inside the SALSA loop (2048 x 2048), from the debug build
`make -C .../csrc timeline` (lib/libsbtv_timeline.so: thread 0 of every workgroup records the 100 MHz clock at entry, after
its region has arrived, after its last iteration and after its stores were issued, and the CU it ran on).
Reads the records of the LAST launch of a solve (the f-writing second launch of the last outer iteration) and prints a
markdown summary: how a workgroup's life divides into load / iterate / store, and how the two workgroups resident on a
CU overlap those phases.        python3 tools/chambolle_timeline.py > profiles/<tag>_chambolle_timeline.md"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd")
os.environ.setdefault("SBTV_LIBRARY", os.path.join(PKG, "lib", "libsbtv_timeline.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
import numpy as np, torch, sbtv, bench

ctx = sbtv.default_context(0)
x, y, sigma, noise = bench.make_problem(1, 2048)
yd, xd = sbtv.to_device(y), sbtv.to_device(x)
A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, *bench.W_TRUE), ctx=ctx)
mu, tau = bench.THETA / 10, bench.THETA * sigma ** 2
def solve(k):
    return sbtv.SALSA_v2(yd, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xd, "ToleranceA", -1.0, "MAXITERA", k,
                         "TVINITIALIZATION", 1, "TViters", 10, ctx=ctx)
solve(300)
solve(40); torch.cuda.synchronize()
ntile = int(os.environ.get("NREC", 18 * 98))
rec = np.zeros((ntile, 8), dtype=np.uint64)
ctx.lib.sbtv_debug_timeline.restype = C.c_int
ctx.lib.sbtv_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
rc = ctx.lib.sbtv_debug_timeline(ctx.h, rec.ctypes.data_as(C.c_void_p), ntile)
assert rc == 0, rc
t = rec[:, :4].astype(np.int64)
t -= t[:, 0].min()
us = t / 100.0                                           # 100 MHz ticks -> microseconds
load, comp, store = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
life = us[:, 3] - us[:, 0]
span = us[:, 3].max()
hw, xcc = rec[:, 4].astype(np.int64), rec[:, 5].astype(np.int64) & 0xF
cu_key = (xcc << 8) | (((hw >> 13) & 0x7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
keys = np.unique(cu_key)
def q(a): return "%.1f / %.1f / %.1f" % (np.percentile(a, 10), np.median(a), np.percentile(a, 90))
print("# Fused Chambolle kernel: where a workgroup's time goes (2048², in the SALSA loop, one launch of 1 764 workgroups)\n")
print("`tools/chambolle_timeline.py` on the debug build (`make timeline`): thread 0 of every workgroup stamps the 100 MHz clock at")
print("entry, when its region (g, px, py) has arrived, after its 5 iterations, and when its stores are issued.  Last launch of a")
print("40-iteration solve (the second, f-writing launch of an outer iteration).\n")
print("| quantity | value |\n|---|---|")
print(f"| launch, first entry to last exit | {span:.1f} µs |")
print(f"| CUs seen / workgroups per CU | {len(keys)} / {ntile / len(keys):.2f} |")
print(f"| workgroup life (10 % / median / 90 %) | {q(life)} µs |")
print(f"| … waiting for its region | {q(load)} µs ({100 * load.sum() / life.sum():.0f} % of all workgroup time) |")
print(f"| … 5 iterations | {q(comp)} µs ({100 * comp.sum() / life.sum():.0f} %) |")
print(f"| … issuing its stores (f, px, py) | {q(store)} µs ({100 * store.sum() / life.sum():.0f} %) |")
# per CU: time with 0 / 1 / 2 workgroups in their iteration phase
grid = np.arange(0.0, span, 0.05)
busy = np.zeros((3,))
resident = np.zeros((3,))
for k in keys:
    m = cu_key == k
    c = np.zeros_like(grid); r = np.zeros_like(grid)
    for a, b_, lo, hi in zip(us[m, 1], us[m, 2], us[m, 0], us[m, 3]):
        c += (grid >= a) & (grid < b_)
        r += (grid >= lo) & (grid < hi)
    for n in range(3):
        busy[n] += np.mean(np.minimum(c, 2) == n)
        resident[n] += np.mean(np.minimum(r, 2) == n)
busy /= len(keys); resident /= len(keys)
print(f"| share of the launch a CU has 2 / 1 / 0 workgroups resident | {100 * resident[2]:.0f} % / {100 * resident[1]:.0f} % / {100 * resident[0]:.0f} % |")
print(f"| share of the launch a CU has 2 / 1 / 0 workgroups ITERATING | {100 * busy[2]:.0f} % / {100 * busy[1]:.0f} % / {100 * busy[0]:.0f} % |")
starts = np.sort(us[:, 0])
print(f"| entry times: first 512 workgroups within | {starts[511]:.1f} µs; the last workgroup enters at {starts[-1]:.1f} µs |")
print(f"| time after the last workgroup ENTERED (tail) | {span - starts[-1]:.1f} µs |")
# is the XCD of a workgroup what the tile order assumes (workgroup id mod 8)?
ids = np.arange(ntile)
same = np.mean(xcc == (ids & 7))
print(f"| workgroups whose XCC_ID equals (workgroup id mod 8) | {100 * same:.1f} % |")
for off in range(8):
    m = np.mean(xcc == ((ids + off) & 7))
    if m > 0.5 and off: print(f"| … equals (workgroup id + {off}) mod 8 | {100 * m:.1f} % |")
print(f"| distinct XCC_IDs seen | {len(np.unique(xcc))} |")
# what the first-round stagger assumes (tv.hip fused_stagger): workgroups 0..255 land on 256 different CUs, and workgroups
# 256..511 are the SECOND workgroup of those CUs
if ntile >= 512:
    first, second = cu_key[:256], cu_key[256:512]
    print(f"| distinct CUs of workgroups 0..255 / 256..511 | {len(np.unique(first))} / {len(np.unique(second))} |")
    print(f"| workgroups 256..511 on a CU that already hosts one of 0..255 | {100 * np.mean(np.isin(second, first)):.1f} % |")
    e0, e1 = us[:256, 0], us[256:512, 0]
    print(f"| entry of workgroups 0..255 / 256..511 (median) | {np.median(e0):.2f} / {np.median(e1):.2f} µs |")
    l0, l1 = us[:256, 1] - us[:256, 0], us[256:512, 1] - us[256:512, 0]
    print(f"| their wait for the region (median) | {np.median(l0):.2f} / {np.median(l1):.2f} µs |")

# ---- per-wave stamps around the two barriers of every fused iteration (every 16th workgroup, shader-clock counter)
wave = None
try:
    ctx.lib.sbtv_debug_timeline_waves.restype = C.c_int
    ctx.lib.sbtv_debug_timeline_waves.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    ns_, ev_, sl_ = C.c_int(0), C.c_int(0), C.c_int(0)
    raw = np.zeros(128 * 16 * 24, dtype=np.uint64)
    assert ctx.lib.sbtv_debug_timeline_waves(ctx.h, raw.ctypes.data_as(C.c_void_p), C.byref(ns_), C.byref(ev_), C.byref(sl_)) == 0
    tw = raw.reshape(ns_.value, 16, sl_.value)[:min(ns_.value, (ntile + ev_.value - 1) // ev_.value), :8, :21].astype(np.int64)
    ok = np.all(tw[:, :, 0] > 0, axis=1)
    tw = tw[ok]
    # per step: compute A = (after barrier 2 of the previous step | loop entry) -> before barrier 1 ; wait 1 ; compute B =
    # after barrier 1 -> before barrier 2 ; wait 2
    a = np.stack([tw[:, :, 1 + 4 * s] - (tw[:, :, 0] if s == 0 else tw[:, :, 4 * s]) for s in range(5)], axis=2)
    w1 = np.stack([tw[:, :, 2 + 4 * s] - tw[:, :, 1 + 4 * s] for s in range(5)], axis=2)
    b_ = np.stack([tw[:, :, 3 + 4 * s] - tw[:, :, 2 + 4 * s] for s in range(5)], axis=2)
    w2 = np.stack([tw[:, :, 4 + 4 * s] - tw[:, :, 3 + 4 * s] for s in range(5)], axis=2)
    tot = (tw[:, :, 20] - tw[:, :, 0]).astype(float)
    wave = dict(n=int(tw.shape[0]), a=a, w1=w1, b=b_, w2=w2, tot=tot)
    print("\n## Inside the five iterations: every wave's time between and AT the two barriers (shader-clock ticks)\n")
    print(f"{tw.shape[0]} sampled workgroups x 8 waves.  A = from barrier 2 to barrier 1 (split schedule: columns 2..0 + u of column 0 of "
          "the next step; first schedule: only the exchange of the seam column); B = from barrier 1 to barrier 2 (split schedule: "
          "the last column; first schedule: all four columns).\n")
    print("| wave | total ticks (5 its) | in A | waiting at barrier 1 | in B | waiting at barrier 2 |\n|---|---|---|---|---|---|")
    for wv in range(8):
        t = tot[:, wv].mean()
        print(f"| {wv} | {t:.0f} | {100 * a[:, wv].sum(axis=1).mean() / t:.1f} % | {100 * w1[:, wv].sum(axis=1).mean() / t:.1f} % | "
              f"{100 * b_[:, wv].sum(axis=1).mean() / t:.1f} % | {100 * w2[:, wv].sum(axis=1).mean() / t:.1f} % |")
    t = tot.mean()
    print(f"| all | {t:.0f} | {100 * a.sum(axis=2).mean() / t:.1f} % | {100 * w1.sum(axis=2).mean() / t:.1f} % | "
          f"{100 * b_.sum(axis=2).mean() / t:.1f} % | {100 * w2.sum(axis=2).mean() / t:.1f} % |")
    print(f"\nB per step and wave: median {np.median(b_):.0f} ticks (10 % {np.percentile(b_, 10):.0f}, 90 % {np.percentile(b_, 90):.0f}); "
          f"A: median {np.median(a):.0f}; barrier-1 wait median {np.median(w1):.0f}, barrier-2 wait median {np.median(w2):.0f}.")
except Exception as e:      # an older debug library without the per-wave stamps
    print(f"\n(per-wave stamps not available: {e})")

# machine-readable summary for bench.py's roofline block (tail_frac), stamped with the kernel sources it was measured on
if os.environ.get("TIMELINE_JSON"):
    import json
    json.dump({"source_sha256": bench.source_sha(), "image": [2048, 2048], "workgroups": int(ntile),
               "launch_us": float(span), "tail_us": float(span - starts[-1]), "last_entry_us": float(starts[-1]),
               "workgroup_life_us_median": float(np.median(life)), "load_wait_us_median": float(np.median(load)),
               "iterate_us_median": float(np.median(comp)), "store_issue_us_median": float(np.median(store)),
               "cu_two_workgroups_iterating_frac": float(busy[2]), "cu_no_workgroup_iterating_frac": float(busy[0]),
               "wave_phase_fracs": None if wave is None else {
                   "exchange_phase": float(wave["a"].sum(axis=2).mean() / wave["tot"].mean()),
                   "wait_barrier1": float(wave["w1"].sum(axis=2).mean() / wave["tot"].mean()),
                   "column_phase": float(wave["b"].sum(axis=2).mean() / wave["tot"].mean()),
                   "wait_barrier2": float(wave["w2"].sum(axis=2).mean() / wave["tot"].mean())},
               "how": "tools/chambolle_timeline.py on `make timeline` (clock stamps by thread 0 of every workgroup), last "
                      "launch of a 40-iteration SALSA solve at 2048^2"},
              open(os.environ["TIMELINE_JSON"], "w"), indent=1)
