"""CPU: register / scratch budget of the kernels on the benchmarked path (hipcc cross-compiles gfx950 without a GPU).

These kernels sit at the edge of their register budgets on purpose (the fused Chambolle kernel needs <= 128 VGPRs for four
waves per SIMD, the inverse column pass with the SALSA bookkeeping <= 256 without spilling); twice in this project a
harmless-looking edit made the compiler spill 40-90 registers to scratch and cost 5-10 % of the SALSA iteration without
failing any numerical test (round 3: a run-time branch around one store).  This test reads the
compiler's own resource report (-Rpass-analysis=kernel-resource-usage) and fails on scratch use or a lost occupancy."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _report(unit):
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.devnull, os.path.join(CSRC, unit)],
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    out, cur = {}, None
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def _find(rep, *parts):
    names = [n for n in rep if all(p in n for p in parts)]
    assert len(names) == 1, (parts, names)
    return rep[names[0]]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_fused_chambolle_kernel_keeps_four_waves_per_simd_without_scratch():
    rep = _report("tv.hip")
    # template <FCJ, FNW, MINW, FAST, MIX, ESUB>: ...Lb1ELb0ELb0E sums the error of every pixel (exact launches, SAPG),
    # ...Lb1ELb0ELb1E of a subset (the optimistic launches of the SALSA / FISTA / ADMM loops: what bench.py times);
    # MIX = true (mixed launch whose last workgroups run the one-row-per-lane body) exists in the lab build only
    for name in ("chambolle_fused_kernelILi4ELi8ELi4ELb1ELb0ELb0E", "chambolle_fused_kernelILi4ELi8ELi4ELb1ELb0ELb1E"):
        k = _find(rep, name)
        assert k["ScratchSize"] == 0 and k["VGPRs Spill"] == 0, (name, k)
        assert k["VGPRs"] <= 128 and k["Occupancy"] == 4, (name, k)
        assert k["LDS Size"] <= 80 * 1024, (name, k)            # two workgroups per CU


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_fft_passes_of_the_salsa_loop_do_not_spill():
    rep = _report("fft.hip")
    for parts in (("cols_inv_wave_kernelILi10ELi16ELi3E",), ("cols_inv_wave_kernelILi10ELi16ELi35E",),
                  ("cols_inv_wave_kernelILi10ELi16ELi1E",), ("cols_fwd_wave_kernelILi10ELi16E",),
                  ("cols_inv_wave_kernelILi9ELi8ELi3E",)):
        k = _find(rep, *parts)
        assert k["ScratchSize"] == 0 and k["VGPRs Spill"] == 0, (parts, k)
        assert k["Occupancy"] >= 2, (parts, k)
    # the pipelined row pass lives AT its 256-register limit (128 of them operands in flight): the
    # compiler parks 4 registers (20 bytes per lane) in scratch; more than a handful would show in the loop
    k = _find(rep, "rows_pipe_kernelILi11ELi4E")
    assert k["ScratchSize"] <= 32 and k["VGPRs Spill"] <= 6 and k["Occupancy"] == 2, k
    # OP_CSALSA carries a fourth spectrum (read and written): 27 registers in scratch with Y requested after the forward
    # transform (46 with all operands up front: DESIGN.md section 3.4); the 1024-point variant has room
    k = _find(rep, "rows_pipe_kernelILi11ELi9E")
    assert k["ScratchSize"] <= 128 and k["VGPRs Spill"] <= 32, k
    k = _find(rep, "rows_pipe_kernelILi10ELi9E")
    assert k["ScratchSize"] == 0, k
    # C-SALSA's inverse column pass with x - bu as its epilogue
    k = _find(rep, "cols_inv_wave_kernelILi10ELi16ELi64E")
    assert k["ScratchSize"] == 0 and k["Occupancy"] >= 2, k
