"""GPU: the kernel variants selected by environment hooks agree with the default build.
(The hooks are read once per process, so each variant runs in a child process.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import LAB_LIB

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = {"SBTV_LIBRARY": LAB_LIB}      # variants only the lab build carries

CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "semi-blind-image-deblurring-problems-with-tv_amd"))
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import sbtv
from conftest import synth_image
g = synth_image(256, 192, 5) + np.random.default_rng(0).standard_normal((256, 192))
f, px, py, k, err = sbtv.chambolle_prox_TV_stop(g, "lambda", 9.0, "maxiter", 12, return_info=True)
x = synth_image(128, 128, 6)
A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
st = sbtv.demo_setup("gaussian", x, np.random.default_rng(1).standard_normal(x.shape), evMax=1.0)
mu = 0.003
out = sbtv.SALSA_v2(st["y"], A, 0.03 * st["sigma"] ** 2, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", x,
                    "ToleranceA", 1e-5, "MAXITERA", 100, "TVINITIALIZATION", 1, "TViters", 10)
np.savez(sys.argv[1], f=f, px=px, py=py, k=k, err=err, x=out[0], obj=out[3])
"""


def _run(tmp_path, name, env):
    out = str(tmp_path / (name + ".npz"))
    e = dict(os.environ)
    e.update(env)
    subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, out], check=True, env=e, timeout=600)
    return np.load(out)


def test_exact_fast_single_step_and_tile_variants_agree(tmp_path):
    ref = _run(tmp_path, "default", {})
    variants = {
        "exact": {"SBTV_EXACT": "1"},                       # IEEE div/sqrt, no FMA contraction
        "single": {"SBTV_SINGLE_STEP": "1"},                # one-iteration kernels (the fallback path)
        "tile_8_4": dict(LAB, SBTV_FUSED_VARIANT="8,4,2"),  # other tile geometries of the fused kernel (lab build)
        "tile_16_4": dict(LAB, SBTV_FUSED_VARIANT="16,4,1"),
        "nospec": dict(LAB, SBTV_FUSED_VARIANT="4,8,2"),
        "lab_default": dict(LAB),                           # the lab build without any switch = the default kernels
        "rows1": {"SBTV_FUSED_VARIANT": "4,8,4,1"},         # one row per lane (64-row tiles; the default on small grids)
        "rows2": {"SBTV_FUSED_VARIANT": "4,8,4"},           # the 128-row tiles large images get, forced on these small ones
        "inline": {"SBTV_INLINE_CTRL": "1"},                # stop rule applied by the last workgroup of a launch
        "separate": {"SBTV_INLINE_CTRL": "0"},              # ... or always by the separate control kernel
    }
    for name, env in variants.items():
        got = _run(tmp_path, name, env)
        assert int(got["k"][0]) == int(ref["k"][0]) == 12
        assert float(got["err"][0]) == pytest.approx(float(ref["err"][0]), rel=1e-12), name
        np.testing.assert_allclose(got["px"], ref["px"], rtol=1e-12, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(got["f"], ref["f"], rtol=1e-12, atol=1e-10, err_msg=name)
        assert len(got["obj"]) == len(ref["obj"]), name      # same SALSA stopping iteration
        np.testing.assert_allclose(got["obj"], ref["obj"], rtol=1e-10, err_msg=name)
        assert np.max(np.abs(got["x"] - ref["x"])) < 1e-8, name


GRAPH_CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "semi-blind-image-deblurring-problems-with-tv_amd"))
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import sbtv
from conftest import synth_image
x = synth_image(128, 128, 6)
res = {}
for kind, p, names in (("gaussian", (0.4, 0.3), ("w1", "w2")), ("moffat", (0.4, 3.5), ("alpha", "beta"))):
    st = sbtv.demo_setup(kind, x, np.random.default_rng(1).standard_normal(x.shape), evMax=1.0)
    if kind == "gaussian":
        A = sbtv.BlurOperator(sbtv.psf_family(kind, 7, p)[0])
        mu = 0.003
        for crit in (1, 2):
            out = sbtv.SALSA_v2(st["y"], A, 0.03 * st["sigma"] ** 2, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", x,
                                "StopCriterion", crit, "ToleranceA", 1e-5 if crit == 1 else 1e-4, "MAXITERA", 80,
                                "TVINITIALIZATION", 1, "TViters", 10)
            res["salsa_x%%d" %% crit], res["salsa_obj%%d" %% crit], res["salsa_mse%%d" %% crit] = out[0], out[3], out[6]
    # SAPG with the device Philox stream: PSF parameters fixed (gaussian demo) and estimated (moffat demo)
    op = dict(samples=40, warmup=12, burnIn=20, psf_size=7, phi=0.0, gamma=st["gamma"], th_init=0.01, min_th=1e-3,
              max_th=1.0, sigma=st["sigma"], sigma_init=st["sigma_init"], sigma_min=st["sigma_min"],
              sigma_max=st["sigma_max"], d_scale=1.0, d_exp=0.8, fix_sigma=0, seed=7)
    op["lambda"] = st["lambda"]
    fix = kind == "gaussian"
    for q, nm in enumerate(names):
        op[nm] = p[q]
        op[nm + "_init"] = p[q] if fix else (1.0, 10.0)[q]
        op["min_" + nm], op["max_" + nm], op["fix_" + nm] = (0.1, 0.1)[q] if fix else (1e-2, 0.1)[q], (1.0, 10.0)[q], int(fix)
    c = dict(theta=0.01, w1=10.0, w2=10.0, alpha=10.0, beta=1e4, sigma=1000.0, lam=1.0, gam=1.0)
    fn = sbtv.SAPG_algorithm_Guassian if fix else sbtv.SAPG_algorithm_moffat
    r = fn(st["y"], op, c)[-1]
    for key in ("thetas", "sigmas", "logPiTraceX", "logPiTrace_WU", names[0] + "s", names[1] + "s", "Xlast_sample"):
        res[kind + "_" + key] = np.asarray(r[key])
np.savez(sys.argv[1], **res)
"""


def test_graph_replay_is_bit_identical_to_eager_launches(tmp_path):
    """Small problems replay a captured hipGraph per iteration (SALSA_v2, SAPG warm-up and main loop); the
    kernels and their order are the same, so every trace must be bit-identical to eager launching."""
    outs = {}
    for name, env in (("eager", {"SBTV_GRAPH": "0"}), ("graph", {"SBTV_GRAPH": "1"}), ("default", {})):
        out = str(tmp_path / (name + ".npz"))
        e = dict(os.environ)
        e.pop("SBTV_GRAPH", None)
        e.update(env)
        subprocess.run([sys.executable, "-c", GRAPH_CHILD % {"root": ROOT}, out], check=True, env=e, timeout=600)
        outs[name] = np.load(out)
    for name in ("graph", "default"):
        for key in outs["eager"].files:
            np.testing.assert_array_equal(outs[name][key], outs["eager"][key], err_msg=f"{name}:{key}")
    assert len(outs["eager"]["salsa_obj1"]) > 10 and outs["eager"]["moffat_alphas"][-1] != outs["eager"]["moffat_alphas"][0]


LOOP_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "semi-blind-image-deblurring-problems-with-tv_amd"))
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import sbtv
from conftest import synth_image
M = N = 1024                                   # a size of the wave-granular column pass (fused epilogues)
x = synth_image(M, N, 5)
res = {}
st = sbtv.demo_setup("moffat", x, np.random.default_rng(1).standard_normal(x.shape), evMax=1.0)
A = sbtv.BlurOperator(sbtv.psf_family("moffat", 7, (0.4, 3.5))[0])
xf, obj, times, mses = sbtv.my_fista(st["y"], A, A.T, 0.03 * st["sigma"] ** 2, 1.0, sbtv.TVnorm, sbtv.Psi_TV(25), 1, 0.0, 5, x)
res["fista_x"], res["fista_obj"], res["fista_mse"] = xf, obj, mses
st = sbtv.demo_setup("gaussian", x, np.random.default_rng(2).standard_normal(x.shape), evMax=1.0)
A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
out = sbtv.SALSA_v2(st["y"], A, 0.03 * st["sigma"] ** 2, "MU", 0.003, "AT", A.T, "LS", A.LS(0.003), "True_x", x,
                    "ToleranceA", -1.0, "MAXITERA", 6, "TVINITIALIZATION", 1, "TViters", 10)
res["salsa_x"], res["salsa_obj"] = out[0], out[3]
for fixed in (1, 0):
    op = dict(samples=6, warmup=3, burnIn=2, psf_size=7, phi=0.0, gamma=st["gamma"], th_init=0.01, min_th=1e-3, max_th=1.0,
              sigma=st["sigma"], sigma_init=st["sigma_init"], sigma_min=st["sigma_min"], sigma_max=st["sigma_max"], d_scale=1.0,
              d_exp=0.8, fix_sigma=0, seed=7, w1=0.4, w2=0.3, w1_init=0.5, w2_init=0.3, min_w1=0.1, min_w2=0.1, max_w1=1.0,
              max_w2=1.0, fix_w1=fixed, fix_w2=fixed)
    op["lambda"] = st["lambda"]
    c = dict(theta=0.01, w1=10.0, w2=10.0, sigma=1000.0, lam=1.0, gam=1.0)
    r = sbtv.SAPG_algorithm_Guassian(st["y"], op, c)[-1]
    r = r[0] if isinstance(r, list) else r
    for key in ("thetas", "sigmas", "logPiTraceX", "gXTrace", "Xlast_sample", "w1s"):
        res["sapg%%d_%%s" %% (fixed, key)] = np.asarray(r[key])
np.savez(sys.argv[1], **res)
"""


def test_loop_hooks_agree_with_the_default_loops(tmp_path):
    """The launch-saving forms of the solver loops against their plain forms, each selected by its environment hook in a
    child process: host one iteration late / synchronised (FISTA), gradient step and MYULA step as epilogues of the inverse
    column pass / as element-wise kernels, stop rule of the SAPG prox deferred / right after the launches, optimistic /
    exact prox launches, riding / stand-alone collector.  Same bits, except the fused MYULA epilogue (<= 2 ulp of X)."""
    def run(name, env):
        out = str(tmp_path / (name + ".npz"))
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, "-c", LOOP_CHILD % {"root": ROOT}, out], check=True, env=e, timeout=600)
        return np.load(out)
    ref = run("default", {})
    exact = {"fista_lag0": {"SBTV_FISTA_LAG": "0"}, "fista_two_pass": {"SBTV_FISTA_FUSED_STEP": "0"},
             "sapg_rule_kernel": {"SBTV_SAPG_DEFER": "0"}, "exact_prox": {"SBTV_PROX_SPEC": "0"},
             "own_collector": {"SBTV_COLLECT_RIDE": "0"}}
    for name, env in exact.items():
        got = run(name, env)
        for key in ref.files:
            np.testing.assert_array_equal(got[key], ref[key], err_msg=name + " " + key)
    got = run("myula_two_pass", {"SBTV_SAPG_FUSED_MYULA": "0"})
    for key in ref.files:
        if key.startswith("sapg"):
            np.testing.assert_allclose(got[key], ref[key], rtol=1e-11, atol=1e-11, err_msg=key)
        else:
            np.testing.assert_array_equal(got[key], ref[key], err_msg=key)


ROWS_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "semi-blind-image-deblurring-problems-with-tv_amd"))
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import sbtv
from conftest import synth_image
res = {}
for tag, (M, N) in (("a", (2048, 1024)), ("b", (1024, 2048))):
    x = synth_image(M, N, 9)
    st = sbtv.demo_setup("gaussian", x, np.random.default_rng(3).standard_normal(x.shape), evMax=1.0)
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
    mu = 0.003
    res[tag + "_Ax"], res[tag + "_ATx"], res[tag + "_LSx"] = A(x), A.T(x), A.LS(mu)(x)
    out = sbtv.SALSA_v2(st["y"], A, 0.03 * st["sigma"] ** 2, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", x,
                        "ToleranceA", 1e-9, "MAXITERA", 8, "TVINITIALIZATION", 1, "TViters", 10)
    res[tag + "_x"], res[tag + "_obj"] = out[0], out[3]
    got = sbtv.csalsa(st["y"], A, 0.5, 0.5, st["sigma"], "AT", A.T, "LS", A.invLS, "TVINITIALIZATION", 1, "TVITERS", 5,
                      "STOPCRITERION", 3, "TOLERANCEA", 1e-9, "MAXITERA", 8, "TRUE_X", x, "VERBOSE", 0)
    res[tag + "_cx"], res[tag + "_cobj"], res[tag + "_ccrit"] = got[0], got[3], got[6]
np.savez(sys.argv[1], **res)
"""


def test_lab_row_pass_variants_agree_at_the_wave_granular_sizes(tmp_path):
    """The row passes the lab build keeps for M, N in {1024, 2048} against the default library's pipelined kernel: four
    wave-local sub-transforms per row on sub-row-major operator spectra (SBTV_ROWS_SUB=1, round 3) and the workgroup kernel
    on the tiled layout (SBTV_ROWS_PIPE=0); plain operator applications, SALSA (OP_SALSA) and C-SALSA (OP_CSALSA)."""
    def run(name, env):
        out = str(tmp_path / (name + ".npz"))
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, "-c", ROWS_CHILD % {"root": ROOT}, out], check=True, env=e, timeout=900)
        return np.load(out)
    ref = run("default", {})
    for name, env in (("rows_sub", dict(LAB, SBTV_ROWS_SUB="1")), ("rows_wg", dict(LAB, SBTV_ROWS_PIPE="0"))):
        got = run(name, env)
        for key in ref.files:
            scale = float(np.max(np.abs(ref[key])))
            assert np.max(np.abs(got[key] - ref[key])) <= 1e-10 * scale, (name, key)


TAIL_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "semi-blind-image-deblurring-problems-with-tv_amd"))
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import sbtv
from conftest import synth_image
g = synth_image(2048, 2048, 13) + np.random.default_rng(5).standard_normal((2048, 2048))
f, px, py, k, err = sbtv.chambolle_prox_TV_stop(g, "lambda", 7.0, "maxiter", 25, return_info=True)
f2, px2, py2, k2, err2 = sbtv.chambolle_prox_TV_stop(g, "lambda", 7.0, "maxiter", 10, "dualvars", np.hstack([px, py]), return_info=True)
np.savez(sys.argv[1], f=f, px=px, py=py, k=k, err=err, f2=f2, px2=px2, k2=k2, err2=err2)
"""


def test_lab_mixed_tiling_agrees_with_the_128_row_tiles(tmp_path):
    """SBTV_TAIL_HALF=1 (lab build): the last workgroups of a launch work on 64-row tiles (one row per lane) below the
    128-row ones, three splits of the tile rows; cold prox of 25 iterations and a warm-started one of 10 at 2048 x 2048."""
    def run(name, env):
        out = str(tmp_path / (name + ".npz"))
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, "-c", TAIL_CHILD % {"root": ROOT}, out], check=True, env=e, timeout=900)
        return np.load(out)
    ref = run("default", {})
    for name, env in (("auto", dict(LAB, SBTV_TAIL_HALF="1")), ("rows1", dict(LAB, SBTV_TAIL_HALF="1", SBTV_TAIL_ROWS="1")),
                      ("rows2", dict(LAB, SBTV_TAIL_HALF="1", SBTV_TAIL_ROWS="2"))):
        got = run(name, env)
        assert int(got["k"][0]) == int(ref["k"][0]) == 25 and int(got["k2"][0]) == int(ref["k2"][0]) == 10
        for key in ("err", "err2"):
            assert float(got[key][0]) == pytest.approx(float(ref[key][0]), rel=1e-12), (name, key)
        for key in ("f", "px", "py", "f2", "px2"):
            np.testing.assert_allclose(got[key], ref[key], rtol=1e-12, atol=1e-12, err_msg=name + " " + key)
