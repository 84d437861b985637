"""GPU parity of the TV-prox kernel that the benchmark times: the default-geometry temporally fused kernel
`chambolle_fused_kernel<4, 8, 4, FAST>` (two rows per lane, 128-row x 32-column regions), which `prox_plan`
selects when an image has at least 256 tiles (from about 1024^2 up) and which the small-image tests in
test_gpu_tv.py therefore never reach.  Whole-image comparison with the oracle
(utils/chambolle_prox_TV_stop.m:120-149): k, err, px, py, f at the same bars as the small cases.

Arithmetic note: the default build of the fused kernels is FAST (v_rcp_f64 / v_rsq_f64 + Newton / Goldschmidt steps,
FMA contraction), every operation within ~1 ulp of IEEE; SBTV_EXACT=1 selects IEEE div / sqrt without contraction.
Both meet the bars below."""
import numpy as np
import pytest

from conftest import synth_image

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-12, atol=1e-12)


def _assert_default_geometry(ctx, M, N, batch=1):
    """These sizes take the default-geometry TILE kernel - or, when the whole file is re-run with SBTV_PROX_PIPE=1
    (test_pipeline_kernel_passes_the_prox_parity_suite), the streaming pipeline kernel."""
    import os
    v = ctx.prox_variant(M, N, batch)
    if os.environ.get("SBTV_PROX_PIPE") == "1":
        assert v["kind"] == "pipeline", v
        return v
    assert v["kind"] == "tile" and (v["cols_per_wave"], v["waves"], v["waves_per_simd"], v["rows_per_lane"]) == (4, 8, 4, 2), v
    assert v["tiles"] * batch >= 256, v
    return v


def _image(M, N, seed):
    return synth_image(M, N, seed) + np.random.default_rng(seed).standard_normal((M, N))


def _compare(got, want, ftol=1e-10):
    f, px, py, k, err = got
    fo, pxo, pyo, ko, erro = want
    assert int(np.ravel(k)[0]) == ko
    assert float(np.ravel(err)[0]) == pytest.approx(erro, rel=1e-12)
    np.testing.assert_allclose(px, pxo, **TOL)
    np.testing.assert_allclose(py, pyo, **TOL)
    np.testing.assert_allclose(f, fo, rtol=1e-12, atol=ftol)


@pytest.mark.parametrize("shape", [(1024, 1024), (1100, 900)])
def test_default_fused_kernel_cold_25_matches_oracle(ctx, shape):
    """K = 25 cold start (the SAPG / FISTA prox, run_Gaussian_demo.m:188-192) on a whole image; 1100 x 900 is
    even but a multiple of neither the 116-row nor the 21-column core tile."""
    import sbtv
    import sbtv_oracle as o
    M, N = shape
    _assert_default_geometry(ctx, M, N)
    g = _image(M, N, 31)
    lam = 7.5
    got = sbtv.chambolle_prox_TV_stop(g, "lambda", lam, "maxiter", 25, return_info=True)
    want = o.chambolle_prox_TV_stop(g, lam=lam, maxiter=25, return_info=True)
    _compare(got, want)


@pytest.mark.parametrize("shape", [(1024, 1024), (1100, 900)])
def test_default_fused_kernel_warm_started_twice_matches_oracle(ctx, shape):
    """K = 10 three times, the 2nd and 3rd call warm-started from the duals of the previous one: what SALSA_v2 does
    in every outer iteration (SALSA_v2.m:429), here with g changing between the calls as it does there."""
    import sbtv
    import sbtv_oracle as o
    M, N = shape
    _assert_default_geometry(ctx, M, N)
    g = _image(M, N, 32)
    rng = np.random.default_rng(5)
    lam = 10.0
    px = py = pxo = pyo = None
    for call in range(3):
        gc = g + 0.5 * call * rng.standard_normal((M, N))
        # device-resident buffers: 'dualvars' is then the (px, py) pair, which also works for M != N (the [px py]
        # array form splits with M, quirk Q2, and is covered by the square test below)
        gd = sbtv.to_device(gc)
        if call == 0:
            got = sbtv.chambolle_prox_TV_stop(gd, "lambda", lam, "maxiter", 10, return_info=True)
            want = o.chambolle_prox_TV_stop(gc, lam=lam, maxiter=10, return_info=True)
        else:
            got = sbtv.chambolle_prox_TV_stop(gd, "lambda", lam, "maxiter", 10, "dualvars", (px, py), return_info=True)
            want = o.chambolle_prox_TV_stop(gc, lam=lam, maxiter=10, dualvars=(pxo, pyo), return_info=True)
        px, py = got[1], got[2]
        _compare((sbtv.to_host(got[0]), sbtv.to_host(px), sbtv.to_host(py), got[3], got[4]), want)
        pxo, pyo = want[1], want[2]


def test_default_fused_kernel_square_dualvars_split(ctx):
    """The reference splits 'dualvars' = [px py] with M for the columns (quirk Q2, square images only): the mirror
    accepts the same M x 2N array."""
    import sbtv
    import sbtv_oracle as o
    M = N = 1024
    g = _image(M, N, 33)
    f1, px1, py1 = sbtv.chambolle_prox_TV_stop(g, "lambda", 4.0, "maxiter", 5)
    got = sbtv.chambolle_prox_TV_stop(g, "lambda", 4.0, "maxiter", 5, "dualvars", np.hstack([px1, py1]), return_info=True)
    want = o.chambolle_prox_TV_stop(g, lam=4.0, maxiter=10, return_info=True)      # 5 + 5 warm == 10 cold
    _compare(got[:3] + (np.array([want[3]]), got[4]), want)
    assert int(got[3][0]) == 5


@pytest.mark.parametrize("kstop", [3, 5, 7, 10, 11])
def test_default_fused_kernel_stop_rule_mid_launch_and_at_boundary(ctx, kstop):
    """`cont = (k<MaxIter) & (err>tol)` (:131) with MaxIter = 15 = three fused launches of 5: a tolerance that is
    met in the middle of a launch (3, 7, 11: the over-run launch is repeated for exactly that many steps by the
    redo pass), exactly at a launch boundary (5, 10: the finish-only pass writes f) must give the reference's
    k, err, p and f on the default-geometry kernel."""
    import sbtv
    import sbtv_oracle as o
    M, N = 1100, 900
    _assert_default_geometry(ctx, M, N)
    g = _image(M, N, 34)
    lam, K = 8.0, 15
    # err_k of the first K iterations from single warm-started oracle iterations
    errs, p = [], None
    for k in range(K):
        _, pxo, pyo, _, e = o.chambolle_prox_TV_stop(g, lam=lam, maxiter=1, tol=0.0, dualvars=p, return_info=True)
        p = (pxo, pyo)
        errs.append(e)
    assert all(a > b for a, b in zip(errs, errs[1:])), "test assumes a decreasing err sequence"
    tol = 0.5 * (errs[kstop - 1] + errs[kstop - 2])        # err_{kstop-1} > tol > err_kstop
    want = o.chambolle_prox_TV_stop(g, lam=lam, maxiter=K, tol=tol, return_info=True)
    assert want[3] == kstop
    got = sbtv.chambolle_prox_TV_stop(g, "lambda", lam, "maxiter", K, "tol", tol, return_info=True)
    _compare(got, want)


def test_default_fused_kernel_batch_of_two_1024(ctx):
    """Batch path of the same kernel: two images with their own lambda, one of them stopping early."""
    import sbtv
    import sbtv_oracle as o
    M = N = 1024
    _assert_default_geometry(ctx, M, N, 2)
    g = np.stack([_image(M, N, 35), np.full((M, N), 17.0)])
    lam = np.array([6.0, 3.0])
    f, px, py, k, err = sbtv.chambolle_prox_TV_stop(g, "lambda", lam, "maxiter", 10, return_info=True)
    want0 = o.chambolle_prox_TV_stop(g[0], lam=6.0, maxiter=10, return_info=True)
    _compare((f[0], px[0], py[0], k[:1], err[:1]), want0)
    assert k[1] == 1 and err[1] == 0.0 and np.array_equal(f[1], g[1])      # constant image: err = 0 at k = 1


def test_pipeline_kernel_2048_matches_oracle(ctx, man512):
    """The streaming pipeline kernel (csrc/tv_pipe.inc, opt-in) at the benchmark size: 19 bands x 13 column segments,
    ten iterations per launch.  Cold K = 10 with f, then warm-started K = 10 on a changed g (what
    SALSA_v2.m:429 does every outer iteration), then K = 25 cold (three launches: 9 + 8 + 8) against the oracle."""
    import os
    import sbtv
    import sbtv_oracle as o
    if os.environ.get("SBTV_PROX_PIPE") != "1":
        pytest.skip("runs in the child process of test_pipeline_kernel_passes_the_prox_parity_suite (SBTV_PROX_PIPE=1)")
    M = N = 2048
    v = ctx.prox_variant(M, N, 1)
    assert v["kind"] == "pipeline" and v["tiles"] <= 256, v
    g = np.tile(man512, (4, 4)) + np.random.default_rng(41).standard_normal((M, N))
    lam = 9.0
    gd = sbtv.to_device(g)
    got = sbtv.chambolle_prox_TV_stop(gd, "lambda", lam, "maxiter", 10, return_info=True)
    want = o.chambolle_prox_TV_stop(g, lam=lam, maxiter=10, return_info=True)
    _compare((sbtv.to_host(got[0]), sbtv.to_host(got[1]), sbtv.to_host(got[2]), got[3], got[4]), want)
    g2 = g + 0.7 * np.random.default_rng(42).standard_normal((M, N))
    got2 = sbtv.chambolle_prox_TV_stop(sbtv.to_device(g2), "lambda", lam, "maxiter", 10, "dualvars", (got[1], got[2]),
                                       return_info=True)
    want2 = o.chambolle_prox_TV_stop(g2, lam=lam, maxiter=10, dualvars=(want[1], want[2]), return_info=True)
    _compare((sbtv.to_host(got2[0]), sbtv.to_host(got2[1]), sbtv.to_host(got2[2]), got2[3], got2[4]), want2)
    got3 = sbtv.chambolle_prox_TV_stop(gd, "lambda", lam, "maxiter", 25, return_info=True)
    want3 = o.chambolle_prox_TV_stop(g, lam=lam, maxiter=25, return_info=True)
    _compare((sbtv.to_host(got3[0]), sbtv.to_host(got3[1]), sbtv.to_host(got3[2]), got3[3], got3[4]), want3)


def test_pipeline_kernel_passes_the_prox_parity_suite():
    """Every TV-prox parity test of this file and of test_gpu_tv.py (ragged sizes from 2 x 2, warm start, batch,
    early exit, the stop rule firing inside / between launches) once more with the streaming pipeline kernel forced
    for every even M (SBTV_PROX_PIPE=1 is read once per process, hence the child process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from conftest import LAB_LIB
    env = dict(os.environ, SBTV_PROX_PIPE="1", SBTV_LIBRARY=LAB_LIB)      # the pipeline kernel lives in the lab build
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_tv.py"),
                        os.path.join(root, "tests", "test_gpu_tv_large.py"), "-m", "gpu", "-x", "-q", "-k",
                        "not passes_the_prox_parity_suite"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
