"""GPU parity: TV prox (K1/K2) and periodic TV norm (K3) through the C-ABI vs the oracle."""
import numpy as np
import pytest

from conftest import synth_image

pytestmark = pytest.mark.gpu

# Arithmetic: the default build of the fused kernels is FAST (v_rcp_f64 / v_rsq_f64 seeds refined by Newton /
# Goldschmidt steps, FMA contraction on): every operation stays within ~1 ulp of the IEEE result the NumPy oracle
# computes, and the dual iteration is non-expansive, so element-wise results agree to ~1e-14; the one-iteration
# kernels (odd M) and SBTV_EXACT=1 use IEEE sqrt / div with contraction off.  The reduction order of err / TVnorm
# differs from NumPy's in every build.  These sizes take the one-row-per-lane fused kernel (fewer than 256 tiles)
# or the scalar kernel; the two-rows-per-lane kernel that large images take is covered by test_gpu_tv_large.py.
TOL = dict(rtol=1e-12, atol=1e-12)


def test_kat_3x3(ctx):
    import sbtv
    g = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 10.0]])
    f, px, py, k, err = sbtv.chambolle_prox_TV_stop(g, "lambda", 2, "maxiter", 1, return_info=True)
    assert k[0] == 1 and err[0] == pytest.approx(4.183300132670378, rel=1e-14)
    assert f[0, 0] == pytest.approx(1.714642641645, abs=1e-11)
    assert px[0, 0] == pytest.approx(-0.267990990617, abs=1e-11)
    assert py[2, 1] == pytest.approx(-0.199359487590, abs=1e-11)
    f, px, py, k, err = sbtv.chambolle_prox_TV_stop(g, "LAMBDA", 2, "MaxIter", 3, return_info=True)
    assert err[0] == pytest.approx(1.943468133580, abs=1e-11) and f[0, 0] == pytest.approx(2.464881599466, abs=1e-11)
    assert sbtv.TVnorm(g) == pytest.approx(41.28554656058845, rel=1e-14)


@pytest.mark.parametrize("shape", [(2, 2), (3, 7), (17, 12), (64, 64), (129, 33), (128, 16), (130, 18), (256, 200)])
@pytest.mark.parametrize("K", [1, 5])
def test_prox_matches_oracle_ragged(ctx, shape, K):
    import sbtv
    import sbtv_oracle as o
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    g = rng.uniform(0, 255, shape)
    lam = 6.5
    f, px, py, k, err = sbtv.chambolle_prox_TV_stop(g, "lambda", lam, "maxiter", K, return_info=True)
    fo, pxo, pyo, ko, erro = o.chambolle_prox_TV_stop(g, lam=lam, maxiter=K, return_info=True)
    assert k[0] == ko
    assert err[0] == pytest.approx(erro, rel=1e-12)
    np.testing.assert_allclose(px, pxo, **TOL)
    np.testing.assert_allclose(py, pyo, **TOL)
    np.testing.assert_allclose(f, fo, rtol=1e-12, atol=1e-10)


def test_prox_warm_start_and_batch(ctx, cman256):
    import sbtv
    import sbtv_oracle as o
    rng = np.random.default_rng(7)
    g = np.stack([cman256, cman256[::-1].copy() + rng.standard_normal(cman256.shape)])
    lam = np.array([4.0, 9.0])
    # cold 10 iterations, then warm 10 more == reference called twice with dualvars
    f1, px1, py1 = sbtv.chambolle_prox_TV_stop(g, "lambda", lam, "maxiter", 10)
    f2, px2, py2, k, err = sbtv.chambolle_prox_TV_stop(g, "lambda", lam, "maxiter", 10,
                                                      "dualvars", np.concatenate([px1, py1], axis=2), return_info=True)
    for b in range(2):
        fo, pxo, pyo = o.chambolle_prox_TV_stop(g[b], lam=lam[b], maxiter=10)
        np.testing.assert_allclose(f1[b], fo, rtol=1e-12, atol=1e-10)
        fo2, pxo2, pyo2, ko, erro = o.chambolle_prox_TV_stop(g[b], lam=lam[b], maxiter=10,
                                                            dualvars=np.hstack([pxo, pyo]), return_info=True)
        assert k[b] == ko and err[b] == pytest.approx(erro, rel=1e-11)
        np.testing.assert_allclose(px2[b], pxo2, **TOL)
        np.testing.assert_allclose(py2[b], pyo2, **TOL)
        np.testing.assert_allclose(f2[b], fo2, rtol=1e-12, atol=1e-10)


def test_prox_early_exit_and_structure(ctx):
    import sbtv
    # g == 0 and constant images stop at k = 1 with err = 0 (device-side stop rule, :131)
    for g in (np.zeros((64, 48)), np.full((64, 48), 3.5)):
        f, px, py, k, err = sbtv.chambolle_prox_TV_stop(g, "lambda", 2.0, "maxiter", 10, return_info=True)
        assert k[0] == 1 and err[0] == 0.0
        assert np.array_equal(f, g) and not px.any() and not py.any()
    # mixed batch: image 0 stops at once, image 1 runs all iterations
    g = np.stack([np.zeros((32, 32)), synth_image(32, 32)])
    f, px, py, k, err = sbtv.chambolle_prox_TV_stop(g, "lambda", 5.0, "maxiter", 7, return_info=True)
    assert list(k) == [1, 7]
    # tolerance reached mid-way: huge tol stops after the first iteration
    g = synth_image(48, 40, 3)
    f, px, py, k, err = sbtv.chambolle_prox_TV_stop(g, "lambda", 5.0, "maxiter", 7, "tol", 1e9, return_info=True)
    import sbtv_oracle as o
    fo, pxo, pyo, ko, _ = o.chambolle_prox_TV_stop(g, lam=5.0, maxiter=7, tol=1e9, return_info=True)
    assert k[0] == ko == 1
    np.testing.assert_allclose(f, fo, rtol=1e-12, atol=1e-10)
    # |p| <= 1, last row of px / last col of py are zero, corner untouched (Q3)
    g = synth_image(100, 60, 4)
    f, px, py = sbtv.chambolle_prox_TV_stop(g, "lambda", 12.0, "maxiter", 25)
    assert np.all(np.sqrt(px ** 2 + py ** 2) <= 1 + 1e-12)
    assert not px[-1, :].any() and not py[:, -1].any() and f[-1, -1] == g[-1, -1]


@pytest.mark.parametrize("kstop", [2, 4, 5, 6, 8, 10, 13])
def test_prox_tolerance_stop_inside_and_between_fused_launches(ctx, kstop):
    """The stop rule (:131) firing in the middle of a fused 5-iteration launch (redo pass), exactly at a
    launch boundary (finish-only pass) and in a later launch must give the reference's k, err, p and f."""
    import sbtv
    import sbtv_oracle as o
    g = synth_image(96, 80, 21) + np.random.default_rng(3).standard_normal((96, 80))
    lam, K = 8.0, 15
    errs = []
    for k in range(1, K + 1):
        errs.append(o.chambolle_prox_TV_stop(g, lam=lam, maxiter=k, return_info=True)[4])
    assert all(a > b for a, b in zip(errs, errs[1:])), "test assumes a decreasing err sequence"
    tol = 0.5 * (errs[kstop - 1] + errs[kstop - 2])        # err_{kstop-1} > tol > err_kstop
    fo, pxo, pyo, ko, erro = o.chambolle_prox_TV_stop(g, lam=lam, maxiter=K, tol=tol, return_info=True)
    assert ko == kstop
    f, px, py, k, err = sbtv.chambolle_prox_TV_stop(g, "lambda", lam, "maxiter", K, "tol", tol, return_info=True)
    assert k[0] == kstop and err[0] == pytest.approx(erro, rel=1e-12)
    np.testing.assert_allclose(px, pxo, **TOL)
    np.testing.assert_allclose(py, pyo, **TOL)
    np.testing.assert_allclose(f, fo, rtol=1e-12, atol=1e-10)
    # batch: image 0 stops at kstop, image 1 (other lambda => other errs) runs its own course
    g2 = np.stack([g, g[::-1].copy()])
    f2, px2, py2, k2, err2 = sbtv.chambolle_prox_TV_stop(g2, "lambda", [lam, 3.0], "maxiter", K, "tol", tol,
                                                        return_info=True)
    fo1, pxo1, pyo1, ko1, erro1 = o.chambolle_prox_TV_stop(g2[1], lam=3.0, maxiter=K, tol=tol, return_info=True)
    assert list(k2) == [kstop, ko1]
    np.testing.assert_allclose(f2[0], fo, rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(f2[1], fo1, rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(px2[1], pxo1, **TOL)


def test_prox_errors(ctx):
    import sbtv
    g = np.zeros((8, 8))
    with pytest.raises(sbtv.SbtvError) as e:
        sbtv.chambolle_prox_TV_stop(g, "lambda", 1.0)            # Q1
    assert e.value.code == -3
    with pytest.raises(sbtv.SbtvError) as e:
        sbtv.chambolle_prox_TV_stop(g, "lambda", 1.0, "maxiter", 2, "dualvars", np.zeros((8, 12)))
    assert e.value.code == -4
    with pytest.raises(ValueError):
        sbtv.chambolle_prox_TV_stop(g, "lambda")


@pytest.mark.parametrize("shape", [(2, 2), (5, 9), (64, 64), (129, 33), (256, 256), (512, 512)])
def test_tvnorm_matches_oracle(ctx, shape):
    import sbtv
    import sbtv_oracle as o
    rng = np.random.default_rng(11)
    x = rng.uniform(0, 255, shape)
    assert sbtv.TVnorm(x) == pytest.approx(o.TVnorm(x), rel=1e-13)


def test_device_pointer_path_and_full_size_properties(ctx):
    """2048 x 2048 (BASELINE size): device-resident buffers, size-independent checks."""
    import torch
    import sbtv
    M = N = 2048
    g = synth_image(M, N, 5)
    gd = sbtv.to_device(g)
    f, px, py, k, err = sbtv.chambolle_prox_TV_stop(gd, "lambda", 10.0, "maxiter", 10, return_info=True)
    torch.cuda.synchronize()
    assert k[0] == 10
    pxh, pyh, fh = sbtv.to_host(px), sbtv.to_host(py), sbtv.to_host(f)
    assert np.all(np.sqrt(pxh ** 2 + pyh ** 2) <= 1 + 1e-12)
    assert not pxh[-1, :].any() and not pyh[:, -1].any() and fh[-1, -1] == g[-1, -1]
    # f = g - lambda div p, recomputed on the host from the returned duals
    import sbtv_oracle as o
    np.testing.assert_allclose(fh, g - 10.0 * o.DivergenceIm(pxh, pyh), rtol=1e-13, atol=1e-10)
    # translation of the stencil: a crop far from the borders matches the oracle run on the whole image crop?
    # (cheap exact check instead) first 128 rows/cols of one oracle iteration chain on a 256x256 corner
    # is not separable; use the 1-iteration identity: after ONE iteration from p=0, p depends on g only locally.
    f1, px1, py1 = sbtv.chambolle_prox_TV_stop(gd, "lambda", 10.0, "maxiter", 1)
    c = g[:256, :256]
    _, pxo, pyo = o.chambolle_prox_TV_stop(c, lam=10.0, maxiter=1)
    np.testing.assert_allclose(sbtv.to_host(px1)[:255, :255], pxo[:255, :255], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(sbtv.to_host(py1)[:255, :255], pyo[:255, :255], rtol=1e-12, atol=1e-12)
    # TVnorm is invariant under circular shifts
    t0 = sbtv.TVnorm(gd)
    t1 = sbtv.TVnorm(sbtv.to_device(np.roll(np.roll(g, 37, axis=0), 501, axis=1)))
    assert t0 == pytest.approx(t1, rel=1e-13)
