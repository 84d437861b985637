"""CPU: the vectorised oracle against a second, literal scalar-loop reading of the same MATLAB lines
(oracle/sbtv_oracle_loops.py, 1-based indices emulated).  With no MATLAB output to compare with ("parity
unpinned"), two independent readings agreeing to rounding is what guards the oracle against vectorisation slips."""
import numpy as np
import pytest

import sbtv_oracle as o
import sbtv_oracle_loops as L


def _img(m, n, seed):
    return np.random.default_rng(seed).uniform(0, 255, (m, n))


@pytest.mark.parametrize("m,n", [(5, 4), (4, 7), (2, 2), (3, 9)])
def test_gradient_divergence_match(m, n):
    rng = np.random.default_rng(m * 10 + n)
    p1, p2, u = rng.standard_normal((3, m, n))
    d = L.DivergenceIm(L.M1.of(p1), L.M1.of(p2)).tolist()
    np.testing.assert_allclose(o.DivergenceIm(p1, p2), d, rtol=0, atol=1e-15)
    gx, gy = L.GradientIm(L.M1.of(u))
    ox, oy = o.GradientIm(u)
    np.testing.assert_array_equal(ox, gx.tolist())
    np.testing.assert_array_equal(oy, gy.tolist())


@pytest.mark.parametrize("m,n,K", [(5, 4, 1), (5, 4, 4), (4, 7, 3), (6, 6, 7)])
def test_chambolle_matches(m, n, K):
    g = _img(m, n, 3)
    f, px, py, k, err = L.chambolle_prox_TV_stop(L.M1.of(g), 7.5, K)
    of, opx, opy, ok, oerr = o.chambolle_prox_TV_stop(g, lam=7.5, maxiter=K, return_info=True)
    assert k == ok == K
    assert oerr == pytest.approx(err, rel=1e-13)
    np.testing.assert_allclose(opx, px.tolist(), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(opy, py.tolist(), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(of, f.tolist(), rtol=1e-13, atol=1e-12)


def test_chambolle_early_exit_matches():
    g = np.zeros((4, 5))
    f, px, py, k, err = L.chambolle_prox_TV_stop(L.M1.of(g), 2.0, 9)
    of, opx, opy, ok, oerr = o.chambolle_prox_TV_stop(g, lam=2.0, maxiter=9, return_info=True)
    assert k == ok == 1 and err == oerr == 0.0


@pytest.mark.parametrize("m,n", [(5, 4), (3, 8), (6, 6)])
def test_tvnorm_and_conv2c_match(m, n):
    x = _img(m, n, 5)
    assert o.TVnorm(x) == pytest.approx(L.TVnorm(L.M1.of(x)), rel=1e-14)
    for h in ([[0, 1, -1]], [[0], [1], [-1]], [[1, 2, 3], [4, 5, 6], [7, 8, 9.5]]):
        ref = L.conv2c(L.M1.of(x), L.M1.of(h)).tolist()
        np.testing.assert_allclose(o.conv2c(x, np.array(h, dtype=float)), ref, rtol=1e-14, atol=1e-12)
    np.testing.assert_allclose(o.diffh(x), L.conv2c(L.M1.of(x), L.M1.of([[0, 1, -1]])).tolist(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(o.diffv(x), L.conv2c(L.M1.of(x), L.M1.of([[0], [1], [-1]])).tolist(), rtol=0, atol=1e-12)


def test_gaussian_psf_and_operator_match():
    for w1, w2, phi in ((0.4, 0.3, 0.0), (0.7, 0.2, 0.6)):
        k = L.Gaussian_psf(7, w1, w2, phi)
        np.testing.assert_allclose(o.Gaussian_psf(7, w1, w2, phi), k.tolist(), rtol=1e-14)
    x = _img(16, 32, 8)
    k = L.Gaussian_psf(7, 0.4, 0.3, 0.0)
    model = o.BlurModel("gaussian", x.shape)
    ax = model.A(x, 0.4, 0.3)
    np.testing.assert_allclose(ax, L.A_spatial(L.M1.of(x), k).tolist(), rtol=1e-12, atol=1e-10)
    # AT is the adjoint of that circular convolution
    z = _img(16, 32, 9)
    assert float(np.sum(ax * z)) == pytest.approx(float(np.sum(x * model.AT(z, 0.4, 0.3))), rel=1e-12)
