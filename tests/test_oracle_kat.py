"""CPU tests: pin the oracle (oracle/sbtv_oracle.py) with the analytic /
hand-checkable known-answer tests of SURVEY.md §8c.  The reference ships no
tests or golden vectors (MATLAB only), so these KATs are what anchors parity.
"""
import math

import numpy as np
import pytest

import sbtv_oracle as o

G3 = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 10.0]])


def test_chambolle_3x3_hand_case():
    # one iteration from p = 0: u = -g/lambda, err^2 = sum |grad u|^2 = 17.5 (worked by hand)
    f, px, py, k, err = o.chambolle_prox_TV_stop(G3, lam=2, maxiter=1, return_info=True)
    assert k == 1
    assert err == pytest.approx(math.sqrt(17.5), rel=1e-15)
    assert err == pytest.approx(4.183300132670378, rel=1e-15)
    assert f[0, 0] == pytest.approx(1.714642641645, abs=1e-11)
    assert px[0, 0] == pytest.approx(-0.267990990617, abs=1e-11)
    assert py[2, 1] == pytest.approx(-0.199359487590, abs=1e-11)
    # hand check of px(0,0): upx = -1.5, upy = -0.5, tmp = sqrt(2.5); px = tau*upx/(1+tau*tmp)
    t = math.sqrt(2.5)
    assert px[0, 0] == pytest.approx(0.249 * -1.5 / (1 + 0.249 * t), rel=1e-15)


def test_chambolle_3x3_three_iterations():
    errs = []
    for K in (1, 2, 3):
        f, _, _, k, err = o.chambolle_prox_TV_stop(G3, lam=2, maxiter=K, return_info=True)
        errs.append(err)
    assert errs == pytest.approx([4.183300132670, 2.791819209734, 1.943468133580], abs=1e-11)
    assert f[0, 0] == pytest.approx(2.464881599466, abs=1e-11)


def test_tvnorm_periodic_kat():
    assert o.TVnorm(G3) == pytest.approx(41.28554656058845, rel=1e-15)
    # diffh/diffv are periodic backward differences (conv2c with [0 1 -1])
    x = np.random.default_rng(0).standard_normal((6, 5))
    assert np.allclose(o.diffh(x), o.conv2c(x, np.array([[0, 1, -1.0]])))
    assert np.allclose(o.diffv(x), o.conv2c(x, np.array([[0, 1, -1.0]]).T))
    assert np.allclose(o.diffh(x), x - np.roll(x, 1, axis=1))


def test_chambolle_structural_properties():
    rng = np.random.default_rng(1)
    g = rng.uniform(0, 255, (17, 12))
    f, px, py, k, err = o.chambolle_prox_TV_stop(g, lam=7.0, maxiter=25, return_info=True)
    assert k == 25
    assert np.all(np.sqrt(px ** 2 + py ** 2) <= 1 + 1e-12)      # |p| <= 1
    assert np.all(px[-1, :] == 0) and np.all(py[:, -1] == 0)     # last row / col of the duals stay 0
    assert f[-1, -1] == g[-1, -1]                                 # quirk Q3: corner untouched
    # constant image: f = g, p = 0, err = 0 -> exits at k = 1
    c = np.full((8, 8), 3.5)
    f, px, py, k, err = o.chambolle_prox_TV_stop(c, lam=2.0, maxiter=10, return_info=True)
    assert k == 1 and err == 0 and np.array_equal(f, c) and not px.any() and not py.any()
    # g == 0 exits at k = 1
    _, _, _, k, _ = o.chambolle_prox_TV_stop(np.zeros((8, 8)), lam=2.0, maxiter=10, return_info=True)
    assert k == 1


def test_chambolle_quirks():
    with pytest.raises(NameError):                 # Q1: maxiter required
        o.chambolle_prox_TV_stop(G3, lam=1.0)
    with pytest.raises(ValueError):                # dual size check (:103)
        o.chambolle_prox_TV_stop(G3, lam=1.0, maxiter=2, dualvars=np.zeros((3, 5)))
    # divergence is NOT the adjoint of the gradient (Q3)
    rng = np.random.default_rng(2)
    p1, p2, u = rng.standard_normal((3, 5, 6))
    gx, gy = o.GradientIm(u)
    lhs = np.sum(gx * p1 + gy * p2)
    rhs = -np.sum(u * o.DivergenceIm(p1, p2))
    assert abs(lhs - rhs) > 1e-3


@pytest.mark.parametrize("kind,params", [("gaussian", (0.4, 0.3)), ("moffat", (0.4, 3.5)), ("laplace", (0.3,))])
def test_operator_kats(kind, params):
    M, N = 32, 48
    m = o.BlurModel(kind, (M, N))
    taps = m.taps(*params)
    assert taps.shape == (7, 7) and np.sum(taps) == pytest.approx(1.0, rel=1e-15)
    H = m.H_FFT(*params)
    assert H[0, 0] == pytest.approx(1.0) and np.max(np.abs(H)) == pytest.approx(1.0)
    d = np.zeros((M, N)); d[0, 0] = 1
    assert np.allclose(m.A(d, *params)[:7, :7], taps, atol=1e-15)          # A(delta) = taps at TOP-LEFT (Q6)
    assert np.allclose(m.A(np.full((M, N), 2.5), *params), 2.5)             # A(const) = const
    rng = np.random.default_rng(3)
    x, z = rng.standard_normal((2, M, N))
    assert np.sum(m.A(x, *params) * z) == pytest.approx(np.sum(x * m.AT(z, *params)), rel=1e-12)   # adjoint
    # FFT operator == spatial circular 7x7 sum  (A x)(i,j) = sum h(m,n) x(i-m, j-n)
    ref = np.zeros((M, N))
    for a in range(7):
        for b in range(7):
            ref += taps[a, b] * np.roll(np.roll(x, a, axis=0), b, axis=1)
    assert np.allclose(m.A(x, *params), ref, atol=1e-13)
    # PSF-derivative taps vs central finite differences
    for i in range(len(params)):
        hstep = 1e-6
        pp = list(params); pm = list(params)
        pp[i] += hstep; pm[i] -= hstep
        fd = (m.taps(*pp) - m.taps(*pm)) / (2 * hstep)
        # quirk Q10: utils/diff_moffat_alpha.m:17 carries an extra factor 2 in a denominator; after the
        # quotient rule the reference's d/d(alpha) is exactly HALF the true derivative.  We reproduce it.
        scale = 0.5 if (kind == "moffat" and i == 0) else 1.0
        assert np.allclose(m.dtaps(i, *params), scale * fd, atol=1e-8)


def test_gaussian_axes_convention():
    # w1 scales the COLUMN axis, w2 the ROW axis (ndgrid order)
    k = o.Gaussian_psf(7, 0.9, 0.2)
    assert k[3, 0] < k[0, 3]           # faster decay along columns (w1 larger)
    assert np.allclose(k, k[::-1, :]) and np.allclose(k, k[:, ::-1])


def test_salsa_identity_psf_tiny_tau():
    # identity PSF (single tap at (1,1)), tiny tau -> x -> y
    rng = np.random.default_rng(4)
    y = rng.uniform(0, 255, (16, 16))
    A = lambda v: v.copy()
    mu = 0.05
    invLS = lambda v: v / (1 + mu)
    out = o.SALSA_v2(y, A, 1e-9, mu=mu, AT=A, invLS=invLS, tolA=1e-12, maxiter=400, TViters=5)
    assert np.max(np.abs(out["x"] - y)) < 1e-6
    assert out["numAt"] == 1 and out["numA"] == out["n_outer"] + 1


def test_metrics():
    x = np.array([[0, 10.0], [20, 30]])
    y = x + 1
    assert o.PSNR(x, y) == pytest.approx(10 * math.log10(900) - 0.0)
    assert o.MSE(x, y) == pytest.approx(0.0)
    assert o.l2(np.eye(2) * 3, np.zeros((2, 2))) == pytest.approx(9.0)      # spectral norm (Q9)


def test_demo_setup_and_sapg_smoke():
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 255, (16, 16))
    noise = rng.standard_normal((16, 16))
    st = o.demo_setup("laplace", x, noise, evMax=0.99)
    # Laplace demo: Lf = max(...), lambdaMax = 0.1, gamma x10 (quirk Q8)
    assert st["Lf"] == pytest.approx(0.99 ** 2 / min(st["sigma_min"], st["sigma_max"]))
    assert st["lam"] <= 0.1
    res = o.SAPG_algorithm(st, samples=4, warmup=3, burnIn=2, randn=lambda s: rng.standard_normal(s), chambolleit=3)
    assert res["thetas"].shape == (4,) and np.all(res["thetas"] >= 1e-3)
    assert np.all(res["Xlast_sample"] >= 0)      # abs() in the MYULA step
