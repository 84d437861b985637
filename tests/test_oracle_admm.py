"""CPU tests of the oracle's C-SALSA / CoRAL restatements and of the host-side diagnostics
(f-4).  The reference holds no outputs for either ("parity unpinned"); these pin them by properties the
algorithms must have: the constrained solver lands on the epsilon sphere, the compound solver with
tau1 + tau2 = tau agrees with SALSA_v2(tau), and hand-checkable first iterations."""
import os
import sys

import numpy as np
import pytest

import sbtv_oracle as o
from conftest import PKG, synth_image


def _setup(n=32, seed=2):
    x = synth_image(n, n, seed)
    rng = np.random.default_rng(seed)
    st = o.demo_setup("gaussian", x, rng.standard_normal(x.shape), evMax=1.0, BSNR=30.0, true_params=(0.4, 0.3))
    m, p = st["model"], st["p_true"]
    H2 = np.abs(m.H_FFT(*p)) ** 2
    A = lambda z: m.A(z, *p)
    AT = lambda z: m.AT(z, *p)
    invLS = lambda r, mu: np.real(o.ifft2(o.fft2(r) / (H2 + mu)))
    return st, A, AT, invLS


def test_csalsa_first_iteration_by_hand():
    st, A, AT, invLS = _setup()
    y = st["y"]
    r = o.CSALSA_v2(y, A, 0.7, 0.3, st["sigma"], AT=AT, invLS=invLS, maxiter=2, TViters=3)
    # outer = 2 from x = u = bu = v = bv = 0 (CSALSA_v2.m:404-408,467-471): x = invLS(mu2*AT(y), mu1)
    x1 = invLS(0.3 * AT(y), 0.7)
    np.testing.assert_allclose(r["x"], x1, rtol=0, atol=1e-12)
    u1, _, _ = o.chambolle_prox_TV_stop(x1, lam=1 / 0.7, maxiter=3)
    np.testing.assert_allclose(r["u"], u1, rtol=0, atol=1e-12)
    ve = A(x1) - y
    eps = np.sqrt(y.size + 8 * np.sqrt(y.size)) * st["sigma"]
    assert r["epsilon"] == pytest.approx(eps, rel=1e-15)
    v1 = ve if np.linalg.norm(ve) <= eps else ve / np.linalg.norm(ve) * eps
    np.testing.assert_allclose(r["v"], v1, rtol=0, atol=1e-12)
    assert r["n_outer"] == 2 and (r["numA"], r["numAt"]) == (3, 2)
    assert len(r["objective"]) == 2 and r["objective"][0] == 0.0                # TV of the zero image
    assert r["criterion"][0] == pytest.approx(np.linalg.norm(y), rel=1e-14)     # ||A 0 - y||


def test_csalsa_lands_on_the_epsilon_sphere():
    st, A, AT, invLS = _setup()
    r = o.CSALSA_v2(st["y"], A, 1.0, 1.0, st["sigma"], AT=AT, invLS=invLS, true_x=st["x"], stopcriterion=3,
                    tolA=1e-5, maxiter=400, TViters=10)
    # the residual norm approaches epsilon (from above here, so the `criterion <= epsilon` stop never fires)
    assert abs(r["criterion"][-1] / r["epsilon"] - 1) < 1e-3
    assert o.PSNR(st["x"], r["x"]) > o.PSNR(st["x"], st["y"])
    assert r["mses"][-1] < r["mses"][0]


def test_coral_initial_multipliers_and_first_iteration():
    st, A, AT, invLS = _setup()
    y = st["y"]
    r = o.CoRAL_v2(y, A, 2.0, 1.0, mu1=0.2, mu2=0.1, AT=AT, invLS=lambda z: invLS(z, 0.3), maxiter=1,
                   initialization=2)
    # bu = u = x0 and bv = v = x0 (CoRAL_v2.m:353-359): the first prox inputs are zero images, so u = v = 0 and
    # x1 = invLS(ATy + mu1*x0 + mu2*x0)
    x0 = AT(y)
    np.testing.assert_array_equal(r["u"], np.zeros_like(y))
    np.testing.assert_array_equal(r["v"], np.zeros_like(y))
    np.testing.assert_allclose(r["x"], invLS(x0 + 0.2 * x0 + 0.1 * x0, 0.3), rtol=0, atol=1e-10)
    assert (r["numA"], r["numAt"]) == (2, 2)
    resid = y - A(x0)
    assert r["objective"][0] == pytest.approx(0.5 * np.sum(resid ** 2) + 3.0 * o.TVnorm(x0), rel=1e-14)


def test_coral_with_split_tau_agrees_with_salsa():
    st, A, AT, invLS = _setup()
    tau, mu = 0.03 * st["sigma"] ** 2, 0.003
    s = o.SALSA_v2(st["y"], A, tau, mu=mu, AT=AT, invLS=lambda z: invLS(z, mu), tolA=1e-9, maxiter=1500, TViters=10)
    c = o.CoRAL_v2(st["y"], A, tau / 2, tau / 2, mu1=mu / 2, mu2=mu / 2, AT=AT, invLS=lambda z: invLS(z, mu),
                   tolA=1e-9, maxiter=1500, TViters1=10, TViters2=10)
    assert o.PSNR(s["x"], c["x"]) > 50.0
    assert c["objective"][-1] == pytest.approx(s["objective"][-1], rel=1e-3)


def test_admm_option_errors():
    st, A, AT, invLS = _setup(16)
    with pytest.raises(ValueError, match="Unknown stopping criterion"):
        o.CSALSA_v2(st["y"], A, 1, 1, 1, AT=AT, invLS=invLS, stopcriterion=4)
    with pytest.raises(ValueError, match="transpose of A is missing"):
        o.CoRAL_v2(st["y"], A, 1, 1, invLS=lambda z: z)
    with pytest.raises(ValueError, match="Unknown 'Initialization' option"):
        o.CoRAL_v2(st["y"], A, 1, 1, AT=AT, invLS=lambda z: z, initialization=1)


# ---------------------------------------------------------------------------------------------
# f-4 diagnostics (host NumPy code of the package; no GPU, no library needed)
def _diagnostics():
    import importlib.util
    spec = importlib.util.spec_from_file_location("sbtv_diagnostics", os.path.join(PKG, "sbtv", "diagnostics.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_ssim_properties():
    d = _diagnostics()
    x = synth_image(64, 48, 3) / 255.0
    rng = np.random.default_rng(0)
    assert d.ssim(x, x) == pytest.approx(1.0, abs=1e-12)
    n1 = x + 0.02 * rng.standard_normal(x.shape)
    n2 = x + 0.10 * rng.standard_normal(x.shape)
    s1, s2 = d.ssim(n1, x), d.ssim(n2, x)
    assert 0 < s2 < s1 < 1
    assert d.ssim(x, n1) == pytest.approx(s1, rel=1e-12)                        # symmetric
    val, smap = d.ssim(n1, x, return_map=True)
    assert smap.shape == x.shape and val == pytest.approx(float(smap.mean()))
    # a constant image against itself plus an offset: only the luminance term differs, closed form
    a, b = np.full((32, 32), 0.5), np.full((32, 32), 0.6)
    C1 = 0.01 ** 2
    assert d.ssim(a, b) == pytest.approx((2 * 0.5 * 0.6 + C1) / (0.25 + 0.36 + C1), rel=1e-12)
    with pytest.raises(ValueError):
        d.ssim(x, x[:-1])


def test_results_round_trip(tmp_path):
    d = _diagnostics()
    res = dict(theta_EB=0.0123, thetas=np.arange(5.0), xMAP=np.ones((4, 3)), options=dict(a=1))
    p = d.save_results(str(tmp_path / "res.npz"), res, psnr=30.5)
    back = d.load_results(p)
    assert back["theta_EB"] == 0.0123 and back["psnr"] == 30.5
    np.testing.assert_array_equal(back["thetas"], res["thetas"])
    np.testing.assert_array_equal(back["xMAP"], res["xMAP"])
    assert "options" not in back
    # .mat (what SALSA/salsa_m.m:346 `save`s): MATLAB v5 file with the reference's field names as variables
    pm = d.save_results(str(tmp_path / "res.mat"), res, psnr=30.5)
    from scipy.io import loadmat
    raw = loadmat(pm)
    assert raw["thetas"].shape == (1, 5) and raw["xMAP"].shape == (4, 3)      # traces are row vectors in MATLAB
    backm = d.load_results(pm)
    assert backm["theta_EB"] == 0.0123 and backm["psnr"] == 30.5
    np.testing.assert_array_equal(backm["thetas"], res["thetas"])
    np.testing.assert_array_equal(backm["xMAP"], res["xMAP"])


def test_trace_plot_and_image_files(tmp_path):
    d = _diagnostics()
    res = dict(thetas=0.01 + 0.03 * (1 - np.exp(-np.arange(5000) / 800.0)), theta_EB=0.0398,
               w1s=np.full(5000, 0.4), w1_EB=0.4, sigmas=np.linspace(2.5, 1.9, 5000), sigma_EB=1.93)
    p = d.plot_traces(str(tmp_path / "traces.svg"), res, true_values=dict(w1s=0.4, sigmas=1.9776))
    txt = open(p).read()
    assert txt.startswith("<svg") and txt.count("<polyline") == 3 and txt.count('stroke="red"') == 2
    assert "EB estimate 0.0398" in txt
    img = np.arange(12.0).reshape(3, 4)
    q = d.save_image(str(tmp_path / "x.pgm"), img)
    raw = open(q, "rb").read()
    assert raw.startswith(b"P5\n4 3\n255\n") and raw[-1] == 255 and raw[len(b"P5\n4 3\n255\n")] == 0
