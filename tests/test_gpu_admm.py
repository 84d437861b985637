"""GPU parity of the two other TV front-ends: csalsa (SALSA/CSALSA_v2.m) and CoRAL
(SALSA/CoRAL_v2.m) through the C-ABI vs the oracle's op-for-op restatements.  Same bar as SALSA_v2: same
stopping iteration, PSNR within 1e-3 dB, traces to float tolerance."""
import numpy as np
import pytest

from conftest import synth_image

pytestmark = pytest.mark.gpu

PSNR_TOL_DB = 1e-3


def _setup(x, params=(0.4, 0.3), seed=3):
    import sbtv_oracle as o
    rng = np.random.default_rng(seed)
    return o.demo_setup("gaussian", x, rng.standard_normal(x.shape), evMax=1.0, BSNR=30.0, true_params=params)


def _oracle_handles(st):
    import sbtv_oracle as o
    m, p = st["model"], st["p_true"]
    H = m.H_FFT(*p)
    H2 = np.abs(H) ** 2
    A = lambda z: m.A(z, *p)
    AT = lambda z: m.AT(z, *p)
    invLS = lambda r, mu: np.real(o.ifft2(o.fft2(r) / (H2 + mu)))
    return A, AT, invLS


@pytest.mark.parametrize("stop,init,delta", [(3, 0, 1.0), (1, 2, 1.0), (2, 0, 1.02)])
def test_csalsa_matches_oracle(ctx, stop, init, delta):
    import sbtv
    import sbtv_oracle as o
    x = synth_image(128, 128, 4)
    st = _setup(x)
    A, AT, invLS = _oracle_handles(st)
    mu1, mu2, K = 0.5, 0.5, 40
    ref = o.CSALSA_v2(st["y"], A, mu1, mu2, st["sigma"], AT=AT, invLS=invLS, true_x=x, stopcriterion=stop, tolA=1e-3,
                      maxiter=K, TViters=5, initialization=init, continuationfactor=delta)
    op = sbtv.BlurOperator(sbtv.psf_family("gaussian", 7, st["p_true"])[0])
    got = sbtv.csalsa(st["y"], op, mu1, mu2, st["sigma"], "AT", op.T, "LS", op.invLS, "TVINITIALIZATION", 1,
                      "TVITERS", 5, "STOPCRITERION", stop, "TOLERANCEA", 1e-3, "MAXITERA", K, "TRUE_X", x,
                      "INITIALIZATION", init, "CONTINUATIONFACTOR", delta, "VERBOSE", 0)
    xg, numA, numAt, objective, d1, d2, crit, times, mses = got
    assert len(objective) == ref["n_outer"] == len(ref["objective"]), "different stopping iteration"
    assert (numA, numAt) == (ref["numA"], ref["numAt"])
    np.testing.assert_allclose(objective, ref["objective"], rtol=1e-9)
    np.testing.assert_allclose(crit, ref["criterion"], rtol=1e-8)
    np.testing.assert_allclose(d1, ref["distance1"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(d2, ref["distance2"], rtol=1e-8)
    np.testing.assert_allclose(mses, ref["mses"], rtol=1e-9)
    assert abs(o.PSNR(x, xg) - o.PSNR(x, ref["x"])) <= PSNR_TOL_DB
    assert np.max(np.abs(xg - ref["x"])) < 1e-7
    assert times[0] == 0 and np.all(np.diff(times) >= 0)


def test_csalsa_reaches_the_constraint_set(ctx, cman256):
    """Domain property at a larger size: the iterates approach ||Ax - y|| = epsilon and improve on y."""
    import sbtv
    import sbtv_oracle as o
    st = _setup(cman256, (1 / 1.6, 1 / 1.6))
    op = sbtv.BlurOperator(sbtv.psf_family("gaussian", 7, st["p_true"])[0])
    got = sbtv.csalsa(st["y"], op, 1.0, 1.0, st["sigma"], "AT", op.T, "LS", op.invLS, "TVINITIALIZATION", 1,
                      "TVITERS", 10, "STOPCRITERION", 3, "TOLERANCEA", 1e-4, "MAXITERA", 300, "VERBOSE", 0)
    xg, crit = got[0], got[6]
    eps = np.sqrt(cman256.size + 8 * np.sqrt(cman256.size)) * st["sigma"]
    assert 0.9 * eps < crit[-1] < eps * (1 + 5e-3)     # feasible (the stop rule needs criterion <= epsilon) and near the sphere
    assert o.PSNR(cman256, xg) > o.PSNR(cman256, st["y"]) + 1.0


@pytest.mark.parametrize("stop,init,k2", [(1, 0, 5), (2, 2, 3), (3, 0, 5)])
def test_coral_matches_oracle(ctx, stop, init, k2):
    import sbtv
    import sbtv_oracle as o
    x = synth_image(64, 64, 5)                      # square: the warm-start dual split needs it (quirk Q2)
    st = _setup(x)
    A, AT, invLS2 = _oracle_handles(st)
    theta, s2 = 0.03, st["sigma"] ** 2
    tau1, tau2 = 0.6 * theta * s2, 0.4 * theta * s2
    mu1, mu2, K = theta / 10, theta / 20, 50
    tolA = 1e-4 if stop != 3 else 0.0
    ref = o.CoRAL_v2(st["y"], A, tau1, tau2, mu1=mu1, mu2=mu2, AT=AT, invLS=lambda r: invLS2(r, mu1 + mu2), true_x=x,
                     stopcriterion=stop, tolA=tolA, maxiter=K, TViters1=5, TViters2=k2, initialization=init)
    op = sbtv.BlurOperator(sbtv.psf_family("gaussian", 7, st["p_true"])[0])
    got = sbtv.CoRAL(st["y"], op, tau1, tau2, "MU1", mu1, "MU2", mu2, "AT", op.T, "LS", op.LS(mu1 + mu2),
                     "TVINITIALIZATION1", 1, "TVITERS1", 5, "TVINITIALIZATION2", 1, "TVITERS2", k2, "STOPCRITERION", stop,
                     "TOLERANCEA", tolA, "MAXITERA", K, "TRUE_X", x, "INITIALIZATION", init, "VERBOSE", 0)
    xg, numA, numAt, objective, distance, times, mses = got
    assert len(objective) == len(ref["objective"]), "different stopping iteration"
    assert (numA, numAt) == (ref["numA"], ref["numAt"])
    np.testing.assert_allclose(objective, ref["objective"], rtol=1e-9)
    np.testing.assert_allclose(mses, ref["mses"], rtol=1e-9)
    np.testing.assert_allclose(distance, ref["distance"], rtol=1e-7)
    assert abs(o.PSNR(x, xg) - o.PSNR(x, ref["x"])) <= PSNR_TOL_DB
    assert np.max(np.abs(xg - ref["x"])) < 1e-7


def test_coral_split_equals_salsa_fixed_point(ctx, cman256):
    """tau1 + tau2 = tau poses the same problem as SALSA_v2 with tau: both front-ends land on (nearly) the same image."""
    import sbtv
    import sbtv_oracle as o
    st = _setup(cman256, (1 / 1.6, 1 / 1.6))
    theta, s2 = 0.03, st["sigma"] ** 2
    tau, mu = theta * s2, theta / 10
    op = sbtv.BlurOperator(sbtv.psf_family("gaussian", 7, st["p_true"])[0])
    xs = sbtv.SALSA_v2(st["y"], op, tau, "MU", mu, "AT", op.T, "LS", op.LS(mu), "TVINITIALIZATION", 1, "TVITERS", 10,
                       "TOLERANCEA", 1e-7, "MAXITERA", 1500, "VERBOSE", 0)[0]
    xc = sbtv.CoRAL(st["y"], op, tau / 2, tau / 2, "MU1", mu / 2, "MU2", mu / 2, "AT", op.T, "LS", op.LS(mu),
                    "TVINITIALIZATION1", 1, "TVITERS1", 10, "TVINITIALIZATION2", 1, "TVITERS2", 10,
                    "TOLERANCEA", 1e-7, "MAXITERA", 1500, "VERBOSE", 0)[0]
    assert abs(o.PSNR(cman256, xs) - o.PSNR(cman256, xc)) < 0.05
    assert o.PSNR(xs, xc) > 45.0


def test_admm_error_paths(ctx):
    import sbtv
    x = synth_image(32, 32, 1)
    op = sbtv.BlurOperator(sbtv.psf_family("gaussian", 7, (0.4, 0.3))[0])
    with pytest.raises(sbtv.SbtvError, match="transpose of A is missing"):
        sbtv.csalsa(x, op, 1.0, 1.0, 1.0, "LS", op.invLS, "TVINITIALIZATION", 1)
    with pytest.raises(sbtv.SbtvError, match="must be specified as a function handle"):
        sbtv.csalsa(x, op, 1.0, 1.0, 1.0, "AT", op.T, "TVINITIALIZATION", 1)
    with pytest.raises(sbtv.SbtvError, match="Unknown stopping criterion"):
        sbtv.csalsa(x, op, 1.0, 1.0, 1.0, "AT", op.T, "LS", op.invLS, "TVINITIALIZATION", 1, "STOPCRITERION", 4)
    with pytest.raises(sbtv.SbtvError, match="must be specified as a function handle"):
        sbtv.CoRAL(x, op, 1.0, 1.0, "AT", op.T, "TVINITIALIZATION1", 1, "TVINITIALIZATION2", 1)
    with pytest.raises(sbtv.SbtvError, match="Unknown 'Initialization' option"):
        sbtv.CoRAL(x, op, 1.0, 1.0, "AT", op.T, "LS", op.LS(0.1), "TVINITIALIZATION1", 1, "TVINITIALIZATION2", 1,
                   "INITIALIZATION", 7)
