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


def test_csalsa_spectral_form_matches_oracle_at_the_pipelined_size(ctx, man512):
    """1024 x 1024 runs the pipelined row kernel with OP_CSALSA (the constraint split as a spectrum, csrc/admm.hip); 20
    iterations against the oracle's image-domain restatement, projection active (epsilon below ||Ax - y|| at the start)."""
    import sbtv
    import sbtv_oracle as o
    x = np.tile(man512, (2, 2))
    st = _setup(x, seed=5)
    A, AT, invLS = _oracle_handles(st)
    mu1, mu2, K = 0.4, 0.7, 20
    ref = o.CSALSA_v2(st["y"], A, mu1, mu2, st["sigma"], AT=AT, invLS=invLS, true_x=x, stopcriterion=3, tolA=1e-9,
                      maxiter=K, TViters=5, initialization=2)
    op = sbtv.BlurOperator(sbtv.psf_family("gaussian", 7, st["p_true"])[0])
    got = sbtv.csalsa(st["y"], op, mu1, mu2, st["sigma"], "AT", op.T, "LS", op.invLS, "TVINITIALIZATION", 1,
                      "TVITERS", 5, "STOPCRITERION", 3, "TOLERANCEA", 1e-9, "MAXITERA", K, "TRUE_X", x,
                      "INITIALIZATION", 2, "VERBOSE", 0)
    xg, numA, numAt, objective, d1, d2, crit, times, mses = got
    assert len(objective) == ref["n_outer"] == K
    assert np.any(np.asarray(ref["criterion"]) > ref["epsilon"]) and np.any(np.asarray(ref["criterion"]) <= ref["epsilon"] * 1.2)
    np.testing.assert_allclose(objective, ref["objective"], rtol=1e-9)
    np.testing.assert_allclose(crit, ref["criterion"], rtol=1e-8)
    np.testing.assert_allclose(d1, ref["distance1"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(d2, ref["distance2"], rtol=1e-8)
    np.testing.assert_allclose(mses, ref["mses"], rtol=1e-9)
    assert np.max(np.abs(xg - ref["x"])) < 1e-7


CS_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "semi-blind-image-deblurring-problems-with-tv_amd"))
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import sbtv
from conftest import synth_image
res = {}
for tag, (M, N) in (("a", (2048, 2048)), ("b", (256, 192)), ("c", (100, 90))):
    x = synth_image(M, N, 8)
    st = sbtv.demo_setup("gaussian", x, np.random.default_rng(2).standard_normal(x.shape), evMax=1.0)
    op = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
    got = sbtv.csalsa(st["y"], op, 0.5, 0.5, st["sigma"], "AT", op.T, "LS", op.invLS, "TVINITIALIZATION", 1, "TVITERS", 10,
                      "STOPCRITERION", 2, "TOLERANCEA", 1e-9, "MAXITERA", 15, "TRUE_X", x, "VERBOSE", 0)
    for nm, v in zip(("x", "obj", "d1", "d2", "crit", "mses"), (got[0], got[3], got[4], got[5], got[6], got[8])):
        res[tag + "_" + nm] = np.asarray(v)
np.savez(sys.argv[1], **res)
"""


def test_csalsa_spectral_and_image_forms_agree(tmp_path):
    """The default (spectral constraint split, one FFT triple per iteration) against SBTV_CSALSA_SPECTRAL=0 (v, bv, Ax as
    images, three FFT triples) at 2048^2 (pipelined row kernel), 256 x 192 (workgroup row kernel) and 100 x 90 (chirp-z
    path: both runs take the image form there)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(name, env):
        out = str(tmp_path / (name + ".npz"))
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, "-c", CS_CHILD % {"root": root}, out], check=True, env=e, timeout=900)
        return np.load(out)
    a, b = run("spectral", {}), run("images", {"SBTV_CSALSA_SPECTRAL": "0"})
    for key in a.files:
        if key.endswith("_x"):
            assert np.max(np.abs(a[key] - b[key])) < 1e-8, key
        elif key.endswith("_d1"):
            np.testing.assert_allclose(a[key], b[key], rtol=1e-6, atol=1e-9, err_msg=key)
        else:
            np.testing.assert_allclose(a[key], b[key], rtol=1e-9, err_msg=key)
    np.testing.assert_array_equal(a["c_x"], b["c_x"])


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


def test_coral_batched_proxes_match_oracle_at_1024(ctx, man512):
    """Equal TViters: the two proxes of an outer iteration run as one batch of two images with their own thresholds
    (tau1/mu1 != tau2/mu2), optimistic launches from the second iteration on; TV(u), TV(v) come from the pass that forms s."""
    import sbtv
    import sbtv_oracle as o
    x = np.tile(man512, (2, 2))
    st = _setup(x, seed=7)
    A, AT, invLS2 = _oracle_handles(st)
    theta, s2 = 0.03, st["sigma"] ** 2
    tau1, tau2 = 0.7 * theta * s2, 0.3 * theta * s2
    mu1, mu2, K = theta / 10, theta / 25, 12
    ref = o.CoRAL_v2(st["y"], A, tau1, tau2, mu1=mu1, mu2=mu2, AT=AT, invLS=lambda r: invLS2(r, mu1 + mu2), true_x=x,
                     stopcriterion=1, tolA=0.0, maxiter=K, TViters1=10, TViters2=10, initialization=2)
    op = sbtv.BlurOperator(sbtv.psf_family("gaussian", 7, st["p_true"])[0])
    got = sbtv.CoRAL(st["y"], op, tau1, tau2, "MU1", mu1, "MU2", mu2, "AT", op.T, "LS", op.LS(mu1 + mu2),
                     "TVINITIALIZATION1", 1, "TVITERS1", 10, "TVINITIALIZATION2", 1, "TVITERS2", 10, "STOPCRITERION", 1,
                     "TOLERANCEA", 0.0, "MAXITERA", K, "TRUE_X", x, "INITIALIZATION", 2, "VERBOSE", 0)
    xg, numA, numAt, objective, distance, times, mses = got
    assert len(objective) == len(ref["objective"]) == K + 1
    assert (numA, numAt) == (ref["numA"], ref["numAt"])
    np.testing.assert_allclose(objective, ref["objective"], rtol=1e-9)
    np.testing.assert_allclose(mses, ref["mses"], rtol=1e-9)
    np.testing.assert_allclose(distance, ref["distance"], rtol=1e-7)
    assert np.max(np.abs(xg - ref["x"])) < 1e-7


CORAL_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "semi-blind-image-deblurring-problems-with-tv_amd"))
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import sbtv
from conftest import synth_image
res = {}
for tag, (M, N) in (("a", (2048, 1024)), ("b", (192, 256)), ("c", (100, 90))):
    x = synth_image(M, N, 11)
    st = sbtv.demo_setup("gaussian", x, np.random.default_rng(4).standard_normal(x.shape), evMax=1.0)
    op = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
    theta, s2 = 0.03, st["sigma"] ** 2
    got = sbtv.CoRAL(st["y"], op, 0.6 * theta * s2, 0.4 * theta * s2, "MU1", theta / 10, "MU2", theta / 20, "AT", op.T,
                     "LS", op.LS(theta / 10 + theta / 20), "TVINITIALIZATION1", 1, "TVITERS1", 10, "TVINITIALIZATION2", 1,
                     "TVITERS2", 10, "STOPCRITERION", 2, "TOLERANCEA", 0.0, "MAXITERA", 12, "TRUE_X", x, "VERBOSE", 0)
    for nm, v in zip(("x", "obj", "dist", "mses"), (got[0], got[3], got[4], got[6])):
        res[tag + "_" + nm] = np.asarray(v)
np.savez(sys.argv[1], **res)
"""


def test_coral_batched_and_separate_proxes_agree(tmp_path):
    """One batch of two images (default when TViters1 == TViters2) against one plan per prox (SBTV_CORAL_BATCH=0): the prox of
    an image does not depend on what else is in its batch, so the runs agree to the last bit - at 2048 x 1024 (wave-granular
    FFT kernels), 192 x 256 and 100 x 90 (chirp-z path)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(name, env):
        out = str(tmp_path / (name + ".npz"))
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, "-c", CORAL_CHILD % {"root": root}, out], check=True, env=e, timeout=900)
        return np.load(out)
    a, b = run("batched", {}), run("separate", {"SBTV_CORAL_BATCH": "0"})
    for key in a.files:
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)


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
