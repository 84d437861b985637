"""The committed known-answer file tests/golden/oracle_kat.json (made by tests/golden/make_golden.py from the
oracle on the reference's own cman.png): the oracle must still reproduce it (CPU), and the HIP path must hit the
same numbers through the C-ABI (GPU), independently of the live oracle run in the other parity tests."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def kat():
    return json.load(open(os.path.join(GOLDEN, "oracle_kat.json")))


def test_oracle_reproduces_committed_answers(kat, cman256):
    import sbtv_oracle as o
    c = kat["chambolle_cman256_lambda10_K10"]
    f, px, py, k, err = o.chambolle_prox_TV_stop(cman256, lam=10.0, maxiter=10, return_info=True)
    assert k == c["k"] and err == pytest.approx(c["err"], rel=1e-13)
    assert f.sum() == pytest.approx(c["sum_f"], rel=1e-13) and px.sum() == pytest.approx(c["sum_px"], rel=1e-10)
    np.testing.assert_allclose(f[100:108, 100:108], c["f_crop_8x8"], rtol=1e-13)
    assert o.TVnorm(cman256) == pytest.approx(kat["tvnorm_cman256"], rel=1e-14)


@pytest.mark.gpu
def test_hip_path_hits_committed_answers(kat, cman256, ctx):
    import sbtv
    c = kat["chambolle_cman256_lambda10_K10"]
    f, px, py, k, err = sbtv.chambolle_prox_TV_stop(cman256, "lambda", 10.0, "maxiter", 10, return_info=True)
    assert int(k[0]) == c["k"] and float(err[0]) == pytest.approx(c["err"], rel=1e-11)
    assert f.sum() == pytest.approx(c["sum_f"], rel=1e-12) and py.sum() == pytest.approx(c["sum_py"], rel=1e-9)
    np.testing.assert_allclose(f[100:108, 100:108], c["f_crop_8x8"], rtol=1e-11)
    assert sbtv.TVnorm(cman256) == pytest.approx(kat["tvnorm_cman256"], rel=1e-13)
    # configs[0]: cman 256^2, Gaussian blur sigma 1.6, SALSA_v2 with the demo's settings
    s = kat["salsa_cman256_gaussian_sigma1p6"]
    st = sbtv.demo_setup("gaussian", cman256, np.random.default_rng(1).standard_normal(cman256.shape), evMax=1.0,
                         BSNR=30.0, true_params=(1 / 1.6, 1 / 1.6))
    assert st["sigma"] == pytest.approx(s["sigma"], rel=1e-12)
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 1 / 1.6, 1 / 1.6))
    theta, mu = 0.03, 0.003
    x, numA, numAt, obj, dist, times, mses = sbtv.SALSA_v2(
        st["y"], A, theta * st["sigma"] ** 2, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", cman256, "StopCriterion", 1,
        "ToleranceA", 1e-5, "MAXITERA", 500, "TVINITIALIZATION", 1, "TViters", 10, "VERBOSE", 0)
    assert len(obj) - 1 == s["n_outer"]                                    # same stopping iteration
    assert obj[0] == pytest.approx(s["objective_first"], rel=1e-10) and obj[-1] == pytest.approx(s["objective_last"], rel=1e-9)
    assert abs(sbtv.PSNR(cman256, x) - s["psnr_db"]) <= 1e-3               # the north-star tolerance
    assert mses[-1] == pytest.approx(s["mse_last"], rel=1e-9)
    np.testing.assert_allclose(x[60:64, 60:64], s["x_crop_4x4"], rtol=0, atol=1e-6)
