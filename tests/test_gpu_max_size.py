"""GPU: the loops at the LARGEST supported image size, 4096 x 4096 (include/sbtv.h; the operators alone are covered in
test_gpu_fft.py).
  * TV prox: K Chambolle iterations from a cold start have a dependency radius of K + 1 pixels, so any interior window of the
    4096^2 result must equal the oracle's prox of that window plus a margin - exact locality, no 4096^2 oracle run needed;
    the last rows / columns (quirk Q3, the non-adjoint divergence) the same way with the margin on the inner sides only;
  * SALSA_v2: two outer iterations against the live oracle (objective, mse, crops of x) - ~10 s of oracle time."""
import numpy as np
import pytest

from conftest import synth_image

pytestmark = pytest.mark.gpu
S = 4096


def _big(seed):
    rng = np.random.default_rng(seed)
    base = synth_image(512, 512, seed)
    x = np.tile(base, (8, 8)) + 6.0 * rng.standard_normal((S, S))        # no exact periodicity left
    return np.clip(x, 0, 255)


@pytest.mark.parametrize("K", [10, 25])
def test_prox_at_4096_equals_the_oracle_on_windows(ctx, K):
    import sbtv
    import sbtv_oracle as o
    g = _big(3)
    lam = 7.5
    f, px, py, k, err = sbtv.chambolle_prox_TV_stop(g, "lambda", lam, "maxiter", K, return_info=True, ctx=ctx)
    assert int(np.ravel(k)[0]) == K
    m = K + 2
    # interior windows, among them one across the seams of the 128-row x 21-column tiles and one in the middle
    for i0, j0 in ((1000, 2000), (2040, 2040), (116 * 17 - 20, 21 * 97 - 10), (3000, 70)):
        h = 96
        win = g[i0 - m:i0 + h + m, j0 - m:j0 + h + m]
        rf, rpx, rpy = o.chambolle_prox_TV_stop(win, lam=lam, maxiter=K)[:3]
        np.testing.assert_allclose(f[i0:i0 + h, j0:j0 + h], rf[m:m + h, m:m + h], rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(px[i0:i0 + h, j0:j0 + h], rpx[m:m + h, m:m + h], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(py[i0:i0 + h, j0:j0 + h], rpy[m:m + h, m:m + h], rtol=1e-11, atol=1e-13)
    # the bottom-right corner: the image boundary is the window's boundary there (Q3: last row / column unsmoothed)
    h = 96
    win = g[S - h - m:, S - h - m:]
    rf, rpx, rpy = o.chambolle_prox_TV_stop(win, lam=lam, maxiter=K)[:3]
    np.testing.assert_allclose(f[S - h:, S - h:], rf[m:, m:], rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(px[S - h:, S - h:], rpx[m:, m:], rtol=1e-11, atol=1e-13)
    assert f[S - 1, S - 1] == g[S - 1, S - 1] and np.all(px[S - 1, :] == 0.0) and np.all(py[:, S - 1] == 0.0)
    # and the top-left one
    win = g[:h + m, :h + m]
    rf = o.chambolle_prox_TV_stop(win, lam=lam, maxiter=K)[0]
    np.testing.assert_allclose(f[:h, :h], rf[:h, :h], rtol=1e-12, atol=1e-10)


def test_salsa_at_4096_first_iterations_match_the_oracle(ctx):
    import sbtv
    import sbtv_oracle as o
    x = _big(5)
    st = o.demo_setup("gaussian", x, np.random.default_rng(9).standard_normal(x.shape), evMax=1.0)
    theta, s2 = 0.03, st["sigma"] ** 2
    ref = o.salsa_from_estimates(st, theta, st["p_true"], s2, tol=0.0, outeriters=2, TViters=10)
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3), ctx=ctx)
    mu = theta / 10
    got = sbtv.SALSA_v2(sbtv.to_device(st["y"]), A, theta * s2, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", sbtv.to_device(x),
                        "ToleranceA", -1.0, "MAXITERA", 2, "TVINITIALIZATION", 1, "TViters", 10, ctx=ctx)
    np.testing.assert_allclose(got[3], ref["objective"], rtol=1e-9)
    np.testing.assert_allclose(got[6], ref["mses"], rtol=1e-9)
    xg = sbtv.to_host(got[0])
    for si, sj in ((slice(0, 16), slice(0, 16)), (slice(2040, 2056), slice(2040, 2056)), (slice(S - 16, S), slice(S - 16, S)),
                   (slice(1000, 1016), slice(3000, 3016))):
        np.testing.assert_allclose(xg[si, sj], ref["x"][si, sj], rtol=0, atol=1e-7)
