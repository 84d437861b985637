"""GPU parity of the arbitrary-size blur operator (chirp-z path, csrc/fft_any.inc).  The reference's closures work for
ANY image size (utils/resize.m:1-12 pads the kernel to the image's size; run_Gaussian_demo.m:136-139 are plain
fft2 / ifft2): sizes that are not powers of two take a full complex 2-D DFT by Bluestein's identity."""
import numpy as np
import pytest

from conftest import synth_image

pytestmark = pytest.mark.gpu

SHAPES = [(384, 640), (500, 500), (30, 50), (17, 12), (333, 501), (8, 8), (100, 4096)]


@pytest.mark.parametrize("shape", SHAPES)
def test_operator_closures_any_size_match_oracle(ctx, shape):
    import sbtv
    import sbtv_oracle as o
    M, N = shape
    rng = np.random.default_rng(M * 31 + N)
    x = rng.uniform(0, 255, (M, N))
    for kind, params in (("gaussian", (0.4, 0.3)), ("moffat", (0.4, 3.5))):
        model = o.BlurModel(kind, (M, N))
        taps, dtaps = sbtv.psf_family(kind, 7, params)
        op = sbtv.BlurOperator(taps)
        np.testing.assert_allclose(op.A(x), model.A(x, *params), rtol=0, atol=5e-11)
        np.testing.assert_allclose(op.AT(x), model.AT(x, *params), rtol=0, atol=5e-11)
        np.testing.assert_allclose(sbtv.BlurOperator(dtaps[0]).A(x), model.dA(0, x, *params), rtol=0, atol=5e-10)
        mu = 0.003
        H = model.H_FFT(*params)
        ref = np.real(o.ifft2(o.fft2(x) / (np.abs(H) ** 2 + mu)))
        got = op.invLS(x, mu)
        assert np.max(np.abs(got - ref)) / np.max(np.abs(ref)) < 1e-11
    d = np.zeros((M, N)); d[0, 0] = 1
    np.testing.assert_allclose(op.A(d)[:7, :7], taps, atol=1e-14)          # A(delta) = taps at the top-left (Q6)
    np.testing.assert_allclose(op.A(np.full((M, N), 2.5)), 2.5, atol=1e-12)
    z = rng.standard_normal((M, N))
    assert np.sum(op.A(x) * z) == pytest.approx(np.sum(x * op.AT(z)), rel=1e-10)
    # batch of two images with their own PSF
    xb = rng.uniform(0, 255, (2, M, N))
    t2 = np.stack([sbtv.Gaussian_psf(7, 0.4, 0.3), sbtv.psf_laplace(7, 0.3)])
    got = sbtv.BlurOperator(t2).A(xb)
    for b, (kind, params) in enumerate((("gaussian", (0.4, 0.3)), ("laplace", (0.3,)))):
        np.testing.assert_allclose(got[b], o.BlurModel(kind, (M, N)).A(xb[b], *params), rtol=0, atol=5e-11)


@pytest.mark.parametrize("shape", [(100, 100), (75, 75), (126, 126)])
def test_salsa_any_size_matches_oracle(ctx, shape):
    """SALSA_v2 on non-power-of-two images (even and odd M: the TV prox takes the fused resp. the scalar kernels).
    Square only: the reference's warm start splits 'dualvars' = [px py] with M (quirk Q2, chambolle_prox_TV_stop.m:
    105-107), so SALSA_v2 itself fails on a rectangular image in MATLAB and the oracle restates that."""
    import sbtv
    import sbtv_oracle as o
    M, N = shape
    x = synth_image(M, N, 9)
    st = o.demo_setup("gaussian", x, np.random.default_rng(1).standard_normal((M, N)), evMax=1.0)
    theta, s2 = 0.03, st["sigma"] ** 2
    ref = o.salsa_from_estimates(st, theta, st["p_true"], s2, outeriters=60)
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
    mu = theta / 10
    xg, numA, numAt, obj, dist, times, mses = sbtv.SALSA_v2(
        st["y"], A, theta * s2, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", x, "ToleranceA", 1e-5, "MAXITERA", 60,
        "TVINITIALIZATION", 1, "TViters", 10, "VERBOSE", 0)
    assert len(obj) == len(ref["objective"]), "different stopping iteration"
    np.testing.assert_allclose(obj, ref["objective"], rtol=1e-9)
    np.testing.assert_allclose(mses, ref["mses"], rtol=1e-9)
    np.testing.assert_allclose(dist, ref["distance"], rtol=1e-7)
    assert np.max(np.abs(xg - ref["x"])) < 1e-6
    assert abs(o.PSNR(x, xg) - o.PSNR(x, ref["x"])) <= 1e-3


def test_fista_and_sapg_any_size_match_oracle(ctx):
    """FISTA and the injected-noise SAPG loop at 96 x 80 / 60 x 44 (the element-wise passes of these loops move two
    pixels per lane, so they need an even number of pixels; an odd count is refused with SBTV_ERR_SIZE)."""
    import sbtv
    import sbtv_oracle as o
    from test_gpu_sapg_fista import _op_struct
    M, N = 96, 80
    x = synth_image(M, N, 4)
    rng = np.random.default_rng(3)
    st = o.demo_setup("moffat", x, rng.standard_normal((M, N)), evMax=1.0)
    p, model = st["p_true"], st["model"]
    tau = 0.03 * st["sigma"] ** 2
    Psi = lambda v, th: o.chambolle_prox_TV_stop(v, lam=th, maxiter=25)[0]
    ref = o.my_fista(st["y"], lambda v: model.A(v, *p), lambda v: model.AT(v, *p), tau, 1.0, o.TVnorm, Psi, 1, 1e-4, 25, x)
    A = sbtv.BlurOperator(model.taps(*p))
    xg, obj, times, mses = sbtv.my_fista(st["y"], A, A.T, tau, 1.0, sbtv.TVnorm, sbtv.Psi_TV(25), 1, 1e-4, 25, x)
    assert len(obj) == len(ref["objective"])
    np.testing.assert_allclose(obj, ref["objective"], rtol=1e-9)
    assert np.max(np.abs(xg - ref["x"])) < 1e-7
    M, N = 60, 44
    x = synth_image(M, N, 7)
    st = o.demo_setup("laplace", x, rng.standard_normal((M, N)), evMax=0.99)
    samples, warmup, burnIn = 8, 4, 5
    nz = rng.standard_normal((warmup - 1 + samples - 1, M, N))
    it = iter(nz)
    ref = o.SAPG_algorithm(st, samples=samples, warmup=warmup, burnIn=burnIn, randn=lambda s: next(it))
    op, c, names = _op_struct("laplace", st, samples, warmup, burnIn)
    res = sbtv.SAPG_algorithm_laplace(st["y"], op, c, noise=nz)[-1]
    np.testing.assert_allclose(res["thetas"], ref["thetas"], rtol=1e-9)
    np.testing.assert_allclose(res["bs"], ref["ps"][0], rtol=1e-8)
    np.testing.assert_allclose(res["sigmas"], ref["sigmas"], rtol=1e-9)
    np.testing.assert_allclose(res["logPiTraceX"], ref["logPiTraceX"], rtol=1e-9)
    np.testing.assert_allclose(res["Xlast_sample"], ref["Xlast_sample"], rtol=1e-8, atol=1e-8)
    with pytest.raises(sbtv.SbtvError) as e:                     # 15 x 13 pixels: odd count
        sbtv.SAPG_algorithm_laplace(st["y"][:15, :13], op, c)
    assert e.value.code == -2


def test_size_limits(ctx):
    import sbtv
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
    with pytest.raises(sbtv.SbtvError) as e:
        A(np.zeros((4097, 16)))
    assert e.value.code == -2
    with pytest.raises(sbtv.SbtvError):                          # the 7 x 7 mask does not fit a 6 x 6 image (conv2c.m:15)
        A(np.zeros((6, 6)))
