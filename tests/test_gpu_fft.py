"""GPU parity: hand-written 2-D FFT kernels (K5/K6), spectral operator (K7), tap spectrum (K8)."""
import numpy as np
import pytest

from conftest import synth_image

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(16, 16), (16, 64), (32, 16), (64, 128), (128, 32), (256, 256), (512, 1024),
                                   (1024, 512), (2048, 2048), (4096, 64), (4096, 2048), (16, 4096),
                                   (4096, 4096)])   # last: the largest supported size
def test_rfft2_matches_numpy(ctx, shape):
    import sbtv
    M, N = shape
    rng = np.random.default_rng(M * 7 + N)
    x = rng.standard_normal((2, M, N)) if M * N <= 512 * 1024 else rng.standard_normal((1, M, N))
    raw = sbtv.rfft2_packed(x)
    U = sbtv.unpack_half_spectrum(raw, M, N)
    ref = np.fft.fft2(x, axes=(1, 2))[:, :M // 2 + 1, :]
    scale = np.sqrt(M * N)                     # typical magnitude of a spectrum entry
    assert np.max(np.abs(U - ref)) / scale < 5e-14 * np.log2(M * N)
    # inverse of the packed spectrum returns the image
    back = sbtv.rfft2_packed(np.transpose(raw, (0, 2, 1)), inverse=True)
    np.testing.assert_allclose(np.transpose(back, (0, 2, 1)), x, rtol=0, atol=1e-13 * np.log2(M * N))


@pytest.mark.parametrize("kind,params", [("gaussian", (0.4, 0.3)), ("moffat", (0.4, 3.5)), ("laplace", (0.3,))])
@pytest.mark.parametrize("shape", [(16, 16), (64, 32), (256, 256)])
def test_operator_closures_match_oracle(ctx, kind, params, shape):
    import sbtv
    import sbtv_oracle as o
    M, N = shape
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 255, (M, N))
    model = o.BlurModel(kind, (M, N))
    taps, dtaps = sbtv.psf_family(kind, 7, params)
    op = sbtv.BlurOperator(taps)
    np.testing.assert_allclose(op.A(x), model.A(x, *params), rtol=0, atol=2e-11)
    np.testing.assert_allclose(op.AT(x), model.AT(x, *params), rtol=0, atol=2e-11)
    for i, dt in enumerate(dtaps):
        np.testing.assert_allclose(sbtv.BlurOperator(dt).A(x), model.dA(i, x, *params), rtol=0, atol=2e-10)
    mu = 0.003
    H = model.H_FFT(*params)
    ref = np.real(o.ifft2(o.fft2(x) / (np.abs(H) ** 2 + mu)))
    got = op.invLS(x, mu)
    assert np.max(np.abs(got - ref)) / np.max(np.abs(ref)) < 1e-12
    # analytic KATs through the GPU path: A(delta) = taps at top-left (Q6), A(const) = const, adjointness
    d = np.zeros((M, N)); d[0, 0] = 1
    np.testing.assert_allclose(op.A(d)[:7, :7], taps, atol=1e-15)
    np.testing.assert_allclose(op.A(np.full((M, N), 2.5)), 2.5, atol=1e-13)
    z = rng.standard_normal((M, N))
    assert np.sum(op.A(x) * z) == pytest.approx(np.sum(x * op.AT(z)), rel=1e-11)


def test_operator_batch_per_image_psf_and_counter(ctx):
    import sbtv
    import sbtv_oracle as o
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 255, (3, 64, 64))
    ps = [(0.4, 0.3), (0.9, 0.2), (0.15, 0.6)]
    taps = np.stack([sbtv.Gaussian_psf(7, *p) for p in ps])
    op = sbtv.BlurOperator(taps)
    ctx.reset_calls()
    y = op.A(x)
    assert ctx.calls == 3          # callcounter.m semantics: one per operator application
    m = o.BlurModel("gaussian", (64, 64))
    for b in range(3):
        np.testing.assert_allclose(y[b], m.A(x[b], *ps[b]), atol=2e-11)


def test_operator_errors(ctx):
    import sbtv
    op = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
    with pytest.raises(sbtv.SbtvError) as e:
        op.A(np.zeros((4100, 48)))                  # larger than 4096 (any other size runs: tests/test_gpu_anysize.py)
    assert e.value.code == -2
    with pytest.raises(sbtv.SbtvError) as e:
        sbtv.rfft2_packed(np.zeros((48, 48)))       # the packed half-spectrum format is a power-of-two format
    assert e.value.code == -2
    with pytest.raises(sbtv.SbtvError) as e:
        op.apply(np.zeros((32, 32)), 4)             # A_wrapper.m:15
    assert e.value.code == -5


def test_full_size_linearity_and_roundtrip(ctx):
    """2048^2: size-independent properties (linearity, Parseval-type adjoint identity, shift covariance)."""
    import sbtv
    M = N = 2048
    op = sbtv.BlurOperator(sbtv.psf_moffat(7, 0.4, 3.5))
    rng = np.random.default_rng(9)
    x = sbtv.to_device(synth_image(M, N, 1))
    z = sbtv.to_device(rng.standard_normal((M, N)))
    Ax, Az = op.A(x), op.A(z)
    lin = op.A(2.0 * x - 0.5 * z)
    assert float((lin - (2.0 * Ax - 0.5 * Az)).abs().max()) < 1e-10
    assert float((Ax * z).sum()) == pytest.approx(float((x * op.AT(z)).sum()), rel=1e-11)
    # circular shift covariance: A(shift x) = shift A(x)
    xs = sbtv.to_device(np.roll(sbtv.to_host(x), (5, 1000), axis=(0, 1)))
    assert np.max(np.abs(sbtv.to_host(op.A(xs)) - np.roll(sbtv.to_host(Ax), (5, 1000), axis=(0, 1)))) < 1e-10
    # invLS inverts (A'A + mu I)
    mu = 0.01
    w = op.invLS(op.AT(op.A(z)) + mu * z, mu)
    assert float((w - z).abs().max()) < 1e-10


@pytest.mark.parametrize("M,N", [(4096, 2048), (4096, 4096)])
def test_maximum_size_operator(ctx, M, N):
    """The largest images the blur operator accepts (M, N <= 4096, sbtv.h): A against the spatial 7x7 circular
    sum on a sample of pixels, the adjoint identity and invLS(mu) as the inverse of A'A + mu I."""
    import sbtv
    rng = np.random.default_rng(5)
    x, z = rng.uniform(0, 255, (M, N)), rng.standard_normal((M, N))
    taps = sbtv.Gaussian_psf(7, 0.4, 0.3)
    A = sbtv.BlurOperator(taps)
    ax, atz = A.A(x), A.AT(z)
    assert float(np.sum(ax * z)) == pytest.approx(float(np.sum(x * atz)), rel=1e-11)
    for i, j in [(0, 0), (1, 5), (M - 1, N - 1), (M // 2, N // 2), (3, N - 2), (M - 3, 2)]:
        ref = sum(taps[m, n] * x[(i - m) % M, (j - n) % N] for m in range(7) for n in range(7))
        assert ax[i, j] == pytest.approx(ref, rel=1e-12)
    mu = 0.05
    r = A.AT(A.A(x)) + mu * x
    np.testing.assert_allclose(A.invLS(r, mu), x, rtol=0, atol=1e-9)
    with pytest.raises(sbtv.SbtvError):
        A.A(np.zeros((8192, 16)))                       # beyond the supported size
    with pytest.raises(sbtv.SbtvError):
        A.A(np.zeros((16, 8192)))


def test_conv2c_diffh_diffv_match_oracle(ctx):
    """SALSA/conv2c.m (mask origin at the mask centre) and diffh / diffv through the spectral operator."""
    import sbtv
    import sbtv_oracle as o
    x = synth_image(64, 32, 4)
    for h in ([[0, 1, -1]], [[0], [1], [-1]], [[1, 2, 3], [4, 5, 6], [7, 8, 9.5]], [[1, -2, 0.5, 3, 1]],
              np.random.default_rng(1).standard_normal((4, 6))):
        np.testing.assert_allclose(sbtv.conv2c(x, h), o.conv2c(x, np.asarray(h, dtype=float)), rtol=0, atol=5e-10)
    np.testing.assert_allclose(sbtv.diffh(x), x - np.roll(x, 1, axis=1), rtol=0, atol=1e-10)
    np.testing.assert_allclose(sbtv.diffv(x), x - np.roll(x, 1, axis=0), rtol=0, atol=1e-10)
    xd = sbtv.to_device(x)
    np.testing.assert_allclose(sbtv.to_host(sbtv.diffh(xd)), x - np.roll(x, 1, axis=1), rtol=0, atol=1e-10)
    tv = np.sum(np.sqrt(sbtv.diffh(x) ** 2 + sbtv.diffv(x) ** 2))                  # utils/TVnorm.m:2
    assert sbtv.TVnorm(x) == pytest.approx(tv, rel=1e-12)
    with pytest.raises(sbtv.SbtvError):
        sbtv.conv2c(np.zeros((16, 16)), np.ones((17, 1)))
