"""Host mirror of the blur-operator "plugin API" (L2): PSF builders and the
A / AT / dA / invLS closures of run_*_demo.m, backed by the HIP FFT kernels.

In the reference an operator is a MATLAB function handle; here it is a small
object that carries the PSF taps (the device code builds the spectrum of
`resize(h)` itself) and is callable like the handle it replaces.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

PSF_GAUSSIAN, PSF_MOFFAT, PSF_LAPLACE = 0, 1, 2
_KINDS = {"gaussian": 0, "moffat": 1, "laplace": 2}


def _taps(kind, taille, params):
    lib = L.load_library()
    p = (C.c_double * 3)(*([float(v) for v in params] + [0.0] * (3 - len(params))))
    n = taille * taille
    t, d0, d1 = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    rc = lib.sbtv_psf_taps(kind, taille, p, t, d0, d1)
    if rc != 0:
        raise L.SbtvError(rc, lib.sbtv_last_error(None).decode())
    cm = lambda a: np.array(a[:]).reshape(taille, taille).T.copy()   # column-major -> [i, j]
    return cm(t), cm(d0), cm(d1)


def Gaussian_psf(taille, w1, w2, phi=0.0):
    """kernel = Gaussian_psf(taille, w1, w2, phi)   (utils/Gaussian_psf.m:2-19)."""
    return _taps(PSF_GAUSSIAN, taille, (w1, w2, phi))[0]


psf_gaussian = Gaussian_psf   # utils/psf_gaussian.m is the same function


def psf_moffat(size, a, b):
    """kernel = psf_moffat(size, a, b)   (utils/psf_moffat.m:2-20)."""
    return _taps(PSF_MOFFAT, size, (a, b))[0]


def psf_laplace(size, b):
    """lap = psf_laplace(size, b)   (utils/psf_laplace.m:1-13)."""
    return _taps(PSF_LAPLACE, size, (b,))[0]


def psf_family(kind, taille, params):
    """(taps, [derivative taps...]) of a PSF family: the spatial kernels behind
    diff_fftgaus_w1/w2.m, diff_moffat_alpha/beta.m, diff_laplace_b.m."""
    k = _KINDS[kind] if isinstance(kind, str) else int(kind)
    t, d0, d1 = _taps(k, taille, params)
    return t, ([d0] if k == PSF_LAPLACE else [d0, d1])


class BlurOperator:
    """The closures A, AT, dA/dp, invLS over one PSF (run_Gaussian_demo.m:126-139,224-225).

    taps: (taille,taille) array, or (B,taille,taille) for per-image PSFs.
    """

    def __init__(self, taps, ctx=None):
        t = np.asarray(taps, dtype=np.float64)
        if t.ndim == 2:
            t = t[None]
        if t.shape[1] != t.shape[2]:
            raise ValueError("PSF taps must be square")
        self.taps = t
        self.taille = t.shape[1]
        self.ctx = ctx

    def _cm(self, B):
        t = self.taps
        if t.shape[0] == 1 and B > 1:
            t = np.repeat(t, B, axis=0)
        if t.shape[0] != B:
            raise ValueError("number of PSFs does not match the image batch")
        return np.ascontiguousarray(np.transpose(t, (0, 2, 1)))    # column-major taps per image

    def apply(self, x, mode, mu=None):
        ctx = self.ctx or L.default_context()
        xi = L.Images(x)
        out = L.empty_like_images(xi)
        taps = self._cm(xi.B)
        mu_a, mu_p = (None, None) if mu is None else L.dvec(mu, xi.B)
        ctx.check(ctx.lib.sbtv_A_wrapper(ctx.h, L.vptr(taps), self.taille, mu_p, xi.ptr, out.ptr,
                                         xi.M, xi.N, xi.B, int(mode), xi.flags), xi.flags)
        sq = (x.dim() == 2) if xi.torch else xi.squeeze
        return L.images_result(out, sq)

    def A(self, x):
        """real(ifft2(H_FFT .* fft2(x)))   (run_Gaussian_demo.m:136)"""
        return self.apply(x, 1)

    def AT(self, x):
        """real(ifft2(conj(H_FFT) .* fft2(x)))   (run_Gaussian_demo.m:137)"""
        return self.apply(x, 2)

    def invLS(self, x, mu):
        """real(ifft2(fft2(x) ./ (abs(H_FFT).^2 + mu)))   (run_Gaussian_demo.m:224-225)"""
        return self.apply(x, 9, mu)

    __call__ = A

    @classmethod
    def from_handle(cls, A, shape, like=None, ctx=None, max_taille=15, rtol=1e-10):
        """Recover the PSF taps from a plain function handle of the blur: the reference builds A as
        real(ifft2(resize(h) .* fft2(x))) with the kernel zero-padded into the TOP-LEFT corner (utils/resize.m:8-11,
        quirk Q6), so A(delta) IS the padded kernel.  One probe call; the response must be confined (to `rtol` of its
        largest entry) to a top-left square of at most 15 x 15, otherwise the handle is not a compact circular blur
        and cannot run on the GPU path.  `like`: an image of the caller (numpy array or CUDA tensor) whose kind the
        probe takes."""
        M, N = int(shape[0]), int(shape[1])
        d = np.zeros((M, N))
        d[0, 0] = 1.0
        if like is not None and L._is_torch(like):
            r = L.to_host(A(L.to_device(d, like.device)))
        else:
            r = np.asarray(A(d), dtype=np.float64)
        if r.shape != (M, N):
            raise L.SbtvError(-10, "A(delta) has the wrong size: the handle is not an image-to-image blur")
        big = float(np.max(np.abs(r)))
        sig = np.abs(r) > rtol * big
        rows, cols = np.nonzero(sig)
        t = int(max(rows.max(), cols.max())) + 1 if rows.size else 1
        if t > max_taille or t > M or t > N:
            raise L.SbtvError(-10, "A(delta) is not confined to a top-left square of at most 15 x 15: the handle is "
                                   "not a compact circular blur built by utils/resize.m (Mask does not fit)")
        return cls(r[:t, :t].copy(), ctx=ctx)

    def mu_of_handle(self, LS, shape, like=None, rtol=1e-8):
        """The mu a plain invLS handle was built with (run_Gaussian_demo.m:222-225: filter = 1./(abs(H).^2 + mu)):
        the sum of invLS(delta) is the DC gain 1 / (H(0,0)^2 + mu) with H(0,0) = sum of the taps.  The whole response is
        then checked against this operator's own invLS (on the GPU) so that an unrelated handle is refused."""
        M, N = int(shape[0]), int(shape[1])
        d = np.zeros((M, N))
        d[0, 0] = 1.0
        tor = like is not None and L._is_torch(like)
        r = L.to_host(LS(L.to_device(d, like.device))) if tor else np.asarray(LS(d), dtype=np.float64)
        h00 = float(np.sum(self.taps[0]))
        mu = 1.0 / float(np.sum(r)) - h00 * h00
        if not (mu > 0):
            raise L.SbtvError(-9, "(A^T A + \\mu I)^(-1): the LS handle does not have the form 1./(abs(H).^2 + mu)")
        mine = np.asarray(self.invLS(d, mu), dtype=np.float64)
        if np.max(np.abs(mine - r)) > rtol * np.max(np.abs(r)):
            raise L.SbtvError(-9, "(A^T A + \\mu I)^(-1): the LS handle is not the inverse filter of A")
        return mu

    def check_adjoint_handle(self, AT, shape, like=None, rtol=1e-8):
        """A plain AT handle must be the adjoint of this blur (AT(delta) == this operator's AT(delta))."""
        M, N = int(shape[0]), int(shape[1])
        d = np.zeros((M, N))
        d[0, 0] = 1.0
        tor = like is not None and L._is_torch(like)
        r = L.to_host(AT(L.to_device(d, like.device))) if tor else np.asarray(AT(d), dtype=np.float64)
        mine = np.asarray(self.AT(d), dtype=np.float64)
        if r.shape != mine.shape or np.max(np.abs(mine - r)) > rtol * np.max(np.abs(mine)):
            raise L.SbtvError(-8, "The function handle for transpose of A is not the adjoint of A")

    @property
    def T(self):
        return _Adjoint(self)

    def LS(self, mu):
        return _InvLS(self, mu)


class _Adjoint:
    def __init__(self, op):
        self.op = op

    def __call__(self, x):
        return self.op.AT(x)


class _InvLS:
    def __init__(self, op, mu):
        self.op = op
        self.mu = mu

    def __call__(self, x):
        return self.op.invLS(x, self.mu)


def A_wrapper(A, AT, x, M1, N1, M2, N2, mode):
    """g = A_wrapper(A, AT, x, M1, N1, M2, N2, mode)   (SALSA/A_wrapper.m:5-17):
    vectorised adaptor, mode 1 -> A, mode 2 -> AT; x is a column-major vector."""
    x = np.asarray(x, dtype=np.float64)
    if mode == 1:
        xt = x.reshape((M1, N1), order="F")
        gt = A(xt)
        return np.asarray(gt).reshape((M2 * N2, 1), order="F")
    if mode == 2:
        xt = x.reshape((M2, N2), order="F")
        gt = AT(xt)
        return np.asarray(gt).reshape((M1 * N1, 1), order="F")
    raise L.SbtvError(-5, "The value of parameter mode must be 1 or 2.")


def rfft2_packed(x, inverse=False, ctx=None):
    """Debug/test access to the FFT kernels: packed half spectrum (see sbtv.h)."""
    ctx = ctx or L.default_context()
    xi = L.Images(x)
    out = L.empty_like_images(xi)
    ctx.check(ctx.lib.sbtv_rfft2_packed(ctx.h, xi.ptr, out.ptr, xi.M, xi.N, xi.B, 1 if inverse else 0, xi.flags), xi.flags)
    return out.buf if not out.torch else out.t     # raw (B, N, M) memory for numpy


def unpack_half_spectrum(buf, M, N):
    """(B,N,M) raw doubles of a packed spectrum -> complex (B, M/2+1, N) true rows 0..M/2."""
    B = buf.shape[0]
    n1 = M // 2
    S = buf.reshape(B, N, n1, 2)
    S = (S[..., 0] + 1j * S[..., 1]).transpose(0, 2, 1)        # (B, n1, N): S[k, l]
    U = np.zeros((B, n1 + 1, N), dtype=complex)
    U[:, 1:n1] = S[:, 1:]
    P = S[:, 0]
    Q = np.conj(S[:, 0][:, (N - np.arange(N)) % N])
    U[:, 0] = (P + Q) / 2
    U[:, n1] = (P - Q) / 2j
    return U


def conv2c(x, h, ctx=None):
    """y = conv2c(x, h)  (SALSA/conv2c.m:1-50): circular 2-D convolution with the mask origin at floor((1+size)/2).

    Runs on the spectral blur operator (whose mask origin is the top-left tap, utils/resize.m:8): the mask is embedded
    in a square tap array and the result rotated back by the origin offset.  Needs power-of-two image sizes like the
    operator; masks up to 15 x 15."""
    hh = np.atleast_2d(np.asarray(h, dtype=np.float64))
    mm, nm = hh.shape
    xi = L.Images(x)
    if mm > xi.M or nm > xi.N:
        raise L.SbtvError(-10, "Mask does not fit inside array")                               # conv2c.m:15
    T = max(mm, nm)
    taps = np.zeros((T, T))
    taps[:mm, :nm] = hh
    out = BlurOperator(taps, ctx=ctx).A(x)
    mo, no = (1 + mm) // 2 - 1, (1 + nm) // 2 - 1                                              # conv2c.m:19-20
    if L._is_torch(out):
        import torch
        return torch.roll(out, shifts=(-mo, -no), dims=(-2, -1))
    return np.roll(out, shift=(-mo, -no), axis=(-2, -1))


def diffh(x, ctx=None):
    """sol = diffh(x) = conv2c(x, [0 1 -1])  (SALSA/diffh.m:1-3): x(i,j) - x(i,j-1), circular."""
    return conv2c(x, np.array([[0.0, 1.0, -1.0]]), ctx=ctx)


def diffv(x, ctx=None):
    """sol = diffv(x) = conv2c(x, [0 1 -1]')  (SALSA/diffv.m:1-3): x(i,j) - x(i-1,j), circular."""
    return conv2c(x, np.array([[0.0], [1.0], [-1.0]]), ctx=ctx)
