"""CPU oracle for the SALSA / SAPG / FISTA TV-deblurring hot path.

TEST INFRASTRUCTURE ONLY.  This module is a float64 NumPy restatement of the
reference's MATLAB arithmetic for the path named in BASELINE.json.  It is the
checker for the HIP path; nothing under `semi-blind-image-deblurring-problems-
with-tv_amd/` may import it.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` use it.

PARITY STATUS: **parity unpinned against MATLAB output**.  The reference is
MATLAB-only, ships no tests, no golden vectors and no stored results, and
neither MATLAB nor Octave exists in the build container.  The
restatement is pinned instead by (1) the analytic / hand-checkable known-answer
tests listed in SURVEY.md §8c (tests/test_oracle_kat.py) and (2) line-by-line
correspondence with the reference sources cited on every function below
(paths relative to /root/reference).

Conventions: arrays are indexed [i, j] = MATLAB (i+1, j+1).  Memory order is
irrelevant here; the device boundary uses column-major (Fortran) buffers.
`fft2`/`ifft2` are scipy.fft with `workers` threads (MATLAB's fft2 is
multithreaded as well); they are mathematically defined so any conforming FFT
pins them to rounding.
"""
from __future__ import annotations

import math
import os
import time

import numpy as np
import scipy.fft as _sfft

def _host_cores() -> int:
    """Cores this process may really use: the affinity mask, capped by a cgroup CPU quota if one is set
    (a container that sees 256 CPUs but owns a 16-core share would otherwise oversubscribe the FFT pool)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


_WORKERS = _host_cores()


def set_workers(n: int) -> None:
    """Number of threads scipy.fft may use (cpu_baseline reports it)."""
    global _WORKERS
    _WORKERS = max(1, int(n))


def get_workers() -> int:
    return _WORKERS


def fft2(x):
    return _sfft.fft2(x, workers=_WORKERS)


def ifft2(x):
    return _sfft.ifft2(x, workers=_WORKERS)


# --------------------------------------------------------------------------
# L1: TV primitives
# --------------------------------------------------------------------------

def GradientIm(u):
    """utils/chambolle_prox_TV_stop.m:161-166 — forward differences, last
    row of dux and last column of duy are zero."""
    dux = np.zeros_like(u)
    dux[:-1, :] = u[1:, :] - u[:-1, :]
    duy = np.zeros_like(u)
    duy[:, :-1] = u[:, 1:] - u[:, :-1]
    return dux, duy


def DivergenceIm(p1, p2):
    """utils/chambolle_prox_TV_stop.m:152-159.  Quirk Q3: the last entry along
    each axis is ``-p(end)`` (not ``-p(end-1)``), i.e. NOT the adjoint of
    GradientIm."""
    v = np.empty_like(p2)
    v[:, 0] = p2[:, 0]
    v[:, 1:-1] = p2[:, 1:-1] - p2[:, :-2]
    v[:, -1] = -p2[:, -1]
    u = np.empty_like(p1)
    u[0, :] = p1[0, :]
    u[1:-1, :] = p1[1:-1, :] - p1[:-2, :]
    u[-1, :] = -p1[-1, :]
    return v + u


def chambolle_prox_TV_stop(g, lam=1.0, maxiter=None, tol=1e-3, tau=0.249,
                           dualvars=None, return_info=False):
    """utils/chambolle_prox_TV_stop.m:1-149.

    ``maxiter`` is REQUIRED (quirk Q1: the default `maxiter = 10` at :80 is
    dead, the loop reads `MaxIter` which only the 'maxiter' option sets, :95-96).
    ``dualvars`` is the M x 2N horizontal concatenation [px py]; the split uses
    M for the column index (quirk Q2, :105-107).  A (px, py) TUPLE is taken as the two
    arrays directly: that is what the C-ABI passes (separate pointers, so rectangular
    images can be warm-started too) and has no counterpart in the reference's option list.
    Returns (f, px, py) like the reference; with return_info also (k, err).
    """
    if maxiter is None:
        raise NameError("MaxIter undefined: 'maxiter' must be given (chambolle_prox_TV_stop.m:95,131)")
    g = np.asarray(g, dtype=np.float64)
    px = np.zeros_like(g)
    py = np.zeros_like(g)
    if isinstance(dualvars, tuple):
        px = np.array(dualvars[0], dtype=np.float64)
        py = np.array(dualvars[1], dtype=np.float64)
        if px.shape != g.shape or py.shape != g.shape:
            raise ValueError("Wrong size of the dual variables")
    elif dualvars is not None:
        M, N = g.shape
        Maux, Naux = dualvars.shape
        if M != Maux or Naux != 2 * N:
            raise ValueError("Wrong size of the dual variables")
        py = np.array(dualvars[:, M:], dtype=np.float64)
        px = np.array(dualvars[:, :M], dtype=np.float64)
    cont = True
    k = 0
    err = 0.0
    while cont:
        k += 1
        divp = DivergenceIm(px, py)                     # :123
        u = divp - g / lam                              # :124
        upx, upy = GradientIm(u)                        # :126
        tmp = np.sqrt(upx ** 2 + upy ** 2)              # :127
        err = math.sqrt(float(np.sum((-upx + tmp * px) ** 2 + (-upy + tmp * py) ** 2)))  # :128 (old p)
        px = (px + tau * upx) / (1 + tau * tmp)         # :129
        py = (py + tau * upy) / (1 + tau * tmp)         # :130
        cont = (k < maxiter) and (err > tol)            # :131
    f = g - lam * DivergenceIm(px, py)                  # :149
    if return_info:
        return f, px, py, k, err
    return f, px, py


def diffh(x):
    """SALSA/diffh.m:1-3 -> conv2c(x,[0 1 -1]) == x(i,j) - x(i,j-1), circular."""
    return x - np.roll(x, 1, axis=1)


def diffv(x):
    """SALSA/diffv.m:1-3 -> conv2c(x,[0 1 -1]') == x(i,j) - x(i-1,j), circular."""
    return x - np.roll(x, 1, axis=0)


def conv2c(x, h):
    """SALSA/conv2c.m:1-50 — circular 2-D convolution with the mask origin at
    floor((1+size)/2) (1-based), written as an explicit circular sum."""
    x = np.asarray(x, dtype=np.float64)
    h = np.atleast_2d(np.asarray(h, dtype=np.float64))
    mm, nm = h.shape
    if mm > x.shape[0] or nm > x.shape[1]:
        raise ValueError("Mask does not fit inside array")
    mo = (1 + mm) // 2 - 1
    no = (1 + nm) // 2 - 1
    y = np.zeros_like(x)
    for a in range(mm):
        for b in range(nm):
            if h[a, b] != 0.0:
                y += h[a, b] * np.roll(np.roll(x, a - mo, axis=0), b - no, axis=1)
    return y


def TVnorm(x):
    """utils/TVnorm.m:1-2 — isotropic TV with PERIODIC backward differences."""
    return float(np.sum(np.sqrt(diffh(x) ** 2 + diffv(x) ** 2)))


# --------------------------------------------------------------------------
# L2: PSF builders and the FFT blur operator
# --------------------------------------------------------------------------

def resize(kernel, im_shape):
    """utils/resize.m:1-12 — zero-pad into the TOP-LEFT corner, fft2 (no centring)."""
    taille = kernel.shape[0]
    full = np.zeros(im_shape, dtype=np.float64)
    full[:taille, :taille] = kernel
    return fft2(full)


def _grid(taille, phi):
    center = (taille + 1) / 2
    x = np.arange(-taille + center, taille - center + 1)
    v, u = np.meshgrid(x, x, indexing="ij")          # ndgrid: v rows, u cols
    U = u * math.cos(phi) - v * math.sin(phi)
    V = u * math.sin(phi) + v * math.cos(phi)
    return U, V


def Gaussian_psf(taille, w1, w2, phi=0.0):
    """utils/Gaussian_psf.m:2-19 (== utils/psf_gaussian.m:2-19)."""
    U, V = _grid(taille, phi)
    c = w1 ** 2 * U ** 2 + w2 ** 2 * V ** 2
    kernel = ((w1 * w2) / (2 * math.pi)) * np.exp(-c / 2)
    return kernel / np.sum(kernel)


psf_gaussian = Gaussian_psf


def Sum_gauss_psf(taille, w1, w2, phi=0.0):
    """utils/Sum_gauss_psf.m:1-28."""
    U, V = _grid(taille, phi)
    c = w1 ** 2 * U ** 2 + w2 ** 2 * V ** 2
    f = ((w1 * w2) / (2 * math.pi)) * np.exp(-c / 2)
    diffw2 = (w1 / (2 * math.pi)) * (1 - w2 ** 2 * V ** 2) * np.exp(-c / 2)
    diffw1 = (w2 / (2 * math.pi)) * (1 - w1 ** 2 * U ** 2) * np.exp(-c / 2)
    return float(np.sum(f)), float(np.sum(diffw1)), float(np.sum(diffw2))


def diff_gaus_w1_taps(taille, w1, w2, phi=0.0):
    """utils/diff_fftgaus_w1.m:2-24 (spatial taps before `resize`)."""
    U, V = _grid(taille, phi)
    sum_psf, sum_d_w1, _ = Sum_gauss_psf(taille, w1, w2, phi)
    c = w1 ** 2 * U ** 2 + w2 ** 2 * V ** 2
    f = ((w1 * w2) / (2 * math.pi)) * np.exp(-c / 2)
    diff = (w2 / (2 * math.pi)) * (1 - w1 ** 2 * U ** 2) * np.exp(-c / 2)
    return (diff * sum_psf - f * sum_d_w1) / (sum_psf ** 2)


def diff_gaus_w2_taps(taille, w1, w2, phi=0.0):
    """utils/diff_fftgaus_w2.m:2-24."""
    U, V = _grid(taille, phi)
    sum_psf, _, sum_d_w2 = Sum_gauss_psf(taille, w1, w2, phi)
    c = w1 ** 2 * U ** 2 + w2 ** 2 * V ** 2
    f = ((w1 * w2) / (2 * math.pi)) * np.exp(-c / 2)
    diff = (w1 / (2 * math.pi)) * (1 - w2 ** 2 * V ** 2) * np.exp(-c / 2)
    return (diff * sum_psf - f * sum_d_w2) / (sum_psf ** 2)


def diff_fftgaus_w1(im_size, taille, w1, w2, phi=0.0):
    """utils/diff_fftgaus_w1.m:25."""
    return resize(diff_gaus_w1_taps(taille, w1, w2, phi), im_size)


def diff_fftgaus_w2(im_size, taille, w1, w2, phi=0.0):
    """utils/diff_fftgaus_w2.m:25."""
    return resize(diff_gaus_w2_taps(taille, w1, w2, phi), im_size)


def psf_moffat(size, a, b):
    """utils/psf_moffat.m:2-20 (== taps of utils/moffat_psf.m:2-20)."""
    center = (size + 1) / 2
    X = np.arange(-size + center, size - center + 1)
    xy = X[:, None] ** 2 + X[None, :] ** 2
    b2 = b + 2
    kernel = a ** 2 * (xy * a ** 2 / b + 1) ** (-b2 / 2) / (2 * math.pi)
    return kernel / np.sum(kernel)


def moffat_psf(im_shape, size, a, b):
    """utils/moffat_psf.m:2-23."""
    return resize(psf_moffat(size, a, b), im_shape)


def sum_mof_psf(size, a, b):
    """utils/sum_mof_psf.m:1-40."""
    center = (size + 1) / 2
    X = np.arange(-size + center, size - center + 1)
    xy = X[:, None] ** 2 + X[None, :] ** 2
    b2 = b + 2
    kernel = a ** 2 * ((xy * a ** 2) / b + 1) ** (-b2 / 2) / (2 * math.pi)
    dal = (2 - (((b + 2) * xy * a ** 2) / (2 * (b + xy * a ** 2)))) * (1 + xy * a ** 2 / b) ** (-(b + 2) / 2) * (a / (2 * math.pi))
    cons1 = a ** 2 / (4 * math.pi)
    dbe = (-np.log(xy * a ** 2 / b + 1) + (b2 * xy * a ** 2) / (b * (b + xy * a ** 2))) * (xy * a ** 2 / b + 1) ** (-b2 / 2) * cons1
    return float(np.sum(kernel)), float(np.sum(dal)), float(np.sum(dbe))


def diff_moffat_alpha_taps(size, a, b):
    """utils/diff_moffat_alpha.m:1-20."""
    sum_psf, sum_da, _ = sum_mof_psf(size, a, b)
    center = (size + 1) / 2
    X = np.arange(-size + center, size - center + 1)
    xy = X[:, None] ** 2 + X[None, :] ** 2
    b2 = b + 2
    f = a ** 2 * ((xy * a ** 2) / b + 1) ** (-b2 / 2) / (2 * math.pi)
    diff = (2 - (((b + 2) * xy * a ** 2) / (2 * (b + xy * a ** 2)))) * (1 + xy * a ** 2 / b) ** (-(b + 2) / 2) * (a / (2 * math.pi))
    return (diff * sum_psf - f * sum_da) / sum_psf ** 2


def diff_moffat_beta_taps(size, a, b):
    """utils/diff_moffat_beta.m:1-21."""
    sum_psf, _, sum_db = sum_mof_psf(size, a, b)
    center = (size + 1) / 2
    X = np.arange(-size + center, size - center + 1)
    xy = X[:, None] ** 2 + X[None, :] ** 2
    b2 = b + 2
    f = a ** 2 * ((xy * a ** 2) / b + 1) ** (-b2 / 2) / (2 * math.pi)
    cons1 = a ** 2 / (4 * math.pi)
    diff = (-np.log(xy * a ** 2 / b + 1) + (b2 * xy * a ** 2) / (b * (b + xy * a ** 2))) * (xy * a ** 2 / b + 1) ** (-b2 / 2) * cons1
    return (diff * sum_psf - f * sum_db) / sum_psf ** 2


def diff_moffat_alpha(im_shape, size, a, b):
    return resize(diff_moffat_alpha_taps(size, a, b), im_shape)


def diff_moffat_beta(im_shape, size, a, b):
    return resize(diff_moffat_beta_taps(size, a, b), im_shape)


def psf_laplace(size, b):
    """utils/psf_laplace.m:1-13 (== taps of utils/laplace_psf.m:1-12)."""
    center = (size + 1) / 2
    x = np.arange(-size + center, size - center + 1)
    s = np.abs(x)[:, None] + np.abs(x)[None, :]
    lap = (b ** 2 / 4) * np.exp(-b * s)
    return lap / np.sum(lap)


def laplace_psf(im_shape, size, b):
    """utils/laplace_psf.m:1-15."""
    return resize(psf_laplace(size, b), im_shape)


def sum_lap_psf(size, b):
    """utils/sum_lap_psf.m:1-28."""
    center = (size + 1) / 2
    x = np.arange(-size + center, size - center + 1)
    s = np.abs(x)[:, None] + np.abs(x)[None, :]
    kernel = (b ** 2 / 4) * np.exp(-b * s)
    d = ((2 * b - b ** 2 * s) / 4) * np.exp(-b * s)
    return float(np.sum(kernel)), float(np.sum(d))


def diff_laplace_b_taps(size, b):
    """utils/diff_laplace_b.m:1-14."""
    center = (size + 1) / 2
    x = np.arange(-size + center, size - center + 1)
    s = np.abs(x)[:, None] + np.abs(x)[None, :]
    sum_psf, sum_db = sum_lap_psf(size, b)
    f = (b ** 2 / 4) * np.exp(-b * s)
    diff = ((2 * b - b ** 2 * s) / 4) * np.exp(-b * s)
    return (diff * sum_psf - f * sum_db) / sum_psf ** 2


def diff_laplace_b(im_shape, size, b):
    return resize(diff_laplace_b_taps(size, b), im_shape)


PSF_TAPS = {
    # kind -> (taps(params), [derivative taps(params), ...])
    "gaussian": (lambda t, p: Gaussian_psf(t, p[0], p[1], 0.0 if len(p) < 3 else p[2]),
                 [lambda t, p: diff_gaus_w1_taps(t, p[0], p[1], 0.0 if len(p) < 3 else p[2]),
                  lambda t, p: diff_gaus_w2_taps(t, p[0], p[1], 0.0 if len(p) < 3 else p[2])]),
    "moffat": (lambda t, p: psf_moffat(t, p[0], p[1]),
               [lambda t, p: diff_moffat_alpha_taps(t, p[0], p[1]),
                lambda t, p: diff_moffat_beta_taps(t, p[0], p[1])]),
    "laplace": (lambda t, p: psf_laplace(t, p[0]),
                [lambda t, p: diff_laplace_b_taps(t, p[0])]),
}


class BlurModel:
    """The operator closures of the demos, parameterised by PSF family.

    run_Gaussian_demo.m:126-139, run_moffat_demo.m:122-137,
    run_laplace_demo.m:96-107.  Like the reference, the PSF spectrum is
    RECOMPUTED (a full-size fft2 of the padded taps) on every operator call;
    `calls_fft` counts 2-D FFTs so the redundancy is visible to the baseline.
    """

    def __init__(self, kind, im_shape, psf_size=7):
        self.kind = kind
        self.im_shape = tuple(im_shape)
        self.psf_size = psf_size
        self._taps, self._dtaps = PSF_TAPS[kind]

    def taps(self, *p):
        return self._taps(self.psf_size, p)

    def dtaps(self, i, *p):
        return self._dtaps[i](self.psf_size, p)

    def H_FFT(self, *p):
        return resize(self.taps(*p), self.im_shape)

    def HC_FFT(self, *p):
        return np.conj(self.H_FFT(*p))

    def A(self, x, *p):
        return np.real(ifft2(self.H_FFT(*p) * fft2(x)))

    def AT(self, x, *p):
        return np.real(ifft2(self.HC_FFT(*p) * fft2(x)))

    def dA(self, i, x, *p):
        return np.real(ifft2(resize(self.dtaps(i, *p), self.im_shape) * fft2(x)))


def max_eigenval(A, At, params, im_size, tol, max_iter, randn):
    """utils/max_eigenval_Gaussian_Moffat.m:1-27 / max_eigenval_Laplace.m:1-28.
    `randn(shape)` supplies the start vector (MATLAB's stream is unpinned)."""
    x = randn(im_size)
    x = x / np.linalg.norm(x.ravel())
    init_val = 1.0
    val = init_val
    for _ in range(int(max_iter)):
        y = A(x, *params)
        x = At(y, *params)
        val = float(np.linalg.norm(x.ravel()))
        rel_var = abs(val - init_val) / init_val
        if rel_var < tol:
            break
        init_val = val
        x = x / val
    return val


# --------------------------------------------------------------------------
# metrics
# --------------------------------------------------------------------------

def PSNR(x, y):
    """utils/PSNR.m:2-4 (peak = max of the TRUE image x)."""
    x = np.asarray(x, dtype=np.float64)
    mse = 10 * math.log10(float(np.max(x)) ** 2)
    return mse - 10 * math.log10(float(np.sum((x - y) ** 2)) / x.size)


def MSE(x_true, x_app):
    """utils/MSE.m:1-4 (in dB)."""
    return 10 * math.log10(float(np.sum((x_true - x_app) ** 2)) / x_true.size)


def l2(x, y):
    """utils/l2.m:1-3 — norm() of a MATRIX is the spectral norm (quirk Q9)."""
    return float(np.linalg.norm(x - y, 2)) ** 2


# --------------------------------------------------------------------------
# L3: SALSA_v2 (TV path) and FISTA
# --------------------------------------------------------------------------

def SALSA_v2(y, A, tau, mu=1e-3, AT=None, invLS=None, true_x=None,
             stopcriterion=1, tolA=0.001, maxiter=10000, TViters=5,
             initialization=0, max_time=None):
    """SALSA/SALSA_v2.m:156-494 with 'TVINITIALIZATION' = 1 (the only mode the
    demos use; user Psi/Phi are then ignored, quirk Q7, :318-320,354-359).

    Returns dict with x, numA, numAt, objective, distance, times, mses,
    n_outer.  `max_time` (seconds) is an oracle-only knob that truncates the
    loop for the bounded cpu_baseline sample; it never fires in tests.
    """
    if stopcriterion not in (1, 2, 3):
        raise ValueError("Unknown stopping criterion")
    if AT is None:
        raise ValueError("The function handle for transpose of A is missing")
    if invLS is None:
        raise ValueError("(A^T A + \\mu I)^(-1) must be specified as a function handle.")
    numA = 0
    numAt = 0
    ATy = AT(y)                                              # :288
    numAt += 1
    dummy = invLS(ATy)                                       # :298
    if dummy.shape != ATy.shape:
        raise ValueError("Specified function handle for solving the LS step does not seem compatible")
    phi = TVnorm                                             # :358
    if isinstance(initialization, np.ndarray):
        x = np.array(initialization, dtype=np.float64)       # :220-226,374-378
    elif initialization == 0:
        x = AT(np.zeros_like(y))                             # :369
    elif initialization == 2:
        x = ATy.copy()                                       # :373
    else:
        raise ValueError("Unknown 'Initialization' option")
    if true_x is not None and true_x.shape != x.shape:
        raise ValueError("Initial x has incompatible size")
    u = x.copy()                                             # :392
    bu = np.zeros_like(u)                                    # :393
    threshold = tau / mu                                     # :394
    resid = y - A(x)                                         # :399
    numA += 1
    prev_f = 0.5 * float(np.sum(resid * resid)) + tau * phi(u)   # :401
    t0 = time.process_time()
    times = [0.0]
    objective = [prev_f]
    mses = []
    if true_x is not None:
        mses.append(float(np.sum((x - true_x) ** 2)) / x.size)   # :414
    pux = np.zeros_like(u)                                   # :418-421
    puy = np.zeros_like(u)
    distance = []
    criterion = [1.0]
    n_outer = 0
    wall0 = time.perf_counter()
    for outer in range(1, int(maxiter) + 1):                 # :423
        n_outer = outer
        xprev = x
        u, pux, puy = chambolle_prox_TV_stop(np.real(x - bu), lam=threshold, maxiter=TViters,
                                             dualvars=np.hstack([pux, puy]))        # :429
        r = ATy + mu * (u + bu)                              # :434
        x = invLS(r)                                         # :436
        bu = bu + (u - x)                                    # :440
        resid = y - A(x)                                     # :442
        numA += 1
        objective.append(0.5 * float(np.sum(resid * resid)) + tau * phi(u))          # :444
        if true_x is not None:
            e = x - true_x
            mses.append(float(np.sum(e * e)) / x.size)       # :446-449
        distance.append(float(np.linalg.norm((x - u).ravel())) /
                        math.sqrt(float(np.sum(x * x)) + float(np.sum(u * u))))      # :451
        if outer > 1:                                        # :453
            if stopcriterion == 1:
                crit = abs(objective[outer] - objective[outer - 1]) / objective[outer - 1]
            elif stopcriterion == 2:
                crit = abs(float(np.linalg.norm((x - xprev).ravel())) / float(np.linalg.norm(x.ravel())))
            else:
                crit = objective[outer]
            criterion.append(crit)
            if crit < tolA:                                  # :472
                times.append(time.process_time() - t0)
                break
        times.append(time.process_time() - t0)               # :493
        if max_time is not None and time.perf_counter() - wall0 > max_time:
            break
    return dict(x=x, numA=numA, numAt=numAt, objective=np.array(objective), wall_loop=time.perf_counter() - wall0,
                distance=np.array(distance), times=np.array(times),
                mses=np.array(mses), n_outer=n_outer, u=u, bu=bu, pux=pux, puy=puy,
                criterion=np.array(criterion))


def CSALSA_v2(y, A, mu1, mu2, sigma, AT=None, invLS=None, true_x=None, stopcriterion=3, tolA=0.001,
              maxiter=10000, TViters=5, initialization=0, continuationfactor=1.0, epsilon=0.0):
    """SALSA/CSALSA_v2.m:160-561 (`csalsa`, constrained problem min TV(x) s.t. ||Ax-y|| <= epsilon) with
    'TVINITIALIZATION' = 1 and P = PT = identity.  `invLS(r, mu)` takes the weight as second argument
    (:310,471).  All traces are 1-based in MATLAB: index 0 here is the state before the loop, and the loop
    runs outer = 2..maxiter (:461), i.e. at most maxiter-1 updates."""
    if stopcriterion not in (1, 2, 3):
        raise ValueError("Unknown stopping criterion")                   # :258
    if AT is None:
        raise ValueError("The function handle for transpose of A is missing")
    if invLS is None:
        raise ValueError("(A^T A + \\mu I)^(-1) must be specified as a function handle.")
    numA = numAt = 0
    ATy = AT(y)                                                          # :301
    numAt += 1
    dummy = invLS(ATy, mu1)                                              # :310
    if dummy.shape != ATy.shape:
        raise ValueError("Specified function handle for solving the LS step does not seem compatible")
    phi = TVnorm                                                         # :371
    if isinstance(initialization, np.ndarray):
        x = np.array(initialization, dtype=np.float64)
    elif initialization == 0:
        x = AT(np.zeros_like(y))                                         # :380
    elif initialization == 2:
        x = ATy.copy()                                                   # :384
        numAt += 1
    else:
        raise ValueError("Unknown 'Initialization' option")
    u = np.zeros_like(x)                                                 # :404-408
    bu = np.zeros_like(x)
    v = np.zeros_like(y)
    bv = np.zeros_like(y)
    if not epsilon:
        epsilon = math.sqrt(y.size + 8 * math.sqrt(y.size)) * sigma      # :413
    Ax = A(x)                                                            # :416
    numA += 1                                                            # :421
    objective = [phi(x)]                                                 # :423
    mses = [float(np.sum((x - true_x) ** 2)) / x.size] if true_x is not None else []   # :436
    pux = np.zeros_like(u)
    puy = np.zeros_like(u)
    Ax = A(x)                                                            # :444
    numA += 1
    criterion = [float(np.linalg.norm((Ax - y).ravel()))]                # :446
    distance1 = [float(np.linalg.norm((Ax - y - v).ravel()))]            # :447
    distance2 = [float(np.linalg.norm((x - u).ravel()))]                 # :448
    delta = continuationfactor
    n_outer = 1
    for outer in range(2, int(maxiter) + 1):                             # :461
        n_outer = outer
        xprev = x
        r = mu1 * (u + bu) + mu2 * AT(y + v + bv)                        # :467
        numAt += 1
        x = invLS(r, mu1)                                                # :471
        u, pux, puy = chambolle_prox_TV_stop(np.real(x - bu), lam=1.0 / mu1, maxiter=TViters,
                                             dualvars=np.hstack([pux, puy]))          # :476
        Ax = A(x)                                                        # :481
        numA += 1
        ve = Ax - y - bv                                                 # :483
        n_ve = float(np.linalg.norm(ve.ravel()))
        v = ve if n_ve <= epsilon else ve / n_ve * epsilon               # :485-489
        bv = bv - (Ax - y - v)                                           # :491
        bu = bu - (x - u)                                                # :492
        criterion.append(float(np.linalg.norm((Ax - y).ravel())))        # :494
        distance1.append(float(np.linalg.norm((Ax - y - v).ravel())))    # :495
        distance2.append(float(np.linalg.norm((x - u).ravel())))         # :497
        objective.append(phi(x))                                         # :498
        if true_x is not None:
            mses.append(float(np.linalg.norm((x - true_x).ravel())) ** 2 / true_x.size)   # :501
        mu1 = mu1 * delta                                                # :517-518
        mu2 = mu2 * delta
        k = outer - 1                                                    # 0-based position of `outer`
        if stopcriterion == 1:
            sc = abs(objective[k] - objective[k - 1]) / objective[k]     # :527
        elif stopcriterion == 2:
            sc = abs(float(np.linalg.norm((x - xprev).ravel())) / float(np.linalg.norm(x.ravel())))   # :534
        else:
            sc = abs(criterion[k] - criterion[k - 1]) / criterion[k]     # :539
        if sc < tolA and criterion[k] <= epsilon:                        # :529,535,541
            break
    return dict(x=x, numA=numA, numAt=numAt, objective=np.array(objective), distance1=np.array(distance1),
                distance2=np.array(distance2), criterion=np.array(criterion), mses=np.array(mses),
                n_outer=n_outer, epsilon=epsilon, u=u, v=v)


def CoRAL_v2(y, A, tau1, tau2, mu1=1e-3, mu2=1e-3, AT=None, invLS=None, true_x=None, stopcriterion=1,
             tolA=0.001, maxiter=10000, TViters1=5, TViters2=5, initialization=0):
    """SALSA/CoRAL_v2.m:2-476 (compound regulariser, two TV terms: 'TVINITIALIZATION1' = 'TVINITIALIZATION2' = 1,
    P1 = P2 = identity).  Note the initial scaled multipliers are bu = u = x and bv = v = x (:353-359),
    not zero as in SALSA_v2."""
    if stopcriterion not in (1, 2, 3):
        raise ValueError("Unknown stopping criterion")                   # :133
    if AT is None:
        raise ValueError("The function handle for transpose of A is missing")
    if invLS is None:
        raise ValueError("(A^T A + \\mu I)^(-1) must be specified as a function handle.")
    numA = numAt = 0
    ATy = AT(y)                                                          # :189
    numAt += 1
    dummy = invLS(ATy)                                                   # :199
    if dummy.shape != ATy.shape:
        raise ValueError("Specified function handle for solving the LS step does not seem compatible")
    ATy = AT(y)                                                          # :219 (computed twice)
    numAt += 1
    if isinstance(initialization, np.ndarray):
        x = np.array(initialization, dtype=np.float64)
    elif initialization == 0:
        x = AT(np.zeros_like(y))                                         # :329
    elif initialization == 2:
        x = ATy.copy()                                                   # :333
    else:
        raise ValueError("Unknown 'Initialization' option")
    u = x.copy()                                                         # :353-361
    bu = u.copy()
    threshold1 = tau1 / mu1
    v = x.copy()
    bv = v.copy()
    threshold2 = tau2 / mu2
    resid = y - A(x)                                                     # :366
    numA += 1
    objective = [0.5 * float(np.sum(resid * resid)) + tau1 * TVnorm(u) + tau2 * TVnorm(v)]   # :368
    mses = [float(np.sum((x - true_x) ** 2)) / x.size] if true_x is not None else []
    pux = np.zeros_like(u)
    puy = np.zeros_like(u)
    pvx = np.zeros_like(v)
    pvy = np.zeros_like(v)
    distance = []
    criterion = [1.0]
    n_outer = 0
    for outer in range(1, int(maxiter) + 1):                             # :394
        n_outer = outer
        xprev = x
        u, pux, puy = chambolle_prox_TV_stop(np.real(x - bu), lam=threshold1, maxiter=TViters1,
                                             dualvars=np.hstack([pux, puy]))          # :401
        v, pvx, pvy = chambolle_prox_TV_stop(np.real(x - bv), lam=threshold2, maxiter=TViters2,
                                             dualvars=np.hstack([pvx, pvy]))          # :406
        r = ATy + mu1 * (u + bu) + mu2 * (v + bv)                        # :411
        x = invLS(r)                                                     # :413
        bu = bu + (u - x)                                                # :420
        bv = bv + (v - x)                                                # :421
        resid = y - A(x)                                                 # :423
        numA += 1
        objective.append(0.5 * float(np.sum(resid * resid)) + tau1 * TVnorm(u) + tau2 * TVnorm(v))   # :425
        if true_x is not None:
            e = x - true_x
            mses.append(float(np.sum(e * e)) / x.size)                   # :428-429
        nx2 = float(np.sum(x * x))
        distance.append((float(np.linalg.norm((x - u).ravel())) / math.sqrt(nx2 + float(np.sum(u * u))),
                         float(np.linalg.norm((x - v).ravel())) / math.sqrt(nx2 + float(np.sum(v * v)))))   # :432-433
        if outer > 1:                                                    # :435
            if stopcriterion == 1:
                crit = abs(objective[outer] - objective[outer - 1]) / objective[outer - 1]   # :441
            elif stopcriterion == 2:
                crit = abs(float(np.linalg.norm((x - xprev).ravel())) / float(np.linalg.norm(x.ravel())))
            else:
                crit = objective[outer]
            criterion.append(crit)
            if crit < tolA:                                              # :453
                break
    return dict(x=x, numA=numA, numAt=numAt, objective=np.array(objective), distance=np.array(distance),
                mses=np.array(mses), n_outer=n_outer, u=u, v=v, criterion=np.array(criterion))


def my_fista(b, A, AT, tau, L, Phi, Psi, stopcriterion, tolerance, maxiters, true, x0=None):
    """SALSA/my_fista.m:5-56.  `x0=None` -> x = AT(b) (:7); passing zeros gives
    the my_deblur_fista.m:21 start."""
    x = AT(b) if x0 is None else np.array(x0, dtype=np.float64)
    yv = x.copy()
    t = 1.0
    Ax = A(x)
    objective = [0.5 * float(np.sum((Ax - b) ** 2)) + tau * Phi(x)]
    mses = [float(np.sum((x - true) ** 2)) / true.size]
    k = 1
    for k in range(2, int(maxiters) + 1):
        x_old = x
        t_old = t
        yv = yv - (1.0 / L) * AT(A(yv) - b)                  # :25
        x = Psi(yv, tau / L)                                 # :26
        t = 0.5 * (1 + math.sqrt(1 + 4 * t_old ** 2))        # :28
        yv = x + ((t_old - 1) / t) * (x - x_old)             # :29
        Ax = A(x)
        objective.append(0.5 * float(np.sum((Ax - b) ** 2)) + tau * Phi(x))           # :32
        mses.append(float(np.sum((x - true) ** 2)) / true.size)
        if stopcriterion == 1:
            criterion = abs(objective[-1] - objective[-2]) / objective[-1]            # :38
        elif stopcriterion == 2:
            criterion = float(np.linalg.norm(x - x_old)) / math.sqrt(float(np.sum(x ** 2)))
        elif stopcriterion == 3:
            criterion = objective[-1]
        else:
            raise ValueError("Invalid stopping criterion!")
        if criterion < tolerance:
            break
    return dict(x=x, objective=np.array(objective), mses=np.array(mses), n_iter=k)


def my_deblur_fista(b, h, tau, Phi_TV, Psi_TV, stopcriterion, tolerance, maxiters, true):
    """SALSA/my_deblur_fista.m:5-68 — full-size kernel h, L = 1, zero start."""
    H = fft2(h)
    HC = np.conj(H)
    A = lambda x: np.real(ifft2(H * fft2(x)))
    AT = lambda x: np.real(ifft2(HC * fft2(x)))
    return my_fista(b, A, AT, tau, 1.0, Phi_TV, Psi_TV, stopcriterion, tolerance, maxiters, true,
                    x0=np.zeros_like(b))


# --------------------------------------------------------------------------
# L4/L5: demo set-up and SAPG (MYULA) loops
# --------------------------------------------------------------------------

DEMO = {
    # constants of run_*_demo.m
    "gaussian": dict(true=(0.4, 0.3), init=(0.4, 0.3), pmin=(0.1, 0.1), pmax=(1.0, 1.0), fix=(True, True),
                     ev_params=(1.0, 1.0), bsnr_min=15, bsnr_max=45, lambdaMax=2.0, gammaFrac=0.98,
                     lf="min", gamma_mult=1.0, c_theta=0.01, c_p=(10.0, 10.0), c_sigma=1000.0),
    "moffat": dict(true=(0.4, 3.5), init=(1.0, 10.0), pmin=(1e-2, 0.1), pmax=(1.0, 10.0), fix=(False, False),
                   ev_params=(1.0, 5.0), bsnr_min=18, bsnr_max=35, lambdaMax=2.0, gammaFrac=0.98,
                   lf="min", gamma_mult=1.0, c_theta=0.1, c_p=(10.0, 1e4), c_sigma=1e4),
    "laplace": dict(true=(0.3,), init=(0.1,), pmin=(1e-3,), pmax=(1.0,), fix=(False,),
                    ev_params=(1.0,), bsnr_min=15, bsnr_max=45, lambdaMax=0.1, gammaFrac=0.98,
                    lf="max", gamma_mult=10.0, c_theta=0.01, c_p=(100.0,), c_sigma=1e4),
}


def demo_setup(kind, x, noise, evMax, BSNR=30.0, true_params=None, th_init=0.01):
    """Data synthesis + step sizes of run_Gaussian_demo.m:145-184,
    run_moffat_demo.m:139-176, run_laplace_demo.m:109-142."""
    d = DEMO[kind]
    p_true = tuple(d["true"] if true_params is None else true_params)
    model = BlurModel(kind, x.shape)
    dimX = x.size
    Ax = model.A(x, *p_true)
    nrm = float(np.linalg.norm(Ax - np.mean(np.mean(Ax, axis=0)), "fro"))
    sigma = nrm / math.sqrt(dimX * 10 ** (BSNR / 10))
    sigma_min = nrm / math.sqrt(dimX * 10 ** (d["bsnr_min"] / 10))
    sigma_max = nrm / math.sqrt(dimX * 10 ** (d["bsnr_max"] / 10))
    y = Ax + sigma * noise
    lf = lambda s2: evMax ** 2 / s2
    pick = min if d["lf"] == "min" else max
    Lf = pick(lf(sigma_min ** 2), lf(sigma_max ** 2))
    lam = min(5 / Lf, d["lambdaMax"])
    gamma_max = 1 / (Lf + 1 / lam)
    gamma = d["gamma_mult"] * d["gammaFrac"] * gamma_max
    return dict(kind=kind, model=model, y=y, x=x, sigma=sigma, sigma_min=sigma_min ** 2,
                sigma_max=sigma_max ** 2, sigma_init=(sigma_min ** 2 + sigma_max ** 2) / 2,
                Lf=Lf, lam=lam, gamma=gamma, p_true=p_true, th_init=th_init, dimX=dimX,
                d_exp=0.8, d_scale=0.01 / th_init, min_th=1e-3, max_th=1.0)


def SAPG_algorithm(setup, samples, warmup, burnIn, randn, chambolleit=25, p_init=None, fix=None,
                   fix_sigma=False, c=None, X0=None, sigma_init=None, iter_offset=0, keep_X=()):
    """SAPG/SAPG_algorithm_Guassian.m:7-308, SAPG_algorithm_moffat.m:7-297,
    SAPG_algorithm_laplace.m:7-268 (one body; the three files differ only in
    the number of PSF parameters and the step-scale constants).

    `randn(shape)` is called once per MYULA step, warm-up first (the MATLAB
    stream cannot be reproduced: parity is per-step with injected noise).
    Every closure recomputes A(x) exactly like the reference (24 FFTs per
    iteration for two-parameter PSFs, 18 for Laplace).

    X0: the reference's op.X0 (start image, default y; SAPG_algorithm_Guassian.m:10-12).  sigma_init overrides the
    demo's (sigma_min + sigma_max) / 2.  Not in the reference (test infrastructure for re-anchored segments of a long
    chain, include/sbtv.h `iter_offset`): iter_offset shifts the step, delta(ii + iter_offset); keep_X = 1-based
    iteration numbers whose sample X is returned in `X_at` (iteration 1 = the state the SAPG loop starts from)."""
    kind = setup["kind"]
    d = DEMO[kind]
    model = setup["model"]
    y = setup["y"]
    dimX = setup["dimX"]
    lamb = setup["lam"]
    gam = setup["gamma"]
    npar = len(d["true"])
    p_init = tuple(d["init"] if p_init is None else p_init)
    fixp = tuple(d["fix"] if fix is None else fix)
    c_theta = d["c_theta"] if c is None else c["theta"]
    c_p = d["c_p"] if c is None else c["p"]
    c_sigma = d["c_sigma"] if c is None else c["sigma"]
    p_true = setup["p_true"]
    min_sigma = min(setup["sigma_min"], setup["sigma_max"])
    max_sigma = max(setup["sigma_min"], setup["sigma_max"])
    if sigma_init is None:
        sigma_init = setup["sigma"] ** 2 if fix_sigma else setup["sigma_init"]

    A, AT = model.A, model.AT
    f = lambda x, p, s2: float(np.linalg.norm(y - A(x, *p), "fro")) ** 2 / (2 * s2)
    gradF = lambda x, p, s2: np.real(AT(A(x, *p) - y, *p) / s2)
    grad_p = lambda i, x, p, s2: float(np.real(np.sum(model.dA(i, x, *p) * (A(x, *p) - y)) / s2))
    gradF_sigma = lambda x, p, s2: float(np.linalg.norm(y - A(x, *p), "fro")) ** 2 / (2 * s2 ** 2) - dimX / (2 * s2)
    proxG = lambda x, theta: chambolle_prox_TV_stop(x, lam=lamb * theta, maxiter=chambolleit)[0]
    logPi = lambda x, theta, p, s2: -f(x, p, s2) - theta * TVnorm(x)
    delta = lambda i: setup["d_scale"] * (((i + iter_offset) ** (-setup["d_exp"])) / dimX)
    X_at = {}

    # ---- warm-up (SAPG_algorithm_Guassian.m:66-93)
    X = y.copy() if X0 is None else np.array(X0, dtype=np.float64)
    logPiTrace_WU = np.zeros(max(warmup, 1))
    if warmup > 0:
        prox = proxG(X, setup["th_init"])
        for ii in range(2, warmup + 1):
            X = np.abs(X + gam * (prox - X) / lamb - gam * gradF(X, p_init, sigma_init)
                       + math.sqrt(2 * gam) * randn(X.shape))
            prox = proxG(X, setup["th_init"])
            logPiTrace_WU[ii - 1] = logPi(X, setup["th_init"], p_init, sigma_init)

    # ---- SAPG loop (:158-248)
    thetas = np.zeros(samples); thetas[0] = setup["th_init"]
    sigmas = np.zeros(samples); sigmas[0] = sigma_init
    ps = np.zeros((npar, samples)); ps[:, 0] = p_init
    logPiTraceX = np.zeros(samples)
    gX = np.zeros(samples)
    grads = np.zeros((npar + 2, samples))
    logPiTraceX[0] = logPi(X, thetas[0], tuple(ps[:, 0]), sigmas[0])
    prox = proxG(X, thetas[0])
    if 1 in keep_X:
        X_at[1] = X.copy()
    for ii in range(2, samples + 1):
        i0 = ii - 1          # 0-based slot of iteration ii
        pm = tuple(ps[:, i0 - 1])
        Z = randn(X.shape)
        X = np.abs(X + gam * (prox - X) / lamb - gam * gradF(X, pm, sigmas[i0 - 1]) + math.sqrt(2 * gam) * Z)
        prox = proxG(X, thetas[i0 - 1])
        G_t = dimX / thetas[i0 - 1] - TVnorm(X)
        thetas[i0] = min(max(thetas[i0 - 1] + c_theta * delta(ii) * G_t, setup["min_th"]), setup["max_th"])
        for q in range(npar):
            G = grad_p(q, X, pm, sigmas[i0 - 1])
            grads[1 + q, i0] = G
            pq = p_true[q] if fixp[q] else ps[q, i0 - 1] - c_p[q] * delta(ii) * G
            ps[q, i0] = min(max(pq, d["pmin"][q]), d["pmax"][q])
        G_s = gradF_sigma(X, pm, sigmas[i0 - 1])
        s_new = sigma_init if fix_sigma else sigmas[i0 - 1] + c_sigma * delta(ii) * G_s
        sigmas[i0] = min(max(s_new, min_sigma), max_sigma)
        grads[0, i0] = G_t
        grads[npar + 1, i0] = G_s
        logPiTraceX[i0] = logPi(X, thetas[i0 - 1], pm, sigmas[i0 - 1])
        gX[i0 - 1] = TVnorm(X)
        if ii in keep_X:
            X_at[ii] = X.copy()
    b0 = int(burnIn) - 1
    return dict(theta_EB=float(np.mean(thetas[b0:])), p_EB=[float(np.mean(ps[q, b0:])) for q in range(npar)],
                sigma_EB=float(np.mean(sigmas[b0:])), thetas=thetas, ps=ps, sigmas=sigmas,
                logPiTraceX=logPiTraceX, logPiTrace_WU=logPiTrace_WU, gXTrace=gX, grads=grads,
                Xlast_sample=X, prox_last=prox, err_psf=err_psf_trace(kind, ps, p_true, model.psf_size), X_at=X_at)


def SAPG_algorithm_shared(setup, chains, samples, warmup, burnIn, randn, chambolleit=25, p_init=None, fix=None,
                          fix_sigma=False, c=None):
    """BASELINE.json configs[4] / SURVEY.md §8e row 2: `chains` MYULA chains on ONE image whose gradients are
    averaged before the (common) parameter update.  This is the reference's multi-sample update
    `g_*(jj) = ...; G_* = mean(g_*)` (SAPG/SAPG_algorithm_moffat.m:143-173, SAPG_algorithm_laplace.m:146-166,
    there with `for jj = 1:1`) with the samples drawn by `chains` independent chains instead of one: every chain
    has its own X, prox and noise, all chains share thetas / PSF parameters / sigmas.  Everything else follows
    SAPG_algorithm (same line citations).  `randn(shape, chain)` is called once per MYULA step and chain,
    chains in order 0..chains-1 inside a step, warm-up first.  mean() = sequential sum / n (MATLAB mean)."""
    kind = setup["kind"]
    d = DEMO[kind]
    model = setup["model"]
    y = setup["y"]
    dimX = setup["dimX"]
    lamb = setup["lam"]
    gam = setup["gamma"]
    npar = len(d["true"])
    p_init = tuple(d["init"] if p_init is None else p_init)
    fixp = tuple(d["fix"] if fix is None else fix)
    c_theta = d["c_theta"] if c is None else c["theta"]
    c_p = d["c_p"] if c is None else c["p"]
    c_sigma = d["c_sigma"] if c is None else c["sigma"]
    p_true = setup["p_true"]
    min_sigma = min(setup["sigma_min"], setup["sigma_max"])
    max_sigma = max(setup["sigma_min"], setup["sigma_max"])
    sigma_init = setup["sigma"] ** 2 if fix_sigma else setup["sigma_init"]
    A, AT = model.A, model.AT
    f = lambda x, p, s2: float(np.linalg.norm(y - A(x, *p), "fro")) ** 2 / (2 * s2)
    gradF = lambda x, p, s2: np.real(AT(A(x, *p) - y, *p) / s2)
    grad_p = lambda i, x, p, s2: float(np.real(np.sum(model.dA(i, x, *p) * (A(x, *p) - y)) / s2))
    gradF_sigma = lambda x, p, s2: float(np.linalg.norm(y - A(x, *p), "fro")) ** 2 / (2 * s2 ** 2) - dimX / (2 * s2)
    proxG = lambda x, theta: chambolle_prox_TV_stop(x, lam=lamb * theta, maxiter=chambolleit)[0]
    logPi = lambda x, theta, p, s2: -f(x, p, s2) - theta * TVnorm(x)
    delta = lambda i: setup["d_scale"] * ((i ** (-setup["d_exp"])) / dimX)
    mean = lambda v: sum(v[1:], v[0]) / len(v)

    C = int(chains)
    Xs = [y.copy() for _ in range(C)]
    proxs = [None] * C
    if warmup > 0:
        for k in range(C):
            proxs[k] = proxG(Xs[k], setup["th_init"])
        for ii in range(2, warmup + 1):
            for k in range(C):
                Xs[k] = np.abs(Xs[k] + gam * (proxs[k] - Xs[k]) / lamb - gam * gradF(Xs[k], p_init, sigma_init)
                               + math.sqrt(2 * gam) * randn(y.shape, k))
                proxs[k] = proxG(Xs[k], setup["th_init"])
    thetas = np.zeros(samples); thetas[0] = setup["th_init"]
    sigmas = np.zeros(samples); sigmas[0] = sigma_init
    ps = np.zeros((npar, samples)); ps[:, 0] = p_init
    logPiTraceX = np.zeros((C, samples))
    gX = np.zeros((C, samples))
    grads = np.zeros((npar + 2, samples))
    for k in range(C):
        logPiTraceX[k, 0] = logPi(Xs[k], thetas[0], tuple(ps[:, 0]), sigmas[0])
        proxs[k] = proxG(Xs[k], thetas[0])
    for ii in range(2, samples + 1):
        i0 = ii - 1
        pm = tuple(ps[:, i0 - 1])
        g_t, g_s = [0.0] * C, [0.0] * C
        g_p = [[0.0] * C for _ in range(npar)]
        for k in range(C):                                   # the `for jj` loop, one sample per chain
            Z = randn(y.shape, k)
            Xs[k] = np.abs(Xs[k] + gam * (proxs[k] - Xs[k]) / lamb - gam * gradF(Xs[k], pm, sigmas[i0 - 1])
                           + math.sqrt(2 * gam) * Z)
            proxs[k] = proxG(Xs[k], thetas[i0 - 1])
            for q in range(npar):
                g_p[q][k] = grad_p(q, Xs[k], pm, sigmas[i0 - 1])
            g_s[k] = gradF_sigma(Xs[k], pm, sigmas[i0 - 1])
            g_t[k] = dimX / thetas[i0 - 1] - TVnorm(Xs[k])
        G_t, G_s = mean(g_t), mean(g_s)
        thetas[i0] = min(max(thetas[i0 - 1] + c_theta * delta(ii) * G_t, setup["min_th"]), setup["max_th"])
        for q in range(npar):
            G = mean(g_p[q])
            grads[1 + q, i0] = G
            pq = p_true[q] if fixp[q] else ps[q, i0 - 1] - c_p[q] * delta(ii) * G
            ps[q, i0] = min(max(pq, d["pmin"][q]), d["pmax"][q])
        s_new = sigma_init if fix_sigma else sigmas[i0 - 1] + c_sigma * delta(ii) * G_s
        sigmas[i0] = min(max(s_new, min_sigma), max_sigma)
        grads[0, i0] = G_t
        grads[npar + 1, i0] = G_s
        for k in range(C):
            logPiTraceX[k, i0] = logPi(Xs[k], thetas[i0 - 1], pm, sigmas[i0 - 1])
            gX[k, i0 - 1] = TVnorm(Xs[k])
    b0 = int(burnIn) - 1
    return dict(theta_EB=float(np.mean(thetas[b0:])), p_EB=[float(np.mean(ps[q, b0:])) for q in range(npar)],
                sigma_EB=float(np.mean(sigmas[b0:])), thetas=thetas, ps=ps, sigmas=sigmas,
                logPiTraceX=logPiTraceX, gXTrace=gX, grads=grads, Xlast_samples=Xs)


def err_psf_trace(kind, ps, p_true, psf_size=7):
    """The PSF-tracking trace `results.err_psf`: l2 (squared SPECTRAL norm, quirk Q9) between the PSF at the current
    parameters and the true one.  Per family:
      gaussian  err(1) = l2(psf(w1s(1), w2s(1)), true); err(ii) = l2(psf(w1s(ii), w2s(ii-1)), true)  -- the w2 of the
                PREVIOUS iteration, quirk Q8 (SAPG_algorithm_Guassian.m:144-146,203-204)
      moffat    the initial value is stored under another name (`psf_err(1)`, SAPG_algorithm_moffat.m:156), so
                err_psf(1) stays 0; err(ii) = l2(psf(alphas(ii), betas(ii)), true)   (:204-205)
      laplace   err(ii) = l2(psf(bs(ii)), true) for every ii   (SAPG_algorithm_laplace.m:134-136,190-191)"""
    ps = np.atleast_2d(np.asarray(ps, dtype=np.float64))
    builder = PSF_TAPS[kind][0]
    true = builder(psf_size, tuple(p_true))
    n = ps.shape[1]
    out = np.zeros(n)
    for i in range(n):
        if kind == "gaussian":
            q = (ps[0, i], ps[1, i - 1] if i > 0 else ps[1, 0])
        else:
            q = tuple(ps[:, i])
        if kind == "moffat" and i == 0:
            continue
        out[i] = l2(builder(psf_size, q), true)
    return out


def myula(op, im, randn):
    """SALSA/myula.m:1-22 (plain MYULA chain at fixed theta / PSF parameter, the last sample is returned).
    op: dict with y, lambda, gamma, theta_op, tau_op, samples, gradF(x, tau), proxG(x, lambda, theta)
    (closures as in SALSA/run_deblur_tv.m:126,131); randn(shape) supplies z (MATLAB's stream is unpinned)."""
    x = np.array(op["y"], dtype=np.float64)                  # :3,11
    lam, gam, theta, tau = op["lambda"], op["gamma"], op["theta_op"], op["tau_op"]
    for ii in range(2, int(op["samples"])):                  # ii = 2 : sample-1   (:13)
        z = randn(im.shape)                                  # :14
        prox = op["proxG"](x, lam, theta)                    # :15
        x = (1 - gam / lam) * x - gam * (op["gradF"](x, tau) - prox / lam) + math.sqrt(2 * gam) * z   # :16
    return x


def salsa_from_estimates(setup, theta_EB, p_EB, sigma2_EB, tol=1e-5, outeriters=500, TViters=10, max_time=None):
    """run_Gaussian_demo.m:210-242 (and the Moffat / Laplace twins)."""
    model = setup["model"]
    p = tuple(p_EB)
    calls = [0]

    def counted(fn):
        def w(v):
            calls[0] += 1
            return fn(v)
        return w
    A1 = counted(lambda v: model.A(v, *p))
    AT1 = counted(lambda v: model.AT(v, *p))
    mu = theta_EB / 10
    filter_FFT = 1.0 / (np.abs(model.H_FFT(*p)) ** 2 + mu)
    invLS = counted(lambda v: np.real(ifft2(filter_FFT * fft2(v))))
    out = SALSA_v2(setup["y"], A1, theta_EB * sigma2_EB, mu=mu, AT=AT1, invLS=invLS, true_x=setup["x"],
                   stopcriterion=1, tolA=tol, maxiter=outeriters, TViters=TViters, max_time=max_time)
    out["calls"] = calls[0]
    return out
