"""Host mirror of SALSA_v2 (SALSA/SALSA_v2.m:156-494) over the device-resident loop."""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib as L
from .operators import BlurOperator, _Adjoint, _InvLS
from .tv import _parse_varargin

_OPTIONS = {"P", "PT", "PSI", "PHI", "TVINITIALIZATION", "TVITERS", "MU", "STOPCRITERION", "TOLERANCEA",
            "MAXITERA", "INITIALIZATION", "TRUE_X", "AT", "VERBOSE", "LS", "SEED", "SPECULATE"}


def SALSA_v2(y, A, tau, *varargin, ctx=None, **kw):
    """[x, numA, numAt, objective, distance, times, mses] = SALSA_v2(y, A, tau, ...)

    Same name/value options as SALSA/SALSA_v2.m:196-241 (case-insensitive):
    'MU', 'AT', 'LS', 'STOPCRITERION', 'TOLERANCEA', 'MAXITERA', 'TRUE_X',
    'INITIALIZATION', 'TVINITIALIZATION', 'TVITERS', 'PSI', 'PHI', 'VERBOSE', 'P', 'PT'.

    `A` is a sbtv.BlurOperator (the FFT closure over PSF taps) with 'AT' its `.T` and 'LS' its `.LS(mu)` - or, as in
    run_Gaussian_demo.m:229-242, three plain function handles: the taps are then recovered by probing A(delta)
    (the reference pads the kernel into the top-left corner, utils/resize.m:8-11), mu by probing LS(delta), and
    'AT' / 'LS' are checked against the recovered operator (three probe calls; the handles themselves are never
    called inside the GPU loop, so a handle that is NOT a compact circular blur is refused, not emulated).
    'MU' must be the mu the 'LS' filter was built with: the C-ABI has one mu (the reference's demos pass the same
    value to both, run_Gaussian_demo.m:219-225); a mismatch raises instead of silently preferring one.
    Only the TV path ('TVINITIALIZATION' = 1) is implemented; as in the
    reference a user 'PSI'/'PHI' is then ignored with a warning (quirk Q7,
    SALSA_v2.m:318-320).  y may be (M,N), a batch (B,M,N) (tau, 'MU' and the
    PSF may then be per image) or a column-major CUDA tensor.
    """
    opts = _parse_varargin(varargin, _OPTIONS)
    for k, v in kw.items():
        opts[k.upper()] = v
    ctx = ctx or L.default_context()
    AT = opts.get("AT", 0)
    LS = opts.get("LS", None)
    mu = opts.get("MU", 1e-3)                                                                 # :176
    if not isinstance(A, BlurOperator):
        if not callable(A):
            raise TypeError("A must be a sbtv.BlurOperator or a function handle (a matrix A is not supported)")
        if not callable(AT):
            raise L.SbtvError(-8, "The function handle for transpose of A is missing")       # SALSA_v2.m:262
        if not callable(LS):
            raise L.SbtvError(-9, "(A^T A + \\mu I)^(-1) must be specified as a function handle.")  # :296
        shape = tuple(y.shape[-2:])
        like = y if L._is_torch(y) else None
        A = BlurOperator.from_handle(A, shape, like=like, ctx=ctx)
        A.check_adjoint_handle(AT, shape, like=like)
        AT, LS = A.T, A.LS(A.mu_of_handle(LS, shape, like=like))
    if not isinstance(AT, _Adjoint) or AT.op is not A:
        raise L.SbtvError(-8, "The function handle for transpose of A is missing")           # SALSA_v2.m:262
    if not isinstance(LS, _InvLS) or LS.op is not A:
        raise L.SbtvError(-9, "(A^T A + \\mu I)^(-1) must be specified as a function handle.")  # :296
    if "MU" not in opts:
        mu = LS.mu
    elif (abs(mu - LS.mu) > 1e-6 * abs(LS.mu)) if (isinstance(mu, float) and isinstance(LS.mu, float)) else \
            not np.allclose(np.asarray(mu, dtype=np.float64), np.asarray(LS.mu, dtype=np.float64), rtol=1e-6, atol=0.0):
        raise L.SbtvError(-9, "'MU' differs from the mu the 'LS' filter was built with: the GPU path has one mu "
                              "(run_Gaussian_demo.m:219-225 passes the same value to both)")
    if ("P" in opts) != ("PT" in opts):
        raise ValueError("If you give P you must also give PT, and vice versa.")               # :252
    if "P" in opts:
        raise NotImplementedError("'P'/'PT' other than the identity are not supported")
    if not opts.get("TVINITIALIZATION", 0):
        raise NotImplementedError("only the TV path ('TVINITIALIZATION', 1) runs on the GPU")
    if "PSI" in opts:
        print("Warning: user specified Phi and Psi will not be used as TV with initialization flag has been set to 1.",
              file=sys.stdout)                                                                 # :320
    so = L.sbtv_salsa_opts()
    ctx.lib.sbtv_salsa_opts_default(C.byref(so))
    so.stopcriterion = int(opts.get("STOPCRITERION", 1))
    if so.stopcriterion not in (1, 2, 3):
        raise L.SbtvError(-6, "Unknown stopping criterion")                                   # :246
    so.maxiter = int(opts.get("MAXITERA", 10000))
    so.TViters = int(opts.get("TVITERS", 5))
    so.tolA = float(opts.get("TOLERANCEA", 0.001))
    # extension (no counterpart in the reference): sbtv_salsa_opts.speculate, bit 0 host-side lag, bit 1 exact prox launches
    so.speculate = int(opts.get("SPECULATE", so.speculate))
    yi = L.Images(y)
    B, M, N = yi.B, yi.M, yi.N
    init = opts.get("INITIALIZATION", 0)
    xinit = None
    if np.ndim(init) > 0 or L._is_torch(init):
        xinit = L.Images(init)
        so.initialization = 33333                                                             # :221
    else:
        so.initialization = int(init)
        if so.initialization == 1:
            # x = randn(size(AT(zeros(size(y)))))  (:371).  MATLAB's stream cannot be reproduced: the start image is
            # drawn with NumPy (seed via the extra option 'SEED', default 0) and handed over like a given initial x
            rng = np.random.default_rng(int(opts.get("SEED", 0)))
            x0 = rng.standard_normal((yi.B, yi.M, yi.N))
            xinit = L.Images(L.to_device(x0, yi.t.device) if yi.torch else x0)
            so.initialization = 33333
        elif so.initialization not in (0, 2):
            raise L.SbtvError(-7, "Unknown 'Initialization' option")                         # :382
    true = opts.get("TRUE_X", None)
    ti = L.Images(true) if true is not None else None
    if ti is not None and (ti.M, ti.N) != (M, N):
        raise ValueError("Initial x has incompatible size")                                   # :386
    so.compute_mse = 1 if ti is not None else 0
    for other in (xinit, ti):
        if other is not None and other.flags != yi.flags:
            raise ValueError("all image arguments must live in the same memory space")
    xo = L.empty_like_images(yi)
    K = so.maxiter
    objective = np.zeros((B, K + 1))
    distance = np.zeros((B, K))
    times = np.zeros((B, K + 1))
    mses = np.zeros((B, K + 1))
    numA = (C.c_int * B)()
    numAt = (C.c_int * B)()
    nout = (C.c_int * B)()
    taps = A._cm(B)
    tau_a, tau_p = L.dvec(tau, B)
    mu_a, mu_p = L.dvec(mu, B)
    vp = L.vptr
    if getattr(ctx, "is_group", False):
        # several GPUs behind this process (sbtv.Group): images dealt to the devices in contiguous blocks
        if yi.flags != L.SBTV_HOST_PTRS:
            raise ValueError("a sbtv.Group takes host (NumPy) images")
        ctx.check(ctx.lib.sbtv_SALSA_v2_sharded(ctx.h, yi.ptr, M, N, B, vp(taps), A.taille, tau_p, mu_p, C.byref(so),
                                                ti.ptr if ti else None, xinit.ptr if xinit else None, xo.ptr,
                                                vp(objective), vp(distance), vp(times), vp(mses) if ti else None,
                                                numA, numAt, nout))
    else:
        ctx.check(ctx.lib.sbtv_SALSA_v2(ctx.h, yi.ptr, M, N, B, vp(taps), A.taille, tau_p, mu_p, C.byref(so),
                                        ti.ptr if ti else None, xinit.ptr if xinit else None, xo.ptr,
                                        vp(objective), vp(distance), vp(times), vp(mses) if ti else None,
                                        numA, numAt, nout, yi.flags), yi.flags)
    sq = (y.dim() == 2) if yi.torch else yi.squeeze
    x = L.images_result(xo, sq)
    n = np.array(nout[:])
    if sq or B == 1:
        k = int(n[0])
        return (x, int(numA[0]), int(numAt[0]), objective[0, :k + 1].copy(), distance[0, :k].copy(),
                times[0, :k + 1].copy(), mses[0, :k + 1].copy() if ti else np.array([]))
    return (x, np.array(numA[:]), np.array(numAt[:]), [objective[b, :n[b] + 1].copy() for b in range(B)],
            [distance[b, :n[b]].copy() for b in range(B)], [times[b, :n[b] + 1].copy() for b in range(B)],
            [mses[b, :n[b] + 1].copy() for b in range(B)] if ti else [])
