"""Host mirrors of the two other ADMM front-ends that use the TV prox:
`csalsa` (SALSA/CSALSA_v2.m:160-561) and `CoRAL` (SALSA/CoRAL_v2.m:2-476).  The reference's demos never
call them; they run on the same device kernels as SALSA_v2."""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib as L
from .operators import BlurOperator, _Adjoint, _InvLS
from .tv import _parse_varargin

_CSALSA_OPTIONS = {"P", "PT", "PSI", "PHI", "TVINITIALIZATION", "TVITERS", "STOPCRITERION", "TOLERANCEA", "MAXITERA",
                   "INITIALIZATION", "TRUE_X", "AT", "LS", "VERBOSE", "CONTINUATIONFACTOR", "EPSILON", "SEED"}
_CORAL_OPTIONS = {"W", "WT", "P1", "P1T", "P2", "P2T", "PSI1", "PHI1", "TVINITIALIZATION1", "TVITERS1", "PSI2", "PHI2",
                  "TVINITIALIZATION2", "TVITERS2", "MU1", "MU2", "STOPCRITERION", "TOLERANCEA", "INNERITERS",
                  "MAXITERA", "INITIALIZATION", "TRUE_X", "AT", "VERBOSE", "LS", "SEED"}

_vp = L.vptr


def _common(y, A, opts, ctx, default_stop):
    if not isinstance(A, BlurOperator):
        raise TypeError("A must be a sbtv.BlurOperator (matrix / generic handle A is not supported on the GPU path)")
    AT = opts.get("AT", 0)
    if not isinstance(AT, _Adjoint) or AT.op is not A:
        raise L.SbtvError(-8, "The function handle for transpose of A is missing")
    so = L.sbtv_salsa_opts()
    ctx.lib.sbtv_salsa_opts_default(C.byref(so))
    so.stopcriterion = int(opts.get("STOPCRITERION", default_stop))
    if so.stopcriterion not in (1, 2, 3):
        raise L.SbtvError(-6, "Unknown stopping criterion")
    so.maxiter = int(opts.get("MAXITERA", 10000))
    so.tolA = float(opts.get("TOLERANCEA", 0.001))
    yi = L.Images(y)
    init = opts.get("INITIALIZATION", 0)
    xinit = None
    if np.ndim(init) > 0 or L._is_torch(init):
        xinit = L.Images(init)
        so.initialization = 33333
    else:
        so.initialization = int(init)
        if so.initialization == 1:
            # random start (CSALSA_v2.m:382, CoRAL_v2.m:331): NumPy normals (option 'SEED'), handed over as a given x
            x0 = np.random.default_rng(int(opts.get("SEED", 0))).standard_normal((yi.B, yi.M, yi.N))
            xinit = L.Images(L.to_device(x0, yi.t.device) if yi.torch else x0)
            so.initialization = 33333
        elif so.initialization not in (0, 2):
            raise L.SbtvError(-7, "Unknown 'Initialization' option")
    true = opts.get("TRUE_X", None)
    ti = L.Images(true) if true is not None else None
    if ti is not None and (ti.M, ti.N) != (yi.M, yi.N):
        raise ValueError("Initial x has incompatible size")
    so.compute_mse = 1 if ti is not None else 0
    for other in (xinit, ti):
        if other is not None and other.flags != yi.flags:
            raise ValueError("all image arguments must live in the same memory space")
    return so, yi, xinit, ti


def csalsa(y, A, mu1, mu2, sigma, *varargin, ctx=None, **kw):
    """[x, numA, numAt, objective, distance1, distance2, criterion, times, mses] = csalsa(y, A, mu1, mu2, sigma, ...)

    Options as in SALSA/CSALSA_v2.m:200-250.  'LS' is the two-argument handle invLS(r, mu) (:310,471): pass
    `A.invLS` of the same BlurOperator.  Only 'TVINITIALIZATION' = 1 runs on the GPU.  Traces are 1-based in
    the reference; the returned arrays hold entries 1..outer in positions 0..outer-1."""
    opts = _parse_varargin(varargin, _CSALSA_OPTIONS)
    for k, v in kw.items():
        opts[k.upper()] = v
    ctx = ctx or L.default_context()
    so, yi, xinit, ti = _common(y, A, opts, ctx, 3)                                           # default :171
    LS = opts.get("LS", None)
    if getattr(LS, "__self__", None) is not A or getattr(LS, "__func__", None) is not BlurOperator.invLS:
        raise L.SbtvError(-9, "(A^T A + \\mu I)^(-1) must be specified as a function handle.")   # :308
    if ("P" in opts) != ("PT" in opts):
        raise ValueError("If you give P you must also give PT, and vice versa.")               # :265
    if "P" in opts:
        raise NotImplementedError("'P'/'PT' other than the identity are not supported")
    if not opts.get("TVINITIALIZATION", 0):
        raise NotImplementedError("only the TV path ('TVINITIALIZATION', 1) runs on the GPU")
    if "PSI" in opts:
        print("Warning: user specified Phi and Psi will not be used as TV with initialization flag has been set to 1.",
              file=sys.stdout)                                                                 # :333
    so.TViters = int(opts.get("TVITERS", 5))
    B, M, N = yi.B, yi.M, yi.N
    K = so.maxiter
    xo = L.empty_like_images(yi)
    tr = {k: np.zeros((B, K)) for k in ("objective", "distance1", "distance2", "criterion", "times", "mses")}
    numA, numAt, nout = (C.c_int * B)(), (C.c_int * B)(), (C.c_int * B)()
    taps = A._cm(B)
    keep = [L.dvec(v, B) for v in (mu1, mu2, sigma, opts.get("EPSILON", 0.0))]     # (array, pointer) pairs stay alive
    m1, m2, sg, ep = (k[1] for k in keep)
    args = (ctx.h, yi.ptr, M, N, B, _vp(taps), A.taille, m1, m2, sg, ep, float(opts.get("CONTINUATIONFACTOR", 1.0)),
            C.byref(so), ti.ptr if ti else None, xinit.ptr if xinit else None, xo.ptr, _vp(tr["objective"]),
            _vp(tr["distance1"]), _vp(tr["distance2"]), _vp(tr["criterion"]), _vp(tr["times"]),
            _vp(tr["mses"]) if ti else None, numA, numAt, nout)
    if getattr(ctx, "is_group", False):          # sbtv.Group: images dealt to the devices in contiguous blocks
        if yi.flags != L.SBTV_HOST_PTRS:
            raise ValueError("a sbtv.Group takes host (NumPy) images")
        ctx.check(ctx.lib.sbtv_CSALSA_v2_sharded(*args))
    else:
        ctx.check(ctx.lib.sbtv_CSALSA_v2(*args, yi.flags), yi.flags)
    sq = (y.dim() == 2) if yi.torch else yi.squeeze
    x = L.images_result(xo, sq)
    n = np.array(nout[:])
    cut = lambda name, b: tr[name][b, :n[b]].copy()
    names = ("objective", "distance1", "distance2", "criterion", "times")
    if sq or B == 1:
        return (x, int(numA[0]), int(numAt[0])) + tuple(cut(k, 0) for k in names) + \
               ((cut("mses", 0) if ti else np.array([])),)
    return (x, np.array(numA[:]), np.array(numAt[:])) + tuple([cut(k, b) for b in range(B)] for k in names) + \
           (([cut("mses", b) for b in range(B)] if ti else []),)


CSALSA_v2 = csalsa


def CoRAL(y, A, tau1, tau2, *varargin, ctx=None, **kw):
    """[x, numA, numAt, objective, distance, times, mses] = CoRAL(y, A, tau1, tau2, ...)

    Options as in SALSA/CoRAL_v2.m:52-128.  'LS' is `A.LS(mu)`; the reference's convention is mu = mu1+mu2
    (:137).  Only the TV/TV configuration ('TVINITIALIZATION1' = 'TVINITIALIZATION2' = 1) runs on the GPU.
    distance is (outer, 2) like the reference (:432-433)."""
    opts = _parse_varargin(varargin, _CORAL_OPTIONS)
    for k, v in kw.items():
        opts[k.upper()] = v
    ctx = ctx or L.default_context()
    so, yi, xinit, ti = _common(y, A, opts, ctx, 1)
    LS = opts.get("LS", None)
    if not isinstance(LS, _InvLS) or LS.op is not A:
        raise L.SbtvError(-9, "(A^T A + \\mu I)^(-1) must be specified as a function handle.")   # :197
    for a, b in (("P1", "P1T"), ("P2", "P2T")):
        if (a in opts) != (b in opts):
            raise ValueError(f"If you give {a} you must also give {b}, and vice versa.")       # :143,153
        if a in opts:
            raise NotImplementedError("'P1'/'P2' other than the identity are not supported")
    if not (opts.get("TVINITIALIZATION1", 0) and opts.get("TVINITIALIZATION2", 0)):
        raise NotImplementedError("only the TV/TV configuration runs on the GPU")
    print("WARNING: TV with initialization has been specified for both Phi1 and Phi2. Try reformulating with a "
          "single regularizer term for efficiency.", file=sys.stdout)                          # :224
    so.TViters = int(opts.get("TVITERS1", 5))
    B, M, N = yi.B, yi.M, yi.N
    K = so.maxiter
    xo = L.empty_like_images(yi)
    objective, times, mses = np.zeros((B, K + 1)), np.zeros((B, K + 1)), np.zeros((B, K + 1))
    distance = np.zeros((B, K, 2))
    numA, numAt, nout = (C.c_int * B)(), (C.c_int * B)(), (C.c_int * B)()
    taps = A._cm(B)
    keep = [L.dvec(v, B) for v in (tau1, tau2, opts.get("MU1", 1e-3), opts.get("MU2", 1e-3), LS.mu)]
    t1, t2, m1, m2, ml = (k[1] for k in keep)
    args = (ctx.h, yi.ptr, M, N, B, _vp(taps), A.taille, t1, t2, m1, m2, ml, int(opts.get("TVITERS2", 5)), C.byref(so),
            ti.ptr if ti else None, xinit.ptr if xinit else None, xo.ptr, _vp(objective), _vp(distance), _vp(times),
            _vp(mses) if ti else None, numA, numAt, nout)
    if getattr(ctx, "is_group", False):          # sbtv.Group: images dealt to the devices in contiguous blocks
        if yi.flags != L.SBTV_HOST_PTRS:
            raise ValueError("a sbtv.Group takes host (NumPy) images")
        ctx.check(ctx.lib.sbtv_CoRAL_v2_sharded(*args))
    else:
        ctx.check(ctx.lib.sbtv_CoRAL_v2(*args, yi.flags), yi.flags)
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


CoRAL_v2 = CoRAL
