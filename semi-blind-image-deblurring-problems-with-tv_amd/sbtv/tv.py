"""Host mirror of the reference's TV primitives (L1), calling the HIP kernels.

`chambolle_prox_TV_stop(g, 'lambda', L, 'maxiter', K, ...)` keeps the MATLAB
name/value calling convention of utils/chambolle_prox_TV_stop.m:1,87-109
(option names are case-insensitive like the reference's `upper(varargin{i})`).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _parse_varargin(varargin, known):
    if len(varargin) % 2 == 1:
        raise ValueError("Optional parameters should always go by pairs")
    out = {}
    for i in range(0, len(varargin), 2):
        key = str(varargin[i]).upper()
        if key not in known:
            raise ValueError(f"Unrecognized option: '{varargin[i]}'")
        out[key] = varargin[i + 1]
    return out


def chambolle_prox_TV_stop(g, *varargin, ctx=None, return_info=False, **kw):
    """[f, px, py] = chambolle_prox_TV_stop(g, 'lambda', lam, 'maxiter', K,
    'tol', tol, 'tau', tau, 'dualvars', [px py])   (utils/chambolle_prox_TV_stop.m)

    g: (M,N) or (B,M,N) numpy array, or a column-major CUDA tensor (sbtv.to_device).
    'lambda' may be a scalar or one value per image.  'maxiter' is required
    (quirk Q1: the reference's default is dead code and the call errors without it).
    'dualvars' is the M x 2N concatenation [px py]; like the reference the split
    uses M, so it is only meaningful for square images (quirk Q2).
    Keyword form (lam=..., maxiter=...) is accepted too.
    """
    opts = _parse_varargin(varargin, {"LAMBDA", "VERBOSE", "TOL", "MAXITER", "TAU", "DUALVARS"})
    for k, v in kw.items():
        opts[{"lam": "LAMBDA"}.get(k, k.upper())] = v
    ctx = ctx or L.default_context()
    gi = L.Images(g)
    B, M, N = gi.B, gi.M, gi.N
    lam = opts.get("LAMBDA", 1.0)          # :79
    tol = float(opts.get("TOL", 1e-3))     # :78
    tau = float(opts.get("TAU", 0.249))    # :77
    if "MAXITER" not in opts:
        # MATLAB: "Undefined function or variable 'MaxIter'" (:131)
        raise L.SbtvError(-3, "chambolle_prox_TV_stop: 'maxiter' is required (MaxIter undefined, quirk Q1)")
    maxiter = int(opts["MAXITER"])
    lam_a, lam_p = L.dvec(lam, B)
    px = L.empty_like_images(gi)
    py = L.empty_like_images(gi)
    f = L.empty_like_images(gi)
    warm = 0
    if "DUALVARS" in opts and opts["DUALVARS"] is not None:
        dv = opts["DUALVARS"]
        if gi.torch:
            pxi, pyi = dv          # device path: a (px, py) pair of column-major tensors
            px, py = L.Images(pxi.clone()), L.Images(pyi.clone())
        else:
            dv = np.asarray(dv, dtype=np.float64)
            if dv.ndim == 2:
                dv = dv[None]
            if dv.shape[1] != M or dv.shape[2] != 2 * N:
                raise L.SbtvError(-4, "Wrong size of the dual variables")
            px = L.Images(dv[:, :, :M])          # px = px(:,1:M)      (:107)
            py = L.Images(dv[:, :, M:])          # py = px(:,M+1:end)  (:106)
            if px.N != N or py.N != N:
                raise L.SbtvError(-4, "Wrong size of the dual variables (non-square image, quirk Q2)")
            px.squeeze = py.squeeze = gi.squeeze
        warm = 1
    k = (C.c_int * B)()
    err = (C.c_double * B)()
    ctx.check(ctx.lib.sbtv_chambolle_prox_TV_stop(ctx.h, gi.ptr, M, N, B, lam_p, maxiter, tol, tau, warm,
                                                  px.ptr, py.ptr, f.ptr, k, err, gi.flags), gi.flags)
    sq = (g.dim() == 2) if gi.torch else gi.squeeze
    res = (L.images_result(f, sq), L.images_result(px, sq), L.images_result(py, sq))
    if return_info:
        return res + (np.array(k[:]), np.array(err[:]))
    return res


def TVnorm(x, ctx=None):
    """y = TVnorm(x)  (utils/TVnorm.m:1-2): isotropic TV with periodic differences."""
    ctx = ctx or L.default_context()
    xi = L.Images(x)
    out = (C.c_double * xi.B)()
    ctx.check(ctx.lib.sbtv_TVnorm(ctx.h, xi.ptr, xi.M, xi.N, xi.B, out, xi.flags), xi.flags)
    if xi.B == 1 and (getattr(xi, "squeeze", False) or (xi.torch and x.dim() == 2)):
        return float(out[0])
    return np.array(out[:])
