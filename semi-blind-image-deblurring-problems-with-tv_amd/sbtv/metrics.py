"""utils/PSNR.m and utils/MSE.m on the device."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _metric(fn_name, x_true, x, ctx=None):
    ctx = ctx or L.default_context()
    a, b = L.Images(x_true), L.Images(x)
    if (a.B, a.M, a.N) != (b.B, b.M, b.N) or a.flags != b.flags:
        raise ValueError("x_true and x must have the same shape and memory space")
    out = (C.c_double * a.B)()
    ctx.check(getattr(ctx.lib, fn_name)(ctx.h, a.ptr, b.ptr, a.M, a.N, a.B, out, a.flags), a.flags)
    return float(out[0]) if a.B == 1 else np.array(out[:])


def PSNR(x, y, ctx=None):
    """psnr = PSNR(x, y): 10log10(max(x)^2) - 10log10(||x-y||^2/n)   (utils/PSNR.m:2-4)."""
    return _metric("sbtv_PSNR", x, y, ctx)


def MSE(x_true, x_app, ctx=None):
    """mse = MSE(x_true, x_app) in dB   (utils/MSE.m:1-4)."""
    return _metric("sbtv_MSE", x_true, x_app, ctx)
