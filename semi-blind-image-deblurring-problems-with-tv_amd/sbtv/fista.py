"""Host mirror of SALSA/my_fista.m and SALSA/my_deblur_fista.m over the device-resident loop."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .operators import BlurOperator, _Adjoint
from .tv import TVnorm


class Psi_TV:
    """The demos' TV prox handle  Psi = @(x,th) chambolle_prox_TV_stop(x,'lambda',th,'maxiter',K)
    (run_Gaussian_demo.m:192, run_moffat_demo.m:182): a cold-start Chambolle prox with K iterations."""

    def __init__(self, maxiter=25):
        self.maxiter = int(maxiter)

    def __call__(self, x, th):
        from .tv import chambolle_prox_TV_stop
        return chambolle_prox_TV_stop(x, "lambda", th, "maxiter", self.maxiter)[0]


def my_fista(b, A, AT, tau, L_, Phi, Psi, stopcriterion, tolerance, maxiters, true, verbose=0, ctx=None,
             _zero_start=False, exact_prox=False):
    """[x, objective, times, mses] = my_fista(b,A,AT,tau,L,Phi,Psi,stopcriterion,tolerance,maxiters,true,verbose)
    (SALSA/my_fista.m:5-56).  A: sbtv.BlurOperator, AT: A.T, Phi: sbtv.TVnorm, Psi: sbtv.Psi_TV(K).
    exact_prox=True (extension, SBTV_FISTA_EXACT_PROX): a stop-rule kernel after every Chambolle launch instead of the
    default optimistic launches (same result; the default repeats the call this way if the rule fires inside a prox)."""
    ctx = ctx or L.default_context()
    if not isinstance(A, BlurOperator) or not isinstance(AT, _Adjoint) or AT.op is not A:
        raise TypeError("A must be a sbtv.BlurOperator and AT its .T")
    if Phi is not TVnorm:
        raise TypeError("Phi must be sbtv.TVnorm on the GPU path")
    if not isinstance(Psi, Psi_TV):
        raise TypeError("Psi must be a sbtv.Psi_TV (Chambolle TV prox) on the GPU path")
    if stopcriterion not in (1, 2, 3):
        raise L.SbtvError(-6, "Invalid stopping criterion!")                      # my_fista.m:45
    bi, ti = L.Images(b), L.Images(true)
    if (bi.B, bi.M, bi.N) != (ti.B, ti.M, ti.N) or bi.flags != ti.flags:
        raise ValueError("b and true must have the same shape and memory space")
    B = bi.B
    xo = L.empty_like_images(bi)
    K = int(maxiters)
    objective = np.zeros((B, K))
    mses = np.zeros((B, K))
    nit = (C.c_int * B)()
    taps = A._cm(B)
    tau_a, tau_p = L.dvec(tau, B)
    vp = L.vptr
    import time
    t0 = time.perf_counter()
    if getattr(ctx, "is_group", False):
        # several GPUs behind this process (sbtv.Group): images dealt to the devices in contiguous blocks, no exchange
        if bi.flags != L.SBTV_HOST_PTRS or exact_prox:
            raise ValueError("a sbtv.Group takes host (NumPy) images (and no exact_prox flag)")
        ctx.check(ctx.lib.sbtv_fista_tv_sharded(ctx.h, bi.ptr, bi.M, bi.N, B, vp(taps), A.taille, tau_p, float(L_),
                                                Psi.maxiter, int(stopcriterion), float(tolerance), K,
                                                1 if _zero_start else 0, ti.ptr, xo.ptr, vp(objective), vp(mses), nit))
    else:
        ctx.check(ctx.lib.sbtv_fista_tv(ctx.h, bi.ptr, bi.M, bi.N, B, vp(taps), A.taille, tau_p, float(L_), Psi.maxiter,
                                        int(stopcriterion), float(tolerance), K, 1 if _zero_start else 0, ti.ptr, xo.ptr,
                                        vp(objective), vp(mses), nit,
                                        bi.flags | (L.FISTA_EXACT_PROX if exact_prox else 0)), bi.flags)
    wall = time.perf_counter() - t0
    sq = (b.dim() == 2) if bi.torch else bi.squeeze
    x = L.images_result(xo, sq)
    n = np.array(nit[:])
    times = [np.linspace(0.0, wall, int(k)) for k in n]       # per-iteration cputime is not tracked on the device
    if sq or B == 1:
        k = int(n[0])
        return x, objective[0, :k].copy(), times[0], mses[0, :k].copy()
    return x, [objective[i, :n[i]].copy() for i in range(B)], times, [mses[i, :n[i]].copy() for i in range(B)]


def my_deblur_fista(b, h, tau, Phi_TV, Psi_TV_, stopcriterion, tolerance, maxiters, true, verbose=0, ctx=None):
    """[x, objective, times, mses] = my_deblur_fista(b,h,tau,Phi_TV,Psi_TV,stopcriterion,tolerance,maxiters,true,verbose)
    (SALSA/my_deblur_fista.m:5-68): h is the FULL-SIZE kernel (H_FFT = fft2(h)), L = 1, zero start.
    The kernel's support must lie in the top-left 15 x 15 corner (the demos' resize() layout)."""
    h = np.asarray(h, dtype=np.float64)
    nz = np.argwhere(h != 0)
    t = int(nz.max()) + 1 if nz.size else 1
    if t > 15:
        raise L.SbtvError(-10, "my_deblur_fista: kernel support larger than 15 x 15 is not supported")
    A = BlurOperator(h[:t, :t], ctx=ctx)
    return my_fista(b, A, A.T, tau, 1.0, Phi_TV, Psi_TV_, stopcriterion, tolerance, maxiters, true, verbose, ctx=ctx,
                    _zero_start=True)
