"""Host mirror of SAPG/SAPG_algorithm_{Guassian,moffat,laplace}.m, utils/max_eigenval_*.m and the
data-synthesis / step-size block of run_*_demo.m, over the device-resident MYULA/SAPG loop.

`op` is the reference's option struct (a dict or any attribute bag with the same field names);
its function handles (op.gradF, op.proxG, op.logPi, ...) are not used: the C-ABI rebuilds them
from the PSF family, `y` and the scalar fields.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib as L
from .operators import BlurOperator, psf_family

_KIND = {"gaussian": 0, "moffat": 1, "laplace": 2}


def _get(op, name, default=None):
    if isinstance(op, dict):
        return op.get(name, default)
    return getattr(op, name, default)


def max_eigenval(A, At, params, im_size, tol, max_iter, verbose=0, x0=None, ctx=None):
    """val = max_eigenval(A, At, a, b, im_size, tol, max_iter, verbose)
    (utils/max_eigenval_Gaussian_Moffat.m:1-27, utils/max_eigenval_Laplace.m:1-28).
    A: callable (x, *params) -> BlurOperator application is not needed: pass A as a
    sbtv.BlurOperator built at `params` (At is then ignored).  x0: start vector
    (the reference draws randn(im_size); MATLAB's stream is unpinned -> pass it explicitly)."""
    ctx = ctx or L.default_context()
    if not isinstance(A, BlurOperator):
        raise TypeError("A must be a sbtv.BlurOperator built at the requested parameters")
    if x0 is None:
        x0 = np.random.default_rng(0).standard_normal(im_size)
    xi = L.Images(x0)
    taps = A._cm(1)
    val = C.c_double(0.0)
    it = C.c_int(0)
    ctx.check(ctx.lib.sbtv_max_eigenval(ctx.h, L.vptr(taps), A.taille, xi.ptr, xi.M, xi.N,
                                        float(tol), int(max_iter), C.byref(val), C.byref(it), xi.flags), xi.flags)
    if verbose:
        print(f"Norm = {val.value:e} ({it.value} iterations)")
    return val.value


def demo_setup(kind, x, noise, evMax, BSNR=30.0, true_params=None, BSNR_min=None, BSNR_max=None, lambdaMax=None,
               gammaFrac=0.98, th_init=0.01, ctx=None):
    """The observation model and MYULA step sizes of run_Gaussian_demo.m:145-184,
    run_moffat_demo.m:139-176, run_laplace_demo.m:109-142 (blur on the GPU, scalars on the host).
    Returns a dict with y, sigma, sigma_min/max/init (variances), Lf, lambda, gamma."""
    defaults = {"gaussian": dict(true=(0.4, 0.3), bmin=15, bmax=45, lmax=2.0, lf=min, gmul=1.0),
                "moffat": dict(true=(0.4, 3.5), bmin=18, bmax=35, lmax=2.0, lf=min, gmul=1.0),
                "laplace": dict(true=(0.3,), bmin=15, bmax=45, lmax=0.1, lf=max, gmul=10.0)}[kind]
    p = tuple(defaults["true"] if true_params is None else true_params)
    bmin = defaults["bmin"] if BSNR_min is None else BSNR_min
    bmax = defaults["bmax"] if BSNR_max is None else BSNR_max
    lmax = defaults["lmax"] if lambdaMax is None else lambdaMax
    x = np.asarray(x, dtype=np.float64)
    taps, _ = psf_family(kind, 7, p)
    Ax = BlurOperator(taps, ctx=ctx).A(x)
    dimX = x.size
    nrm = float(np.linalg.norm(Ax - np.mean(Ax)))
    sigma = nrm / math.sqrt(dimX * 10 ** (BSNR / 10))
    smin = nrm / math.sqrt(dimX * 10 ** (bmin / 10))
    smax = nrm / math.sqrt(dimX * 10 ** (bmax / 10))
    y = Ax + sigma * np.asarray(noise, dtype=np.float64)
    lf = lambda s2: evMax ** 2 / s2
    Lf = defaults["lf"](lf(smin ** 2), lf(smax ** 2))
    lam = min(5 / Lf, lmax)
    gamma = defaults["gmul"] * gammaFrac * (1 / (Lf + 1 / lam))
    return dict(kind=kind, y=y, x=x, sigma=sigma, sigma_min=smin ** 2, sigma_max=smax ** 2,
                sigma_init=(smin ** 2 + smax ** 2) / 2, Lf=Lf, **{"lambda": lam}, gamma=gamma, p_true=p,
                th_init=th_init, dimX=dimX)


def _sapg(kind, y, op, c, noise=None, x0=None, share_gradients=False, reduce_fn=None, ctx=None, reduce_dev_fn=None,
          host_loop=False):
    """reduce_fn(user, buf, n): host all-reduce of the shared-gradient sums (selects the host-side parameter loop);
    reduce_dev_fn(user, dev_ptr, n, stream): in-stream all-reduce on the device buffer (`dist.make_device_allreduce_fn`),
    the loop stays device-resident; host_loop=True forces the host-side loop (SBTV_SAPG_HOST_LOOP)."""
    if reduce_fn is not None and reduce_dev_fn is not None:
        raise ValueError("give reduce_fn or reduce_dev_fn, not both")
    ctx = ctx or L.default_context()
    yi = L.Images(y)
    B, M, N = yi.B, yi.M, yi.N
    o = L.sbtv_sapg_opts()
    o.kind = _KIND[kind]
    o.psf_size = int(_get(op, "psf_size", 7))
    o.samples = int(_get(op, "samples"))
    o.warmup = int(_get(op, "warmup", 100))                       # default 100 (SAPG_algorithm_Guassian.m:20)
    o.burnIn = int(_get(op, "burnIn"))
    o.chambolleit = int(_get(op, "chambolleit", 25))              # run_Gaussian_demo.m:188
    o.share_gradients = 1 if share_gradients else 0
    o.lambda_ = float(c.get("lam", 1.0)) * float(_get(op, "lambda"))     # lamb = c.lam*op.lambda (:30)
    o.gamma = float(c.get("gam", 1.0)) * float(_get(op, "gamma"))
    o.th_init = float(_get(op, "th_init"))
    o.min_th = float(_get(op, "min_th"))
    o.max_th = float(_get(op, "max_th"))
    names = {"gaussian": ("w1", "w2"), "moffat": ("alpha", "beta"), "laplace": ("b",)}[kind]
    for q, nm in enumerate(names):
        o.p_init[q] = float(_get(op, nm + "_init"))
        o.p_min[q] = float(_get(op, "min_" + nm))
        o.p_max[q] = float(_get(op, "max_" + nm))
        o.p_true[q] = float(np.ravel(_get(op, nm))[0])
        o.fix_p[q] = int(bool(_get(op, "fix_" + nm, 0)))
        o.c_p[q] = float(c[nm])
    if len(names) == 1:
        o.fix_p[1] = 1
    o.phi = float(_get(op, "phi", 0.0))
    o.fix_sigma = int(bool(_get(op, "fix_sigma", 0)))
    o.sigma2_true = float(_get(op, "sigma")) ** 2                 # sigma = op.sigma^2 (:49)
    o.sigma2_init = float(_get(op, "sigma_init"))
    o.sigma2_min = float(_get(op, "sigma_min"))
    o.sigma2_max = float(_get(op, "sigma_max"))
    o.d_scale = float(_get(op, "d_scale"))
    o.d_exp = float(_get(op, "d_exp"))
    o.c_theta = float(c["theta"])
    o.c_sigma = float(c["sigma"])
    o.seed = int(_get(op, "seed", 1))
    o.chain_offset = int(_get(op, "chain_offset", 0))              # first chain of this process (dist.split_chains)
    o.iter_offset = int(_get(op, "iter_offset", 0))                # resume: iteration ii of this call steps with delta(ii + offset)
    S, W = o.samples, max(o.warmup, 1)
    nch = B
    thetas = np.zeros((nch, S)); sigmas = np.zeros((nch, S)); ps = np.zeros((nch, 2, S))
    logpi = np.zeros((nch, S)); logpi_wu = np.zeros((nch, W)); gx = np.zeros((nch, S)); grads = np.zeros((nch, 4, S))
    eb = np.zeros((nch, 4))
    xl = L.empty_like_images(yi)
    x0i = L.Images(x0) if x0 is not None else None
    if _get(op, "X0") is not None and x0 is None:
        x0i = L.Images(_get(op, "X0"))
    nz_ptr, nz_keep = None, None
    if noise is not None:
        if L._is_torch(noise):
            nz_keep = noise
            nz_ptr = C.c_void_p(noise.data_ptr())
        else:
            a = np.asarray(noise, dtype=np.float64)           # (steps, B, M, N) or (steps, M, N)
            if a.ndim == 3:
                a = a[:, None]
            nz_keep = L.column_major_images(a.reshape((-1,) + a.shape[2:])).reshape(a.shape[:2] + (a.shape[3], a.shape[2]))   # column-major images
            nz_ptr = L.vptr(nz_keep)
    cb = L.ALLREDUCE_FN(reduce_fn) if reduce_fn is not None else L.ALLREDUCE_FN()
    xflags = L.SAPG_HOST_LOOP if host_loop else 0
    if reduce_dev_fn is not None:
        cb_dev = L.ALLREDUCE_DEV_FN(reduce_dev_fn)            # kept alive until the call returns
        cb = C.cast(cb_dev, L.ALLREDUCE_FN)
        xflags |= L.REDUCE_DEVICE
    vp = L.vptr
    yptr = yi.ptr
    if share_gradients:
        # all chains sample ONE image: y is a single image, the batch size is the chain count
        nch = int(_get(op, "chains", 1))
        thetas = np.zeros((nch, S)); sigmas = np.zeros((nch, S)); ps = np.zeros((nch, 2, S))
        logpi = np.zeros((nch, S)); logpi_wu = np.zeros((nch, W)); gx = np.zeros((nch, S)); grads = np.zeros((nch, 4, S))
        eb = np.zeros((nch, 4))
        if yi.torch:
            import torch
            xl = L.Images(torch.empty((nch, N, M), dtype=torch.float64, device=yi.t.device).permute(0, 2, 1))
        else:
            xl = L.Images(np.zeros((nch, M, N)))
    if getattr(ctx, "is_group", False):
        # several GPUs behind this process (sbtv.Group): images / chains dealt to the devices in contiguous blocks; with
        # share_gradients the six gradient sums are exchanged in-process, in-stream (no reduce_fn needed or accepted)
        if yi.flags != L.SBTV_HOST_PTRS or (noise is not None and L._is_torch(noise)):
            raise ValueError("a sbtv.Group takes host (NumPy) images")
        if reduce_fn is not None or reduce_dev_fn is not None or host_loop:
            raise ValueError("a sbtv.Group exchanges the shared gradients itself: no reduce_fn / host_loop")
        ctx.check(ctx.lib.sbtv_SAPG_algorithm_sharded(ctx.h, yptr, M, N, nch, C.byref(o), x0i.ptr if x0i else None, nz_ptr,
                                                      vp(thetas), vp(ps), vp(sigmas), vp(logpi), vp(logpi_wu), vp(gx),
                                                      vp(grads), vp(eb), xl.ptr))
    else:
        ctx.check(ctx.lib.sbtv_SAPG_algorithm(ctx.h, yptr, M, N, nch, C.byref(o), x0i.ptr if x0i else None, nz_ptr,
                                              vp(thetas), vp(ps), vp(sigmas), vp(logpi), vp(logpi_wu), vp(gx), vp(grads),
                                              vp(eb), xl.ptr, cb, None, yi.flags | xflags), yi.flags)
    results = []
    for b in range(nch):
        r = dict(theta_EB=eb[b, 0], sigma_EB=eb[b, 3], last_samp=S, thetas=thetas[b], sigmas=sigmas[b],
                 logPiTraceX=logpi[b], logPiTrace_WU=logpi_wu[b, :o.warmup] if o.warmup > 0 else np.array([]),
                 gXTrace=gx[b], grad_theta=grads[b, 0], grad_sigma=grads[b, 3], last_theta=thetas[b, -1],
                 last_sigma=sigmas[b, -1], c_theta=o.c_theta, c_sigma=o.c_sigma, options=op)
        for q, nm in enumerate(names):
            r[nm + "_EB"] = eb[b, 1 + q]
            r[nm + "s"] = ps[b, q]
            r["last_" + nm] = ps[b, q, -1]
            r["grad_" + nm] = grads[b, 1 + q]
        results.append(r)
    xs = L.images_result(xl, False)
    for b in range(nch):
        results[b]["Xlast_sample"] = xs[b]
        # running means from burnIn and their relative change, as the reference logs them while iterating
        # (SAPG_algorithm_Guassian.m:217-244: mean_* over burnIn..ii for ii > burnIn; tol_*(ii) = relative change
        # of that mean, NaN while the window is empty)
        for key, tolkey in [("thetas", "tol_thetas"), ("sigmas", "tol_sigma")] + [(nm + "s", "tol_" + nm + "s") for nm in names]:
            m, t = _running_mean_and_tol(results[b][key], o.burnIn)
            results[b]["mean_" + key] = m
            results[b][tolkey] = t
        results[b]["err_psf"] = _err_psf(kind, o.psf_size, ps[b, :len(names)], [o.p_true[q] for q in range(len(names))],
                                         o.phi)
    return results


def _running_mean_and_tol(trace, burnIn):
    """mean_x(ii - burnIn) = mean(x(burnIn:ii)) for ii > burnIn and tol_x(ii) = |mean(x(burnIn:ii)) - mean(x(burnIn:ii-1))|
    / mean(x(burnIn:ii-1)) for ii >= 2 (1-based; NaN where a window is empty, like MATLAB's mean of an empty range)."""
    x = np.asarray(trace, dtype=np.float64)
    n, b0 = x.size, int(burnIn) - 1                      # 0-based index of burnIn
    run = np.full(n, np.nan)                             # run[i] = mean(x[b0 .. i]) for i >= b0
    if b0 < n:
        run[b0:] = np.cumsum(x[b0:]) / np.arange(1, n - b0 + 1)
    tol = np.zeros(n)
    with np.errstate(invalid="ignore", divide="ignore"):
        tol[1:] = np.abs(run[1:] - run[:-1]) / run[:-1]
    return run[b0 + 1:].copy(), tol


def _err_psf(kind, taille, ps, p_true, phi):
    """results.err_psf: l2 = norm(.)^2 with the MATRIX 2-norm (utils/l2.m, quirk Q9) between the PSF at the traced
    parameters and the true PSF, evaluated on the host from the parameter traces.  Family quirks as in the
    reference: Gaussian pairs w1s(ii) with w2s(ii-1) (SAPG_algorithm_Guassian.m:203, Q8); Moffat stores its first
    value under another name, so err_psf(1) = 0 (SAPG_algorithm_moffat.m:156); Laplace is plain (:136,191)."""
    lib = L.load_library()
    tr = np.zeros((2, ps.shape[1]))
    tr[:ps.shape[0]] = ps                                           # [first-parameter trace | second-parameter trace]
    pt = np.zeros(2)
    pt[:len(p_true)] = p_true
    out = np.zeros(ps.shape[1])
    rc = lib.sbtv_err_psf(_KIND[kind], int(taille), L.vptr(tr), int(ps.shape[1]),
                          L.vptr(pt), float(phi), L.vptr(out))
    if rc != 0:
        raise L.SbtvError(rc, lib.sbtv_last_error(None).decode())
    return out


def _unpack(kind, res, batched):
    names = {"gaussian": ("w1", "w2"), "moffat": ("alpha", "beta"), "laplace": ("b",)}[kind]
    if not batched:
        r = res[0]
        return (r["theta_EB"],) + tuple(r[n + "_EB"] for n in names) + (r["sigma_EB"], r)
    return ([r["theta_EB"] for r in res],) + tuple([r[n + "_EB"] for r in res] for n in names) + \
        ([r["sigma_EB"] for r in res], res)


def SAPG_algorithm_Guassian(y, op, c, **kw):
    """[theta_EB, w1_EB, w2_EB, sigma_EB, results] = SAPG_algorithm_Guassian(y, op, c)
    (SAPG/SAPG_algorithm_Guassian.m:7-308).  c: dict(theta, w1, w2, sigma, lam, gam) (run_Gaussian_demo.m:34-39)."""
    res = _sapg("gaussian", y, op, c, **kw)
    return _unpack("gaussian", res, len(res) > 1)


def SAPG_algorithm_moffat(y, op, c=None, **kw):
    """[theta_EB, alpha_EB, beta_EB, sigma2_EB, results] = SAPG_algorithm_moffat(y, op)
    (SAPG/SAPG_algorithm_moffat.m:7-297; step scales hard-coded at :135-138)."""
    c = c or dict(theta=0.1, alpha=10.0, beta=1e4, sigma=1e4, lam=1.0, gam=1.0)
    res = _sapg("moffat", y, op, c, **kw)
    return _unpack("moffat", res, len(res) > 1)


def SAPG_algorithm_laplace(y, op, c=None, **kw):
    """[theta_EB, b_EB, sigma2_EB, results] = SAPG_algorithm_laplace(y, op)
    (SAPG/SAPG_algorithm_laplace.m:7-268; step scales hard-coded at :139-141)."""
    c = c or dict(theta=0.01, b=100.0, sigma=1e4, lam=1.0, gam=1.0)
    res = _sapg("laplace", y, op, c, **kw)
    return _unpack("laplace", res, len(res) > 1)


def myula(op, im=None, noise=None, ctx=None):
    """xMAP = myula(op, im)  (SALSA/myula.m:1-22): the plain MYULA chain at fixed theta / PSF, last sample returned.

    op: dict / object with y, lambda, gamma, theta_op, samples as in the reference, plus what the closures
    op.proxG / op.gradF of SALSA/run_deblur_tv.m:126,131 are built from: `A` (the sbtv.BlurOperator at tau_op),
    `sigma2` (or `sigma`), `chambolleit` (25).  `im` is only used for its shape in the reference and is ignored.
    noise: optional (samples-2, [B,] M, N) array instead of the device Philox stream (op.seed, op.chain_offset)."""
    ctx = ctx or L.default_context()
    A = _get(op, "A")
    if not isinstance(A, BlurOperator):
        raise TypeError("op.A must be the sbtv.BlurOperator the closures op.gradF / op.proxG are built from")
    yi = L.Images(_get(op, "y"))
    B, M, N = yi.B, yi.M, yi.N
    s2 = _get(op, "sigma2")
    if s2 is None:
        s2 = np.asarray(_get(op, "sigma"), dtype=np.float64) ** 2
    keep = [L.dvec(_get(op, "theta_op"), B), L.dvec(s2, B)]
    taps = A._cm(B)
    xo = L.empty_like_images(yi)
    nz_ptr, nz_keep = None, None
    if noise is not None:
        if L._is_torch(noise):
            nz_keep, nz_ptr = noise, C.c_void_p(noise.data_ptr())
        else:
            a = np.asarray(noise, dtype=np.float64)
            if a.ndim == 3:
                a = a[:, None]
            nz_keep = L.column_major_images(a.reshape((-1,) + a.shape[2:])).reshape(a.shape[:2] + (a.shape[3], a.shape[2]))
            nz_ptr = L.vptr(nz_keep)
    ctx.check(ctx.lib.sbtv_myula(ctx.h, yi.ptr, M, N, B, L.vptr(taps), A.taille,
                                 float(_get(op, "lambda")), float(_get(op, "gamma")), keep[0][1], keep[1][1],
                                 int(_get(op, "samples")), int(_get(op, "chambolleit", 25)), int(_get(op, "seed", 1)),
                                 int(_get(op, "chain_offset", 0)), nz_ptr, xo.ptr, yi.flags), yi.flags)
    y = _get(op, "y")
    return L.images_result(xo, (y.dim() == 2) if yi.torch else yi.squeeze)
