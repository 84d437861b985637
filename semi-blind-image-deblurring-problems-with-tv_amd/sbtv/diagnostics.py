"""Host-side diagnostics of the demos: SSIM and result files.

These run on the host (NumPy): they are reporting code around the hot path, not part of it.

`ssim`: the demos call the Image Processing Toolbox `ssim(x, xMAP)` (run_Gaussian_demo.m:245).  That
function is closed source and absent here, so this is an independent implementation of the published
definition (Wang et al. 2004) with MATLAB's documented defaults — Gaussian window of standard
deviation 1.5 (11 x 11), exponents 1, replicate padding, and `DynamicRange = 1` for floating-point
inputs (so for the demos' 0..255 doubles C1, C2 are tiny, exactly as in the reference).
**Parity unpinned**: no reference output exists to compare with.
"""
from __future__ import annotations

import numpy as np


def _gauss_filter(a, sigma=1.5):
    r = int(np.ceil(3 * sigma))                               # MATLAB: filter size 2*ceil(3*sigma)+1 = 11
    k = np.exp(-(np.arange(-r, r + 1) ** 2) / (2 * sigma * sigma))
    k /= k.sum()
    p = np.pad(a, r, mode="edge")                             # imfilter(..., 'replicate')
    out = np.zeros_like(a, dtype=np.float64)
    tmp = np.zeros((p.shape[0], a.shape[1]))
    for i, w in enumerate(k):
        tmp += w * p[:, i:i + a.shape[1]]
    for i, w in enumerate(k):
        out += w * tmp[i:i + a.shape[0], :]
    return out


def ssim(A, ref, dynamic_range=1.0, return_map=False):
    """[ssimval, ssimmap] = ssim(A, ref)  (Image Processing Toolbox defaults; see module docstring)."""
    A = np.asarray(A, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if A.shape != ref.shape:
        raise ValueError("A and ref must have the same size")
    C1, C2 = (0.01 * dynamic_range) ** 2, (0.03 * dynamic_range) ** 2
    mux, muy = _gauss_filter(A), _gauss_filter(ref)
    sxx = _gauss_filter(A * A) - mux * mux
    syy = _gauss_filter(ref * ref) - muy * muy
    sxy = _gauss_filter(A * ref) - mux * muy
    m = ((2 * mux * muy + C1) * (2 * sxy + C2)) / ((mux * mux + muy * muy + C1) * (sxx + syy + C2))
    val = float(np.mean(m))
    return (val, m) if return_map else val


def save_results(path, results, **extra):
    """The legacy scripts `save` their results struct to a .mat (SALSA/salsa_m.m:346, run_deblur_tv.m:169);
    here a results dict (as returned by SAPG_algorithm_* / SALSA_v2 wrappers) goes to a compressed .npz."""
    flat = {}
    for k, v in dict(results, **extra).items():
        if isinstance(v, dict):
            continue                                         # nested option structs are not arrays
        try:
            flat[k] = np.asarray(v)
        except Exception:
            pass
    np.savez_compressed(path, **flat)
    return path


def load_results(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}
