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
    """The legacy scripts `save` their results struct to a .mat (SALSA/salsa_m.m:346, run_deblur_tv.m:169).
    A path ending in `.mat` writes a MATLAB v5 file whose variables are the fields of the results dict (as returned
    by the SAPG_algorithm_* / SALSA_v2 wrappers; arrays keep their MATLAB orientation, traces are row vectors), so
    `load results.mat` gives the reference's field names back; any other path writes a compressed `.npz`."""
    flat = {}
    for k, v in dict(results, **extra).items():
        if isinstance(v, dict):
            continue                                         # nested option structs are not arrays
        try:
            flat[k] = np.asarray(v)
        except Exception:
            pass
    if str(path).lower().endswith(".mat"):
        from scipy.io import savemat
        savemat(path, {k: (v if v.ndim != 1 else v[None, :]) for k, v in flat.items() if v.dtype != object},
                do_compression=True, oned_as="row")
        return path
    np.savez_compressed(path, **flat)
    return path


def load_results(path):
    if str(path).lower().endswith(".mat"):
        from scipy.io import loadmat
        z = loadmat(path)
        out = {}
        for k, v in z.items():
            if k.startswith("__"):
                continue
            v = np.asarray(v)
            out[k] = v.item() if v.size == 1 else (v[0] if v.ndim == 2 and v.shape[0] == 1 else v)
        return out
    with np.load(path, allow_pickle=False) as z:
        return {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}


# ---- figures of the demos (run_Gaussian_demo.m:247-301) without a plotting library -------------------------
def _polyline(vals, x0, y0, w, h, lo, hi, n_max=2000):
    v = np.asarray(vals, dtype=np.float64)
    if v.size > n_max:                                       # decimate long traces for the picture only
        v = v[np.linspace(0, v.size - 1, n_max).astype(int)]
    xs = x0 + w * (np.arange(v.size) / max(v.size - 1, 1))
    ys = y0 + h - h * (v - lo) / (hi - lo if hi > lo else 1.0)
    return " ".join(f"{a:.1f},{b:.1f}" for a, b in zip(xs, ys))


def plot_traces(path, results, true_values=None, names=None):
    """One panel per trace of a SAPG `results` dict (thetas, <param>s, sigmas) with the EB estimate in the title and
    the true value as a red line, like the demo's figSigma / figTheta / figw1 / figw2; written as a plain SVG."""
    true_values = true_values or {}
    if names is None:
        names = [k for k in ("thetas", "w1s", "w2s", "alphas", "betas", "bs", "sigmas") if k in results]
    W, H, pad = 420, 240, 48
    parts = [f'<svg xmlns="http://www.w3.org/2000/svg" width="{W * len(names)}" height="{H}" font-family="sans-serif" '
             f'font-size="11">', f'<rect width="{W * len(names)}" height="{H}" fill="white"/>']
    for i, nm in enumerate(names):
        v = np.asarray(results[nm], dtype=np.float64)
        tv = true_values.get(nm)
        lo, hi = float(v.min()), float(v.max())
        if tv is not None:
            lo, hi = min(lo, tv), max(hi, tv)
        if hi <= lo:
            hi = lo + 1.0
        x0, y0, w, h = i * W + pad, 24, W - pad - 12, H - 24 - 36
        eb = results.get(nm[:-1] + "_EB")
        title = nm[:-1] + "_n" + (f"   (EB estimate {eb:.5g})" if eb is not None else "")
        parts.append(f'<rect x="{x0}" y="{y0}" width="{w}" height="{h}" fill="none" stroke="#888"/>')
        parts.append(f'<text x="{x0}" y="16">{title}</text>')
        parts.append(f'<text x="{x0 - 4}" y="{y0 + 10}" text-anchor="end">{hi:.4g}</text>')
        parts.append(f'<text x="{x0 - 4}" y="{y0 + h}" text-anchor="end">{lo:.4g}</text>')
        parts.append(f'<text x="{x0 + w / 2}" y="{y0 + h + 26}" text-anchor="middle">Iteration (n), 1..{v.size}</text>')
        if tv is not None:
            yt = y0 + h - h * (tv - lo) / (hi - lo)
            parts.append(f'<line x1="{x0}" y1="{yt:.1f}" x2="{x0 + w}" y2="{yt:.1f}" stroke="red"/>')
        parts.append(f'<polyline fill="none" stroke="blue" stroke-width="1.5" points="{_polyline(v, x0, y0, w, h, lo, hi)}"/>')
    parts.append("</svg>")
    with open(path, "w") as f:
        f.write("\n".join(parts))
    return path


def save_image(path, x, vmin=None, vmax=None):
    """`imagesc(x), colormap gray` as a binary PGM (readable by any image viewer); scaled to [vmin, vmax]."""
    a = np.asarray(x, dtype=np.float64)
    lo = float(a.min()) if vmin is None else vmin
    hi = float(a.max()) if vmax is None else vmax
    g = np.clip(np.rint(255 * (a - lo) / (hi - lo if hi > lo else 1.0)), 0, 255).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (g.shape[1], g.shape[0]))
        f.write(g.tobytes())
    return path
