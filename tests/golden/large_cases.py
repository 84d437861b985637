"""Seeded INPUTS of the full-size parity cases (BASELINE.json configs[1..4] at their real sizes).

Shared by `make_golden_large.py` (runs the oracle on them in the build container and commits the expected outputs as
`large_configs.npz`) and by `tests/test_gpu_large_fixtures.py` (rebuilds the same inputs on the GPU box and holds the
HIP path against the committed outputs — the oracle needs minutes per case at these sizes and is not run there).
The inputs come from `man_512.npy`, `numpy.random.default_rng(seed)` and the data-synthesis step (one blur + the BSNR
rule: `bench.make_problem` / the oracle's `demo_setup`), so they are the same arrays on both machines (same image, same
NumPy / SciPy).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FIXTURE = os.path.join(HERE, "large_configs.npz")


def man512():
    return np.load(os.path.join(HERE, "man_512.npy")).astype(np.float64)


def tiled(size):
    r = size // 512
    return np.tile(man512(), (r, r))[:size, :size]


def crops(M, N):
    """Where the arrays are compared element for element: an interior seam of the fused Chambolle tiles (core tile
    116 x 21: rows 110..122 / columns 16..26 straddle the first seam in both directions), the seam of the 4 x 4 image
    tiling, the last rows / columns (quirk Q3: the non-adjoint divergence leaves them unsmoothed) and the origin."""
    return {
        "seam": (slice(110, 123), slice(16, 27)),
        "mid": (slice(M // 2 - 4, M // 2 + 4), slice(N // 2 - 4, N // 2 + 4)),
        "last": (slice(M - 8, M), slice(N - 8, N)),
        "origin": (slice(0, 8), slice(0, 8)),
    }


# ---- headline: the problem bench.py times (2048^2 Gaussian SALSA, run_Gaussian_demo.m:229-245) -----------------
def salsa2048():
    import bench
    x, y, sigma, noise = bench.make_problem(seed=1)
    theta = bench.THETA
    return dict(x=x, y=y, sigma=sigma, w=bench.W_TRUE, theta=theta, mu=theta / 10, tau=theta * sigma ** 2,
                tol=1e-5, maxiter=500, TViters=10)


# ---- configs[1]: 512^2 man.png, the same solve (bench.py extra_512) ---------------------------------------------
def salsa512():
    import bench
    x, y, sigma, noise = bench.make_problem(seed=1, size=512)
    theta = bench.THETA
    return dict(x=x, y=y, sigma=sigma, w=bench.W_TRUE, theta=theta, mu=theta / 10, tau=theta * sigma ** 2,
                tol=1e-5, maxiter=500, TViters=10)


# ---- configs[2]: FISTA + cold TV prox(25), 2048^2, Moffat PSF (SALSA/my_fista.m:21-56) --------------------------
FISTA_ITERS = 4          # my_fista's `maxiters`: objective(1) + 3 updates


def _oracle():
    """Data synthesis (run_*_demo.m:145-184: one blur + the BSNR rule) is the oracle's `demo_setup` on both machines,
    so both see the same bits of y; it costs three FFTs."""
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)
    import sbtv_oracle
    return sbtv_oracle


def fista2048():
    o = _oracle()
    x = tiled(2048)
    st = o.demo_setup("moffat", x, np.random.default_rng(1).standard_normal(x.shape), evMax=1.0)
    return dict(x=x, y=st["y"], sigma=st["sigma"], p=(0.4, 3.5), tau=0.03 * st["sigma"] ** 2, L=1.0, st=st)


# ---- configs[3]: Laplace SAPG, a GPU's share (8) of the batch of 64 independent 1024^2 images --------------------
# step scales: see SAPG_S below (the in-function c_b = 100, c_sigma = 1e4 of SAPG_algorithm_laplace.m:139-141 put b and
# sigma^2 on their bounds in the first update of a chain that starts at X = y)
SAPG_L = dict(samples=3, warmup=2, burnIn=2, batch=8, c=dict(theta=0.01, p=(0.1,), sigma=100.0))


# the same batch with the reference's OWN step scales (SAPG_algorithm_laplace.m:139-141): b and sigma^2 are thrown onto their
# projection bounds by the first update, the following updates run with the projected values (round-3 review, weak #2)
SAPG_L_REF = dict(samples=4, warmup=2, burnIn=2, batch=8, c=dict(theta=0.01, p=(100.0,), sigma=1e4))


def sapg_laplace_1024x8(S=None):
    S = S or SAPG_L
    o = _oracle()
    x0 = tiled(1024)
    rng = np.random.default_rng(4)
    mods = ((1.0, 0.0), (0.7, 30.0), (0.5, 90.0), (0.9, 10.0), (0.6, 60.0), (0.8, 5.0), (0.4, 120.0), (1.0, -20.0))
    xs = [np.clip((x0 if b % 2 == 0 else x0[::-1, ::-1]) * s + off, 0, 255) for b, (s, off) in enumerate(mods)]
    sts = [o.demo_setup("laplace", x, rng.standard_normal(x.shape), evMax=0.99) for x in xs]
    # one option struct per call (include/sbtv.h): the step sizes and the sigma^2 bounds of image 0 serve the batch
    for st in sts[1:]:
        for k in ("lam", "gamma", "sigma", "sigma_init", "sigma_min", "sigma_max"):
            st[k] = sts[0][k]
    steps = S["warmup"] - 1 + S["samples"] - 1
    noise = np.random.default_rng(40).standard_normal((steps, S["batch"], 1024, 1024))   # [step][image]
    return dict(xs=xs, sts=sts, noise=noise)


# ---- configs[4]: MYULA chains on ONE 2048^2 image with averaged PSF-parameter gradients (diff_fftgaus_w1/w2) ------
# step scales: the demo's c.w1 = c.w2 = 10, c.sigma = 1000 (run_Gaussian_demo.m:34-39) throw all three parameters onto
# their projection bounds in the first update of a chain that starts at X = y on 4 M pixels (G_w1 = 1.2e6); smaller
# scales keep them inside, so that the second update runs with a PSF that the first one really moved
SAPG_S = dict(samples=3, warmup=2, burnIn=2, chains=2, p_init=(0.5, 0.35), c=dict(theta=0.01, p=(0.3, 0.3), sigma=100.0))


# ... and with the demo's own c.w1 = c.w2 = 10, c.sigma = 1000 (run_Gaussian_demo.m:34-39; both widths free, init 0.5 / 0.3 of
# :68-69): all three parameters hit their bounds in the first update
SAPG_S_REF = dict(samples=4, warmup=2, burnIn=2, chains=2, p_init=(0.5, 0.3), c=dict(theta=0.01, p=(10.0, 10.0), sigma=1000.0))


def sapg_shared_2048x2(S=None):
    S = S or SAPG_S
    o = _oracle()
    x = tiled(2048)
    st = o.demo_setup("gaussian", x, np.random.default_rng(6).standard_normal(x.shape), evMax=0.99)
    steps = S["warmup"] - 1 + S["samples"] - 1
    noise = np.random.default_rng(60).standard_normal((steps, S["chains"], 2048, 2048))  # [step][chain]
    return dict(x=x, st=st, noise=noise)
