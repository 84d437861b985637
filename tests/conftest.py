import os
import sys

# BLAS / OpenMP pools busy-wait after every call; with 64+ threads inside a CPU-quota container that throttles the whole
# process, also the thread that feeds the GPU (profiles/r03_slow_mode_512.md).  The oracle's FFTs use scipy.fft's own
# `workers` pool and are not affected.
for _v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "OMP_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the lab build (make -C csrc lab; __graft_entry__.build() makes it too): the default library plus the kernel variants
# that were measured and lost; child processes select it with SBTV_LIBRARY
LAB_LIB = os.path.join(PKG, "lib", "libsbtv_lab.so")


try:                                      # NumPy may have been imported (by a plugin) before the variables were set
    import threadpoolctl
    threadpoolctl.threadpool_limits(limits=1, user_api="blas")
except Exception:                         # pragma: no cover
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def man512():
    return np.load(os.path.join(GOLDEN, "man_512.npy")).astype(np.float64)


@pytest.fixture(scope="session")
def cman256():
    return np.load(os.path.join(GOLDEN, "cman_256.npy")).astype(np.float64)


@pytest.fixture(scope="session")
def ctx():
    """One GPU context for the whole session (fails loudly without the HIP library / GPU)."""
    import sbtv
    return sbtv.default_context(0)


def synth_image(M, N, seed=0):
    """Piecewise-constant blocks + smooth texture in [0,255] (SURVEY.md §8d generator)."""
    rng = np.random.default_rng(seed)
    i = np.arange(M)[:, None]
    j = np.arange(N)[None, :]
    blocks = np.zeros((M, N))
    for _ in range(12):
        a, b = rng.integers(0, M), rng.integers(0, N)
        h, w = rng.integers(max(2, M // 8), max(3, M // 2)), rng.integers(max(2, N // 8), max(3, N // 2))
        blocks[a:a + h, b:b + w] += rng.uniform(-0.5, 0.5)
    x = 0.5 + 0.25 * np.sin(2 * np.pi * i / 97) * np.cos(2 * np.pi * j / 61) + 0.25 * blocks
    return 255.0 * np.clip(x, 0, 1)
