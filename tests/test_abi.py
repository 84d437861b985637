"""CPU tests of the boundary: libsbtv.so loads without a GPU, exports every
symbol include/sbtv.h declares, and refuses to create a context when there is
no device (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sbtv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sbtv_[A-Za-z0-9_]+)\s*\(", text)) - {"sbtv_allreduce_fn"})


def test_header_symbols_exported():
    import sbtv
    lib = sbtv.load_library()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sbtv.h but not exported by libsbtv.so"
    # the ctypes binding covers exactly the declared functions
    from sbtv import _lib
    assert sorted(_lib.SIGNATURES) == names
    assert lib.sbtv_version() == 100


def test_struct_layouts_match_header():
    from sbtv import _lib
    # 6 ints + 3 doubles
    assert C.sizeof(_lib.sbtv_salsa_opts) == 6 * 4 + 3 * 8
    so = _lib.sbtv_salsa_opts()
    _lib.load_library().sbtv_salsa_opts_default(C.byref(so))
    assert (so.stopcriterion, so.maxiter, so.TViters, so.initialization) == (1, 10000, 5, 0)
    assert (so.tolA, so.chambolle_tol, so.chambolle_tau) == (0.001, 1e-3, 0.249)
    # compile the header with gcc and compare sizeof / a late field offset with the ctypes mirror
    import subprocess
    import tempfile
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "sbtv.h"\nint main(void){printf("%zu %zu %zu %zu\\n",'
           'sizeof(sbtv_sapg_opts), offsetof(sbtv_sapg_opts, seed), sizeof(sbtv_salsa_opts),'
           'offsetof(sbtv_sapg_opts, lambda));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o",
                        os.path.join(d, "t")], check=True)
        out = subprocess.run([os.path.join(d, "t")], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == C.sizeof(_lib.sbtv_sapg_opts)
    assert int(out[1]) == _lib.sbtv_sapg_opts.seed.offset
    assert int(out[2]) == C.sizeof(_lib.sbtv_salsa_opts)
    assert int(out[3]) == _lib.sbtv_sapg_opts.lambda_.offset


def test_psf_taps_match_oracle():
    """sbtv_psf_taps is host arithmetic: compare with the oracle's restatement."""
    import sbtv
    import sbtv_oracle as o
    for kind, p in (("gaussian", (0.4, 0.3)), ("gaussian", (0.7, 0.2, 0.3)), ("moffat", (0.4, 3.5)), ("laplace", (0.3,))):
        taps, d = sbtv.psf_family(kind, 7, p)
        ref, dref = o.PSF_TAPS[kind]
        assert np.allclose(taps, ref(7, p), rtol=1e-14, atol=1e-17)
        for a, fn in zip(d, dref):
            assert np.allclose(a, fn(7, p), rtol=1e-12, atol=1e-16)
    assert np.allclose(sbtv.Gaussian_psf(7, 0.4, 0.3), o.Gaussian_psf(7, 0.4, 0.3), rtol=1e-14)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import sbtv
    with pytest.raises(sbtv.SbtvError) as e:
        sbtv.Context(0)
    assert e.value.code == -12          # SBTV_ERR_NODEVICE


def test_a_wrapper_vectorised_adaptor_errors():
    import sbtv
    with pytest.raises(sbtv.SbtvError):
        sbtv.A_wrapper(lambda v: v, lambda v: v, np.zeros(4), 2, 2, 2, 2, 3)     # A_wrapper.m:15
    g = sbtv.A_wrapper(lambda v: 2 * v, lambda v: v, np.arange(6.0), 2, 3, 2, 3, 1)
    assert g.shape == (6, 1) and np.allclose(g[:, 0], 2 * np.arange(6.0))


def _build_c_host(tmp_path, name="c_host"):
    import subprocess
    exe = str(tmp_path / name)
    libdir = os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", name + ".c"), "-L" + libdir, "-lsbtv", "-lm",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_plain_c_host_compiles_and_links(tmp_path):
    """The boundary is usable from C without Python or torch: examples/c_host.c builds against include/sbtv.h
    and links libsbtv.so (running it needs the GPU: tests/test_gpu_c_host.py)."""
    assert os.path.exists(_build_c_host(tmp_path))
    assert os.path.exists(_build_c_host(tmp_path, "c_host_multi"))      # the single-process multi-GPU host (sbtv_group)


@pytest.mark.parametrize("kind", ["gaussian", "moffat", "laplace"])
def test_err_psf_host_entry_matches_oracle(kind):
    """sbtv_err_psf is host arithmetic (PSF builders + a Jacobi spectral norm): it runs without a GPU and must
    reproduce the oracle's results.err_psf trace incl. the per-family quirks (Q8, Q9, the Moffat first entry)."""
    import sbtv
    import sbtv_oracle as o
    from sbtv.sapg import _err_psf
    rng = np.random.default_rng(3)
    n = 40
    lo, hi = {"gaussian": (0.1, 1.0), "moffat": (0.05, 8.0), "laplace": (0.05, 1.0)}[kind]
    npar = 1 if kind == "laplace" else 2
    ps = rng.uniform(lo, hi, (npar, n))
    ps[:, 5] = ps[:, 4]                                   # repeated parameters
    p_true = {"gaussian": (0.4, 0.3), "moffat": (0.4, 3.5), "laplace": (0.3,)}[kind]
    got = _err_psf(kind, 7, ps, p_true, 0.0)
    ref = o.err_psf_trace(kind, ps, p_true, 7)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-20)
    if kind == "moffat":
        assert got[0] == 0.0
