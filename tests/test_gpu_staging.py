"""GPU: staging of large pageable host arrays through the copy lanes of a context (csrc/ctx.hip `stage_copy`: four
threads, two pinned 4 MB chunks each, own streams).  A MATLAB / NumPy host always takes this path (SBTV_HOST_PTRS)."""
import ctypes as C
import time

import numpy as np
import pytest

from conftest import synth_image

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nbytes", [8 << 20, (8 << 20) + 8, (37 << 20) + 1048, 4 << 20, (4 << 20) - 8, 1000])
def test_host_device_round_trip_of_every_chunking(ctx, nbytes):
    """Whole chunks, a ragged tail, fewer chunks than threads, and sizes below the threshold (plain copy)."""
    import sbtv
    from sbtv import _lib as L
    rng = np.random.default_rng(nbytes)
    src = rng.integers(0, 256, nbytes, dtype=np.uint8)
    dst = np.zeros(nbytes + 64, dtype=np.uint8)
    dst[:] = 0xA5
    p = C.c_void_p()
    ctx.check(ctx.lib.sbtv_malloc(ctx.h, nbytes, C.byref(p)))
    try:
        s0 = ctx.stage_stats()
        ctx.check(ctx.lib.sbtv_memcpy_h2d(ctx.h, p, L.vptr(src), nbytes))
        ctx.check(ctx.lib.sbtv_memcpy_d2h(ctx.h, L.vptr(dst[32:]), p, nbytes))
        s1 = ctx.stage_stats()
    finally:
        ctx.check(ctx.lib.sbtv_free(ctx.h, p))
    np.testing.assert_array_equal(dst[32:32 + nbytes], src)
    assert np.all(dst[:32] == 0xA5) and np.all(dst[32 + nbytes:] == 0xA5)          # nothing written past the ends
    moved = nbytes if nbytes >= (4 << 20) else 0
    assert s1["bytes_in"] - s0["bytes_in"] == moved and s1["bytes_out"] - s0["bytes_out"] == moved


def test_salsa_with_host_images_equals_device_images_and_reports_its_staging():
    """3 images of 1024 x 768 (18.9 MB per array: a ragged tail chunk, split over two lanes): host arrays in and out
    give the bits of the device-resident call."""
    import sbtv
    import sbtv_oracle as o
    c = sbtv.Context(0)
    try:
        M, N, B = 1024, 768, 3
        xs = np.stack([synth_image(M, N, 50 + b) for b in range(B)])
        ys = np.stack([o.demo_setup("gaussian", xs[b], np.random.default_rng(b).standard_normal((M, N)), evMax=1.0)["y"]
                       for b in range(B)])
        A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3), ctx=c)
        args = (0.05, "MU", 0.003, "AT", A.T, "LS", A.LS(0.003), "ToleranceA", -1.0, "MAXITERA", 6, "TVINITIALIZATION", 1,
                "TViters", 10)
        dev = sbtv.SALSA_v2(sbtv.to_device(ys), A, *args, "True_x", sbtv.to_device(xs), ctx=c)
        s0 = c.stage_stats()
        host = sbtv.SALSA_v2(ys, A, *args, "True_x", xs, ctx=c)
        s1 = c.stage_stats()
        np.testing.assert_array_equal(host[0], sbtv.to_host(dev[0]))
        for b in range(B):
            np.testing.assert_array_equal(host[3][b], dev[3][b])
        assert s1["bytes_in"] - s0["bytes_in"] == 2 * ys.nbytes and s1["bytes_out"] - s0["bytes_out"] == ys.nbytes
        # pageable staging at more than the single-thread memcpy rate the plain path is bound by (~9 GB/s measured)
        gbs = (s1["bytes_in"] - s0["bytes_in"]) / (s1["s_in"] - s0["s_in"]) / 1e9
        print(f"host -> device {gbs:.1f} GB/s, device -> host "
              f"{(s1['bytes_out'] - s0['bytes_out']) / (s1['s_out'] - s0['s_out']) / 1e9:.1f} GB/s")
    finally:
        c.close()
