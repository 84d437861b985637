"""GPU: BASELINE configs[3] and configs[4] at their FULL unit counts on one GPU (the oracle needs hours there), through
properties that do not depend on the size:
  configs[3]  64 independent 1024 x 1024 images, Laplace SAPG, ONE call (two lanes of 32): image k of the batch is bit for bit
              the single-image call that draws Philox stream k - checked at the lane seam, the ends and the middle;
  configs[4]  32 MYULA chains on one 2048 x 2048 image with chain-averaged gradients: every chain carries the SAME parameter
              traces (the update is common), the chains themselves differ, and the call split 16 + 16 over the two lanes
              (in-stream exchange of the six sums) reproduces the one-stream call to the rounding of the sum order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tiled(size):
    import os
    man = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "man_512.npy")).astype(np.float64)
    r = size // 512
    return np.tile(man, (r, r))


def _op(kind, st, samples, warmup):
    import sbtv_oracle as o
    d = o.DEMO[kind]
    names = {"gaussian": ("w1", "w2"), "laplace": ("b",)}[kind]
    op = dict(samples=samples, warmup=warmup, burnIn=2, chambolleit=25, psf_size=7, phi=0.0, gamma=st["gamma"],
              th_init=0.01, min_th=1e-3, max_th=1.0, sigma=st["sigma"], sigma_init=st["sigma_init"],
              sigma_min=st["sigma_min"], sigma_max=st["sigma_max"], d_scale=1.0, d_exp=0.8, fix_sigma=0, seed=9)
    op["lambda"] = st["lambda"]
    for q, nm in enumerate(names):
        op[nm], op[nm + "_init"] = st["p_true"][q], d["init"][q] if kind == "laplace" else st["p_true"][q]
        op["min_" + nm], op["max_" + nm], op["fix_" + nm] = d["pmin"][q], d["pmax"][q], 0 if kind == "laplace" else 1
    c = dict(theta=d["c_theta"], sigma=d["c_sigma"], lam=1.0, gam=1.0, **{nm: d["c_p"][q] for q, nm in enumerate(names)})
    return op, c


def test_config3_full_batch_of_64_images_equals_their_single_image_calls():
    import sbtv
    import torch
    ctx = sbtv.Context(0)
    try:
        x = _tiled(1024)
        rng = np.random.default_rng(3)
        st = sbtv.demo_setup("laplace", x, rng.standard_normal(x.shape), evMax=0.99, ctx=ctx)
        op, c = _op("laplace", st, 4, 2)
        # 64 different observations from one: scaled and shifted copies (the per-image state is what must not mix)
        y0 = sbtv.to_device(st["y"])
        yb = torch.empty((64, 1024, 1024), dtype=torch.float64, device=y0.device).permute(0, 2, 1)    # column-major images
        for k in range(64):
            yb[k].copy_(y0 * (1.0 - 0.004 * k) + 0.25 * k)
        assert yb.shape == (64, 1024, 1024)
        out = sbtv.SAPG_algorithm_laplace(yb, op, c, ctx=ctx)[-1]
        assert len(out) == 64 and len({float(r["thetas"][-1]) for r in out}) == 64
        ctx.set_lanes(1)
        for k in (0, 31, 32, 63, 17):
            one = sbtv.SAPG_algorithm_laplace(yb[k], dict(op, chain_offset=k), c, ctx=ctx)[-1]
            for key in ("thetas", "bs", "sigmas", "logPiTraceX", "gXTrace"):
                np.testing.assert_array_equal(out[k][key], one[key], err_msg=f"image {k}: {key}")
            assert torch.equal(out[k]["Xlast_sample"], one["Xlast_sample"])
    finally:
        ctx.close()


def test_config4_full_32_chains_on_one_2048_image():
    import sbtv
    ctx = sbtv.Context(0)
    try:
        x = _tiled(2048)
        st = sbtv.demo_setup("gaussian", x, np.random.default_rng(4).standard_normal(x.shape), evMax=0.99, ctx=ctx)
        op, c = _op("gaussian", st, 4, 2)
        op.update(chains=32, fix_w1=0, fix_w2=0, w1_init=0.5, w2_init=0.35)
        c.update(w1=0.3, w2=0.3, sigma=100.0)
        yd = sbtv.to_device(st["y"])
        one = sbtv.SAPG_algorithm_Guassian(yd, op, c, share_gradients=True, ctx=ctx)[-1]
        assert len(one) == 32
        for k in range(1, 32):                                   # one common update: identical parameter traces
            for key in ("thetas", "w1s", "w2s", "sigmas"):
                np.testing.assert_array_equal(one[k][key], one[0][key])
        lp = np.array([r["logPiTraceX"][-1] for r in one])
        assert len(set(lp.tolist())) == 32 and np.all(np.isfinite(lp))          # 32 different chains
        assert one[0]["w1s"][-1] != one[0]["w1s"][0] and 0.1 < one[0]["w1s"][-1] < 1.0
        ctx.set_lanes(2)                                          # 16 + 16 chains on the two lanes
        two = sbtv.SAPG_algorithm_Guassian(yd, op, c, share_gradients=True, ctx=ctx)[-1]
        for k in (0, 15, 16, 31):
            for key in ("thetas", "w1s", "w2s", "sigmas", "logPiTraceX"):
                np.testing.assert_allclose(two[k][key], one[k][key], rtol=1e-10, err_msg=f"chain {k}: {key}")
    finally:
        ctx.close()
