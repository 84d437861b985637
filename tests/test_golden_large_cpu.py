"""CPU side of the full-size fixtures (`tests/golden/large_configs.npz`): the file is what `make_golden_large.py` writes
(every case present, shapes as the GPU tests expect), and the case the oracle finishes in seconds (512^2 SALSA) is
recomputed here, so a change of the oracle or of the inputs shows up on every CPU run.  The 2048^2 / 8 x 1024^2 cases take
the oracle ~15 minutes: regenerate them with the script when the oracle changes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import large_cases as lc  # noqa: E402


def _fx():
    with np.load(lc.FIXTURE) as f:       # plain arrays: no pickle involved
        return {k: f[k] for k in f.files}


def test_fixture_inventory():
    fx = _fx()
    for tag, size in (("salsa2048", 2048), ("salsa512", 512)):
        n = int(fx[f"{tag}.n_outer"])
        assert fx[f"{tag}.objective"].shape == (n + 1,) and fx[f"{tag}.mses"].shape == (n + 1,)
        assert fx[f"{tag}.distance"].shape == (n,)
        for arr in ("x", "u", "bu"):
            for name, (si, sj) in lc.crops(size, size).items():
                assert fx[f"{tag}.{arr}.{name}"].shape == (si.stop - si.start, sj.stop - sj.start)
        assert abs(fx[f"{tag}.criterion"][-1]) < 1e-5 <= abs(fx[f"{tag}.criterion"][-2])     # stopped by tolA, not before
    assert int(fx["salsa2048.n_outer"]) == 33 and abs(float(fx["salsa2048.psnr"]) - 29.5076) < 1e-3
    assert fx["fista2048.objective"].shape == (lc.FISTA_ITERS,) and np.all(np.diff(fx["fista2048.objective"]) < 0)
    S = lc.SAPG_L
    for b in range(S["batch"]):
        assert fx[f"sapg_l.{b}.thetas"].shape == (S["samples"],) and fx[f"sapg_l.{b}.grads"].shape == (3, S["samples"])
        bs = fx[f"sapg_l.{b}.bs"]
        assert np.all((bs > 1e-3) & (bs < 1.0)) and bs[0] != bs[1] != bs[2]
    assert not np.array_equal(fx["sapg_l.0.thetas"], fx["sapg_l.1.thetas"])
    S = lc.SAPG_S
    assert fx["sapg_s.ps"].shape == (2, S["samples"]) and fx["sapg_s.logPi"].shape == (S["chains"], S["samples"])
    assert np.all((fx["sapg_s.ps"] > 0.1) & (fx["sapg_s.ps"] < 1.0))
    # the same two cases with the reference's own step scales: the first update throws the PSF parameters and sigma^2 onto
    # their projection bounds (SAPG_algorithm_Guassian.m:166-194), the following ones run with the projected values
    S = lc.SAPG_L_REF
    for b in range(S["batch"]):
        bs = fx[f"sapg_l_ref.{b}.bs"]
        assert bs.shape == (S["samples"],) and bs[0] == 0.1 and np.all(bs[1:] == 1e-3)
        assert np.all(np.diff(fx[f"sapg_l_ref.{b}.thetas"][1:]) < 0)                # theta keeps moving
    S = lc.SAPG_S_REF
    assert fx["sapg_s_ref.ps"].shape == (2, S["samples"]) and np.all(fx["sapg_s_ref.ps"][:, 1:] == 0.1)
    assert fx["sapg_s_ref.sigmas"][1] == fx["sapg_s_ref.sigmas"][2] > fx["sapg_s_ref.sigmas"][0]      # on its upper bound
    assert os.path.getsize(lc.FIXTURE) < 300 * 1024


def test_salsa512_fixture_is_what_the_oracle_gives():
    import sbtv_oracle as o
    fx = _fx()
    pr = lc.salsa512()
    st = o.demo_setup("gaussian", pr["x"], np.zeros_like(pr["x"]), evMax=1.0, true_params=pr["w"])
    st["y"] = pr["y"]
    res = o.salsa_from_estimates(st, pr["theta"], pr["w"], pr["sigma"] ** 2, tol=pr["tol"], outeriters=pr["maxiter"],
                                 TViters=pr["TViters"])
    assert res["n_outer"] == int(fx["salsa512.n_outer"])
    np.testing.assert_allclose(res["objective"], fx["salsa512.objective"], rtol=1e-12)
    np.testing.assert_allclose(res["mses"], fx["salsa512.mses"], rtol=1e-12)
    assert abs(o.PSNR(pr["x"], res["x"]) - float(fx["salsa512.psnr"])) < 1e-9
    for arr in ("x", "u", "bu"):
        for name, (si, sj) in lc.crops(512, 512).items():
            np.testing.assert_allclose(res[arr][si, sj], fx[f"salsa512.{arr}.{name}"], rtol=0, atol=1e-9)
