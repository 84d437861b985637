"""CPU side of the long-chain SAPG fixtures (`tests/golden/sapg_long.npz`, made by `make_golden_sapg.py`): inventory, the
properties that make the cases worth having (projections engage and release under the reference's step scales), and the
per-step case of every PSF family recomputed by the oracle (seconds), so that a change of the oracle or of the inputs
shows up on every CPU run.  The 24 statistical chains (~8 s each) are regenerated with the script only."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import sapg_cases as sc  # noqa: E402


@pytest.fixture(scope="module")
def fx():
    with np.load(sc.FIXTURE) as f:       # plain arrays: no pickle involved
        return {k: f[k] for k in f.files}


def test_fixture_inventory(fx):
    T, S = sc.TRACE, sc.STAT
    assert T["samples"] >= 300 and S["samples"] >= 1500 and S["chains"] >= 8
    for kind in sc.KINDS:
        npar = len(sc.NAMES[kind])
        t = f"trace.{kind}"
        assert fx[f"{t}.thetas"].shape == (T["samples"],) and fx[f"{t}.ps"].shape == (npar, T["samples"])
        assert fx[f"{t}.grads"].shape == (npar + 2, T["samples"]) and fx[f"{t}.eb"].shape == (npar + 2,)
        assert fx[f"{t}.mean_thetas"].shape == (T["samples"] - T["burnIn"],)
        assert fx[f"{t}.X"].shape == (sc.SIZE, sc.SIZE)
        h = int(fx[f"{t}.horizon"])
        assert 30 <= h < T["samples"] and fx[f"{t}.sens"][-40:].max() > 1e-3      # all free: chaotic after the horizon
        assert fx[f"trace_fs.{kind}.sens"].max() < 3e-10                           # sigma^2 fixed: stable throughout
        assert np.all(fx[f"trace_fs.{kind}.sigmas"] == fx[f"trace_fs.{kind}.sigmas"][0])
        eb = fx[f"stat.{kind}.eb"]
        assert eb.shape == (S["chains"], npar + 2) and np.all(np.isfinite(eb))
        assert np.all(eb.std(0, ddof=1) > 0)                       # eight different chains
    assert os.path.getsize(sc.FIXTURE) < 600 * 1024


@pytest.mark.parametrize("kind", sc.KINDS)
def test_trace_case_is_what_the_oracle_gives_and_projections_engage_and_release(fx, kind):
    import sbtv_oracle as o
    o.set_workers(1)
    st = sc.setup(kind)
    T = sc.TRACE
    it = iter(sc.trace_noise(kind))
    fr = sc.FREE[kind]
    r = o.SAPG_algorithm(st, samples=T["samples"], warmup=T["warmup"], burnIn=T["burnIn"], randn=lambda s: next(it),
                         fix=fr["fix"], p_init=fr["p_init"])
    t = f"trace.{kind}"
    np.testing.assert_allclose(r["thetas"], fx[f"{t}.thetas"], rtol=1e-12)
    np.testing.assert_allclose(r["ps"], fx[f"{t}.ps"], rtol=1e-12)
    np.testing.assert_allclose(r["sigmas"], fx[f"{t}.sigmas"], rtol=1e-12)
    np.testing.assert_allclose(r["logPiTraceX"], fx[f"{t}.logPi"], rtol=1e-12)
    assert r["theta_EB"] == pytest.approx(fx[f"{t}.eb"][0], rel=1e-12)
    # the reference's step scales (SAPG_algorithm_moffat.m:135-138, _laplace.m:139-141, run_Gaussian_demo.m:34-39) throw
    # the PSF parameters and sigma^2 onto their bounds early in the chain (min(max(.)) of SAPG_algorithm_Guassian.m:166-194)
    # and the decaying step delta(i) lets them go again
    d = o.DEMO[kind]
    lo, hi = min(st["sigma_min"], st["sigma_max"]), max(st["sigma_min"], st["sigma_max"])
    on_s = (r["sigmas"] == lo) | (r["sigmas"] == hi)
    assert on_s[:100].any() and not on_s[-20:].all()
    for q in range(len(d["true"])):
        on_p = (r["ps"][q] == d["pmin"][q]) | (r["ps"][q] == d["pmax"][q])
        assert on_p[:100].any() and not on_p[-20:].any()
        assert np.any(np.diff(on_p.astype(int)) == -1)             # released at least once
    # mean_* / tol_* logs as the reference keeps them (:217-244): spot values written out
    th, b = r["thetas"], T["burnIn"]
    assert fx[f"{t}.mean_thetas"][0] == pytest.approx(np.mean(th[b - 1:b + 1]), rel=1e-13)
    assert fx[f"{t}.mean_thetas"][-1] == pytest.approx(np.mean(th[b - 1:]), rel=1e-13) == pytest.approx(r["theta_EB"], rel=1e-13)
    assert np.all(np.isnan(fx[f"{t}.tol_thetas"][1:b])) and fx[f"{t}.tol_thetas"][0] == 0.0
    # the fixed-sigma twin
    it = iter(sc.trace_noise(kind))
    r = o.SAPG_algorithm(st, samples=T["samples"], warmup=T["warmup"], burnIn=T["burnIn"], randn=lambda s: next(it),
                         fix=fr["fix"], p_init=fr["p_init"], fix_sigma=True)
    np.testing.assert_allclose(r["thetas"], fx[f"trace_fs.{kind}.thetas"], rtol=1e-12)
    np.testing.assert_allclose(r["ps"], fx[f"trace_fs.{kind}.ps"], rtol=1e-12)


def test_re_anchored_segment_continues_the_long_chain():
    """X0 / sigma_init / iter_offset / keep_X of the oracle (what the GPU segment test leans on): a segment started from
    the long chain's state at iteration s reproduces the long chain's next parameter values but for the theta used by
    its first prox (theta(s) instead of theta(s-1)) - and exactly so for the quantities that do not depend on it."""
    import sbtv_oracle as o
    o.set_workers(1)
    kind = "laplace"
    st = sc.setup(kind)
    nz = sc.trace_noise(kind)[:40]
    it = iter(nz)
    long = o.SAPG_algorithm(st, samples=30, warmup=12, burnIn=20, randn=lambda s: next(it), keep_X={1, 17})
    assert sorted(long["X_at"]) == [1, 17]
    s, L = 17, 5
    seg_nz = nz[12 - 1 + s - 1: 12 - 1 + s - 1 + L]
    it2 = iter(seg_nz)
    seg = o.SAPG_algorithm(dict(st, th_init=long["thetas"][s - 1]), samples=L + 1, warmup=0, burnIn=1,
                           randn=lambda z: next(it2), p_init=tuple(long["ps"][:, s - 1]), sigma_init=long["sigmas"][s - 1],
                           X0=long["X_at"][s], iter_offset=s - 1)
    # delta(ii + offset): theta's first update uses G_t of the new X, which depends on the first prox only through X
    np.testing.assert_allclose(seg["thetas"][1:], long["thetas"][s:s + L], rtol=2e-2)
    np.testing.assert_allclose(seg["ps"][0, 1:], long["ps"][0, s:s + L], rtol=2e-2)
    # with the SAME first prox the continuation is exact: start one iteration earlier is not possible without theta(s-2),
    # so check exactness where the prox does not enter - the step size itself
    d0 = st["d_scale"] * ((2 + s - 1) ** (-st["d_exp"])) / st["dimX"]
    g = seg["grads"][0, 1]
    assert seg["thetas"][1] == min(max(long["thetas"][s - 1] + o.DEMO[kind]["c_theta"] * d0 * g, 1e-3), 1.0)
