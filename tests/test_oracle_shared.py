"""CPU: the shared-gradient SAPG restatement (configs[4]) is pinned to the single-chain one."""
import numpy as np

from conftest import synth_image


def test_shared_oracle_with_one_chain_is_the_single_chain_oracle():
    """`mean(g_*)` over one sample is the sample (SAPG_algorithm_moffat.m:158-173 with `for jj = 1:1`)."""
    import sbtv_oracle as o
    M = N = 24
    x = synth_image(M, N, 5)
    rng = np.random.default_rng(3)
    for kind in ("moffat", "laplace"):
        st = o.demo_setup(kind, x, rng.standard_normal((M, N)), evMax=0.99)
        samples, warmup, burnIn = 6, 3, 3
        nz = rng.standard_normal((warmup - 1 + samples - 1, M, N))
        it = iter(nz)
        one = o.SAPG_algorithm(st, samples, warmup, burnIn, lambda s: next(it))
        it2 = iter(nz)
        sh = o.SAPG_algorithm_shared(st, 1, samples, warmup, burnIn, lambda s, k: next(it2))
        for key in ("thetas", "sigmas", "ps", "grads"):
            np.testing.assert_array_equal(sh[key], one[key])
        np.testing.assert_array_equal(sh["logPiTraceX"][0], one["logPiTraceX"])
        np.testing.assert_array_equal(sh["Xlast_samples"][0], one["Xlast_sample"])


def test_shared_oracle_averages_the_chain_gradients():
    """Two chains fed the SAME noise are the same chain twice: the mean of two equal gradients is that gradient, so
    the trajectory equals the single-chain one; with different noise the parameter paths differ from both."""
    import sbtv_oracle as o
    M = N = 24
    x = synth_image(M, N, 6)
    rng = np.random.default_rng(4)
    st = o.demo_setup("laplace", x, rng.standard_normal((M, N)), evMax=0.99)
    samples, warmup, burnIn = 5, 2, 3
    nz = rng.standard_normal((warmup - 1 + samples - 1, 2, M, N))
    cnt = [0, 0]

    def same(shape, k):
        z = nz[cnt[k], 0]
        cnt[k] += 1
        return z
    it = iter(nz[:, 0])
    one = o.SAPG_algorithm(st, samples, warmup, burnIn, lambda s: next(it))
    two = o.SAPG_algorithm_shared(st, 2, samples, warmup, burnIn, same)
    np.testing.assert_allclose(two["thetas"], one["thetas"], rtol=1e-14)
    np.testing.assert_allclose(two["ps"], one["ps"], rtol=1e-14)
    cnt[:] = [0, 0]

    def own(shape, k):
        z = nz[cnt[k], k]
        cnt[k] += 1
        return z
    mixed = o.SAPG_algorithm_shared(st, 2, samples, warmup, burnIn, own)
    assert not np.allclose(mixed["thetas"], one["thetas"], rtol=1e-9, atol=0)
    assert np.max(np.abs(mixed["Xlast_samples"][0] - mixed["Xlast_samples"][1])) > 1e-3
