"""GPU: the two lanes of a context (include/sbtv.h `sbtv_ctx_set_lanes`, csrc/group.hip) and the sharded variants of the
remaining loops.

A call with batch >= 2 independent items is dealt in two contiguous halves to two internal contexts on the same GPU (own
stream, workspaces, host thread).  Nothing about the results may change: image k of a batched call is computed bit for bit
like image k alone, so lanes on (default) == lanes off (`set_lanes(1)`), traces, counters and all.  The same holds for
`sbtv_fista_tv_sharded`, `sbtv_CSALSA_v2_sharded`, `sbtv_CoRAL_v2_sharded` over virtual shards (SALSA/my_fista.m:5,
SALSA/CSALSA_v2.m:160, SALSA/CoRAL_v2.m:2: independent images, no exchange)."""
import numpy as np
import pytest

from conftest import synth_image
from test_gpu_group import _salsa_problem, _sapg_op

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx1():
    """A context of its own with the lanes switched off (one stream), closed after the test."""
    import sbtv
    c = sbtv.Context(0)
    c.set_lanes(1)
    yield c
    c.close()


@pytest.fixture()
def ctx2():
    """A fresh context with the default policy (two lanes for independent items)."""
    import sbtv
    c = sbtv.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def group2():
    import sbtv
    g = sbtv.Group([0, 0])
    yield g
    g.close()


def _salsa_args(A, xs):
    return ("MU", 0.003, "AT", A.T, "LS", A.LS(0.003), "True_x", xs, "ToleranceA", 1e-4, "MAXITERA", 80,
            "TVINITIALIZATION", 1, "TViters", 10)


@pytest.mark.parametrize("n", [2, 3, 5])
def test_salsa_batch_in_two_lanes_is_bit_equal_to_one_stream(ctx1, ctx2, n):
    import sbtv
    xs, ys, taus = _salsa_problem(n)
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
    ctx1.reset_calls()
    ctx2.reset_calls()
    one = sbtv.SALSA_v2(ys, A, taus, *_salsa_args(A, xs), ctx=ctx1)
    two = sbtv.SALSA_v2(ys, A, taus, *_salsa_args(A, xs), ctx=ctx2)
    np.testing.assert_array_equal(two[0], one[0])
    assert list(two[1]) == list(one[1]) and list(two[2]) == list(one[2])            # numA, numAt per image
    for b in range(n):
        for k in (3, 4, 6):                                                          # objective, distance, mses
            np.testing.assert_array_equal(two[k][b], one[k][b])
    assert ctx2.calls == ctx1.calls > 0                                              # callcounter.m total of the batch
    t1, t2 = ctx1.last_timing(), ctx2.last_timing()
    assert t2["loop_ms"] > 0 and t2["chambolle_launches"] == pytest.approx(t1["chambolle_launches"], rel=1e-12)
    assert t2["chambolle_bytes"] == t1["chambolle_bytes"]
    # the single image never takes the lanes
    s1 = sbtv.SALSA_v2(ys[0], A, taus[0], *_salsa_args(A, xs[0]), ctx=ctx2)
    np.testing.assert_array_equal(s1[0], one[0][0])


def test_salsa_device_resident_batch_in_two_lanes(ctx1, ctx2):
    """torch tensors (SBTV_DEVICE_PTRS): a lane's block is an offset into the caller's device arrays."""
    import sbtv
    xs, ys, taus = _salsa_problem(4, 128, 128)
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
    yd, xd = sbtv.to_device(ys), sbtv.to_device(xs)
    one = sbtv.SALSA_v2(yd, A, taus, *_salsa_args(A, xd), ctx=ctx1)
    two = sbtv.SALSA_v2(yd, A, taus, *_salsa_args(A, xd), ctx=ctx2)
    np.testing.assert_array_equal(sbtv.to_host(two[0]), sbtv.to_host(one[0]))
    host = sbtv.SALSA_v2(ys, A, taus, *_salsa_args(A, xs), ctx=ctx2)
    np.testing.assert_array_equal(host[0], sbtv.to_host(one[0]))
    for b in range(4):
        np.testing.assert_array_equal(two[3][b], one[3][b])


def test_lane_error_reaches_the_caller(ctx2):
    """A failure inside a lane comes back as the call's status with the lane's message."""
    import sbtv
    xs, ys, taus = _salsa_problem(3)
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
    with pytest.raises(sbtv.SbtvError) as e:
        sbtv.SALSA_v2(ys, A, taus, "MU", 0.003, "AT", A.T, "LS", A.LS(0.003), "MAXITERA", 5, "TVINITIALIZATION", 1,
                      "TViters", 0, ctx=ctx2)
    assert "TViters" in str(e.value)


def _sapg_case(nch):
    import sbtv_oracle as o
    M = N = 32
    sts = [o.demo_setup("laplace", synth_image(M, N, 30 + b), np.random.default_rng(b).standard_normal((M, N)), evMax=0.99)
           for b in range(nch)]
    op, c, names = _sapg_op("laplace", sts[0], 14, 5, 8)
    op["seed"] = 11
    return np.stack([st["y"] for st in sts]), op, c


@pytest.mark.parametrize("noise", ["philox", "injected"])
def test_sapg_independent_chains_in_two_lanes_bit_equal(ctx1, ctx2, noise):
    import sbtv
    y, op, c = _sapg_case(5)
    nz = None
    if noise == "injected":
        nz = np.random.default_rng(9).standard_normal((op["warmup"] - 1 + op["samples"] - 1, 5, 32, 32))
    one = sbtv.SAPG_algorithm_laplace(y, op, c, noise=nz, ctx=ctx1)[-1]
    two = sbtv.SAPG_algorithm_laplace(y, op, c, noise=nz, ctx=ctx2)[-1]
    for b in range(5):
        for key in ("thetas", "bs", "sigmas", "logPiTraceX", "gXTrace", "Xlast_sample", "logPiTrace_WU", "grad_b"):
            np.testing.assert_array_equal(two[b][key], one[b][key], err_msg=f"{b}:{key}")
        assert two[b]["theta_EB"] == one[b]["theta_EB"]
    assert not np.array_equal(one[0]["thetas"], one[1]["thetas"])


def test_sapg_shared_chains_split_only_on_request(ctx1, ctx2):
    """share_gradients = 1: default policy keeps the chains on one stream (bit-equal to lanes off); set_lanes(2) splits them
    2 + 2 with the in-stream exchange of group.hip (sum order differs: rtol 1e-11)."""
    import sbtv
    import sbtv_oracle as o
    M = N = 32
    st = o.demo_setup("gaussian", synth_image(M, N, 3), np.random.default_rng(2).standard_normal((M, N)), evMax=0.99)
    op, c, names = _sapg_op("gaussian", st, 10, 4, 6)
    op.update(chains=4, seed=3, fix_w1=0, fix_w2=0, w1_init=0.5, w2_init=0.35)
    c.update(w1=0.3, w2=0.3, sigma=100.0)
    one = sbtv.SAPG_algorithm_Guassian(st["y"], op, c, share_gradients=True, ctx=ctx1)[-1]
    dflt = sbtv.SAPG_algorithm_Guassian(st["y"], op, c, share_gradients=True, ctx=ctx2)[-1]
    ctx2.set_lanes(2)
    split = sbtv.SAPG_algorithm_Guassian(st["y"], op, c, share_gradients=True, ctx=ctx2)[-1]
    for k in range(4):
        for key in ("thetas", "w1s", "w2s", "sigmas", "logPiTraceX"):
            np.testing.assert_array_equal(dflt[k][key], one[k][key])
            np.testing.assert_allclose(split[k][key], one[k][key], rtol=1e-11, err_msg=f"{k}:{key}")
    assert not np.array_equal(one[0]["logPiTraceX"], one[3]["logPiTraceX"])
    assert one[0]["w1s"][-1] != one[0]["w1s"][0]


def _fista_problem(n, M=64, N=64):
    import sbtv_oracle as o
    xs, ys, taus = [], [], []
    for b in range(n):
        x = synth_image(M, N, 40 + b)
        st = o.demo_setup("moffat", x, np.random.default_rng(b).standard_normal(x.shape), evMax=1.0)
        xs.append(x)
        ys.append(st["y"])
        taus.append(0.03 * st["sigma"] ** 2 * (1 + 0.3 * b))
    return np.stack(xs), np.stack(ys), taus


def test_fista_lanes_and_sharded_are_bit_equal_to_one_stream(ctx1, ctx2, group2):
    import sbtv
    xs, ys, taus = _fista_problem(3)
    A = sbtv.BlurOperator(sbtv.psf_moffat(7, 0.4, 3.5))
    run = lambda c: sbtv.my_fista(ys, A, A.T, taus, 1.0, sbtv.TVnorm, sbtv.Psi_TV(25), 1, 1e-4, 30, xs, ctx=c)
    one, two, shd = run(ctx1), run(ctx2), run(group2)
    for got in (two, shd):
        np.testing.assert_array_equal(got[0], one[0])
        for b in range(3):
            np.testing.assert_array_equal(got[1][b], one[1][b])          # objective traces


def test_csalsa_and_coral_lanes_and_sharded_are_bit_equal_to_one_stream(ctx1, ctx2, group2):
    import sbtv
    xs, ys, taus = _salsa_problem(3, 64, 64)
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
    sig = [float(np.std(ys[b] - xs[b])) for b in range(3)]
    common = ("AT", A.T, "TVINITIALIZATION", 1, "TVITERS", 10, "TRUE_X", xs, "MAXITERA", 40, "TOLERANCEA", 1e-4)
    cs = lambda c: sbtv.csalsa(ys, A, 1.0, 1.0, sig, "LS", A.invLS, *common, "STOPCRITERION", 3, ctx=c)
    co = lambda c: sbtv.CoRAL(ys, A, [0.4 * t for t in taus], [0.6 * t for t in taus], "MU1", 0.003, "MU2", 0.004,
                              "AT", A.T, "LS", A.LS(0.007), "TVINITIALIZATION1", 1, "TVITERS1", 10, "TVINITIALIZATION2", 1,
                              "TVITERS2", 10, "TRUE_X", xs, "MAXITERA", 40, "TOLERANCEA", 1e-4, "STOPCRITERION", 1, ctx=c)
    for fn in (cs, co):
        one, two, shd = fn(ctx1), fn(ctx2), fn(group2)
        for got in (two, shd):
            np.testing.assert_array_equal(got[0], one[0])
            for b in range(3):
                np.testing.assert_array_equal(got[3][b], one[3][b])      # objective traces


def test_lanes_stress_alternating_batches_and_entry_points(ctx1, ctx2):
    """Forty calls on ONE pair of contexts with changing batch sizes, image sizes and entry points (the lanes' workspaces grow,
    plans and order tables are rebuilt, host threads are started and joined every call): every result equals the one-stream
    context's bit for bit."""
    import sbtv
    rng = np.random.default_rng(123)
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
    for it in range(40):
        n = int(rng.integers(2, 7))
        M, N = [(64, 48), (128, 128), (96, 80), (256, 64)][int(rng.integers(0, 4))]
        xs, ys, taus = _salsa_problem(n, M, N)
        if it % 3 == 2:
            f = lambda c: sbtv.chambolle_prox_TV_stop(ys, "lambda", 8.0, "maxiter", 12, ctx=c)       # (no lanes: a single pass)
            a, b = f(ctx1), f(ctx2)
            np.testing.assert_array_equal(b[0], a[0])
            continue
        args = ("MU", 0.003, "AT", A.T, "LS", A.LS(0.003), "True_x", xs, "ToleranceA", 1e-3, "MAXITERA", 25,
                "TVINITIALIZATION", 1, "TViters", int(rng.integers(3, 11)))
        one = sbtv.SALSA_v2(ys, A, taus, *args, ctx=ctx1)
        two = sbtv.SALSA_v2(ys, A, taus, *args, ctx=ctx2)
        np.testing.assert_array_equal(two[0], one[0], err_msg=f"call {it}: {n} x {M} x {N}")
        for b in range(n):
            np.testing.assert_array_equal(two[3][b], one[3][b])


def test_two_host_threads_with_a_context_each_use_their_lanes_concurrently():
    """Two application threads, each with its own context (four lane threads, four streams on the one GPU), solving
    different batches at the same time: each gets the result of its solitary run."""
    import threading
    import sbtv
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3))
    probs = [_salsa_problem(3, 128, 128), _salsa_problem(4, 96, 80)]
    ctxs = [sbtv.Context(0), sbtv.Context(0)]
    try:
        def solve(i):
            xs, ys, taus = probs[i]
            return sbtv.SALSA_v2(ys, A, taus, "MU", 0.003, "AT", A.T, "LS", A.LS(0.003), "True_x", xs, "ToleranceA", 1e-4,
                                 "MAXITERA", 60, "TVINITIALIZATION", 1, "TViters", 10, ctx=ctxs[i])
        alone = [solve(0), solve(1)]
        got = [None, None]
        for rep in range(3):
            th = [threading.Thread(target=lambda i=i: got.__setitem__(i, solve(i))) for i in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            for i in range(2):
                np.testing.assert_array_equal(got[i][0], alone[i][0])
                for b in range(len(alone[i][3])):
                    np.testing.assert_array_equal(got[i][3][b], alone[i][3][b])
    finally:
        for c in ctxs:
            c.close()
