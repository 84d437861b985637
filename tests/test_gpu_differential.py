"""Differential tests of the launch-saving modes against the exact modes on seeded random configurations.

The solver loops have two ways of running the same arithmetic: the default (optimistic Chambolle launches whose stop
rule `cont = (k<MaxIter) & (err>tol)`, chambolle_prox_TV_stop.m:131, is applied afterwards; the collector riding on
the next launch; the host one iteration behind; the device-resident SAPG loop) and the exact one (a stop-rule kernel
after every launch, one synchronisation per iteration, the host-side loop).  Both must produce the SAME BITS whatever
the size, batch, number of TV iterations, stopping criterion, start and stopping iteration - including the
configurations in which the rule does fire inside a prox (small lambda) and the optimistic path has to fall back.
"""
import numpy as np
import pytest

from conftest import synth_image

pytestmark = pytest.mark.gpu

SIZES = [(32, 32), (64, 48), (96, 160), (128, 128), (256, 192), (512, 256), (60, 44), (100, 36)]


def _salsa_config(seed):
    rng = np.random.default_rng(1000 + seed)
    M, N = SIZES[rng.integers(len(SIZES))]
    B = int(rng.integers(1, 4))
    kind = ["gaussian", "moffat", "laplace"][rng.integers(3)]
    params = {"gaussian": (0.4, 0.3), "moffat": (0.5, 4.0), "laplace": (0.35,)}[kind]
    xs = np.stack([synth_image(M, N, 50 + seed * 7 + b) * rng.uniform(0.3, 1.0) for b in range(B)])
    return dict(M=M, N=N, B=B, kind=kind, params=params, xs=xs, rng=rng,
                tviters=int(rng.choice([1, 3, 5, 7, 10, 12, 20, 25])), crit=int(rng.choice([1, 2, 3])),
                maxiter=int(rng.choice([1, 2, 3, 9, 30])), init=rng.choice([0, 2, 1]),
                theta=rng.uniform(0.01, 0.2, size=B))


@pytest.mark.parametrize("seed", range(14))
def test_salsa_default_and_exact_launch_modes_give_the_same_bits(ctx, seed):
    import sbtv
    c = _salsa_config(seed)
    rng, B = c["rng"], c["B"]
    sts = [sbtv.demo_setup(c["kind"], c["xs"][b], rng.standard_normal((c["M"], c["N"])), evMax=1.0, true_params=c["params"])
           for b in range(B)]
    A = sbtv.BlurOperator(sbtv.psf_family(c["kind"], 7, c["params"])[0])
    y = np.stack([st["y"] for st in sts])
    tau = [float(th * st["sigma"] ** 2) for th, st in zip(c["theta"], sts)]
    mu = [float(th / 10) for th in c["theta"]]
    # criterion 3 compares the objective itself with the tolerance (SALSA_v2.m:470-472): put the threshold inside the
    # range the objective passes through so that images stop at different iterations
    tol = 1e-4 if c["crit"] != 3 else None
    outs = []
    for spec in (1, 3, 0, 2):
        args = [y, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", c["xs"], "StopCriterion", c["crit"],
                "MAXITERA", c["maxiter"], "TVINITIALIZATION", 1, "TViters", c["tviters"], "INITIALIZATION", int(c["init"]),
                "SPECULATE", spec]
        if tol is None:
            # first run: find the objective range with an unreachable tolerance, then stop half way
            probe = sbtv.SALSA_v2(*args, "ToleranceA", 0.0) if not outs else None
            if probe is not None:
                objs = probe[3] if B > 1 else [probe[3]]
                tol3 = float(np.median([0.5 * (o[0] + o[-1]) for o in objs]))
            args += ["ToleranceA", tol3]
        else:
            args += ["ToleranceA", tol]
        outs.append(sbtv.SALSA_v2(*args))
    ref = outs[0]
    for o in outs[1:]:
        np.testing.assert_array_equal(np.asarray(o[0]), np.asarray(ref[0]))
        np.testing.assert_array_equal(np.asarray(o[1]), np.asarray(ref[1]))
        np.testing.assert_array_equal(np.asarray(o[2]), np.asarray(ref[2]))
        for field in (3, 4, 6):
            a, r = (o[field], ref[field]) if B > 1 else ([o[field]], [ref[field]])
            assert len(a) == len(r)
            for ab, rb in zip(a, r):
                np.testing.assert_array_equal(np.asarray(ab), np.asarray(rb))


def test_salsa_small_lambda_rule_fires_inside_the_prox(ctx):
    """Nearly flat observations (a constant plus structure of amplitude 1e-6 ... 1e-2): the Chambolle error falls below
    its tolerance after a few iterations, at a k that varies with the image and the outer iteration, so the optimistic
    launches over-run and the solve must fall back to the exact launches - with the same bits as a solve that used
    them from the start."""
    import sbtv
    import sbtv._lib as L
    fired = 0
    for seed in range(10):
        rng = np.random.default_rng(2000 + seed)
        M, N = SIZES[rng.integers(len(SIZES))]
        B = int(rng.integers(1, 3))
        y = np.stack([np.full((M, N), rng.uniform(10, 200)) + 10.0 ** rng.uniform(-6, -2) * (synth_image(M, N, seed + b) - 128)
                      for b in range(B)])
        A = sbtv.BlurOperator(sbtv.psf_family("gaussian", 7, (0.4, 0.3))[0])
        mu, tau = 0.003, float(rng.uniform(0.05, 0.5))
        tv = int(rng.choice([5, 10, 15]))
        outs = []
        for spec in (1, 3, 0, 2):
            outs.append(sbtv.SALSA_v2(y, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "StopCriterion", 1, "ToleranceA", 1e-12,
                                      "MAXITERA", 7, "TVINITIALIZATION", 1, "TViters", tv, "SPECULATE", spec))
            if spec == 1:
                # Chambolle iterations booked per image: fewer than 7 x TViters = the rule stopped a prox early
                fired += L.default_context().last_timing()["chambolle_launches"] < 7 * tv - (tv - 1) - 0.5
        for o in outs[1:]:
            np.testing.assert_array_equal(np.asarray(o[0]), np.asarray(outs[0][0]), err_msg="seed %d" % seed)
            a, r = (o[3], outs[0][3]) if B > 1 else ([o[3]], [outs[0][3]])
            for ab, rb in zip(a, r):
                np.testing.assert_array_equal(np.asarray(ab), np.asarray(rb), err_msg="seed %d" % seed)
    assert fired >= 4, "the configurations are meant to make the rule fire inside a prox (it did in %d of 10)" % fired


@pytest.mark.parametrize("seed", range(8))
def test_fista_optimistic_and_exact_prox_give_the_same_bits(ctx, seed):
    import sbtv
    rng = np.random.default_rng(3000 + seed)
    M, N = SIZES[rng.integers(6)]                      # this entry point: power-of-two-friendly sizes of the list
    B = int(rng.integers(1, 3))
    kind = ["gaussian", "moffat", "laplace"][rng.integers(3)]
    params = {"gaussian": (0.4, 0.3), "moffat": (0.5, 4.0), "laplace": (0.35,)}[kind]
    flat = seed % 4 == 3                               # every fourth case: a nearly flat image (the rule fires early)
    xs = np.stack([(np.full((M, N), 90.0) + 1e-3 * synth_image(M, N, seed + b)) if flat else synth_image(M, N, 70 + seed + b)
                   for b in range(B)])
    sts = [sbtv.demo_setup(kind, xs[b], rng.standard_normal((M, N)) * (1e-3 if flat else 1.0), evMax=1.0, true_params=params)
           for b in range(B)]
    A = sbtv.BlurOperator(sbtv.psf_family(kind, 7, params)[0])
    y = np.stack([st["y"] for st in sts])
    tau = [float(rng.uniform(0.01, 0.1) * st["sigma"] ** 2) for st in sts]
    K = int(rng.choice([5, 10, 25, 30]))
    crit = int(rng.choice([1, 2]))
    tol = float(rng.choice([1e-3, 1e-5]))
    nit = int(rng.choice([2, 6, 25]))
    a = sbtv.my_fista(y, A, A.T, tau, 1.0, sbtv.TVnorm, sbtv.Psi_TV(K), crit, tol, nit, xs)
    e = sbtv.my_fista(y, A, A.T, tau, 1.0, sbtv.TVnorm, sbtv.Psi_TV(K), crit, tol, nit, xs, exact_prox=True)
    np.testing.assert_array_equal(np.asarray(a[0]), np.asarray(e[0]))
    for field in (1, 3):
        av, ev = (a[field], e[field]) if B > 1 else ([a[field]], [e[field]])
        assert len(av) == len(ev)
        for ab, eb in zip(av, ev):
            np.testing.assert_array_equal(np.asarray(ab), np.asarray(eb))


@pytest.mark.parametrize("seed", range(6))
def test_sapg_device_loop_and_host_loop_give_the_same_bits_with_fixed_psf(ctx, seed):
    import sbtv
    import sbtv_oracle as o
    from test_gpu_sapg_fista import _op_struct
    rng = np.random.default_rng(4000 + seed)
    M, N = SIZES[rng.integers(6)]
    kind = ["gaussian", "moffat", "laplace"][rng.integers(3)]
    B = int(rng.integers(1, 4))
    x = synth_image(M, N, 20 + seed)
    st = o.demo_setup(kind, x, rng.standard_normal((M, N)), evMax=0.99)
    warmup, burnIn = int(rng.integers(2, 8)), int(rng.integers(2, 8))
    samples = burnIn + int(rng.integers(4, 24))
    op, c, names = _op_struct(kind, st, samples, warmup, burnIn)
    op["seed"] = int(seed)
    for nm in names:
        op["fix_" + nm] = 1
    fn = {"gaussian": sbtv.SAPG_algorithm_Guassian, "moffat": sbtv.SAPG_algorithm_moffat,
          "laplace": sbtv.SAPG_algorithm_laplace}[kind]
    ys = np.stack([st["y"] if b % 2 == 0 else st["y"][::-1, ::-1].copy() for b in range(B)])
    dev = fn(ys, op, c)[-1]
    host = fn(ys, op, c, host_loop=True)[-1]
    if B == 1 and isinstance(dev, dict):
        dev, host = [dev], [host]
    for b in range(B):
        for key in ("thetas", "sigmas", "logPiTraceX", "gXTrace", "grad_theta", "grad_sigma", "theta_EB", "sigma_EB",
                    "Xlast_sample"):
            np.testing.assert_array_equal(np.asarray(dev[b][key]), np.asarray(host[b][key]), err_msg=key)


def test_shortest_calls_device_resident(ctx):
    """One and two iterations with device-resident images: the SALSA iteration numbered MAXITERA writes x straight into
    x_out, FISTA evaluates its stopping rule one iteration late - the shortest calls are their corner cases.  Checked
    against the same calls on host arrays (which go through the staging copies and the copy out of the double buffer)."""
    import sbtv
    M, N = 128, 96
    x = synth_image(M, N, 3)
    st = sbtv.demo_setup("gaussian", x, np.random.default_rng(8).standard_normal((M, N)), evMax=1.0)
    A = sbtv.BlurOperator(sbtv.psf_family("gaussian", 7, (0.4, 0.3))[0])
    mu, tau = 0.003, 0.03 * st["sigma"] ** 2
    yd, xd = sbtv.to_device(st["y"]), sbtv.to_device(x)
    for K in (1, 2, 3):
        for init in (0, 2):
            h = sbtv.SALSA_v2(st["y"], A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", x, "ToleranceA", -1.0,
                              "MAXITERA", K, "TVINITIALIZATION", 1, "TViters", 10, "INITIALIZATION", init)
            d = sbtv.SALSA_v2(yd, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xd, "ToleranceA", -1.0,
                              "MAXITERA", K, "TVINITIALIZATION", 1, "TViters", 10, "INITIALIZATION", init)
            assert len(h[3]) == K + 1 == len(d[3])
            np.testing.assert_array_equal(sbtv.to_host(d[0]), h[0])
            np.testing.assert_array_equal(d[3], h[3])
            np.testing.assert_array_equal(d[6], h[6])
        hf = sbtv.my_fista(st["y"], A, A.T, tau, 1.0, sbtv.TVnorm, sbtv.Psi_TV(25), 1, 0.0, K, x)
        df = sbtv.my_fista(yd, A, A.T, tau, 1.0, sbtv.TVnorm, sbtv.Psi_TV(25), 1, 0.0, K, xd)
        assert len(hf[1]) == K == len(df[1])
        np.testing.assert_array_equal(sbtv.to_host(df[0]), hf[0])
        np.testing.assert_array_equal(df[1], hf[1])
    # an output that overlaps an input falls back to the copy: x_out = the `true` image's memory
    # (the Python mirror always allocates its own output, so this goes through the C-ABI directly)
    import ctypes as C
    import sbtv._lib as L
    c = L.default_context()
    so = L.sbtv_salsa_opts()
    c.lib.sbtv_salsa_opts_default(C.byref(so))
    so.maxiter, so.TViters, so.tolA = 3, 10, -1.0
    ref = sbtv.SALSA_v2(yd, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xd, "ToleranceA", -1.0, "MAXITERA", 3,
                        "TVINITIALIZATION", 1, "TViters", 10)
    yi, ti = L.Images(yd), L.Images(xd.clone())
    taps = A._cm(1)
    tau_a, tau_p = L.dvec(tau, 1)
    mu_a, mu_p = L.dvec(mu, 1)
    obj = np.zeros(4)
    nout = (C.c_int * 1)()
    c.check(c.lib.sbtv_SALSA_v2(c.h, yi.ptr, M, N, 1, L.vptr(taps), A.taille, tau_p, mu_p, C.byref(so), ti.ptr, None, ti.ptr,
                                L.vptr(obj), None, None, None, None, None, nout, yi.flags), yi.flags)
    np.testing.assert_array_equal(sbtv.to_host(ti.t[0]), sbtv.to_host(ref[0]))
    np.testing.assert_array_equal(obj, ref[3])
