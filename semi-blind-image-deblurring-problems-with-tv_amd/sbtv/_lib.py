"""ctypes binding of libsbtv.so (the C-ABI declared in include/sbtv.h).

There is no CPU fallback anywhere in this package: if the shared library is
missing, or no gfx950 GPU is visible, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SBTV_LIBRARY selects another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("SBTV_LIBRARY") or os.path.join(os.path.dirname(_HERE), "lib", "libsbtv.so")

SBTV_HOST_PTRS = 0
SBTV_DEVICE_PTRS = 1

ERRORS = {
    -1: "SBTV_ERR_BADARG", -2: "SBTV_ERR_SIZE", -3: "SBTV_ERR_MAXITER", -4: "SBTV_ERR_DUALVARS",
    -5: "SBTV_ERR_MODE", -6: "SBTV_ERR_STOPCRITERION", -7: "SBTV_ERR_INIT", -8: "SBTV_ERR_MISSING_AT",
    -9: "SBTV_ERR_MISSING_LS", -10: "SBTV_ERR_PSF", -11: "SBTV_ERR_NOMEM", -12: "SBTV_ERR_NODEVICE",
    -13: "SBTV_ERR_CANARY", -14: "SBTV_ERR_PEER",
}


class SbtvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{ERRORS.get(code, code)}] {msg}")
        self.code = code
        self.msg = msg


class sbtv_salsa_opts(C.Structure):
    _fields_ = [("stopcriterion", C.c_int), ("maxiter", C.c_int), ("TViters", C.c_int),
                ("initialization", C.c_int), ("compute_mse", C.c_int), ("speculate", C.c_int),
                ("tolA", C.c_double), ("chambolle_tol", C.c_double), ("chambolle_tau", C.c_double)]


class sbtv_sapg_opts(C.Structure):
    _fields_ = [("kind", C.c_int), ("psf_size", C.c_int), ("samples", C.c_int), ("warmup", C.c_int),
                ("burnIn", C.c_int), ("chambolleit", C.c_int), ("fix_p", C.c_int * 2), ("fix_sigma", C.c_int),
                ("share_gradients", C.c_int),
                ("lambda_", C.c_double), ("gamma", C.c_double),
                ("th_init", C.c_double), ("min_th", C.c_double), ("max_th", C.c_double),
                ("p_init", C.c_double * 2), ("p_min", C.c_double * 2), ("p_max", C.c_double * 2),
                ("p_true", C.c_double * 2), ("phi", C.c_double),
                ("sigma2_init", C.c_double), ("sigma2_min", C.c_double), ("sigma2_max", C.c_double),
                ("sigma2_true", C.c_double),
                ("d_scale", C.c_double), ("d_exp", C.c_double),
                ("c_theta", C.c_double), ("c_p", C.c_double * 2), ("c_sigma", C.c_double),
                ("seed", C.c_ulonglong), ("chain_offset", C.c_int), ("iter_offset", C.c_int)]


def vptr(a):
    """void* of a NumPy array's data.  (numpy's `a.ctypes` builds a helper object on every access, ~25 us a time: with
    seven output arrays that was a third of the fixed cost of a SALSA call.)  The caller keeps `a` alive."""
    return C.c_void_p(a.__array_interface__["data"][0])


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
# sbtv_allreduce_dev_fn (flags & REDUCE_DEVICE): (user, device address of the 6 doubles, n, hipStream_t)
ALLREDUCE_DEV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
REDUCE_DEVICE = 2        # include/sbtv.h SBTV_REDUCE_DEVICE
SAPG_HOST_LOOP = 4       # include/sbtv.h SBTV_SAPG_HOST_LOOP
FISTA_EXACT_PROX = 2     # include/sbtv.h SBTV_FISTA_EXACT_PROX

_P = C.c_void_p
_D = C.c_double
_I = C.c_int

# name -> (restype, argtypes); the list is also what tests check against include/sbtv.h
SIGNATURES = {
    "sbtv_version": (_I, []),
    "sbtv_ctx_create": (_I, [_I, C.POINTER(_P)]),
    "sbtv_ctx_destroy": (_I, [_P]),
    "sbtv_last_error": (C.c_char_p, [_P]),
    "sbtv_ctx_set_stream": (_I, [_P, _P]),
    "sbtv_ctx_sync": (_I, [_P]),
    "sbtv_ctx_set_lanes": (_I, [_P, _I]),
    "sbtv_callcounter_get": (_I, [_P, C.POINTER(C.c_longlong)]),
    "sbtv_callcounter_reset": (_I, [_P]),
    "sbtv_last_timing": (_I, [_P, C.POINTER(_D)]),
    "sbtv_malloc": (_I, [_P, C.c_size_t, C.POINTER(_P)]),
    "sbtv_free": (_I, [_P, _P]),
    "sbtv_memcpy_h2d": (_I, [_P, _P, _P, C.c_size_t]),
    "sbtv_memcpy_d2h": (_I, [_P, _P, _P, C.c_size_t]),
    "sbtv_chambolle_prox_TV_stop": (_I, [_P, _P, _I, _I, _I, _P, _I, _D, _D, _I, _P, _P, _P, _P, _P, _I]),
    "sbtv_TVnorm": (_I, [_P, _P, _I, _I, _I, _P, _I]),
    "sbtv_A_wrapper": (_I, [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I]),
    "sbtv_psf_taps": (_I, [_I, _I, _P, _P, _P, _P]),
    "sbtv_err_psf": (_I, [_I, _I, _P, _I, _P, _D, _P]),
    "sbtv_rfft2_packed": (_I, [_P, _P, _P, _I, _I, _I, _I, _I]),
    "sbtv_salsa_opts_default": (None, [C.POINTER(sbtv_salsa_opts)]),
    "sbtv_SALSA_v2": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, C.POINTER(sbtv_salsa_opts), _P, _P, _P, _P, _P, _P,
                           _P, _P, _P, _P, _I]),
    "sbtv_CSALSA_v2": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _D, C.POINTER(sbtv_salsa_opts), _P, _P, _P,
                            _P, _P, _P, _P, _P, _P, _P, _P, _P, _I]),
    "sbtv_CoRAL_v2": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, C.POINTER(sbtv_salsa_opts), _P, _P, _P,
                           _P, _P, _P, _P, _P, _P, _P, _I]),
    "sbtv_fista_tv": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _D, _I, _I, _D, _I, _I, _P, _P, _P, _P, _P, _I]),
    "sbtv_SAPG_algorithm": (_I, [_P, _P, _I, _I, _I, C.POINTER(sbtv_sapg_opts), _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                 _P, _P, ALLREDUCE_FN, _P, _I]),
    "sbtv_myula": (_I, [_P, _P, _I, _I, _I, _P, _I, _D, _D, _P, _P, _I, _I, C.c_ulonglong, _I, _P, _P, _I]),
    "sbtv_max_eigenval": (_I, [_P, _P, _I, _P, _I, _I, _D, _I, _P, _P, _I]),
    "sbtv_PSNR": (_I, [_P, _P, _P, _I, _I, _I, _P, _I]),
    "sbtv_MSE": (_I, [_P, _P, _P, _I, _I, _I, _P, _I]),
    "sbtv_group_create": (_I, [C.POINTER(_I), _I, C.POINTER(_P)]),
    "sbtv_group_destroy": (_I, [_P]),
    "sbtv_group_size": (_I, [_P]),
    "sbtv_group_ctx": (_P, [_P, _I]),
    "sbtv_group_last_error": (C.c_char_p, [_P]),
    "sbtv_group_shard_of": (_I, [_P, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "sbtv_SALSA_v2_sharded": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, C.POINTER(sbtv_salsa_opts), _P, _P, _P, _P, _P,
                                   _P, _P, _P, _P, _P]),
    "sbtv_SALSA_v2_sharded_dev": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, C.POINTER(sbtv_salsa_opts), _P, _P, _P, _P, _P,
                                       _P, _P, _P, _P, _P]),
    "sbtv_SAPG_algorithm_sharded": (_I, [_P, _P, _I, _I, _I, C.POINTER(sbtv_sapg_opts), _P, _P, _P, _P, _P, _P, _P, _P,
                                         _P, _P, _P]),
    "sbtv_fista_tv_sharded": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _D, _I, _I, _D, _I, _I, _P, _P, _P, _P, _P]),
    "sbtv_CSALSA_v2_sharded": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _D, C.POINTER(sbtv_salsa_opts), _P, _P,
                                    _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sbtv_CoRAL_v2_sharded": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, C.POINTER(sbtv_salsa_opts), _P,
                                   _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sbtv_diag_stage_stats": (_I, [_P, C.POINTER(_D)]),
    "sbtv_host_transpose": (_I, [_P, _P, _I, _I, _I]),
    "sbtv_diag_solve_stats": (_I, [_P, C.POINTER(_D)]),
    "sbtv_diag_canary": (_I, [_P, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "sbtv_diag_prox_variant": (_I, [_P, _I, _I, _I, C.POINTER(_I)]),
    "sbtv_last_host_stats": (_I, [_P, C.POINTER(_D)]),
    "sbtv_diag_workspace": (_I, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "sbtv_diag_switches": (_I, [C.c_char_p, C.c_size_t]),
    "sbtv_diag_time_pass": (_I, [_P, _I, _I, _I, _I, _I, C.POINTER(_D), C.POINTER(_D)]),
}

_lib = None
_lib_lock = threading.Lock()


def load_library():
    """dlopen libsbtv.so (built by `make -C csrc` / __graft_entry__.build())."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; two HIP runtimes in one process do not
        # share the device.  Import torch FIRST (when present) so that libsbtv's DT_NEEDED
        # libamdhip64.so.7 resolves to the copy torch already loaded and both use one runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                "sbtv has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


class Context:
    """One GPU context (sbtv_ctx)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = _P()
        rc = self.lib.sbtv_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise SbtvError(rc, self.lib.sbtv_last_error(None).decode())
        self.h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None):
            self.lib.sbtv_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, flags=0):
        """Raise on a non-zero status.  With device pointers the C call is asynchronous on the
        context stream: wait for it so that torch (which runs on its own stream) may read the result."""
        if rc != 0:
            raise SbtvError(rc, self.lib.sbtv_last_error(self.h).decode())
        if flags & SBTV_DEVICE_PTRS:
            self.sync()

    def sync(self):
        self.check(self.lib.sbtv_ctx_sync(self.h))

    def set_lanes(self, mode):
        """0 (default): a batch of independent items is dealt to two internal streams; 1: one stream; 2: shared-gradient
        chains are split as well (sbtv_ctx_set_lanes)."""
        self.check(self.lib.sbtv_ctx_set_lanes(self.h, int(mode)))

    def solve_stats(self):
        """Cumulative: exact_restarts (solves repeated with exact Chambolle launches), esub_off (switches from subset error
        sums back to full sums) of this context and its lanes (sbtv_diag_solve_stats)."""
        out = (C.c_double * 4)()
        self.check(self.lib.sbtv_diag_solve_stats(self.h, out))
        return dict(exact_restarts=int(out[0]), esub_off=int(out[1]))

    def stage_stats(self):
        """Cumulative staging of large host arrays by this context and its lanes (sbtv_diag_stage_stats)."""
        out = (C.c_double * 4)()
        self.check(self.lib.sbtv_diag_stage_stats(self.h, out))
        return dict(bytes_in=out[0], s_in=out[1], bytes_out=out[2], s_out=out[3])

    def set_stream(self, stream_ptr):
        self.check(self.lib.sbtv_ctx_set_stream(self.h, _P(stream_ptr)))

    @property
    def calls(self):
        v = C.c_longlong(0)
        self.check(self.lib.sbtv_callcounter_get(self.h, C.byref(v)))
        return v.value

    def reset_calls(self):
        self.check(self.lib.sbtv_callcounter_reset(self.h))

    def canary(self, poke=False):
        """Guard-band check of every device workspace (active when SBTV_CANARY=1 was set before the context was
        created): dict(enabled, buffers, bad_bytes).  poke=True damages one guard first (self-test) and repairs it."""
        en, nb, bad = _I(0), _I(0), _I(0)
        self.check(self.lib.sbtv_diag_canary(self.h, int(bool(poke)), C.byref(en), C.byref(nb), C.byref(bad)))
        return dict(enabled=bool(en.value), buffers=nb.value, bad_bytes=bad.value)

    PASSES = {"cols_fwd": 0, "rows_salsa": 1, "cols_inv_post": 2, "cols_inv": 3, "rows_fwd": 4, "rows_grad": 5,
              "rows_gradf": 6, "prox10_warm": 7, "prox25_cold": 8,
              "fused_steps1": 11, "fused_steps2": 12, "fused_steps3": 13, "fused_steps4": 14, "fused_steps5": 15}

    def time_pass(self, name, M, N, batch=1, reps=20):
        """Average ms per launch and algorithmic bytes of one pass of the hot path (sbtv_diag_time_pass)."""
        ms, by = _D(0.0), _D(0.0)
        self.check(self.lib.sbtv_diag_time_pass(self.h, self.PASSES[name], int(M), int(N), int(batch), int(reps),
                                                C.byref(ms), C.byref(by)))
        return dict(ms=ms.value, bytes=by.value, gbs=by.value / (ms.value * 1e-3) / 1e9 if ms.value > 0 else 0.0)

    def prox_variant(self, M, N, batch=1):
        """Which TV-prox kernel an M x N x batch problem takes (sbtv_diag_prox_variant)."""
        out = (_I * 6)()
        self.check(self.lib.sbtv_diag_prox_variant(self.h, int(M), int(N), int(batch), out))
        return dict(cols_per_wave=out[0], waves=out[1], waves_per_simd=out[2], rows_per_lane=out[3], tiles=out[4],
                    fused=bool(out[5]), kind={0: "single-step", 1: "tile", 2: "pipeline"}[out[5]])

    def last_timing(self):
        out = (C.c_double * 4)()
        self.check(self.lib.sbtv_last_timing(self.h, out))
        return dict(loop_ms=out[0], chambolle_ms=out[1], chambolle_launches=out[2], chambolle_bytes=out[3])


    HOST_STATS = ("waits", "ready_at_once", "waits_slept", "sleeps", "stream_queries", "wait_s", "wait_max_s",
                  "enqueue_s", "enqueue_max_s", "wait_max_outer", "nvcsw", "nivcsw", "minflt", "majflt")

    def last_host_stats(self):
        """How the host side of the last SALSA_v2 call waited for the device (sbtv_last_host_stats)."""
        out = (C.c_double * 14)()
        self.check(self.lib.sbtv_last_host_stats(self.h, out))
        return dict(zip(self.HOST_STATS, out))

    def workspace(self, name, M, N, batch=1):
        """Host copy of an internal workspace that holds `batch` column-major M x N images (sbtv_diag_workspace),
        e.g. "salsa.u" / "salsa.bu" after a SALSA_v2 call -> (batch, M, N) array (or (M, N))."""
        p, nb = _P(), C.c_size_t(0)
        self.check(self.lib.sbtv_diag_workspace(self.h, name.encode(), C.byref(p), C.byref(nb)))
        need = 8 * M * N * batch
        if nb.value < need:
            raise ValueError(f"workspace {name} holds {nb.value} bytes, {need} wanted")
        buf = np.empty((batch, N, M), dtype=np.float64)
        self.sync()
        self.check(self.lib.sbtv_memcpy_d2h(self.h, vptr(buf), p, need))
        a = np.transpose(buf, (0, 2, 1))
        return a[0] if batch == 1 else a


class Group:
    """Several GPUs behind this one process (sbtv_group): pass it as `ctx=` to SALSA_v2 or the SAPG_algorithm_*
    functions with HOST (NumPy) images; items are dealt to the devices in contiguous blocks.  An ordinal may repeat
    ("virtual shards" on one GPU)."""
    is_group = True

    def __init__(self, devices):
        self.lib = load_library()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = _P()
        rc = self.lib.sbtv_group_create(devs, len(devices), C.byref(h))
        if rc != 0:
            raise SbtvError(rc, self.lib.sbtv_last_error(None).decode())
        self.h = h
        self.devices = [int(d) for d in devices]

    def __len__(self):
        return self.lib.sbtv_group_size(self.h)

    def shard_of(self, n_items, item):
        s, f, c = _I(0), _I(0), _I(0)
        self.check(self.lib.sbtv_group_shard_of(self.h, int(n_items), int(item), C.byref(s), C.byref(f), C.byref(c)))
        return dict(shard=s.value, first=f.value, count=c.value)

    def check(self, rc, flags=0):
        if rc != 0:
            raise SbtvError(rc, self.lib.sbtv_group_last_error(self.h).decode())

    def blocks(self, n_items):
        """[(first, count)] of the shards that take part in a call with n_items items."""
        out, i = [], 0
        while i < n_items:
            s = self.shard_of(n_items, i)
            out.append((s["first"], s["count"]))
            i = s["first"] + s["count"]
        return out

    def SALSA_v2_device(self, y_shards, taps, tau, mu, maxiter, TViters=10, tolA=1e-3, stopcriterion=1, true_shards=None):
        """sbtv_SALSA_v2_sharded_dev: `y_shards[r]` is a float64 torch tensor ON shard r's device holding that shard's block
        of images (count_r, M, N), column-major image memory (sbtv.to_device); n_items = sum of the counts, dealt as
        `blocks(n_items)` says.  taps: (taille, taille) PSF for all images; tau, mu: scalars or per-image sequences.
        Returns (x_shards, objective[n_items][...], n_outer): nothing crosses the host."""
        import torch
        ys = [Images(t) for t in y_shards]
        n_items = sum(i.B for i in ys)
        blk = self.blocks(n_items)
        if [i.B for i in ys] != [c for _, c in blk]:
            raise ValueError(f"shard r must hold the images of its block: counts {[c for _, c in blk]}")
        M, N = ys[0].M, ys[0].N
        ts = [Images(t) for t in true_shards] if true_shards is not None else None
        xs = [empty_like_images(i) for i in ys]
        for t in y_shards:
            torch.cuda.synchronize(t.device)
        so = sbtv_salsa_opts()
        self.lib.sbtv_salsa_opts_default(C.byref(so))
        so.stopcriterion, so.maxiter, so.TViters, so.tolA = int(stopcriterion), int(maxiter), int(TViters), float(tolA)
        so.compute_mse = 1 if ts is not None else 0
        K = so.maxiter
        objective = np.zeros((n_items, K + 1)); distance = np.zeros((n_items, K)); times = np.zeros((n_items, K + 1))
        mses = np.zeros((n_items, K + 1))
        nA, nAt, nout = (C.c_int * n_items)(), (C.c_int * n_items)(), (C.c_int * n_items)()
        tp = np.ascontiguousarray(np.broadcast_to(np.asarray(taps, dtype=np.float64).T, (n_items,) + np.shape(taps)[::-1]))
        tau_a, tau_p = dvec(tau, n_items)
        mu_a, mu_p = dvec(mu, n_items)
        arr = lambda imgs: (C.c_void_p * len(imgs))(*[i.ptr.value for i in imgs])
        ya, xa = arr(ys), arr(xs)
        ta = arr(ts) if ts is not None else None
        self.check(self.lib.sbtv_SALSA_v2_sharded_dev(self.h, ya, M, N, n_items, vptr(tp), int(np.shape(taps)[0]), tau_p, mu_p,
                                                      C.byref(so), ta, None, xa, vptr(objective), vptr(distance), vptr(times),
                                                      vptr(mses) if ts is not None else None, nA, nAt, nout))
        n = np.array(nout[:])
        return [x.t for x in xs], [objective[b, :n[b] + 1].copy() for b in range(n_items)], n

    def close(self):
        if getattr(self, "h", None):
            self.lib.sbtv_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def switches():
    """The SBTV_* environment switches set in this process ("" = default kernels); sbtv_diag_switches."""
    buf = C.create_string_buffer(2048)
    load_library().sbtv_diag_switches(buf, 2048)
    return buf.value.decode()


_default_ctx = {}


def default_context(device=None) -> Context:
    if device is None:
        device = int(os.environ.get("SBTV_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


# ---------------------------------------------------------------------------
# image marshalling: MATLAB column-major doubles
# ---------------------------------------------------------------------------
def _is_torch(x):
    return type(x).__module__.startswith("torch")


def column_major_images(a):
    """(B, M, N) float64 array in any layout -> C-contiguous (B, N, M) array = column-major images.  Arrays that already are
    column-major per image (Fortran-ordered images, e.g. what `images_result` returns) pass through without a copy; C-ordered
    ones go through the library's blocked, threaded transpose (numpy's strided copy is ~40 x slower)."""
    t = np.transpose(a, (0, 2, 1))
    if t.flags.c_contiguous:
        return t
    if a.flags.c_contiguous and a.size >= 4096:
        out = np.empty((a.shape[0], a.shape[2], a.shape[1]), dtype=np.float64)
        rc = load_library().sbtv_host_transpose(vptr(a), vptr(out), a.shape[0], a.shape[1], a.shape[2])
        if rc == 0:
            return out
    return np.ascontiguousarray(t)


class Images:
    """A batch of M x N images ready for the C-ABI.

    numpy input: (M,N) or (B,M,N) arrays (any order/dtype) -> host pointer to a
    (B, N, M) C-contiguous float64 copy (== column-major per image).
    torch CUDA input: tensor whose memory already is column-major per image,
    i.e. shape (M,N) with strides (1,M) or (B,M,N) with strides (M*N,1,M)
    (use `to_device`) -> device pointer, no copy.
    """

    def __init__(self, x, like=None, _fresh=False):
        self.torch = _is_torch(x)
        if self.torch:
            if x.dtype is not __import__("torch").float64:
                raise TypeError("device images must be float64")
            if x.dim() == 2:
                x = x.unsqueeze(0)
            B, M, N = x.shape
            st = x.stride()
            if st[1:] != (1, M) or (B > 1 and st[0] != M * N):
                raise ValueError("device images must be column-major per image: use sbtv.to_device()")
            if not x.is_cuda:
                raise ValueError("torch images must live on the GPU (numpy arrays are the host path)")
            if not _fresh:          # (a buffer this module has just allocated has no producer)
                __import__("torch").cuda.current_stream(x.device).synchronize()   # producer kernels done
            self.t = x
            self.B, self.M, self.N = B, M, N
            self.ptr = _P(x.data_ptr())
            self.flags = SBTV_DEVICE_PTRS
        else:
            a = np.asarray(x, dtype=np.float64)
            self.squeeze = (a.ndim == 2)
            if a.ndim == 2:
                a = a[None]
            if a.ndim != 3:
                raise ValueError("images must be (M,N) or (B,M,N)")
            self.B, self.M, self.N = a.shape
            self.buf = column_major_images(a)                             # (B,N,M): column-major images
            self.ptr = vptr(self.buf)
            self.flags = SBTV_HOST_PTRS


def empty_like_images(ref: Images):
    """Output buffer matching `ref` (host numpy or device torch)."""
    if ref.torch:
        import torch
        t = torch.empty((ref.B, ref.N, ref.M), dtype=torch.float64, device=ref.t.device).permute(0, 2, 1)
        return Images(t, _fresh=True)
    out = Images.__new__(Images)
    out.torch = False
    out.squeeze = getattr(ref, "squeeze", False)
    out.B, out.M, out.N = ref.B, ref.M, ref.N
    out.buf = np.empty((ref.B, ref.N, ref.M), dtype=np.float64)
    out.ptr = vptr(out.buf)
    out.flags = SBTV_HOST_PTRS
    return out


def images_result(img: Images, squeeze=None):
    """Back to the caller's convention: numpy (M,N)/(B,M,N) or the torch tensor."""
    if img.torch:
        return img.t[0] if squeeze else img.t
    # a VIEW of the column-major buffer the library wrote (shape (B, M, N), Fortran-ordered images): a row-major copy of
    # four 2048^2 images costs more than their whole solve (tools/bench_hostcall.py)
    a = np.transpose(img.buf, (0, 2, 1))
    if squeeze is None:
        squeeze = getattr(img, "squeeze", False)
    return a[0] if squeeze else a


def to_device(x, device="cuda:0"):
    """numpy (M,N)/(B,M,N) -> torch float64 CUDA tensor with column-major image memory."""
    import torch
    a = np.asarray(x, dtype=np.float64)
    sq = a.ndim == 2
    if sq:
        a = a[None]
    t = torch.from_numpy(np.ascontiguousarray(column_major_images(a))).to(device).permute(0, 2, 1)
    return t[0] if sq else t


def to_host(t):
    return t.detach().cpu().numpy().copy()


def dvec(v, n):
    """host double array of length n from a scalar or a sequence."""
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64), (n,)))
    return a, vptr(a)
