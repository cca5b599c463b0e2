"""ctypes binding of libklhr_hip.so (C ABI: include/klhr.h).  No CPU fallback: importing the package is
always possible, but any compute call raises KlhrError when the HIP library or a gfx950 device is missing."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KLHR_LIB") or os.path.join(HERE, "lib", "libklhr_hip.so")   # KLHR_LIB: diagnostic builds

NDIAG = 20
DIAG_COLS = ["x0", "mode", "hinv1", "nfev1", "nit1", "status1", "init0", "init1", "init2", "init3", "nfev2", "nit2",
             "status2", "fun2", "zp", "r", "logu", "accept", "fallbacks", "passes"]
FIELDS = dict(theta=0, rng=1, mean=2, cov=3, eigvecs=4, eigvals=5, om_m=6, om_v=7, omd_m=8, omd_v=9, pca_v=10, counters=11,
              acc=12, grad_evals=13, rho=14, eta=15, diag=16, status=17, kstate=18)
MODEL_IDS = {"funnel": 0, "ill-normal": 1, "glmm-poisson": 2, "normal": 3, "ar1": 4, "earnings": 5}


class KlhrError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("model_id", C.c_int32), ("chains", C.c_int32),
                ("N", C.c_int32), ("K", C.c_int32), ("J", C.c_int32), ("max_init_tries", C.c_int32),
                ("l", C.c_double), ("initscale", C.c_double), ("tol", C.c_double), ("grad_clip", C.c_double),
                ("scale_clip", C.c_double),
                ("warmup", C.c_int64), ("windowsize", C.c_int64), ("windowscale", C.c_int64),
                ("scale_dir_cov", C.c_int32), ("overrelaxed", C.c_int32), ("eigen_method_one", C.c_int32),
                ("variant", C.c_int32),
                ("quad_x", C.POINTER(C.c_double)), ("quad_w", C.POINTER(C.c_double)),
                ("quad_asinh_x", C.POINTER(C.c_double))]


_lib = None


def lib():
    """Load the HIP library (built by __graft_entry__.build() / klhr_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KlhrError(f"{LIB_PATH} is missing: build it with `python -m klhr_amd.build` (hipcc, gfx950). "
                        "There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, dp, u64p, i64p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
    sig = {
        "klhr_create": (C.c_int, [C.POINTER(Config), dp, C.c_size_t, C.POINTER(vp)]),
        "klhr_destroy": (None, [vp]),
        "klhr_last_error": (C.c_char_p, [vp]),
        "klhr_abi_version": (C.c_int, []),
        "klhr_build_id": (C.c_char_p, []),
        "klhr_device_count": (C.c_int, []),
        "klhr_dim": (C.c_int, [vp]),
        "klhr_chains": (C.c_int, [vp]),
        "klhr_sync": (C.c_int, [vp]),
        "klhr_kernel_variant": (C.c_int, [vp]),
        "klhr_tile_width": (C.c_int, [vp]),
        "klhr_seed": (C.c_int, [vp, u64p, C.c_int]),
        "klhr_seed_state": (None, [C.c_uint64, u64p]),
        "klhr_window_closures": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, i64p, C.c_int]),
        "klhr_step": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int]),
        "klhr_get_draws": (C.c_int, [vp, dp, C.c_int64, C.c_int64]),
        "klhr_reserve_draws": (C.c_int, [vp, C.c_int64]),
        "klhr_random_direction": (C.c_int, [vp, dp]),
        "klhr_fit": (C.c_int, [vp, dp, dp]),
        "klhr_metropolis": (C.c_int, [vp, dp, dp]),
        "klhr_adapt": (C.c_int, [vp]),
        "klhr_set_slice": (C.c_int, [vp, C.c_double, C.c_double, C.c_double, C.c_int64]),
        "klhr_slice_step": (C.c_int, [vp, dp]),
        "klhr_kl_eval": (C.c_int, [vp, dp, dp, dp, dp]),
        "klhr_line_eval": (C.c_int, [vp, dp, dp, dp]),
        "klhr_log_density_gradient": (C.c_int, [vp, dp, C.c_int64, dp, dp]),
        "klhr_primitives": (C.c_int, [vp, dp, dp, C.c_int64, dp]),
        "klhr_get_state": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
        "klhr_set_state": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
        "klhr_mh_seed": (C.c_int, [vp, u64p]),
        "klhr_mh_step": (C.c_int, [vp, C.c_double, C.c_int64, C.c_int]),
        "klhr_pool_stats": (C.c_int, [vp, dp, C.c_size_t]),
        "klhr_pool_apply": (C.c_int, [vp, dp, dp]),
        "klhr_pool_gram": (C.c_int, [vp, C.c_int64, C.c_int64, dp, C.c_size_t]),
        "klhr_pool_apply_full": (C.c_int, [vp, dp, dp, dp, dp]),
        "klhr_last_step_timing": (C.c_int, [vp, dp, C.POINTER(C.c_int)]),
        "klhr_comm_unique_id": (C.c_int, [vp, C.c_size_t]),
        "klhr_comm_init": (C.c_int, [vp, vp, C.c_size_t, C.c_int, C.c_int]),
        "klhr_comm_world": (C.c_int, [vp]),
        "klhr_pool_reset": (C.c_int, [vp]),
        "klhr_pool_accumulate": (C.c_int, [vp, C.c_int64, C.c_int64]),
        "klhr_pool_allreduce": (C.c_int, [vp]),
        "klhr_pool_fetch": (C.c_int, [vp, dp, C.c_size_t]),
        "klhr_set_kernel_profiling": (C.c_int, [vp, C.c_int]),
        "klhr_last_kernel_times": (C.c_int, [vp, dp, C.POINTER(C.c_int)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTS = ["klhr_create", "klhr_destroy", "klhr_last_error", "klhr_abi_version", "klhr_build_id", "klhr_device_count", "klhr_dim", "klhr_chains", "klhr_sync",
           "klhr_kernel_variant", "klhr_tile_width",
           "klhr_seed", "klhr_seed_state", "klhr_window_closures", "klhr_step", "klhr_get_draws", "klhr_reserve_draws", "klhr_random_direction",
           "klhr_fit", "klhr_metropolis", "klhr_adapt", "klhr_set_slice", "klhr_slice_step", "klhr_kl_eval", "klhr_line_eval", "klhr_log_density_gradient",
           "klhr_primitives", "klhr_get_state", "klhr_set_state", "klhr_mh_seed", "klhr_mh_step", "klhr_pool_stats",
           "klhr_pool_apply", "klhr_pool_gram", "klhr_pool_apply_full", "klhr_last_step_timing", "klhr_set_kernel_profiling",
           "klhr_last_kernel_times", "klhr_comm_unique_id", "klhr_comm_init", "klhr_comm_world", "klhr_pool_reset", "klhr_pool_accumulate",
           "klhr_pool_allreduce", "klhr_pool_fetch"]


def device_count():
    return lib().klhr_device_count()


def build_id():
    """sha256 of the sources the loaded library was built from (klhr_build_id)"""
    return lib().klhr_build_id().decode()


def comm_unique_id():
    """128 bytes identifying a new RCCL communicator (rank 0 creates it, every rank passes it to Sampler.comm_init)."""
    buf = C.create_string_buffer(128)
    rc = lib().klhr_comm_unique_id(buf, 128)
    if rc != 0:
        raise KlhrError(f"klhr_comm_unique_id failed ({rc}): {lib().klhr_last_error(None).decode()}")
    return buf.raw


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def seed_state(seed):
    out = np.zeros(4, dtype=np.uint64)
    lib().klhr_seed_state(int(seed), out.ctypes.data_as(C.POINTER(C.c_uint64)))
    return out


def window_closures(warmup, windowsize=25, windowscale=2):
    """WindowedAdaptation(warmup, windowsize, windowscale)._closures: the count first, then a buffer of that size
    (windowscale = 1 gives warmup / windowsize closures: any number)."""
    L = lib()
    n = L.klhr_window_closures(int(warmup), int(windowsize), int(windowscale), None, 0)
    if n <= 0:
        return []
    buf = (C.c_int64 * n)()
    L.klhr_window_closures(int(warmup), int(windowsize), int(windowscale), buf, n)
    return [int(buf[i]) for i in range(n)]


class Sampler:
    """Thin RAII wrapper over a klhr_sampler handle; all arrays are host NumPy, chain-major."""

    def __init__(self, model_id, model_data, chains, device=0, N=8, K=10, J=2, l=4, initscale=0.1, warmup=1000,
                 windowsize=50, windowscale=2, tol=1e-10, grad_clip=1e15, scale_clip=300, scale_dir_cov=False,
                 overrelaxed=False, eigen_method_one=False, max_init_tries=100, quad=None, variant=0):
        L = lib()
        self._L = L
        self._h = C.c_void_p()
        if quad is None:
            from numpy.polynomial.hermite import hermgauss
            x, w = hermgauss(N)           # klhr_sinh.py:44-47
            x = x * np.sqrt(2)
            w = w / np.sqrt(np.pi)
        else:
            x, w = quad
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.w = np.ascontiguousarray(w, dtype=np.float64)
        self._asx = np.ascontiguousarray(np.arcsinh(self.x))
        self._data = np.ascontiguousarray(model_data, dtype=np.float64)
        cfg = Config()
        cfg.struct_size = C.sizeof(Config)
        cfg.device, cfg.model_id, cfg.chains = int(device), int(model_id), int(chains)
        cfg.N, cfg.K, cfg.J, cfg.max_init_tries = int(N), int(K), int(J), int(max_init_tries)
        cfg.l, cfg.initscale, cfg.tol, cfg.grad_clip, cfg.scale_clip = float(l), float(initscale), float(tol), \
            float(grad_clip), float(scale_clip)
        cfg.warmup, cfg.windowsize, cfg.windowscale = int(warmup), int(windowsize), int(windowscale)
        cfg.scale_dir_cov, cfg.overrelaxed, cfg.eigen_method_one = int(bool(scale_dir_cov)), int(bool(overrelaxed)), \
            int(bool(eigen_method_one))
        cfg.variant = int(variant)
        cfg.quad_x, cfg.quad_w, cfg.quad_asinh_x = dptr(self.x), dptr(self.w), dptr(self._asx)
        rc = L.klhr_create(C.byref(cfg), dptr(self._data), self._data.size, C.byref(self._h))
        if rc != 0:
            raise KlhrError(f"klhr_create failed ({rc}): {L.klhr_last_error(None).decode()}")
        self.C = int(chains)
        self.D = L.klhr_dim(self._h)
        self.J = int(J)
        self.ncol = self.J + (1 if eigen_method_one else 0)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.klhr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise KlhrError(f"{what} failed ({rc}): {self._L.klhr_last_error(self._h).decode()}")

    # ---- seeding / stepping
    def seed(self, seeds, theta_given=False):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert seeds.shape == (self.C,)
        self._ck(self._L.klhr_seed(self._h, seeds.ctypes.data_as(C.POINTER(C.c_uint64)), int(theta_given)), "klhr_seed")

    def step(self, n, record_draws=False, record_diag=False, sync=True):
        self._ck(self._L.klhr_step(self._h, int(n), int(record_draws), int(record_diag)), "klhr_step")
        if sync:
            self.sync()

    def sync(self):
        self._ck(self._L.klhr_sync(self._h), "klhr_sync")

    def reserve_draws(self, n):
        self._ck(self._L.klhr_reserve_draws(self._h, int(n)), "klhr_reserve_draws")

    def kernel_names(self):
        """the hot-path kernels of this handle by name, in the slot order of kernel_times()"""
        stream = "klhr_stream16_kernel" if self._L.klhr_tile_width(self._h) == 16 else "klhr_stream_kernel"
        return {0: ["klhr_iter_kernel"], 1: ["klhr_lane_kernel"], 2: [stream, "(none)", "klhr_fit12_kernel"],
                3: [stream, "(none)", "klhr_fitn_kernel"],
                5: ["klhr_tile_kernel"]}[self._L.klhr_kernel_variant(self._h)]

    def kernel_name(self):
        return "+".join(n for n in self.kernel_names() if n != "(none)")

    def draws(self, first=0, n=None, total=None):
        n = int(n)
        out = np.empty((n, self.C, self.D))
        self._ck(self._L.klhr_get_draws(self._h, dptr(out), int(first), n), "klhr_get_draws")
        return out

    def kernel_profiling(self, on=True):
        self._ck(self._L.klhr_set_kernel_profiling(self._h, int(bool(on))), "klhr_set_kernel_profiling")

    def kernel_times(self):
        """mean ms and launch counts of [stream, fit1, fit2] (split pipeline) or [hot kernel, 0, 0] over the last step()"""
        ms = (C.c_double * 3)()
        n = (C.c_int * 3)()
        self._ck(self._L.klhr_last_kernel_times(self._h, ms, n), "klhr_last_kernel_times")
        return [ms[i] for i in range(3)], [n[i] for i in range(3)]

    def timing(self):
        ms = C.c_double()
        nl = C.c_int()
        self._ck(self._L.klhr_last_step_timing(self._h, C.byref(ms), C.byref(nl)), "klhr_last_step_timing")
        return ms.value, nl.value

    # ---- pieces
    def random_direction(self):
        rho = np.empty((self.C, self.D))
        self._ck(self._L.klhr_random_direction(self._h, dptr(rho)), "klhr_random_direction")
        return rho

    def fit(self, rho):
        rho = np.ascontiguousarray(rho, dtype=np.float64).reshape(self.C, self.D)
        eta = np.empty((self.C, 4))
        self._ck(self._L.klhr_fit(self._h, dptr(rho), dptr(eta)), "klhr_fit")
        return eta

    def metropolis(self, eta, rho):
        rho = np.ascontiguousarray(rho, dtype=np.float64).reshape(self.C, self.D)
        eta = np.ascontiguousarray(eta, dtype=np.float64).reshape(self.C, 4)
        self._ck(self._L.klhr_metropolis(self._h, dptr(eta), dptr(rho)), "klhr_metropolis")

    def adapt(self):
        self._ck(self._L.klhr_adapt(self._h), "klhr_adapt")

    def set_slice(self, w=1.0, lower=-np.inf, upper=np.inf, max_evals=100000):
        self._ck(self._L.klhr_set_slice(self._h, float(w), float(lower), float(upper), int(max_evals)), "klhr_set_slice")

    def slice_step(self, rho):
        rho = np.ascontiguousarray(rho, dtype=np.float64).reshape(self.C, self.D)
        self._ck(self._L.klhr_slice_step(self._h, dptr(rho)), "klhr_slice_step")

    def kl_eval(self, eta, rho):
        rho = np.ascontiguousarray(rho, dtype=np.float64).reshape(self.C, self.D)
        eta = np.ascontiguousarray(eta, dtype=np.float64).reshape(self.C, 4)
        v, g = np.empty(self.C), np.empty((self.C, 4))
        self._ck(self._L.klhr_kl_eval(self._h, dptr(eta), dptr(rho), dptr(v), dptr(g)), "klhr_kl_eval")
        return v, g

    def line_eval(self, t, rho):
        rho = np.ascontiguousarray(rho, dtype=np.float64).reshape(self.C, self.D)
        t = np.ascontiguousarray(t, dtype=np.float64).reshape(self.C)
        out = np.empty((self.C, 2))
        self._ck(self._L.klhr_line_eval(self._h, dptr(t), dptr(rho), dptr(out)), "klhr_line_eval")
        return out

    def log_density_gradient(self, thetas, grad=True):
        thetas = np.ascontiguousarray(thetas, dtype=np.float64).reshape(-1, self.D)
        n = thetas.shape[0]
        lp = np.empty(n)
        g = np.empty((n, self.D)) if grad else None
        self._ck(self._L.klhr_log_density_gradient(self._h, dptr(thetas), n, dptr(lp), dptr(g) if grad else None),
                 "klhr_log_density_gradient")
        return lp, g

    def primitives(self, x, eta):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        eta = np.ascontiguousarray(eta, dtype=np.float64).reshape(-1, 4)
        assert eta.shape[0] == x.size
        out = np.empty((x.size, 12))
        self._ck(self._L.klhr_primitives(self._h, dptr(x), dptr(eta), x.size, dptr(out)), "klhr_primitives")
        return out

    # ---- state
    def _shape(self, name):
        C_, D = self.C, self.D
        return {"theta": ((C_, D), np.float64), "rng": ((C_, 4), np.uint64), "mean": ((C_, D), np.float64),
                "cov": ((C_, D), np.float64), "eigvecs": ((C_, D, self.ncol), np.float64),
                "eigvals": ((C_, self.ncol), np.float64), "om_m": ((C_, D), np.float64), "om_v": ((C_, D), np.float64),
                "omd_m": ((C_, D), np.float64), "omd_v": ((C_, D), np.float64), "pca_v": ((C_, D, self.J), np.float64),
                "counters": ((C_, 4), np.int64), "acc": ((C_,), np.float64), "grad_evals": ((C_,), np.int64),
                "rho": ((C_, D), np.float64), "eta": ((C_, 4), np.float64), "diag": ((C_, NDIAG), np.float64),
                "status": ((C_,), np.int32), "kstate": ((C_, 4), np.float64)}[name]

    def get(self, name):
        shape, dt = self._shape(name)
        out = np.empty(shape, dtype=dt)
        self._ck(self._L.klhr_get_state(self._h, FIELDS[name], out.ctypes.data_as(C.c_void_p), out.nbytes), f"get {name}")
        return out

    def set(self, name, value):
        shape, dt = self._shape(name)
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=dt), shape))
        self._ck(self._L.klhr_set_state(self._h, FIELDS[name], v.ctypes.data_as(C.c_void_p), v.nbytes), f"set {name}")

    # ---- MH comparator
    def mh_seed(self, seeds):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        self._ck(self._L.klhr_mh_seed(self._h, seeds.ctypes.data_as(C.POINTER(C.c_uint64))), "klhr_mh_seed")

    def mh_step(self, stepsize, n, record_draws=False, sync=True):
        self._ck(self._L.klhr_mh_step(self._h, float(stepsize), int(n), int(record_draws)), "klhr_mh_step")
        if sync:
            self.sync()

    # ---- pooled adaptation (extension)
    def pool_stats(self):
        out = np.empty(1 + 2 * self.D)
        self._ck(self._L.klhr_pool_stats(self._h, dptr(out), out.size), "klhr_pool_stats")
        return out

    def pool_apply(self, mean, cov):
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        cov = np.ascontiguousarray(cov, dtype=np.float64)
        self._ck(self._L.klhr_pool_apply(self._h, dptr(mean), dptr(cov)), "klhr_pool_apply")

    def pool_gram(self, first, n):
        """[count, sum theta (D), sum theta theta^T (D, D)] over recorded draws [first, first + n) — fp64 MFMA Gram kernel."""
        out = np.empty(1 + self.D + self.D * self.D)
        self._ck(self._L.klhr_pool_gram(self._h, int(first), int(n), dptr(out), out.size), "klhr_pool_gram")
        return out

    # ---- the collective inside the library (RCCL)
    def comm_init(self, id_bytes, rank, world):
        buf = C.create_string_buffer(bytes(id_bytes), 128)
        self._ck(self._L.klhr_comm_init(self._h, buf, 128, int(rank), int(world)), "klhr_comm_init")

    def comm_world(self):
        return self._L.klhr_comm_world(self._h)

    def pool_reset(self):
        self._ck(self._L.klhr_pool_reset(self._h), "klhr_pool_reset")

    def pool_accumulate(self, first, n):
        self._ck(self._L.klhr_pool_accumulate(self._h, int(first), int(n)), "klhr_pool_accumulate")

    def pool_allreduce(self):
        self._ck(self._L.klhr_pool_allreduce(self._h), "klhr_pool_allreduce")

    def pool_fetch(self):
        out = np.empty(1 + self.D + self.D * self.D)
        self._ck(self._L.klhr_pool_fetch(self._h, dptr(out), out.size), "klhr_pool_fetch")
        return out

    def pool_apply_full(self, mean, cov_diag, eigvecs, eigvals):
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (mean, cov_diag, eigvecs, eigvals)]
        assert a[2].shape == (self.D, self.ncol) and a[3].shape == (self.ncol,)
        self._ck(self._L.klhr_pool_apply_full(self._h, dptr(a[0]), dptr(a[1]), dptr(a[2]), dptr(a[3])), "klhr_pool_apply_full")
