"""ctypes front-end of the CPU oracle (oracle/klhr_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package klhr_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np
from numpy.polynomial.hermite import hermgauss

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libklhr_oracle.so")

MODEL_IDS = {"funnel": 0, "ill-normal": 1, "glmm-poisson": 2, "normal": 3, "ar1": 4, "earnings": 5}

F = dict(theta=0, rng=1, mean=2, cov=3, eigvecs=4, eigvals=5, om_m=6, om_v=7, om_N=8, omd_m=9, omd_v=10, omd_N=11,
         pca_v=12, pca_n=13, draw=14, acc=15, grad_evals=16, rho=17, eta=18, diag=19, cidx=20, prev_theta=21)
DIAG = ["x0", "mode", "hinv1", "nfev1", "nit1", "status1", "init0", "init1", "init2", "init3", "nfev2", "nit2",
        "status2", "fun2", "zp", "r", "logu", "accept", "ls_fallbacks"]


def build(force=False):
    src = os.path.join(HERE, "klhr_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "libklhr_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        dp, u64p, vp = C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_void_p
        L.klhr_oracle_create.restype = vp
        L.klhr_oracle_create.argtypes = [C.c_int, dp, C.c_int, C.c_int, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double,
                                         C.c_long, C.c_long, C.c_long, C.c_double, C.c_double, C.c_double, C.c_int,
                                         C.c_int, C.c_int, C.c_int]
        L.klhr_oracle_destroy.argtypes = [vp]
        L.klhr_oracle_seed.argtypes = [vp, C.c_uint64, dp]
        L.klhr_oracle_seed.restype = C.c_int
        L.klhr_oracle_dim.argtypes = [vp]
        L.klhr_oracle_draw.argtypes = [vp, dp]
        L.klhr_oracle_draw_many.argtypes = [vp, C.c_long, dp]
        L.klhr_oracle_random_direction.argtypes = [vp, dp]
        L.klhr_oracle_fit.argtypes = [vp, dp, dp]
        L.klhr_oracle_metropolis.argtypes = [vp, dp, dp]
        L.klhr_oracle_adapt.argtypes = [vp]
        L.klhr_oracle_kl.argtypes = [vp, dp, dp, dp]
        L.klhr_oracle_kl.restype = C.c_double
        L.klhr_oracle_line.argtypes = [vp, C.c_double, dp, dp]
        L.klhr_oracle_logp_grad.argtypes = [vp, dp, dp]
        L.klhr_oracle_logp_grad.restype = C.c_double
        L.klhr_oracle_primitives.argtypes = [vp, C.c_double, dp, dp]
        L.klhr_oracle_primitives_inv.argtypes = [vp, C.c_double, dp, dp]
        L.klhr_oracle_get.argtypes = [vp, C.c_int, vp]
        L.klhr_oracle_set.argtypes = [vp, C.c_int, vp]
        L.klhr_oracle_set_variant.argtypes = [vp, C.c_int]
        L.klhr_oracle_get_K.argtypes = [vp]
        L.klhr_oracle_ndtr.argtypes = [C.c_double]
        L.klhr_oracle_ndtr.restype = C.c_double
        L.klhr_oracle_ndtri.argtypes = [C.c_double]
        L.klhr_oracle_ndtri.restype = C.c_double
        L.klhr_oracle_seed_state.argtypes = [C.c_uint64, u64p]
        L.klhr_oracle_rng_raw.argtypes = [u64p, C.c_long, u64p]
        L.klhr_oracle_rng_double.argtypes = [u64p, C.c_long, dp]
        L.klhr_oracle_rng_normal.argtypes = [u64p, C.c_long, dp]
        L.klhr_oracle_set_slice.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_long]
        L.klhr_oracle_slice_evals.argtypes = [vp]
        L.klhr_oracle_slice_evals.restype = C.c_long
        L.klhr_oracle_slice_status.argtypes = [vp]
        L.klhr_oracle_uni_slice.argtypes = [vp, dp]
        L.klhr_oracle_rng_exponential.argtypes = [u64p, C.c_long, dp]
        L.klhr_oracle_rng_uniform_range.argtypes = [u64p, C.c_double, C.c_double]
        L.klhr_oracle_rng_uniform_range.restype = C.c_double
        L.klhr_oracle_rng_uniform.argtypes = [u64p]
        L.klhr_oracle_rng_uniform.restype = C.c_double
        L.klhr_oracle_om_stream.argtypes = [C.c_int, C.c_long, dp, dp, dp, dp]
        L.klhr_oracle_pca_stream.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_long, dp, dp, dp]
        L.klhr_oracle_window_closures.argtypes = [C.c_long, C.c_long, C.c_long, C.POINTER(C.c_long), C.c_int]
        L.klhr_oracle_window_closures.restype = C.c_int
        L.klhr_oracle_mh_seed.argtypes = [vp, C.c_uint64]
        L.klhr_oracle_mh_draw.argtypes = [vp, C.c_double, C.c_long, dp]
        L.klhr_oracle_draw_batch.argtypes = [C.POINTER(vp), C.c_int, C.c_long]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def quadrature(N=8):
    """klhr_sinh.py:44-47"""
    x, w = hermgauss(N)
    x *= np.sqrt(2)
    w /= np.sqrt(np.pi)
    return x, w


def model_data(stem, data):
    """Flat fp64 data buffer for a model (same packing as klhr_amd.bsmodel)."""
    if stem in ("funnel", "ill-normal", "normal"):
        return np.array([float(data["D"])])
    if stem == "glmm-poisson":
        return np.concatenate([[float(data["n"])], np.asarray(data["year"], float), np.asarray(data["C"], float)])
    if stem == "ar1":
        return np.array([float(data["N"])])
    if stem == "earnings":
        return np.concatenate([[float(data["N"])], np.asarray(data["earn"], float), np.asarray(data["height"], float)])
    raise KeyError(stem)


class Chain:
    """One KLHRSINH chain (constructor defaults of klhr_sinh.py:15-32, except overrelaxed which defaults to False:
    the over-relaxed variates use this build's own RNG convention, see klhr_oracle.c: overrelaxed_zp)."""

    def __init__(self, stem, data, seed=None, N=8, K=10, J=2, l=4, initscale=0.1, warmup=1000, windowsize=50,
                 windowscale=2, tol=1e-10, grad_clip=1e15, scale_clip=300, scale_dir_cov=False, overrelaxed=False,
                 eigen_method_one=False, max_init_tries=100, theta_given=False, variant="sinh"):
        L = lib()
        self._L = L
        self._data = np.ascontiguousarray(model_data(stem, data), dtype=np.float64)
        self.x, self.w = quadrature(N)
        self.J = J
        self.ncol = J + (1 if eigen_method_one else 0)
        self._h = L.klhr_oracle_create(MODEL_IDS[stem], _dp(self._data), self._data.size, N, _dp(self.x), _dp(self.w),
                                       K, J, float(l), initscale, warmup, windowsize, windowscale, tol, grad_clip,
                                       scale_clip, int(scale_dir_cov), int(overrelaxed), int(eigen_method_one),
                                       max_init_tries)
        if not self._h:
            raise ValueError("oracle: bad configuration")
        self.D = L.klhr_oracle_dim(self._h)
        self.variant = variant
        if variant == "gauss":
            L.klhr_oracle_set_variant(self._h, 1)
        elif variant == "sub":
            L.klhr_oracle_set_variant(self._h, 2)
        self.init_tries = None
        if seed is not None:
            self.init_tries = L.klhr_oracle_seed(self._h, int(seed), None)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.klhr_oracle_destroy(self._h)
            self._h = None

    # ---- state
    def get(self, name):
        f = F[name]
        if name == "rng":
            out = np.zeros(4, dtype=np.uint64)
        elif name in ("eigvecs",):
            out = np.zeros((self.D, self.ncol))
        elif name == "pca_v":
            out = np.zeros((self.D, self.J))
        elif name == "eigvals":
            out = np.zeros(self.ncol)
        elif name == "eta":
            out = np.zeros(4)
        elif name == "diag":
            out = np.zeros(len(DIAG))
        elif name in ("om_N", "omd_N", "pca_n", "draw", "acc", "grad_evals", "cidx"):
            out = np.zeros(1)
        else:
            out = np.zeros(self.D)
        assert self._L.klhr_oracle_get(self._h, f, out.ctypes.data_as(C.c_void_p)) == 0
        if name == "diag":
            return dict(zip(DIAG, out))
        if out.size == 1 and name not in ("theta",) and out.ndim == 1 and name in ("om_N", "omd_N", "pca_n", "draw", "acc", "grad_evals", "cidx"):
            return float(out[0])
        return out

    def set(self, name, value):
        if name == "rng":
            v = np.ascontiguousarray(value, dtype=np.uint64)
        else:
            v = np.ascontiguousarray(np.atleast_1d(value), dtype=np.float64)
        assert self._L.klhr_oracle_set(self._h, F[name], v.ctypes.data_as(C.c_void_p)) == 0

    # ---- algorithm
    def draw(self):
        out = np.zeros(self.D)
        self._L.klhr_oracle_draw(self._h, _dp(out))
        return out

    def draw_many(self, n, record=True):
        out = np.zeros((n, self.D)) if record else None
        self._L.klhr_oracle_draw_many(self._h, n, _dp(out) if record else None)
        return out

    def random_direction(self):
        rho = np.zeros(self.D)
        self._L.klhr_oracle_random_direction(self._h, _dp(rho))
        return rho

    def uni_slice(self, rho):
        """slice.py:86-141 from the current theta along rho (increments the draw counter)."""
        rho = np.ascontiguousarray(rho, dtype=np.float64)
        self._L.klhr_oracle_uni_slice(self._h, _dp(rho))
        return self.get("theta")

    def slice_evals(self):
        return int(self._L.klhr_oracle_slice_evals(self._h))

    def fit(self, rho):
        rho = np.ascontiguousarray(rho, dtype=np.float64)
        eta = np.zeros(4)
        self._L.klhr_oracle_fit(self._h, _dp(rho), _dp(eta))
        return eta

    def metropolis(self, eta, rho):
        eta = np.ascontiguousarray(eta, dtype=np.float64)
        rho = np.ascontiguousarray(rho, dtype=np.float64)
        self._L.klhr_oracle_metropolis(self._h, _dp(eta), _dp(rho))

    def adapt(self):
        self._L.klhr_oracle_adapt(self._h)

    def KL(self, eta, rho):
        eta = np.ascontiguousarray(eta, dtype=np.float64)
        rho = np.ascontiguousarray(rho, dtype=np.float64)
        g = np.zeros(4)
        v = self._L.klhr_oracle_kl(self._h, _dp(eta), _dp(rho), _dp(g))
        return v, g

    def logp_grad_rho(self, t, rho):
        rho = np.ascontiguousarray(rho, dtype=np.float64)
        out = np.zeros(2)
        self._L.klhr_oracle_line(self._h, float(t), _dp(rho), _dp(out))
        return out[0], out[1]

    def log_density_gradient(self, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        g = np.zeros(self.D)
        lp = self._L.klhr_oracle_logp_grad(self._h, _dp(theta), _dp(g))
        return lp, g

    def primitives(self, x, eta):
        eta = np.ascontiguousarray(eta, dtype=np.float64)
        out = np.zeros(10)
        self._L.klhr_oracle_primitives(self._h, float(x), _dp(eta), _dp(out))
        return dict(T=out[0], log_abs_jac=out[1], grad_T=out[2:6].copy(), grad_log_abs_jac=out[6:10].copy())

    def primitives_inv(self, y, eta):
        eta = np.ascontiguousarray(eta, dtype=np.float64)
        out = np.zeros(2)
        self._L.klhr_oracle_primitives_inv(self._h, float(y), _dp(eta), _dp(out))
        return dict(T_inv=out[0], log_q=out[1])

    # ---- MH comparator (mh.py)
    def mh_seed(self, seed):
        self._L.klhr_oracle_mh_seed(self._h, int(seed))

    def mh_draw(self, stepsize, n):
        out = np.zeros((n, self.D))
        self._L.klhr_oracle_mh_draw(self._h, float(stepsize), n, _dp(out))
        return out


def KLHRChain(stem, data, seed=None, **kw):
    """A chain of the Gaussian-proposal sampler with klhr.py's constructor defaults (klhr.py:16-34)."""
    d = dict(tol=1e-12, scale_clip=600, overrelaxed=False, eigen_method_one=True, variant="gauss")
    d.update(kw)
    D = {"funnel": lambda: data["D"] + 1, "glmm-poisson": lambda: data["n"] + 5, "ar1": lambda: data["N"],
         "earnings": lambda: 4}.get(stem, lambda: data["D"])()
    d["J"] = min(d.get("J", 2), D - 1)          # klhr.py:39
    return Chain(stem, data, seed=seed, **d)


def SliceChain(stem, data, seed=None, w=1.0, lower=-np.inf, upper=np.inf, maxiter=100000, **kw):
    """A chain of the slice sampler with slice.py's constructor defaults (slice.py:14-32); m = inf only."""
    d = dict(tol=1e-12, eigen_method_one=True, overrelaxed=False)
    d.update(kw)
    c = Chain(stem, data, seed=None, **d)
    c._L.klhr_oracle_set_slice(c._h, float(w), float(lower), float(upper), int(maxiter))
    c.variant = "slice"
    if seed is not None:
        c.init_tries = c._L.klhr_oracle_seed(c._h, int(seed), None)
    return c


def draw_batch(chains, niters):
    """Advance many chains niters draws each, OpenMP threads over chains (OMP_NUM_THREADS)."""
    arr = (C.c_void_p * len(chains))(*[c._h for c in chains])
    lib().klhr_oracle_draw_batch(arr, len(chains), niters)


# ---- RNG helpers
def seed_state(seed):
    out = np.zeros(4, dtype=np.uint64)
    lib().klhr_oracle_seed_state(int(seed), _u64p(out))
    return out


def rng_raw(state4, n):
    st = np.array(state4, dtype=np.uint64)
    out = np.zeros(n, dtype=np.uint64)
    lib().klhr_oracle_rng_raw(_u64p(st), n, _u64p(out))
    return out, st


def rng_exponential(state4, n):
    st = np.array(state4, dtype=np.uint64)
    out = np.zeros(n)
    lib().klhr_oracle_rng_exponential(_u64p(st), n, _dp(out))
    return out, st


def rng_uniform_range(state4, low, high):
    st = np.array(state4, dtype=np.uint64)
    v = lib().klhr_oracle_rng_uniform_range(_u64p(st), float(low), float(high))
    return v, st


def rng_double(state4, n):
    st = np.array(state4, dtype=np.uint64)
    out = np.zeros(n)
    lib().klhr_oracle_rng_double(_u64p(st), n, _dp(out))
    return out, st


def rng_normal(state4, n):
    st = np.array(state4, dtype=np.uint64)
    out = np.zeros(n)
    lib().klhr_oracle_rng_normal(_u64p(st), n, _dp(out))
    return out, st


def rng_uniform(state4):
    st = np.array(state4, dtype=np.uint64)
    u = lib().klhr_oracle_rng_uniform(_u64p(st))
    return u, st


def om_stream(X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, D = X.shape
    m, v, var = np.zeros((n, D)), np.zeros((n, D)), np.zeros((n, D))
    lib().klhr_oracle_om_stream(D, n, _dp(X), _dp(m), _dp(v), _dp(var))
    return m, v, var


def pca_stream(X, K=2, l=4, tol=1e-10):
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, D = X.shape
    v, vals = np.zeros((n, D, K)), np.zeros((n, K))
    lib().klhr_oracle_pca_stream(D, K, float(l), tol, n, _dp(X), _dp(v), _dp(vals))
    return v, vals


def window_closures(warmup, windowsize=25, windowscale=2):
    buf = (C.c_long * 64)()
    n = lib().klhr_oracle_window_closures(warmup, windowsize, windowscale, buf, 64)
    return [buf[i] for i in range(min(n, 64))]
