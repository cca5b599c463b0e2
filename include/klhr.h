/*
 * klhr.h — C ABI of libklhr_hip.so: the MI355X (gfx950) many-chain KL Hit-and-Run sampler.
 *
 * The reference (roualdes/klhr) is pure Python; its hot path crosses exactly one FFI boundary,
 * Python -> BridgeStan (ctypes) at every log-density call (bsmodel.py:15-30).  A device function
 * cannot cross ctypes, so this library moves the whole per-iteration path
 *     KLHRSINH.draw()  (klhr_sinh.py:262-289)
 * behind one C ABI and batches it over `chains` independent chains.  Each entry point below names
 * the reference interface it replaces.  Conventions: plain pointers and sizes, caller-owned HOST
 * buffers (row-major, chain-major: [chain][dim]), library-owned device buffers, `int` status
 * (0 = ok, <0 = error; see klhr_last_error), no exceptions, one host thread per handle, a handle is
 * bound to one device.  HIP streams are a per-device pool of four shared by every handle on the device
 * (created once per process: see klhr_api.hip): handles on one device serialise in stream order, and
 * klhr_sync / klhr_last_step_timing of one handle wait for / include work that another handle has put on
 * the same streams — timing is meaningful only while no other handle of the device is active.
 *
 * There is NO CPU fallback: every compute entry point runs HIP kernels and fails with
 * KLHR_ERR_HIP when no gfx950 device is usable.
 */
#ifndef KLHR_H
#define KLHR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KLHR_ABI_VERSION 1

#if defined(__GNUC__)
#define KLHR_API __attribute__((visibility("default")))
#else
#define KLHR_API
#endif

enum klhr_status {
    KLHR_OK = 0,
    KLHR_ERR_ARG = -1,     /* bad argument / unsupported configuration */
    KLHR_ERR_HIP = -2,     /* HIP runtime error (no device, OOM, launch failure) */
    KLHR_ERR_STATE = -3,   /* call sequence error (e.g. step before seed) */
    KLHR_ERR_INIT = -4     /* some chain failed _initialize after max_init_tries (klhr_sinh.py:96-98) */
};

/* Built-in log densities: the device-side replacement of BSModel(stan_file, data_file)
 * (bsmodel.py:6-13).  `model_data` packing (fp64):
 *   FUNNEL        [D_alpha]                       stan/funnel.stan      (dim = D_alpha + 1)
 *   ILL_NORMAL    [D]                             stan/ill-normal.stan
 *   GLMM_POISSON  [n, year_1..year_n, C_1..C_n]   stan/glmm-poisson.stan (dim = n + 5)
 *   NORMAL        [D]                             stan/normal.stan
 *   AR1           [N]                             stan/ar1.stan          (dim = N)
 *   EARNINGS      [N, earn_1..N, height_1..N]     stan/earnings.stan     (dim = 4: beta_1, beta_2, log sigma, log s) */
enum klhr_model { KLHR_MODEL_FUNNEL = 0, KLHR_MODEL_ILL_NORMAL = 1, KLHR_MODEL_GLMM_POISSON = 2, KLHR_MODEL_NORMAL = 3,
                  KLHR_MODEL_AR1 = 4, KLHR_MODEL_EARNINGS = 5 };

/* Mirrors KLHRSINH.__init__ keyword arguments (klhr_sinh.py:15-32) plus `chains` and `device`. */
typedef struct klhr_config {
    uint32_t struct_size;      /* = sizeof(klhr_config) */
    int32_t device;            /* HIP device ordinal */
    int32_t model_id;          /* enum klhr_model */
    int32_t chains;            /* C: number of independent chains (extension; reference = 1) */
    int32_t N;                 /* quadrature nodes; this build supports N == 8 (one lane per node) */
    int32_t K;                 /* over-relaxation K (klhr_sinh.py:216); used when overrelaxed != 0 */
    int32_t J;                 /* online-PCA eigenvectors; J <= 3 */
    int32_t max_init_tries;
    double l;                  /* CCIPCA amnesia */
    double initscale, tol, grad_clip, scale_clip;
    int64_t warmup, windowsize, windowscale;
    int32_t scale_dir_cov, overrelaxed, eigen_method_one;   /* overrelaxed: binomial/beta variates come from the
                                                              * chain's own PCG64 stream (the reference uses SciPy's unseeded
                                                              * global RNG there: distribution-exact, not stream-exact) */
    int32_t variant;           /* 0: KLHRSINH, 4-parameter sinh-arcsinh proposal (klhr_sinh.py)
                                * 1: KLHR, 2-parameter Gaussian proposal (klhr.py:15-223); eta is then (m, log s, 0, 0)
                                * 2: SUBKLHRSINH, 3-parameter sinh-arcsinh, d == 1 (sub_klhr_sinh.py); eta = (m, log s, e, 0)
                                * 3: Slice, Neal's univariate slice sampler along the same direction draw (slice.py:12-183);
                                *    no fit; K, N, grad_clip, scale_clip, overrelaxed are ignored; see klhr_set_slice */
    const double *quad_x;      /* N probabilists' Gauss-Hermite nodes   (klhr_sinh.py:44-47) */
    const double *quad_w;      /* N weights                                                  */
    const double *quad_asinh_x;/* N values asinh(x_n) (host-computed, np.arcsinh)            */
} klhr_config;

typedef struct klhr_sampler klhr_sampler;

/* ---- lifetime ------------------------------------------------------------------------------- */
/* replaces: BSModel(...) + KLHRSINH(bsmodel, **kw) object construction, minus RNG seeding        */
KLHR_API int klhr_create(const klhr_config *cfg, const double *model_data, size_t n_model_data, klhr_sampler **out);
KLHR_API void klhr_destroy(klhr_sampler *s);
KLHR_API const char *klhr_last_error(const klhr_sampler *s);   /* s may be NULL: last create() error */
KLHR_API int klhr_abi_version(void);
/* sha256 of the library's sources (klhr_amd/build.py: source_hash) — names the build a profile was taken on
 * (profiles/traffic.json; bench.py quotes measured HBM traffic only for the build that is loaded) */
KLHR_API const char *klhr_build_id(void);
KLHR_API int klhr_device_count(void);                          /* visible HIP devices (0 when none / no driver) */
KLHR_API int klhr_dim(const klhr_sampler *s);                  /* replaces BSModel.dim() (bsmodel.py:39-40) */
KLHR_API int klhr_chains(const klhr_sampler *s);
KLHR_API int klhr_sync(klhr_sampler *s);                       /* wait for the handle's stream */
/* which hot-path kernels this handle launches: 0 = octet-per-chain (klhr_iter_kernel), 1 = lane-per-chain
 * (klhr_lane_kernel, model dimension <= 26; separable targets only from 16 385 chains on — below that they run as 5, with
 * the same bits), 2 = split-phase pipeline: streaming kernel + klhr_fit12_kernel (both fit
 * stages lane = chain; funnel / ill-normal / normal / ar1 above dimension 26), 3 = split-phase pipeline with both fit
 * stages in the node-parallel klhr_fitn_kernel (up to 32 768 chains), 5 = as 3, and klhr_step runs whole iterations per
 * 16-chain tile in one launch (klhr_tile_kernel: up to 16 384 chains, ill-normal / normal / ar1 up to 32 768; the pieces
 * of the C ABI and diagnostic steps use the kernels of 3).  (4 was the hand-over mode removed in round 3.)  Results agree within the stated tolerances
 * (2, 3, 5 and, for funnel / ill-normal / normal without scale_dir_cov, 1: bit for bit); only the mapping to the machine differs.  KLHR_KERNEL = octet | lane | mono | split in the
 * environment overrides.
 * klhr_tile_width: chains per tile of the split pipeline's state layout — 64 (klhr_stream_kernel) or 16
 * (klhr_stream16_kernel / klhr_tile_kernel, four lanes per chain; the range of variant 5); 0 for the monolithic kernels. */
KLHR_API int klhr_kernel_variant(const klhr_sampler *s);
KLHR_API int klhr_tile_width(const klhr_sampler *s);

/* ---- seeding / initialisation --------------------------------------------------------------- */
/* replaces: MCMCBase.__init__ (mcmc.py:4-17: default_rng(seed), D normals consumed unless a theta
 * was given) followed by KLHRSINH._initialize (klhr_sinh.py:86-98), per chain, on device.
 * seeds[c] is the integer passed to np.random.default_rng; SeedSequence hashing is done here.      */
KLHR_API int klhr_seed(klhr_sampler *s, const uint64_t *seeds, int theta_given);
/* host helper: np.random.default_rng(seed).bit_generator.state as [state_hi,state_lo,inc_hi,inc_lo] */
KLHR_API void klhr_seed_state(uint64_t seed, uint64_t out4[4]);
/* host helper: WindowedAdaptation(warmup, windowsize, windowscale)._closures (windowedadaptation.py:12-28) */
KLHR_API int klhr_window_closures(int64_t warmup, int64_t windowsize, int64_t windowscale, int64_t *out, int cap);

/* ---- the hot path ---------------------------------------------------------------------------- */
/* replaces: `for m in range(n): thetas[m] = algo.draw()` (experiment_funnel.py:49-52) for all chains.
 * record_draws: keep the n×C×D draws in a device buffer (fetch with klhr_get_draws).
 * record_diag : also store per-chain diagnostics of the LAST iteration (rho, eta, counters).       */
KLHR_API int klhr_step(klhr_sampler *s, int64_t n_iters, int record_draws, int record_diag);
KLHR_API int klhr_get_draws(klhr_sampler *s, double *host_out, int64_t first_iter, int64_t n_iters); /* [n][C][D] */
/* allocate the device buffer for n_iters recorded draws ahead of time (the counterpart of `thetas = np.zeros((M, D))`,
 * experiment_funnel.py:48); without it the first recording klhr_step allocates it.                  */
KLHR_API int klhr_reserve_draws(klhr_sampler *s, int64_t n_iters);

/* pieces of draw(), for teacher-forced parity tests and for callers of the reference's methods:
 *   klhr_random_direction  <- KLHRSINH._random_direction()         klhr_sinh.py:203-213
 *   klhr_fit               <- KLHRSINH.fit(rho)                     klhr_sinh.py:182-201
 *   klhr_metropolis        <- KLHRSINH._metropolis_step(eta, rho)   klhr_sinh.py:242-260 (also _draw += 1)
 *   klhr_adapt             <- closure-or-update tail of draw()      klhr_sinh.py:268-287
 * rho is C×D, eta is C×4 (host).  Outputs may be NULL.                                             */
KLHR_API int klhr_random_direction(klhr_sampler *s, double *rho_out);
KLHR_API int klhr_fit(klhr_sampler *s, const double *rho, double *eta_out);
KLHR_API int klhr_metropolis(klhr_sampler *s, const double *eta, const double *rho);
KLHR_API int klhr_adapt(klhr_sampler *s);

/* Slice sampler (variant 3) — replaces Slice.__init__'s w / lower / upper (slice.py:19-22; m = inf only: the reference's
 * finite-m branch refers to undefined names and cannot run) and Slice._uni_slice(rho) (slice.py:86-141, also _draw += 1).
 * max_evals caps the log-density evaluations of one draw (the reference loops forever on a density that never
 * compares); reaching it sets value 4 (bit 2) in the chain's status.  klhr_step / klhr_random_direction / klhr_adapt serve
 * draw() (slice.py:160-183) as for the other variants; grad_evals counts log-density evaluations.               */
KLHR_API int klhr_set_slice(klhr_sampler *s, double w, double lower, double upper, int64_t max_evals);
KLHR_API int klhr_slice_step(klhr_sampler *s, const double *rho);

/* replaces: KLHRSINH.KL(eta, rho) (klhr_sinh.py:163-176) at each chain's current theta             */
KLHR_API int klhr_kl_eval(klhr_sampler *s, const double *eta, const double *rho, double *value_out, double *grad_out);
/* replaces: KLHRSINH.logp_grad_rho(t, rho) (klhr_sinh.py:178-180); out is C×2 (-logp, -grad.rho)   */
KLHR_API int klhr_line_eval(klhr_sampler *s, const double *t, const double *rho, double *out2);
/* replaces: BSModel.log_density_gradient(theta) (bsmodel.py:24-30) for n points; grad may be NULL
 * (then it is BSModel.log_density, bsmodel.py:15-22)                                              */
KLHR_API int klhr_log_density_gradient(klhr_sampler *s, const double *thetas, int64_t n, double *lp_out, double *grad_out);
/* replaces: KLHRSINH._T/_T_inv/_log_abs_jac/_log_q/_grad_T/_grad_log_abs_jac (klhr_sinh.py:112-156,230-240)
 * for n (x, eta) pairs. out is n×12: [T, T_inv, log_abs_jac, log_q, grad_T[4], grad_log_abs_jac[4]]   */
KLHR_API int klhr_primitives(klhr_sampler *s, const double *x, const double *eta, int64_t n, double *out12);

/* ---- state access (teacher forcing, checkpoint/restore) ------------------------------------- */
enum klhr_field {
    KLHR_F_THETA = 0,      /* double  C×D                                   */
    KLHR_F_RNG = 1,        /* uint64  C×4  [state_hi,state_lo,inc_hi,inc_lo] */
    KLHR_F_MEAN = 2,       /* double  C×D   _mean                            */
    KLHR_F_COV = 3,        /* double  C×D   _cov (set also recomputes the SVD rank permutation) */
    KLHR_F_EIGVECS = 4,    /* double  C×D×ncol  (ncol = J, or J+1 with eigen_method_one) */
    KLHR_F_EIGVALS = 5,    /* double  C×ncol */
    KLHR_F_OM_M = 6, KLHR_F_OM_V = 7,          /* OnlineMoments(theta) m, v      C×D */
    KLHR_F_OMD_M = 8, KLHR_F_OMD_V = 9,        /* OnlineMoments(clipped grad)    C×D (scale_dir_cov only) */
    KLHR_F_PCA_V = 10,     /* double  C×D×J   OnlinePCA.v */
    KLHR_F_COUNTERS = 11,  /* int64   C×4  [_draw, OnlineMoments.N, OnlinePCA.n, window index] */
    KLHR_F_ACC = 12,       /* double  C    acceptance_probability */
    KLHR_F_GRAD_EVALS = 13,/* int64   C    grad_evals */
    KLHR_F_RHO = 14,       /* double  C×D  last direction (record_diag) */
    KLHR_F_ETA = 15,       /* double  C×4  last fitted eta (record_diag) */
    KLHR_F_DIAG = 16,      /* double  C×KLHR_NDIAG (record_diag), layout below */
    KLHR_F_STATUS = 17,    /* int32   C    bit0 (1) init failed, bit1 (2) theta non-finite, bit2 (4) slice evaluation cap, bit3 (8) fit safety valve */
    KLHR_F_KSTATE = 18     /* double  C×4  KLHR variant: [Smoother x, Smoother count, current K, moved flag] (klhr.py:212-221) */
};
/* KLHR_F_DIAG columns */
enum { KLHR_D_X0 = 0, KLHR_D_MODE, KLHR_D_HINV1, KLHR_D_NFEV1, KLHR_D_NIT1, KLHR_D_STATUS1, KLHR_D_INIT0, KLHR_D_INIT1,
       KLHR_D_INIT2, KLHR_D_INIT3, KLHR_D_NFEV2, KLHR_D_NIT2, KLHR_D_STATUS2, KLHR_D_FUN2, KLHR_D_ZP, KLHR_D_R,
       KLHR_D_LOGU, KLHR_D_ACCEPT, KLHR_D_FALLBACKS, KLHR_D_PASSES, KLHR_NDIAG };
KLHR_API int klhr_get_state(klhr_sampler *s, int field, void *host_out, size_t bytes);
KLHR_API int klhr_set_state(klhr_sampler *s, int field, const void *host_in, size_t bytes);

/* ---- MH comparator (mh.py:7-37) on the same models ------------------------------------------ */
KLHR_API int klhr_mh_seed(klhr_sampler *s, const uint64_t *seeds);                 /* MCMCBase.__init__ */
KLHR_API int klhr_mh_step(klhr_sampler *s, double stepsize, int64_t n_iters, int record_draws);  /* MH.draw() × n */

/* ---- multi-GPU pooled adaptation (extension, default off) ------------------------ */
/* Per-GPU sufficient statistics of the current window over this handle's chains:
 * out = [n_chains_x_N, sum_i theta (D), sum_i theta^2 (D)] — to be all-reduced by the caller.      */
KLHR_API int klhr_pool_stats(klhr_sampler *s, double *out, size_t n_out);
/* Overwrite every chain's direction mean/cov with pooled values (D each) at a closure.            */
KLHR_API int klhr_pool_apply(klhr_sampler *s, const double *mean, const double *cov);
/* Full-covariance variant: sufficient statistics of the draws recorded by the last recording call,
 * iterations [first_iter, first_iter + n_iters): out = [count, sum theta (D), sum theta theta^T (D x D, row major)],
 * computed as one fp64 MFMA Gram product over the (n_iters*C) x D draws matrix (v_mfma_f64_16x16x4_f64).           */
KLHR_API int klhr_pool_gram(klhr_sampler *s, int64_t first_iter, int64_t n_iters, double *out, size_t n_out);
/* ... and the matching setter: pooled mean (D), diagonal of the pooled covariance (D), unit eigenvectors (D x ncol,
 * row major, ncol = J or J + 1 with eigen_method_one) and eigenvalues (ncol) for every chain.                         */
KLHR_API int klhr_pool_apply_full(klhr_sampler *s, const double *mean, const double *cov_diag, const double *eigvecs, const double *eigvals);

/* ---- the collective of the pooled mode, inside the library (RCCL over xGMI; librccl.so is loaded on first use) ----------
 * One process per GPU.  Rank 0 calls klhr_comm_unique_id and hands the bytes to the other ranks by any host channel
 * (bench.py: the torch.distributed gloo group it already has for its barrier; a file works too); every rank then calls
 * klhr_comm_init on its handle.  Per adaptation window:  klhr_pool_reset  ->  klhr_pool_accumulate for each recorded
 * segment (LDS-tiled v_mfma_f64_16x16x4_f64 Gram kernel, X read once, statistics stay on the device)  ->
 * klhr_pool_allreduce (packs [count, sum theta, sum theta theta^T] and sums it over the ranks with ONE ncclAllReduce of
 * 1 + D + D^2 doubles on the handle's stream; without a communicator it only packs)  ->  klhr_pool_fetch (one D2H copy)
 * -> host eigen-decomposition  ->  klhr_pool_apply_full.  No PyTorch anywhere on this path.
 * replaces: nothing in the reference (its fan-out is `parallel -j` over independent processes, run_experiments:27);
 * this is the north-star's "RCCL all-reduce ... to pool the online-PCA covariance across shards".                         */
#define KLHR_COMM_ID_BYTES 128
KLHR_API int klhr_comm_unique_id(void *id_out, size_t bytes);
KLHR_API int klhr_comm_init(klhr_sampler *s, const void *id_in, size_t bytes, int rank, int world);
KLHR_API int klhr_comm_world(const klhr_sampler *s);            /* 0: no communicator */
KLHR_API int klhr_pool_reset(klhr_sampler *s);
KLHR_API int klhr_pool_accumulate(klhr_sampler *s, int64_t first_iter, int64_t n_iters);
KLHR_API int klhr_pool_allreduce(klhr_sampler *s);
KLHR_API int klhr_pool_fetch(klhr_sampler *s, double *out, size_t n_out);

/* ---- timing ---------------------------------------------------------------------------------- */
/* HIP-event time (ms) of the kernels launched by the most recent klhr_step on the handle's stream,
 * and the number of kernel launches it made.                                                      */
KLHR_API int klhr_last_step_timing(klhr_sampler *s, double *kernel_ms, int *n_launches);
/* Per-kernel timing for the roofline of bench.py.  The split-phase pipeline (klhr_kernel_variant >= 2) runs two or three kernels per
 * iteration; with profiling enabled klhr_step runs its chains as one group and brackets every kernel with HIP events on the
 * handle's stream.  klhr_last_kernel_times: mean duration (ms) and launch count of [streaming kernel, mode-search kernel,
 * KL-fit kernel] over the most recent klhr_step (monolithic kernels: slot 0 = the hot kernel).                            */
KLHR_API int klhr_set_kernel_profiling(klhr_sampler *s, int enable);
KLHR_API int klhr_last_kernel_times(klhr_sampler *s, double ms3[3], int n3[3]);

#ifdef __cplusplus
}
#endif
#endif /* KLHR_H */
