// klhr_api.hip — host side of libklhr_hip.so: the C ABI declared in include/klhr.h.
//
// Owns all device memory of a sampler handle (chain-major SoA in HBM; the HIP streams are a per-device pool shared by all handles), and
// launches the gfx950 kernels of klhr_kernels.hpp.  No PyTorch, no CPU fallback: every compute entry
// point fails with KLHR_ERR_HIP when the device is unusable.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "klhr_kernels.hpp"

namespace klhr {   // hot-path launchers, one per (variant, over-relaxation) translation unit (klhr_hot.hip)
void launch_hot_v0_o0(const KParams &, int, bool, hipStream_t);
void launch_hot_v0_o1(const KParams &, int, bool, hipStream_t);
void launch_hot_v1_o0(const KParams &, int, bool, hipStream_t);
void launch_hot_v1_o1(const KParams &, int, bool, hipStream_t);
void launch_hot_v2_o0(const KParams &, int, bool, hipStream_t);
void launch_hot_v2_o1(const KParams &, int, bool, hipStream_t);
void launch_hot_v3_o0(const KParams &, int, bool, hipStream_t);
// split-phase pipeline (klhr_split.hip), one unit per variant: what = 0 stream, 3 probe, 4 / 8 fit, 7 whole iterations per tile
void launch_split_v0(const KParams &, int, int, bool, int, int, hipStream_t);
void launch_split_v1(const KParams &, int, int, bool, int, int, hipStream_t);
void launch_split_v2(const KParams &, int, int, bool, int, int, hipStream_t);
}  // namespace klhr

using namespace klhr;

static thread_local std::string g_create_err;

// RCCL is loaded on first use (dlopen): the library has no link-time dependency on it, single-GPU users never touch it.
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {
struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi &rccl() {
    static RcclApi a;
    if (a.h) return a;
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        a.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (a.h) break;
    }
    if (!a.h) return a;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.h, "ncclCommInitRank");
    a.AllReduce = (decltype(a.AllReduce))dlsym(a.h, "ncclAllReduce");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.h, "ncclCommDestroy");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.h, "ncclGetErrorString");
    a.ok = a.GetUniqueId && a.CommInitRank && a.AllReduce && a.CommDestroy && a.GetErrorString;
    return a;
}
}  // namespace

// HIP streams of a device, shared by every handle on it and kept for the life of the process.  HIP maps streams onto a few
// hardware queues in creation order; streams created and destroyed per handle ended up, for later handles of a process,
// with two chain groups on ONE hardware queue — their kernels then ran one after the other (measured: funnel-100 x 131 072
// dropped from 1.39e8 to 0.90e8 it/s for the fourth sampler of a process).  Slot 0 is a handle's main stream, slots 1-3
// carry the chain groups of the split pipeline.  Handles used at the same time share the streams: their work interleaves
// in stream order, which is correct (every call forks and joins through events) if not concurrent.
static hipStream_t g_stream_pool[64][4];
static std::mutex g_stream_pool_mutex;          // handles may be created from several host threads
static int stream_from_pool(int device, int slot, hipStream_t *out) {
    if (device < 0 || device >= 64 || slot < 0 || slot >= 4) return -1;
    std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
    if (!g_stream_pool[device][slot]) {
        // all four at once, at the first handle of the process on this device: created later, after other work had gone
        // through the device, the group streams shared hardware queues (funnel-100 x 131 072: 0.87e8 instead of 1.36e8 it/s)
        for (int k = 0; k < 4; k++)
            if (!g_stream_pool[device][k] && hipStreamCreateWithFlags(&g_stream_pool[device][k], hipStreamNonBlocking) != hipSuccess) return -1;
    }
    *out = g_stream_pool[device][slot];
    return 0;
}

struct klhr_sampler {
    klhr_config cfg{};
    int C = 0, D = 0, ncol = 0, S = 0, S_lane = 0, n_model = 0;
    bool use_lane = false;       // lane-per-chain kernel (small D) instead of the octet-per-chain kernel
    bool split = false;          // split-phase pipeline (klhr_split.hpp): tiled state layout, two kernels per iteration
    size_t Cp = 0;               // chains the [C][D...] arrays are allocated for (whole tiles in the tiled layout)
    int tw = 64;                 // tile width of the tiled layout: chains per tile = lanes of a streaming wavefront in use
    int groups = 1;              // klhr_step runs the chains as this many groups on separate HIP streams (split pipeline)
    hipStream_t gstream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t gev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr;
    // per-kernel timing of the split pipeline (klhr_set_kernel_profiling): events between the kernels of every iteration
    // pooled adaptation: device-resident Gram (Dp x Dp) and packed statistics (1 + D + D^2), RCCL communicator
    double *d_gram = nullptr, *d_pool = nullptr;
    void *comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    bool fitn = false;           // split pipeline: both fit stages in the node-parallel kernel (few chains: klhr_fitn_kernel)
    bool last_fused = false;     // the most recent klhr_step ran as one klhr_tile_kernel launch
    long long small_max = 16384; // split pipeline: up to this many chains 16-chain tiles and the tile kernel
    bool fused = false;          // split pipeline, few chains: whole iterations per tile in one launch (klhr_tile_kernel)
    bool profile = false;
    std::vector<hipEvent_t> pev;
    int pev_used = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing_pending = false;
    double last_ms = 0;
    int last_launches = 0;
    bool seeded = false;
    std::string err;
    KParams kp{};
    double *d_aux = nullptr;
    double *d_draws = nullptr;
    size_t draws_cap = 0;          // bytes
    int64_t draws_iters = 0;       // iterations recorded by the last recording call
    std::vector<void *> allocs;
};

#define HIPCHECK(s, call)                                                                                  \
    do {                                                                                                   \
        hipError_t _e = (call);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            char _b[512];                                                                                  \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            (s)->err = _b;                                                                                 \
            return KLHR_ERR_HIP;                                                                           \
        }                                                                                                  \
    } while (0)

// ---- NumPy SeedSequence(seed).generate_state(4, uint64) -> PCG64 (numpy/random/bit_generator.pyx, _pcg64.pyx)
extern "C" void klhr_seed_state(uint64_t seed, uint64_t out4[4]) {
    const uint32_t INIT_A = 0x43b0d7e5u, MULT_A = 0x931e8875u, INIT_B = 0x8b51f9ddu, MULT_B = 0x58f38dedu;
    const uint32_t MIX_L = 0xca01f9ddu, MIX_R = 0x4973f715u;
    uint32_t entropy[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    const int n_entropy = entropy[1] ? 2 : 1;
    uint32_t hash_const = INIT_A;
    auto hashmix = [&](uint32_t v) {
        v ^= hash_const;
        hash_const *= MULT_A;
        v *= hash_const;
        v ^= v >> 16;
        return v;
    };
    auto mix = [&](uint32_t x, uint32_t y) {
        uint32_t r = MIX_L * x - MIX_R * y;
        r ^= r >> 16;
        return r;
    };
    uint32_t pool[4];
    for (int i = 0; i < 4; i++) pool[i] = hashmix(i < n_entropy ? entropy[i] : 0u);
    for (int src = 0; src < 4; src++)
        for (int dst = 0; dst < 4; dst++)
            if (src != dst) pool[dst] = mix(pool[dst], hashmix(pool[src]));
    uint32_t words[8];
    uint32_t hb = INIT_B;
    for (int i = 0; i < 8; i++) {
        uint32_t v = pool[i & 3];
        v ^= hb;
        hb *= MULT_B;
        v *= hb;
        v ^= v >> 16;
        words[i] = v;
    }
    uint64_t q[4];
    for (int i = 0; i < 4; i++) q[i] = (uint64_t)words[2 * i] | ((uint64_t)words[2 * i + 1] << 32);
    typedef unsigned __int128 u128;
    const u128 MULT = (((u128)0x2360ED051FC65DA4ULL) << 64) | (u128)0x4385DF649FCCF645ULL;
    const u128 initstate = ((u128)q[0] << 64) | q[1], initseq = ((u128)q[2] << 64) | q[3];
    u128 inc = (initseq << 1) | 1, state = 0;
    state = state * MULT + inc;
    state += initstate;
    state = state * MULT + inc;
    out4[0] = (uint64_t)(state >> 64);
    out4[1] = (uint64_t)state;
    out4[2] = (uint64_t)(inc >> 64);
    out4[3] = (uint64_t)inc;
}

// WindowedAdaptation._calculate_windows (windowedadaptation.py:12-28): `for w in range(warmup + 1): if w == closewindow:
// record, then closewindow = next`.  Restated as a walk over the closures themselves; it ends when the next closure does
// not lie ahead of the loop variable (windowsize == 0, windowscale <= 0: the reference records one closure and never
// meets another) or lies beyond warmup.  Intermediate products in 128 bits: no overflow for any int64 arguments.
extern "C" int klhr_window_closures(int64_t warmup, int64_t windowsize, int64_t windowscale, int64_t *out, int cap) {
    typedef __int128 i128;
    int n = 0;
    if (!(warmup > windowsize)) return 0;
    i128 ws = windowsize, close = windowsize;
    while (close >= 0 && close <= warmup) {
        if (n < cap && out) out[n] = (int64_t)close;
        n++;
        if (n >= (1 << 24)) break;                      // (a schedule this long is refused by klhr_create)
        ws *= windowscale;
        if (ws > (i128)1 << 100 || ws < -((i128)1 << 100)) ws = ws > 0 ? (i128)1 << 100 : -((i128)1 << 100);
        const i128 next = (close + (i128)windowscale * ws >= warmup) ? (i128)warmup : close + ws;
        if (next <= close) break;                       // the loop variable has passed it: no further closure
        close = next;
    }
    return n;
}

extern "C" int klhr_abi_version(void) { return KLHR_ABI_VERSION; }
#ifndef KLHR_BUILD_ID
#define KLHR_BUILD_ID "unknown"
#endif
extern "C" const char *klhr_build_id(void) { return KLHR_BUILD_ID; }
extern "C" int klhr_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
extern "C" const char *klhr_last_error(const klhr_sampler *s) { return s ? s->err.c_str() : g_create_err.c_str(); }
extern "C" int klhr_dim(const klhr_sampler *s) { return s ? s->D : KLHR_ERR_ARG; }
extern "C" int klhr_chains(const klhr_sampler *s) { return s ? s->C : KLHR_ERR_ARG; }
extern "C" int klhr_tile_width(const klhr_sampler *s) { return s ? (s->split ? s->tw : 0) : KLHR_ERR_ARG; }
extern "C" int klhr_kernel_variant(const klhr_sampler *s) { return s ? (s->split ? (s->fused ? 5 : (s->fitn ? 3 : 2)) : (s->use_lane ? 1 : 0)) : KLHR_ERR_ARG; }

// ---- host <-> device copies of [C][W] state arrays: the host side is always chain-major (the NumPy layout); the device
// side is tiled [tile][w][64 lanes] for handles of the split pipeline
template <class T>
static void host_to_tiled(const T *src, T *dst, size_t C, size_t W, size_t tw) {
    for (size_t c = 0; c < C; c++)
        for (size_t w = 0; w < W; w++) dst[((c / tw) * W + w) * tw + (c % tw)] = src[c * W + w];
}
template <class T>
static void tiled_to_host(const T *src, T *dst, size_t C, size_t W, size_t tw) {
    for (size_t c = 0; c < C; c++)
        for (size_t w = 0; w < W; w++) dst[c * W + w] = src[((c / tw) * W + w) * tw + (c % tw)];
}

template <class T>
static int dalloc(klhr_sampler *s, T **p, size_t n) {
    void *q = nullptr;
    HIPCHECK(s, hipMalloc(&q, n * sizeof(T) ? n * sizeof(T) : 8));
    s->allocs.push_back(q);
    *p = (T *)q;
    return 0;
}

// the direction cache follows eigvecs / eigvals / cov (split-pipeline handles; the streaming kernel's closure writes it itself)
static int refresh_dircache(klhr_sampler *s) {
    if (!s->kp.dir_m) return 0;
    const int bs = 64;
    hipLaunchKernelGGL(klhr_dircache_kernel, dim3((s->C + bs - 1) / bs), dim3(bs), 0, s->stream, s->kp);
    HIPCHECK(s, hipGetLastError());
    return 0;
}

static int reset_state(klhr_sampler *s) {
    KParams &k = s->kp;
    const size_t Cc = s->C, D = s->D, nc = s->ncol, J = s->cfg.J;
    const size_t C = s->Cp;        // allocation size of the [C][D...] arrays (== chains, or whole tiles)
    std::vector<double> ones(C * (D > nc ? D : nc), 1.0);        // cov (C x D) and eigvals (C x ncol; ncol may exceed D)
    std::vector<int> ident(C * D);
    for (size_t c = 0; c < C; c++)
        for (size_t i = 0; i < D; i++) ident[s->split ? ((c / s->tw) * D + i) * s->tw + (c % s->tw) : c * D + i] = (int)i;
    HIPCHECK(s, hipMemcpyAsync(k.cov, ones.data(), C * D * 8, hipMemcpyHostToDevice, s->stream));
    HIPCHECK(s, hipMemcpyAsync(k.rank, ident.data(), C * D * 4, hipMemcpyHostToDevice, s->stream));
    HIPCHECK(s, hipMemcpyAsync(k.eigvals, ones.data(), Cc * nc * 8, hipMemcpyHostToDevice, s->stream));
    if (k.fitbuf) {
        HIPCHECK(s, hipMemsetAsync(k.fitbuf, 0, (size_t)F_NFIELDS * C * 8, s->stream));
    }
    HIPCHECK(s, hipMemsetAsync(k.theta, 0, C * D * 8, s->stream));
    HIPCHECK(s, hipMemsetAsync(k.mean, 0, C * D * 8, s->stream));
    HIPCHECK(s, hipMemsetAsync(k.eigvecs, 0, C * D * nc * 8, s->stream));
    HIPCHECK(s, hipMemsetAsync(k.om_m, 0, C * D * 8, s->stream));
    HIPCHECK(s, hipMemsetAsync(k.om_v, 0, C * D * 8, s->stream));
    if (k.omd_m) {
        HIPCHECK(s, hipMemsetAsync(k.omd_m, 0, C * D * 8, s->stream));
        HIPCHECK(s, hipMemsetAsync(k.omd_v, 0, C * D * 8, s->stream));
    }
    HIPCHECK(s, hipMemsetAsync(k.pca_v, 0, C * D * J * 8, s->stream));
    HIPCHECK(s, hipMemsetAsync(k.acc, 0, C * 8, s->stream));
    HIPCHECK(s, hipMemsetAsync(k.counters, 0, C * 4 * 8, s->stream));
    HIPCHECK(s, hipMemsetAsync(k.grad_evals, 0, C * 8, s->stream));
    HIPCHECK(s, hipMemsetAsync(k.status, 0, C * 4, s->stream));
    HIPCHECK(s, hipMemsetAsync(k.rng, 0, C * 4 * 8, s->stream));
    HIPCHECK(s, hipMemsetAsync(k.d_rho, 0, C * D * 8, s->stream));
    HIPCHECK(s, hipMemsetAsync(k.d_eta, 0, C * 4 * 8, s->stream));
    HIPCHECK(s, hipMemsetAsync(k.d_diag, 0, C * KLHR_NDIAG * 8, s->stream));
    {   // KLHR variant: Smoother(K) state per chain = [x = K, count = 0, K, moved = 0]
        std::vector<double> ks(C * 4, 0.0);
        for (size_t c = 0; c < Cc; c++) { ks[c * 4 + 0] = (double)s->cfg.K; ks[c * 4 + 2] = (double)s->cfg.K; }
        HIPCHECK(s, hipMemcpyAsync(k.kstate, ks.data(), C * 4 * 8, hipMemcpyHostToDevice, s->stream));
    }
    { int rc = refresh_dircache(s); if (rc) return rc; }
    HIPCHECK(s, hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int klhr_create(const klhr_config *cfg, const double *model_data, size_t n_model_data, klhr_sampler **out) {
    if (out) *out = nullptr;
    auto fail = [&](const char *m) { g_create_err = m; return KLHR_ERR_ARG; };
    if (!cfg || !out || cfg->struct_size != sizeof(klhr_config)) return fail("klhr_create: bad config struct");
    if (cfg->N != 8) return fail("klhr_create: this build maps one lane per quadrature node and supports N == 8 only");
    if (cfg->J < 1 || cfg->J > 3) return fail("klhr_create: need 1 <= J <= 3");
    if (cfg->overrelaxed && (cfg->K < 1 || cfg->K > 64)) return fail("klhr_create: overrelaxed needs 1 <= K <= 64");
    if (cfg->chains < 1) return fail("klhr_create: chains < 1");
    if (cfg->variant < 0 || cfg->variant > 3) return fail("klhr_create: variant must be 0 (KLHRSINH), 1 (KLHR), 2 (SUBKLHRSINH) or 3 (Slice)");
    if (!model_data || n_model_data < 1 || !cfg->quad_x || !cfg->quad_w || !cfg->quad_asinh_x) return fail("klhr_create: null data");
    int D = 0, n = 0;
    std::vector<double> aux;
    switch (cfg->model_id) {
    case KLHR_MODEL_FUNNEL: n = (int)model_data[0]; D = n + 1; break;
    case KLHR_MODEL_NORMAL: D = (int)model_data[0]; break;
    case KLHR_MODEL_ILL_NORMAL:
        D = (int)model_data[0];
        for (int i = 0; i < D; i++) aux.push_back(1.0 / ((double)(i + 1) / std::sqrt((double)D)));
        break;
    case KLHR_MODEL_GLMM_POISSON:
        n = (int)model_data[0];
        if (n < 1 || n_model_data < (size_t)(1 + 2 * n)) return fail("klhr_create: glmm-poisson data too short");
        D = n + 5;
        aux.resize(4 * (size_t)n);
        for (int i = 0; i < n; i++) {
            double y = model_data[1 + i];
            aux[i] = y; aux[n + i] = y * y; aux[2 * n + i] = y * y * y; aux[3 * n + i] = model_data[1 + n + i];
        }
        break;
    case KLHR_MODEL_AR1: D = (int)model_data[0]; break;
    case KLHR_MODEL_EARNINGS: {
        n = (int)model_data[0];
        if (n < 2 || n_model_data < (size_t)(1 + 2 * n)) return fail("klhr_create: earnings data too short");
        D = 4;
        double eb = 0, hb = 0;
        for (int i = 0; i < n; i++) { eb += model_data[1 + i]; hb += model_data[1 + n + i]; }
        eb /= n; hb /= n;
        double see = 0, seh = 0, shh = 0;
        for (int i = 0; i < n; i++) { double ec = model_data[1 + i] - eb, hc = model_data[1 + n + i] - hb; see += ec * ec; seh += ec * hc; shh += hc * hc; }
        aux = {(double)n, eb, hb, see, seh, shh};
        break; }
    default: return fail("klhr_create: unknown model_id");
    }
    if (D < 1) return fail("klhr_create: model dimension < 1");

    klhr_sampler *s = new klhr_sampler();
    s->cfg = *cfg;
    s->cfg.quad_x = s->cfg.quad_w = s->cfg.quad_asinh_x = nullptr;
    s->C = cfg->chains; s->D = D; s->ncol = cfg->J + (cfg->eigen_method_one ? 1 : 0);
    // LDS per chain: D double2 (theta, rho) + D doubles scratch; stride staggered over banks (S % 32 == 4)
    int S = 3 * D;
    while (S % 32 != 4) S++;
    s->S = S;
    // lane-per-chain kernel: 64 chains per wavefront, stride ≡ 2 (mod 32) doubles; used while it fits 64 KB of LDS
    int SL = 3 * D;
    while (SL % 32 != 2) SL++;
    s->S_lane = SL;
    s->use_lane = (D <= 26) && (((size_t)64 * SL + 1072) * 8 <= 64 * 1024);   // + LANE_LDS_EXTRA (klhr_lane_kernel.hpp)
    // split-phase pipeline: line-separable targets, any variant but the slice sampler; one [D][64] LDS column block per tile
    // (ar1: its z-transform is linear on a line, so the same closed form applies; the element-wise gradient that scale_dir_cov
    //  tracks needs a coordinate's neighbours, which the streaming kernel's per-element pass does not carry: octet kernel then)
    const bool split_ok = (cfg->model_id == KLHR_MODEL_FUNNEL || cfg->model_id == KLHR_MODEL_ILL_NORMAL || cfg->model_id == KLHR_MODEL_NORMAL ||
                           (cfg->model_id == KLHR_MODEL_AR1 && !cfg->scale_dir_cov)) &&
                          cfg->variant != 3 && ((size_t)D * 16 + 512) * 8 <= 64 * 1024;
    // default: the lane-per-chain kernel while a chain's row fits its LDS share (D <= 26) — from 16 385 chains on for the separable
    // targets, see below —, the split pipeline above that dimension for the separable targets, the octet kernel for everything else.  KLHR_KERNEL = octet | lane | mono | split overrides
    // (mono: the monolithic kernel the dimension selects; split: the split pipeline wherever it is supported) — used by
    // the parity tests, which run every supported mapping.
    // Dimension <= 26, separable target: lane kernel and split pipeline return the same bits (the lane kernel takes its line sums from the
    // unnormalised direction as the streaming kernel does, and every form patches the clip bound per node: tools/lane_vs_split.py) — so the
    // mapping may follow the chain count.  Up to 16 384 chains a lane-per-chain wavefront is 64 chains on one SIMD of 1 024 and an iteration
    // lasts its 158 us whatever the count; the tile kernel spreads a chain over 4-8 lanes: funnel-10, us per iteration, lane kernel | tile
    // kernel: 256-8 192 chains 158 | 78, 16 384: 158 | 129, 32 768: 169 | 277.  (Not with scale_dir_cov — the gradient's moments are formed
    // differently in the two — and not ar1, which the lane kernel evaluates element-wise.)
    const bool lane_equals_split = (cfg->model_id == KLHR_MODEL_FUNNEL || cfg->model_id == KLHR_MODEL_ILL_NORMAL || cfg->model_id == KLHR_MODEL_NORMAL) &&
                                   !cfg->scale_dir_cov;
    s->split = split_ok && (!s->use_lane || (lane_equals_split && s->C <= 16384));
    if (cfg->variant == 3) s->use_lane = false;          // the slice sampler runs on the octet kernel only
    else
    if (const char *kk = getenv("KLHR_KERNEL")) {
        if (!strcmp(kk, "octet")) { s->use_lane = false; s->split = false; }
        else if (!strcmp(kk, "lane")) { s->split = false; if (((size_t)64 * SL + 1072) * 8 <= 64 * 1024) s->use_lane = true; }
        else if (!strcmp(kk, "mono")) s->split = false;
        else if (!strcmp(kk, "split")) s->split = split_ok;
    }
    if (s->split) s->use_lane = false;
    if (s->split) {
        // Tile width of the streaming kernel: 64 lanes per tile use the whole wavefront; with few chains a launch lasts as long
        // as one wavefront's walk through its phases whatever the width, and 16-chain tiles can run a tile's tail and head
        // on two wavefronts (pair mode below, up to 16 384 chains).  Measured, ill-normal-100 (us per iteration, 16 | 64):
        // 32 768 chains 299 | 270, 65 536: 642 | 420, 131 072: 1 114 | 644.
        // (16-chain tiles are the tile kernel's: whole iterations per tile, quad mode.  Against the 64-chain kernel sequence it wins up to
        //  16 384 chains on every target and up to 32 768 on the Gaussian ones, whose fits are short — measured, us per iteration,
        //  sequence | tile kernel: ill-normal-100 x 24 576: 221 | 188, x 32 768: 266 | 194; funnel-100 x 24 576: 373 | 330, x 32 768: 409 | 420;
        //  funnel-50 x 32 768: 317 | 359)
        s->small_max = cfg->model_id == KLHR_MODEL_FUNNEL ? 16384 : 32768;
        int tw = s->C <= s->small_max ? 16 : 64;
        if (const char *e = getenv("KLHR_TILE")) { const int v = atoi(e); if (v == 16 || v == 64) tw = v; }
        if (((size_t)D * tw + 512) * 8 > 160 * 1024) tw = 16;       // (one column block + the tables must fit a CU's LDS)
        s->tw = tw;
    }
    s->Cp = s->split ? ((size_t)s->C + s->tw - 1) / s->tw * s->tw : (size_t)s->C;
    auto bail = [&](int rc) { g_create_err = s->err; klhr_destroy(s); return rc; };
    {
        hipError_t e = hipSetDevice(cfg->device);
        if (e != hipSuccess) { s->err = std::string("hipSetDevice failed: ") + hipGetErrorString(e); return bail(KLHR_ERR_HIP); }
    }
    {
        // octet kernel: 8 chain rows of S doubles + the staged model data (ill-normal D, glmm-poisson 4n doubles) in 64 KB
        const size_t naux = cfg->model_id == KLHR_MODEL_GLMM_POISSON ? (size_t)4 * n : (cfg->model_id == KLHR_MODEL_ILL_NORMAL ? (size_t)D : 0);
        if (!s->split && ((size_t)8 * S + naux) * 8 > 64 * 1024) {
            s->err = "klhr_create: model dimension too large for the LDS layout of the octet kernel (8 rows of 3 D doubles + staged model data "
                     "must fit 64 KB: D <= ~340; ill-normal D <= ~320, glmm-poisson n <= ~290)";
            return bail(KLHR_ERR_ARG);
        }
    }
#define TRY(x) do { int _rc = (x); if (_rc) return bail(_rc); } while (0)
    auto hip_try = [&](hipError_t e, const char *what) { if (e != hipSuccess) { s->err = std::string(what) + ": " + hipGetErrorString(e); return (int)KLHR_ERR_HIP; } return 0; };
    if (stream_from_pool(cfg->device, 0, &s->stream)) { s->err = "hipStreamCreate failed"; return bail(KLHR_ERR_HIP); }
    TRY(hip_try(hipEventCreate(&s->ev0), "hipEventCreate"));
    TRY(hip_try(hipEventCreate(&s->ev1), "hipEventCreate"));
    KParams &k = s->kp;
    const size_t C = s->Cp, nc = s->ncol, J = cfg->J;
    k.C = s->C; k.D = D; k.S = S; k.ncol = s->ncol; k.J = cfg->J;
    k.tiled = s->split ? 1 : 0; k.tw = s->tw; k.Cpad = (long long)s->Cp; k.fstride = (long long)s->Cp; k.draw_slot = 0;
    k.draw_stride = (long long)s->C * D;
    if (s->split) {
        TRY(dalloc(s, &k.fitbuf, (size_t)F_NFIELDS * C));
        TRY(dalloc(s, &k.dir_m, C * D)); TRY(dalloc(s, &k.dir_s, C * D));        // direction cache (KParams::dir_m)
        // few chains: the fit's latency (its slowest chain's sequence of passes) is the launch's duration — 8 lanes per chain
        // shorten the pass (klhr_split.hpp: klhr_fitn_kernel).  Measured, ill-normal-100, both stages: 16 384 chains 77 us against
        // 159, 32 768: 118 / 171, 65 536: 204 / 183, funnel-100 x 131 072: 952 / 676 — from 65 536 chains on lane = chain has the throughput
        s->fitn = C <= 32768;
        if (const char *e = getenv("KLHR_FITN")) s->fitn = atoi(e) != 0;
        // few chains: klhr_step runs whole iterations per 16-chain tile in ONE launch (klhr_tile_kernel: the pair-mode streaming
        // body and the node-parallel fit body in a loop over the draws, one workgroup per tile).  Needs what those two need.
        // Up to 16 384 chains all tiles are resident (2 048 wavefronts: two per SIMD, the kernel's 256-register build; measured in
        // klhr_split.hpp); Gaussian targets up to 32 768 (a second round of workgroups, see the tile width above).
        const bool fused_ok = s->tw == 16 && s->fitn && ((size_t)2 * D * 16 + 512 + 66) * 8 <= 160 * 1024;
        long long fused_max = s->small_max;
        if (const char *e = getenv("KLHR_FUSED_MAX")) fused_max = atoll(e);                  // (tuning aid)
        s->fused = fused_ok && s->C <= fused_max;
        if (const char *e = getenv("KLHR_FUSED")) s->fused = s->fused && atoi(e) != 0;      // (0: the kernel sequence; used by the parity tests)
        // Chain groups of klhr_step: the streaming kernel is bound by memory, the fit kernels by the fp64 ALUs; with the chains
        // in two groups on two HIP streams one group's fit overlaps the other group's streaming.  (Tuning aid: KLHR_GROUPS.)
        s->groups = s->C >= 32768 ? 4 : (s->C >= 8192 ? 2 : 1);          // (measured: funnel-100 x 131 072: 1.0e8 / 1.2e8 / 1.4e8 it/s with 1 / 2 / 4 groups)
        if (const char *e = getenv("KLHR_GROUPS")) { const int v = atoi(e); if (v >= 1 && v <= 4) s->groups = v; }
        for (int g = 1; g < s->groups; g++) {
            if (stream_from_pool(cfg->device, g, &s->gstream[g])) { s->err = "hipStreamCreate failed"; return bail(KLHR_ERR_HIP); }
            TRY(hip_try(hipEventCreateWithFlags(&s->gev[g], hipEventDisableTiming), "hipEventCreate"));
        }
        TRY(hip_try(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming), "hipEventCreate"));
    }
    {   // streaming kernel: as many tiles per workgroup (1-4 wavefronts) as fit 160 KB of LDS next to the shared 4 KB ziggurat
        // tables; small column blocks: 4 wavefronts per workgroup, several workgroups per CU
        const size_t zb = (size_t)D * s->tw * 8, cap = 160 * 1024 - 4096;
        size_t nw = zb ? cap / zb : 4;
        k.stream_waves = (int)(nw < 1 ? 1 : (nw > 4 ? 4 : nw));
        if (const char *e = getenv("KLHR_STREAM_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 4) k.stream_waves = v; }
        // two wavefronts per tile (tail | head) in the streaming kernel, for launches with a tail and a head on 16-chain tiles
        // (klhr_split.hip) while both wavefronts of every tile are resident at once: up to 16 384 chains (1 024 tiles, 2 048
        // wavefronts at two per SIMD).  ill-normal-100: 4 096 chains 213 -> 145 us per iteration, 8 192: 233 -> 152, 16 384: 224 -> 177
        // (159 with two chain groups), 32 768: 296 -> 311 (second round of wavefronts: off).
        k.stream_pair = s->C <= 16384;
        if (const char *e = getenv("KLHR_STREAM_PAIR")) k.stream_pair = atoi(e) != 0;
    }
    k.flags = (cfg->scale_dir_cov ? FLAG_SCALE_DIR_COV : 0) | (cfg->eigen_method_one ? FLAG_EIGEN_ONE : 0) |
              (cfg->overrelaxed ? FLAG_OVERRELAXED : 0);
    k.K = cfg->K;
    k.tol = cfg->tol; k.scale_clip = cfg->scale_clip; k.grad_clip = cfg->grad_clip; k.initscale = cfg->initscale; k.l = cfg->l;
    for (int i = 0; i < 8; i++) { k.xq[i] = cfg->quad_x[i]; k.wq[i] = cfg->quad_w[i]; k.asx[i] = cfg->quad_asinh_x[i]; }
    k.variant = cfg->variant;
    k.sl_w = 1.0; k.sl_lower = -INFINITY; k.sl_upper = INFINITY; k.sl_maxiter = 100000;      // slice.py:19-22 defaults
    TRY(dalloc(s, &s->d_aux, aux.size()));
    if (!aux.empty()) TRY(hip_try(hipMemcpy(s->d_aux, aux.data(), aux.size() * 8, hipMemcpyHostToDevice), "hipMemcpy aux"));
    k.md.aux = s->d_aux; k.md.D = D; k.md.n = n;
    TRY(dalloc(s, &k.theta, C * D)); TRY(dalloc(s, &k.rng, C * 4)); TRY(dalloc(s, &k.mean, C * D)); TRY(dalloc(s, &k.cov, C * D));
    TRY(dalloc(s, &k.eigvecs, C * D * nc)); TRY(dalloc(s, &k.eigvals, C * nc > C * D ? C * nc : C * D));
    TRY(dalloc(s, &k.om_m, C * D)); TRY(dalloc(s, &k.om_v, C * D));
    if (cfg->scale_dir_cov) { TRY(dalloc(s, &k.omd_m, C * D)); TRY(dalloc(s, &k.omd_v, C * D)); }
    TRY(dalloc(s, &k.pca_v, C * D * J)); TRY(dalloc(s, &k.acc, C)); TRY(dalloc(s, &k.rank, C * D));
    TRY(dalloc(s, &k.counters, C * 4)); TRY(dalloc(s, &k.grad_evals, C)); TRY(dalloc(s, &k.status, C));
    TRY(dalloc(s, &k.d_rho, C * D)); TRY(dalloc(s, &k.d_eta, C * 4)); TRY(dalloc(s, &k.d_diag, C * KLHR_NDIAG));
    TRY(dalloc(s, &k.kstate, C * 4));
    k.warmup = cfg->warmup; k.windowsize = cfg->windowsize;
    // window-closure draws (windowedadaptation.py:22-37): any number of them (windowscale = 1 gives warmup / windowsize)
    int ncl = klhr_window_closures(cfg->warmup, cfg->windowsize, cfg->windowscale, nullptr, 0);
    if (ncl < 0 || ncl > (1 << 20)) { s->err = "klhr_create: bad adaptation-window schedule"; return bail(KLHR_ERR_ARG); }
    std::vector<int64_t> cl((size_t)(ncl > 0 ? ncl : 1), 0);
    if (ncl > 0) klhr_window_closures(cfg->warmup, cfg->windowsize, cfg->windowscale, cl.data(), ncl);
    k.nclos = ncl;
    {
        long long *d_cl = nullptr;
        TRY(dalloc(s, &d_cl, cl.size()));
        std::vector<long long> hcl(cl.begin(), cl.end());
        TRY(hip_try(hipMemcpy(d_cl, hcl.data(), hcl.size() * sizeof(long long), hipMemcpyHostToDevice), "hipMemcpy closures"));
        k.closures = d_cl;
    }
    TRY(reset_state(s));
#undef TRY
    *out = s;
    return KLHR_OK;
}

extern "C" void klhr_destroy(klhr_sampler *s) {
    if (!s) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (hipEvent_t e : s->pev) (void)hipEventDestroy(e);
    if (s->comm && rccl().ok) (void)rccl().CommDestroy((ncclComm_t)s->comm);
    if (s->d_gram) (void)hipFree(s->d_gram);
    if (s->d_pool) (void)hipFree(s->d_pool);
    for (void *p : s->allocs) (void)hipFree(p);
    if (s->d_draws) (void)hipFree(s->d_draws);
    for (int g = 1; g < 4; g++) {
        if (s->gstream[g]) (void)hipStreamSynchronize(s->gstream[g]);          // (pooled streams are not destroyed)
        if (s->gev[g]) (void)hipEventDestroy(s->gev[g]);
    }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
}

extern "C" int klhr_sync(klhr_sampler *s) {
    if (!s) return KLHR_ERR_ARG;
    HIPCHECK(s, hipStreamSynchronize(s->stream));
    return KLHR_OK;
}

// ---- launches ------------------------------------------------------------------------------------
static int launch_iter(klhr_sampler *s, const KParams &kp_in) {
    KParams kp = kp_in;
    kp.S = s->use_lane ? s->S_lane : s->S;
    const bool o = s->cfg.overrelaxed != 0;
    if (s->cfg.variant == 3) launch_hot_v3_o0(kp, s->cfg.model_id, false, s->stream);
    else if (s->cfg.variant == 2) (o ? launch_hot_v2_o1 : launch_hot_v2_o0)(kp, s->cfg.model_id, s->use_lane, s->stream);
    else if (s->cfg.variant == 1) (o ? launch_hot_v1_o1 : launch_hot_v1_o0)(kp, s->cfg.model_id, s->use_lane, s->stream);
    else (o ? launch_hot_v0_o1 : launch_hot_v0_o0)(kp, s->cfg.model_id, s->use_lane, s->stream);
    HIPCHECK(s, hipGetLastError());
    return 0;
}

static void launch_split(klhr_sampler *s, const KParams &kp, int what, int do_tail, int do_head, hipStream_t st = nullptr) {
    const bool o = s->cfg.overrelaxed != 0;
    if (!st) st = s->stream;
    if (s->cfg.variant == 2) launch_split_v2(kp, s->cfg.model_id, what, o, do_tail, do_head, st);
    else if (s->cfg.variant == 1) launch_split_v1(kp, s->cfg.model_id, what, o, do_tail, do_head, st);
    else launch_split_v0(kp, s->cfg.model_id, what, o, do_tail, do_head, st);
}

// The view of chain group g of G (whole tiles, contiguous) of a split-pipeline handle: every pointer advanced to the
// group's first chain, C / Cpad the group's own.  Returns false for an empty group.
static bool group_view(const klhr_sampler *s, const KParams &kp, int g, int G, KParams &out) {
    out = kp;
    if (G <= 1) return g == 0;
    const long long tw = s->tw, tiles = (long long)s->Cp / tw, tg = (tiles + G - 1) / G;
    const long long c0 = g * tg * tw;
    if (c0 >= s->C) return false;
    long long c1 = (g + 1) * tg * tw;
    if (c1 > (long long)s->Cp) c1 = (long long)s->Cp;
    const long long D = s->D, nc = s->ncol, J = s->cfg.J;
    out.C = (int)((c1 < s->C ? c1 : s->C) - c0);
    out.Cpad = c1 - c0;
    out.theta += c0 * D; out.mean += c0 * D; out.cov += c0 * D; out.om_m += c0 * D; out.om_v += c0 * D; out.d_rho += c0 * D;
    if (out.omd_m) { out.omd_m += c0 * D; out.omd_v += c0 * D; }
    out.rank += c0 * D; out.eigvecs += c0 * D * nc; out.pca_v += c0 * D * J; out.eigvals += c0 * nc;
    out.rng += c0 * 4; out.counters += c0 * 4; out.kstate += c0 * 4; out.d_eta += c0 * 4;
    out.acc += c0; out.grad_evals += c0; out.status += c0; out.d_diag += c0 * KLHR_NDIAG;
    out.fitbuf += c0;
    if (out.dir_m) { out.dir_m += c0 * D; out.dir_s += c0 * D; }
    if (out.draws) out.draws += c0 * D;
    return true;
}

static int ensure_draws(klhr_sampler *s, int64_t n_iters) {
    size_t need = (size_t)n_iters * s->C * s->D * 8;
    if (need > s->draws_cap) {
        if (s->d_draws) { HIPCHECK(s, hipStreamSynchronize(s->stream)); HIPCHECK(s, hipFree(s->d_draws)); s->d_draws = nullptr; s->draws_cap = 0; }
        HIPCHECK(s, hipMalloc((void **)&s->d_draws, need));
        s->draws_cap = need;
    }
    return 0;
}

extern "C" int klhr_reserve_draws(klhr_sampler *s, int64_t n_iters) {
    if (!s) return KLHR_ERR_ARG;
    if (n_iters < 0) { s->err = "klhr_reserve_draws: n_iters < 0"; return KLHR_ERR_ARG; }
    return ensure_draws(s, n_iters);
}

static int run_phases(klhr_sampler *s, int phases, int64_t n_iters, int record_draws, int record_diag) {
    if (!s->seeded && (phases & (PH_DIR | PH_FIT | PH_MH))) { s->err = "klhr: chains are not seeded (call klhr_seed or set RNG/theta state)"; return KLHR_ERR_STATE; }
    HIPCHECK(s, hipSetDevice(s->cfg.device));
    if (record_draws) { int rc = ensure_draws(s, n_iters); if (rc) return rc; s->draws_iters = n_iters; }
    if (s->split && s->fused && phases == PH_ALL && !record_diag) {
        // (a launch is kept to a few tens of milliseconds: 256 draws; a call of more draws is a sequence of launches, each
        //  ending with the tail of its last draw — the same as consecutive calls)
        const int64_t chunk = 256;
        HIPCHECK(s, hipEventRecord(s->ev0, s->stream));
        int launches = 0;
        for (int64_t done = 0; done < n_iters; done += chunk) {
            KParams kp = s->kp;
            kp.phases = phases;
            kp.n_iters = (int)((n_iters - done) < chunk ? (n_iters - done) : chunk);
            kp.draws = record_draws ? s->d_draws + (size_t)done * s->C * s->D : nullptr;
            launch_split(s, kp, 7, 1, 1, s->stream);
            HIPCHECK(s, hipGetLastError());
            launches++;
        }
        HIPCHECK(s, hipEventRecord(s->ev1, s->stream));
        s->timing_pending = true;
        s->last_launches = launches;
        s->last_fused = true;
        return KLHR_OK;
    }
    s->last_fused = false;
    if (s->split) {
        // Split-phase pipeline (klhr_split.hpp).  Iteration k = [stream kernel: tail of k-1 (MH, adaptation, draw output) +
        // head of k (direction, line sums, the fit's normals)] -> mode-search kernel -> KL-fit kernel; one last stream launch
        // runs the tail of the final iteration.  Everything is enqueued on the handle's stream without host synchronisation.
        const int tailp = phases & (PH_MH | PH_ADAPT | PH_FIT), headp = phases & (PH_DIR | PH_FIT | PH_MH | PH_KL | PH_LINE);      // (a tail after a fit also books its evaluation counts)
        // whole draws without diagnostics run as chain groups on their own streams (fork / join through events);
        // per-kernel profiling runs one group so that the kernels of an iteration follow each other on one stream
        const int G = (phases == PH_ALL && !record_diag && !s->profile) ? s->groups : 1;
        const bool prof = s->profile && phases == PH_ALL;
        if (prof) {
            const size_t need = (size_t)(n_iters + 1) * 4;
            while (s->pev.size() < need) { hipEvent_t e; HIPCHECK(s, hipEventCreate(&e)); s->pev.push_back(e); }
            s->pev_used = 0;
        }
        HIPCHECK(s, hipEventRecord(s->ev0, s->stream));
        if (G > 1) {
            HIPCHECK(s, hipEventRecord(s->ev_fork, s->stream));
            for (int g = 1; g < G; g++) HIPCHECK(s, hipStreamWaitEvent(s->gstream[g], s->ev_fork, 0));
        }
        // (an error between the fork and the join must not leave the group streams unjoined for the next call: the loop records
        //  the first failure, the join below runs in every case)
        int rc_loop = KLHR_OK;
        auto hipok = [&](hipError_t e, const char *what) {
            if (e == hipSuccess || rc_loop != KLHR_OK) return e == hipSuccess;
            s->err = std::string(what) + " failed: " + hipGetErrorString(e);
            rc_loop = KLHR_ERR_HIP;
            return false;
        };
        auto mark = [&](int slot) { if (prof && hipok(hipEventRecord(s->pev[slot], s->stream), "hipEventRecord")) s->pev_used = slot + 1; };
        for (int64_t it = 0; it <= n_iters && rc_loop == KLHR_OK; it++) {
            KParams kp0 = s->kp;
            kp0.phases = phases;
            kp0.n_iters = 1;
            if (record_diag) kp0.flags |= FLAG_DIAG;
            const int do_tail = it > 0 && tailp, do_head = it < n_iters && headp;
            kp0.draws = (record_draws && do_tail) ? s->d_draws : nullptr;
            kp0.draw_slot = (int)(it - 1);
            for (int g = 0; g < G; g++) {
                KParams kp;
                if (!group_view(s, kp0, g, G, kp)) continue;
                hipStream_t st = g == 0 ? s->stream : s->gstream[g];
                mark((int)it * 4 + 0);
                if (do_tail || do_head) launch_split(s, kp, 0, do_tail, do_head, st);
                mark((int)it * 4 + 1);
                if (it < n_iters && (phases & PH_FIT)) {
                    mark((int)it * 4 + 2);
                    launch_split(s, kp, s->fitn ? 4 : 8, 0, 0, st);          // both fit stages: node-parallel (few chains) or lane = chain
                    mark((int)it * 4 + 3);
                }
                if (it < n_iters && (phases & (PH_KL | PH_LINE))) {
                    const char *rep = getenv("KLHR_PROBE_REPEAT");
                    kp.n_iters = rep ? atoi(rep) : 1;
                    launch_split(s, kp, 3, 0, 0, st);
                }
            }
            hipok(hipGetLastError(), "kernel launch");
        }
        for (int g = 1; g < G; g++) {
            hipok(hipEventRecord(s->gev[g], s->gstream[g]), "hipEventRecord");
            hipok(hipStreamWaitEvent(s->stream, s->gev[g], 0), "hipStreamWaitEvent");
        }
        if (rc_loop != KLHR_OK) return rc_loop;
        HIPCHECK(s, hipEventRecord(s->ev1, s->stream));
        s->timing_pending = true;
        s->last_launches = (int)n_iters;
        return KLHR_OK;
    }
    // Chunk the iterations so that a single launch stays short (tens of milliseconds); chain state persists in HBM between
    // launches.  With one wavefront per SIMD a launch lasts as long as its slowest wavefront, and the spread of the
    // wavefronts' summed fit lengths shrinks like 1/sqrt(iterations), so small problems take more iterations per launch:
    // 50 for C*D >= 2.6e6 doubles of state, up to 200 below (funnel-10 x 65 536: 200).  KLHR_CHUNK overrides.
    int64_t chunk_max = 50;
    {
        const double cd = (double)s->C * (double)s->D;
        if (cd > 0 && cd < 2621440.0) chunk_max = (int64_t)(50.0 * 2621440.0 / cd);
        if (chunk_max > 200) chunk_max = 200;
        if (const char *e = getenv("KLHR_CHUNK")) { const long v = atol(e); if (v > 0) chunk_max = v; }
    }
    HIPCHECK(s, hipEventRecord(s->ev0, s->stream));
    int launches = 0;
    for (int64_t done = 0; done < n_iters; done += chunk_max) {
        KParams kp = s->kp;
        kp.phases = phases;
        kp.n_iters = (int)((n_iters - done) < chunk_max ? (n_iters - done) : chunk_max);
        if (record_diag) kp.flags |= FLAG_DIAG;
        kp.draws = record_draws ? s->d_draws + (size_t)done * s->C * s->D : nullptr;
        if (getenv("KLHR_TRACE_LAUNCH")) fprintf(stderr, "[klhr] launch phases=%d n_iters=%d flags=%d chains=%d\n", kp.phases, kp.n_iters, kp.flags, kp.C);
        int rc = launch_iter(s, kp);
        if (rc) return rc;
        launches++;
    }
    HIPCHECK(s, hipEventRecord(s->ev1, s->stream));
    s->timing_pending = true;
    s->last_launches = launches;
    return KLHR_OK;
}

extern "C" int klhr_step(klhr_sampler *s, int64_t n_iters, int record_draws, int record_diag) {
    if (!s || n_iters < 0) return KLHR_ERR_ARG;
    if (n_iters == 0) return KLHR_OK;
    int phases = PH_ALL;
    // profiling aid (tools/phase_insitu.py): run the iteration loop with some phases left out to time the others in situ
    if (const char *dbg = getenv("KLHR_DEBUG_PHASES")) phases = atoi(dbg) & PH_ALL;
    return run_phases(s, phases, n_iters, record_draws, record_diag);
}

extern "C" int klhr_last_step_timing(klhr_sampler *s, double *kernel_ms, int *n_launches) {
    if (!s) return KLHR_ERR_ARG;
    if (s->timing_pending) {
        HIPCHECK(s, hipEventSynchronize(s->ev1));
        float ms = 0;
        HIPCHECK(s, hipEventElapsedTime(&ms, s->ev0, s->ev1));
        s->last_ms = ms;
        s->timing_pending = false;
    }
    if (kernel_ms) *kernel_ms = s->last_ms;
    if (n_launches) *n_launches = s->last_launches;
    return KLHR_OK;
}

// Per-kernel timing (split pipeline): with profiling on, klhr_step runs its chains as ONE group and brackets every kernel
// with HIP events on the handle's stream; klhr_last_kernel_times returns the mean duration (ms) and the launch count of
// [streaming kernel, mode-search kernel, KL-fit kernel] over the most recent klhr_step.  Monolithic kernels: slot 0 only.
extern "C" int klhr_set_kernel_profiling(klhr_sampler *s, int enable) {
    if (!s) return KLHR_ERR_ARG;
    s->profile = enable != 0;
    return KLHR_OK;
}
extern "C" int klhr_last_kernel_times(klhr_sampler *s, double ms3[3], int n3[3]) {
    if (!s || !ms3 || !n3) return KLHR_ERR_ARG;
    for (int k = 0; k < 3; k++) { ms3[k] = 0; n3[k] = 0; }
    if (!s->split || s->last_fused) {
        double ms = 0; int nl = 0;
        int rc = klhr_last_step_timing(s, &ms, &nl);
        if (rc) return rc;
        ms3[0] = nl ? ms / nl : 0; n3[0] = nl;
        return KLHR_OK;
    }
    if (s->pev_used < 4) return KLHR_OK;
    HIPCHECK(s, hipStreamSynchronize(s->stream));
    double tot[3] = {0, 0, 0};
    const int iters = s->pev_used / 4;
    for (int it = 0; it < iters; it++)
        for (int k = 0; k < 3; k++) {
            float ms = 0;
            if (it * 4 + k + 1 >= s->pev_used) break;
            HIPCHECK(s, hipEventElapsedTime(&ms, s->pev[it * 4 + k], s->pev[it * 4 + k + 1]));
            tot[k] += ms; n3[k] += 1;
        }
    tot[1] = 0; n3[1] = 0;          // (one kernel runs both fit stages: slot 2; slot 1 is kept for the ABI's layout)
    for (int k = 0; k < 3; k++) ms3[k] = n3[k] ? tot[k] / n3[k] : 0;
    return KLHR_OK;
}

extern "C" int klhr_get_draws(klhr_sampler *s, double *host_out, int64_t first_iter, int64_t n_iters) {
    if (!s || !host_out || first_iter < 0 || n_iters < 0 || first_iter + n_iters > s->draws_iters) return KLHR_ERR_ARG;
    const size_t per = (size_t)s->C * s->D;
    HIPCHECK(s, hipMemcpyAsync(host_out, s->d_draws + (size_t)first_iter * per, (size_t)n_iters * per * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHECK(s, hipStreamSynchronize(s->stream));
    return KLHR_OK;
}

// ---- seeding -------------------------------------------------------------------------------------
template <class F>
static int launch_by_model(klhr_sampler *s, F f) {
    switch (s->cfg.model_id) {
    case KLHR_MODEL_FUNNEL: f(std::integral_constant<int, MODEL_FUNNEL>()); break;
    case KLHR_MODEL_ILL_NORMAL: f(std::integral_constant<int, MODEL_ILL_NORMAL>()); break;
    case KLHR_MODEL_GLMM_POISSON: f(std::integral_constant<int, MODEL_GLMM_POISSON>()); break;
    case KLHR_MODEL_AR1: f(std::integral_constant<int, MODEL_AR1>()); break;
    case KLHR_MODEL_EARNINGS: f(std::integral_constant<int, MODEL_EARNINGS>()); break;
    default: f(std::integral_constant<int, MODEL_NORMAL>()); break;
    }
    HIPCHECK(s, hipGetLastError());
    return 0;
}

static int upload_seeds(klhr_sampler *s, const uint64_t *seeds) {
    std::vector<uint64_t> st((size_t)s->C * 4);
    for (int c = 0; c < s->C; c++) klhr_seed_state(seeds[c], &st[(size_t)c * 4]);
    HIPCHECK(s, hipMemcpyAsync(s->kp.rng, st.data(), st.size() * 8, hipMemcpyHostToDevice, s->stream));
    HIPCHECK(s, hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int klhr_seed(klhr_sampler *s, const uint64_t *seeds, int theta_given) {
    if (!s || !seeds) return KLHR_ERR_ARG;
    HIPCHECK(s, hipSetDevice(s->cfg.device));
    int rc = reset_state(s);
    if (rc) return rc;
    rc = upload_seeds(s, seeds);
    if (rc) return rc;
    const int bs = 64;
    const dim3 grid((s->C + bs - 1) / bs), block(bs);
    KParams kp = s->kp;
    const int mt = s->cfg.max_init_tries;
    rc = launch_by_model(s, [&](auto M) { hipLaunchKernelGGL(klhr_init_kernel<decltype(M)::value>, grid, block, 0, s->stream, kp, theta_given, mt); });
    if (rc) return rc;
    std::vector<int> st(s->C);
    HIPCHECK(s, hipMemcpyAsync(st.data(), s->kp.status, (size_t)s->C * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHECK(s, hipStreamSynchronize(s->stream));
    s->seeded = true;
    for (int c = 0; c < s->C; c++)
        if (st[c] & 1) { s->err = "klhr_seed: failed to initialize (some chain found no finite log density within max_init_tries)"; return KLHR_ERR_INIT; }
    return KLHR_OK;
}

// ---- pieces --------------------------------------------------------------------------------------
static int h2d(klhr_sampler *s, void *dst, const void *src, size_t bytes) {
    HIPCHECK(s, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s->stream));
    HIPCHECK(s, hipStreamSynchronize(s->stream));   // the caller's buffer may be pageable and reused immediately
    return 0;
}
static int d2h(klhr_sampler *s, void *dst, const void *src, size_t bytes) {
    HIPCHECK(s, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s->stream));
    HIPCHECK(s, hipStreamSynchronize(s->stream));
    return 0;
}

// [C][W] arrays of doubles crossing the ABI: chain-major on the host, tiled on the device for split-pipeline handles
static int up_cw(klhr_sampler *s, double *dst, const double *src, size_t W) {
    if (!s->split) return h2d(s, dst, src, (size_t)s->C * W * 8);
    std::vector<double> t(s->Cp * W, 0.0);
    host_to_tiled(src, t.data(), (size_t)s->C, W, (size_t)s->tw);
    return h2d(s, dst, t.data(), t.size() * 8);
}
static int down_cw(klhr_sampler *s, double *dst, const double *src, size_t W) {
    if (!s->split) return d2h(s, dst, src, (size_t)s->C * W * 8);
    std::vector<double> t(s->Cp * W);
    int rc = d2h(s, t.data(), src, t.size() * 8);
    if (rc) return rc;
    tiled_to_host(t.data(), dst, (size_t)s->C, W, (size_t)s->tw);
    return 0;
}

extern "C" int klhr_random_direction(klhr_sampler *s, double *rho_out) {
    if (!s) return KLHR_ERR_ARG;
    int rc = run_phases(s, PH_DIR, 1, 0, 1);
    if (rc) return rc;
    return rho_out ? down_cw(s, rho_out, s->kp.d_rho, (size_t)s->D) : klhr_sync(s);
}

extern "C" int klhr_fit(klhr_sampler *s, const double *rho, double *eta_out) {
    if (!s || !rho) return KLHR_ERR_ARG;
    int rc = up_cw(s, s->kp.d_rho, rho, (size_t)s->D);
    if (rc) return rc;
    rc = run_phases(s, PH_FIT, 1, 0, 1);
    if (rc) return rc;
    return eta_out ? d2h(s, eta_out, s->kp.d_eta, (size_t)s->C * 4 * 8) : klhr_sync(s);
}

extern "C" int klhr_metropolis(klhr_sampler *s, const double *eta, const double *rho) {
    if (!s || !eta || !rho) return KLHR_ERR_ARG;
    int rc = up_cw(s, s->kp.d_rho, rho, (size_t)s->D);
    if (rc) return rc;
    rc = h2d(s, s->kp.d_eta, eta, (size_t)s->C * 4 * 8);
    if (rc) return rc;
    rc = run_phases(s, PH_MH, 1, 0, 1);
    if (rc) return rc;
    return klhr_sync(s);
}

extern "C" int klhr_set_slice(klhr_sampler *s, double w, double lower, double upper, int64_t max_evals) {
    if (!s) return KLHR_ERR_ARG;
    if (s->cfg.variant != 3) { s->err = "klhr_set_slice: handle was not created with variant 3 (Slice)"; return KLHR_ERR_STATE; }
    if (!(w > 0) || !(lower < upper) || max_evals < 3 || max_evals > 2000000000) { s->err = "klhr_set_slice: need w > 0, lower < upper, 3 <= max_evals"; return KLHR_ERR_ARG; }
    s->kp.sl_w = w; s->kp.sl_lower = lower; s->kp.sl_upper = upper; s->kp.sl_maxiter = (int)max_evals;
    return KLHR_OK;
}

extern "C" int klhr_slice_step(klhr_sampler *s, const double *rho) {
    if (!s || !rho) return KLHR_ERR_ARG;
    if (s->cfg.variant != 3) { s->err = "klhr_slice_step: handle was not created with variant 3 (Slice)"; return KLHR_ERR_STATE; }
    int rc = up_cw(s, s->kp.d_rho, rho, (size_t)s->D);
    if (rc) return rc;
    rc = run_phases(s, PH_MH, 1, 0, 1);
    if (rc) return rc;
    return klhr_sync(s);
}

extern "C" int klhr_adapt(klhr_sampler *s) {
    if (!s) return KLHR_ERR_ARG;
    int rc = run_phases(s, PH_ADAPT, 1, 0, 0);
    if (rc) return rc;
    return klhr_sync(s);
}

static int probe(klhr_sampler *s, int phase, const double *eta_or_t, int eta_cols, const double *rho, double *out, int out_cols) {
    const size_t C = s->C;
    std::vector<double> e4(C * 4, 0.0);
    for (size_t c = 0; c < C; c++)
        for (int k = 0; k < eta_cols; k++) e4[c * 4 + k] = eta_or_t[c * eta_cols + k];
    int rc = up_cw(s, s->kp.d_rho, rho, (size_t)s->D);
    if (rc) return rc;
    rc = h2d(s, s->kp.d_eta, e4.data(), C * 4 * 8);
    if (rc) return rc;
    // KLHR_PROBE_REPEAT (diagnostics only): repeat the probe evaluation n times in one launch to time a bare evaluation
    const char *rep = getenv("KLHR_PROBE_REPEAT");
    rc = run_phases(s, phase, rep ? atoi(rep) : 1, 0, 0);
    if (rc) return rc;
    std::vector<double> dg(C * KLHR_NDIAG);
    rc = d2h(s, dg.data(), s->kp.d_diag, C * KLHR_NDIAG * 8);
    if (rc) return rc;
    for (size_t c = 0; c < C; c++)
        for (int k = 0; k < out_cols; k++) out[c * out_cols + k] = dg[c * KLHR_NDIAG + k];
    return KLHR_OK;
}

extern "C" int klhr_kl_eval(klhr_sampler *s, const double *eta, const double *rho, double *value_out, double *grad_out) {
    if (!s || !eta || !rho || !value_out || !grad_out) return KLHR_ERR_ARG;
    std::vector<double> o((size_t)s->C * 5);
    int rc = probe(s, PH_KL, eta, 4, rho, o.data(), 5);
    if (rc) return rc;
    for (int c = 0; c < s->C; c++) {
        value_out[c] = o[(size_t)c * 5];
        for (int k = 0; k < 4; k++) grad_out[(size_t)c * 4 + k] = o[(size_t)c * 5 + 1 + k];
    }
    return KLHR_OK;
}

extern "C" int klhr_line_eval(klhr_sampler *s, const double *t, const double *rho, double *out2) {
    if (!s || !t || !rho || !out2) return KLHR_ERR_ARG;
    return probe(s, PH_LINE, t, 1, rho, out2, 2);
}

extern "C" int klhr_log_density_gradient(klhr_sampler *s, const double *thetas, int64_t n, double *lp_out, double *grad_out) {
    if (!s || !thetas || !lp_out || n < 0) return KLHR_ERR_ARG;
    if (n == 0) return KLHR_OK;
    HIPCHECK(s, hipSetDevice(s->cfg.device));
    double *d_th = nullptr, *d_lp = nullptr, *d_g = nullptr;
    const size_t D = s->D;
    HIPCHECK(s, hipMalloc((void **)&d_th, (size_t)n * D * 8));
    HIPCHECK(s, hipMalloc((void **)&d_lp, (size_t)n * 8));
    if (grad_out) HIPCHECK(s, hipMalloc((void **)&d_g, (size_t)n * D * 8));
    int rc = h2d(s, d_th, thetas, (size_t)n * D * 8);
    if (!rc) {
        const int bs = 64;
        const dim3 grid((unsigned)((n + bs - 1) / bs)), block(bs);
        ModelData md = s->kp.md;
        rc = launch_by_model(s, [&](auto M) { hipLaunchKernelGGL(klhr_logp_kernel<decltype(M)::value>, grid, block, 0, s->stream, md, d_th, (long long)n, d_lp, d_g); });
    }
    if (!rc) rc = d2h(s, lp_out, d_lp, (size_t)n * 8);
    if (!rc && grad_out) rc = d2h(s, grad_out, d_g, (size_t)n * D * 8);
    (void)hipFree(d_th); (void)hipFree(d_lp); if (d_g) (void)hipFree(d_g);
    return rc;
}

extern "C" int klhr_primitives(klhr_sampler *s, const double *x, const double *eta, int64_t n, double *out12) {
    if (!s || !x || !eta || !out12 || n < 0) return KLHR_ERR_ARG;
    if (n == 0) return KLHR_OK;
    HIPCHECK(s, hipSetDevice(s->cfg.device));
    double *d_x = nullptr, *d_e = nullptr, *d_o = nullptr;
    HIPCHECK(s, hipMalloc((void **)&d_x, (size_t)n * 8));
    HIPCHECK(s, hipMalloc((void **)&d_e, (size_t)n * 32));
    HIPCHECK(s, hipMalloc((void **)&d_o, (size_t)n * 96));
    int rc = h2d(s, d_x, x, (size_t)n * 8);
    if (!rc) rc = h2d(s, d_e, eta, (size_t)n * 32);
    if (!rc) {
        const int bs = 64;
        hipLaunchKernelGGL(klhr_prim_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, s->stream, d_x, d_e, (long long)n, s->cfg.tol, s->cfg.scale_clip, d_o, s->cfg.variant == 2 ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { s->err = hipGetErrorString(e); rc = KLHR_ERR_HIP; }
    }
    if (!rc) rc = d2h(s, out12, d_o, (size_t)n * 96);
    (void)hipFree(d_x); (void)hipFree(d_e); (void)hipFree(d_o);
    return rc;
}

// ---- state access --------------------------------------------------------------------------------
static bool field_info(klhr_sampler *s, int field, void **ptr, size_t *bytes) {
    const size_t C = s->C, D = s->D, nc = s->ncol, J = s->cfg.J;
    KParams &k = s->kp;
    switch (field) {
    case KLHR_F_THETA: *ptr = k.theta; *bytes = C * D * 8; return true;
    case KLHR_F_RNG: *ptr = k.rng; *bytes = C * 32; return true;
    case KLHR_F_MEAN: *ptr = k.mean; *bytes = C * D * 8; return true;
    case KLHR_F_COV: *ptr = k.cov; *bytes = C * D * 8; return true;
    case KLHR_F_EIGVECS: *ptr = k.eigvecs; *bytes = C * D * nc * 8; return true;
    case KLHR_F_EIGVALS: *ptr = k.eigvals; *bytes = C * nc * 8; return true;
    case KLHR_F_OM_M: *ptr = k.om_m; *bytes = C * D * 8; return true;
    case KLHR_F_OM_V: *ptr = k.om_v; *bytes = C * D * 8; return true;
    case KLHR_F_OMD_M: *ptr = k.omd_m; *bytes = C * D * 8; return k.omd_m != nullptr;
    case KLHR_F_OMD_V: *ptr = k.omd_v; *bytes = C * D * 8; return k.omd_v != nullptr;
    case KLHR_F_PCA_V: *ptr = k.pca_v; *bytes = C * D * J * 8; return true;
    case KLHR_F_COUNTERS: *ptr = k.counters; *bytes = C * 32; return true;
    case KLHR_F_ACC: *ptr = k.acc; *bytes = C * 8; return true;
    case KLHR_F_GRAD_EVALS: *ptr = k.grad_evals; *bytes = C * 8; return true;
    case KLHR_F_RHO: *ptr = k.d_rho; *bytes = C * D * 8; return true;
    case KLHR_F_ETA: *ptr = k.d_eta; *bytes = C * 32; return true;
    case KLHR_F_DIAG: *ptr = k.d_diag; *bytes = C * KLHR_NDIAG * 8; return true;
    case KLHR_F_STATUS: *ptr = k.status; *bytes = C * 4; return true;
    case KLHR_F_KSTATE: *ptr = k.kstate; *bytes = C * 32; return true;
    }
    return false;
}

// width W of a tiled [C][W] field (0: the field is chain-major / per-chain scalars in either layout)
static size_t tiled_width(klhr_sampler *s, int field) {
    if (!s->split) return 0;
    const size_t D = s->D;
    switch (field) {
    case KLHR_F_THETA: case KLHR_F_MEAN: case KLHR_F_COV: case KLHR_F_OM_M: case KLHR_F_OM_V: case KLHR_F_OMD_M: case KLHR_F_OMD_V:
    case KLHR_F_RHO: return D;
    case KLHR_F_EIGVECS: return D * s->ncol;
    case KLHR_F_PCA_V: return D * s->cfg.J;
    }
    return 0;
}

extern "C" int klhr_get_state(klhr_sampler *s, int field, void *host_out, size_t bytes) {
    if (!s || !host_out) return KLHR_ERR_ARG;
    void *p; size_t nb;
    if (!field_info(s, field, &p, &nb) || nb != bytes) { s->err = "klhr_get_state: unknown field or size mismatch"; return KLHR_ERR_ARG; }
    if (const size_t W = tiled_width(s, field)) return down_cw(s, (double *)host_out, (const double *)p, W);
    return d2h(s, host_out, p, nb);
}

extern "C" int klhr_set_state(klhr_sampler *s, int field, const void *host_in, size_t bytes) {
    if (!s || !host_in) return KLHR_ERR_ARG;
    void *p; size_t nb;
    if (!field_info(s, field, &p, &nb) || nb != bytes) { s->err = "klhr_set_state: unknown field or size mismatch"; return KLHR_ERR_ARG; }
    int rc;
    if (const size_t W = tiled_width(s, field)) rc = up_cw(s, (double *)p, (const double *)host_in, W);
    else rc = h2d(s, p, host_in, nb);
    if (rc) return rc;
    if (field == KLHR_F_COV) {
        const int bs = 64;
        hipLaunchKernelGGL(klhr_rank_kernel, dim3((s->C + bs - 1) / bs), dim3(bs), 0, s->stream, s->kp);
        HIPCHECK(s, hipGetLastError());
    }
    if (field == KLHR_F_COV || field == KLHR_F_EIGVECS || field == KLHR_F_EIGVALS) {
        rc = refresh_dircache(s);
        if (rc) return rc;
        HIPCHECK(s, hipStreamSynchronize(s->stream));
    }
    if (field == KLHR_F_RNG || field == KLHR_F_THETA) s->seeded = true;
    return KLHR_OK;
}

// ---- MH comparator ---------------------------------------------------------------------------------
extern "C" int klhr_mh_seed(klhr_sampler *s, const uint64_t *seeds) {
    if (!s || !seeds) return KLHR_ERR_ARG;
    HIPCHECK(s, hipSetDevice(s->cfg.device));
    int rc = reset_state(s);
    if (rc) return rc;
    rc = upload_seeds(s, seeds);
    if (rc) return rc;
    const int bs = 64;
    const dim3 grid((s->C + bs - 1) / bs), block(bs);
    KParams kp = s->kp;
    kp.n_iters = 0; kp.draws = nullptr;
    rc = launch_by_model(s, [&](auto M) { hipLaunchKernelGGL(klhr_mh_kernel<decltype(M)::value>, grid, block, 0, s->stream, kp, 0.0, 1); });
    if (rc) return rc;
    s->seeded = true;
    return klhr_sync(s);
}

extern "C" int klhr_mh_step(klhr_sampler *s, double stepsize, int64_t n_iters, int record_draws) {
    if (!s || n_iters < 0) return KLHR_ERR_ARG;
    if (!s->seeded) { s->err = "klhr_mh_step: not seeded"; return KLHR_ERR_STATE; }
    HIPCHECK(s, hipSetDevice(s->cfg.device));
    if (record_draws) { int rc = ensure_draws(s, n_iters); if (rc) return rc; s->draws_iters = n_iters; }
    const int bs = 64;
    const dim3 grid((s->C + bs - 1) / bs), block(bs);
    const int64_t chunk_max = 4096;
    HIPCHECK(s, hipEventRecord(s->ev0, s->stream));
    int launches = 0;
    for (int64_t done = 0; done < n_iters; done += chunk_max) {
        KParams kp = s->kp;
        kp.n_iters = (int)((n_iters - done) < chunk_max ? (n_iters - done) : chunk_max);
        kp.draws = record_draws ? s->d_draws + (size_t)done * s->C * s->D : nullptr;
        int rc = launch_by_model(s, [&](auto M) { hipLaunchKernelGGL(klhr_mh_kernel<decltype(M)::value>, grid, block, 0, s->stream, kp, stepsize, 0); });
        if (rc) return rc;
        launches++;
    }
    HIPCHECK(s, hipEventRecord(s->ev1, s->stream));
    s->timing_pending = true;
    s->last_launches = launches;
    return KLHR_OK;
}

// ---- pooled adaptation (extension) ------------------------------------------------------------------
extern "C" int klhr_pool_stats(klhr_sampler *s, double *out, size_t n_out) {
    if (!s || !out || n_out != (size_t)(1 + 2 * s->D)) return KLHR_ERR_ARG;
    HIPCHECK(s, hipSetDevice(s->cfg.device));
    double *d_o = nullptr;
    HIPCHECK(s, hipMalloc((void **)&d_o, n_out * 8));
    hipLaunchKernelGGL(klhr_pool_kernel, dim3(s->D), dim3(256), 0, s->stream, s->kp, d_o);
    int rc = 0;
    { hipError_t e = hipGetLastError(); if (e != hipSuccess) { s->err = hipGetErrorString(e); rc = KLHR_ERR_HIP; } }
    if (!rc) rc = d2h(s, out, d_o, n_out * 8);
    (void)hipFree(d_o);
    return rc;
}

extern "C" int klhr_pool_apply(klhr_sampler *s, const double *mean, const double *cov) {
    if (!s || !mean || !cov) return KLHR_ERR_ARG;
    HIPCHECK(s, hipSetDevice(s->cfg.device));
    double *d_m = nullptr;
    const size_t D = s->D;
    HIPCHECK(s, hipMalloc((void **)&d_m, 2 * D * 8));
    int rc = h2d(s, d_m, mean, D * 8);
    if (!rc) rc = h2d(s, d_m + D, cov, D * 8);
    if (!rc) {
        const int bs = 64;
        hipLaunchKernelGGL(klhr_pool_apply_kernel, dim3((s->C + bs - 1) / bs), dim3(bs), 0, s->stream, s->kp, d_m, d_m + D);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { s->err = hipGetErrorString(e); rc = KLHR_ERR_HIP; }
    }
    if (!rc) rc = refresh_dircache(s);
    if (!rc) rc = klhr_sync(s);
    (void)hipFree(d_m);
    return rc;
}

// ---- pooled full covariance: device-resident statistics, LDS-tiled MFMA Gram, RCCL all-reduce inside the library ------

extern "C" int klhr_comm_unique_id(void *id_out, size_t bytes) {
    if (!id_out || bytes < sizeof(ncclUniqueId)) return KLHR_ERR_ARG;
    RcclApi &r = rccl();
    if (!r.ok) { g_create_err = "klhr_comm_unique_id: librccl.so could not be loaded"; return KLHR_ERR_HIP; }
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) { g_create_err = "ncclGetUniqueId failed"; return KLHR_ERR_HIP; }
    memset(id_out, 0, bytes);
    memcpy(id_out, &id, sizeof id);
    return KLHR_OK;
}

extern "C" int klhr_comm_init(klhr_sampler *s, const void *id_in, size_t bytes, int rank, int world) {
    if (!s || !id_in || bytes < sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world) return KLHR_ERR_ARG;
    RcclApi &r = rccl();
    if (!r.ok) { s->err = "klhr_comm_init: librccl.so could not be loaded"; return KLHR_ERR_HIP; }
    HIPCHECK(s, hipSetDevice(s->cfg.device));
    if (s->comm) { (void)r.CommDestroy((ncclComm_t)s->comm); s->comm = nullptr; }
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    ncclComm_t c = nullptr;
    const ncclResult_t rc = r.CommInitRank(&c, world, id, rank);
    if (rc != ncclSuccess) { s->err = std::string("ncclCommInitRank failed: ") + r.GetErrorString(rc); return KLHR_ERR_HIP; }
    s->comm = c; s->comm_rank = rank; s->comm_world = world;
    return KLHR_OK;
}
extern "C" int klhr_comm_world(const klhr_sampler *s) { return s ? s->comm_world : KLHR_ERR_ARG; }

static int pool_buffers(klhr_sampler *s) {
    if (s->d_pool) return 0;
    const int D = s->D, Dp = ((D + 1 + 15) / 16) * 16;
    HIPCHECK(s, hipMalloc((void **)&s->d_gram, (size_t)Dp * Dp * 8));
    HIPCHECK(s, hipMalloc((void **)&s->d_pool, ((size_t)1 + D + (size_t)D * D) * 8));
    HIPCHECK(s, hipMemsetAsync(s->d_gram, 0, (size_t)Dp * Dp * 8, s->stream));
    HIPCHECK(s, hipMemsetAsync(s->d_pool, 0, ((size_t)1 + D + (size_t)D * D) * 8, s->stream));
    return 0;
}

extern "C" int klhr_pool_reset(klhr_sampler *s) {
    if (!s) return KLHR_ERR_ARG;
    HIPCHECK(s, hipSetDevice(s->cfg.device));
    int rc = pool_buffers(s);
    if (rc) return rc;
    const int D = s->D, Dp = ((D + 1 + 15) / 16) * 16;
    HIPCHECK(s, hipMemsetAsync(s->d_gram, 0, (size_t)Dp * Dp * 8, s->stream));
    return KLHR_OK;
}

template <int T>
static void launch_gram(klhr_sampler *s, const double *X, long long nrows, long long rps, unsigned slabs, double *G) {
    constexpr int DpS = T * 16 + ((T % 2 == 0) ? 16 : 0);
    const size_t lds = (size_t)GRAM_ROWS * DpS * sizeof(double);
    hipLaunchKernelGGL(klhr_gram_lds_kernel<T>, dim3(slabs), dim3(256), lds, s->stream, X, nrows, s->D, rps, s->kp.status, s->C, G);
}

// += Gram statistics of the recorded draws [first_iter, first_iter + n_iters) into the handle's device-resident buffer;
// asynchronous on the handle's stream, no allocation, no host copy
extern "C" int klhr_pool_accumulate(klhr_sampler *s, int64_t first_iter, int64_t n_iters) {
    if (!s || first_iter < 0 || n_iters < 1 || first_iter + n_iters > s->draws_iters) return KLHR_ERR_ARG;
    HIPCHECK(s, hipSetDevice(s->cfg.device));
    int rc = pool_buffers(s);
    if (rc) return rc;
    const int D = s->D, Dp = ((D + 1 + 15) / 16) * 16, T = Dp / 16;
    const long long nrows = (long long)n_iters * s->C;
    const double *X = s->d_draws + (size_t)first_iter * s->C * D;
    if (T <= 8) {
        // LDS-tiled kernel: X read once.  Slabs: ~4 workgroups per CU, whole chunks of GRAM_ROWS rows
        long long slabs = 1024;
        long long rps = (nrows + slabs - 1) / slabs;
        rps = (rps + GRAM_ROWS - 1) / GRAM_ROWS * GRAM_ROWS;
        slabs = (nrows + rps - 1) / rps;
        switch (T) {
        case 1: launch_gram<1>(s, X, nrows, rps, (unsigned)slabs, s->d_gram); break;
        case 2: launch_gram<2>(s, X, nrows, rps, (unsigned)slabs, s->d_gram); break;
        case 3: launch_gram<3>(s, X, nrows, rps, (unsigned)slabs, s->d_gram); break;
        case 4: launch_gram<4>(s, X, nrows, rps, (unsigned)slabs, s->d_gram); break;
        case 5: launch_gram<5>(s, X, nrows, rps, (unsigned)slabs, s->d_gram); break;
        case 6: launch_gram<6>(s, X, nrows, rps, (unsigned)slabs, s->d_gram); break;
        case 7: launch_gram<7>(s, X, nrows, rps, (unsigned)slabs, s->d_gram); break;
        default: launch_gram<8>(s, X, nrows, rps, (unsigned)slabs, s->d_gram); break;
        }
    } else {
        // D > 127: one wavefront per tile pair and slab (re-reads X once per tile column)
        const int tiles = Dp / 16;
        long long slabs = (nrows + 8191) / 8192;
        if (slabs > 4096) slabs = 4096;
        long long rps = (nrows + slabs - 1) / slabs;
        rps = (rps + 3) / 4 * 4;
        slabs = (nrows + rps - 1) / rps;
        hipLaunchKernelGGL(klhr_gram_mfma_kernel, dim3(tiles * tiles, (unsigned)slabs), dim3(64), 0, s->stream, X, nrows, D, Dp, rps, s->kp.status, s->C, s->d_gram);
    }
    HIPCHECK(s, hipGetLastError());
    return KLHR_OK;
}

// pack the accumulated Gram into [count, sum theta (D), sum theta theta^T (D x D)] and sum it over the ranks of the
// communicator: one ncclAllReduce of 1 + D + D^2 doubles on the handle's stream, in place on the device (RCCL over xGMI).
// Without a communicator (single GPU) only the packing runs.
extern "C" int klhr_pool_allreduce(klhr_sampler *s) {
    if (!s) return KLHR_ERR_ARG;
    HIPCHECK(s, hipSetDevice(s->cfg.device));
    int rc = pool_buffers(s);
    if (rc) return rc;
    const int D = s->D, Dp = ((D + 1 + 15) / 16) * 16, n = 1 + D + D * D;
    hipLaunchKernelGGL(klhr_gram_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->d_gram, D, Dp, s->d_pool);
    HIPCHECK(s, hipGetLastError());
    if (s->comm) {
        // (a one-rank communicator goes through RCCL as well: the path the multi-GPU job takes, exercised by the 1-GPU test)
        RcclApi &r = rccl();
        const ncclResult_t nrc = r.AllReduce(s->d_pool, s->d_pool, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)s->comm, s->stream);
        if (nrc != ncclSuccess) { s->err = std::string("ncclAllReduce failed: ") + r.GetErrorString(nrc); return KLHR_ERR_HIP; }
    }
    return KLHR_OK;
}

extern "C" int klhr_pool_fetch(klhr_sampler *s, double *out, size_t n_out) {
    if (!s || !out || n_out != (size_t)(1 + s->D + (size_t)s->D * s->D)) return KLHR_ERR_ARG;
    if (!s->d_pool) { s->err = "klhr_pool_fetch: nothing accumulated"; return KLHR_ERR_STATE; }
    return d2h(s, out, s->d_pool, n_out * 8);
}

// one-call form (kept from ABI version 1): statistics of a range of recorded draws, this GPU only
extern "C" int klhr_pool_gram(klhr_sampler *s, int64_t first_iter, int64_t n_iters, double *out, size_t n_out) {
    if (!s || !out || first_iter < 0 || n_iters < 1 || first_iter + n_iters > s->draws_iters) return KLHR_ERR_ARG;
    if (n_out != (size_t)(1 + s->D + (size_t)s->D * s->D)) return KLHR_ERR_ARG;
    int rc = klhr_pool_reset(s);
    if (!rc) rc = klhr_pool_accumulate(s, first_iter, n_iters);
    if (rc) return rc;
    const int D = s->D, Dp = ((D + 1 + 15) / 16) * 16, n = 1 + D + D * D;
    hipLaunchKernelGGL(klhr_gram_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->d_gram, D, Dp, s->d_pool);
    HIPCHECK(s, hipGetLastError());
    return klhr_pool_fetch(s, out, n_out);
}

extern "C" int klhr_pool_apply_full(klhr_sampler *s, const double *mean, const double *cov_diag, const double *eigvecs, const double *eigvals) {
    if (!s || !mean || !cov_diag || !eigvecs || !eigvals) return KLHR_ERR_ARG;
    HIPCHECK(s, hipSetDevice(s->cfg.device));
    const size_t D = s->D, nc = s->ncol;
    double *d_m = nullptr;
    const size_t tot = 2 * D + D * nc + nc;
    HIPCHECK(s, hipMalloc((void **)&d_m, tot * 8));
    int rc = h2d(s, d_m, mean, D * 8);
    if (!rc) rc = h2d(s, d_m + D, cov_diag, D * 8);
    if (!rc) rc = h2d(s, d_m + 2 * D, eigvecs, D * nc * 8);
    if (!rc) rc = h2d(s, d_m + 2 * D + D * nc, eigvals, nc * 8);
    if (!rc) {
        const int bs = 64;
        hipLaunchKernelGGL(klhr_pool_apply_full_kernel, dim3((s->C + bs - 1) / bs), dim3(bs), 0, s->stream, s->kp, d_m, d_m + D, d_m + 2 * D,
                           d_m + 2 * D + D * nc);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { s->err = hipGetErrorString(e); rc = KLHR_ERR_HIP; }
    }
    if (!rc) rc = refresh_dircache(s);
    if (!rc) rc = klhr_sync(s);
    (void)hipFree(d_m);
    return rc;
}
