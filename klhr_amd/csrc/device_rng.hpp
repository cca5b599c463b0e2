// device_rng.hpp — NumPy-compatible PCG64 + 256-layer ziggurat normal on gfx950.
//
// The reference draws every random number from np.random.default_rng(seed) (mcmc.py:9-15;
// klhr_sinh.py:90,184,191,212,246,254).  To match it chain-for-chain the device generator must be
// NumPy's: PCG64 (128-bit LCG, XSL-RR output) and the ziggurat standard normal of
// numpy/random/src/distributions/distributions.c.  rocRAND cannot reproduce this stream.
// One generator per chain; all 8 lanes of a chain's octet hold an identical copy in registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KLHR_ZIG_QUAL static __device__ const
#include "zig_tables.h"

// NumPy's generator code is plain C without FMA contraction: every operation here rounds on its own
#pragma clang fp contract(off)
namespace klhr {

typedef unsigned __int128 u128;

struct Pcg64 {
    uint64_t sh, sl, ih, il;   // state hi/lo, increment hi/lo
};

__device__ __forceinline__ uint64_t pcg_next64(Pcg64 &r) {
    const u128 MULT = (((u128)0x2360ED051FC65DA4ULL) << 64) | (u128)0x4385DF649FCCF645ULL;
    u128 st = (((u128)r.sh) << 64) | r.sl;
    u128 inc = (((u128)r.ih) << 64) | r.il;
    st = st * MULT + inc;
    r.sh = (uint64_t)(st >> 64);
    r.sl = (uint64_t)st;
    uint64_t x = r.sh ^ r.sl;
    unsigned rot = (unsigned)(r.sh >> 58);
    return (x >> rot) | (x << ((-rot) & 63));
}

__device__ __forceinline__ double pcg_next_double(Pcg64 &r) {
    return (double)(pcg_next64(r) >> 11) * (1.0 / 9007199254740992.0);
}

// Generator.standard_normal()
__device__ __forceinline__ double rng_std_normal_body(Pcg64 &g) {
    const double ZIG_R = 3.6541528853610088, ZIG_INV_R = 0.27366123732975828;
    for (;;) {
        uint64_t r = pcg_next64(g);
        int idx = (int)(r & 0xff);
        r >>= 8;
        int sign = (int)(r & 1);
        uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
        double x = (double)rabs * KLHR_ZIG_WI[idx];
        if (sign) x = -x;
        if (rabs < KLHR_ZIG_KI[idx]) return x;          // ~98.8 % of draws
        if (idx == 0) {
            for (;;) {
                double xx = -ZIG_INV_R * log1p(-pcg_next_double(g));
                double yy = -log1p(-pcg_next_double(g));
                if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? -(ZIG_R + xx) : ZIG_R + xx;
            }
        } else {
            if (((KLHR_ZIG_FI[idx - 1] - KLHR_ZIG_FI[idx]) * pcg_next_double(g) + KLHR_ZIG_FI[idx]) < exp(-0.5 * x * x))
                return x;
        }
    }
}

__device__ inline double rng_std_normal(Pcg64 &g) { return rng_std_normal_body(g); }

// Out-of-line by-value form for the lane-per-chain kernel: one copy of the ziggurat loop (tail + wedge branches) per
// kernel instead of one per call site keeps those branches out of the register allocation around the seven call sites
// (+4 % on funnel-10; the octet kernel, which draws D normals per lane and iteration, is 3-10 % faster inlined and
// uses this form only for the fit's x0 and the MH proposal: the exit blocks of those two inlined loops are where the
// register allocator misplaced AGPR copies, see tools/exec_lint.py).
struct NormOut { double z; Pcg64 g; };
__device__ __attribute__((noinline)) NormOut rng_std_normal_ool(Pcg64 g) {
    NormOut o;
    o.z = rng_std_normal_body(g);
    o.g = g;
    return o;
}
__device__ __forceinline__ double rng_std_normal_call(Pcg64 &g) {
    const NormOut o = rng_std_normal_ool(g);
    g = o.g;
    return o.z;
}

// Direction draw of the octet kernel: D normals per iteration, one after the other (the stream is sequential).  The
// two table look-ups of the fast path (98.8 % of the draws) come from LDS copies of WI / KI (4 KB per wavefront) instead of
// the vector cache — a wavefront alone on its SIMD otherwise waits out two dependent global-load latencies per normal
// (23 % of an iteration at D = 100).  A draw that fails the fast test continues out of line, from the global tables,
// with exactly the steps of rng_std_normal_body: same values, same stream position.
#define KLHR_ZTAB_DOUBLES 512
__device__ __attribute__((noinline)) NormOut rng_std_normal_slow(Pcg64 g, int idx, uint64_t rabs, double x) {
    const double ZIG_R = 3.6541528853610088, ZIG_INV_R = 0.27366123732975828;
    NormOut o;
    for (;;) {
        if (idx == 0) {
            for (;;) {
                double xx = -ZIG_INV_R * log1p(-pcg_next_double(g));
                double yy = -log1p(-pcg_next_double(g));
                if (yy + yy > xx * xx) { o.z = ((rabs >> 8) & 1) ? -(ZIG_R + xx) : ZIG_R + xx; o.g = g; return o; }
            }
        } else if (((KLHR_ZIG_FI[idx - 1] - KLHR_ZIG_FI[idx]) * pcg_next_double(g) + KLHR_ZIG_FI[idx]) < exp(-0.5 * x * x)) {
            o.z = x; o.g = g; return o;
        }
        uint64_t r = pcg_next64(g);
        idx = (int)(r & 0xff);
        r >>= 8;
        const int sign = (int)(r & 1);
        rabs = (r >> 1) & 0x000fffffffffffffULL;
        x = (double)rabs * KLHR_ZIG_WI[idx];
        if (sign) x = -x;
        if (rabs < KLHR_ZIG_KI[idx]) { o.z = x; o.g = g; return o; }
    }
}
__device__ __forceinline__ double rng_std_normal_tab(Pcg64 &g, const double *wi_l, const uint64_t *ki_l) {
    uint64_t r = pcg_next64(g);
    const int idx = (int)(r & 0xff);
    r >>= 8;
    const int sign = (int)(r & 1);
    const uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
    double x = (double)rabs * wi_l[idx];
    if (sign) x = -x;
    if (rabs < ki_l[idx]) return x;
    const NormOut o = rng_std_normal_slow(g, idx, rabs, x);
    g = o.g;
    return o.z;
}

// D standard normals per lane into a strided LDS / memory column (dst[k * stride]), for wavefronts whose lanes own
// DIFFERENT chains (lane-per-chain kernels).  The ziggurat's fast test fails for 1.2 % of the draws; executed naively,
// one such lane makes the whole wavefront walk the slow branch (tail / wedge, ~600 instructions) — at 64 lanes that is
// every second draw.  Here a lane whose draw fails the fast test PARKS (idx, rabs, x) and waits, the others keep drawing,
// and the slow branch runs once for all parked lanes when every unfinished lane is parked or every KLHR_ZIG_BATCH trips:
// ~D/KLHR_ZIG_BATCH executions of the slow branch per column instead of ~D/2.  Each lane still consumes its own stream in
// order: same values, same stream position as Generator.standard_normal() D times.
#ifndef KLHR_ZIG_BATCH
#define KLHR_ZIG_BATCH 12
#endif
__device__ inline void rng_std_normal_column(Pcg64 &g, double *dst, int stride, int D, const double *wi_l, const uint64_t *ki_l) {
    int k = 0, idx = 0, trip = 0;
    uint64_t rabs = 0;
    double x = 0.0;
    bool parked = false;
    for (;;) {
        const bool todo = k < D;
        if (todo && !parked) {
            uint64_t r = pcg_next64(g);
            idx = (int)(r & 0xff);
            r >>= 8;
            const int sign = (int)(r & 1);
            rabs = (r >> 1) & 0x000fffffffffffffULL;
            x = (double)rabs * wi_l[idx];
            if (sign) x = -x;
            if (rabs < ki_l[idx]) { dst[(long long)k * stride] = x; k++; }
            else parked = true;
        }
        ++trip;
        const bool any_parked = __any(parked);
        const bool all_stuck = !__any(k < D && !parked);
        if (any_parked && (all_stuck || trip % KLHR_ZIG_BATCH == 0)) {
            if (parked) {
                const NormOut o = rng_std_normal_slow(g, idx, rabs, x);
                g = o.g;
                dst[(long long)k * stride] = o.z;
                k++;
                parked = false;
            }
        }
        if (!__any(k < D)) break;
    }
}

// D standard normals of ONE chain drawn by the four lanes of its DPP quad (quad mode of the streaming kernel: lane = 4 * chain +
// q), into dst[k * stride].  PCG64 is a 128-bit LCG: lane q holds the state q + 1 steps ahead of the chain's and advances four
// steps at a time (state * M^4 + inc * (1 + M + M^2 + M^3)), so one round yields the chain's next four raw outputs and — when all
// four pass the ziggurat's fast test, 95 % of the rounds — its next four normals.  When lane f fails, the lanes below it keep
// their normals, the quad takes lane f's state and draw, PARKS (as in rng_std_normal_column) and finishes that normal with
// rng_std_normal_slow — consuming however many further outputs that takes, sequentially — before the four lanes fan out again
// from the state it leaves.  Same values in the same order as Generator.standard_normal() D times, and the generator ends at
// the same position.  g: the chain's generator, identical in the quad's four lanes on entry and on return.
#ifndef KLHR_ZIGQ_BATCH
#define KLHR_ZIGQ_BATCH 4
#endif
__device__ __forceinline__ uint64_t quad_lane_u64(uint64_t v, int src_lane) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)v);
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)(v >> 32));
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ inline void rng_std_normal_quad(Pcg64 &g, double *dst, int stride, int D, int lane, const double *wi_l, const uint64_t *ki_l) {
    const u128 M1 = (((u128)0x2360ED051FC65DA4ULL) << 64) | (u128)0x4385DF649FCCF645ULL, G1 = 1;
    const u128 M2 = (((u128)0x17BCE35BDF69743CULL) << 64) | (u128)0x529ED9EB20E0AE99ULL, G2 = (((u128)0x2360ED051FC65DA4ULL) << 64) | (u128)0x4385DF649FCCF646ULL;
    const u128 M3 = (((u128)0x25F041404BD80E82ULL) << 64) | (u128)0xEB5AE837ED42153DULL, G3 = (((u128)0x3B1DD060FF2FD1E0ULL) << 64) | (u128)0x9624B94FC0ADA4DFULL;
    const u128 M4 = (((u128)0xF4DD417327DB7A9BULL) << 64) | (u128)0xD194DFBE42D45771ULL, G4 = (((u128)0x610E11A14B07E063ULL) << 64) | (u128)0x817FA187ADEFBA1CULL;
    const int q = lane & 3, qbase = lane & ~3;
    const u128 inc = (((u128)g.ih) << 64) | g.il;
    const u128 inc_g4 = inc * G4;
    const u128 Mq = q == 0 ? M1 : (q == 1 ? M2 : (q == 2 ? M3 : M4));
    const u128 inc_gq = inc * (q == 0 ? G1 : (q == 1 ? G2 : (q == 2 ? G3 : G4)));
    uint64_t bh = g.sh, bl = g.sl;          // the chain's state (all four lanes)
    uint64_t th, tl;                        // the lane's state: q + 1 steps past (bh, bl); parked: the failed draw's state (all four lanes)
    { const u128 t = ((((u128)bh) << 64) | bl) * Mq + inc_gq; th = (uint64_t)(t >> 64); tl = (uint64_t)t; }
    int k = 0, trip = 0, idx = 0;
    uint64_t rabs = 0;
    double x = 0.0;
    bool parked = false;
    for (;;) {
        if (k < D && !parked) {
            const uint64_t xo = th ^ tl;
            const unsigned rot = (unsigned)(th >> 58);
            uint64_t r = (xo >> rot) | (xo << ((-rot) & 63));
            idx = (int)(r & 0xff);
            const double wv = wi_l[idx];
            const uint64_t kv = ki_l[idx];
            // the stride-4 successor, computed while the two table reads are in flight (used unless a lane of the quad fails)
            uint64_t nh, nl;
            { const u128 t = ((((u128)th) << 64) | tl) * M4 + inc_g4; nh = (uint64_t)(t >> 64); nl = (uint64_t)t; }
            asm volatile("" : "+v"(nh), "+v"(nl));
            r >>= 8;
            const int sign = (int)(r & 1);
            rabs = (r >> 1) & 0x000fffffffffffffULL;
            x = (double)rabs * wv;
            if (sign) x = -x;
            const int rem = D - k;
            const bool fail = q < rem && !(rabs < kv);
            const unsigned fm = (unsigned)(__ballot(fail) >> qbase) & 0xfu;          // the quad's lanes whose draw needs the slow branch
            const int f = fm ? __ffs((int)fm) - 1 : 4;
            if (q < f && q < rem) dst[(long long)(k + q) * stride] = x;
            if (fm == 0 && rem > 4) {
                k += 4;
                th = nh; tl = nl;
            } else {
                // lane f's draw goes to the slow branch — or, with no failure, the column ends in this round with lane rem - 1's state
                const int src = qbase + (fm ? f : rem - 1);
                th = quad_lane_u64(th, src); tl = quad_lane_u64(tl, src);
                if (fm) {
                    k += f;
                    idx = __builtin_amdgcn_ds_bpermute(src << 2, idx);
                    rabs = quad_lane_u64(rabs, src);
                    x = __longlong_as_double((long long)quad_lane_u64((uint64_t)__double_as_longlong(x), src));
                    parked = true;
                } else {
                    k = D; bh = th; bl = tl;
                }
            }
        }
        ++trip;
        const bool any_parked = __any(parked);
        const bool all_stuck = !__any(k < D && !parked);
        if (any_parked && (all_stuck || trip % KLHR_ZIGQ_BATCH == 0)) {
            if (parked) {
                Pcg64 t; t.sh = th; t.sl = tl; t.ih = g.ih; t.il = g.il;
                const NormOut o = rng_std_normal_slow(t, idx, rabs, x);
                dst[(long long)k * stride] = o.z;
                k++;
                bh = o.g.sh; bl = o.g.sl;
                const u128 u = ((((u128)bh) << 64) | bl) * Mq + inc_gq;
                th = (uint64_t)(u >> 64); tl = (uint64_t)u;
                parked = false;
            }
        }
        if (!__any(k < D)) break;
    }
    g.sh = bh; g.sl = bl;
}

// Generator.uniform() == 0.0 + (1.0 - 0.0) * next_double
__device__ __forceinline__ double rng_uniform(Pcg64 &g) { return pcg_next_double(g); }

// Generator.uniform(low, high): low + (high - low) * next_double, unfused (slice.py:95,130)
__device__ __forceinline__ double rng_uniform_range(Pcg64 &g, double low, double high) {
    return add_rn(low, mul_rn(high - low, pcg_next_double(g)));
}

// Generator.exponential(): NumPy's 256-layer ziggurat for the standard exponential (distributions.c:
// random_standard_exponential / standard_exponential_unlikely), used by the reference's slice.py:93
__device__ inline double rng_std_exponential(Pcg64 &g) {
    const double ZIG_EXP_R = 7.69711747013104972;
    for (;;) {
        uint64_t ri = pcg_next64(g);
        ri >>= 3;
        const int idx = (int)(ri & 0xff);
        ri >>= 8;
        const double x = mul_rn((double)ri, KLHR_ZIG_WE[idx]);
        if (ri < KLHR_ZIG_KE[idx]) return x;            // ~98.9 % of draws
        if (idx == 0) return ZIG_EXP_R - log1p(-pcg_next_double(g));
        if (add_rn(mul_rn(KLHR_ZIG_FE[idx - 1] - KLHR_ZIG_FE[idx], pcg_next_double(g)), KLHR_ZIG_FE[idx]) < exp(-x)) return x;
    }
}

}  // namespace klhr
#pragma clang fp contract(fast)
