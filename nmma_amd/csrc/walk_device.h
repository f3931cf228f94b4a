// walk_device.h -- device functions of the lock-step ensemble walk and of the constraint programs, shared by walk_kernels.hip (the
// per-step kernels) and em_kernels.hip (the MCMC step fused into the likelihood kernel's epilogue: em_logl<..., WALKF>).  Both units
// are built with -ffp-contract=off, so a chain is bit-identical whichever kernel advances it.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/nmma_hip.h"

namespace nmma {

struct WalkSpec {
    nmma_walk_prior p[NMMA_WALK_MAX_DIM];
    int32_t ndim;
};

__device__ __forceinline__ uint64_t walk_mix64(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// draw k of step `step` of the chain with key `key`: sampler.py:counter_uniforms, bit for bit
__device__ __forceinline__ double walk_uniform(const uint64_t key, const uint64_t step, const uint64_t k) {
    const uint64_t G = 0x9E3779B97F4A7C15ull;
    uint64_t x = walk_mix64(walk_mix64(key * G + G) ^ (step * 0xD1342543DE82EF95ull + k * 0xA0761D6478BD642Full + G));
    x = walk_mix64(x);
    return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double floored_mod(const double x, const double m) {       // np.mod for m > 0
    const double r = fmod(x, m);
    return r < 0.0 ? r + m : r;
}
// bilby/core/prior/analytical.py: rescale(val) of the analytic priors
__device__ inline double walk_rescale(const nmma_walk_prior& p, const double u) {
    switch (p.kind) {
        case NMMA_PRIOR_UNIFORM: return p.a + u * (p.b - p.a);
        case NMMA_PRIOR_SINE: { const double norm = 1.0 / (cos(p.a) - cos(p.b)); return acos(cos(p.a) - u / norm); }
        case NMMA_PRIOR_COSINE: { const double norm = 1.0 / (sin(p.b) - sin(p.a)); return asin(u / norm + sin(p.a)); }
        case NMMA_PRIOR_POWERLAW:
            if (p.alpha == -1.0) return p.a * exp(u * log(p.b / p.a));
            return pow(pow(p.a, 1.0 + p.alpha) + u * (pow(p.b, 1.0 + p.alpha) - pow(p.a, 1.0 + p.alpha)), 1.0 / (1.0 + p.alpha));
        case NMMA_PRIOR_GAUSSIAN: return p.a + erfinv(2.0 * u - 1.0) * 1.4142135623730951 * p.b;      // mu, sigma
        case NMMA_PRIOR_TRUNC_GAUSSIAN: return erfinv(2.0 * u * p.alpha + p.c) * 1.4142135623730951 * p.b + p.a;   // bilby TruncatedGaussian.rescale
        case NMMA_PRIOR_LOGNORMAL: return exp(p.a + sqrt(2.0 * p.b * p.b) * erfinv(2.0 * u - 1.0));                // bilby LogNormal.rescale
        case NMMA_PRIOR_HALF_GAUSSIAN: return erfinv(u) * 1.4142135623730951 * p.b;                                // bilby HalfGaussian.rescale
        default: return p.a;                                                                        // NMMA_PRIOR_DELTA: peak
    }
}

// A GROUP of T = 8 / 16 / 32 lanes per chain (the smallest that holds the dimensions: lane d owns dimension d), 256 / T chains per
// workgroup.  One thread per chain made this kernel a 10 us chain of dependent loads and a serial loop over the dimensions around a
// 29 us likelihood launch; with a lane per dimension the loads of a row are one coalesced access and the prior transforms run side by
// side (rocprofv3: 10.1 -> see DESIGN.md section 6).  Every lane draws the chain's seven uniforms itself (integer hashing, no traffic).
// The prior table is staged from the kernel arguments into LDS so that lanes can index it by their dimension.
__device__ __forceinline__ int walk_group(const int D) { return D <= 8 ? 8 : D <= 16 ? 16 : 32; }

static_assert(sizeof(nmma_walk_prior) == 40 && NMMA_WALK_MAX_DIM * 10 <= 512, "at most two dwords of the table per thread of the workgroup");
__device__ __forceinline__ void walk_stage_spec(const WalkSpec& S, nmma_walk_prior* sp) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&S.p[0]);          // (the kernel-argument segment, read per thread)
    for (int j = threadIdx.x; j < S.ndim * 10; j += 256) reinterpret_cast<uint32_t*>(sp)[j] = src[j];
    __syncthreads();
}

// Constraint priors: the postfix program of include/nmma_hip.h (nmma_con_op) on one row of theta.  Uniform control flow (every
// lane runs the same program); the stack is a handful of doubles.
__device__ inline bool con_row_ok(const nmma_con_op* __restrict__ ops, const int n_ops, const double* __restrict__ row) {
    double st[NMMA_CON_MAX_STACK];
    int sp = 0;
    bool ok = true;
    for (int i = 0; i < n_ops; ++i) {
        const nmma_con_op o = ops[i];
        switch (o.op) {
            case NMMA_CON_PUSH_COL: st[sp++] = row[o.col]; break;
            case NMMA_CON_PUSH_CONST: st[sp++] = o.value; break;
            case NMMA_CON_ADD: st[sp - 2] = st[sp - 2] + st[sp - 1]; --sp; break;
            case NMMA_CON_SUB: st[sp - 2] = st[sp - 2] - st[sp - 1]; --sp; break;
            case NMMA_CON_MUL: st[sp - 2] = st[sp - 2] * st[sp - 1]; --sp; break;
            case NMMA_CON_DIV: st[sp - 2] = st[sp - 2] / st[sp - 1]; --sp; break;
            case NMMA_CON_POW: st[sp - 2] = pow(st[sp - 2], st[sp - 1]); --sp; break;
            case NMMA_CON_MIN: st[sp - 2] = fmin(st[sp - 2], st[sp - 1]); --sp; break;
            case NMMA_CON_MAX: st[sp - 2] = fmax(st[sp - 2], st[sp - 1]); --sp; break;
            case NMMA_CON_NEG: st[sp - 1] = -st[sp - 1]; break;
            case NMMA_CON_ABS: st[sp - 1] = fabs(st[sp - 1]); break;
            case NMMA_CON_SQRT: st[sp - 1] = sqrt(st[sp - 1]); break;
            case NMMA_CON_LOG10: st[sp - 1] = log10(st[sp - 1]); break;
            case NMMA_CON_LOG: st[sp - 1] = log(st[sp - 1]); break;
            case NMMA_CON_EXP: st[sp - 1] = exp(st[sp - 1]); break;
            case NMMA_CON_SIN: st[sp - 1] = sin(st[sp - 1]); break;
            case NMMA_CON_COS: st[sp - 1] = cos(st[sp - 1]); break;
            case NMMA_CON_ACOS: st[sp - 1] = acos(st[sp - 1]); break;
            case NMMA_CON_ASIN: st[sp - 1] = asin(st[sp - 1]); break;
            case NMMA_CON_SIGN: { const double x = st[sp - 1]; st[sp - 1] = x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : x); } break;
            case NMMA_CON_CHECK_GT: ok = ok && (st[sp - 1] > o.value); break;
            default: ok = ok && (st[sp - 1] < o.value); --sp; break;          // NMMA_CON_CHECK_LT
        }
    }
    return ok;
}

// con_row_ok with its evaluation stack in LDS: st[level * stride] (one stack per chain of a round, the chains' stacks interleaved).
// For the likelihood kernel's fused MCMC step: as a private array the 16-deep fp64 stack and its dynamic indexing cost 270 VGPRs, which
// the kernel's budget of 128 cannot hold.  Same operations in the same order as con_row_ok: the same verdict, bit for bit.
__device__ inline bool con_row_ok_lds(const nmma_con_op* __restrict__ ops, const int n_ops, const double* __restrict__ row, double* st_generic,
                                      const int stride) {
    typedef __attribute__((address_space(3))) double* lds_dp;
    const lds_dp st = (lds_dp)st_generic;
    int sp = 0;
    bool ok = true;
    for (int i = 0; i < n_ops; ++i) {
        const nmma_con_op o = ops[i];
        const int t1 = (sp - 1) * stride, t2 = (sp - 2) * stride;
        switch (o.op) {
            case NMMA_CON_PUSH_COL: st[sp * stride] = row[o.col]; ++sp; break;
            case NMMA_CON_PUSH_CONST: st[sp * stride] = o.value; ++sp; break;
            case NMMA_CON_ADD: st[t2] = st[t2] + st[t1]; --sp; break;
            case NMMA_CON_SUB: st[t2] = st[t2] - st[t1]; --sp; break;
            case NMMA_CON_MUL: st[t2] = st[t2] * st[t1]; --sp; break;
            case NMMA_CON_DIV: st[t2] = st[t2] / st[t1]; --sp; break;
            case NMMA_CON_POW: st[t2] = pow(st[t2], st[t1]); --sp; break;
            case NMMA_CON_MIN: st[t2] = fmin(st[t2], st[t1]); --sp; break;
            case NMMA_CON_MAX: st[t2] = fmax(st[t2], st[t1]); --sp; break;
            case NMMA_CON_NEG: st[t1] = -st[t1]; break;
            case NMMA_CON_ABS: st[t1] = fabs(st[t1]); break;
            case NMMA_CON_SQRT: st[t1] = sqrt(st[t1]); break;
            case NMMA_CON_LOG10: st[t1] = log10(st[t1]); break;
            case NMMA_CON_LOG: st[t1] = log(st[t1]); break;
            case NMMA_CON_EXP: st[t1] = exp(st[t1]); break;
            case NMMA_CON_SIN: st[t1] = sin(st[t1]); break;
            case NMMA_CON_COS: st[t1] = cos(st[t1]); break;
            case NMMA_CON_ACOS: st[t1] = acos(st[t1]); break;
            case NMMA_CON_ASIN: st[t1] = asin(st[t1]); break;
            case NMMA_CON_SIGN: { const double x = st[t1]; st[t1] = x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : x); } break;
            case NMMA_CON_CHECK_GT: ok = ok && (st[t1] > o.value); break;
            default: ok = ok && (st[t1] < o.value); --sp; break;          // NMMA_CON_CHECK_LT
        }
    }
    return ok;
}

// proposal in the unit cube (differential evolution between two other live points), boundary conditions, inside-the-cube flag, prior
// transform.  A proposal outside the cube keeps the chain's current point in `theta` (the lock-step likelihood launch evaluates every
// chain; the accept kernel ignores that row).
__device__ __forceinline__ void walk_propose_one(const nmma_walk_prior* sp, const int D, const int T, const long c, const int lane,
                                                 const double* __restrict__ live, const long n_live, const double* u, const double* v,
                                                 const uint64_t* __restrict__ key, const uint64_t step, double* prop, double* theta,
                                                 int32_t* inside) {
    const uint64_t kc = key[c];
    double r[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) r[k] = walk_uniform(kc, step, (uint64_t)k);
    long i = (long)(r[0] * (double)n_live);
    i = i > n_live - 1 ? n_live - 1 : i;
    long jj = (long)(r[1] * (double)(n_live - 1));
    jj = jj > n_live - 2 ? n_live - 2 : jj;
    const long j = (i + 1 + jj) % n_live;                                         // a different live point
    const double gamma = r[2] < 0.5 ? 1.0 : 2.38 / sqrt(2.0 * (double)D) * (-0.25 * log(r[3] * r[4] * r[5] * r[6]));   // Gamma(4, 1/4)
    int in = 1;
    double x = 0.0;
    if (lane < D) {
        x = u[c * D + lane] + gamma * (live[j * D + lane] - live[i * D + lane]);
        const int32_t bc = sp[lane].boundary;
        if (bc == NMMA_BOUNDARY_PERIODIC) x = floored_mod(x, 1.0);
        else if (bc == NMMA_BOUNDARY_REFLECTIVE) { const double q = floored_mod(x, 2.0); x = q > 1.0 ? 2.0 - q : q; }
        prop[c * D + lane] = x;
        in = (x >= 0.0) && (x <= 1.0);
    }
    for (int m = T >> 1; m > 0; m >>= 1) in &= __shfl_xor(in, m, 64);           // (groups are aligned powers of two: the exchange stays inside)
    if (lane == 0) inside[c] = in;
    if (lane < D) theta[c * D + lane] = in ? walk_rescale(sp[lane], x) : v[c * D + lane];
}

// accept when the proposal was inside the cube and its likelihood beats the chain's bound (dynesty: logl > loglstar)
__device__ __forceinline__ void walk_accept_one(const int D, const long c, const int lane, const double* prop, const double* theta,
                                                const int32_t* inside, const double lp_in, const double* __restrict__ loglstar,
                                                double* u, double* v, double* __restrict__ logl, int32_t* __restrict__ counts,
                                                const int32_t* __restrict__ n_steps, const uint64_t step,
                                                const nmma_con_op* __restrict__ con_ops = nullptr, const int n_con_ops = 0) {
    if (n_steps != nullptr && step > (uint64_t)n_steps[c]) return;       // this chain's walk is over (walk lengths may differ per chain)
    const int in = inside[c];
    double lp = lp_in;
    // a Constraint prior the proposal violates: the reference's likelihood returns the floor for it (core/base.py:77-82)
    if (n_con_ops > 0 && in && !con_row_ok(con_ops, n_con_ops, theta + c * D)) lp = NMMA_LOGL_FLOOR;
    const bool acc = in && lp > loglstar[c];
    if (acc && lane < D) { u[c * D + lane] = prop[c * D + lane]; v[c * D + lane] = theta[c * D + lane]; }
    if (lane == 0) {
        int32_t* cnt = counts + 4 * c;                      // {accept, reject, nfail, ncall}
        if (!in) cnt[2] += 1;
        else {
            cnt[3] += 1;
            if (acc) { logl[c] = lp; cnt[0] += 1; }
            else cnt[1] += 1;
        }
    }
}

// The same step in two phases, so that everything that does not depend on the proposal's log-likelihood -- the chain's seven
// uniforms, the two live points of the differential-evolution move, the chain's current state -- is in flight BEFORE the
// likelihood is known: walk_step_pre issues the loads, walk_step_post (with log L) decides, moves and proposes.  The arithmetic
// is walk_accept_one followed by walk_propose_one, operation for operation (x = u_after_accept + gamma (live_j - live_i), the same
// boundary folds, the same transform), so a chain is bit-identical whichever form advanced it.  Used by the likelihood kernel's
// fused epilogue (em_logl<..., WALKF>), whose first likelihood wave runs the pre-phase while it waits for the tile's last tasks.
struct WalkPre {
    double gamma, li, lj, uu, vv, pp, th, lstar;
    int32_t in0, active, cnt0, cnt1, cnt2, cnt3;
};

// (three sub-steps, so that a caller with several rounds of chains can issue ALL rounds' loads of a sub-step back to back: the key
//  and state loads, then the hashes and the live-point gathers they address, then the arithmetic on the uniforms)
struct WalkPreKey { uint64_t kc; };
__device__ __forceinline__ void walk_step_pre_a(const int D, const long c, const int lane, const uint64_t* __restrict__ key, const double* u,
                                                const double* v, const double* prop, const double* theta, const int32_t* inside,
                                                const double* __restrict__ loglstar, const int32_t* __restrict__ counts,
                                                const int32_t* __restrict__ n_steps, const uint64_t step, WalkPre& w, WalkPreKey& k) {
    k.kc = key[c];
    w.in0 = inside[c];
    w.lstar = loglstar[c];
    w.active = !(n_steps != nullptr && step > (uint64_t)n_steps[c]);
    w.cnt0 = w.cnt1 = w.cnt2 = w.cnt3 = 0;
    if (lane == 0) { const int32_t* cnt = counts + 4 * c; w.cnt0 = cnt[0]; w.cnt1 = cnt[1]; w.cnt2 = cnt[2]; w.cnt3 = cnt[3]; }
    w.uu = w.vv = w.pp = w.th = 0.0;
    if (lane < D) { w.uu = u[c * D + lane]; w.vv = v[c * D + lane]; w.pp = prop[c * D + lane]; w.th = theta[c * D + lane]; }
}
// The chain's seven uniforms, ONE per lane of its group (lane k draws number k) and handed round by lane shuffles: the fused
// epilogue runs on a wave whose every vector instruction takes an issue slot from the MFMA stream of its SIMD, and seven 64-bit
// hashes per lane were most of this phase's instructions.  Same numbers as walk_uniform(key, step, k) anywhere else.
__device__ __forceinline__ void walk_step_pre_b(const int D, const int lane, const double* __restrict__ live, const long n_live,
                                                const uint64_t rng_step, const WalkPreKey& k, WalkPre& w, double (&r)[7]) {
    const double mine = walk_uniform(k.kc, rng_step, (uint64_t)(lane & 7));
    const int g0 = (int)(threadIdx.x & 63) & ~7;              // (groups of at least 8 lanes: T >= 8)
#pragma unroll
    for (int q = 0; q < 7; ++q) r[q] = __shfl(mine, g0 + q, 64);
    long i = (long)(r[0] * (double)n_live);
    i = i > n_live - 1 ? n_live - 1 : i;
    long jj = (long)(r[1] * (double)(n_live - 1));
    jj = jj > n_live - 2 ? n_live - 2 : jj;
    const long j = (i + 1 + jj) % n_live;                                         // a different live point
    w.li = w.lj = 0.0;
    if (lane < D) { w.lj = live[j * D + lane]; w.li = live[i * D + lane]; }
}
__device__ __forceinline__ void walk_step_pre_c(const int D, const double (&r)[7], WalkPre& w) {
    w.gamma = r[2] < 0.5 ? 1.0 : 2.38 / sqrt(2.0 * (double)D) * (-0.25 * log(r[3] * r[4] * r[5] * r[6]));   // Gamma(4, 1/4)
}
__device__ __forceinline__ void walk_step_pre(const int D, const long c, const int lane, const double* __restrict__ live, const long n_live,
                                              const uint64_t* __restrict__ key, const uint64_t rng_step, const double* u, const double* v,
                                              const double* prop, const double* theta, const int32_t* inside,
                                              const double* __restrict__ loglstar, const int32_t* __restrict__ counts,
                                              const int32_t* __restrict__ n_steps, const uint64_t step, WalkPre& w) {
    WalkPreKey k;
    double r[7];
    walk_step_pre_a(D, c, lane, key, u, v, prop, theta, inside, loglstar, counts, n_steps, step, w, k);
    walk_step_pre_b(D, lane, live, n_live, rng_step, k, w, r);
    walk_step_pre_c(D, r, w);
}

template <bool CON = true>
__device__ __forceinline__ void walk_step_post(const nmma_walk_prior* sp, const int D, const int T, const long c, const int lane, const double lp_in,
                                               const WalkPre& w, double* u, double* v, double* __restrict__ logl, int32_t* __restrict__ counts,
                                               double* prop, double* theta, int32_t* inside, const nmma_con_op* __restrict__ con_ops,
                                               const int n_con_ops, const bool propose, double* cstack = nullptr, const int cstride = 1) {
    bool acc = false;
    // a Constraint prior the proposal violates: the reference's likelihood returns the floor for it (core/base.py:77-82).  The first
    // lane of the chain's group runs the program (its stack in LDS: cstack, cstride) and hands the verdict to the group.
    int con_ok = 1;
    if constexpr (CON) {
        if (n_con_ops > 0) {
            if (w.active && w.in0 && lane == 0) con_ok = con_row_ok_lds(con_ops, n_con_ops, theta + c * D, cstack, cstride) ? 1 : 0;
            con_ok = __shfl(con_ok, (int)(threadIdx.x & 63) & ~(T - 1), 64);
        }
    }
    if (w.active) {
        double lp = lp_in;
        if (!con_ok) lp = NMMA_LOGL_FLOOR;
        acc = w.in0 && lp > w.lstar;
        if (acc && lane < D) { u[c * D + lane] = w.pp; v[c * D + lane] = w.th; }
        if (lane == 0) {
            int32_t* cnt = counts + 4 * c;                      // {accept, reject, nfail, ncall}
            if (!w.in0) cnt[2] = w.cnt2 + 1;
            else {
                cnt[3] = w.cnt3 + 1;
                if (acc) { logl[c] = lp; cnt[0] = w.cnt0 + 1; }
                else cnt[1] = w.cnt1 + 1;
            }
        }
    }
    if (!propose) return;
    const double u_now = acc ? w.pp : w.uu, v_now = acc ? w.th : w.vv;
    int in = 1;
    double x = 0.0;
    if (lane < D) {
        x = u_now + w.gamma * (w.lj - w.li);
        const int32_t bc = sp[lane].boundary;
        if (bc == NMMA_BOUNDARY_PERIODIC) x = floored_mod(x, 1.0);
        else if (bc == NMMA_BOUNDARY_REFLECTIVE) { const double q = floored_mod(x, 2.0); x = q > 1.0 ? 2.0 - q : q; }
        prop[c * D + lane] = x;
        in = (x >= 0.0) && (x <= 1.0);
    }
    for (int m = T >> 1; m > 0; m >>= 1) in &= __shfl_xor(in, m, 64);           // (groups are aligned powers of two: the exchange stays inside)
    if (lane == 0) inside[c] = in;
    if (lane < D) theta[c * D + lane] = in ? walk_rescale(sp[lane], x) : v_now;
}

}  // namespace nmma
