// Lock-step ensemble walk on the device (the sampler-side batching seam).
//
// The host adapter (nmma_amd/sampler.py: _LockstepWalk.run_many) spends ~0.5 ms of numpy per MCMC step for 4 096 chains -- counter
// hash, differential-evolution proposal, prior transform, accept bookkeeping -- around a 31 us likelihood launch.  These
// kernels move that bookkeeping next to the likelihood: one step is nmma_*_loglike -> accept + next proposal (nmma_walk_step), two
// launches on one stream, no host round trip.  The random numbers are the SAME counter hash as sampler.py:counter_uniforms (SplitMix64 of
// (chain key, step, draw)), the proposal is sampler.py:_propose (dynesty's "rwalk"-style differential evolution as
// bilby/core/sampler/dynesty_utils.py implements it; the reference builds those walker objects at mpi_setup.py:202-245), and the
// prior transform covers the analytic bilby priors by their published ``rescale`` formulas (bilby/core/prior/analytical.py).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/nmma_hip.h"
#include "nmma_common.h"
#include "walk_device.h"

namespace nmma {

__global__ __launch_bounds__(256) void con_floor_kernel(const nmma_con_op* __restrict__ ops, const int n_ops, const double* __restrict__ theta,
                                                        const long B, const long ld, double* __restrict__ logl) {
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    if (b < B && !con_row_ok(ops, n_ops, theta + b * ld)) logl[b] = NMMA_LOGL_FLOOR;
}

__global__ __launch_bounds__(256) void walk_propose_kernel(const WalkSpec S, const double* __restrict__ live, const long n_live,
                                                           const double* __restrict__ u, const double* __restrict__ v,
                                                           const uint64_t* __restrict__ key, const long n, const uint64_t step,
                                                           double* __restrict__ prop, double* __restrict__ theta, int32_t* __restrict__ inside) {
    __shared__ nmma_walk_prior sp[NMMA_WALK_MAX_DIM];
    walk_stage_spec(S, sp);
    const int T = walk_group(S.ndim);
    const long c = (long)blockIdx.x * (256 / T) + threadIdx.x / T;
    if (c < n) walk_propose_one(sp, S.ndim, T, c, threadIdx.x % T, live, n_live, u, v, key, step, prop, theta, inside);
}

__global__ __launch_bounds__(256) void walk_accept_kernel(const int D, const long n, const double* __restrict__ prop,
                                                          const double* __restrict__ theta, const int32_t* __restrict__ inside,
                                                          const double* __restrict__ l_prop, const double* __restrict__ loglstar,
                                                          double* __restrict__ u, double* __restrict__ v, double* __restrict__ logl,
                                                          int32_t* __restrict__ counts, const int32_t* __restrict__ n_steps,
                                                          const uint64_t step, const nmma_con_op* __restrict__ con_ops, const int n_con_ops) {
    const int T = walk_group(D);
    const long c = (long)blockIdx.x * (256 / T) + threadIdx.x / T;
    if (c < n) walk_accept_one(D, c, threadIdx.x % T, prop, theta, inside, l_prop[c], loglstar, u, v, logl, counts, n_steps, step, con_ops, n_con_ops);
}

// End of a queue (sampler.py: run_many): a chain that never accepted returns a fresh draw from the prior -- u = the counter hash of
// (key, step 0, dimension), v = theta = its prior transform; the likelihood launch that follows evaluates it and walk_fresh_logl_kernel
// files the value.  A group of lanes per chain as in walk_step_kernel.
__global__ __launch_bounds__(256) void walk_fresh_kernel(const WalkSpec S, const uint64_t* __restrict__ key, const long n, double* u, double* v,
                                                         double* theta, const int32_t* __restrict__ counts, int32_t* __restrict__ stuck) {
    __shared__ nmma_walk_prior sp[NMMA_WALK_MAX_DIM];
    walk_stage_spec(S, sp);
    const int D = S.ndim, T = walk_group(D);
    const long c = (long)blockIdx.x * (256 / T) + threadIdx.x / T;
    if (c >= n) return;
    const int lane = threadIdx.x % T;
    const int never = counts[4 * c] == 0;
    if (lane == 0) stuck[c] = never;
    if (never && lane < D) {
        const double x = walk_uniform(key[c], 0ull, (uint64_t)lane);
        const double t = walk_rescale(sp[lane], x);
        u[c * D + lane] = x; v[c * D + lane] = t; theta[c * D + lane] = t;
    }
}
__global__ __launch_bounds__(256) void walk_fresh_logl_kernel(const long n, const int D, const int32_t* __restrict__ stuck, const double* __restrict__ l_prop,
                                                              const double* __restrict__ theta, double* __restrict__ logl, int32_t* __restrict__ counts,
                                                              const nmma_con_op* __restrict__ con_ops, const int n_con_ops) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= n || !stuck[c]) return;
    double lp = l_prop[c];
    if (n_con_ops > 0 && !con_row_ok(con_ops, n_con_ops, theta + c * D)) lp = NMMA_LOGL_FLOOR;
    logl[c] = lp;
    counts[4 * c + 3] += 1;
}

// A queue's records as ONE device buffer (nmma_walk_queue::records_dev): row c = [u[D] | v[D] | logl | counts], the four int32 counters in
// the bit patterns of two doubles -- what a rank of a sharded queue hands to the all-gather (parallel.ShardedQueue: the shards' records are
// exchanged on the device, RCCL over xGMI, and downloaded once).  One thread per (chain, column).
__global__ __launch_bounds__(256) void walk_pack_records_kernel(const long n, const int D, const double* __restrict__ u, const double* __restrict__ v,
                                                                const double* __restrict__ logl, const int32_t* __restrict__ counts,
                                                                double* __restrict__ rec) {
    const int W = 2 * D + 3;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * W) return;
    const long c = i / W;
    const int j = (int)(i - c * W);
    double x;
    if (j < D) x = u[c * D + j];
    else if (j < 2 * D) x = v[c * D + (j - D)];
    else if (j == 2 * D) x = logl[c];
    else x = reinterpret_cast<const double*>(counts + 4 * c)[j - 2 * D - 1];      // (counts + 4 c is 16-byte aligned)
    rec[i] = x;
}

// accept of step `step` and proposal of step `step + 1` in one launch (a chain's accept touches only its own row, and its next proposal
// reads that row and the fixed live points; lane d of the group owns element d of both): an MCMC step is then two launches -- this one
// and the likelihood.  (The group reads inside[c] for the accept before its lane 0 stores the next flag: one wavefront, program order.)
__global__ __launch_bounds__(256) void walk_step_kernel(const WalkSpec S, const double* __restrict__ live, const long n_live,
                                                        const uint64_t* __restrict__ key, const long n, double* prop, double* theta,
                                                        int32_t* inside, const double* __restrict__ l_prop,
                                                        const double* __restrict__ loglstar, double* u, double* v, double* __restrict__ logl,
                                                        int32_t* __restrict__ counts, const int32_t* __restrict__ n_steps, const uint64_t step,
                                                        const uint64_t first_step, const nmma_con_op* __restrict__ con_ops, const int n_con_ops) {
    __shared__ nmma_walk_prior sp[NMMA_WALK_MAX_DIM];
    walk_stage_spec(S, sp);
    const int T = walk_group(S.ndim);
    const long c = (long)blockIdx.x * (256 / T) + threadIdx.x / T;
    if (c >= n) return;
    const int lane = threadIdx.x % T;
    walk_accept_one(S.ndim, c, lane, prop, theta, inside, l_prop[c], loglstar, u, v, logl, counts, n_steps, step, con_ops, n_con_ops);
    walk_propose_one(sp, S.ndim, T, c, lane, live, n_live, u, v, key, first_step + step, prop, theta, inside);
}

// The accept step of bilby's AcceptanceTrackingRWalk ("rwalk", mpi_setup.py:234-245): as walk_accept_kernel for the chains still
// running, then the chain's autocorrelation estimate from its running acceptance ratio (bilby dynesty_utils.estimate_nmcmc with
// safety = 1, smoothed over tau calls with the estimate old_act the previous queue left behind; old_act < 0: none) and whether it
// goes on: step < nact * act and accept + reject <= maxmcmc (sampler.py: AcceptanceTrackingRWalk._after_step / _continues).
__global__ __launch_bounds__(256) void walk_accept_rwalk_kernel(const int D, const long n, const double* __restrict__ prop,
                                                                const double* __restrict__ theta, const int32_t* __restrict__ inside,
                                                                const double* __restrict__ l_prop, const double* __restrict__ loglstar,
                                                                double* __restrict__ u, double* __restrict__ v, double* __restrict__ logl,
                                                                int32_t* __restrict__ counts, double* __restrict__ act,
                                                                int32_t* __restrict__ active, const uint64_t step, const double nact,
                                                                const int32_t maxmcmc, const double tau, const double old_act) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n || !active[c]) return;
    int32_t* cnt = counts + 4 * c;                          // {accept, reject, nfail, ncall}
    if (!inside[c]) cnt[2] += 1;
    else {
        cnt[3] += 1;
        if (l_prop[c] > loglstar[c]) {
            for (int d = 0; d < D; ++d) { u[c * D + d] = prop[c * D + d]; v[c * D + d] = theta[c * D + d]; }
            logl[c] = l_prop[c];
            cnt[0] += 1;
        } else cnt[1] += 1;
    }
    const double a = (double)cnt[0], r = (double)cnt[1], f = (double)cnt[2];
    if (a + r > nact) {
        const double ratio = a / (a + r + f);
        double n_exact;
        if (ratio == 0.0) n_exact = old_act < 0.0 ? HUGE_VAL : (1.0 + 1.0 / tau) * old_act;
        else {
            n_exact = 2.0 / ratio - 1.0;                      // safety = 1
            if (old_act >= 0.0) n_exact = (1.0 - 1.0 / tau) * old_act + n_exact / tau;
        }
        const double capped = n_exact < (double)maxmcmc ? n_exact : (double)maxmcmc;
        act[c] = capped > 1.0 ? capped : 1.0;
    }
    active[c] = ((double)step < nact * act[c] && cnt[0] + cnt[1] <= maxmcmc) ? 1 : 0;
}

// nmma_walk_accept_rwalk for step `step` and the proposal of step `step + 1` in one launch, a group of lanes per chain as in
// walk_step_kernel (a chain that has stopped still gets a proposal; its accept ignores it).
__global__ __launch_bounds__(256) void walk_step_rwalk_kernel(const WalkSpec S, const double* __restrict__ live, const long n_live,
                                                              const uint64_t* __restrict__ key, const long n, double* prop, double* theta,
                                                              int32_t* inside, const double* __restrict__ l_prop,
                                                              const double* __restrict__ loglstar, double* u, double* v,
                                                              double* __restrict__ logl, int32_t* __restrict__ counts, double* __restrict__ act,
                                                              int32_t* active, const uint64_t step, const double nact, const int32_t maxmcmc,
                                                              const double tau, const double old_act) {
    __shared__ nmma_walk_prior sp[NMMA_WALK_MAX_DIM];
    walk_stage_spec(S, sp);
    const int D = S.ndim, T = walk_group(D);
    const long c = (long)blockIdx.x * (256 / T) + threadIdx.x / T;
    if (c >= n) return;
    const int lane = threadIdx.x % T;
    if (active[c]) {                                           // (uniform over the chain's lanes; lane 0 stores the new flag last)
        const int in = inside[c];
        const double lp = l_prop[c];
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
            const double a = (double)cnt[0], r = (double)cnt[1], f = (double)cnt[2];
            double ac = act[c];
            if (a + r > nact) {
                const double ratio = a / (a + r + f);
                double n_exact;
                if (ratio == 0.0) n_exact = old_act < 0.0 ? HUGE_VAL : (1.0 + 1.0 / tau) * old_act;
                else {
                    n_exact = 2.0 / ratio - 1.0;                  // safety = 1
                    if (old_act >= 0.0) n_exact = (1.0 - 1.0 / tau) * old_act + n_exact / tau;
                }
                const double capped = n_exact < (double)maxmcmc ? n_exact : (double)maxmcmc;
                ac = capped > 1.0 ? capped : 1.0;
                act[c] = ac;
            }
            active[c] = ((double)step < nact * ac && cnt[0] + cnt[1] <= maxmcmc) ? 1 : 0;
        }
    }
    walk_propose_one(sp, D, T, c, lane, live, n_live, u, v, key, step + 1, prop, theta, inside);
}

// the prior transform alone (start points, fresh draws): theta = rescale(u)
__global__ __launch_bounds__(256) void walk_rescale_kernel(const WalkSpec S, const long n, const double* __restrict__ u, double* __restrict__ theta) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    for (int d = 0; d < S.ndim; ++d) theta[c * S.ndim + d] = walk_rescale(S.p[d], u[c * S.ndim + d]);
}

static unsigned walk_blocks(int64_t n, int32_t ndim) {       // 256 / T chains per workgroup (walk_group)
    const int per = 256 / (ndim <= 8 ? 8 : ndim <= 16 ? 16 : 32);
    return (unsigned)((n + per - 1) / per);
}

static int walk_spec(const nmma_walk_prior* priors, int32_t ndim, WalkSpec* S, const char* what) {
    if (!priors || ndim < 1 || ndim > NMMA_WALK_MAX_DIM) return fail(std::string(what) + ": 1 .. NMMA_WALK_MAX_DIM dimensions");
    S->ndim = ndim;
    for (int d = 0; d < ndim; ++d) {
        if (priors[d].kind < NMMA_PRIOR_UNIFORM || priors[d].kind > NMMA_PRIOR_HALF_GAUSSIAN) return fail(std::string(what) + ": unknown prior kind");
        S->p[d] = priors[d];
    }
    return 0;
}

}  // namespace nmma

extern "C" {

int32_t nmma_walk_propose(const nmma_walk_prior* priors, int32_t ndim, const double* live_dev, int64_t n_live, const double* u_dev,
                          const double* v_dev, const uint64_t* key_dev, int64_t n, uint64_t step, double* prop_dev, double* theta_dev,
                          int32_t* inside_dev, int32_t device, void* stream) {
    using namespace nmma;
    WalkSpec S;
    if (walk_spec(priors, ndim, &S, "nmma_walk_propose")) return 1;
    if (!live_dev || !u_dev || !v_dev || !key_dev || !prop_dev || !theta_dev || !inside_dev || n < 0 || n_live < 3)
        return fail("nmma_walk_propose: bad argument (at least three live points)");
    if (n == 0) return 0;
    if (hipSetDevice(device) != hipSuccess) return fail("nmma_walk_propose: hipSetDevice failed");
    hipLaunchKernelGGL(walk_propose_kernel, dim3(walk_blocks(n, ndim)), dim3(256), 0, static_cast<hipStream_t>(stream), S, live_dev,
                       (long)n_live, u_dev, v_dev, key_dev, (long)n, step, prop_dev, theta_dev, inside_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string("nmma_walk_propose launch failed: ") + hipGetErrorString(e));
    return 0;
}

int32_t nmma_walk_accept(int32_t ndim, int64_t n, const double* prop_dev, const double* theta_dev, const int32_t* inside_dev,
                         const double* logl_prop_dev, const double* loglstar_dev, double* u_dev, double* v_dev, double* logl_dev,
                         int32_t* counts_dev, const int32_t* n_steps_dev, uint64_t step, int32_t device, void* stream) {
    using namespace nmma;
    if (ndim < 1 || ndim > NMMA_WALK_MAX_DIM || n < 0 || !prop_dev || !theta_dev || !inside_dev || !logl_prop_dev || !loglstar_dev || !u_dev ||
        !v_dev || !logl_dev || !counts_dev) return fail("nmma_walk_accept: bad argument");
    if (n == 0) return 0;
    if (hipSetDevice(device) != hipSuccess) return fail("nmma_walk_accept: hipSetDevice failed");
    hipLaunchKernelGGL(walk_accept_kernel, dim3(walk_blocks(n, ndim)), dim3(256), 0, static_cast<hipStream_t>(stream), (int)ndim, (long)n,
                       prop_dev, theta_dev, inside_dev, logl_prop_dev, loglstar_dev, u_dev, v_dev, logl_dev, counts_dev, n_steps_dev, step,
                       (const nmma_con_op*)nullptr, 0);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string("nmma_walk_accept launch failed: ") + hipGetErrorString(e));
    return 0;
}

int32_t nmma_walk_step(const nmma_walk_prior* priors, int32_t ndim, const double* live_dev, int64_t n_live, const uint64_t* key_dev, int64_t n,
                       double* prop_dev, double* theta_dev, int32_t* inside_dev, const double* logl_prop_dev, const double* loglstar_dev,
                       double* u_dev, double* v_dev, double* logl_dev, int32_t* counts_dev, const int32_t* n_steps_dev, uint64_t step,
                       uint64_t first_step, int32_t device, void* stream) {
    using namespace nmma;
    WalkSpec S;
    if (walk_spec(priors, ndim, &S, "nmma_walk_step")) return 1;
    if (!live_dev || !key_dev || !prop_dev || !theta_dev || !inside_dev || !logl_prop_dev || !loglstar_dev || !u_dev || !v_dev || !logl_dev ||
        !counts_dev || n < 0 || n_live < 3) return fail("nmma_walk_step: bad argument (at least three live points)");
    if (n == 0) return 0;
    if (hipSetDevice(device) != hipSuccess) return fail("nmma_walk_step: hipSetDevice failed");
    hipLaunchKernelGGL(walk_step_kernel, dim3(walk_blocks(n, ndim)), dim3(256), 0, static_cast<hipStream_t>(stream), S, live_dev,
                       (long)n_live, key_dev, (long)n, prop_dev, theta_dev, inside_dev, logl_prop_dev, loglstar_dev, u_dev, v_dev, logl_dev,
                       counts_dev, n_steps_dev, step, first_step, (const nmma_con_op*)nullptr, 0);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string("nmma_walk_step launch failed: ") + hipGetErrorString(e));
    return 0;
}

int32_t nmma_walk_accept_rwalk(int32_t ndim, int64_t n, const double* prop_dev, const double* theta_dev, const int32_t* inside_dev,
                               const double* logl_prop_dev, const double* loglstar_dev, double* u_dev, double* v_dev, double* logl_dev,
                               int32_t* counts_dev, double* act_dev, int32_t* active_dev, uint64_t step, double nact, int32_t maxmcmc,
                               double tau, double old_act, int32_t device, void* stream) {
    using namespace nmma;
    if (ndim < 1 || ndim > NMMA_WALK_MAX_DIM || n < 0 || !prop_dev || !theta_dev || !inside_dev || !logl_prop_dev || !loglstar_dev || !u_dev ||
        !v_dev || !logl_dev || !counts_dev || !act_dev || !active_dev || !(tau > 0) || maxmcmc < 1) return fail("nmma_walk_accept_rwalk: bad argument");
    if (n == 0) return 0;
    if (hipSetDevice(device) != hipSuccess) return fail("nmma_walk_accept_rwalk: hipSetDevice failed");
    hipLaunchKernelGGL(walk_accept_rwalk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), (int)ndim,
                       (long)n, prop_dev, theta_dev, inside_dev, logl_prop_dev, loglstar_dev, u_dev, v_dev, logl_dev, counts_dev, act_dev, active_dev,
                       step, nact, maxmcmc, tau, old_act);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string("nmma_walk_accept_rwalk launch failed: ") + hipGetErrorString(e));
    return 0;
}

int32_t nmma_walk_step_rwalk(const nmma_walk_prior* priors, int32_t ndim, const double* live_dev, int64_t n_live, const uint64_t* key_dev,
                             int64_t n, double* prop_dev, double* theta_dev, int32_t* inside_dev, const double* logl_prop_dev,
                             const double* loglstar_dev, double* u_dev, double* v_dev, double* logl_dev, int32_t* counts_dev, double* act_dev,
                             int32_t* active_dev, uint64_t step, double nact, int32_t maxmcmc, double tau, double old_act, int32_t device,
                             void* stream) {
    using namespace nmma;
    WalkSpec S;
    if (walk_spec(priors, ndim, &S, "nmma_walk_step_rwalk")) return 1;
    if (!live_dev || !key_dev || !prop_dev || !theta_dev || !inside_dev || !logl_prop_dev || !loglstar_dev || !u_dev || !v_dev || !logl_dev ||
        !counts_dev || !act_dev || !active_dev || n < 0 || n_live < 3 || !(tau > 0) || maxmcmc < 1)
        return fail("nmma_walk_step_rwalk: bad argument (at least three live points)");
    if (n == 0) return 0;
    if (hipSetDevice(device) != hipSuccess) return fail("nmma_walk_step_rwalk: hipSetDevice failed");
    hipLaunchKernelGGL(walk_step_rwalk_kernel, dim3(walk_blocks(n, ndim)), dim3(256), 0, static_cast<hipStream_t>(stream), S, live_dev, (long)n_live,
                       key_dev, (long)n, prop_dev, theta_dev, inside_dev, logl_prop_dev, loglstar_dev, u_dev, v_dev, logl_dev, counts_dev, act_dev,
                       active_dev, step, nact, maxmcmc, tau, old_act);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string("nmma_walk_step_rwalk launch failed: ") + hipGetErrorString(e));
    return 0;
}

int32_t nmma_walk_rescale(const nmma_walk_prior* priors, int32_t ndim, const double* u_dev, int64_t n, double* theta_dev, int32_t device,
                          void* stream) {
    using namespace nmma;
    WalkSpec S;
    if (walk_spec(priors, ndim, &S, "nmma_walk_rescale")) return 1;
    if (!u_dev || !theta_dev || n < 0) return fail("nmma_walk_rescale: bad argument");
    if (n == 0) return 0;
    if (hipSetDevice(device) != hipSuccess) return fail("nmma_walk_rescale: hipSetDevice failed");
    hipLaunchKernelGGL(walk_rescale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), S, (long)n, u_dev,
                       theta_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string("nmma_walk_rescale launch failed: ") + hipGetErrorString(e));
    return 0;
}

// ---- constraint programs -------------------------------------------------------------------------------------------------------------
struct nmma_con_program {
    int32_t device = 0, n_ops = 0, n_cols = 0;
    nmma_con_op* ops_d = nullptr;
};

int32_t nmma_con_create(const nmma_con_op* ops, int32_t n_ops, int32_t n_cols, int32_t device, nmma_con_program** out) {
    using namespace nmma;
    if (!ops || !out || n_ops < 1 || n_ops > NMMA_CON_MAX_OPS || n_cols < 1) return fail("nmma_con_create: bad argument (1 .. NMMA_CON_MAX_OPS operations)");
    *out = nullptr;
    int sp = 0;                       // the program is validated on the host: the kernel then runs it unchecked
    for (int i = 0; i < n_ops; ++i) {
        const int op = ops[i].op;
        if (op == NMMA_CON_PUSH_COL || op == NMMA_CON_PUSH_CONST) {
            if (op == NMMA_CON_PUSH_COL && (ops[i].col < 0 || ops[i].col >= n_cols)) return fail("nmma_con_create: column out of range");
            if (++sp > NMMA_CON_MAX_STACK) return fail("nmma_con_create: expression too deep (NMMA_CON_MAX_STACK)");
        } else if (op >= NMMA_CON_ADD && op <= NMMA_CON_MAX) {
            if (sp < 2) return fail("nmma_con_create: stack underflow");
            --sp;
        } else if ((op >= NMMA_CON_NEG && op <= NMMA_CON_SIGN) || op == NMMA_CON_CHECK_GT) {
            if (sp < 1) return fail("nmma_con_create: stack underflow");
        } else if (op == NMMA_CON_CHECK_LT) {
            if (sp < 1) return fail("nmma_con_create: stack underflow");
            --sp;
        } else return fail("nmma_con_create: unknown operation");
    }
    if (sp != 0) return fail("nmma_con_create: values left on the stack (every expression ends in CHECK_GT, CHECK_LT)");
    if (hipSetDevice(device) != hipSuccess) return fail("nmma_con_create: hipSetDevice failed (no CPU fallback)");
    auto* p = new nmma_con_program();
    p->device = device; p->n_ops = n_ops; p->n_cols = n_cols;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->ops_d), sizeof(nmma_con_op) * n_ops);
    if (e == hipSuccess) e = hipMemcpy(p->ops_d, ops, sizeof(nmma_con_op) * n_ops, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (p->ops_d) (void)hipFree(p->ops_d);
        delete p;
        return fail(std::string("nmma_con_create: ") + hipGetErrorString(e));
    }
    *out = p;
    return 0;
}

void nmma_con_destroy(nmma_con_program* p) {
    if (!p) return;
    if (p->ops_d) { (void)hipSetDevice(p->device); (void)hipFree(p->ops_d); }
    delete p;
}

int32_t nmma_con_floor(const nmma_con_program* p, const double* theta_dev, int64_t B, int64_t ld, double* logl_dev, void* stream) {
    using namespace nmma;
    if (!p || !theta_dev || !logl_dev || B < 0 || ld < p->n_cols) return fail("nmma_con_floor: bad argument (ld must cover the program's columns)");
    if (B == 0) return 0;
    if (hipSetDevice(p->device) != hipSuccess) return fail("nmma_con_floor: hipSetDevice failed");
    hipLaunchKernelGGL(con_floor_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p->ops_d, (int)p->n_ops,
                       theta_dev, (long)B, (long)ld, logl_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string("nmma_con_floor launch failed: ") + hipGetErrorString(e));
    return 0;
}

// ---- a whole queue of the nested sampler -----------------------------------------------------------------------------------------------
// Workspace: ONE device allocation and ONE pinned host mirror with the same layout,
//   [ live | loglstar | key | walks | u ][ v | logl | counts ]  prop | theta | l_prop | inside | stuck
//   '------------- upload ------------'
//                               '------- download ---------'
// so that a queue costs one host-to-device and one device-to-host copy.
struct nmma_walk_ws {
    int32_t device = 0;
    size_t cap = 0;                 // bytes of each of the two buffers
    unsigned char* dev = nullptr;
    unsigned char* pin = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // a queue enqueued by nmma_em_walk_queue_begin and not yet collected by nmma_em_walk_queue_end
    bool pending = false;
    hipStream_t pend_stream = nullptr;
    long pend_n = 0, pend_d = 0;
    size_t pend_u = 0, pend_v = 0, pend_logl = 0, pend_cnt = 0;
    bool pend_dev_records = false;   // the records went to nmma_walk_queue::records_dev: nothing to unpack
};

int32_t nmma_walk_ws_create(int32_t device, nmma_walk_ws** out) {
    using namespace nmma;
    if (!out) return fail("nmma_walk_ws_create: null argument");
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return fail("nmma_walk_ws_create: hipSetDevice failed (no CPU fallback)");
    auto* ws = new nmma_walk_ws();
    ws->device = device;
    if (hipEventCreate(&ws->ev0) != hipSuccess || hipEventCreate(&ws->ev1) != hipSuccess) { delete ws; return fail("nmma_walk_ws_create: hipEventCreate failed"); }
    *out = ws;
    return 0;
}

void nmma_walk_ws_destroy(nmma_walk_ws* ws) {
    if (!ws) return;
    (void)hipSetDevice(ws->device);
    if (ws->dev) (void)hipFree(ws->dev);
    if (ws->pin) (void)hipHostFree(ws->pin);
    if (ws->ev0) (void)hipEventDestroy(ws->ev0);
    if (ws->ev1) (void)hipEventDestroy(ws->ev1);
    delete ws;
}

// nmma_em_walk_queue in two halves, so that ONE host thread can keep several devices busy (a queue sharded over the devices'
// handles: begin on every device, then end on every device).  begin packs and uploads, enqueues the whole step loop and the
// download on `stream` and returns without waiting; end waits for that stream, unpacks into q's output arrays and checks the handle.
int32_t nmma_em_walk_queue_begin(nmma_em_handle* h, nmma_walk_ws* ws, const nmma_walk_queue* q, void* stream) {
    using namespace nmma;
    if (!h || !ws || !q) return fail("nmma_em_walk_queue: null argument");
    if (ws->pending) return fail("nmma_em_walk_queue_begin: the workspace already holds a queue that was not collected (nmma_em_walk_queue_end)");
    WalkSpec S;
    if (walk_spec(q->priors, q->ndim, &S, "nmma_em_walk_queue")) return 1;
    const long n = (long)q->n, D = q->ndim, NL = (long)q->n_live;
    const bool dev_rec = q->records_dev != nullptr;
    if (n < 0 || NL < 3 || !q->live || !q->u0 || !q->loglstar || !q->key || (!dev_rec && (!q->u || !q->v || !q->logl || !q->counts)) ||
        (!q->walks_per_chain && q->walks < 1))
        return fail("nmma_em_walk_queue: bad argument (at least three live points, walks >= 1, host outputs or records_dev)");
    if (nmma_em_device(h) != ws->device) return fail("nmma_em_walk_queue: the workspace belongs to another device than the likelihood handle");
    if (q->constraints && (q->constraints->device != ws->device || q->constraints->n_cols > D))
        return fail("nmma_em_walk_queue: the constraint program belongs to another device or reads columns the walk does not sample");
    ws->pend_n = n; ws->pend_d = D; ws->pend_stream = static_cast<hipStream_t>(stream); ws->pend_dev_records = dev_rec;
    if (n == 0) { ws->pending = true; return 0; }
    if (hipSetDevice(ws->device) != hipSuccess) return fail("nmma_em_walk_queue: hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    size_t off = 0;
    const size_t o_wf = off;    off = al(off + sizeof(nmma_walk_fuse));        // the fused step's view of the chains (device copy)
    const size_t o_live = off;  off = al(off + sizeof(double) * NL * D);
    const size_t o_star = off;  off = al(off + sizeof(double) * n);
    const size_t o_key = off;   off = al(off + sizeof(uint64_t) * n);
    const size_t o_walks = off; off = al(off + sizeof(int32_t) * n);
    const size_t o_u = off;     off = al(off + sizeof(double) * n * D);
    const size_t up_end = off;
    const size_t o_v = off;     off = al(off + sizeof(double) * n * D);
    const size_t o_logl = off;  off = al(off + sizeof(double) * n);
    const size_t o_cnt = off;   off = al(off + sizeof(int32_t) * 4 * n);
    const size_t down_end = off;
    const size_t o_prop = off;  off = al(off + sizeof(double) * n * D);
    const size_t o_theta = off; off = al(off + sizeof(double) * n * D);
    const size_t o_lp = off;    off = al(off + sizeof(double) * n);
    const size_t o_in = off;    off = al(off + sizeof(int32_t) * n);
    const size_t o_stuck = off; off = al(off + sizeof(int32_t) * n);
    if (off > ws->cap) {
        if (hipStreamSynchronize(s) != hipSuccess) return fail("nmma_em_walk_queue: stream error before growing the workspace");
        if (ws->dev) (void)hipFree(ws->dev);
        if (ws->pin) (void)hipHostFree(ws->pin);
        ws->dev = ws->pin = nullptr; ws->cap = 0;
        const size_t want = off + off / 4;
        if (hipMalloc(reinterpret_cast<void**>(&ws->dev), want) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&ws->pin), want, hipHostMallocDefault) != hipSuccess)
            return fail("nmma_em_walk_queue: out of memory for the walk workspace");
        ws->cap = want;
    }
    unsigned char *d = ws->dev, *p = ws->pin;
    memcpy(p + o_live, q->live, sizeof(double) * NL * D);
    memcpy(p + o_star, q->loglstar, sizeof(double) * n);
    memcpy(p + o_key, q->key, sizeof(uint64_t) * n);
    int max_walks = q->walks;
    if (q->walks_per_chain) {
        memcpy(p + o_walks, q->walks_per_chain, sizeof(int32_t) * n);
        max_walks = 0;
        for (long c = 0; c < n; ++c) max_walks = q->walks_per_chain[c] > max_walks ? q->walks_per_chain[c] : max_walks;
    }
    memcpy(p + o_u, q->u0, sizeof(double) * n * D);
    // NaN = "never moved" (sampler.py: run_many); the counts start at zero
    { double* lg = reinterpret_cast<double*>(p + o_logl); for (long c = 0; c < n; ++c) lg[c] = __builtin_nan(""); }
    memset(p + o_cnt, 0, sizeof(int32_t) * 4 * n);
    {
        nmma_walk_fuse* wf = reinterpret_cast<nmma_walk_fuse*>(p + o_wf);
        memset(wf, 0, sizeof(*wf));
        for (int dd = 0; dd < D; ++dd) wf->priors[dd] = S.p[dd];
        wf->ndim = (int32_t)D;
        wf->n_con_ops = q->constraints ? q->constraints->n_ops : 0;
        wf->live = reinterpret_cast<const double*>(d + o_live); wf->n_live = NL;
        wf->key = reinterpret_cast<const uint64_t*>(d + o_key);
        wf->prop = reinterpret_cast<double*>(d + o_prop); wf->inside = reinterpret_cast<int32_t*>(d + o_in);
        wf->loglstar = reinterpret_cast<const double*>(d + o_star);
        wf->u = reinterpret_cast<double*>(d + o_u); wf->v = reinterpret_cast<double*>(d + o_v);
        wf->logl = reinterpret_cast<double*>(d + o_logl); wf->counts = reinterpret_cast<int32_t*>(d + o_cnt);
        wf->n_steps = q->walks_per_chain ? reinterpret_cast<const int32_t*>(d + o_walks) : nullptr;
        wf->con_ops = q->constraints ? q->constraints->ops_d : nullptr;
        wf->first_step = q->first_step;
    }
#define NMQ(call, what) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return fail(std::string("nmma_em_walk_queue: ") + what + ": " + hipGetErrorString(e_)); } while (0)
    // (everything from here on is enqueued work that reads ws->pin / ws->dev: a failure part-way drains the stream before it returns, so
    //  that a retry never writes into a pinned buffer a copy is still reading)
    auto enqueue = [&]() -> int {
    NMQ(hipEventRecord(ws->ev0, s), "event");
    NMQ(hipMemcpyAsync(d, p, up_end, hipMemcpyHostToDevice, s), "upload");
    NMQ(hipMemcpyAsync(d + o_logl, p + o_logl, o_prop - o_logl, hipMemcpyHostToDevice, s), "upload of the initial state");
    double *live_d = reinterpret_cast<double*>(d + o_live), *star_d = reinterpret_cast<double*>(d + o_star), *u_d = reinterpret_cast<double*>(d + o_u),
           *v_d = reinterpret_cast<double*>(d + o_v), *logl_d = reinterpret_cast<double*>(d + o_logl), *prop_d = reinterpret_cast<double*>(d + o_prop),
           *theta_d = reinterpret_cast<double*>(d + o_theta), *lp_d = reinterpret_cast<double*>(d + o_lp);
    const uint64_t* key_d = reinterpret_cast<const uint64_t*>(d + o_key);
    const int32_t* walks_d = q->walks_per_chain ? reinterpret_cast<const int32_t*>(d + o_walks) : nullptr;
    int32_t *cnt_d = reinterpret_cast<int32_t*>(d + o_cnt), *in_d = reinterpret_cast<int32_t*>(d + o_in), *stuck_d = reinterpret_cast<int32_t*>(d + o_stuck);
    const nmma_con_op* con_ops = q->constraints ? q->constraints->ops_d : nullptr;
    const int n_con = q->constraints ? q->constraints->n_ops : 0;
    const dim3 grid_g(walk_blocks(n, (int32_t)D)), grid_t((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(walk_rescale_kernel, grid_t, block, 0, s, S, n, u_d, v_d);
    hipLaunchKernelGGL(walk_propose_kernel, grid_g, block, 0, s, S, live_d, NL, u_d, v_d, key_d, n, q->first_step, prop_d, theta_d, in_d);
    // One launch per MCMC step where the handle's task flavour carries the fused step (accept + next proposal in the likelihood
    // kernel's epilogue, nmma_em_loglike_walk); else the likelihood launch followed by walk_step_kernel.  Same device functions,
    // same bits.
    // (queues beyond 4096 chains -- one round of 16-sample tiles -- run the fused step on the likelihood's 32-sample tiles; walking them
    //  in chunks of 4096 chains lost to two launches per step: 8192 chains x 100 steps 6.15 against 5.37 ms for config 2,
    //  profiles/r05_fused_mcmc_step.md)
    bool fused = true;
    const nmma_walk_fuse* wf_d = reinterpret_cast<const nmma_walk_fuse*>(d + o_wf);
    for (int k = 1; k <= max_walks; ++k) {
        if (fused) {
            const int rc = nmma_em_loglike_walk(h, theta_d, n, D, lp_d, wf_d, (uint64_t)k, k == max_walks ? 1 : 0, n_con, stream);
            if (rc == 1) return 1;
            if (rc == 0) continue;
            fused = false;
        }
        if (nmma_em_loglike(h, theta_d, n, D, lp_d, stream)) return 1;
        if (k < max_walks)
            hipLaunchKernelGGL(walk_step_kernel, grid_g, block, 0, s, S, live_d, NL, key_d, n, prop_d, theta_d, in_d, lp_d, star_d, u_d, v_d, logl_d, cnt_d,
                               walks_d, (uint64_t)k, q->first_step, con_ops, n_con);
        else
            hipLaunchKernelGGL(walk_accept_kernel, grid_g, block, 0, s, (int)D, n, prop_d, theta_d, in_d, lp_d, star_d, u_d, v_d, logl_d, cnt_d, walks_d,
                               (uint64_t)k, con_ops, n_con);
    }
    hipLaunchKernelGGL(walk_fresh_kernel, grid_g, block, 0, s, S, key_d, n, u_d, v_d, theta_d, cnt_d, stuck_d);
    if (nmma_em_loglike(h, theta_d, n, D, lp_d, stream)) return 1;
    hipLaunchKernelGGL(walk_fresh_logl_kernel, grid_t, block, 0, s, n, (int)D, stuck_d, lp_d, theta_d, logl_d, cnt_d, con_ops, n_con);
    NMQ(hipGetLastError(), "launch");
    if (dev_rec) {      // the records stay on the device, packed for the caller's collective; no download
        const long cells = n * (2 * D + 3);
        hipLaunchKernelGGL(walk_pack_records_kernel, dim3((unsigned)((cells + 255) / 256)), block, 0, s, n, (int)D, u_d, v_d, logl_d, cnt_d, q->records_dev);
        NMQ(hipGetLastError(), "launch of the record packing");
    } else {
        NMQ(hipMemcpyAsync(p + o_u, d + o_u, down_end - o_u, hipMemcpyDeviceToHost, s), "download");
    }
    NMQ(hipEventRecord(ws->ev1, s), "event");
    return 0;
    };
    if (enqueue()) { (void)hipStreamSynchronize(s); return 1; }
    ws->pend_u = o_u; ws->pend_v = o_v; ws->pend_logl = o_logl; ws->pend_cnt = o_cnt;
    ws->pending = true;
    return 0;
}

int32_t nmma_em_walk_queue_end(nmma_em_handle* h, nmma_walk_ws* ws, nmma_walk_queue* q) {
    using namespace nmma;
    if (!h || !ws || !q) return fail("nmma_em_walk_queue_end: null argument");
    if (!ws->pending) return fail("nmma_em_walk_queue_end: no queue was begun on this workspace");
    ws->pending = false;
    q->gpu_ms = 0.0;
    const long n = ws->pend_n, D = ws->pend_d;
    if (n == 0) return 0;
    const bool dev_rec = ws->pend_dev_records;
    if (n != (long)q->n || D != q->ndim || (dev_rec ? q->records_dev == nullptr : (!q->u || !q->v || !q->logl || !q->counts)))
        return fail("nmma_em_walk_queue_end: the queue record does not match the one that was begun");
    if (hipSetDevice(ws->device) != hipSuccess) return fail("nmma_em_walk_queue_end: hipSetDevice failed");
    NMQ(hipStreamSynchronize(ws->pend_stream), "stream");
    float ms = 0.f;
    NMQ(hipEventElapsedTime(&ms, ws->ev0, ws->ev1), "event time");
#undef NMQ
    q->gpu_ms = ms;
    if (dev_rec) return nmma_em_check(h) ? 1 : 0;
    const unsigned char* p = ws->pin;
    memcpy(q->u, p + ws->pend_u, sizeof(double) * n * D);
    memcpy(q->v, p + ws->pend_v, sizeof(double) * n * D);
    memcpy(q->logl, p + ws->pend_logl, sizeof(double) * n);
    memcpy(q->counts, p + ws->pend_cnt, sizeof(int32_t) * 4 * n);
    return nmma_em_check(h) ? 1 : 0;
}

int32_t nmma_em_walk_queue(nmma_em_handle* h, nmma_walk_ws* ws, nmma_walk_queue* q, void* stream) {
    if (nmma_em_walk_queue_begin(h, ws, q, stream)) return 1;
    return nmma_em_walk_queue_end(h, ws, q);
}

}  // extern "C"
