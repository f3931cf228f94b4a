// em_lc.h -- the likelihood-from-curves kernel em_lc_loglike<G, NM, SD, SA> (its body is also the second half of stack2_redo's tile work,
// em_kernels.hip) and its launcher.  Included by em_kernels.hip, which only uses the body and calls the launcher, and by em_lc.hip,
// which instantiates the fourteen launcher / kernel variants: a translation unit of its own, as for em_logl -- inside em_kernels.hip
// they made that unit the build's critical path.
#pragma once
#include "em_host.h"

namespace nmma {

// =======================================================================================
// em_lc_loglike: likelihood from SUPPLIED source-frame light curves lc[B][M][NS] (absolute
// magnitudes on the handle's sample_times, +inf / NaN where the model has no value) --
// the generic tail of the reference path for models whose light curve is produced by
// another kernel (Me2017) or by the caller (GRB afterglow, combined models):
//   combine_detector_data (model.py:381-404), sanity_check (em_likelihood.py:305-311),
//   autocomplete_data with its finite mask (utils.py:626-645), band_log_likelihood (:337-352).
// One wave per parameter vector; lanes stride over the data of each observed filter.
// HBM-bound by design (the curves are read once: B x M x NS doubles): each wave first puts ALL of its sample's curves in
// flight (coalesced, one LDS slab per wave) together with the block's copy of the sample-time and cosmology grids, so that
// the serial parts that follow -- per-sample scalars, bracket searches, finite-node walks -- run on LDS latency
// (`stage_all`; a configuration whose curves do not fit keeps the per-filter copy from global memory).
// =======================================================================================
__device__ __forceinline__ double wave_sum(double v) {
    v = group_sum(v, 64);
    // total sits in the last row; broadcast lane 63
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// em_lc_loglike<G, NM, SD, SA>: the generic tail of the reference path from detector-frame curves on -- sanity_check, autocomplete_data's
// dynamic finite mask, systematics, Gaussian / truncated / upper-limit terms, floor -- for curves another kernel produced or the caller
// supplies (Me2017, combined models).  G lanes per parameter vector (64: a wave per sample; 32 / 16: two / four samples per wave), 256
// threads per workgroup; a sample's terms are added in one order whatever G (group_total_canon).  One wave of the workgroup runs the
// per-sample scalar chains while the other three stage all the workgroup's curves into LDS.  At 8192 rows every workgroup of the launch
// is resident at once, so the kernel's time is a workgroup's chain of phases (HBM-bound staging, then latency and fp64 issue), not a
// throughput: DESIGN 3.3, profiles/r04_config3_tail.md.
// NM: how the sample's curves come about -- 1: set 0 as it is; 2 / 0: the flux sum of two / of n_sets (<= 8) sets, node by node
// (stack_magnitudes, model.py:1486-1510, with its per-model gap filling: lc_stack_node) WHILE the curves are staged into LDS, so
// that a combined model's stacked set is never written to memory and read back (72.5 MB -> 48.6 MB per call at config 3's shape,
// one launch instead of two).  SD / SA: photometry / curves staged in LDS (compile-time, so that the pointers are LDS pointers to
// the compiler).  bad_rows (or NULL): rows whose sub-model delivered no light curve (floor).
#ifdef NMMA_DBG_LC_STAMPS      // measurement builds: cycle stamps of workgroup 300's four waves at the phase boundaries
__device__ unsigned long long g_lc_stamps[4 * 16];
#define LC_STAMP(i) do { if (blockIdx.x == 300 && lane == 0) g_lc_stamps[wave * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LC_STAMP(i) do { } while (0)
#endif
// A sample's sum in ONE order whatever the grouping: virtual lane v = (index of the term) mod 64 adds its terms in increasing
// index order; a group of G lanes carries 64 / G virtual lanes per lane (a[q]: v = lane-in-group + G q); the 64 partial sums are
// added as four DPP rows of 16, then (R0 + R1) + (R2 + R3).  16, 32 or 64 lanes per sample then give the same bits, so the grouping
// can follow the batch size without a row's value depending on the size of its batch.
template <int G>
__device__ __forceinline__ double group_total_canon(const double (&a)[64 / G]) {
    double R0, R1, R2, R3;
    if constexpr (G == 16) {
        R0 = group_sum(a[0], 16); R1 = group_sum(a[1], 16); R2 = group_sum(a[2], 16); R3 = group_sum(a[3], 16);
    } else if constexpr (G == 32) {
        const double s0 = group_sum(a[0], 16), s1 = group_sum(a[1], 16);
        const int base = (int)(threadIdx.x & 32);
        R0 = __shfl(s0, base, 64); R1 = __shfl(s0, base + 16, 64); R2 = __shfl(s1, base, 64); R3 = __shfl(s1, base + 16, 64);
    } else {
        const double s0 = group_sum(a[0], 16);
        R0 = __shfl(s0, 0, 64); R1 = __shfl(s0, 16, 64); R2 = __shfl(s0, 32, 64); R3 = __shfl(s0, 48, 64);
    }
    return (R0 + R1) + (R2 + R3);
}

// The kernel proper for row block `bidx` (4 * 64 / G parameter vectors).  ONLY: only the rows with only_rows[b] != 0 are stored.
template <int G, int NM, bool SD, bool SA, bool ONLY>
__device__ __forceinline__ void em_lc_loglike_body(
    const EmDev* __restrict__ Pp, const double* __restrict__ theta, const long B, const long ld,
    const LcSets sets, const int n_sets, const unsigned char* __restrict__ bad_rows, const int lds_per_sample,
    const int always_floor, double* __restrict__ out, double* __restrict__ chi_parts, double* __restrict__ gp_parts,
    const unsigned char* __restrict__ only_rows, const unsigned bidx) {
    static_assert(G == 64 || G == 32 || G == 16, "a wave per sample, or two / four samples per wave");
    constexpr int SPW = 64 / G;                        // samples per wave
    const EmDev& P = *Pp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1), grp = lane / G;     // lane within the sample's group; the group within the wave
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long b_raw = ((long)bidx * 4 + wave) * SPW + grp;
    const long b = b_raw < B ? b_raw : B - 1;          // (a group beyond the batch recomputes the last row and stores nothing)
    const int NS = P.NS, M = P.M;
    const double* __restrict__ lc = sets.p[0];
    // block LDS: stl[NS] | dist_grid[n_cosmo] | z_grid[n_cosmo] (the cosmology grid only when it fits STAGE_COSMO nodes),
    // then per sample: app[NS] | estacc[nf_max] | praw[8] | scal[8] | curves[M][NS] (stage_all)
    const bool cosmo_lds = P.redshift_mode == NMMA_Z_GRID && P.n_cosmo <= STAGE_COSMO;
    double* stl = reinterpret_cast<double*>(smem);
    double* dgl = stl + NS;
    double* zgl = dgl + (cosmo_lds ? P.n_cosmo : 0);
    // ... | photometry {t, m, sigma, sigma_tot, ln sigma_tot}[n_data] | first work item of each datum | item descriptors (stage_dat):
    // a datum's term then runs on LDS latency -- from global memory every datum was a chain of L2 round trips, and the kernel was
    // bound by that latency at the few waves per CU its LDS slabs allow
    const int grid_bytes = ((NS + (cosmo_lds ? 2 * P.n_cosmo : 0)) * 8 + 15) & ~15;
    const int ND = P.n_data;
    double* pho = reinterpret_cast<double*>(smem + grid_bytes);
    int* ditl = reinterpret_cast<int*>(pho + 5 * ND);
    int* itl = ditl + ((ND + 3) & ~3);
    const int shared_bytes = SD ? ((grid_bytes + 5 * ND * 8 + ((ND + 3) & ~3) * 4 + P.n_items * ITEM_WORDS * 4 + 15) & ~15) : grid_bytes;
    // (SD is a compile-time switch: the staged pointers are LDS pointers to the compiler, not a run-time choice of address space)
    const double* dt_p = SD ? pho : P.dt;
    const double* dm_p = SD ? pho + ND : P.dm;
    const double* dsig_p = SD ? pho + 2 * ND : P.dsig;
    const double* dsigtot_p = SD ? pho + 3 * ND : P.dsigtot;
    const double* dlogsig_p = SD ? pho + 4 * ND : P.dlogsig;
    const int* d_item_p = SD ? ditl : P.d_item;
    const ItemDesc* item_p = SD ? reinterpret_cast<const ItemDesc*>(itl) : P.item_desc;
    double* app = reinterpret_cast<double*>(smem + shared_bytes + (size_t)(wave * SPW + grp) * lds_per_sample);
    double* estacc = app + NS;
    double* praw = estacc + P.lc_nf_max;
    double* scal = praw + 8;
    double* curves = scal + 8;
    (void)estacc; (void)app;
    const double* row = theta + b * ld;
    auto group_count = [&](const bool pred) -> int {   // lanes of THIS sample's group with pred (a ballot: no cross-lane fp64 reduction)
        unsigned long long m = __ballot(pred);
        if constexpr (G < 64) m = (m >> (grp * G)) & ((1ull << G) - 1ull);
        return __popcll(m);
    };
    // (SA -- the sample's curves staged in LDS -- is a compile-time switch like SD: `cur` below is then an LDS pointer to the compiler
    //  and a node costs a ds_read; as a run-time choice between LDS and global memory it was a flat load, several times the latency,
    //  on the serial bracket / finite-node walks of every datum)
    // (two sets: the flux-sum table of lc_stack_node, complete before any curve is stacked.  It is only read while the curves are
    //  staged, so it borrows the LDS of the staged photometry, which is filled afterwards -- a table of its own behind the slabs took
    //  the fourth workgroup per CU away at config 3's shape; behind the last sample's slab when the photometry region is too small)
    // (the cosmology grid is read by one wave only -- the block's scalar chains below -- which stages it itself: no block barrier between)
    // Which wave: the workgroups that share a CU should pick different ones (wave i sits on SIMD i, and the SIMD's issue slots are
    // what the chains cost).  Workgroups go round-robin over the 8 XCDs; k = blockIdx / 8 counts within the XCD, and whether the
    // XCD's 32 CUs are then filled round-robin (k, k + 32, ... share a CU) or one after the other (4c .. 4c + 3), (k + k / 32) & 3
    // differs among the workgroups of a CU.
    const int kx = bidx >> 3;
    const int pro_wave = (kx + (kx >> 5)) & 3;
    LC_STAMP(0);
    // That wave first copies what its chains read from memory into LDS -- the cosmology grid, the theta rows of the block's samples
    // (into each sample's own slab) -- so that the chains themselves wait for LDS and scalar loads only and run WHILE the curve loads
    // issued before them are in flight: all the workgroups of a launch of 8192 rows are resident at once and move in phase, so the
    // chip was either loading curves (HBM-bound) or running the chains (latency-bound), never both.
    constexpr int SPB = 4 * SPW;
    const bool th_lds = NS + P.lc_nf_max >= P.D;
    unsigned char bad_s = 0;
    if (wave == pro_wave) {
        if (cosmo_lds)
            for (int j = lane; j < P.n_cosmo; j += 64) { dgl[j] = P.dist_grid[j]; zgl[j] = P.z_grid[j]; }
        if (th_lds)
            for (int idx = lane; idx < SPB * P.D; idx += 64) {
                const int sx = idx / P.D, c = idx - sx * P.D;
                const long bx = (long)bidx * SPB + sx;
                reinterpret_cast<double*>(smem + shared_bytes + (size_t)sx * lds_per_sample)[c] = theta[(bx < B ? bx : B - 1) * ld + c];
            }
        if (lane < SPB && bad_rows != nullptr) {
            const long bx = (long)bidx * SPB + lane;
            bad_s = bad_rows[bx < B ? bx : B - 1];
        }
    }
    // The per-sample scalar chains (sample_scalars: conversions, z(d_L), distance modulus -- several hundred dependent instructions
    // with ONE useful lane per sample) of ALL the block's samples run side by side on the first lanes of that one wave.
    auto scalar_chains = [&]() {
        if (wave != pro_wave) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // (the LDS copies above, by other lanes of this wave)
        __builtin_amdgcn_wave_barrier();
        if (lane >= SPB) return;
        const long bs_raw = (long)bidx * SPB + lane;
        const long bs = bs_raw < B ? bs_raw : B - 1;
        double* slab_s = reinterpret_cast<double*>(smem + shared_bytes + (size_t)lane * lds_per_sample);
        double* praw_s = slab_s + NS + P.lc_nf_max;
        double* scal_s = praw_s + 8;
        auto chains = [&](const double* row_s) {
#ifdef NMMA_DBG_LC_NOPRO      // (measurement builds: what the per-sample scalar chain costs)
            scal_s[S_ZP1] = 1.0; scal_s[S_IZP1] = 1.0; scal_s[S_TS] = row_s[0] * 1e-30; scal_s[S_DMOD] = 0.0; scal_s[S_RC] = 0.0; scal_s[S_EBV] = 0.0; scal_s[S_BAD] = 0.0;
#else
            double chk;
            if (cosmo_lds) sample_scalars(P, row_s, praw_s, scal_s, chk, dgl, zgl);
            else sample_scalars(P, row_s, praw_s, scal_s, chk);
            for (int o = 0; o < P.O; ++o)         // (sampled time nodes may be non-finite: autocomplete_data masks them)
                if (P.sys_kind[o] != NMMA_SYS_NODES)
                    for (int q = P.sys_off[o]; q < P.sys_off[o + 1]; ++q) chk += apply_slot(P.sys_slots[q], row_s);
            scal_s[S_BAD] = (chk - chk == 0.0) ? 0.0 : 1.0;
            if (bad_s != 0) scal_s[S_BAD] = 1.0;
#endif
        };
        if (th_lds) chains(slab_s);            // (two inlined copies: the row is an LDS pointer to the compiler in this one)
        else chains(theta + bs * ld);
    };
    const bool tab_in_pho = SD && shared_bytes - grid_bytes >= STACK2_LDS_BYTES;
    double* tab2 = tab_in_pho ? pho : reinterpret_cast<double*>(smem + shared_bytes + (size_t)(4 * SPW) * lds_per_sample);
    if constexpr (SA && NM == 2) {
        stack2_stage(tab2, threadIdx.x, 256);
        __syncthreads();
    }
    (void)tab2; (void)tab_in_pho;
    if constexpr (SA) {
        // ... and the other three waves meanwhile stage the curves of ALL the block's samples (stacking two sets on the way), every
        // load of a trip issued before the first is used.  Work item n = sample * NNP + node, NNP = the nodes of a sample rounded up
        // to whole waves; the three waves take items 192 apart.
        if (wave == pro_wave) {
            scalar_chains();
        } else {
            const int NN = M * NS, NNP = (NN + 63) & ~63, n_items = SPB * NNP;
            const float inv_nnp = 1.0f / (float)NNP;
            const int sl = ((((wave - pro_wave) & 3) - 1) << 6) + lane;
            constexpr int KM = NM > 0 ? NM : 8;
            const int n_models = NM > 0 ? NM : n_sets;
            constexpr int NPT = NM == 0 ? 2 : 8;       // items per lane and trip (12 or 16 -- more bytes in flight -- measured slower: 38.4 against 35.4 us)
            const int curves_off = NS + P.lc_nf_max + 16;
            const long blk_base = (long)bidx * SPB * NN;
            const long left = B - (long)bidx * SPB;
            const int n_own = left < SPB ? (int)left : SPB;            // samples of this block inside the batch (>= 1)
            for (int n0 = sl; n0 < n_items; n0 += NPT * 192) {
                double v[NPT][KM];
                unsigned slow = 0u;
                (void)slow;
                // (addresses as a uniform base per set -- the block's first sample -- plus a 32-bit offset: ONE register per item; as
                //  64-bit addresses per item and set they took the registers the loads in flight need)
#pragma unroll
                for (int i = 0; i < NPT; ++i) {
                    const int n = n0 + i * 192;
                    const int sx = (int)(((float)n + 0.5f) * inv_nnp), j = n - sx * NNP;      // (exact: n < 2^20)
                    const int sxe = sx < n_own ? sx : n_own - 1;           // (a sample beyond the batch re-reads the last row)
                    const unsigned off = (unsigned)(sxe * NN + j);
                    const bool ok = n < n_items && j < NN;
#pragma unroll
                    for (int k = 0; k < KM; ++k) v[i][k] = (ok && k < n_models) ? (sets.p[k] + blk_base)[off] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < NPT; ++i) {
                    const int n = n0 + i * 192;
                    const int sx = (int)(((float)n + 0.5f) * inv_nnp), j = n - sx * NNP;
                    const long bx = (long)bidx * SPB + sx;
                    const long g = (bx < B ? bx : B - 1) * NN + j;
                    if (n < n_items && j < NN) {
                        double* dst = reinterpret_cast<double*>(smem + shared_bytes + (size_t)sx * lds_per_sample) + curves_off + j;
                        if constexpr (NM == 1) *dst = v[i][0];
                        else if constexpr (NM == 2) {
                            double r;
                            if (stack2_fast(v[i][0], v[i][1], tab2, r)) *dst = r;
                            else slow |= 1u << i;
                        } else *dst = lc_stack_node<KM>(P, sets, n_models, g, v[i]);
                    }
                }
                if constexpr (NM == 2) {               // (the few nodes with a gap to fill: one copy of the general code, values re-read)
#pragma nounroll
                    for (int i = 0; (slow >> i) != 0u; ++i) {
                        if (((slow >> i) & 1u) == 0u) continue;
                        const int n = n0 + i * 192;
                        const int sx = (int)(((float)n + 0.5f) * inv_nnp), j = n - sx * NNP;
                        const long bx = (long)bidx * SPB + sx;
                        const long g = (bx < B ? bx : B - 1) * NN + j;
                        const double vv[2] = {sets.p[0][g], sets.p[1][g]};
                        reinterpret_cast<double*>(smem + shared_bytes + (size_t)sx * lds_per_sample)[curves_off + j] = lc_stack_node<2>(P, sets, 2, g, vv, tab2);
                    }
                }
            }
        }
    } else {
        scalar_chains();
    }
    LC_STAMP(1);
    for (int j = threadIdx.x; j < NS; j += 256) stl[j] = P.st[j];
    if constexpr (SA && NM == 2) {
        if (tab_in_pho) __syncthreads();               // every wave is done with the table before the photometry overwrites it
    }
    if constexpr (SD) {
        for (int j = threadIdx.x; j < ND; j += 256) {
            pho[j] = P.dt[j]; pho[ND + j] = P.dm[j]; pho[2 * ND + j] = P.dsig[j]; pho[3 * ND + j] = P.dsigtot[j]; pho[4 * ND + j] = P.dlogsig[j];
            ditl[j] = P.d_item[j];
        }
        const int* src = reinterpret_cast<const int*>(P.item_desc);
        for (int j = threadIdx.x; j < P.n_items * ITEM_WORDS; j += 256) itl[j] = src[j];
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    LC_STAMP(2);
    const double zp1 = scal[S_ZP1], tsh = scal[S_TS], dmod = scal[S_DMOD], rc = scal[S_RC], ebv = scal[S_EBV], izp1 = scal[S_IZP1];
    bool bad = always_floor != 0 || scal[S_BAD] != 0.0;
    const bool st_uniform = P.st_uniform != 0;
    const double st0 = P.st0, st_inv_dt = P.st_inv_dt;

    // sanity_check over ALL model filters: fewer than 2 finite magnitudes -> all-inf -> floor
    // (four filters per trip, their reads issued before the first ballot: filter by filter the loop was a chain of LDS round trips,
    //  2 of the 35 us at config 3's shape)
#ifdef NMMA_DBG_LC_NOSANITY
    for (int m0 = 0; m0 < 0; m0 += 4) {
#else
    for (int m0 = 0; m0 < M; m0 += 4) {
#endif
        int nfin[4] = {0, 0, 0, 0};                    // (a ballot per G nodes: no cross-lane fp64 reduction for a count)
        for (int j0 = 0; j0 < NS; j0 += G) {
            const int j = j0 + gl;
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + q;
                const double* cur = SA ? curves + m * NS : lc + ((size_t)b * M + m) * NS;
                v[q] = HUGE_VAL;
                if (m < M && j < NS) v[q] = cur[j];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) nfin[q] += group_count(v[q] - v[q] == 0.0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (m0 + q < M && nfin[q] < 2) bad = true;
    }

    LC_STAMP(3);
    // One datum: interpolate every source curve of the datum's band at its epoch, average, likelihood term.
    // k0 = first work item of the band (its sources are consecutive items); the bracket depends on the epoch only.
    auto datum_term = [&](const int di, const int k0, double& chi, double& gp) {
        const ItemDesc& it0 = item_p[k0];
        const int o = it0.o, nsrc = it0.nsrc, kind = it0.kind;
        const double lim = it0.lim, e_const = it0.e_const;
        const double t = dt_p[di];
        int lo = -1;                                   // t_obs[lo] <= t (<= t_obs[NS - 1]); -1: outside the grid
        if (t == t && NS >= 1 && t >= stl[0] * zp1 + tsh && t <= stl[NS - 1] * zp1 + tsh) {
            if (st_uniform) {
                // equally spaced sample_times: the index guess, then the exact test np.interp's bracket obeys (t_obs[lo] <= t, and
                // t_obs[lo + 1] > t unless lo is the last node) -- the guess is off by at most one node from rounding
                lo = (int)(((t - tsh) * izp1 - st0) * st_inv_dt);      // (a guess: the reciprocal's rounding is corrected below)
                lo = lo < 0 ? 0 : (lo > NS - 1 ? NS - 1 : lo);
                while (lo > 0 && stl[lo] * zp1 + tsh > t) --lo;
                while (lo < NS - 1 && stl[lo + 1] * zp1 + tsh <= t) ++lo;
            } else {
                int hi = NS - 1;
                lo = 0;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (stl[mid] * zp1 + tsh <= t) lo = mid; else hi = mid;
                }
                if (stl[hi] * zp1 + tsh <= t) lo = hi;      // t on the last node
            }
        }
        double acc_e = 0.0;
        for (int ks = 0; ks < nsrc; ++ks) {
            const ItemDesc& it = item_p[k0 + ks];
            const double ext = extinction_mag(P.ext_law, it.ebvc, zp1, ebv);
            // detector-frame curve of this source (model.py:390-397); non-finite stays non-finite
            const double* cur = SA ? curves + it.m * NS : lc + ((size_t)b * M + it.m) * NS;
            auto app = [&](const int j) { double v = cur[j]; if (ext != 0.0) v = v + ext; return (v + dmod) + rc; };
            // np.interp over the FINITE nodes only, left = right = +inf (utils.py:634-645)
            double est = dinf();
            if (t != t) {
                est = t;
            } else if (lo >= 0) {
                int jl = lo;                                  // nearest finite node at or left of t
                while (jl >= 0 && !(cur[jl] - cur[jl] == 0.0)) --jl;
                int jr = lo + 1;                              // nearest finite node right of t
                while (jr < NS && !(cur[jr] - cur[jr] == 0.0)) ++jr;
                if (jl >= 0) {
                    const double x0 = stl[jl] * zp1 + tsh;
                    if (x0 == t) est = app(jl);
                    else if (jr < NS) est = lerp_np(t, x0, stl[jr] * zp1 + tsh, app(jl), app(jr));
                }
            }
            acc_e = ks == 0 ? est : acc_e + est;     // averaged band: (a + b [+ c]) / n  (utils.py:566-584)
        }
        const double est = nsrc > 1 ? acc_e / (double)nsrc : acc_e;
        const double sd = dsig_p[di];
        double e = e_const, sig, lsig;
        if (kind == NMMA_SYS_CONST) {
            sig = dsigtot_p[di]; lsig = dlogsig_p[di];
        } else {
            const nmma_slot* sv = P.sys_slots + P.sys_off[o];
            if (kind == NMMA_SYS_PARAM) {
                e = apply_slot(sv[0], row);
            } else {
                const int K = P.sys_nn[o];
                const int ni = P.sys_nidx[di];
                if (ni < 0) e = apply_slot(sv[0], row);
                else if (ni >= K - 1) e = apply_slot(sv[K - 1], row);
                else {
                    const double v0 = apply_slot(sv[ni], row), v1 = apply_slot(sv[ni + 1], row);
                    e = ((v1 - v0) / P.sys_ndx[di]) * P.sys_noff[di] + v0;
                }
                if (!(e - e == 0.0)) e = masked_nodes_at(sv, P.sys_node_t + P.sys_off[o], K, P.dt[di], row);      // (finite mask on the nodes)
            }
            sig = sqrt(sd * sd + e * e);
            lsig = log(sig);
        }
        const double mobs = dm_p[di];
        if (sig - sig == 0.0) chi += detection_term_tab(mobs, est, sig, lsig, lim, static_cast<const double*>(kLogPhiTab));
        // (log Phi from the polynomial table of logphi_tab.h, read from global memory here -- 2 KB, cache-resident: the out-of-line
        //  scipy formula was most of the 4.9 us this kernel spent on ONE upper limit per sample, profiles/r04_config3_tail.md)
        else gp += upper_limit_term_tab(mobs, est, e, static_cast<const double*>(kLogPhiTab));       // (inline: out of line it costs this kernel 1.8 of the 3.5 us the table saves)
    };

    double chi_tot = 0.0, gp_tot = 0.0;
    if (chi_parts == nullptr) {
        // all photometry points of all bands in one pass over the lanes (a band with a dozen points would otherwise leave
        // most of the wave idle for a whole pass): d_item[di] = first work item of the datum's band
        constexpr int NA = 64 / G;                     // virtual lanes per lane (group_total_canon)
        double chi_a[NA], gp_a[NA];
#pragma unroll
        for (int q = 0; q < NA; ++q) { chi_a[q] = 0.0; gp_a[q] = 0.0; }
        auto add_chi = [&](const int h, const double c) {            // (h: the trip's index mod NA -- uniform over the wave)
            if constexpr (NA == 1) chi_a[0] += c;
            else {
#pragma unroll
                for (int q = 0; q < NA; ++q)
                    if (h == q) chi_a[q] += c;
            }
        };
        auto add_gp = [&](const int h, const double g) {
            if constexpr (NA == 1) gp_a[0] += g;
            else {
#pragma unroll
                for (int q = 0; q < NA; ++q)
                    if (h == q) gp_a[q] += g;
            }
        };
        auto general_term = [&](const int h, const int di) {         // one datum through the general term, into virtual lane h
            double c = 0.0, g = 0.0;
            datum_term(di, d_item_p[di], c, g);
            add_chi(h, c);
            add_gp(h, g);
        };
        // FAST LANE (compile-time staged data only).  The common datum -- a detection inside the model window whose band has one
        // source, a constant systematic, no finite limit, no extinction, and whose two bracket nodes are finite -- needs none of
        // the general term's machinery (source loop, finite-node walks, systematics kinds, truncation mass, upper limits: ~3 000
        // instructions of code of which a lane executes a few hundred).  It is evaluated here with the SAME operations in the same
        // order (lerp_np's quotient, the residual's quotient, scipy's expression of the Gaussian term), so a lane's sum is bit for
        // bit the general term's; lanes that do not qualify -- an upper limit, a non-finite node, an epoch outside the window --
        // take the general term, and the branch is skipped when no lane of the wave needs it.
        constexpr bool FASTLANE = SD && SA;
        const bool fast_cfg = FASTLANE && !P.has_ebv && NS >= 2;
        const double t_first = stl[0] * zp1 + tsh, t_last = stl[NS - 1] * zp1 + tsh;
        int* glist = reinterpret_cast<int*>(app);      // (the sample's first NS + lc_nf_max doubles of LDS: not used otherwise)
        const int gcap = 2 * (NS + P.lc_nf_max);
        int n_gen = 0;
        // ONE loop, two passes, so that the general term (~3 000 instructions of code) is inlined ONCE: pass 0 walks the data (fast lane;
        // what it turns away is queued, or -- queue full, or no fast lane for this configuration -- evaluated in place), pass 1 the
        // queue.  (Three inlined copies -- in place, overflow, queue -- cost 1 us of instruction fetch at config 3's shape.)
        const int cap_eff = fast_cfg ? gcap : 0;
#ifdef NMMA_DBG_LC_NODATA
        const int nd_eff = ND > 100000 ? ND : 0;
#else
        const int nd_eff = ND;
#endif
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                LC_STAMP(4);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the queue: written by other lanes of this wave)
                __builtin_amdgcn_wave_barrier();
#ifdef NMMA_DBG_LC_NOGENERAL
                n_gen = 0;
#endif
            }
            const int n_pass = pass == 0 ? nd_eff : n_gen;
            for (int d0 = 0; d0 < n_pass; d0 += G) {
                const int h = (d0 / G) & (NA - 1);
                int di = d0 + gl;
                bool general = di < n_pass;
                if (pass == 1) {
                    di = general ? glist[di] : 0;      // (the queue is in increasing datum order for every grouping)
                } else if (fast_cfg && general) {
                    const ItemDesc& it0 = item_p[d_item_p[di]];
                    const double t = dt_p[di], sig = dsigtot_p[di];
                    if (it0.nsrc == 1 && it0.kind == NMMA_SYS_CONST && it0.lim == dinf() && t >= t_first && t < t_last && sig > 0.0 && sig < dinf()) {
                        int lo;
                        if (st_uniform) {
                            lo = (int)(((t - tsh) * izp1 - st0) * st_inv_dt);
                            lo = lo < 0 ? 0 : (lo > NS - 2 ? NS - 2 : lo);
                            while (lo > 0 && stl[lo] * zp1 + tsh > t) --lo;
                            while (lo < NS - 2 && stl[lo + 1] * zp1 + tsh <= t) ++lo;
                        } else {
                            int hi = NS - 1;
                            lo = 0;
                            while (hi - lo > 1) {
                                const int mid = (lo + hi) >> 1;
                                if (stl[mid] * zp1 + tsh <= t) lo = mid; else hi = mid;
                            }
                        }
                        const double* cur = curves + it0.m * NS;
                        const double y0 = cur[lo], y1 = cur[lo + 1];
                        if ((y0 - y0 == 0.0) && (y1 - y1 == 0.0)) {
                            const double x0 = stl[lo] * zp1 + tsh, x1 = stl[lo + 1] * zp1 + tsh;
                            const double est = (x0 == t) ? (y0 + dmod) + rc : lerp_np(t, x0, x1, (y0 + dmod) + rc, (y1 + dmod) + rc);
                            const double x = (dm_p[di] - est) / sig;
                            if (est < dinf() && x == x) {
                                add_chi(h, ((-(x * x) / 2.0 - kNormPdfLogC) - 0.0) - dlogsig_p[di]);
                                general = false;
                            }
                        }
                    }
                }
                // The data the fast lane turned away (upper limits, epochs outside the window, non-finite nodes: a few per sample) are
                // queued in the sample's LDS and take the general term densely packed in pass 1: called in place, one such lane made
                // its whole wave walk the general term in every trip.  (The fast lane is bound by the fp64 issue rate of the SIMD --
                // four waves x ~150 instructions a trip -- not by the latency of a trip: two data per lane and trip in a branch-free
                // form, more instructions for shorter chains, took 38.7 instead of 35.1 us at config 3's shape.  A queue of one datum
                // per sample -- the single upper limit of config 3's data set -- costs 5 of the 35 us: the latency of one chain
                // through the general term and log_ndtr; handing the queues of all the block's samples to ONE wave, a lane per sample,
                // did not change the time.)
                bool in_place = general;
                if (pass == 0) {
                    unsigned long long gm = __ballot(general);
                    if constexpr (G < 64) gm = (gm >> (grp * G)) & ((1ull << G) - 1ull);
                    const int slot = n_gen + __popcll(gm & ((1ull << gl) - 1ull));
                    if (general && slot < cap_eff) { glist[slot] = di; in_place = false; }
                    n_gen += __popcll(gm);
                    n_gen = n_gen > cap_eff ? cap_eff : n_gen;
                }
                if (in_place) general_term(h, di);
            }
        }
        LC_STAMP(5);
        chi_tot = group_total_canon<G>(chi_a);
        gp_tot = group_total_canon<G>(gp_a);
    } else {
        // per-filter parts requested: one pass per band
        for (int k = 0; k < P.n_items; ++k) {
            const ItemDesc& it = item_p[k];
            if (it.ks != 0) continue;
            constexpr int NA = 64 / G;
            double chi_a[NA], gp_a[NA];
#pragma unroll
            for (int q = 0; q < NA; ++q) { chi_a[q] = 0.0; gp_a[q] = 0.0; }
            for (int d0 = 0; d0 < it.nf; d0 += G) {
                double c = 0.0, g = 0.0;
                if (d0 + gl < it.nf) datum_term(it.d0 + d0 + gl, k, c, g);
                const int h = (d0 / G) & (NA - 1);
#pragma unroll
                for (int q = 0; q < NA; ++q)
                    if (h == q) { chi_a[q] += c; gp_a[q] += g; }
            }
            const double chi = group_total_canon<G>(chi_a);
            const double gp = group_total_canon<G>(gp_a);
            chi_tot += chi;
            gp_tot += gp;
            if (gl == 0 && b_raw < B) {
                chi_parts[(long)it.o * B + b] = chi;
                gp_parts[(long)it.o * B + b] = gp;
            }
        }
    }
    LC_STAMP(6);
    if (gl == 0 && b_raw < B && (!ONLY || only_rows[b] != 0)) {
        double tot = chi_tot + gp_tot;
        if (bad || !(tot - tot == 0.0)) tot = NMMA_LOGL_FLOOR;      // (a NaN term of any band makes the total NaN)
        out[b] = tot;
    }
}

// em_lc_loglike<G, NM, SD, SA>: one workgroup per row block.
template <int G, int NM, bool SD, bool SA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void em_lc_loglike(
    const EmDev* __restrict__ Pp, const double* __restrict__ theta, const long B, const long ld,
    const LcSets sets, const int n_sets, const unsigned char* __restrict__ bad_rows, const int lds_per_sample,
    const int always_floor, double* __restrict__ out, double* __restrict__ chi_parts, double* __restrict__ gp_parts) {
    em_lc_loglike_body<G, NM, SD, SA, false>(Pp, theta, B, ld, sets, n_sets, bad_rows, lds_per_sample, always_floor, out, chi_parts, gp_parts,
                                             nullptr, blockIdx.x);
}

// stack2_redo<KP, G, SD>: ONE launch that re-evaluates the rows the combined model's one-launch kernel flagged (em_logl<.., 7 | 8>,
// nmma_em_loglike_stack2: rows that met an interior gap of the second transient's curve) with the kernels of the materialising path
// -- per flagged tile of 32 rows a workgroup runs em_fused<MODE_LC_ABS> for every model filter (the surrogate's curves, into kn_ws),
// then em_lc_loglike on {kn_ws, lc2} for the tile's row blocks, storing the flagged rows only: their values are the materialising
// path's, bit for bit.  A small grid: every workgroup looks at each gridDim.x-th tile's flags (64 tiles per load round), so that
// with nothing flagged -- the usual case -- the launch is little more than an empty kernel.  (Two restricted launches, one per
// kernel, cost 5 us EACH at config 3's shape whatever their grid: dependent launches pay the inter-kernel latency.)

// Launcher of one variant (em_api.inc: launch_lc_loglike picks the variant).  Declared in em_host.h; instantiated in em_lc.hip.
template <int G, int NM, bool SD, bool SA>
int launch_lc_loglike_sd(nmma_em_handle* h, const double* theta, int64_t B, int64_t ld, const LcSets& sets, int n_sets,
                         const unsigned char* bad_rows, int per_sample, int lds, double* out, double* chi, double* gp, hipStream_t s) {
    constexpr int SPB = 4 * (64 / G);       // samples per 256-thread block
    if (lds > 64 * 1024)
        NM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&em_lc_loglike<G, NM, SD, SA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    h->g_x = (int)((B + SPB - 1) / SPB); h->g_y = 1; h->g_block = 256; h->g_tile = SPB; h->g_lds = lds;
    hipLaunchKernelGGL((em_lc_loglike<G, NM, SD, SA>), dim3((unsigned)((B + SPB - 1) / SPB)), dim3(256), lds, s, h->dev_d, theta, (long)B, (long)ld, sets,
                       n_sets, bad_rows, per_sample, h->always_floor, out, chi, gp);
    NM_HIP(hipGetLastError());
    return 0;
}

// The variants that exist (launch_lc_loglike_as, em_api.inc): curves staged in LDS (SA) for every form; the unstaged fallback only for
// one materialised set and a wave per sample; the 16- and 32-lane groups only with the photometry staged (SD).
// (two translation units: em_lc.hip holds the sub-wave groups, em_lc64.hip the wave-per-sample forms -- as one unit they were the
//  longest single compile of the library)
#define NMMA_LC_VARIANTS_SUBWAVE(X)                                                       \
    X(16, 0, true, true) X(16, 1, true, true) X(16, 2, true, true)                        \
    X(32, 0, true, true) X(32, 1, true, true) X(32, 2, true, true)
#define NMMA_LC_VARIANTS_WAVE(X)                                                          \
    X(64, 0, true, true) X(64, 1, true, true) X(64, 2, true, true)                        \
    X(64, 0, false, true) X(64, 1, false, true) X(64, 2, false, true)                     \
    X(64, 1, true, false) X(64, 1, false, false)
#define NMMA_LC_VARIANTS(X) NMMA_LC_VARIANTS_SUBWAVE(X) NMMA_LC_VARIANTS_WAVE(X)
#define NMMA_LC_SIGNATURE(G, NM, SD, SA)                                                                                                  \
    launch_lc_loglike_sd<G, NM, SD, SA>(nmma_em_handle*, const double*, int64_t, int64_t, const LcSets&, int, const unsigned char*, int, \
                                        int, double*, double*, double*, hipStream_t)
#ifndef NMMA_LC_INSTANTIATE
#define NMMA_LC_EXTERN(G, NM, SD, SA) extern template int NMMA_LC_SIGNATURE(G, NM, SD, SA);
NMMA_LC_VARIANTS(NMMA_LC_EXTERN)
#undef NMMA_LC_EXTERN
#endif

}  // namespace nmma
