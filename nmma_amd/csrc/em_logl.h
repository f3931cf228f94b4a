// em_logl.h -- the hot-path kernel em_logl<R, KP, NMW, NVW, FASTM, WALKF> (overview: em_common.h) and its launcher.  Included by
// the em_logl_*.hip units, each of which instantiates launch_logl_one for some task flavours (NMMA_LOGL_INSTANCE).
#pragma once
#include "em_host.h"

namespace nmma {

// FASTM = 0: generic item phase; 1: every work item qualifies for the basic fast task (constant systematics,
// at most 2 G points per filter); 2: extended fast task (sampled systematics per datum, any number of points) --
// separate instantiations so that the extensions cost the basic configuration nothing.
// Fast modes: every work item qualifies for the fast path (EmDev::all_fast) -- the generic item phase and its
// LDS table staging are not compiled in, which keeps the register budget small enough for 16-wave workgroups.
// WALKF: the MCMC step fused in (nmma_em_loglike_walk) -- the first likelihood wave, which sums the tile's log L, also runs the accept
// of the walk's step `wstep` and the proposal of the next one for the tile's chains (walk_device.h: the device functions of
// walk_step_kernel, same arithmetic) and writes the tile's theta rows for the next launch.  Its own instantiations: the walk code
// must not touch the register allocation of the tuned flavours.
template <int R, int KP, int NMW, int NVW, int FASTM, int WALKF>
__global__ __launch_bounds__(logl_threads(NMW, NVW), (NMW + NVW) / 4) void em_logl(
    const EmDev* __restrict__ Pp, const double* __restrict__ theta, const long B, const long ld, const LdsW L,
    const int always_floor, double* __restrict__ out, double* __restrict__ chi_parts, double* __restrict__ gp_parts,
    long long* __restrict__ dbg, const nmma_walk_fuse* __restrict__ wf, const unsigned long long wstep, const int wlast, const typename em_aux_of<FASTM>::type aux) {
    constexpr int TS = 16 * R;
    constexpr int PF = (R == 1) ? 8 : 4;
    constexpr bool FAST = FASTM != 0, EXT = FASTM == 2;
    // FASTM == 3: the lean task with its extras compiled in (filters with more than 32 points, a sampled em_syserr); the
    // plain lean kernel (FASTM == 1, BASELINE config 2 and the CLI grid) does not carry them: they cost it 2 % when present
#ifdef NMMA_DBG_NO_SPLIT      // measurement build: the kernel without the band-split forms of its epilogue (what their code costs the full-batch launch)
    constexpr bool SPLITTABLE = false;
#else
    constexpr bool SPLITTABLE = R == 1 && FASTM != 0 && FASTM != 2 && FASTM != 7 && FASTM != 8;    // small batches: one band per workgroup (launch_logl_one)
#endif
    constexpr bool DENSE = FASTM == 6 || FASTM == 9;       // lean task that reconstructs ALL nodes of (item, 16 samples) on the fp64 matrix cores (many points per filter)
    // FASTM == 9: the dense task's ONE variant BASELINE config 4 takes -- constant systematics, equally spaced sample_times -- alone in its
    // kernel: with the other three variants (sampled systematic, unequally spaced grid) inlined next to it, code it never runs cost that
    // variant 3 % (134.2 against 130.5 us at 8192 rows, 964 against 939 at 65 536: tools/experiments/ab_dense_variants.sh)
    constexpr bool LEANX = FASTM >= 3;       // 3: equally spaced sample_times, 4: unequally spaced (fewer inlined variants per kernel)
    // FASTM == 7: the lean task with extras (as 3) for a COMBINED model of two transients that share sample_times and filters
    // (CombinedLightCurveModelContainer, model.py:1411-1459 -- what the reference's drivers build, :1591-1614): the second transient's
    // source-frame curves aux.lc2[B][M][NS] are an operand; every datum loads its two bracket nodes of them and the flux sum
    // (stack_magnitudes, :1486-1510) is formed on those two nodes only, next to the kilonova's two reconstructed nodes -- the
    // kilonova's curves are never written out (em_fused<MODE_LC_ABS> + em_lc_loglike: two launches and 24 + 48 MB of traffic at config 3's shape)
    constexpr bool COMB = FASTM == 7 || FASTM == 8;      // (8: the same on unequally spaced sample_times, as FASTM 4)
    constexpr int NV = 64 * NVW;      // VALU-role threads

    // (blockIdx.y > 0 only in the split launch of small batches: one copy of the configuration per observed band)
    const EmDev& P = Pp[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* praw = reinterpret_cast<double*>(smem + L.praw);
    double* scal = reinterpret_cast<double*>(smem + L.scal);
    double* stl = reinterpret_cast<double*>(smem + L.stl);
    float* part = reinterpret_cast<float*>(smem + L.part);
    double* chi_tot = reinterpret_cast<double*>(smem + L.chi);
    double* gp_tot = reinterpret_cast<double*>(smem + L.gp);
    int* bad = reinterpret_cast<int*>(smem + L.bad);
    double* estb = reinterpret_cast<double*>(smem + L.est);
    unsigned char* tabl = smem + L.tab;
    double* cdl = reinterpret_cast<double*>(smem + L.cdl);
    const ItemDesc* itab = reinterpret_cast<const ItemDesc*>(smem + L.itab);
    int* sync = reinterpret_cast<int*>(smem + L.sync);
    double* xnl = reinterpret_cast<double*>(smem + L.xn);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NBUF = L.nbuf;
    for (int j = tid; j < 7 * P.n_items + 5 + (WALKF ? 1 : 0); j += logl_threads(NMW, NVW)) sync[j] = 0;      // (the last two: fused MCMC step -- first phase done | the tile's totals parked)
    if (tid == 0) g_wd_trip = 0;
    __syncthreads();             // the only workgroup barrier: counters zeroed
    const long tile0 = (long)blockIdx.x * TS;
    const int NP = P.NP, NC = P.NC, NT = P.NT, NS = P.NS;
    const int W = P.n_items;
    gci32p items = as_global(P.items);
    // (WALKF = 8 / 16 [+ 64]: the fused MCMC step keeps a group of 8 / 16 lanes per chain -- one per sampled dimension -- and 16-sample
    //  tiles; + 64: the chains' Constraint program is evaluated in the step, on an fp64 stack in LDS)
    constexpr int WT = WALKF ? (WALKF & 31) : 8;      // lanes per chain (8: up to 8 sampled dimensions, 16: up to 16)
    constexpr bool WCON = (WALKF & 64) != 0;
    constexpr int WCR = 64 / WT;                      // chains per round of one wave
    constexpr int WNR = WALKF ? TS / WCR : 1;         // rounds per tile: the second phase of the step gives each to a likelihood wave of its own
    static_assert(WALKF == 0 || ((WT == 8 || WT == 16) && WNR <= NVW), "lanes per chain of the fused MCMC step");
    constexpr int WPH = WNR > 2 ? 2 : 1;              // likelihood waves that share the step's first phase (two rounds each)
    // (staging area after the prologue: tot[TS] | parked walk state 5 x WNR x 64 + 2 TS doubles + 6 TS ints | prior table)
    nmma_walk_prior* wspl = reinterpret_cast<nmma_walk_prior*>(reinterpret_cast<double*>(smem + L.stage) + TS + 5 * WNR * 64 + 2 * TS + 3 * TS);

    if (wave < NMW) {
        // ============================ MFMA role ============================
        // layer-1 operands straight from theta (no dependency on the other role's prologue)
        double xraw[R][KP];
        auto fill_xraw = [&]() {
#pragma unroll
            for (int rb = 0; rb < R; ++rb) {
                long b = tile0 + rb * 16 + (lane & 15);
                if (b >= B) b = B - 1;
                const double* row = theta + b * ld;
#pragma unroll
                for (int kp = 0; kp < KP; ++kp) {
                    const int p = 4 * kp + (lane >> 4);
                    xraw[rb][kp] = (p < NP) ? apply_slot(P.model_param[p], row) : 0.0;
                }
            }
        };
#ifndef NMMA_DBG_PRELOAD_FIRST
        fill_xraw();
#endif
        switch (P.prio_mfma) {
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            case 3: __builtin_amdgcn_s_setprio(3); break;
            default: break;
        }
#ifdef NMMA_DBG_PRELOAD_FIRST
        mfma_role<R, KP, PF, NMW, NVW, FAST, COMB>(P, xraw, xnl, wave, lane, part, L.nbuf, sync, dbg, fill_xraw);
#else
        mfma_role<R, KP, PF, NMW, NVW, FAST, COMB>(P, xraw, xnl, wave, lane, part, L.nbuf, sync, dbg, [] {});
#endif
        if (!FAST || !P.helpers) return;
        // fast mode: the record stream is done -- join the likelihood workers for the remaining tasks
    }

    // ================================ VALU role ================================
    // Few instructions, long dependency chains: give them the issue slot whenever they are
    // ready; the MFMA waves (lower priority) soak up every other cycle of the SIMD.
    switch (P.prio_valu) {
        case 0: break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
    const bool helper = wave < NMW;              // an MFMA-role wave that finished its stream
    const int vt = tid - 64 * NMW;               // negative for helpers (they skip the prologue)
    const int vwave = wave - NMW;
    // Lean task with extinction: ext_mag[item][sample] of this tile in LDS, filled by the prologue lane that owns the sample's
    // E(B-V) -- coefficient x E(B-V) for the linear law, the pre-pass kernel's value for the Pei-1992 law -- so that a task reads
    // ONE LDS word per slot (anything more inside the task tips hipcc into spilling: see the register-budget test)
    auto fill_ext = [&](const int s_l, const double ebv, const double zp1) {
        double* et = reinterpret_cast<double*>(smem + L.exttab);
        if (!P.has_ebv) { et[s_l] = 0.0; return; }        // one row of zeros that every item reads
        long bb = tile0 + s_l;
        if (bb >= B) bb = B - 1;
        for (int kk = 0; kk < W; ++kk) {
            const int m_k = P.item_desc[kk].m;
            if (P.p92_cheb != nullptr)       // Pei 1992: magnitude per unit E(B-V) from the filter's series in the redshift (no pre-pass launch)
                et[kk * TS + s_l] = (ebv != 0.0) ? cheb14_eval(P.p92_cheb + m_k * 16, zp1 - 1.0) * ebv : 0.0;
            else
                et[kk * TS + s_l] = P.p92_tab ? P.ext_tab[bb * P.M + m_k] : ((ebv != 0.0) ? P.item_desc[kk].ebvc * ebv : 0.0);
        }
    };
    if (!helper) {
        if (dbg && blockIdx.x == 0 && vt == 0) dbg[64] = clock64();
        // ---- prologue: per-sample scalars, accumulators, sample-time grid.
        // The tile's theta rows and the cosmology grid are first staged in LDS by all likelihood waves
        // (ONE memory round trip), so the serial slot/interpolation chain of sample_scalars runs on
        // LDS latency instead of ~40 dependent L2/HBM round trips.
        const bool staged = (ld <= STAGE_COLS) && (P.redshift_mode != NMMA_Z_GRID || P.n_cosmo <= STAGE_COSMO);
        double* thl = reinterpret_cast<double*>(smem + L.stage);
        double* dgl = thl + TS * STAGE_COLS;
        double* zgl = dgl + STAGE_COSMO;
        if (staged) {
            const int ncol = (int)ld;
            for (int idx = vt; idx < TS * ncol; idx += NV) {
                const int sidx = idx / ncol, cidx = idx - sidx * ncol;
                long b = tile0 + sidx;
                if (b >= B) b = B - 1;
                thl[sidx * ncol + cidx] = theta[b * ld + cidx];
            }
            if (P.redshift_mode == NMMA_Z_GRID)
                for (int j = vt; j < P.n_cosmo; j += NV) { dgl[j] = P.dist_grid[j]; zgl[j] = P.z_grid[j]; }
            sync_signal(sync + 2 * W + 2, lane);
        }
        int* badp = bad + TS;        // [4][TS]: non-finite input seen by prologue part 0..3 for sample s
        if (staged) {
            // four waves share the per-sample chain (a divergent split inside one wave would execute all parts
            // one after the other): 0 = redshift branch (grid interpolation, log10), 1 = distance modulus, time
            // shift, E(B-V), 2 = model parameters, 3 = systematics parameters.  Lane = sample.
            if (vwave < 4 && lane < TS) {
                sync_wait(sync + 2 * W + 2, NVW, P.watchdog, 200);
                // (LDS-address-space pointers: ds_read instead of flat_load for the staged rows and grids)
                typedef const __attribute__((address_space(3))) double* lds_cdp;
                const lds_cdp row = (lds_cdp)(thl + lane * (int)ld);
                const lds_cdp dgl_l = (lds_cdp)dgl, zgl_l = (lds_cdp)zgl;
                double* sc = scal + lane * 8;
                double chk = 0.0;
#ifdef NMMA_DBG_NOCHAINS
                // measurement build: what a pre-pass kernel for the per-sample scalars could save AT MOST -- the chains are gone,
                // plausible constants stand in (the tasks run the same instruction stream on wrong numbers)
                if (vwave == 0) { sc[S_ZP1] = 1.0093; sc[S_IZP1] = 1.0 / 1.0093; sc[S_RC] = -0.01; }
                else if (vwave == 1) { sc[S_DMOD] = 33.0; sc[S_TS] = -0.3; sc[S_EBV] = 0.0; if constexpr (LEANX) fill_ext(lane, 0.0, 1.0); }
                else if (vwave == 3) bad[lane] = 0;
                (void)row; (void)dgl_l; (void)zgl_l;
#else
                if (vwave == 0) {
                    const double d_l = apply_slot(P.lumdist, row);
                    double z = 0.0;
                    if (P.redshift_mode == NMMA_Z_SLOT) z = apply_slot(P.redshift, row);
                    else if (P.redshift_mode == NMMA_Z_GRID) {
                        const double d_eff = P.has_h0 ? d_l * apply_slot(P.hubble, row) * P.inv_h0_ref : d_l;
                        z = interp_np(d_eff, dgl_l, zgl_l, P.n_cosmo, zgl_l[0], zgl_l[P.n_cosmo - 1]);
                        if (P.has_h0) z *= d_eff;
                    }
                    sc[S_ZP1] = 1 + z;
                    sc[S_IZP1] = 1.0 / (1 + z);
                    sc[S_RC] = redshift_correction(z);
                    chk = d_l + z;
                    // (the Pei-1992 series needs the redshift: this wave fills the extinction rows then, with its own copy of E(B-V))
                    if constexpr (LEANX) if (P.p92_cheb != nullptr) fill_ext(lane, P.has_ebv ? apply_slot(P.ebv, row) : 0.0, 1 + z);
                } else if (vwave == 1) {
                    const double d_l = apply_slot(P.lumdist, row);
                    sc[S_DMOD] = distance_modulus(d_l);
                    sc[S_TS] = apply_slot(P.timeshift, row);
                    sc[S_EBV] = P.has_ebv ? apply_slot(P.ebv, row) : 0.0;
                    chk = sc[S_TS] + sc[S_EBV];
                    if constexpr (LEANX) if (P.p92_cheb == nullptr) fill_ext(lane, sc[S_EBV], 1.0);
                } else if (vwave == 2) {
                    for (int p = 0; p < P.NP; ++p) chk += apply_slot(P.model_param[p], row);
                } else {
                    // (sampled time nodes are not "inputs that must be finite": autocomplete_data masks them, em/utils.py:634-645)
                    for (int o = 0; o < P.O; ++o) {
                        const int q0 = P.sys_off[o], q1 = P.sys_off[o + 1];
                        const bool nodes = P.sys_kind[o] == NMMA_SYS_NODES;
                        bool odd = false;
                        for (int q = q0; q < q1; ++q) {
                            const double v = apply_slot(P.sys_slots[q], row);
                            if (nodes) odd = odd || !(v - v == 0.0); else chk += v;
                            if constexpr (FAST) reinterpret_cast<double*>(smem + L.epar)[q * TS + lane] = v;   // sysv[slot][sample]
                        }
                        if constexpr (FAST)
                            if (nodes && odd) repair_nodes(reinterpret_cast<double*>(smem + L.epar) + q0 * TS + lane, TS, P.sys_node_t + q0, q1 - q0);
                    }
                    bad[lane] = 0;
                    if constexpr (COMB) bad[5 * TS + lane] = 0;       // gap[TS]: the sample met an interior gap of the second transient's curve
                }
#endif
                badp[vwave * TS + lane] = (chk - chk == 0.0) ? 0 : 1;
            }
        } else if (vt < TS) {
            long b = tile0 + vt;
            if (b >= B) b = B - 1;
            const double* row = theta + b * ld;
            double chk;
            sample_scalars(P, row, praw + vt * 8, scal + vt * 8, chk);
            for (int o = 0; o < P.O; ++o) {
                const int q0 = P.sys_off[o], q1 = P.sys_off[o + 1];
                const bool nodes = P.sys_kind[o] == NMMA_SYS_NODES;
                bool odd = false;
                for (int q = q0; q < q1; ++q) {
                    const double v = apply_slot(P.sys_slots[q], row);
                    if (nodes) odd = odd || !(v - v == 0.0); else chk += v;
                    if constexpr (FAST) reinterpret_cast<double*>(smem + L.epar)[q * TS + vt] = v;
                }
                if constexpr (FAST)
                    if (nodes && odd) repair_nodes(reinterpret_cast<double*>(smem + L.epar) + q0 * TS + vt, TS, P.sys_node_t + q0, q1 - q0);
            }
            if constexpr (LEANX) fill_ext(vt, scal[vt * 8 + S_EBV], scal[vt * 8 + S_ZP1]);
            badp[vt] = (chk - chk == 0.0) ? 0 : 1;
            badp[TS + vt] = 0; badp[2 * TS + vt] = 0; badp[3 * TS + vt] = 0;
            bad[vt] = 0;
            if constexpr (COMB) bad[5 * TS + vt] = 0;
        }
        // The table copies below do not depend on the per-sample chains above: with the staged prologue the chains occupy
        // likelihood waves 0-3 only, so waves 4-7 take ALL the copies and the two run side by side (a band's workgroup of the
        // split launch has nothing to hide its prologue behind, so the prologue costs the longer of the two instead of their sum)
        const bool two_lane_prologue = staged && NVW == 8;
        const int cvt = two_lane_prologue ? vt - 256 : vt;
        const int cnv = two_lane_prologue ? NV - 256 : NV;
        if (cvt >= 0) {
        // (slots: one per work item; some task flavours file a band's sums under its observed-filter index instead, which in a
        //  band's own workgroup of the split launch can lie beyond its item count -- the layout always holds P.O slots)
        for (int j = cvt; j < (W > P.O ? W : P.O) * TS; j += cnv) { chi_tot[j] = 0.0; gp_tot[j] = 0.0; }
#ifdef NMMA_DBG_NODIV     // measurement build: the table of reciprocal spacings without its division (a create-time table would hold them)
        for (int j = cvt; j < NS; j += cnv) { stl[j] = P.st[j]; stl[NS + j] = (j + 1 < NS) ? P.st_inv_dt : 0.0; }
#else
        for (int j = cvt; j < NS; j += cnv) { stl[j] = P.st[j]; stl[NS + j] = (j + 1 < NS) ? 1.0 / (P.st[j + 1] - P.st[j]) : 0.0; }
#endif
        if (!P.st_uniform && P.bguess != nullptr) {
            int* bgl = reinterpret_cast<int*>(stl + 2 * NS);
            gci32p bgs = as_global(P.bguess);
            for (int j = cvt; j <= BG_CELLS; j += cnv) bgl[j] = bgs[j];
        }
        if constexpr (FAST) {
            int* tmap = reinterpret_cast<int*>(smem + L.tmap);
            gci32p src = as_global(P.task_map[R - 1]);
            for (int j = cvt; j < P.n_tasks[R - 1] && j < TMAP_MAX; j += cnv) tmap[j] = src[j];
            if (L.dat >= 0) {
                double* dat = reinterpret_cast<double*>(smem + L.dat);
                const int nd = P.n_data;
                if constexpr (!EXT) {      // lean task: one {t, m, 1/sigma, ln sigma} record per datum
                    if (LEANX && P.dat_in_tab) {      // item-staged photometry: the epochs only, the records come with the item's rows
                        if constexpr (!DENSE) {       // (the dense task reads the epochs from the records too: nothing staged here)
                            gcf64p sdt = as_global(P.dt);
                            for (int j = cvt; j < nd; j += cnv) dat[j] = sdt[j];
                        }
                    } else {
                        gcf64p src4 = as_global(P.dat4);
                        for (int j = cvt; j < 4 * nd; j += cnv) dat[j] = src4[j];
                    }
                } else {
                    gcf64p sdt = as_global(P.dt), sdm = as_global(P.dm), sis = as_global(P.dinvsig), sls = as_global(P.dlogsig);
                    for (int j = cvt; j < nd; j += cnv) { dat[j] = sdt[j]; dat[nd + j] = sdm[j]; dat[2 * nd + j] = sis[j]; dat[3 * nd + j] = sls[j]; }
                }
            }
        }
        {
            gci32p src = as_global(reinterpret_cast<const int*>(P.item_desc));
            int* dst = reinterpret_cast<int*>(smem + L.itab);
            for (int j = cvt; j < W * ITEM_WORDS; j += cnv) dst[j] = src[j];
        }
        if constexpr (COMB) {       // the two-model flux-sum table (stack2_tab.h): every lane gathers its own row from LDS
            double* dst = reinterpret_cast<double*>(smem + L.nodes);
            for (int j = cvt; j < STACK2_NINT * STACK2_ROW; j += cnv) dst[j] = kStack2Tab[j];
            if (P.mass_tab2) {      // ... and log Phi for the upper limits behind it (logphi_tab.h)
                double* dst2 = reinterpret_cast<double*>(smem + L.nodes + STACK2_LDS_BYTES);
                for (int j = cvt; j < LOGPHI_NINT * LOGPHI_ROW; j += cnv) dst2[j] = kLogPhiTab[j];
            }
        }
        if constexpr (FASTM == 1 || FASTM == 3 || FASTM == 4 || FASTM == 5 || FASTM == 6 || FASTM == 9) {       // log Phi for upper limits / detections under a finite limit (logphi_tab.h)
            if (P.mass_tab) {
                // (the dense task keeps it behind its node buffers)
                double* dst = reinterpret_cast<double*>(smem + L.nodes) + (DENSE ? DENSE_NBUF * ((NS + 15) & ~15) * DENSE_STRIDE : 0);
                for (int j = cvt; j < LOGPHI_NINT * LOGPHI_ROW; j += cnv) dst[j] = kLogPhiTab[j];
            }
        }
        }

        // Static tables of a model filter (basis rows, span, mins, stage-1 lerp tables) are
        // copied global -> LDS by LDS-DMA one item ahead, into the other half of a double buffer:
        // no registers, and the L2 latency hides behind the phase of the previous item.
    }
    const int tab_bytes = P.tab_bytes;                 // multiple of 1 KiB (one wave-instruction)
    typedef __attribute__((address_space(3))) unsigned char* lds_p;
    typedef const __attribute__((address_space(1))) unsigned char* gbyte_p;
    auto tab_dma = [&](int k) {
        const int m = items[4 * k + 2];
        gbyte_p src = (gbyte_p)(uintptr_t)(P.tab + (size_t)m * tab_bytes);
        lds_p dst = (lds_p)(tabl + (k & 1) * tab_bytes);
        for (int c = vwave; c * 1024 < tab_bytes; c += NVW)
            __builtin_amdgcn_global_load_lds(src + c * 1024 + lane * 16, dst + c * 1024, 16, 0, 0);
    };
    constexpr bool all_fast = FAST;
    if (!helper) {
        if constexpr (!FAST) tab_dma(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (dbg && blockIdx.x == 0 && vt == 0) dbg[65] = clock64();
        sync_signal(sync + W + 1, lane);     // phase "prologue" of this wave done
    }

    auto sample_bad = [&](const int s_) -> bool {      // a non-finite input of sample s_ (any prologue part)
        const int* badp = bad + TS;
        return (badp[s_] | badp[TS + s_] | badp[2 * TS + s_] | badp[3 * TS + s_]) != 0;
    };
    const bool uniform = P.st_uniform != 0;
    const double st0 = P.st0, inv_dt = P.st_inv_dt;
    gcf64p g_dt = as_global(P.dt), g_dm = as_global(P.dm), g_dsig = as_global(P.dsig);
    gcf64p g_sigtot = as_global(P.dsigtot), g_logsig = as_global(P.dlogsig), g_invsig = as_global(P.dinvsig);

    auto item_phase = [&](auto nct_tag, const int k) {
        constexpr int NCT = decltype(nct_tag)::value;
        const ItemDesc& it = itab[k];
        const int o = it.o, ks = it.ks, nsrc = it.nsrc;
        const float* pbuf = part + (k % NBUF) * (NSLICE * TS * PSTR);
        const int jlo = it.jlo, jhi = it.jhi;
        const bool identity = it.identity != 0;
        const double ebvc = it.ebvc;
        const int G = it.G;                       // lanes per sample, power of two in [16, 64]
        const int gpb = NV / G;
        const int g = vt / G, gi = vt - g * G;
        const int d0 = it.d0;
        const int nf = it.nf;
        const int kind = it.kind;
        const double lim = it.lim;
        const double e_const = it.e_const;
        const int npass = (TS + gpb - 1) / gpb;
        const unsigned char* tb = tabl + (k & 1) * tab_bytes;
        const double* rows_m = reinterpret_cast<const double*>(tb);   // [NT][RS]: VA row | span | mins
        const int RS = P.tab_row_stride;
        const double* s1dx = reinterpret_cast<const double*>(tb + P.tab_off_s1dx);
        const double* s1of = reinterpret_cast<const double*>(tb + P.tab_off_s1of);
        const int* s1i = reinterpret_cast<const int*>(tb + P.tab_off_s1i);
        const float* b2l = reinterpret_cast<const float*>(tb + P.tab_off_b2);
        const bool dbg_on = dbg && blockIdx.x == 0 && vt == 0 && k == W - 1;
        if (dbg_on) dbg[96] = clock64();
        // the lane's first datum is the same for every sample: keep it in registers
        double c_t = 0, c_m = 0, c_sd = 0, c_sig = 0, c_lsig = 0;
        if (gi < nf) {
            const int di = d0 + gi;
            c_t = g_dt[di]; c_m = g_dm[di]; c_sd = g_dsig[di];
            if (kind == NMMA_SYS_CONST) { c_sig = g_sigtot[di]; c_lsig = g_logsig[di]; }
        }
        // (ordinary loads retired before the DMA is issued: a later wait on them would
        //  otherwise drain the DMA queue as well)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(c_t), "+v"(c_m), "+v"(c_sd), "+v"(c_sig), "+v"(c_lsig)::"memory");
        if (k + 1 < W) tab_dma(k + 1);            // next item's tables: in flight during this phase
        if (dbg_on) dbg[97] = clock64();

        for (int pass = 0; pass < npass; ++pass) {
            const int sl = pass * gpb + g;
            const bool active = sl < TS;
            const int s = active ? sl : 0;
            // waves whose groups are all beyond the tile have nothing to do in this pass (uniform)
            if (pass * gpb + (vwave * 64) / G >= TS) continue;
            // slice reduction (fixed order) + bias of the second Dense: lane gi < 16 owns
            // coefficient gi and hands it to its group through LDS (same wave: LDS is in order)
            if (gi < 16) {
                const int rb = s >> 4, sidx = s & 15;
                float cmine = 0.f;
#pragma unroll
                for (int w = 0; w < NSLICE; ++w) cmine += pbuf[((w * R + rb) * 16 + sidx) * PSTR + gi];
                cmine += b2l[gi];
                cdl[(vwave * (64 / 16) + (lane >> 4)) * 16 + gi] = (double)cmine;   // row: 16-lane slot of this wave
                // a model filter nobody observed (nf == 0): its light curve still has to be a usable one -- a non-finite
                // coefficient makes every node non-finite, the reference's sanity_check then floors the sample
                // (em_likelihood.py:305-311; with data the NaN reaches the sum through the terms themselves)
                if (nf == 0 && active && gi < NC && !(cmine - cmine == 0.f)) bad[s] = 1;
            }
            const double* crow = cdl + (vwave * 4 + ((lane & ~(G - 1)) >> 4)) * 16;     // the group's first slot
            // NCT > 0: exactly NCT coefficients, kept in registers; NCT == 0: any NC, re-read from LDS
            constexpr int NREG = NCT > 0 ? NCT : 1;
            double cc[NREG];
            if constexpr (NCT > 0) {
#pragma unroll
                for (int j = 0; j < NCT; ++j) cc[j] = crow[j];
            }

            if (dbg_on && pass == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dbg[98] = clock64(); }
            const double zp1 = scal[s * 8 + S_ZP1], tsh = scal[s * 8 + S_TS];
            const double dmod = scal[s * 8 + S_DMOD], rc = scal[s * 8 + S_RC];
            const double ebv = scal[s * 8 + S_EBV], izp1 = scal[s * 8 + S_IZP1];
            const double ext = extinction_mag(P.ext_law, ebvc, zp1, ebv);
            const double t_lo = stl[jlo] * zp1 + tsh, t_hi = stl[jhi] * zp1 + tsh;
            long brow = tile0 + s;
            if (brow >= B) brow = B - 1;
            const double* row = theta + brow * ld;

            // absolute magnitude at SVD-grid node i: (VA[i,:] . c) * span[i] + mins[i]
            auto mag_abs = [&](int i) -> double {
                const double* vr = rows_m + i * RS;
                double a;
                if constexpr (NCT > 0) {
                    a = vr[0] * cc[0];
#pragma unroll
                    for (int j = 1; j < NCT; ++j) a = fma(vr[j], cc[j], a);
                } else {
                    a = vr[0] * crow[0];
                    for (int j = 1; j < NC; ++j) a = fma(vr[j], crow[j], a);
                }
                return a * vr[NC] + vr[NC + 1];
            };
            // apparent magnitude at sample node j (stage-1 lerp + model.py:374-404)
            auto app_mag = [&](int j) -> double {
                const int i1 = s1i[j];
                double v;
                if (identity) {
                    v = mag_abs(i1);
                } else {
                    const double y0 = mag_abs(i1);
                    const double y1 = mag_abs(i1 + 1 < NT ? i1 + 1 : NT - 1);
                    const double slope = (y1 - y0) / s1dx[j];
                    v = slope * s1of[j] + y0;
                }
                if (ext != 0.0) v = v + ext;
                return (v + dmod) + rc;
            };

            double chi = 0.0, gp = 0.0;
            if (active) {
                for (int dd = gi; dd < nf; dd += G) {
                    const int di = d0 + dd;
                    double t, mobs, sd, sig, lsig;
                    if (dd == gi) { t = c_t; mobs = c_m; sd = c_sd; sig = c_sig; lsig = c_lsig; }
                    else {
                        t = g_dt[di]; mobs = g_dm[di]; sd = g_dsig[di]; sig = 0; lsig = 0;
                        if (kind == NMMA_SYS_CONST) { sig = g_sigtot[di]; lsig = g_logsig[di]; }
                    }
                    // stage-2: np.interp(t, t_obs[jlo..jhi], app, left=right=+inf); straight-line:
                    // bracket (clamped), both nodes reconstructed, result selected at the end
                    const bool inside = (jhi > jlo) && t >= t_lo && t <= t_hi;
                    int lo;   // t_obs[lo] <= t < t_obs[lo+1]  (lo = jhi-1 when t == t_obs[jhi])
                    if (uniform) {
                        lo = (int)floor(((t - tsh) * izp1 - st0) * inv_dt);
                        lo = lo < jlo ? jlo : (lo > jhi - 1 ? jhi - 1 : lo);
                    } else {
                        lo = jlo;
                        int hi = jhi;
                        while (inside && hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (stl[mid] * zp1 + tsh <= t) lo = mid; else hi = mid;
                        }
                    }
                    if (lo > jhi - 1) lo = jhi - 1;
                    if (lo < 0) lo = 0;
                    int hi1 = lo + 1 < NS ? lo + 1 : lo;
                    double x0 = stl[lo] * zp1 + tsh, x1 = stl[hi1] * zp1 + tsh;
                    // exact re-check of the guessed bracket (normally no iteration)
                    while (inside && ((x0 > t && lo > jlo) || (x1 <= t && lo < jhi - 1))) {
                        lo += (x0 > t) ? -1 : 1;
                        hi1 = lo + 1;
                        x0 = stl[lo] * zp1 + tsh; x1 = stl[hi1] * zp1 + tsh;
                    }
                    const double y0 = app_mag(lo), y1 = app_mag(hi1);
                    double est = lerp_np(t, x0, x1, y0, y1);
                    if (x0 == t) est = y0;
                    if (x1 == t) est = y1;          // also the exact right edge (np.interp: fp[-1])
                    if (!inside) est = (t != t) ? t : dinf();
                    if (dbg_on && pass == 0) { asm volatile("" : "+v"(est)); dbg[99] = clock64(); }
                    if (nsrc > 1) {  // averaged band: (a + b [+ c]) / n  (utils.py:566-584)
                        double acc_e = est;
                        if (ks > 0) acc_e = estb[s * L.nf_max + dd] + est;
                        if (ks < nsrc - 1) { estb[s * L.nf_max + dd] = acc_e; continue; }
                        est = acc_e / (double)nsrc;
                    }
                    // systematics (systematics.py:279-296) and combined sigma (em_likelihood.py:341)
                    double e = e_const;
                    if (kind != NMMA_SYS_CONST) {
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
                                const double sl2 = (v1 - v0) / P.sys_ndx[di];
                                e = sl2 * P.sys_noff[di] + v0;
                            }
                            // (a non-finite node value: autocomplete_data's finite mask, em/utils.py:634-645)
                            if (!(e - e == 0.0)) e = masked_nodes_at(sv, P.sys_node_t + P.sys_off[o], K, P.dt[di], row);
                        }
                        sig = sqrt(sd * sd + e * e);
                        lsig = log(sig);
                    }
                    if (sig - sig == 0.0) {   // np.isfinite(data_sigma): detection
                        chi += detection_term_gtab(mobs, est, sig, lsig, lim);       // (log Phi from the table, read from global memory in the fallback flavours)
                    } else {                  // infinite error: upper limit
                        gp += upper_limit_term_gtab(mobs, est, e);
                    }
                }
            }
            if (dbg_on && pass == 0) { asm volatile("" : "+v"(chi), "+v"(gp)); dbg[100] = clock64(); }
            if (ks < nsrc - 1) continue;      // uniform: more sources of this band to come
            // group reduction by DPP (no LDS round trips); the sum lands in the group's LAST row
            chi = group_sum(chi, G);
            gp = group_sum(gp, G);
            if (active && gi == G - 16) {
                // running sums over observed filters, in filter order (em_likelihood.py:337-352)
                chi_tot[k * TS + s] = chi;
                gp_tot[k * TS + s] = gp;
                if (chi != chi) bad[s] = 1;
                if (chi_parts != nullptr && tile0 + s < B) {
                    chi_parts[(long)o * B + tile0 + s] = sample_bad(s) ? dnan() : chi;
                    gp_parts[(long)o * B + tile0 + s] = gp;
                }
            }
            if (dbg_on && pass == 0) dbg[101] = clock64();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA of the next item's tables has landed
        if (dbg_on) dbg[102] = clock64();
    };

    // ---------------------------------------------------------------------------------
    // Fast path (every item qualifies; flag computed at create): NC == 10, one source per band,
    // constant systematics, no detection limit, sample_times equal to an equally spaced SVD grid,
    // no extinction.  The likelihood waves are independent workers: a task is one wave-load of
    // (item k, 64/G consecutive samples), G = 16/32/64 lanes per sample with two data per lane;
    // tasks are dealt round-robin to the waves.  A task has two stages:
    //   P (needs only the prologue): bracket every datum on the sample's observer-frame grid and
    //     fetch its two basis rows [VA | span | mins] from L2 into registers;
    //   Q (needs the coefficients of item k): reduce the 8 slice partial sums, 2 x 10 FMAs per datum,
    //     lerp, likelihood term, DPP group sum.
    // Only Q is on the critical path behind the MFMA role, and it contains no global-memory latency.
    // ---------------------------------------------------------------------------------
    const int nd_l = P.n_data;
    const double* dat_l = (FAST && L.dat >= 0) ? reinterpret_cast<const double*>(smem + L.dat) : nullptr;
    auto fast_task = [&](auto kind_tag, const int k, const int c) {
        // systematics of the item: 0 = constant (1/sigma_tot precomputed), 1 = one sampled parameter, 2 = sampled time nodes
        constexpr int SK = decltype(kind_tag)::value;
        constexpr bool par = SK != 0;
        const bool dbt = dbg && blockIdx.x == 0 && lane == 0 && c == 0 && k == W - 1;
        if (dbt) dbg[96] = clock64();
        const ItemDesc& it = itab[k];
        const int o = it.o;
        if (c == 0) {
            // The wave that owns the item's first task stages its basis rows [VA | span | mins | b2] into ring slot
            // k % NBUF by LDS-DMA (no registers).  Tasks are claimed well before their item is published, so the
            // copy lands while this wave does stage P and waits for the MLP.  (Sharing the copy among the item's
            // tasks would make them wait for each other: with more tasks per item than free waves that deadlocks.)
            if (k >= NBUF) sync_wait(sync + W + 1 + (k - NBUF + 1), itab[k - NBUF].ntask[R - 1], P.watchdog, 800 + k);
            typedef __attribute__((address_space(3))) unsigned char* lds_bp;
            typedef const __attribute__((address_space(1))) unsigned char* gbyte_p;
            gbyte_p src = (gbyte_p)(uintptr_t)(P.tab + (size_t)it.m * P.tab_bytes);
            lds_bp dst = (lds_bp)(tabl + (k % NBUF) * P.tab_fast_bytes);
            for (int q = 0; q * 1024 < P.tab_fast_bytes; ++q)
                __builtin_amdgcn_global_load_lds(src + q * 1024 + lane * 16, dst + q * 1024, 16, 0, 0);
        }
        const float* pbuf = part + (k % NBUF) * (NSLICE * TS * PSTR);
        const int jlo = it.jlo, jhi = it.jhi;
        const int G = it.G, d0 = it.d0, nf = it.nf;         // G = 16, 32 or 64 (a power of two: shifts, no division)
        const int lgG = (G == 16) ? 4 : (G == 32 ? 5 : 6);
        const int g = lane >> lgG, gi = lane & (G - 1);
        const int s = (c << (6 - lgG)) + g;                 // < TS: TS * G is a multiple of 64
        const double st0 = P.st0, inv_dt = P.st_inv_dt;
        const double* sc = scal + s * 8;
        const double zp1 = sc[S_ZP1], tsh = sc[S_TS], izp1 = sc[S_IZP1];
        const double dmrc = sc[S_DMOD] + sc[S_RC];
        const double izdt = izp1 * inv_dt;
        const double t_lo = stl[jlo] * zp1 + tsh, t_hi = stl[jhi] * zp1 + tsh;
        const double* sysv = reinterpret_cast<const double*>(smem + L.epar) + s;        // sysv[slot * TS]: this sample's values
        const int sv0 = par ? P.sys_off[o] : 0;                                            // first slot of the filter's group
        double e_sys = (SK == 1) ? sysv[sv0 * TS] : it.e_const;
        const bool lim_finite = EXT && (it.lim - it.lim == 0.0);
        const bool two = EXT && !(it.identity != 0 && it.same_grid != 0);
        const double ext = (EXT && sc[S_EBV] != 0.0) ? it.ebvc * sc[S_EBV] : 0.0;

        // ---- stage P: this lane's data and their brackets on the sample's observer-frame grid.
        // A lane owns data gi, gi + G, gi + 2G, ...; they are processed in pairs (two slots in registers):
        // the first pair before the item is published, further pairs (more than 2 G points per filter) after it.
        constexpr int NDL = 2;
        double c_t[NDL], c_m[NDL], c_is[NDL], c_ls[NDL], x0[NDL], x1[NDL];
        bool inside[NDL], hit1[NDL];
        int lo_[NDL];
        auto stage_p = [&](const int u0) {
#pragma unroll
            for (int u = 0; u < NDL; ++u) {
                const int dd = gi + (u0 + u) * G;
                const int di = d0 + (dd < nf ? dd : 0);
                if (dat_l != nullptr) { c_t[u] = dat_l[di]; c_m[u] = dat_l[nd_l + di]; c_is[u] = dat_l[2 * nd_l + di]; c_ls[u] = dat_l[3 * nd_l + di]; }
                else { c_t[u] = g_dt[di]; c_m[u] = g_dm[di]; c_is[u] = g_invsig[di]; c_ls[u] = g_logsig[di]; }
            }
#pragma unroll
            for (int u = 0; u < NDL; ++u) {
                if ((u0 + u) * G >= nf) { inside[u] = false; hit1[u] = false; x0[u] = 0; x1[u] = 0; lo_[u] = 0; continue; }   // uniform: slot unused by this item
                const double t = c_t[u];
                inside[u] = (jhi > jlo) && t >= t_lo && t <= t_hi;
                int lo;
                if (!EXT || uniform) {
                    lo = (int)floor(((t - tsh) * izp1 - st0) * inv_dt);
                    lo = lo < jlo ? jlo : (lo > jhi - 1 ? jhi - 1 : lo);
                } else {                     // sample_times not equally spaced (e.g. the CLI's log-spaced grid): bisection
                    lo = jlo;
                    int hi = jhi;
                    while (inside[u] && hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (stl[mid] * zp1 + tsh <= t) lo = mid; else hi = mid;
                    }
                    if (lo > jhi - 1) lo = jhi - 1;
                }
                if (lo < 0) lo = 0;
                double a = stl[lo] * zp1 + tsh, b = stl[lo + 1] * zp1 + tsh;
                for (int it2 = 0; it2 < 4 && inside[u] && ((a > t && lo > jlo) || (b <= t && lo < jhi - 1)); ++it2) {   // exact re-check (the guess is off by at most one)
                    lo += (a > t) ? -1 : 1;
                    a = stl[lo] * zp1 + tsh; b = stl[lo + 1] * zp1 + tsh;
                }
                x0[u] = a; x1[u] = b; hit1[u] = (b == t); lo_[u] = lo;
            }
        };
        stage_p(0);

        if (dbt) { asm volatile("" : "+v"(x0[0]), "+v"(lo_[0])); dbg[97] = clock64(); }
        // ---- stage Q
        if (c == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sync_signal(sync + 2 * W + 4 + k, lane);     // rows of item k staged
        }
        sync_wait(sync + 2 * W + 4 + k, 1, P.watchdog, 350 + k);
        sync_wait(sync + k, NMW, P.watchdog, 300 + k);   // coefficients of item k published
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k] = clock64();
        if (dbt) dbg[98] = clock64();
        const unsigned char* tbl = tabl + (k % NBUF) * P.tab_fast_bytes;
        const double* rows_l = reinterpret_cast<const double*>(tbl);          // [NT][12]: VA row | span | mins
        const float* b2l = reinterpret_cast<const float*>(tbl + P.tab_off_b2);
        const double* s1dx_l = reinterpret_cast<const double*>(tbl + P.tab_off_s1dx);   // (staged only when some item needs them)
        const double* s1of_l = reinterpret_cast<const double*>(tbl + P.tab_off_s1of);
        const int* s1i_l = reinterpret_cast<const int*>(tbl + P.tab_off_s1i);
        if (gi < 16) {
            const int rb = s >> 4, sidx = s & 15;
            const float* pp = pbuf + (rb * 16 + sidx) * PSTR + gi;
            float cmine = 0.f;
#pragma unroll
            for (int w = 0; w < NSLICE; ++w) cmine += pp[w * (R * 16 * PSTR)];
            cmine += b2l[gi];
            cdl[(wave * 4 + (lane >> 4)) * 16 + gi] = (double)cmine;
        }
        const double* crow = cdl + (wave * 4 + ((lane & ~(G - 1)) >> 4)) * 16;
        double cc[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) cc[j] = crow[j];
        if (dbt) { asm volatile("" : "+v"(cc[0]), "+v"(cc[9])); dbg[99] = clock64(); }
        double chi = 0.0, gp = 0.0;
        auto stage_q = [&](const int u0) {
#pragma unroll
            for (int u = 0; u < NDL; ++u) {
                if ((u0 + u) * G >= nf) continue;
                double y0, y1;
                if (EXT && two) {
                    // sample_times differ from the SVD grid (uniform per item): each of the two sample nodes is itself a
                    // lerp between two SVD nodes (stage 1, lightcurve_generation.py:177) -- up to four basis rows
                    auto mag_row = [&](const int i) -> double {
                        const double* r = rows_l + i * 12;
                        double a = r[0] * cc[0];
#pragma unroll
                        for (int j = 1; j < 10; ++j) a = fma(r[j], cc[j], a);
                        return a * r[10] + r[11];
                    };
                    auto stage1 = [&](const int j) -> double {
                        const int i1 = s1i_l[j];
                        const double ya = mag_row(i1);
                        if (it.identity) return ya;
                        const double yb = mag_row(i1 + 1 < NT ? i1 + 1 : NT - 1);
                        return ((yb - ya) / s1dx_l[j]) * s1of_l[j] + ya;
                    };
                    y0 = stage1(lo_[u]);
                    y1 = stage1(lo_[u] + 1);
                } else {
                    const double* r0 = rows_l + lo_[u] * 12;
                    const double* r1 = r0 + 12;
                    double a0 = r0[0] * cc[0], a1 = r1[0] * cc[0];
#pragma unroll
                    for (int j = 1; j < 10; ++j) { a0 = fma(r0[j], cc[j], a0); a1 = fma(r1[j], cc[j], a1); }
                    y0 = a0 * r0[10] + r0[11]; y1 = a1 * r1[10] + r1[11];
                }
                if constexpr (EXT) { if (ext != 0.0) { y0 = y0 + ext; y1 = y1 + ext; } }     // uniform per sample group
                y0 = y0 + dmrc; y1 = y1 + dmrc;
                const double t = c_t[u];
                double est = ((y1 - y0) * izdt) * (t - x0[u]) + y0;
                if (EXT && !uniform) est = ((y1 - y0) / (x1[u] - x0[u])) * (t - x0[u]) + y0;
                if (hit1[u]) est = y1;
                if (!inside[u]) est = (t != t) ? t : dinf();
                if (gi + (u0 + u) * G < nf) {
                    double isig = c_is[u], lsig = c_ls[u];
                    bool sig_bad = false;
                    if constexpr (par) {             // sampled systematic, combined per datum
                        if constexpr (SK == 2) {     // time nodes: constant outside, linear in between (systematics.py:288-291)
                            const int di = d0 + gi + (u0 + u) * G;
                            const int K = P.sys_nn[o], ni = P.sys_nidx[di];
                            if (ni < 0) e_sys = sysv[sv0 * TS];
                            else if (ni >= K - 1) e_sys = sysv[(sv0 + K - 1) * TS];
                            else {
                                const double v0 = sysv[(sv0 + ni) * TS], v1 = sysv[(sv0 + ni + 1) * TS];
                                const double sl2 = (v1 - v0) / P.sys_ndx[di];
                                e_sys = sl2 * P.sys_noff[di] + v0;
                            }
                        }
                        const double sd = c_is[u];   // (the slot carries sigma_data for these filters)
                        const double sig = sqrt(sd * sd + e_sys * e_sys);
                        if (sig - sig == 0.0) { isig = 1.0 / sig; lsig = log(sig); sig_bad = !(sig > 0); }
                        else isig = 0.0;             // infinite data error: upper limit
                        if (sig != sig) sig_bad = true;
                    }
                    // (two explicit accumulations: written as "chi += v" in one branch and "gp += ..." in the other, hipcc merges
                    //  them into ONE add on a two-element private array indexed by the branch -- scratch memory and an
                    //  s_waitcnt vmcnt(0) per datum)
                    double add_chi = 0.0, add_gp = 0.0;
                    if (isig != 0.0 || sig_bad) {
                        double v;
                        if (EXT && lim_finite) {         // uniform: truncated Gaussian with a finite detection limit
                            v = sig_bad ? dnan() : detection_term_gtab(c_m[u], est, 1.0 / isig, lsig, it.lim);
                        } else {
                            const double x = (c_m[u] - est) * isig;
                            v = (-(x * x) / 2.0 - kNormPdfLogC) - lsig;
                            if (!(est < dinf()) || sig_bad) v = dnan();
                        }
                        add_chi = v;
                    } else {
                        add_gp = upper_limit_term_gtab(c_m[u], est, e_sys);
                    }
                    opaque(add_chi); opaque(add_gp);
                    chi += add_chi; gp += add_gp;
                }
            }
        };
        stage_q(0);
        if constexpr (EXT)
            for (int u0 = NDL; u0 * G < nf; u0 += NDL) { stage_p(u0); stage_q(u0); }    // uniform: only beyond 2 G points
        if (dbt) { asm volatile("" : "+v"(chi)); dbg[100] = clock64(); }
        chi = group_sum(chi, G);
        if (it.has_ul) gp = group_sum(gp, G);
        if (dbt) { asm volatile("" : "+v"(chi)); dbg[101] = clock64(); }
        if (gi == G - 16) {
            chi_tot[o * TS + s] = chi;       // slot of the observed filter (= item order of the reference sum)
            gp_tot[o * TS + s] = gp;
            if (chi != chi) bad[s] = 1;
            if (chi_parts != nullptr && tile0 + s < B) {
                chi_parts[(long)o * B + tile0 + s] = sample_bad(s) ? dnan() : chi;
                gp_parts[(long)o * B + tile0 + s] = gp;
            }
        }
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k + 1] = clock64();
        sync_signal(sync + W + 2 + k, lane);     // one signal per task
    };

    // ---------------------------------------------------------------------------------
    // Lean task (FASTM == 1: every item has constant systematics, no detection limit, no extinction, sample_times =
    // the equally spaced SVD grid, at most 32 points per filter, photometry staged in LDS as {t, m, 1/sigma, ln sigma}).
    // Straight-line code, two independent slots per lane whose instruction streams the compiler interleaves:
    //   TYPEB = false (17..32 points): a task = 4 samples x 16 lanes, slot u = datum gi + 16 u of the lane's sample;
    //   TYPEB = true  (<= 16 points) : a task = 8 samples, slot u = datum gi of sample 8 c + 4 u + g.
    // Every VALU instruction of a likelihood wave takes issue time from the f32 MFMA stream of its SIMD (f32 MFMA and
    // VALU share the SIMD's vector pipe: tools/ubench/valu_mix2.hip), and a chain of dependent instructions advances one
    // instruction per MFMA issued in between -- hence few instructions and two chains per wave.
    // The bracket of a datum on the sample's observer-frame grid is taken from the index guess without the exact
    // re-check of the extended task: the guess can differ from np.interp's bracket only when the epoch lies within
    // ~1e-13 of a grid node, where both brackets give the same value to rounding (linear interpolation is continuous).
    // ---------------------------------------------------------------------------------
    auto lean_task = [&](auto typeb_tag, auto two_tag, auto sys_tag, auto nonuni_tag, const int k, const int c) {
        // NONUNI: sample_times not equally spaced (the CLI's default log-spaced grid): branch-free bisection instead of the
        // index guess, and the node spacing from a table
        constexpr bool NONUNI = decltype(nonuni_tag)::value;
        constexpr bool TYPEB = decltype(typeb_tag)::value;
        // SYS: one sampled systematic per filter or shared (em_syserr): sigma_tot = sqrt(sigma_data^2 + e^2) per datum and sample,
        // with the extended task's expressions (the photometry record then carries sigma_data instead of 1 / sigma_tot)
        constexpr bool SYS = decltype(sys_tag)::value;
        // TWO: sample_times differ from the SVD grid -- each of a datum's two sample nodes is a stage-1 lerp between two
        // SVD rows (lightcurve_generation.py:177), evaluated in two passes of the same four FMA chains
        constexpr bool TWO = decltype(two_tag)::value;
        const ItemDesc& it = itab[k];
        const int o = it.o;
#ifdef NMMA_DBG_TASKSTAMPS      // diagnostic build: per-task stage stamps, dbg[128 + 8 (4 k + c) + j]
#define NM_TS(j) do { if (dbg && blockIdx.x == 0 && lane == 0 && k < 6 && c < 4) dbg[128 + 8 * (4 * k + c) + (j)] = clock64(); } while (0)
#else
#define NM_TS(j) do { } while (0)
#endif
        NM_TS(0);
#ifdef NMMA_DBG_TASKSTAMPS
        if (dbg && blockIdx.x == 0 && lane == 0 && k < 6 && c < 4) dbg[128 + 8 * (4 * k + c) + 7] = wave;
#endif
        if (c == 0) {      // this wave stages the item's basis rows (see fast_task)
            if (k >= NBUF) sync_wait(sync + W + 1 + (k - NBUF + 1), itab[k - NBUF].ntask[R - 1], P.watchdog, 800 + k);
            typedef __attribute__((address_space(3))) unsigned char* lds_bp;
            typedef const __attribute__((address_space(1))) unsigned char* gbyte_p;
            // (item-staged photometry: the item's own table, its records behind the filter's rows)
            gbyte_p src = (LEANX && P.dat_in_tab) ? (gbyte_p)(uintptr_t)(P.tabi + (size_t)it.tabi * P.tabi_bytes)
                                                  : (gbyte_p)(uintptr_t)(P.tab + (size_t)it.m * P.tab_bytes);
            lds_bp dst = (lds_bp)(tabl + (k % NBUF) * P.tab_fast_bytes);
            for (int q = 0; q * 1024 < P.tab_fast_bytes; ++q)
                __builtin_amdgcn_global_load_lds(src + q * 1024 + lane * 16, dst + q * 1024, 16, 0, 0);
        }
        NM_TS(1);
        typedef const __attribute__((address_space(3))) double* lds_cdp;
        typedef const __attribute__((address_space(3))) float* lds_cfp;
        typedef __attribute__((address_space(3))) double* lds_dp;
        const lds_cfp pbuf = (lds_cfp)(part + (k % NBUF) * (NSLICE * TS * PSTR));
        // (uniform descriptor words as scalars: comparisons on them are SALU work)
        const int jlo = __builtin_amdgcn_readfirstlane(it.jlo), jhi = __builtin_amdgcn_readfirstlane(it.jhi);
        const int d0 = __builtin_amdgcn_readfirstlane(it.d0), nf = __builtin_amdgcn_readfirstlane(it.nf);
        const int g = lane >> 4, gi = lane & 15;
        const double st0 = P.st0, inv_dt = P.st_inv_dt;
        const lds_cdp stl_l = (lds_cdp)stl;
        typedef __attribute__((ext_vector_type(2))) double f64x2;
        typedef const __attribute__((address_space(3))) f64x2* lds_c2p;
        // photometry: records {t, m | 1/sigma, ln sigma} of all points in LDS -- or, item-staged (EmDev::dat_in_tab), the epochs of
        // all points there (stage P) and the item's records in its ring slot (stage Q)
        const bool item_dat = LEANX && __builtin_amdgcn_readfirstlane(P.dat_in_tab) != 0;
        const lds_cdp tdat = (lds_cdp)(smem + L.dat);
        const int tstride = item_dat ? 1 : 4;                                   // doubles between the epochs of consecutive points
        const lds_c2p dat4 = item_dat ? (lds_c2p)(tabl + (k % NBUF) * P.tab_fast_bytes + P.tab_off_dat)
                                      : (lds_c2p)(smem + L.dat);
        const int dbase = item_dat ? 0 : d0;
        const double st_lo = stl_l[jlo], st_hi = stl_l[jhi];
        const int nbis = NONUNI ? __builtin_amdgcn_readfirstlane(P.bg_nbis) : 0;
        const double bg_inv_h = NONUNI ? P.bg_inv_h : 0.0;
        const bool range_ok = jhi > jlo;
        constexpr int NSL = 2;
        int s_[NSL];
        s_[0] = TYPEB ? 8 * c + g : 4 * c + g;
        s_[1] = TYPEB ? s_[0] + 4 : s_[0];
        // ---- stage P (needs only the prologue)
        // (only what depends on the bracket stays in registers across the wait for the MLP: the photometry record and
        //  the sample scalars are read again from LDS in stage Q -- LDS reads cost the MFMA stream nothing, registers
        //  are what limits the workgroup to 16 waves)
        double dtx_[NSL];
        bool inside_[NSL], valid_[NSL];
        int lo_[NSL];
        lds_c2p D_[NSL];
        // combined model: the second transient's magnitudes at the slot's two bracket nodes, requested in stage P (their L2 / HBM
        // latency passes while the task waits for the surrogate) -- the only global-memory reads of the task
        double g2_[COMB ? NSL : 1][2];
        gcf64p lc2g = nullptr;
        if constexpr (COMB) lc2g = as_global(aux.lc2);
        // The kernels that take further passes over a filter with more than 32 points (both slots of a lane then belong to ONE
        // sample) keep that sample's scalars and window in registers across the passes: 12 fewer VALU instructions per pass.
        constexpr bool HOIST = LEANX && !TYPEB;
        double h_zp1 = 0.0, h_tsh = 0.0, h_izp1 = 0.0, h_tlo = 0.0, h_thi = 0.0, h_dmrc = 0.0, h_izdt = 0.0, h_guess0 = 0.0;
        if constexpr (HOIST) {
            const lds_cdp sc = (lds_cdp)(scal + s_[0] * 8);
            h_zp1 = sc[S_ZP1]; h_tsh = sc[S_TS]; h_izp1 = sc[S_IZP1];
            h_tlo = st_lo * h_zp1 + h_tsh; h_thi = st_hi * h_zp1 + h_tsh;
            h_dmrc = sc[S_DMOD] + sc[S_RC];
            h_izdt = h_izp1 * inv_dt;
            h_guess0 = -((h_tsh * h_izp1 + st0) * inv_dt);
        }
        // (filters with 17 .. 32 points take one pass over the lane's two slots; more points further passes of 32: `pp`)
        auto stage_p = [&](const int pp) {
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const int dd = TYPEB ? gi : gi + 16 * u + 32 * pp;
            valid_[u] = dd < nf;
            const int dix = valid_[u] ? dd : 0;
            D_[u] = dat4 + 2 * (dbase + dix);
            const double t = LEANX ? tdat[(d0 + dix) * tstride] : D_[u][0][0];
            const lds_cdp sc = (lds_cdp)(scal + s_[u] * 8);
            const double zp1 = HOIST ? h_zp1 : sc[S_ZP1], tsh = HOIST ? h_tsh : sc[S_TS], izp1 = HOIST ? h_izp1 : sc[S_IZP1];
            const double t_lo = HOIST ? h_tlo : st_lo * zp1 + tsh, t_hi = HOIST ? h_thi : st_hi * zp1 + tsh;
            inside_[u] = range_ok & (t >= t_lo) & (t <= t_hi);
            int lo;
            if constexpr (NONUNI) {
                // largest node index in [jlo, jhi - 1] whose observer-frame time is <= t (np.interp's bracket)
                // (narrowed first by the lookup over equal cells of the source-frame grid, one node of slack on either side for the
                //  rounding of the source-frame time: bg_nbis steps instead of ceil(log2 NS))
                typedef const __attribute__((address_space(3))) int* lds_cip_bg;
                const lds_cip_bg bgl = (lds_cip_bg)(stl_l + 2 * NS);
                int cq = (int)(((t - tsh) * izp1 - st0) * bg_inv_h);
                cq = cq < 0 ? 0 : (cq > BG_CELLS - 1 ? BG_CELLS - 1 : cq);
                lo = bgl[cq] - 1;
                int hi = bgl[cq + 1] + 2;
                lo = lo < jlo ? jlo : lo;
                hi = hi > jhi ? jhi : hi;
                for (int itb = 0; itb < nbis; ++itb) {         // uniform trip count
                    const int mid = (lo + hi) >> 1;
                    const bool le = (stl_l[mid] * zp1 + tsh) <= t;
                    lo = le ? mid : lo;
                    hi = le ? hi : mid;
                }
                lo = lo > jhi - 1 ? jhi - 1 : lo;
            } else {
                // (HOIST: the same guess from one FMA on per-sample constants -- it may differ from the unfused form only for an
                //  epoch within rounding of a node, where both brackets give the same value)
                lo = HOIST ? (int)floor(fma(t, h_izdt, h_guess0)) : (int)floor(((t - tsh) * izp1 - st0) * inv_dt);
                lo = lo > jhi - 1 ? jhi - 1 : lo;
                lo = lo < jlo ? jlo : lo;
            }
            dtx_[u] = t - (stl_l[lo] * zp1 + tsh);            // t - x0
            lo_[u] = lo;
            asm volatile("" : "+v"(dtx_[u]), "+v"(lo_[u]));   // (evaluated here, before the wait for the MLP)
            if constexpr (COMB) {
                long bb = tile0 + s_[u];
                bb = bb < B ? bb : B - 1;
                const unsigned off = (unsigned)((bb * P.M + it.m) * NS + lo);      // (in doubles from aux.lc2; < 2^32: checked by the launcher)
                g2_[u][0] = lc2g[off]; g2_[u][1] = lc2g[off + 1u];
            }
        }
        };
        stage_p(0);
        NM_TS(2);
        // ---- stage Q (needs the coefficients of item k)
        if (c == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            NM_TS(3);
            sync_signal(sync + 2 * W + 4 + k, lane);     // rows of item k staged
        }
        sync_wait(sync + 2 * W + 4 + k, 1, P.watchdog, 350 + k);
        NM_TS(4);
        sync_wait(sync + k, NMW, P.watchdog, 300 + k);   // coefficients of item k published
        NM_TS(5);
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k] = clock64();
        const unsigned char* tbl = tabl + (k % NBUF) * P.tab_fast_bytes;
        const lds_cfp b2l = (lds_cfp)(tbl + P.tab_off_b2);
        const float b2v = b2l[gi];
        constexpr int NCC = TYPEB ? 2 : 1;
        lds_c2p cc_[NCC];          // the sample's 10 coefficients (fp64) in this wave's LDS slots: [wave][q][g][16]
#pragma unroll
        for (int q = 0; q < NCC; ++q) {
            // slice reduction (fixed order) + bias of the second Dense: lane gi owns coefficient gi of its sample
            const int s = s_[q];
            const lds_cfp pp = pbuf + ((s >> 4) * 16 + (s & 15)) * PSTR + gi;
            float cmine = pp[0];
#pragma unroll
            for (int w = 1; w < NSLICE; ++w) cmine += pp[w * (R * 16 * PSTR)];
            cmine += b2v;
            const lds_dp cslot = (lds_dp)(cdl + ((wave * 2 + q) * 4 + g) * 16);
            cslot[gi] = (double)cmine;
            cc_[q] = (lds_c2p)cslot;
        }
        // the two basis rows of every slot, read as 16-byte pairs [VA[2j], VA[2j+1]] (pair 5 = [span, mins]); the four FMA
        // chains (2 slots x 2 rows) advance together, one pair per step, so that no instruction waits for its predecessor.
        // TWO: pass 0 reconstructs the SVD rows around sample node lo, pass 1 those around node lo + 1.
        const lds_c2p rows2 = (lds_c2p)(tbl);
        const lds_cdp s1of_l = (lds_cdp)(tbl + P.tab_off_s1of), s1inv_l = (lds_cdp)(tbl + P.tab_off_s1inv);
        typedef const __attribute__((address_space(3))) int* lds_cip;
        const lds_cip s1i_l = (lds_cip)(tbl + P.tab_off_s1i);
        double v_[NSL], gp_[NSL], esys_[NSL] = {0.0, 0.0};
        const int sv0 = SYS ? __builtin_amdgcn_readfirstlane(P.sys_off[o]) : 0;      // first slot of the filter's parameter
        const int svl = (SYS && COMB) ? __builtin_amdgcn_readfirstlane(P.sys_nn[o]) - 1 : 0;   // last time node (combined-model flavours)
        const lds_cdp ext_l = (lds_cdp)(smem + L.exttab) + (P.has_ebv ? k : 0) * TS;
        auto stage_q = [&]() {
        double ynode_[2][NSL];            // magnitudes at the two sample nodes of every slot
#pragma unroll
        for (int pass = 0; pass < (TWO ? 2 : 1); ++pass) {
            lds_c2p ra_[NSL], rb_[NSL];
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                if constexpr (TWO) {
                    const int j = lo_[u] + pass;
                    int i1 = s1i_l[j];
                    i1 = i1 < 0 ? 0 : i1;                         // (nodes outside the SVD grid lie outside [jlo, jhi]: never bracketed)
                    const int i2 = i1 + 1 < NT ? i1 + 1 : NT - 1;
                    ra_[u] = rows2 + i1 * 6; rb_[u] = rows2 + i2 * 6;
                } else {
                    ra_[u] = rows2 + lo_[u] * 6; rb_[u] = ra_[u] + 6;
                }
            }
            double a0_[NSL], a1_[NSL];
            f64x2 p0_[NSL], p1_[NSL];
#pragma unroll
            for (int u = 0; u < NSL; ++u) { p0_[u] = ra_[u][0]; p1_[u] = rb_[u][0]; }
            f64x2 cq_[NCC];
#pragma unroll
            for (int q = 0; q < NCC; ++q) cq_[q] = cc_[q][0];
#pragma unroll
            for (int jp = 0; jp < 5; ++jp) {
                f64x2 n0_[NSL], n1_[NSL], nq_[NCC];
#pragma unroll
                for (int u = 0; u < NSL; ++u) { n0_[u] = ra_[u][jp + 1]; n1_[u] = rb_[u][jp + 1]; }
                if (jp < 4) {
#pragma unroll
                    for (int q = 0; q < NCC; ++q) nq_[q] = cc_[q][jp + 1];
                }
#pragma unroll
                for (int u = 0; u < NSL; ++u) {
                    const f64x2 cq = cq_[TYPEB ? u : 0];
                    if (jp == 0) { a0_[u] = p0_[u][0] * cq[0]; a1_[u] = p1_[u][0] * cq[0]; }
                    else { a0_[u] = fma(p0_[u][0], cq[0], a0_[u]); a1_[u] = fma(p1_[u][0], cq[0], a1_[u]); }
                }
#pragma unroll
                for (int u = 0; u < NSL; ++u) {
                    const f64x2 cq = cq_[TYPEB ? u : 0];
                    a0_[u] = fma(p0_[u][1], cq[1], a0_[u]); a1_[u] = fma(p1_[u][1], cq[1], a1_[u]);
                }
#pragma unroll
                for (int u = 0; u < NSL; ++u) { p0_[u] = n0_[u]; p1_[u] = n1_[u]; }
                if (jp < 4) {
#pragma unroll
                    for (int q = 0; q < NCC; ++q) cq_[q] = nq_[q];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                const double ya = a0_[u] * p0_[u][0] + p0_[u][1], yb = a1_[u] * p1_[u][0] + p1_[u][1];   // (VA[i,:].c) span[i] + mins[i]
                if constexpr (TWO) {
                    const int j = lo_[u] + pass;
                    // stage 1: ((yb - ya) / dx) * off + ya with the reciprocal of dx from the table
                    ynode_[pass][u] = ((yb - ya) * s1inv_l[j]) * s1of_l[j] + ya;
                } else {
                    ynode_[0][u] = ya; ynode_[1][u] = yb;
                }
            }
        }
        if constexpr (COMB) {
            // stack_magnitudes (model.py:1486-1510) on the slot's two nodes: min(kn, m2) - g(|kn - m2|) from the table in LDS
            // (stack2_node: the arithmetic of the stacking kernels' stack2_fast, em_common.h).  STRAIGHT-LINE code on purpose: a
            // branch here -- even a one-line conditional store -- splits the task's one scheduling region and hipcc then spills
            // hundreds of registers.  A node where the second transient has no finite value: at the first / last node of the grid
            // it is the edge of that transient's time range -- no flux, the general form's value for (kn, +inf) -- anywhere else it
            // may be a gap that autocomplete_data fills from the transient's finite neighbours (utils.py:634-645): the sample is
            // flagged (an LDS OR issued by every lane) and re-evaluated after this launch by the kernels that materialise the curves.
            // Union grids (nmma_em_config::base_times): the operand went through nmma_lc_regrid (aux.completed: its non-finite nodes are
            // the ends of its time range wherever they lie -- no flux), and the KILONOVA has no value on the nodes outside its own grid
            // (rows [0 | 1 | +inf]): the sum is then the second transient alone, the general form's value for (+inf, m2); a node where
            // neither has a value sends the row to the re-evaluation launch.
            const double* tab2 = reinterpret_cast<const double*>(smem + L.nodes);
            const bool completed = aux.completed != 0;
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                bool gap = false;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const double kn = ynode_[e][u], m2 = g2_[u][e];
                    const int j = lo_[u] + e;
                    const bool m2_fin = (m2 - m2 == 0.0);
                    const bool kn_out = kn == dinf();
                    const bool edge = !m2_fin & (completed | (j == 0) | (j == NS - 1));
                    const double r = stack2_node(kn, m2, tab2);
                    const double alone = stack2_no_flux(edge ? kn : m2);
                    ynode_[e][u] = (edge | kn_out) ? alone : r;
                    gap |= !m2_fin & (!edge | kn_out);
                }
                const int flag = (gap & valid_[u]) ? 1 : 0;
                __hip_atomic_fetch_or((lds_ip)(bad + 5 * TS + s_[u]), flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        double est_[NSL], m_[NSL];
        bool ul_[NSL];
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const f64x2 tm = D_[u][0], sl = D_[u][1];       // {t, m}, {1/sigma, ln sigma}
            const lds_cdp sc = (lds_cdp)(scal + s_[u] * 8);
            const double dmrc = HOIST ? h_dmrc : sc[S_DMOD] + sc[S_RC];
            const double izdt = (HOIST && !NONUNI) ? h_izdt : (HOIST ? h_izp1 : sc[S_IZP1]) * (NONUNI ? stl_l[NS + lo_[u]] : inv_dt);
            double y0 = ynode_[0][u], y1 = ynode_[1][u];
            double est;
            if constexpr (HOIST) {
                // the sample's offsets (extinction of this item + distance modulus + K-correction) added once, to the left node:
                // the slope is the difference of the node magnitudes themselves
                const double yl = (y0 + ext_l[s_[0]]) + dmrc;
                est = ((y1 - y0) * izdt) * dtx_[u] + yl;
            } else {
                if constexpr (LEANX) {       // extinction magnitude of this sample and item (filled by the prologue, model.py:323-342)
                    const double ext = ext_l[s_[u]];       // (no branch here: 0 when there is no extinction)
                    y0 = y0 + ext; y1 = y1 + ext;
                }
                y0 = y0 + dmrc; y1 = y1 + dmrc;
                est = ((y1 - y0) * izdt) * dtx_[u] + y0;
            }
            double isig = sl[0], lsig = sl[1];
            bool sig_bad = false;
            if constexpr (SYS) {
                const double sd = sl[0];                      // sigma_data
                double e_sys;
                if constexpr (COMB) {
                    // (combined-model flavours, round 6: systematics at time nodes too -- the record's fourth word is the datum's node
                    //  index + fraction, written at create; a single parameter: i0 = i1 = 0, fr = 0 -- exactly e0.  lean_gen_task's lines)
                    const double qf = floor(sl[1]), fr = sl[1] - qf;
                    const int i0 = (int)qf, i1 = i0 < svl ? i0 + 1 : svl;
                    const lds_cdp ep = (lds_cdp)(smem + L.epar) + sv0 * TS + s_[u];
                    const double e0 = ep[i0 * TS], e1 = ep[i1 * TS];
                    e_sys = (e1 - e0) * fr + e0;
                } else {
                    e_sys = ((lds_cdp)(smem + L.epar))[sv0 * TS + s_[u]];
                }
                esys_[u] = e_sys;
                // (1 / sigma_tot and ln sigma_tot without the library's sqrt, division and log: rsqrt_pos / log_pos above.  The exact
                //  shape of these five lines matters to hipcc: taking the log of sigma^2 instead, or the finite test from sqrt(s2),
                //  made the lean kernels spill ~770 registers)
                const double s2 = sd * sd + e_sys * e_sys;    // sigma_tot^2
                const double rs = rsqrt_pos(s2);
                const double sig = s2 * rs;
                const bool fin = (s2 - s2 == 0.0);
                isig = fin ? rs : 0.0;                        // infinite data error: upper limit
                lsig = log_pos(sig);
                sig_bad = (fin & !(s2 > 0)) | (s2 != s2);
            }
            const double x = (tm[1] - est) * isig;
            double v = (-(x * x) / 2.0 - kNormPdfLogC) - lsig;
            opaque(v);                                        // (computed on every lane: no exec-masked region around the chain)
            // outside the model window est = +inf: truncnorm.logpdf(loc = inf) = NaN (em_likelihood.py:252-256)
            v = (inside_[u] & !sig_bad) ? v : dnan();
            ul_[u] = valid_[u] & (isig == 0.0) & !sig_bad;    // infinite data error: an upper limit
            v_[u] = (valid_[u] & !ul_[u]) ? v : 0.0;
            est_[u] = est; m_[u] = tm[1];
        }
        if constexpr (COMB) {
            // a finite detection limit on the band (em_likelihood.py:252-256; the general lean task's block, lean_gen_task): truncnorm's mass
            // log Phi((lim - est) / sigma) from the table behind the flux-sum table.  1 / sigma and ln sigma are read again from the
            // datum's record (constant systematics) instead of being kept across the flux sum: registers are what this flavour lacks
            const bool lim_fin = (it.lim - it.lim == 0.0);                // uniform
            if (lim_fin) {
#pragma unroll
                for (int u = 0; u < NSL; ++u)
                    if (valid_[u] & !ul_[u]) {
                        const f64x2 sl = D_[u][1];
                        double isig = sl[0], lsig = sl[1];
                        bool sbad = false;
                        if constexpr (SYS) {
                            const double s2 = sl[0] * sl[0] + esys_[u] * esys_[u];
                            const double rs = rsqrt_pos(s2);
                            isig = rs; lsig = log_pos(s2 * rs);
                            sbad = ((s2 - s2 == 0.0) & !(s2 > 0)) | (s2 != s2);
                        }
                        const double b = (it.lim - est_[u]) * isig;
                        const double x = (m_[u] - est_[u]) * isig;
                        double v = ((-(x * x) / 2.0 - kNormPdfLogC) - log_gauss_mass_tab(b, (lds_cdp)(smem + L.nodes + STACK2_LDS_BYTES))) - lsig;
                        v = x > b ? -dinf() : v;                                               // fainter than the limit, yet detected
                        v_[u] = (inside_[u] & !sbad & (b > -dinf()) & (x == x)) ? v : dnan();
                    }
            }
        }
        gp_[0] = 0.0; gp_[1] = 0.0;
        if (it.has_ul) {                                    // uniform; the term itself only on the lanes that hold a limit
#pragma unroll
            for (int u = 0; u < NSL; ++u)
                if (ul_[u]) {
                    if constexpr (FASTM == 1 || FASTM == 3 || FASTM == 4) {
                        if (P.mass_tab) gp_[u] = upper_limit_term_tab(m_[u], inside_[u] ? est_[u] : dinf(), SYS ? esys_[u] : it.e_const, (lds_cdp)(smem + L.nodes));
                        else gp_[u] = upper_limit_term(m_[u], inside_[u] ? est_[u] : dinf(), SYS ? esys_[u] : it.e_const);
                    } else if constexpr (COMB) {
                        if (P.mass_tab2) gp_[u] = upper_limit_term_tab(m_[u], inside_[u] ? est_[u] : dinf(), SYS ? esys_[u] : it.e_const, (lds_cdp)(smem + L.nodes + STACK2_LDS_BYTES));
                        else gp_[u] = upper_limit_term(m_[u], inside_[u] ? est_[u] : dinf(), SYS ? esys_[u] : it.e_const);
                    } else {
                        gp_[u] = upper_limit_term(m_[u], inside_[u] ? est_[u] : dinf(), SYS ? esys_[u] : it.e_const);
                    }
                }
        }
        };
        stage_q();
        lds_dp chi_l = (lds_dp)chi_tot;
        lds_dp gp_l = (lds_dp)gp_tot;
        if constexpr (TYPEB) {
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                const double chi = group_sum(v_[u], 16);
                double gp = 0.0;
                if (it.has_ul) gp = group_sum(gp_[u], 16);
                if (gi == 0) {
                    const int s = s_[u];
                    chi_l[o * TS + s] = chi;
                    gp_l[o * TS + s] = gp;
                    if (chi != chi) bad[s] = 1;
                    if (chi_parts != nullptr && tile0 + s < B) {
                        chi_parts[(long)o * B + tile0 + s] = sample_bad(s) ? dnan() : chi;
                        gp_parts[(long)o * B + tile0 + s] = gp;
                    }
                }
            }
        } else {
            double vacc = v_[0] + v_[1], gacc = gp_[0] + gp_[1];
            if constexpr (LEANX) {
                for (int pp = 1; pp * 32 < nf; ++pp) {      // uniform: only filters with more than 32 points
                    stage_p(pp);
                    stage_q();
                    vacc += v_[0] + v_[1]; gacc += gp_[0] + gp_[1];
                }
            }
            const double chi = group_sum(vacc, 16);
            double gp = 0.0;
            if (it.has_ul) gp = group_sum(gacc, 16);
            if (gi == 0) {
                const int s = s_[0];
                chi_l[o * TS + s] = chi;
                gp_l[o * TS + s] = gp;
                if (chi != chi) bad[s] = 1;
                if (chi_parts != nullptr && tile0 + s < B) {
                    chi_parts[(long)o * B + tile0 + s] = sample_bad(s) ? dnan() : chi;
                    gp_parts[(long)o * B + tile0 + s] = gp;
                }
            }
        }
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k + 1] = clock64();
        NM_TS(6);
        sync_signal(sync + W + 2 + k, lane);     // one signal per task
#undef NM_TS
    };

    // ---------------------------------------------------------------------------------
    // lean_gen_task (FASTM = 5): the general lean task -- lean_task's extras (passes of 32 points, extinction table, sampled
    // systematics, unequally spaced grids) plus
    //  * averaged bands: an observed filter whose magnitude is the mean of several model filters (utils.py:566-584).  k is the
    //    LAST source item of the band (items k - nsrc + 1 .. k, one surrogate each; nsrc = 1 for an ordinary band); stage Q
    //    walks the sources -- wait for the source's coefficients, node magnitudes with its basis rows, + extinction of its
    //    filter + distance modulus -- sums them in source order and divides the interpolated sum by nsrc (the generic item
    //    phase's order).  The other source items own no tasks; every task of the band signals each of them, which releases
    //    their ring slots.
    //  * time-node systematics (systematics.py:288-291): see SYS below.
    // A copy of lean_task rather than a variant of it: any change to that lambda, even a semantically neutral one, moves
    // hipcc's register allocation off its optimum in every instantiation.
    // ---------------------------------------------------------------------------------
    auto lean_gen_task = [&](auto typeb_tag, auto two_tag, auto sys_tag, auto nonuni_tag, const int k, const int c) {
        // NONUNI: sample_times not equally spaced (the CLI's default log-spaced grid): branch-free bisection instead of the
        // index guess, and the node spacing from a table
        constexpr bool NONUNI = decltype(nonuni_tag)::value;
        constexpr bool TYPEB = decltype(typeb_tag)::value;
        // SYS: sampled systematics -- one parameter per filter or shared (em_syserr), or parameters at time nodes, constant
        // outside and linear in between (systematics.py:288-291): sigma_tot = sqrt(sigma_data^2 + e^2) per datum and sample.
        // The photometry record then carries [t | m | sigma_data | q], q = node index + fraction of the datum's node interval
        // (0 for a single parameter; written by nmma_em_create), and e = v[i] + (v[i + 1] - v[i]) * fraction
        constexpr bool SYS = decltype(sys_tag)::value;
        // TWO: sample_times differ from the SVD grid -- each of a datum's two sample nodes is a stage-1 lerp between two
        // SVD rows (lightcurve_generation.py:177), evaluated in two passes of the same four FMA chains
        constexpr bool TWO = decltype(two_tag)::value;
        const ItemDesc& it = itab[k];
        const int o = it.o;
        const int nsrc = __builtin_amdgcn_readfirstlane(it.nsrc);
        const int k0 = k - (nsrc - 1);
        if (c == 0) {      // this wave stages the basis rows of every source (the host guarantees NBUF >= nsrc)
            typedef __attribute__((address_space(3))) unsigned char* lds_bp;
            typedef const __attribute__((address_space(1))) unsigned char* gbyte_p;
            for (int kk = k0; kk <= k; ++kk) {
                if (kk >= NBUF) sync_wait(sync + W + 1 + (kk - NBUF + 1), itab[kk - NBUF].ntask[R - 1], P.watchdog, 800 + kk);
                gbyte_p src = (gbyte_p)(uintptr_t)(P.tab + (size_t)itab[kk].m * P.tab_bytes);
                lds_bp dst = (lds_bp)(tabl + (kk % NBUF) * P.tab_fast_bytes);
                for (int q = 0; q * 1024 < P.tab_fast_bytes; ++q)
                    __builtin_amdgcn_global_load_lds(src + q * 1024 + lane * 16, dst + q * 1024, 16, 0, 0);
            }
        }
        typedef const __attribute__((address_space(3))) double* lds_cdp;
        typedef const __attribute__((address_space(3))) float* lds_cfp;
        typedef __attribute__((address_space(3))) double* lds_dp;
        // (uniform descriptor words as scalars: comparisons on them are SALU work)
        const int jlo = __builtin_amdgcn_readfirstlane(it.jlo), jhi = __builtin_amdgcn_readfirstlane(it.jhi);
        const int d0 = __builtin_amdgcn_readfirstlane(it.d0), nf = __builtin_amdgcn_readfirstlane(it.nf);
        const int g = lane >> 4, gi = lane & 15;
        const double st0 = P.st0, inv_dt = P.st_inv_dt;
        const lds_cdp stl_l = (lds_cdp)stl;
        typedef __attribute__((ext_vector_type(2))) double f64x2;
        typedef const __attribute__((address_space(3))) f64x2* lds_c2p;
        const lds_c2p dat4 = (lds_c2p)(smem + L.dat);
        const double st_lo = stl_l[jlo], st_hi = stl_l[jhi];
        const int nbis = NONUNI ? __builtin_amdgcn_readfirstlane(P.bg_nbis) : 0;
        const double bg_inv_h = NONUNI ? P.bg_inv_h : 0.0;
        const bool range_ok = jhi > jlo;
        constexpr int NSL = 2;
        int s_[NSL];
        s_[0] = TYPEB ? 8 * c + g : 4 * c + g;
        s_[1] = TYPEB ? s_[0] + 4 : s_[0];
        // ---- stage P (needs only the prologue)
        // (only what depends on the bracket stays in registers across the wait for the MLP: the photometry record and
        //  the sample scalars are read again from LDS in stage Q -- LDS reads cost the MFMA stream nothing, registers
        //  are what limits the workgroup to 16 waves)
        double dtx_[NSL];
        bool inside_[NSL], valid_[NSL];
        int lo_[NSL];
        lds_c2p D_[NSL];
        // (filters with 17 .. 32 points take one pass over the lane's two slots; more points further passes of 32: `pp`)
        auto stage_p = [&](const int pp) {
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const int dd = TYPEB ? gi : gi + 16 * u + 32 * pp;
            valid_[u] = dd < nf;
            D_[u] = dat4 + 2 * (d0 + (valid_[u] ? dd : 0));
            const double t = D_[u][0][0];
            const lds_cdp sc = (lds_cdp)(scal + s_[u] * 8);
            const double zp1 = sc[S_ZP1], tsh = sc[S_TS], izp1 = sc[S_IZP1];
            const double t_lo = st_lo * zp1 + tsh, t_hi = st_hi * zp1 + tsh;
            inside_[u] = range_ok & (t >= t_lo) & (t <= t_hi);
            int lo;
            if constexpr (NONUNI) {
                // largest node index in [jlo, jhi - 1] whose observer-frame time is <= t (np.interp's bracket)
                // (narrowed first by the lookup over equal cells of the source-frame grid, one node of slack on either side for the
                //  rounding of the source-frame time: bg_nbis steps instead of ceil(log2 NS))
                typedef const __attribute__((address_space(3))) int* lds_cip_bg;
                const lds_cip_bg bgl = (lds_cip_bg)(stl_l + 2 * NS);
                int cq = (int)(((t - tsh) * izp1 - st0) * bg_inv_h);
                cq = cq < 0 ? 0 : (cq > BG_CELLS - 1 ? BG_CELLS - 1 : cq);
                lo = bgl[cq] - 1;
                int hi = bgl[cq + 1] + 2;
                lo = lo < jlo ? jlo : lo;
                hi = hi > jhi ? jhi : hi;
                for (int itb = 0; itb < nbis; ++itb) {         // uniform trip count
                    const int mid = (lo + hi) >> 1;
                    const bool le = (stl_l[mid] * zp1 + tsh) <= t;
                    lo = le ? mid : lo;
                    hi = le ? hi : mid;
                }
                lo = lo > jhi - 1 ? jhi - 1 : lo;
            } else {
                lo = (int)floor(((t - tsh) * izp1 - st0) * inv_dt);
                lo = lo > jhi - 1 ? jhi - 1 : lo;
                lo = lo < jlo ? jlo : lo;
            }
            dtx_[u] = t - (stl_l[lo] * zp1 + tsh);            // t - x0
            lo_[u] = lo;
            asm volatile("" : "+v"(dtx_[u]), "+v"(lo_[u]));   // (evaluated here, before the wait for the MLP)
        }
        };
        stage_p(0);
        // ---- stage Q (needs the coefficients of item k)
        if (c == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int kk = k0; kk <= k; ++kk) sync_signal(sync + 2 * W + 4 + kk, lane);     // rows of the band's sources staged
        }
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k] = clock64();
        constexpr int NCC = TYPEB ? 2 : 1;
        // the two basis rows of every slot, read as 16-byte pairs [VA[2j], VA[2j+1]] (pair 5 = [span, mins]); the four FMA
        // chains (2 slots x 2 rows) advance together, one pair per step, so that no instruction waits for its predecessor.
        // TWO: pass 0 reconstructs the SVD rows around sample node lo, pass 1 those around node lo + 1.
        typedef const __attribute__((address_space(3))) int* lds_cip;
        double v_[NSL], gp_[NSL], esys_[NSL] = {0.0, 0.0};
        // a finite detection limit (uniform per band): the truncated Gaussian of em_likelihood.py:252-256 through detection_term,
        // evaluated after the straight-line term like the upper limits -- a call per datum, on the lanes that hold a detection
        const bool lim_fin = (it.lim - it.lim == 0.0);
        double isig_[NSL], lsig_[NSL];
        bool sbad_[NSL];
        const int sv0 = SYS ? __builtin_amdgcn_readfirstlane(P.sys_off[o]) : 0;      // first slot of the filter's parameter(s)
        const int svl = SYS ? __builtin_amdgcn_readfirstlane(P.sys_nn[o]) - 1 : 0;   // last node
        auto stage_q = [&]() {
        double ys_[2][NSL] = {{0.0, 0.0}, {0.0, 0.0}};     // sums over the sources of the node magnitudes
        for (int kk = k0; kk <= k; ++kk) {               // uniform trip count
        sync_wait(sync + 2 * W + 4 + kk, 1, P.watchdog, 350 + kk);
        sync_wait(sync + kk, NMW, P.watchdog, 300 + kk);   // coefficients of source item kk published
        const unsigned char* tbl = tabl + (kk % NBUF) * P.tab_fast_bytes;
        const lds_cfp b2l = (lds_cfp)(tbl + P.tab_off_b2);
        const float b2v = b2l[gi];
        lds_c2p cc_[NCC];          // the sample's 10 coefficients (fp64) in this wave's LDS slots: [wave][q][g][16]
#pragma unroll
        for (int q = 0; q < NCC; ++q) {
            // slice reduction (fixed order) + bias of the second Dense: lane gi owns coefficient gi of its sample
            const int s = s_[q];
            const lds_cfp pp = (lds_cfp)(part + (kk % NBUF) * (NSLICE * TS * PSTR)) + ((s >> 4) * 16 + (s & 15)) * PSTR + gi;
            float cmine = pp[0];
#pragma unroll
            for (int w = 1; w < NSLICE; ++w) cmine += pp[w * (R * 16 * PSTR)];
            cmine += b2v;
            const lds_dp cslot = (lds_dp)(cdl + ((wave * 2 + q) * 4 + g) * 16);
            cslot[gi] = (double)cmine;
            cc_[q] = (lds_c2p)cslot;
        }
        const lds_c2p rows2 = (lds_c2p)(tbl);
        const lds_cdp s1of_l = (lds_cdp)(tbl + P.tab_off_s1of), s1inv_l = (lds_cdp)(tbl + P.tab_off_s1inv);
        const lds_cip s1i_l = (lds_cip)(tbl + P.tab_off_s1i);
        const lds_cdp ext_l = (lds_cdp)(smem + L.exttab) + (P.has_ebv ? kk : 0) * TS;
        double ynode_[2][NSL];            // magnitudes at the two sample nodes of every slot
#pragma unroll
        for (int pass = 0; pass < (TWO ? 2 : 1); ++pass) {
            lds_c2p ra_[NSL], rb_[NSL];
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                if constexpr (TWO) {
                    const int j = lo_[u] + pass;
                    int i1 = s1i_l[j];
                    i1 = i1 < 0 ? 0 : i1;                         // (nodes outside the SVD grid lie outside [jlo, jhi]: never bracketed)
                    const int i2 = i1 + 1 < NT ? i1 + 1 : NT - 1;
                    ra_[u] = rows2 + i1 * 6; rb_[u] = rows2 + i2 * 6;
                } else {
                    ra_[u] = rows2 + lo_[u] * 6; rb_[u] = ra_[u] + 6;
                }
            }
            double a0_[NSL], a1_[NSL];
            f64x2 p0_[NSL], p1_[NSL];
#pragma unroll
            for (int u = 0; u < NSL; ++u) { p0_[u] = ra_[u][0]; p1_[u] = rb_[u][0]; }
            f64x2 cq_[NCC];
#pragma unroll
            for (int q = 0; q < NCC; ++q) cq_[q] = cc_[q][0];
#pragma unroll
            for (int jp = 0; jp < 5; ++jp) {
                f64x2 n0_[NSL], n1_[NSL], nq_[NCC];
#pragma unroll
                for (int u = 0; u < NSL; ++u) { n0_[u] = ra_[u][jp + 1]; n1_[u] = rb_[u][jp + 1]; }
                if (jp < 4) {
#pragma unroll
                    for (int q = 0; q < NCC; ++q) nq_[q] = cc_[q][jp + 1];
                }
#pragma unroll
                for (int u = 0; u < NSL; ++u) {
                    const f64x2 cq = cq_[TYPEB ? u : 0];
                    if (jp == 0) { a0_[u] = p0_[u][0] * cq[0]; a1_[u] = p1_[u][0] * cq[0]; }
                    else { a0_[u] = fma(p0_[u][0], cq[0], a0_[u]); a1_[u] = fma(p1_[u][0], cq[0], a1_[u]); }
                }
#pragma unroll
                for (int u = 0; u < NSL; ++u) {
                    const f64x2 cq = cq_[TYPEB ? u : 0];
                    a0_[u] = fma(p0_[u][1], cq[1], a0_[u]); a1_[u] = fma(p1_[u][1], cq[1], a1_[u]);
                }
#pragma unroll
                for (int u = 0; u < NSL; ++u) { p0_[u] = n0_[u]; p1_[u] = n1_[u]; }
                if (jp < 4) {
#pragma unroll
                    for (int q = 0; q < NCC; ++q) cq_[q] = nq_[q];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                const double ya = a0_[u] * p0_[u][0] + p0_[u][1], yb = a1_[u] * p1_[u][0] + p1_[u][1];   // (VA[i,:].c) span[i] + mins[i]
                if constexpr (TWO) {
                    const int j = lo_[u] + pass;
                    // stage 1: ((yb - ya) / dx) * off + ya with the reciprocal of dx from the table
                    ynode_[pass][u] = ((yb - ya) * s1inv_l[j]) * s1of_l[j] + ya;
                } else {
                    ynode_[0][u] = ya; ynode_[1][u] = yb;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NSL; ++u) {    // + extinction of this source's filter (0 without) + distance modulus (model.py:323-342)
            const lds_cdp sc = (lds_cdp)(scal + s_[u] * 8);
            const double dmrc = sc[S_DMOD] + sc[S_RC], ext = ext_l[s_[u]];
            ys_[0][u] += (ynode_[0][u] + ext) + dmrc; ys_[1][u] += (ynode_[1][u] + ext) + dmrc;
        }
        }
        double est_[NSL], m_[NSL];
        bool ul_[NSL];
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const f64x2 tm = D_[u][0], sl = D_[u][1];       // {t, m}, {1/sigma, ln sigma}
            const lds_cdp sc = (lds_cdp)(scal + s_[u] * 8);
            const double izdt = sc[S_IZP1] * (NONUNI ? stl_l[NS + lo_[u]] : inv_dt);
            const double y0 = ys_[0][u], y1 = ys_[1][u];
            const double est = (((y1 - y0) * izdt) * dtx_[u] + y0) / (double)nsrc;     // (a + b [+ c]) / n  (utils.py:566-584)
            double isig = sl[0], lsig = sl[1];
            bool sig_bad = false;
            if constexpr (SYS) {
                const double sd = sl[0];                      // sigma_data
                const double qf = floor(sl[1]), fr = sl[1] - qf;
                const int i0 = (int)qf, i1 = i0 < svl ? i0 + 1 : svl;
                const lds_cdp ep = (lds_cdp)(smem + L.epar) + sv0 * TS + s_[u];
                const double e0 = ep[i0 * TS], e1 = ep[i1 * TS];
                const double e_sys = (e1 - e0) * fr + e0;     // (a single parameter: i0 = i1 = 0, fr = 0 -- exactly e0)
                esys_[u] = e_sys;
                // (1 / sigma_tot and ln sigma_tot without the library's sqrt, division and log: rsqrt_pos / log_pos above.  The exact
                //  shape of these five lines matters to hipcc: taking the log of sigma^2 instead, or the finite test from sqrt(s2),
                //  made the lean kernels spill ~770 registers)
                const double s2 = sd * sd + e_sys * e_sys;    // sigma_tot^2
                const double rs = rsqrt_pos(s2);
                const double sig = s2 * rs;
                const bool fin = (s2 - s2 == 0.0);
                isig = fin ? rs : 0.0;                        // infinite data error: upper limit
                lsig = log_pos(sig);
                sig_bad = (fin & !(s2 > 0)) | (s2 != s2);
            }
            const double x = (tm[1] - est) * isig;
            double v = (-(x * x) / 2.0 - kNormPdfLogC) - lsig;
            opaque(v);                                        // (computed on every lane: no exec-masked region around the chain)
            // outside the model window est = +inf: truncnorm.logpdf(loc = inf) = NaN (em_likelihood.py:252-256)
            v = (inside_[u] & !sig_bad) ? v : dnan();
            ul_[u] = valid_[u] & (isig == 0.0) & !sig_bad;    // infinite data error: an upper limit
            v_[u] = (valid_[u] & !ul_[u]) ? v : 0.0;
            est_[u] = est; m_[u] = tm[1];
            isig_[u] = isig; lsig_[u] = lsig; sbad_[u] = sig_bad;
        }
        if (lim_fin) {
            // truncnorm.logpdf(m, a = -inf, b = (lim - est) / sigma, loc = est, scale = sigma) (em_math.h: detection_term) with the
            // scaled residuals as products with 1 / sigma, like the untruncated term above, and log Phi(b) from the table in LDS
            // (logphi_tab.h; 48.0 -> 37 us per 4096 rows of the CLI-grid case with finite limits: profiles/r05_limits.md)
#pragma unroll
            for (int u = 0; u < NSL; ++u)
                if (valid_[u] & !ul_[u]) {
                    const double b = (it.lim - est_[u]) * isig_[u];
                    const double x = (m_[u] - est_[u]) * isig_[u];
                    double v = ((-(x * x) / 2.0 - kNormPdfLogC) - log_gauss_mass_tab(b, (lds_cdp)(smem + L.nodes))) - lsig_[u];
                    v = x > b ? -dinf() : v;                                               // fainter than the limit, yet detected
                    v_[u] = (inside_[u] & !sbad_[u] & (b > -dinf()) & (x == x)) ? v : dnan();
                }
        }
        gp_[0] = 0.0; gp_[1] = 0.0;
        if (it.has_ul) {                                    // uniform; the term itself only on the lanes that hold a limit
#pragma unroll
            for (int u = 0; u < NSL; ++u)
                if (ul_[u]) {
                    if (P.mass_tab) gp_[u] = upper_limit_term_tab(m_[u], inside_[u] ? est_[u] : dinf(), SYS ? esys_[u] : it.e_const, (lds_cdp)(smem + L.nodes));
                    else gp_[u] = upper_limit_term(m_[u], inside_[u] ? est_[u] : dinf(), SYS ? esys_[u] : it.e_const);
                }
        }
        };
        stage_q();
        lds_dp chi_l = (lds_dp)chi_tot;
        lds_dp gp_l = (lds_dp)gp_tot;
        if constexpr (TYPEB) {
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                const double chi = group_sum(v_[u], 16);
                double gp = 0.0;
                if (it.has_ul) gp = group_sum(gp_[u], 16);
                if (gi == 0) {
                    const int s = s_[u];
                    chi_l[k * TS + s] = chi;
                    gp_l[k * TS + s] = gp;
                    if (chi != chi) bad[s] = 1;
                    if (chi_parts != nullptr && tile0 + s < B) {
                        chi_parts[(long)o * B + tile0 + s] = sample_bad(s) ? dnan() : chi;
                        gp_parts[(long)o * B + tile0 + s] = gp;
                    }
                }
            }
        } else {
            double vacc = v_[0] + v_[1], gacc = gp_[0] + gp_[1];
            if constexpr (LEANX) {
                for (int pp = 1; pp * 32 < nf; ++pp) {      // uniform: only filters with more than 32 points
                    stage_p(pp);
                    stage_q();
                    vacc += v_[0] + v_[1]; gacc += gp_[0] + gp_[1];
                }
            }
            const double chi = group_sum(vacc, 16);
            double gp = 0.0;
            if (it.has_ul) gp = group_sum(gacc, 16);
            if (gi == 0) {
                const int s = s_[0];
                chi_l[k * TS + s] = chi;
                gp_l[k * TS + s] = gp;
                if (chi != chi) bad[s] = 1;
                if (chi_parts != nullptr && tile0 + s < B) {
                    chi_parts[(long)o * B + tile0 + s] = sample_bad(s) ? dnan() : chi;
                    gp_parts[(long)o * B + tile0 + s] = gp;
                }
            }
        }
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k + 1] = clock64();
        for (int kk = k0; kk <= k; ++kk) sync_signal(sync + W + 2 + kk, lane);     // one signal per task and source item
    };

    // ---------------------------------------------------------------------------------
    // dense_task (FASTM == 6; round 6 -- before, a variant of lean_task): every filter has so many points that reconstructing ALL nodes of
    // (item, 16 samples) beats two basis rows per datum (config 4: 12 filters x 200 points).  The four tasks of a UNIT (item k, 16
    // samples) -- task c: samples 4 c .. 4 c + 3, 16 lanes each -- share
    //   * app[node][sample] = (VA[node, :] o span) . c[sample, :] + mins[node] + offset[sample] -- offset = extinction of the item's
    //     filter + distance modulus + K-correction (model.py:390-397) -- on the fp64 matrix cores, 16 nodes x 16 samples per
    //     v_mfma_f64_16x16x4_f64 with K = NC + 2 in three steps (A operands: EmDev::dva, rows on the SAMPLE grid with the stage-1 lerp
    //     folded in, pre-swizzled at create; its column of ones multiplies the sample's offset), a quarter of the node tiles per task,
    //     into one of DENSE_NBUF LDS buffers (hand-off: unit_prod / unit_done counters);
    // and each then walks the data of its four samples, two slots per lane and pass.  What is left per datum is the interpolation and
    // the term themselves, written for the issue port the task shares with the surrogate's f32 MFMA stream (17.5 vector instructions per
    // datum in the ISA; the variant of lean_task: 48):
    //       u = t c1 + c0 (the datum's position on the sample's observer-frame grid in node units: one FMA on per-sample constants),
    //       lo = clamp(int(u)), frac = u - lo, the two node magnitudes (one LDS read of two), est = y0 + (y1 - y0) frac,
    //       x = (m - est) / sigma, term = -x^2 / 2 + k with k = -(ln sqrt(2 pi) + ln sigma) from the record;
    // the window test of np.interp(left = right = inf) -> NaN (em_likelihood.py:252-256) accumulates in a mask, the records are padded to
    // whole passes (no per-slot validity: the padding's term is exactly 0), upper limits sit behind the detections and are evaluated after
    // the loop (log Phi from the table behind the node buffers).
    // ONE task per unit (no redundant B-operand builds) was measured and lost -- two node buffers then keep only two long tasks in
    // flight and the VALU load no longer spreads over the four SIMDs, whose slowest gates the surrogate's items (profiles/r06_c4.md).
    // Deviation from the reference's association order (est from the fraction instead of slope x (t - x0), the offset added inside the
    // matrix product): a few ulp of a magnitude, DEVIATIONS.md; unequally spaced grids take the bracket search and the exact node times.
    // ---------------------------------------------------------------------------------
    auto dense_task = [&](auto sys_tag, auto nonuni_tag, const int k, const int c) {
        constexpr bool SYS = decltype(sys_tag)::value;           // a sampled em_syserr: sigma_tot per (datum, sample)
        constexpr bool NONUNI = decltype(nonuni_tag)::value;     // unequally spaced sample_times: lookup + bisection, node times from the table
        const ItemDesc& it = itab[k];
        const int o = it.o;
        typedef __attribute__((address_space(3))) unsigned char* lds_bp;
        typedef const __attribute__((address_space(1))) unsigned char* gbyte_p;
        typedef const __attribute__((address_space(3))) double* lds_cdp;
        typedef const __attribute__((address_space(3))) float* lds_cfp;
        typedef __attribute__((address_space(3))) double* lds_dp;
        typedef __attribute__((ext_vector_type(2))) double f64x2;
        typedef const __attribute__((address_space(3))) f64x2* lds_c2p;
        if (c == 0) {      // this wave stages the item's [b2 | records] (LDS-DMA, no registers)
            if (k >= NBUF) sync_wait(sync + W + 1 + (k - NBUF + 1), itab[k - NBUF].ntask[R - 1], P.watchdog, 800 + k);
            gbyte_p src = (gbyte_p)(uintptr_t)(P.tabi + (size_t)it.tabi * P.tabi_bytes + P.tab_off_b2);
            lds_bp dst = (lds_bp)(tabl + (k % NBUF) * P.tab_fast_bytes);
            for (int q = 0; q * 1024 < P.tab_fast_bytes; ++q)
                __builtin_amdgcn_global_load_lds(src + q * 1024 + lane * 16, dst + q * 1024, 16, 0, 0);
        }
        const lds_cfp pbuf = (lds_cfp)(part + (k % NBUF) * (NSLICE * TS * PSTR));
        const int jlo = __builtin_amdgcn_readfirstlane(it.jlo), jhi = __builtin_amdgcn_readfirstlane(it.jhi);
        const int ndet = __builtin_amdgcn_readfirstlane(it.pad_i), nul = __builtin_amdgcn_readfirstlane(it.has_ul);
        const int g = lane >> 4, gi = lane & 15;
        const int s = 4 * c + g;                       // this lane group's sample
        const lds_cdp stl_l = (lds_cdp)stl;
        // the A operands of this task's node tiles, requested before the waits for the surrogate and the node buffer
        const int dn_tt = (NS + 15) >> 4;              // node tiles of the sample grid (<= 16: four per task, checked at create)
        double dav[4][3];
        {
            gcf64p dva = as_global(P.dva) + (size_t)it.m * dn_tt * 3 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tt = (c & 3) + 4 * q;
#pragma unroll
                for (int step = 0; step < 3; ++step) dav[q][step] = tt < dn_tt ? dva[(tt * 3 + step) * 64] : 0.0;
            }
        }
        // per-sample constants of the datum loop
        const lds_cdp sc = (lds_cdp)(scal + s * 8);
        const double zp1 = sc[S_ZP1], tsh = sc[S_TS], izp1 = sc[S_IZP1];
        const double t_lo = stl_l[jlo] * zp1 + tsh, t_hi = stl_l[jhi] * zp1 + tsh;
        const double c1 = izp1 * P.st_inv_dt, c0 = -((tsh * izp1 + P.st0) * P.st_inv_dt);
        const int nbis = NONUNI ? __builtin_amdgcn_readfirstlane(P.bg_nbis) : 0;
        const double bg_inv_h = NONUNI ? P.bg_inv_h : 0.0, st0 = P.st0;
        const double e_sys = SYS ? ((lds_cdp)(smem + L.epar))[__builtin_amdgcn_readfirstlane(P.sys_off[o]) * TS + s] : 0.0;
        // A sampled systematic: sum_i ln sigma_tot,i of the band's detections from the item's Chebyshev table in w = ln e (EmDev::lnsig_tab,
        // built and verified at create) -- one series per (item, sample) instead of a logarithm and a reciprocal square root per (datum,
        // sample): the loop below then needs 1 / sigma_tot^2 only.  A sample whose e lies outside the table (or is not positive and
        // finite) sends the whole task through the per-datum form.  Evaluated HERE, before the waits for the surrogate: the table's
        // global-memory latency passes while the task has nothing else to do.
        double lnsig_sum = 0.0;
        bool tab_all = false;
        if constexpr (SYS) {
            gcf64p ltab = as_global(P.lnsig_tab);
            if (ltab != nullptr) {
                const bool pos = (e_sys > 0.0) & (e_sys - e_sys == 0.0);
                const double w = log_pos(pos ? e_sys : 1.0);
                const double uq = (w - P.lnsig_w0) * 2.0;
                const bool in = pos & (uq >= 0.0) & (uq < (double)LNSIG_NI);
                tab_all = __ballot(!in) == 0;
                if (tab_all) {
                    const int iq = (int)uq;
                    const double x2 = 2.0 * (2.0 * (uq - (double)iq) - 1.0);                    // 2 x, x in [-1, 1)
                    gcf64p row = ltab + ((size_t)it.tabi * LNSIG_NI + iq) * 16;
                    double b1 = 0.0, b2 = 0.0;
#pragma unroll
                    for (int n = LNSIG_DEG; n >= 1; --n) { const double b0 = (x2 * b1 - b2) + row[n]; b2 = b1; b1 = b0; }
                    lnsig_sum = ((0.5 * x2) * b1 - b2) + row[0];
                }
            }
        }
        if (c == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sync_signal(sync + 2 * W + 4 + k, lane);     // [b2 | records] of item k staged
        }
        sync_wait(sync + 2 * W + 4 + k, 1, P.watchdog, 350 + k);
#ifndef NMMA_DENSE_SLEEP
#define NMMA_DENSE_SLEEP NMMA_SYNC_SLEEP
#endif
        sync_wait<NMMA_DENSE_SLEEP>(sync + k, NMW, P.watchdog, 300 + k);   // coefficients of item k published
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k] = clock64();
        const unsigned char* tbl = tabl + (k % NBUF) * P.tab_fast_bytes;
        const lds_cfp b2l = (lds_cfp)tbl;
        const int hh = c >> 2;                                 // tasks 4 hh .. 4 hh + 3 share the 16 samples of half hh
        const int unit = R * k + hh;
        int* const unit_prod = sync + 3 * W + 4 + unit;
        int* const unit_done = sync + 3 * W + 4 + R * W + unit;
        const int nrows = (NS + 15) & ~15;
        const lds_dp nb = (lds_dp)(smem + L.nodes) + (unit % DENSE_NBUF) * (nrows * DENSE_STRIDE);
        {
            // B operands: coefficient 4 step + lane / 16 of sample 16 hh + lane % 16 (slice sums in the fixed order, + b2, as fp64);
            // "coefficient" NC is the constant 1 that multiplies mins, NC + 1 the sample's offset that multiplies the column of ones
            const int sj = 16 * hh + (lane & 15), kq = lane >> 4;
            const lds_cdp scj = (lds_cdp)(scal + sj * 8);
            const double off = ((lds_cdp)(smem + L.exttab))[(P.has_ebv ? k : 0) * TS + sj] + (scj[S_DMOD] + scj[S_RC]);
            double bq[3];
#pragma unroll
            for (int step = 0; step < 3; ++step) {
                const int kc = 4 * step + kq;
                const lds_cfp pp = pbuf + ((sj >> 4) * 16 + (sj & 15)) * PSTR + (kc < 16 ? kc : 0);
                float cm = pp[0];
#pragma unroll
                for (int w = 1; w < NSLICE; ++w) cm += pp[w * (R * 16 * PSTR)];
                cm += b2l[kc < 16 ? kc : 0];
                bq[step] = kc < NC ? (double)cm : (kc == NC ? 1.0 : (kc == NC + 1 ? off : 0.0));
            }
            // the buffer's previous unit has been consumed by all four of its tasks
            if (unit >= DENSE_NBUF) sync_wait(unit_done - DENSE_NBUF, 4, P.watchdog, 360 + k);
            typedef double f64x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tt = (c & 3) + 4 * q;                  // this task's node tiles (uniform)
                if (tt >= dn_tt) break;
                f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int step = 0; step < 3; ++step) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(dav[q][step], bq[step], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) nb[(16 * tt + 4 * r + kq) * DENSE_STRIDE + (lane & 15)] = acc[r];
            }
            sync_signal(unit_prod, lane);
            sync_wait(unit_prod, 4, P.watchdog, 370 + k);        // all node tiles of the unit are in LDS
        }
        // ---- the datum loop: detections in passes of 32 (two slots per lane), then the upper limits
        const lds_c2p recs = (lds_c2p)(tbl + P.tab_off_dat - P.tab_off_b2);
        const lds_cdp ncol = (lds_cdp)nb + (s & 15);
        // a datum's bracket on the sample's observer-frame grid and its position inside it
        auto bracket = [&](const double t, int& lo, double& frac) {
            if constexpr (NONUNI) {
                typedef const __attribute__((address_space(3))) int* lds_cip_bg;
                const lds_cip_bg bgl = (lds_cip_bg)(stl_l + 2 * NS);
                int cq = (int)(((t - tsh) * izp1 - st0) * bg_inv_h);
                cq = cq < 0 ? 0 : (cq > BG_CELLS - 1 ? BG_CELLS - 1 : cq);
                lo = bgl[cq] - 1;
                int hi = bgl[cq + 1] + 2;
                lo = lo < jlo ? jlo : lo;
                hi = hi > jhi ? jhi : hi;
                for (int itb = 0; itb < nbis; ++itb) {         // uniform trip count
                    const int mid = (lo + hi) >> 1;
                    const bool le = (stl_l[mid] * zp1 + tsh) <= t;
                    lo = le ? mid : lo;
                    hi = le ? hi : mid;
                }
                lo = lo > jhi - 1 ? jhi - 1 : lo;
                frac = (t - (stl_l[lo] * zp1 + tsh)) * (izp1 * stl_l[NS + lo]);
            } else {
                const double u = fma(t, c1, c0);
                lo = (int)u;                                    // (truncation = floor where it matters: a negative u is clamped anyway)
                lo = lo > jhi - 1 ? jhi - 1 : lo;
                lo = lo < jlo ? jlo : lo;
                frac = u - (double)lo;
            }
        };
        double vacc0 = 0.0, vacc1 = 0.0;
        bool outside = false;
        const int npad = (ndet + 31) & ~31;
        if (SYS && tab_all) {
#pragma unroll 1
            for (int d0_ = 0; d0_ < npad; d0_ += 32) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int dd = d0_ + gi + 16 * u;
                    const f64x2 tm = recs[2 * dd], sk = recs[2 * dd + 1];       // {t, m}, {sigma_data, 0}
                    int lo;
                    double frac;
                    bracket(tm[0], lo, frac);
                    const lds_cdp nd = ncol + lo * DENSE_STRIDE;
                    const double y0 = nd[0], y1 = nd[DENSE_STRIDE];
                    const double est = fma(y1 - y0, frac, y0);
                    outside = outside | !((tm[0] >= t_lo) & (tm[0] <= t_hi));
                    const double s2 = fma(sk[0], sk[0], e_sys * e_sys);
                    double rc = __builtin_amdgcn_rcp(s2);
                    rc = fma(fma(-s2, rc, 1.0), rc, rc);
                    rc = fma(fma(-s2, rc, 1.0), rc, rc);                      // 1 / sigma_tot^2
                    const double r = tm[1] - est;
                    double v = fma((r * -0.5) * r, rc, -kNormPdfLogC);
                    v = dd < ndet ? v : 0.0;
                    if (u == 0) vacc0 += v; else vacc1 += v;
                }
            }
        } else {
#pragma unroll 1
        for (int d0_ = 0; d0_ < npad; d0_ += 32) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int dd = d0_ + gi + 16 * u;
                const f64x2 tm = recs[2 * dd], sk = recs[2 * dd + 1];       // {t, m}, {1 / sigma, k} (sampled: {sigma_data, 0})
                int lo;
                double frac;
                bracket(tm[0], lo, frac);
                const lds_cdp nd = ncol + lo * DENSE_STRIDE;
                const double y0 = nd[0], y1 = nd[DENSE_STRIDE];
                const double est = fma(y1 - y0, frac, y0);
                outside = outside | !((tm[0] >= t_lo) & (tm[0] <= t_hi));
                double v;
                if constexpr (SYS) {
                    // 1 / sigma_tot and ln sigma_tot per (datum, sample): rsqrt_pos / log_pos (em_common.h)
                    const double s2 = sk[0] * sk[0] + e_sys * e_sys;
                    const double rs = rsqrt_pos(s2);
                    const double lsig = log_pos(s2 * rs);
                    const double x = (tm[1] - est) * rs;
                    v = fma(x * -0.5, x, -kNormPdfLogC) - lsig;
                    v = (s2 > 0.0) ? v : dnan();                 // (sigma_tot = 0, or a NaN systematic: scipy's NaN)
                    v = dd < ndet ? v : 0.0;
                } else {
                    const double x = (tm[1] - est) * sk[0];
                    v = fma(x * -0.5, x, sk[1]);
                }
                if (u == 0) vacc0 += v; else vacc1 += v;
            }
        }
        }
        double vacc = vacc0 + vacc1;
        vacc = outside ? dnan() : vacc;
        double gacc = 0.0;
        if (nul) {                                               // uniform; the term itself only on the lanes that hold a limit
            const lds_cdp ptab = (lds_cdp)(smem + L.nodes) + DENSE_NBUF * nrows * DENSE_STRIDE;
            for (int q0 = 0; q0 < nul; q0 += 16) {
                const int dd = q0 + gi;
                if (dd < nul) {
                    const f64x2 tm = recs[2 * (npad + dd)], sk = recs[2 * (npad + dd) + 1];      // {t, m}, {sigma_sys, 0}
                    int lo;
                    double frac;
                    bracket(tm[0], lo, frac);
                    const lds_cdp nd = ncol + lo * DENSE_STRIDE;
                    const double y0 = nd[0], y1 = nd[DENSE_STRIDE];
                    const bool in = (tm[0] >= t_lo) & (tm[0] <= t_hi);
                    const double est = in ? fma(y1 - y0, frac, y0) : dinf();
                    const double e = SYS ? e_sys : sk[0];
                    gacc += P.mass_tab ? upper_limit_term_tab(tm[1], est, e, ptab) : upper_limit_term(tm[1], est, e);
                }
            }
        }
        double chi = group_sum(vacc, 16);
        if constexpr (SYS) chi = tab_all ? chi - lnsig_sum : chi;      // (the band's sum of ln sigma_tot, once per sample)
        double gp = 0.0;
        if (nul) gp = group_sum(gacc, 16);
        if (gi == 0) {
            lds_dp chi_l = (lds_dp)chi_tot;
            lds_dp gp_l = (lds_dp)gp_tot;
            chi_l[o * TS + s] = chi;
            gp_l[o * TS + s] = gp;
            if (chi != chi) bad[s] = 1;
            if (chi_parts != nullptr && tile0 + s < B) {
                chi_parts[(long)o * B + tile0 + s] = sample_bad(s) ? dnan() : chi;
                gp_parts[(long)o * B + tile0 + s] = gp;
            }
        }
        if (dbg && blockIdx.x == 0 && c == 0 && lane == 0) dbg[66 + 2 * k + 1] = clock64();
        sync_signal(unit_done, lane);            // this task no longer reads the unit's node buffer
        sync_signal(sync + W + 2 + k, lane);     // one signal per task
    };

    if constexpr (FAST) {
        sync_wait(sync + W + 1, NVW, P.watchdog, 400);   // prologue data of every likelihood wave in LDS
        if constexpr (WALKF) {
            // ---- the fused MCMC step, first phase (the LAST likelihood wave, before it claims tasks): everything that does not
            // depend on log L -- the chains' uniforms, the two live points of each move, the chains' state -- is loaded NOW and
            // parked in the prologue's staging area (free from here on), so that the epilogue finds it in LDS instead of waiting for
            // a chain of dependent L2 round trips after the tile's last task.  At priority 0: it has all of the launch to finish
            // and must not take issue slots from the MFMA stream.
            // (16 lanes per chain: four rounds -- the last TWO likelihood waves take two each; on one wave the first phase delayed that
            //  wave's tasks enough to lose what the fused step gains on the dense task: config 4's shape with em_syserr)
            if (!helper && vwave >= NVW - WPH) {
                const int wq0 = (NVW - 1 - vwave) * (WNR / WPH);        // this wave's first round
                __builtin_amdgcn_s_setprio(0);
                double* wl = reinterpret_cast<double*>(smem + L.stage);
                double* pl = wl + TS;                         // [5][WNR * 64]: live_j - live_i | u | v | proposal | theta, per (round, lane)
                double* pcd = pl + 5 * WNR * 64;              // [2][TS]: gamma | bound, per chain
                int* pci = reinterpret_cast<int*>(pcd + 2 * TS);      // [6][TS]: inside | active | counts[4]
                if (vwave == NVW - 1) {
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(&wf->priors[0]);
                    for (int j = lane; j < wf->ndim * 10; j += 64) reinterpret_cast<uint32_t*>(wspl)[j] = src[j];
                }
#ifndef NMMA_DBG_WALK_NOPRE
                constexpr int WPR = WNR < 2 ? WNR : 2;       // rounds in flight together
#pragma unroll
                for (int r0 = wq0; r0 < wq0 + WNR / WPH; r0 += WPR) {
                    WalkPre wq[WPR];
                    WalkPreKey wkey[WPR];
                    double wr[WPR][7];
#pragma unroll
                    for (int rr = 0; rr < WPR; ++rr) {   // the pair's state loads first ...
                        const long c = tile0 + (r0 + rr) * WCR + lane / WT;
                        walk_step_pre_a(wf->ndim, c < B ? c : B - 1, lane & (WT - 1), wf->key, wf->u, wf->v, wf->prop, theta, wf->inside, wf->loglstar, wf->counts,
                                        wf->n_steps, (uint64_t)wstep, wq[rr], wkey[rr]);
                    }
#pragma unroll
                    for (int rr = 0; rr < WPR; ++rr)     // ... then the hashes and the live points they address ...
                        walk_step_pre_b(wf->ndim, lane & (WT - 1), wf->live, (long)wf->n_live, wf->first_step + (uint64_t)wstep, wkey[rr], wq[rr], wr[rr]);
#pragma unroll
                    for (int rr = 0; rr < WPR; ++rr) {   // ... then the move's scale; park everything
                        const int r = r0 + rr;
                        walk_step_pre_c(wf->ndim, wr[rr], wq[rr]);
                        const int e = r * 64 + lane;
                        pl[e] = wq[rr].lj - wq[rr].li; pl[WNR * 64 + e] = wq[rr].uu; pl[2 * WNR * 64 + e] = wq[rr].vv; pl[3 * WNR * 64 + e] = wq[rr].pp;
                        pl[4 * WNR * 64 + e] = wq[rr].th;
                        if ((lane & (WT - 1)) == 0) {
                            const int cl = r * WCR + lane / WT;
                            pcd[cl] = wq[rr].gamma; pcd[TS + cl] = wq[rr].lstar;
                            pci[cl] = wq[rr].in0; pci[TS + cl] = wq[rr].active; pci[2 * TS + cl] = wq[rr].cnt0; pci[3 * TS + cl] = wq[rr].cnt1;
                            pci[4 * TS + cl] = wq[rr].cnt2; pci[5 * TS + cl] = wq[rr].cnt3;
                        }
                    }
                }
#endif
                sync_signal(sync + 7 * W + 4, lane);
                switch (P.prio_valu) {
                    case 0: break;
                    case 1: __builtin_amdgcn_s_setprio(1); break;
                    case 2: __builtin_amdgcn_s_setprio(2); break;
                    default: __builtin_amdgcn_s_setprio(3); break;
                }
            }
        }
        // Tasks (item-major) are claimed from one LDS counter: likelihood waves from the start, MFMA-role
        // waves once their record stream is finished.  Any wave may compute any task (results go to
        // per-(item, sample) slots), so the claim order does not affect the values.
        // (task counts are made explicitly wave-uniform: the claim loop must not be compiled as a divergent loop)
        const int ntot = P.n_tasks[R - 1];
        const int* tmap = reinterpret_cast<const int*>(smem + L.tmap);
        int tstat = vwave; (void)tstat;
        int claims = 0;
        for (;;) {
#ifdef NMMA_DBG_STATIC
            int t = tstat; tstat += NVW;
#else
            // every lane issues the LDS add (lane 0 adds 1, the others 0): no divergent control flow around the claim
            int tv;
            {
                const unsigned addr = (unsigned)(uintptr_t)(lds_ip)(sync + 2 * W + 3);
                const int inc = (lane == 0) ? 1 : 0;
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(tv) : "v"(addr), "v"(inc) : "memory");
            }
            int t = __builtin_amdgcn_readfirstlane(tv);
#endif
            if (dbg && blockIdx.x == 0 && lane == 0 && t < 24) { dbg[16 + t] = clock64(); dbg[40 + t] = wave; }
            if (t >= ntot) break;
            const int t_claim = t;
            if (++claims > ntot + 64) {          // cannot happen; fail loudly instead of spinning
                if (lane == 0) { g_ip wd = (g_ip)(uintptr_t)P.watchdog; wd[0] = 1; wd[1] = 900; wd[2] = (int)blockIdx.x * 64 + wave; wd[3] = t; }
                break;
            }
            int k = 0;
            if (ntot <= TMAP_MAX) {              // one LDS read instead of a serial scan over the item list
                const int e = __builtin_amdgcn_readfirstlane(tmap[t]);
                k = e >> 8; t = e & 255;
            } else {
                for (;; ++k) { const int n = __builtin_amdgcn_readfirstlane(itab[k].ntask[R - 1]); if (t < n) break; t -= n; }
            }
#ifndef NMMA_DBG_NOVALU
            if constexpr (EXT) {
                const int kind = itab[k].kind;
                if (kind == NMMA_SYS_PARAM) fast_task(std::integral_constant<int, 1>{}, k, t);
                else if (kind == NMMA_SYS_NODES) fast_task(std::integral_constant<int, 2>{}, k, t);
                else fast_task(std::integral_constant<int, 0>{}, k, t);
            } else {
                // (round 6: no task is two-stage any more -- where sample_times differ from the SVD grid the handle's table holds rows ON
                //  the sample grid with the stage-1 lerp folded in, nmma_em_create -- so every lean flavour reconstructs two rows per
                //  datum; the TWO = true forms of the lambdas are no longer instantiated: half the inlined task variants per kernel)
                #ifdef NMMA_DBG_NO_SYS_VARIANTS      // measurement build: the constant-systematics variants alone in the kernel
                constexpr bool sysp = false;
#else
                const bool sysp = LEANX && (itab[k].kind == NMMA_SYS_PARAM || ((FASTM == 5 || COMB) && itab[k].kind == NMMA_SYS_NODES));
#endif
                auto run = [&](auto tb) {
                    using T = std::true_type; using F = std::false_type;
                    if constexpr (FASTM == 4 || FASTM == 8) {          // unequally spaced sample_times: bracket by lookup + bisection
                        if (sysp) lean_task(tb, F{}, T{}, T{}, k, t); else lean_task(tb, F{}, F{}, T{}, k, t);
                    } else if constexpr (FASTM == 5) {
#ifdef NMMA_DBG_GEN_PLAIN_ONLY      // measurement build: the general lean task's constant-systematics, equally-spaced variants alone in the kernel
                        lean_gen_task(tb, F{}, F{}, F{}, k, t);
#else
                        if (!P.st_uniform) { if (sysp) lean_gen_task(tb, F{}, T{}, T{}, k, t); else lean_gen_task(tb, F{}, F{}, T{}, k, t); }
                        else if (sysp) lean_gen_task(tb, F{}, T{}, F{}, k, t);
                        else lean_gen_task(tb, F{}, F{}, F{}, k, t);
#endif
                    } else if constexpr (FASTM == 3 || FASTM == 7) {
                        if (sysp) lean_task(tb, F{}, T{}, F{}, k, t); else lean_task(tb, F{}, F{}, F{}, k, t);
                    } else if constexpr (FASTM == 6) {   // dense: more than 16 points in every filter (host); any sample grid -- the
                        // stage-1 lerp lives in the A operands, an unequally spaced grid only changes how a datum finds its bracket
                        if (P.st_uniform) { if (sysp) dense_task(T{}, F{}, k, t); else dense_task(F{}, F{}, k, t); }
                        else { if (sysp) dense_task(T{}, T{}, k, t); else dense_task(F{}, T{}, k, t); }
                    } else if constexpr (FASTM == 9) {   // ... its constant-systematics, equally-spaced variant alone (the launcher checks)
                        dense_task(F{}, F{}, k, t);
                    } else {
                        lean_task(tb, F{}, F{}, F{}, k, t);
                    }
                };
                if constexpr (DENSE) run(std::false_type{});
                else if (itab[k].nf <= 16) run(std::true_type{}); else run(std::false_type{});
            }
#else
            sync_wait(sync + k, NMW); sync_signal(sync + W + 2 + k, lane);
#endif
            if (dbg && blockIdx.x == 0 && lane == 0 && t_claim < 24) dbg[104 + t_claim] = clock64();     // (after the task's done-signal)
        }
    } else {
        for (int k = 0; k < W; ++k) {
            // every likelihood wave finished its previous phase (prologue data; LDS table buffer free)
            sync_wait(sync + W + 1 + k, NVW, P.watchdog, 500 + k);
            sync_wait(sync + k, NMW, P.watchdog, 600 + k);   // coefficients of item k published
            if (dbg && blockIdx.x == 0 && vt == 0) dbg[66 + 2 * k] = clock64();
            if (NC == 10) item_phase(std::integral_constant<int, 10>{}, k);   // the reference default
            else item_phase(std::integral_constant<int, 0>{}, k);
            if (dbg && blockIdx.x == 0 && vt == 0) dbg[66 + 2 * k + 1] = clock64();
            sync_signal(sync + W + 2 + k, lane);
        }
    }
    // ---- sum over filters + floor (core/base.py:178-182)
    bool own_totals = false;     // (wave-uniform; first likelihood wave)
    if (vwave == 0) {            // the first likelihood wave (helpers have vwave < 0)
        for (int k = 0; k < W; ++k) sync_wait(sync + W + 2 + k, all_fast ? itab[k].ntask[R - 1] : NVW, P.watchdog, 700 + k);
        const int nb = SPLITTABLE ? P.n_bands : 1;
        own_totals = !SPLITTABLE || nb <= 1;
        if constexpr (COMB) {
            if (aux.gap_rows != nullptr && vt < TS && tile0 + vt >= B) aux.gap_rows[tile0 + vt] = 0;      // (the flag array is padded to whole tiles)
        }
        if (vt < TS && tile0 + vt < B) {
            bool isbad = always_floor != 0 || bad[vt] != 0 || sample_bad(vt) || g_wd_trip != 0;
            if constexpr (COMB) {       // a sub-model delivered no light curve for this row (model.py:1423-1426)
                if (aux.bad_rows != nullptr && aux.bad_rows[tile0 + vt] != 0) isbad = true;
                // (a flagged row's value -- NaN terms from the nodes without a value, hence the floor -- is replaced by the materialising
                //  kernels that follow this launch; they floor the row themselves where that is the answer)
                if (aux.gap_rows != nullptr) aux.gap_rows[tile0 + vt] = bad[5 * TS + vt] != 0 ? 1 : 0;
                else if (bad[5 * TS + vt] != 0 && always_floor == 0 && !sample_bad(vt) && !(aux.bad_rows != nullptr && aux.bad_rows[tile0 + vt] != 0)) {
                    // the caller promised curves without interior gaps (NMMA_STACK2_GAP_FREE: no re-evaluation launch follows) and this
                    // row met one: the handle is poisoned -- every later call fails with a message -- instead of a silent floor
                    g_ip wd = (g_ip)(uintptr_t)P.watchdog;
                    wd[0] = 1; wd[1] = 950; wd[2] = (int)blockIdx.x; wd[3] = vt;
                }
            }
            if (!SPLITTABLE || nb <= 1) {
                double c = 0.0, g = 0.0;             // running sums in item (= observed-filter) order
                for (int k = 0; k < W; ++k) { c += chi_tot[k * TS + vt]; g += gp_tot[k * TS + vt]; }
                double tot = c + g;
                if (isbad || !(tot - tot == 0.0)) tot = NMMA_LOGL_FLOOR;
                out[tile0 + vt] = tot;
                if constexpr (WALKF) reinterpret_cast<double*>(smem + L.stage)[vt] = tot;      // (the prologue's staging area is free by now)
            } else {
                // split launch: this workgroup owns a GROUP of one to three adjacent bands.  A band's sums sit in the slot of its
                // observed filter (lean_task) or of its last work item (lean_gen_task, FASTM 5); they are parked per observed filter
                // in the workspace [2][P.O][B] that gp_parts points at; NaN marks a bad sample
                constexpr bool SLOT_O = FASTM != 5;
                for (int k = 0; k < W; ++k) {
                    if (itab[k].ks != itab[k].nsrc - 1) continue;
                    const int o = itab[k].o;
                    const int slot = (SLOT_O ? o : k) * TS + vt;
                    gp_parts[(long)o * B + tile0 + vt] = isbad ? dnan() : chi_tot[slot];
                    gp_parts[((long)P.O + o) * B + tile0 + vt] = gp_tot[slot];
                }
            }
        }
        if (SPLITTABLE && nb > 1) {
            // The group that arrives LAST at its tile's counter adds the bands in band order -- the running sums of the fused
            // epilogue above, bit for bit -- and re-arms the counter for the next launch.  (Counters: the 64 KiB in front of the
            // workspace, zeroed when it is allocated.  Release / acquire at agent scope: the bands of a tile run on any XCD.)
            unsigned* cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(gp_parts) - SPLIT_COUNTER_BYTES) + blockIdx.x;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            unsigned prev = 0;
            if (vt == 0) prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            prev = __builtin_amdgcn_readfirstlane(prev);
            if (prev == (unsigned)(nb - 1)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (vt == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (vt < TS && tile0 + vt < B) {
                    double c = 0.0, g = 0.0;
                    const int nO = P.O;
                    for (int y = 0; y < nO; ++y) {
                        c += gp_parts[(long)y * B + tile0 + vt];
                        g += gp_parts[((long)nO + y) * B + tile0 + vt];
                    }
                    double tot = c + g;
                    if (always_floor != 0 || !(tot - tot == 0.0)) tot = NMMA_LOGL_FLOOR;
                    out[tile0 + vt] = tot;
                    if constexpr (WALKF) reinterpret_cast<double*>(smem + L.stage)[vt] = tot;
                }
                own_totals = true;       // (split launch: the group that added the bands owns the tile's MCMC step)
            }
        }
    }
    if constexpr (WALKF != 0) {
        // ---- the MCMC step, second phase: decide, move, propose, leave the tile's theta rows ready for the next launch.  One ROUND
        // (64 / WT chains, a group of WT lanes each) per likelihood wave, WNR waves side by side: on the one wave that sums the tile
        // the rounds ran one after the other -- two for 8 lanes per chain; four for 16, which then cost what the separate walk launch
        // costs (profiles/r04_fused_mcmc_step.md).  The first likelihood wave publishes "totals parked" (1: this workgroup owns the
        // tile's step; 2: split launch, another group of bands does) through an LDS word; LDS operations of a wave execute in order,
        // so the totals are there when the word is.
        if (vwave >= 0 && vwave < WNR) {
            int* const wgo = sync + 7 * W + 5;
            int own = own_totals ? 1 : 2;
            if (vwave == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store((lds_ip)wgo, own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                sync_wait(wgo, 1, P.watchdog, 910);
                own = __hip_atomic_load((lds_ip)wgo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (own == 1) {
                double* totl = reinterpret_cast<double*>(smem + L.stage);
                sync_wait(sync + 7 * W + 4, WPH, P.watchdog, 900);        // (the first phase finished long ago)
                const double* pl = totl + TS;
                const double* pcd = pl + 5 * WNR * 64;
                const int* pci = reinterpret_cast<const int*>(pcd + 2 * TS);
                const int r = vwave;
                const int e = r * 64 + lane, cl = r * WCR + lane / WT;
                WalkPre w;
                w.li = 0.0; w.lj = pl[e]; w.uu = pl[WNR * 64 + e]; w.vv = pl[2 * WNR * 64 + e]; w.pp = pl[3 * WNR * 64 + e]; w.th = pl[4 * WNR * 64 + e];
                w.gamma = pcd[cl]; w.lstar = pcd[TS + cl];
                w.in0 = pci[cl]; w.active = pci[TS + cl]; w.cnt0 = pci[2 * TS + cl]; w.cnt1 = pci[3 * TS + cl]; w.cnt2 = pci[4 * TS + cl];
                w.cnt3 = pci[5 * TS + cl];
#ifndef NMMA_DBG_WALK_NOPOST
                const long c = tile0 + cl;
                // (WCON: the Constraint program's evaluation stack, one per chain of the round, in the partial-sum ring -- every task is done)
                double* const cstack = reinterpret_cast<double*>(smem + L.part) + (size_t)vwave * (NMMA_CON_MAX_STACK * WCR);
                if (c < B)
                    walk_step_post<WCON>(wspl, wf->ndim, WT, c, lane & (WT - 1), totl[cl], w, wf->u, wf->v, wf->logl, wf->counts, wf->prop,
                                         const_cast<double*>(theta), wf->inside, wf->con_ops, wf->n_con_ops, !wlast, cstack + lane / WT, WCR);
#else
                if (lane == 0 && vwave == 0) wf->counts[0] = w.cnt0 + (int)w.gamma;      // (keep the first phase alive)
#endif
            }
        }
    }
}

template <int R, int KP, int NMW, int NVW, int FAST, int WALKF>
hipError_t launch_logl_one(nmma_em_handle* h, const double* theta, int64_t B, int64_t ld, double* out,
                                  double* chi, double* gp, hipStream_t s, const nmma_walk_fuse* wf, uint64_t wstep, int wlast, EmAux aux) {
    constexpr int LOGL_THREADS = logl_threads(NMW, NVW);
    const EmDev& P = h->dev;
    const LdsW L = lds_layout_logl(R, lds_ns_arg(P), h->nf_avg_max, P.tab_bytes, P.tab_fast_bytes, P.n_items, P.M, P.NP, P.all_fast, P.n_data, P.n_sys_slots,
                                   (P.all_fast == 1 && (P.lean_x || FAST == 7 || FAST == 8)) ? (P.has_ebv ? P.n_items : 1) : 0, h->ring_max, P.dense ? 0 : (P.dat_in_tab ? 8 : 32),
                                   P.dense ? ((P.NS + 15) & ~15) : 0, (FAST == 7 || FAST == 8) ? STACK2_LDS_BYTES + (P.mass_tab2 ? LOGPHI_LDS_BYTES : 0) : ((FAST == 1 || FAST == 3 || FAST == 4 || FAST == 5 || FAST == 6 || FAST == 9) && P.mass_tab) ? LOGPHI_LDS_BYTES : 0,
                                   ((WALKF & 31) == 16 || (WALKF != 0 && R == 2)) ? (WALKF & 31) : 0);
    const int TS = 16 * R;
    g_launch_note.clear();
    if (L.total > LDS_DYNAMIC_MAX) {
        g_launch_note = "em_logl: the configuration needs " + std::to_string(L.total) + " bytes of LDS, more than 159 KiB (ring depth " +
                        std::to_string(L.nbuf) + ", staged table " + std::to_string(P.tab_fast_bytes) + " bytes per item)";
        return hipErrorInvalidValue;
    }
    // Small batches: a tile's workgroup walks ALL work items, so anything up to 256 tiles costs the latency of one workgroup
    // (26-27 us for BASELINE config 2).  While the grid leaves CUs idle the bands are dealt to workgroups of their own --
    // grid (tiles, bands), each running this very kernel on a copy of the configuration that holds one band's items -- and
    // the band that finishes a tile last adds the per-band sums in band order (the order of the fused epilogue: same bits).
    // Past 256 / bands tiles one band per workgroup would queue workgroups behind each other; groups of two or three adjacent bands
    // keep every workgroup resident at once up to 256 / 2 tiles (measured: profiles/r03_small_batch.log).
    const int n_bands = h->lvl_n[0];
    const long tiles = (long)((B + TS - 1) / TS);
    const int lvl = (WALKF && !h->walk_split) ? -1 : split_level(h, R, FAST, B, chi != nullptr);
    const bool split = lvl >= 0;
    const int n_groups = split ? h->lvl_n[lvl] : 1;
    if (split) {
        const int64_t need = 2 * (int64_t)n_bands * B;
        if (need > h->split_cap) {
            if (h->split_ws) (void)hipFree(h->split_ws);
            h->split_ws = nullptr; h->split_cap = 0;
            const int64_t cap = std::max<int64_t>(need, 2 * (int64_t)n_bands * 1024);
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->split_ws), SPLIT_COUNTER_BYTES + (size_t)cap * sizeof(double));
            // (the arrival counters start at zero; every launch leaves them at zero again)
            if (e == hipSuccess) e = hipMemsetAsync(h->split_ws, 0, SPLIT_COUNTER_BYTES, s);
            if (e != hipSuccess) { g_launch_note = "split workspace"; return e; }
            h->split_cap = cap;
        }
    }
    typename em_aux_of<FAST>::type kaux{};
    if constexpr (FAST == 7 || FAST == 8) kaux = aux;
    const dim3 grid((unsigned)tiles, (unsigned)n_groups);
    h->g_x = grid.x; h->g_y = grid.y; h->g_block = LOGL_THREADS; h->g_tile = TS; h->g_lds = L.total;
    if (L.total > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&em_logl<R, KP, NMW, NVW, FAST, WALKF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
        if (e != hipSuccess) {
            g_launch_note = "hipFuncSetAttribute(em_logl, max dynamic LDS = " + std::to_string(L.total) + ")";
            return e;
        }
    }
    if (split) {
        double* band_ws = reinterpret_cast<double*>(reinterpret_cast<char*>(h->split_ws) + SPLIT_COUNTER_BYTES);
        hipLaunchKernelGGL((em_logl<R, KP, NMW, NVW, FAST, WALKF>), grid, dim3(LOGL_THREADS), L.total, s, h->band_dev_d + h->lvl_off[lvl], theta, (long)B, (long)ld, L,
                           h->always_floor, out, static_cast<double*>(nullptr), band_ws, h->dbg, wf, (unsigned long long)wstep, wlast, kaux);
    } else {
        hipLaunchKernelGGL((em_logl<R, KP, NMW, NVW, FAST, WALKF>), grid, dim3(LOGL_THREADS), L.total, s, h->dev_d, theta, (long)B, (long)ld, L,
                           h->always_floor, out, chi, gp, h->dbg, wf, (unsigned long long)wstep, wlast, kaux);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        g_launch_note = "em_logl<" + std::to_string(R) + "," + std::to_string(KP) + ",..," + std::to_string(FAST) + ">: grid " +
                        std::to_string(grid.x) + ", " + std::to_string(L.total) + " bytes of LDS, ring depth " + std::to_string(L.nbuf);
    return e;
}

#define NMMA_LOGL_INSTANCE(R, KP, NMW, FASTM, WALKF)                                                                          \
    template hipError_t launch_logl_one<R, KP, NMW, 8, FASTM, WALKF>(nmma_em_handle*, const double*, int64_t, int64_t, double*, double*, \
                                                                     double*, hipStream_t, const nmma_walk_fuse*, uint64_t, int, EmAux)
// a flavour at both tile sizes and both parameter-count classes (+ its fused MCMC step at 16-sample tiles)
#define NMMA_LOGL_FLAVOUR(NMW, FASTM)          \
    NMMA_LOGL_INSTANCE(1, 1, NMW, FASTM, 0);   \
    NMMA_LOGL_INSTANCE(1, 2, NMW, FASTM, 0);   \
    NMMA_LOGL_INSTANCE(2, 1, NMW, FASTM, 0);   \
    NMMA_LOGL_INSTANCE(2, 2, NMW, FASTM, 0)
// (the fused MCMC step: 8 / 16 lanes per chain -- up to 8 / 16 sampled dimensions; + 64: with the chains' Constraint program)
#define NMMA_LOGL_WALK(FASTM)                  \
    NMMA_LOGL_INSTANCE(1, 1, 8, FASTM, 8);     \
    NMMA_LOGL_INSTANCE(1, 2, 8, FASTM, 8);     \
    NMMA_LOGL_INSTANCE(1, 1, 8, FASTM, 16);    \
    NMMA_LOGL_INSTANCE(1, 2, 8, FASTM, 16)
#define NMMA_LOGL_WALK_CON(FASTM)              \
    NMMA_LOGL_INSTANCE(1, 1, 8, FASTM, 72);    \
    NMMA_LOGL_INSTANCE(1, 2, 8, FASTM, 72);    \
    NMMA_LOGL_INSTANCE(1, 1, 8, FASTM, 80);    \
    NMMA_LOGL_INSTANCE(1, 2, 8, FASTM, 80)
// (32-sample tiles -- queues beyond one round of 16-sample tiles, 4096 chains: 16 lanes per chain only, no Constraint program, not the
//  dense task; nmma_em_loglike_walk says why)
#define NMMA_LOGL_WALK2(FASTM)                 \
    NMMA_LOGL_INSTANCE(2, 1, 8, FASTM, 16);    \
    NMMA_LOGL_INSTANCE(2, 2, 8, FASTM, 16)

}  // namespace nmma
