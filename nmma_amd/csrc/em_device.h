// em_device.h -- device-side view of one likelihood (built once by nmma_em_create).
#pragma once

#include <stdint.h>

#include "../../include/nmma_hip.h"

namespace nmma {

// Floats per hidden-block record of the pre-swizzled surrogate weights:
//   [0, 256)            W2 fragment : lane l, r in 0..3 -> W2[hb*16 + (l>>4)*4 + r][coef = l&15]
//   [256, 256+64*KP)    W1 fragments: kp, lane l       -> W1[p = 4*kp + (l>>4)][hb*16 + (l&15)]
//   [.., +16)           b1[hb*16 .. hb*16+15]
// One record feeds (1 or 2) + 4 MFMAs per 16-sample row block; a wave streams its
// records linearly with 16-byte loads (no LDS staging: every wave owns its hidden slice).
__host__ __device__ constexpr int rec_floats(int kp) { return 256 + 64 * kp + 16; }

enum EmMode : int32_t { MODE_LOGL = 0, MODE_COEFF = 1, MODE_LC = 2, MODE_LC_ABS = 3 };

// One work item of em_logl: (observed filter o, its ks-th source model filter m), with
// everything the downstream phase needs about it (copied to LDS once per workgroup).
constexpr int BG_CELLS = 256;

struct ItemDesc {
    int32_t o, ks, m, nsrc;
    int32_t G, d0, nf, kind;
    int32_t jlo, jhi, identity, same_grid;
    double lim, e_const, ebvc;
    int32_t tabi, pad_i;      // item-staged photometry (EmDev::dat_in_tab): this item's table in EmDev::tabi (survives the band split's re-indexing)
    int32_t fast, has_ul;
    int32_t ntask[2];         // fast modes: tasks of this item per tile of 16 / 32 samples (index R - 1)
};
static_assert(sizeof(ItemDesc) == 24 * 4, "ItemDesc must be ITEM_WORDS words");

struct EmDev {
    // dims
    int32_t M, NP, KP, NH_pad, HB, NC, NT, NS, D, O;
    int32_t n_cosmo, redshift_mode, has_ebv, kmax;
    int32_t ext_law, pad_ext;   // enum nmma_extinction_law; for P92 the ebv_coeff array / ItemDesc::ebvc hold filter_nu0
    int32_t st_uniform, pad0;     // sample_times equally spaced: bracket guess by division
    double st0, st_inv_dt;
    // surrogate
    const float* wrec;        // [M][HB + NPAD_REC][rec_floats(KP)]   (zero records: branch-free prefetch)
    int32_t wrec_bytes;
    int32_t prio_valu, prio_mfma;   // s_setprio of the two roles of em_logl (NMMA_EM_PRIO="v,m"; default 3,0)
    int* watchdog;            // [4] device words: {tripped, code, workgroup*64+wave, value*65536+target} (em_logl hand-off waits)
    int32_t helpers;          // MFMA-role waves join the likelihood workers after their stream (NMMA_EM_HELPERS, default 1)
    int32_t all_fast;         // every work item takes em_logl's fast path: 1 = lean task (constant systematics, <= 32 points per
                              // filter, photometry staged in LDS), 2 = extended task (else 0: generic item phase)
    const float* b2;          // [M][16]
    const double* VAt;        // [M][NC][NT]   (transposed: coalesced along the time grid; MODE_LC)
    const double* VA;         // [M][NT][NC]   (rows gathered per datum; MODE_LOGL)
    const double* mins;       // [M][NT]
    const double* span;       // [M][NT]   maxs - mins
    const double* pmin;       // [M][NP]
    const double* pspan;      // [M][NP]   param_maxs - param_mins
    const double* pinv;       // [M][NP]   1 / pspan (em_logl normalises with one multiply)
    const double* ebv_coeff;  // [M]
    const double* ext_tab;    // [B][M] extinction magnitudes of the current batch (p92_tab; written by ext_prepass_kernel)
    // stage-1 interpolation tables (sample_times <- tt), static per model filter
    const double* st;         // [NS] sample times
    const int32_t* s1_idx;    // [M][NS]  left node in tt (-1: outside the SVD grid -> +inf)
    const double* s1_dx;      // [M][NS]  tt[i+1] - tt[i]
    const double* s1_off;     // [M][NS]  st[j] - tt[i]
    const int32_t* s1_range;  // [M][4]   jlo, jhi (finite nodes), identity flag, n_finite
    // cosmology
    const double* dist_grid;
    const double* z_grid;
    // theta mapping
    nmma_slot model_param[NMMA_MAX_PARAMS];
    nmma_slot lumdist, redshift, timeshift, ebv;
    nmma_slot hubble;         // sampled Hubble constant: the z(d_L) grid is read at d_L * H0 * inv_h0_ref (has_h0)
    double inv_h0_ref;
    int32_t has_h0;
    int32_t n_bands;       // > 1 only in the per-band copies of the split launch of small batches (grid.y = n_bands)
    // photometry (CSR over observed filters)
    const int32_t* doff;      // [O+1]
    const double* dt;         // [N]
    const double* dm;         // [N]
    const double* dsig;       // [N]
    const double* dsigtot;    // [N]  sqrt(sig^2 + e^2) for NMMA_SYS_CONST filters
    const double* dlogsig;    // [N]  log of the above
    const double* dinvsig;    // [N]  1 / dsigtot (0 for upper limits: infinite sigma)
    const double* dat4;       // [N][4]  {t, m, 1/sigma_tot, log sigma_tot} per datum (lean fast task: two 16-byte LDS reads)
    const double* lim;        // [O]
    const int32_t* nsrc;      // [O]
    const int32_t* src;       // [O][3]
    const int32_t* group;     // [O]  lanes cooperating on one sample (power of two <= 64)
    // packed per-model-filter static tables staged in LDS by em_logl:
    //   [rows NT x RS f64 (VA row | span | mins, RS = NC+2 rounded up to even) | s1_dx NS f64 | s1_off NS f64 |
    //    s1_idx NS i32 | b2 16 f32], 1-KiB padded
    const unsigned char* tab;
    const int32_t* task_map[2];   // fast mode, per tile size R = 1, 2: task index -> (item << 8 | sample chunk)
    int32_t n_tasks[2];
    int32_t n_data;           // total number of photometry points (all observed filters)
    int32_t tab_fast_bytes;   // 1-KiB-rounded prefix [rows | b2] staged per item by em_logl's fast mode
    int32_t tab_bytes, tab_row_stride, tab_off_s1dx, tab_off_s1of, tab_off_s1i, tab_off_b2;
    int32_t lean_x, p92_tab;          // p92_tab: the P92 extinction magnitudes come from ext_tab (pre-pass kernel) -- lean task           // lean task extras needed (a filter with more than 32 points, a sampled em_syserr): em_logl<.., 3>
    int32_t tab_off_s1inv, any_two;   // 1 / s1dx (lean two-stage task); some item's sample_times differ from the SVD grid
    const ItemDesc* item_desc;   // [n_items]
    int32_t lc_nf_max, model_kind;   // widest observed filter; enum nmma_model_kind
    const double* nu0;           // [M] filter frequencies (Hz) for analytic blackbody models
    // work items of em_logl: (observed filter, source index, model filter, n sources) x n_items
    const int32_t* items;
    int32_t n_items, n_sys_slots;
    // systematics
    const int32_t* sys_kind;  // [O]
    const double* sys_const;  // [O]
    const int32_t* sys_nn;    // [O]
    const int32_t* sys_off;   // [O+1]
    const nmma_slot* sys_slots;
    const int32_t* sys_nidx;  // [N]  node bracket per datum (-1: left of first, K-1: at/after last)
    const double* sys_ndx;    // [N]  node spacing
    const double* sys_noff;   // [N]  t - node time
    const double* sys_node_t; // [n_sys_slots] node times (NMMA_SYS_NODES groups; the finite mask on sampled node values)
    const int32_t* d_item;    // [N] first work item of each datum's observed filter (em_lc_loglike's flat pass over the photometry)
    // Item-staged photometry (lean task with so much photometry that staging ALL of it leaves room for one ring slot only,
    // BASELINE config 4): LDS then holds the epochs of all points (stage P) and each ring slot the {t, m, 1/sigma, ln sigma}
    // records of its own item behind the basis rows -- tabi[k] = [first tab_off_dat bytes of tab[m_k] | records of item k].
    const unsigned char* tabi;
    int32_t tabi_bytes, dat_in_tab, tab_off_dat;
    const double* dva;        // dense lean task: [M][ceil(NT/16)][3][64] MFMA A operands, A[node lane % 16][k = 4 step + lane / 16] of
                              // [VA o span | mins | 0] (the constant-1 "coefficient" NC adds mins)
    int32_t dense;            // dense lean task (em_logl<.., 6>): every filter has so many points that reconstructing ALL nodes of
                              // (item, 16 samples) on the fp64 matrix cores beats two basis rows per datum; implies dat_in_tab
    // Unequally spaced sample_times: a coarse lookup over BG_CELLS equal cells of [st[0], st[NS-1]] narrows a datum's bracket
    // before the bisection -- bguess[q] = last node at or before the left edge of cell q (bguess[BG_CELLS] = NS - 1) --
    // so that the lean tasks bisect bg_nbis times (3 for the CLI's 150 log-spaced nodes) instead of ceil(log2 NS) = 8.
    const int32_t* bguess;
    double bg_inv_h;
    int32_t bg_nbis;
    int32_t mass_tab2;        // combined-model flavour (em_logl<.., 7 | 8>): log Phi for its upper limits from the table it keeps behind the flux-sum table
    int32_t lean_gen, mass_tab;       // (mass_tab: a band has a finite detection limit -- the general lean task reads log Phi from the table it keeps in LDS at LdsW::nodes, logphi_tab.h)  general lean task (averaged bands: several source filters per observed filter; time-node systematics): em_logl<.., 5>
    // Pei-1992 extinction of the lean task without the pre-pass launch: per model filter [z_mid, 1 / z_half, c_0 .. c_13] -- Chebyshev
    // coefficients of the extinction magnitude PER UNIT E(B-V) over the handle's redshift range (nmma_em_create: built and verified
    // against em_math.h:p92_smc_ext_mag on a dense grid; null where the range or the accuracy does not allow it: ext_tab then)
    const double* p92_cheb;   // [M][16]
    // Combined model on a UNION grid (nmma_em_config::base_times): the surrogate lives on its own sample_times bt[NB]; a node of the
    // handle's grid st[NS] is np.interp between two of those (autocomplete_data(..., extrapolate=inf), model.py:1440-1448), each of
    // which is the stage-1 lerp between two SVD nodes.  The lean task has both hops in its basis rows; em_fused (curve outputs, the
    // re-evaluation launch) evaluates them from these tables.  union_grid = 0: all null.
    int32_t union_grid, NB;
    const double* bt;         // [NB]      the surrogate's own sample_times
    const int32_t* b_idx;     // [M][NB]   left node in tt of base node i (-1: outside the SVD grid)
    const double* b_dx;       // [M][NB]   tt[a+1] - tt[a]
    const double* b_off;      // [M][NB]   bt[i] - tt[a]
    const int32_t* u_idx;     // [M][NS]   left base node of grid node j (-1: outside the surrogate's finite window -> +inf)
    const double* u_dx;       // [M][NS]   bt[i+1] - bt[i]
    const double* u_off;      // [M][NS]   st[j] - bt[i]  (0: the grid node IS base node i)
    int32_t n_items_null, pad_null;   // combined-model flavours: the first n_items_null work items are null filters (no record stream: mfma_role<.., SKIPNULL>)
    // Dense task under a sampled systematic (one parameter per filter or shared; em_logl<.., 6> -- constant systematics take <.., 9>): sum_i
    // ln sigma_tot,i of a band's detections is a function of the sampled value e alone -- F(w) = 1/2 sum_i ln(sigma_data,i^2 + e^2), w = ln e,
    // analytic in a strip of half-width pi / 2 about the real axis whatever the data -- tabulated per item as Chebyshev series of degree 14 on
    // LNSIG_NI intervals of width 1/2 in w (built and verified at nmma_em_create; null: not built, or a band failed the check): the datum
    // loop then needs 1 / sigma_tot^2 only -- no square root, no logarithm per (datum, sample).
    const double* lnsig_tab;  // [n_items][LNSIG_NI][16]
    double lnsig_w0;          // left end of the first interval (in w = ln e)
};
constexpr int LNSIG_NI = 34, LNSIG_DEG = 14;      // e from 1e-4 to 2.4e3

}  // namespace nmma
