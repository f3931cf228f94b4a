// em_host.h -- host side shared by the EM translation units of libnmma_hip.so: the handle behind nmma_em_create, error
// plumbing, and the launcher of em_logl, whose instantiations are spread over em_logl_*.hip (one unit compiled the 38
// instantiations in 170 s; the units build concurrently).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>


#include "em_common.h"
#include "nmma_common.h"

namespace nmma {

extern thread_local std::string g_launch_note;      // what a launcher found wrong before it returned a HIP error code (defined in em_api.inc)
#define NM_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t _e = (call);                                                             \
        if (_e != hipSuccess)                                                               \
            return fail(std::string(#call) + " failed: " + hipGetErrorString(_e) +          \
                        (g_launch_note.empty() ? std::string() : " (" + g_launch_note + ")"));  \
    } while (0)

}  // namespace nmma

struct nmma_em_handle {
    nmma::EmDev dev{};
    nmma::EmDev* dev_d = nullptr;   // device copy read by the kernels
    int device = 0;
    int always_floor = 0;
    int nf_avg_max = 0;
    int n_obs_total = 0;
    int64_t flops_per_eval = 0;
    std::vector<void*> owned;
    // workspaces (grown on demand, outside any capture)
    double* chi = nullptr;
    double* gp = nullptr;
    int64_t parts_cap = 0;
    double* lc_ws = nullptr;        // [B][M][NS] model light curves (non-SVD models; the surrogate's curves of rows nmma_em_loglike_stack2 re-evaluates)
    // run-time options (nmma_em_set_option; the environment variables of the same purpose are read ONCE, at nmma_em_create:
    // a getenv per launch raced with the host program's own environment writes)
    int walk_fuse = 1;              // "walk_fuse"  / NMMA_WALK_NO_FUSE : the MCMC step fused into the likelihood launch where an instantiation exists
    int walk_lanes = 0;             // "walk_lanes" : 0 = 8 lanes per chain up to 8 sampled dimensions, 16 beyond; 16 = always 16 (measurement)
    int walk_split = 1;             // "walk_split" / NMMA_WALK_NO_SPLIT: small queues' fused launches split by band
    int lc_group = 0;               // "lc_group"   / NMMA_LC_GROUP, NMMA_LC_NO_GROUPS: lanes per sample of em_lc_loglike (0: by batch size; 16 / 32 / 64)
    int stack2_fixup = 1;           // "stack2_fixup" / NMMA_STACK2_NO_FIXUP: re-evaluation launches of nmma_em_loglike_stack2 (0: measurement only)
    int walk_tab_mask = 0;          // fused-MCMC-step forms (bit 2 (R - 1) + (16 lanes per chain)) that give way to two launches: the log Phi table of the upper limits would cost them a ring slot
    int stack2_ok = 0;              // the handle has the one-launch form of the combined model (em_logl<.., 7>; nmma_em_loglike_stack2)
    std::string stack2_why;         // ... or why not (what nmma_last_error says after status 2)
    unsigned char* gap_ws = nullptr;    // [B] rows em_logl<.., 7> flagged for re-evaluation
    int64_t gap_cap = 0;
    int ring_max = 4;               // NMMA_EM_RING: upper bound on em_logl's LDS ring of item slots
    // nmma_lc_regrid's tables on the device, [src times | source index | n sources], one entry per DISTINCT table set seen (a
    // combined model calls with one set per regridded sub-model, the same sets every batch): keyed by content
    struct RegridSet { std::vector<unsigned char> host; unsigned char* dev = nullptr; };
    std::vector<RegridSet> regrid_sets;
    // small batches: one workgroup per (tile, observed band) instead of one per tile (see launch_logl)
    nmma::EmDev* band_dev_d = nullptr;        // [n_bands] copies of dev restricted to one band's work items
    std::vector<nmma::EmDev> band_dev;        // host mirrors (ext_tab is patched when the P92 table grows)
    int lvl_off[3] = {0, 0, 0}, lvl_n[3] = {0, 0, 0};     // band_dev holds up to three partitions: groups of 1 / 2 / 3 adjacent bands
    std::vector<int> band_k0, band_nitems;    // per band (= observed filter): first work item, number of items
    std::vector<int> grp_k0, grp_nitems;      // per group of every partition
    std::vector<const int32_t*> band_tmap[2];
    std::vector<int> band_ntasks[2];
    double* split_ws = nullptr;     // split launch: [64 KiB of per-tile arrival counters][2][bands][B] per-band sums
    int64_t split_cap = 0;
    int split_mode = -1;            // NMMA_EM_SPLIT: -1 auto, 0 never, 1 whenever the handle can
    int split_fill_wg = 256;        // NMMA_EM_SPLIT_FILL_WG: the coarsest-needed partition is the first whose tiles x groups stays within this
    int split_max_wg = 384;         // NMMA_EM_SPLIT_MAX_WG: auto mode splits while tiles x bands stays within this (measured: profiles/r03_small_batch.log)
    double* ext_ws = nullptr;       // [B][M] extinction magnitudes of the current batch (lean task with the P92 law)
    int64_t ext_cap = 0;
    int64_t lc_cap = 0;
    double* theta_stage = nullptr;
    double* out_stage = nullptr;
    double* theta_pin = nullptr;    // pinned host mirrors of the staging buffers (nmma_em_loglike_host)
    double* out_pin = nullptr;
    hipStream_t host_stream = nullptr;
    int64_t theta_cap = 0, out_cap = 0;
    int host_poll = 1;              // NMMA_EM_HOST_POLL: the zero-copy host call spins on its pinned output instead of waiting for the stream
    // launch geometry of the last call
    int g_x = 0, g_y = 0, g_block = 0, g_tile = 0, g_lds = 0;
    int force_r = 0, force_wpb = 0;
    int n_ul_items = 0;             // work items that hold upper limits
    // profiling
    bool prof_on = false;
    std::vector<hipEvent_t> ev;
    int prof_n = 0, prof_cap = 0, prof_stride = 1, prof_group = 1;
    long prof_calls = 0;
    std::vector<int> same_grid, ranges;   // per model filter (host copies used to build item descriptors)
    std::vector<double> st_host;          // host copy of EmDev::st (nmma_lc_regrid builds its bracket table against it)
    // combined model on a union grid (nmma_em_config::base_times): host copies of EmDev's b_* / u_* tables, used where the lean
    // task's rows are built; stack2_only = the handle serves nmma_em_loglike_stack2 (and the curve outputs) only -- a union grid, null filters,
    // or finite limits / time-node systematics that only the combined-model flavours carry: the entry points that take the surrogate alone refuse it
    std::vector<int32_t> u_idx, b_idx;
    std::vector<double> u_dx, u_off, b_dx, b_off;
    bool stack2_only = false;
    bool dense_plain = false;       // dense task, constant systematics, equally spaced grid: em_logl<.., 9> (that variant alone in its kernel)
    long long* dbg = nullptr;   // device buffer of in-kernel timestamps (nmma_em_debug_timeline)
    int* wd_host = nullptr;     // pinned, device-mapped watchdog words written by em_logl's hand-off waits
};

namespace nmma {

// (the NS argument of the LDS layouts: negative for unequally spaced sample_times, which keep a bracket lookup behind the grid)
static inline int lds_ns_arg(const EmDev& P) { return P.st_uniform ? P.NS : -P.NS; }

// em_logl geometry: 16-sample tiles while they fit one round of workgroups on the 256 CUs; beyond
// that 32-sample tiles (half the weight traffic and MFMA-loop overhead per sample) are faster
// (measured: B = 8192 -> 57 us vs 67 us, B = 65536 -> 428 us vs 504 us).
static int choose_r(const nmma_em_handle* h, int64_t B) {
    if (h->force_r == 1 || h->force_r == 2) return h->force_r;
    return (B + 15) / 16 > 256 ? 2 : 1;
}

// Small batches: which grouping of the observed bands (0: one band per workgroup, 1 / 2: groups of two / three) the launch of B rows
// is split into; -1: one workgroup per tile walks all bands (see launch_logl_one).
static int split_level(const nmma_em_handle* h, int R, int FAST, int64_t B, bool per_filter_parts) {
    const int n_bands = h->lvl_n[0];
    const long tiles = (long)((B + 16 * R - 1) / (16 * R));
    int lvl = -1;
    if (R == 1 && FAST != 0 && FAST != 2 && FAST != 7 && FAST != 8 && !per_filter_parts && h->band_dev_d != nullptr && n_bands >= 2 && h->split_mode != 0 &&
        tiles * (long)sizeof(unsigned) <= SPLIT_COUNTER_BYTES) {
        if (h->split_mode == 1) lvl = 0;
        else {
            for (int g = 0; g < 3 && lvl < 0; ++g)
                if (h->lvl_n[g] >= 2 && tiles * h->lvl_n[g] <= h->split_fill_wg) lvl = g;
            if (lvl < 0 && tiles * n_bands <= h->split_max_wg) lvl = 0;
        }
    }
    return lvl;
}

// em_logl's launcher; defined in em_logl.h and instantiated explicitly by the em_logl_*.hip units (em_kernels.hip only calls it)
template <int R, int KP, int NMW, int NVW, int FAST, int WALKF = 0>
hipError_t launch_logl_one(nmma_em_handle* h, const double* theta, int64_t B, int64_t ld, double* out,
                           double* chi, double* gp, hipStream_t s, const nmma_walk_fuse* wf = nullptr, uint64_t wstep = 0, int wlast = 0,
                           EmAux aux = EmAux{nullptr, nullptr, nullptr});

}  // namespace nmma
