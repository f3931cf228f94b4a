/*
 * nmma_hip.h -- C ABI of libnmma_hip.so: the MI355X (gfx950) implementation of NMMA's
 * per-sample electromagnetic light-curve log-likelihood, evaluated for a whole batch
 * of parameter vectors per call.
 *
 * The reference (nuclear-multimessenger-astronomy/nmma v1.0.1) is pure Python and has no
 * FFI for this path; the boundary it exposes is the pair of Python protocols
 *   - bilby Likelihood:  nmma/core/base.py:77-82, :133-185  (log_likelihood(parameters))
 *   - light-curve model: nmma/em/model.py:175-408, :535-731 (gen_detector_lc(parameters))
 * which nmma_amd/em/ (Python) re-implement on top of the entry points below through ctypes
 * (see INTEGRATION.md for the binding a maintainer would add to the reference).
 *
 * Conventions: every function returns 0 on success, non-zero on failure (message via
 * nmma_last_error(), thread-local); nothing throws across the boundary; the library
 * copies all configuration arrays during nmma_em_create (caller keeps ownership of its
 * buffers); theta / out buffers are owned by the caller; one handle per (process, device);
 * calls on one handle must be serialised by the caller (or issued on one stream).
 */
#ifndef NMMA_HIP_H
#define NMMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMMA_ABI_VERSION 8
#define NMMA_MAX_PARAMS 8      /* surrogate inputs NP (Bu2023Ye: 7; nmma/em/model.py:29-125) */
#define NMMA_MAX_COEFF 16      /* SVD coefficients NC (reference default 10; em_parsing.py:189) */
#define NMMA_MAX_SOURCES 3     /* model bands averaged into one observed band (utils.py:549-563) */

/* The value np.nan_to_num(-np.inf): nmma/core/base.py:82, :181; em_likelihood.py:194,:220,:348 */
#define NMMA_LOGL_FLOOR (-1.7976931348623157e308)

/* How one scalar the path needs is obtained from a row of theta[B, D]. */
enum nmma_slot_op {
    NMMA_OP_IDENT = 0,        /* v = theta[col]                                              */
    NMMA_OP_RAD2DEG = 1,      /* v = theta[col] * 180.0 / pi   (KNtheta <- inclination_EM;
                                 nmma/core/conversion.py:123)                                */
    NMMA_OP_DEG2RAD = 2,      /* v = theta[col] / 180.0 * pi   (conversion.py:125)           */
    NMMA_OP_LOG10 = 3,        /* v = log10(theta[col])         (model.py:279-280)            */
    NMMA_OP_POW10 = 4,        /* v = 10 ** theta[col]          (model.py:281-282)            */
    NMMA_OP_THETAJN2DEG = 5,  /* v = min(t, pi - t) * 180/pi   (conversion.py:120-123)       */
    NMMA_OP_COSTHETAJN2DEG = 6, /* t = arccos(theta[col]) then as 5                          */
    NMMA_OP_ACOS = 7          /* v = arccos(theta[col])        (a sampled cos_theta_jn, GW leg)  */
};

typedef struct nmma_slot {
    int32_t col;     /* column of theta, or -1: use `value` (fixed / default parameter) */
    int32_t op;      /* enum nmma_slot_op                                                */
    double value;    /* constant used when col < 0                                       */
} nmma_slot;

enum nmma_redshift_mode {
    NMMA_Z_ZERO = 0,      /* no distance information: z = 0 (conversion.py:62-64)            */
    NMMA_Z_SLOT = 1,      /* sampled "redshift" (conversion.py:58-59)                         */
    NMMA_Z_GRID = 2       /* z = interp(d_L; dist_grid, z_grid)  (model.py:262-265)           */
};

enum nmma_extinction_law {
    NMMA_EXT_LINEAR = 0,
    NMMA_EXT_P92_SMC_HOST = 1
};

enum nmma_model_kind {
    NMMA_MODEL_SVD = 0,       /* SVDLightCurveModel: surrogate tensors required (model.py:535-731)   */
    NMMA_MODEL_ME2017 = 1,    /* SimpleKilonovaLightCurveModel("Me2017"): analytic, needs filter_nu0
                                 (model.py:1280-1337, lightcurve_generation.py:566-652)              */
    NMMA_MODEL_EXTERNAL = 2   /* light curves supplied per call (nmma_em_loglike_lc): GRB afterglow,
                                 combined models (model.py:1342-1510)                               */
};

enum nmma_sys_kind {
    NMMA_SYS_CONST = 0,   /* FilterSystematicsHandler.from_budget (systematics.py:51,:203-210) */
    NMMA_SYS_PARAM = 1,   /* from_param / from_single_params      (systematics.py:279-286)     */
    NMMA_SYS_NODES = 2    /* from_interpolated_params: K time nodes, constant extrapolation
                             (systematics.py:288-291 -> utils.py:665-668)                      */
};

/* Everything static about one likelihood: surrogate tensors, time grids, photometry,
 * systematics layout, and how theta columns map to physical parameters.
 * All pointers are HOST pointers; arrays are dense, row-major, in the shapes given. */
typedef struct nmma_em_config {
    int32_t abi_version;          /* NMMA_ABI_VERSION */
    int32_t device;               /* HIP device ordinal */

    int32_t model_kind;           /* enum nmma_model_kind; the surrogate tensors below are used by
                                     NMMA_MODEL_SVD only (pass NULL / 0 otherwise, with n_tt = 0)    */
    const double* filter_nu0;     /* [M] observer-frame filter frequencies in Hz (NMMA_MODEL_ME2017;
                                     c / lambda, model.py:226) or NULL                              */

    /* ---- SVD surrogate: eval_svd_model, nmma/em/lightcurve_generation.py:180-217 ---- */
    int32_t n_model_filters;      /* M  */
    int32_t n_params;             /* NP (<= NMMA_MAX_PARAMS) */
    int32_t n_hidden;             /* NH (any value; padded internally) */
    int32_t n_coeff;              /* NC (<= NMMA_MAX_COEFF) */
    int32_t n_tt;                 /* NT: length of the SVD time grid */
    const float* W1;              /* [M][NP][NH]  first Dense kernel (training.py:353-364) */
    const float* b1;              /* [M][NH]                                               */
    const float* W2;              /* [M][NH][NC]  second Dense kernel                      */
    const float* b2;              /* [M][NC]                                               */
    const double* VA;             /* [M][NT][NC]  svd_model["VA"][:, :NC]                  */
    const double* mins;           /* [M][NT]                                               */
    const double* maxs;           /* [M][NT]                                               */
    const double* tt;             /* [M][NT]      svd_model["tt"]                          */
    const double* param_mins;     /* [M][NP]                                               */
    const double* param_maxs;     /* [M][NP]                                               */

    /* ---- model sample times: model.py:655-660, utils.py:72-93 ---- */
    int32_t n_sample_times;       /* NS; 0 => use tt of model filter 0 */
    const double* sample_times;   /* [NS] strictly increasing */

    /* ---- z(d_L) grid: conversion.py:49-55, model.py:255-267 ---- */
    int32_t redshift_mode;        /* enum nmma_redshift_mode */
    int32_t n_cosmo;              /* grid length (50 in the reference) */
    const double* dist_grid;      /* [n_cosmo] increasing, Mpc */
    const double* z_grid;         /* [n_cosmo] */

    /* ---- theta layout: em_parameter_setup, model.py:288-303; combine_lc_params :701-705 ---- */
    int32_t n_dim;                          /* D: columns of theta */
    nmma_slot model_param[NMMA_MAX_PARAMS]; /* surrogate inputs in model_parameters order */
    nmma_slot luminosity_distance;          /* default value 1e-5 Mpc (model.py:291-293)  */
    nmma_slot redshift;                     /* used when redshift_mode == NMMA_Z_SLOT      */
    nmma_slot timeshift;                    /* default 0 (model.py:297)                    */
    nmma_slot ebv;                          /* default 0 (model.py:290)                    */
    /* A sampled Hubble constant (core/base.py:161-164 -> core/conversion.py:57-101: every sample gets the redshift of ITS
     * cosmology): with NMMA_Z_GRID the grid is looked up at d_L * H0 / hubble_reference -- in a flat universe d_L scales
     * as c / H0 at fixed z, so one grid tabulated for hubble_reference serves every H0 (the h^2 dependence of the radiation
     * density is ignored: <= 3e-6 relative on z below z = 0.05); the library interpolates z / d_L, which is nearly constant,
     * instead of z.  hubble_reference = 0: no such scaling. */
    nmma_slot hubble_constant;
    double hubble_reference;

    /* ---- extinction, applied when Ebv != 0 (get_extinction_mags, model.py:323-350):
     *      NMMA_EXT_LINEAR      ext_mag[m] = ebv_coeff[m] * Ebv; coefficients are an input (any law that
     *                           does not depend on the sample, e.g. the Milky-Way foreground G23_MW whose
     *                           curve lives in third-party dust_extinction); ebv_coeff NULL => Ebv ignored
     *      NMMA_EXT_P92_SMC_HOST  Pei (1992) SMC curve at the host-frame wavelength of every sample
     *                           (utils.py:373-428); needs filter_nu0, ebv_coeff is ignored ---- */
    const double* ebv_coeff;      /* [M] or NULL */
    int32_t extinction_law;       /* enum nmma_extinction_law */

    /* ---- photometry in the detector frame: utils.py:255-286 (days since trigger) ---- */
    int32_t n_obs_filters;        /* O  */
    const int32_t* data_offsets;  /* [O+1] CSR offsets into the three arrays below */
    const double* data_times;     /* [N] */
    const double* data_mags;      /* [N] */
    const double* data_sigmas;    /* [N]  +inf marks an upper limit (em_likelihood.py:227-228) */
    const double* detection_limit;/* [O]  +inf = none (em_likelihood.py:299-300)               */
    const int32_t* n_sources;     /* [O]  1, or 2..3 for averaged bands (em_likelihood.py:326-333) */
    const int32_t* sources;       /* [O][NMMA_MAX_SOURCES] model-filter indices */

    /* ---- systematics: systematics.py:194-296 ---- */
    const int32_t* sys_kind;      /* [O] enum nmma_sys_kind */
    const double* sys_const;      /* [O] value for NMMA_SYS_CONST */
    const int32_t* sys_n_nodes;   /* [O] 1 for NMMA_SYS_PARAM, K for NMMA_SYS_NODES, 0 otherwise */
    const int32_t* sys_slot_offsets; /* [O+1] CSR offsets into sys_slots / sys_node_times */
    const nmma_slot* sys_slots;   /* per node: where its sigma_sys comes from */
    const double* sys_node_times; /* per node: time of the node (ignored for NMMA_SYS_PARAM) */

    /* ---- combined models (CombinedLightCurveModelContainer, model.py:1342-1510): 1 = the likelihood's model is the flux sum of
     * this handle's surrogate and ONE more transient whose source-frame curves arrive per call on the handle's own sample_times and
     * model filters (nmma_em_loglike_stack2) -- the handle is then laid out for the one-launch form of that call; 0 otherwise ---- */
    int32_t stack_operands;
    /* ---- ... whose sub-models bring their OWN time grids (model.py:1372-1374: the combination lives on the sorted union of the
     * sub-models' model_times; every sub-model's curves are moved there by autocomplete_data(..., extrapolate=inf), :1440-1448):
     * n_base_times > 0 = the surrogate's own sample_times are base_times[n_base_times] and `sample_times` above is the
     * COMBINATION's grid.  A surrogate node on the combination's grid is np.interp between two of its own sample nodes, each of
     * which is np.interp between two nodes of the SVD grid (lightcurve_generation.py:177) -- two static linear maps of the
     * coefficients, folded into the likelihood task's basis rows at create; nodes outside the surrogate's own grid carry +inf (no
     * flux).  Such a handle serves nmma_em_loglike_stack2, nmma_em_model_lightcurves (the surrogate's curves on the combination's
     * grid), nmma_lc_regrid / nmma_lc_stack / nmma_em_loglike_lc[_sets]; the entry points that evaluate the surrogate ALONE as the
     * likelihood's model refuse it.  Needs stack_operands = 1.  0 = the surrogate lives on sample_times itself. ---- */
    int32_t n_base_times;
    const double* base_times;     /* [n_base_times] strictly increasing */
    /* ---- ... and whose surrogate has NOTHING for some of the combination's filters -- calc_svd_lc's null output "for other
     * filters, especially radio and X-ray filters when using with GRB data" (lightcurve_generation.py:168-169: +inf on every node):
     * null_filters[m] != 0 marks model filter m as such.  Pass zero weights, biases and basis for it, and the tt / param_mins /
     * param_maxs of a real filter (the record stream still walks the item; its inputs must normalise to finite numbers).  Its rows
     * give +inf on every node, so in nmma_em_loglike_stack2 the flux sum of that band is the second transient alone; the entry
     * points that take the surrogate ALONE as the likelihood's model refuse the handle (the reference floors every sample there:
     * sanity_check on the all-inf curve, em_likelihood.py:305-311).  NULL = every model filter has a surrogate.  Needs
     * stack_operands = 1. ---- */
    const int32_t* null_filters;  /* [M] or NULL */
} nmma_em_config;

typedef struct nmma_em_handle nmma_em_handle;

/* ABI / build identification. */
int32_t nmma_abi_version(void);
const char* nmma_build_info(void);
const char* nmma_last_error(void);

/* Replaces object construction: SVDLightCurveModel.__init__ (model.py:568-653) +
 * MultiFilterTransient.__init__ (em_likelihood.py:290-303). */
int32_t nmma_em_create(const nmma_em_config* cfg, nmma_em_handle** out);
void nmma_em_destroy(nmma_em_handle* h);

/* Replaces B calls of NMMALikelihoodMixin.log_likelihood (core/base.py:77-82, :178-182)
 * -> MultiFilterTransient.log_likelihood (em_likelihood.py:186-204, :313-352).
 * theta_dev: device pointer, row-major [B][ld] doubles (ld >= D); out_dev: device [B].
 * stream: hipStream_t (NULL = default stream).  Asynchronous. */
int32_t nmma_em_loglike(nmma_em_handle* h, const double* theta_dev, int64_t B, int64_t ld,
                        double* out_dev, void* stream);

/* Same with host buffers (staged through internal pinned/device buffers; synchronous). */
int32_t nmma_em_loglike_host(nmma_em_handle* h, const double* theta_host, int64_t B, int64_t ld,
                             double* out_host);

/* Per-observed-filter pieces before the cross-filter sum (em_likelihood.py:337-352):
 * chi_dev[O][B] = sum of truncated-Gaussian terms, gp_dev[O][B] = sum of logsf terms. */
int32_t nmma_em_loglike_parts(nmma_em_handle* h, const double* theta_dev, int64_t B, int64_t ld,
                              double* chi_dev, double* gp_dev, void* stream);

/* Replaces B calls of gen_detector_lc (model.py:352-404):
 * obs_times_dev[B][NS], mag_dev[B][M][NS] apparent magnitudes (+inf outside the SVD grid). */
int32_t nmma_em_lightcurves(nmma_em_handle* h, const double* theta_dev, int64_t B, int64_t ld,
                            double* obs_times_dev, double* mag_dev, void* stream);

/* Source-frame absolute-magnitude light curves lc_dev[B][M][NS] of the handle's own model
 * (generate_lightcurve: model.py:707-728 for SVD, :1321-1337 for Me2017). */
int32_t nmma_em_model_lightcurves(nmma_em_handle* h, const double* theta_dev, int64_t B, int64_t ld,
                                  double* lc_dev, void* stream);

/* Likelihood from SUPPLIED source-frame light curves lc_dev[B][M][NS] on the handle's
 * sample_times (+inf / NaN where a model has no value): the tail of the reference path from
 * combine_detector_data on (model.py:381-404; em_likelihood.py:305-352).  Used for models whose
 * curve comes from elsewhere (NMMA_MODEL_EXTERNAL) and for combined models after nmma_lc_stack. */
int32_t nmma_em_loglike_lc(nmma_em_handle* h, const double* theta_dev, int64_t B, int64_t ld,
                           const double* lc_dev, double* out_dev, void* stream);

/* nmma_em_loglike_lc for a COMBINED model (CombinedLightCurveModelContainer.gen_detector_lc + stack_magnitudes,
 * nmma/em/model.py:1411-1510) without materialising the stacked curves: lc_dev_sets = HOST array of n_sets (1..8) device pointers
 * to source-frame sets [B][M][NS] on the handle's grid and filters (nmma_lc_regrid's output); the flux sum of nmma_lc_stack is
 * formed node by node while a sample's curves are staged on chip.  bad_rows_dev (or NULL): [B] bytes, non-zero = a sub-model
 * delivered no light curve for this row (model.py:1423-1426) -> floor. */
int32_t nmma_em_loglike_lc_sets(nmma_em_handle* h, const double* theta_dev, int64_t B, int64_t ld, const double* const* lc_dev_sets,
                                int32_t n_sets, const uint8_t* bad_rows_dev, double* out_dev, void* stream);

/* The COMBINED model of two transients that share sample_times and filters -- what the reference's drivers build: every sub-model
 * gets filters=filters, sample_times=setup_sample_times(args), model.py:1591-1614 -- in ONE launch on an NMMA_MODEL_SVD handle created
 * with stack_operands = 1: replaces B calls of CombinedLightCurveModelContainer.gen_detector_lc + stack_magnitudes (model.py:1411-1459,
 * :1486-1510) -> MultiFilterTransient.log_likelihood (em_likelihood.py:186-204, :313-352).  lc2_dev[B][M][NS]: the second transient's
 * source-frame absolute magnitudes on the handle's sample_times and model filters (+inf / NaN where it has no value: interior gaps
 * are interpolated over its finite nodes, +inf outside them, as autocomplete_data does for the reference, utils.py:626-645).  The
 * likelihood kernel takes the curves as an operand: every datum loads the two nodes it interpolates between and the flux sum is
 * formed on those two nodes next to the kilonova's two reconstructed ones -- the kilonova's curves are never written out.  Rows that
 * meet an interior gap of lc2 (rare) are re-evaluated in the same call by the kernels behind nmma_em_model_lightcurves +
 * nmma_em_loglike_lc_sets, restricted to those rows.  bad_rows_dev (or NULL): [B] bytes, non-zero = a sub-model delivered no light
 * curve for this row (model.py:1423-1426) -> floor.  flags: NMMA_STACK2_GAP_FREE = the caller guarantees that lc2 has no non-finite
 * node strictly inside the grid (an afterglowpy curve is finite or the row fails altogether, lightcurve_generation.py:259-283; +inf /
 * NaN at the first / last node are fine) -- the re-evaluation launch is then not enqueued (4 us of 89 at BASELINE config 3's shape);
 * a row that breaks the promise is floored AND poisons the handle: every later call fails with a message saying so.
 * Returns 2 -- nothing launched, nmma_last_error() says why -- when the handle has no one-launch form (not created with
 * stack_operands = 1; a configuration that needs the general task: averaged bands (finite detection limits and time-node systematics
 * are carried by the one-launch form itself -- such a handle then serves this entry point only),
 * other than 10 coefficients; sample_times reaching beyond the surrogate's grid; more curve nodes per sample than the
 * re-evaluation kernel stages -- 4 x M x NS x 8 bytes within 159 KiB of LDS): the caller then takes nmma_em_model_lightcurves +
 * nmma_em_loglike_lc_sets.  Asynchronous. */
#define NMMA_STACK2_GAP_FREE 1
/* NMMA_STACK2_COMPLETED: lc2 has been through nmma_lc_regrid (autocomplete_data over its finite nodes already applied: interior gaps
 * filled, +inf only before its first / after its last finite node) -- a non-finite node of lc2 then means "no flux" wherever it
 * lies and does not send the row to the re-evaluation launch. */
#define NMMA_STACK2_COMPLETED 2
int32_t nmma_em_loglike_stack2(nmma_em_handle* h, const double* theta_dev, int64_t B, int64_t ld, const double* lc2_dev,
                               const uint8_t* bad_rows_dev, double* out_dev, int32_t flags, void* stream);

/* Flux addition of n_models light-curve sets lc_k[B][M][NS] given on the handle's sample_times
 * (CombinedLightCurveModelContainer.gen_detector_lc + stack_magnitudes, model.py:1440-1448,
 * :1486-1510): interior non-finite nodes of every model are interpolated between its finite
 * neighbours first (autocomplete_data with its finite mask), then
 * out = -2.5 * logsumexp_k(-0.4 ln10 * lc_k) / ln10.  lc_dev_sets is a HOST array of device pointers. */
int32_t nmma_lc_stack(nmma_em_handle* h, const double* const* lc_dev_sets, int32_t n_models, int64_t B,
                      double* out_dev, void* stream);

/* One sub-model's light curves moved onto the handle's sample_times and filter list, as the combined model does before it
 * stacks (model.py:1434-1448 + the per-filter lookup of stack_magnitudes, :1490-1503):
 *   lc_src_dev[B][n_src_filters][n_src_times] on src_times_host[n_src_times] (increasing);
 *   for output filter m: n_sources_host[m] in 0..3 source filters src_index_host[m*3 + k] of the sub-model --
 *     1 = the filter itself (or its renamed equivalent), 2-3 = arithmetic mean of helper bands (utils.average_mags),
 *     0 = the sub-model has nothing for this filter (+inf everywhere: it does not contribute to the flux sum);
 *   every source curve is interpolated with np.interp over its FINITE nodes, +inf outside them and when fewer than two
 *   are finite (autocomplete_data(..., extrapolate=np.inf), utils.py:626-645).
 * out_dev[B][M][NS] in the handle's layout.  The three small host arrays are copied per call. */
int32_t nmma_lc_regrid(nmma_em_handle* h, const double* lc_src_dev, int32_t n_src_filters, int32_t n_src_times,
                       const double* src_times_host, const int32_t* src_index_host, const int32_t* n_sources_host,
                       int64_t B, double* out_dev, void* stream);

/* GW term of the joint likelihood for strain already projected onto each detector: replaces B calls of
 * bilby.gw.likelihood.GravitationalWaveTransient.log_likelihood_ratio (no marginalisation) as wrapped by
 * nmma/gw/gw_likelihood.py:97-247 and summed by MultiMessengerLikelihood (joint/joint_likelihood.py:62-67):
 *   out_dev[b] = sum_ifo ( Re<d|h_b> - <h_b|h_b>/2 ),   <a|b> = 4/duration * sum_f conj(a_f) b_f * weight_f
 * strain_dev[B][n_ifo][n_freq] and data_dev[n_ifo][n_freq] are complex128 (re, im interleaved);
 * weight_dev[n_ifo][n_freq] = 1/S_n(f) inside the detector's frequency mask, 0 outside.
 * Stateless and asynchronous on `stream` of `device`.  Waveform generation and the detector response (lalsimulation,
 * bilby.gw.detector) stay with the caller; parity against bilby is unpinned (absent from the build image). */
int32_t nmma_gw_loglike_ratio(const double* strain_dev, const double* data_dev, const double* weight_dev, int64_t B,
                              int32_t n_ifo, int64_t n_freq, double duration, double* out_dev, int32_t device, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * GW term from PARAMETERS (BASELINE config 5): replaces B calls of
 *   GravitationalWaveTransientLikelihood.log_likelihood (nmma/gw/gw_likelihood.py:97-247) ->
 *   bilby.gw.likelihood.GravitationalWaveTransient.log_likelihood_ratio [+ noise_log_likelihood] (:185-203), i.e.
 *   waveform_generator.frequency_domain_strain (bilby.gw.source.lal_binary_neutron_star, approximant
 *   IMRPhenomD_NRTidalv2 or IMRPhenomD, aligned spins) -> Interferometer.get_detector_response (antenna pattern, arrival-time
 *   shift) -> noise-weighted inner products -> sum over detectors, optionally marginalised over the coalescence phase (:164,
 *   :174-178: phase_marginalization; time and distance marginalisation are refused).
 * The strain is never materialised: one fused kernel evaluates the waveform per (frequency bin, sample) and reduces it.
 * bilby / lalsimulation are absent from the build image: PARITY UNPINNED (oracle/gw_waveform_oracle.py restates the
 * published algorithms). */
#define NMMA_GW_MAX_IFO 4

enum nmma_gw_mass_mode {
    NMMA_GW_CHIRP_MASS_RATIO = 0,   /* mass_a = chirp_mass, mass_b = mass_ratio (bilby: convert_to_lal_binary_neutron_star_parameters) */
    NMMA_GW_COMPONENT_MASSES = 1    /* mass_a = mass_1, mass_b = mass_2 */
};

typedef struct nmma_gw_config {
    int32_t abi_version;          /* NMMA_ABI_VERSION */
    int32_t device;
    int32_t n_ifo;                /* <= NMMA_GW_MAX_IFO */
    int32_t tidal;                /* 1: IMRPhenomD_NRTidalv2, 0: IMRPhenomD */
    int64_t n_freq;               /* bins of the one-sided frequency array f_k = k / duration, k = 0 .. n_freq-1 */
    double duration;              /* strain_data.duration [s] */
    double start_time;            /* strain_data.start_time [GPS s] */
    const double* data;           /* [n_ifo][n_freq][2] frequency_domain_strain (re, im) */
    const double* psd;            /* [n_ifo][n_freq]    power_spectral_density_array */
    const uint8_t* mask;          /* [n_ifo][n_freq]    frequency_mask */
    const double* detector_tensor;/* [n_ifo][9] */
    const double* vertex;         /* [n_ifo][3] metres, geocentric */
    double gmst_ref_time;         /* GMST is linearised about this GPS time: gmst(t) = gmst_ref + gmst_rate (t - ref) */
    double gmst_ref;
    double gmst_rate;
    double reference_frequency;   /* waveform_arguments["reference_frequency"] */
    double waveform_minimum_frequency;   /* waveform_arguments["minimum_frequency"]: zero strain below */
    double waveform_maximum_frequency;   /* ... ["maximum_frequency"] (+inf if absent) */
    int32_t phase_marginalization;
    int32_t mass_mode;            /* enum nmma_gw_mass_mode */
    int32_t n_dim;                /* columns of theta */
    nmma_slot mass_a, mass_b, chi_1, chi_2, lambda_1, lambda_2, luminosity_distance, theta_jn, phase, ra, dec, psi, geocent_time;
    /* Distance marginalisation (bilby GravitationalWaveTransient(distance_marginalization=True), gw_likelihood.py:174-178):
     * n_distance > 0 replaces the likelihood ratio by log sum_j w_j exp(x(d_j)) over the grid, with
     * x(d) = Re<d|h>(d) - <h|h>(d)/2 (ln I0(|<d|h>(d)|) - <h|h>(d)/2 with phase marginalisation), <d|h> ~ 1/d, <h|h> ~ 1/d^2 rescaled
     * from the row's own luminosity_distance (fix it at any value inside the prior, as bilby does with its reference distance), and
     * distance_log_weight[j] = ln(prior(d_j) * delta_d) (-inf outside the prior's support).  bilby tabulates the same sum over
     * (<d|h>, <h|h>) and interpolates; the device evaluates it per row. */
    int32_t n_distance;
    int32_t pad_distance;
    const double* distance_grid;        /* [n_distance] Mpc */
    const double* distance_log_weight;  /* [n_distance] */
    /* Time marginalisation (bilby GravitationalWaveTransient(time_marginalization=True), gw_likelihood.py:174-178):
     * time_log_weight != NULL replaces the likelihood ratio by log sum_j w_j exp(x_j) over the n_freq - 1 coalescence times
     * t_j = start_time + j * duration / (n_freq - 1), x_j = Re F_j - <h|h>/2 (ln I0(|F_j|) - <h|h>/2 with phase marginalisation),
     * F = 4/T * FFT_k( sum_ifo conj(d_k) h_k / S_k ), k = 0 .. n_freq - 2 (bilby/gw/likelihood/base.py: calculate_snrs,
     * time_marginalized_likelihood), with the row's geocent_time fixed at start_time as bilby does, and
     * time_log_weight[j] = ln(prior(t_j) * delta_t) (-inf outside the prior's support).  With n_distance > 0 as well, x_j is the
     * distance-marginalised value at F_j (bilby's order: the distance sum inside the time sum). */
    const double* time_log_weight;      /* [n_freq - 1] or NULL */
    /* bilby's jitter_time: a sampled offset time_jitter in [-delta_t/2, delta_t/2] is added to the waveform's geocent_time and to
     * the times the prior is evaluated at.  On the device the weight of shift j becomes that of the nearest node with a finite
     * weight if time_prior_minimum <= t_j + jitter <= time_prior_maximum and zero otherwise (exact for a uniform time prior).
     * col < 0 and value 0: no jitter. */
    nmma_slot time_jitter;
    double time_prior_minimum, time_prior_maximum;
} nmma_gw_config;

typedef struct nmma_gw_handle nmma_gw_handle;

int32_t nmma_gw_create(const nmma_gw_config* cfg, nmma_gw_handle** out);
void nmma_gw_destroy(nmma_gw_handle* h);

/* out_dev[b] = log-likelihood RATIO of theta row b (add nmma_gw_noise_log_likelihood for log L); rows with non-finite or
 * unphysical parameters get NMMA_LOGL_FLOOR.  Asynchronous on `stream`. */
int32_t nmma_gw_loglike(nmma_gw_handle* h, const double* theta_dev, int64_t B, int64_t ld, double* out_dev, void* stream);
double nmma_gw_noise_log_likelihood(const nmma_gw_handle* h);

/* The projected strain itself, strain_dev[B][n_ifo][n_freq][2] (zero outside the evaluated band): for parity tests and
 * plots, not on the sampler's path. */
int32_t nmma_gw_strain(nmma_gw_handle* h, const double* theta_dev, int64_t B, int64_t ld, double* strain_dev, void* stream);

/* <d|h> (re, im) and <h|h> per row: parts_dev[B][3], before the marginalisation / combination step. */
int32_t nmma_gw_inner_products(nmma_gw_handle* h, const double* theta_dev, int64_t B, int64_t ld, double* parts_dev, void* stream);

/* Introspection: bins inside the evaluated band, and HIP-event timing of the fused kernel on the launch stream. */
int64_t nmma_gw_n_bins(const nmma_gw_handle* h);
int32_t nmma_gw_profile_begin(nmma_gw_handle* h, int32_t max_launches);
int32_t nmma_gw_profile_end(nmma_gw_handle* h, double* kernel_ms_total, int32_t* n_launches);

/* ---- Lock-step ensemble walk on the device (the sampler-side batching seam; the reference builds dynesty's walker objects at
 * core/mpi_setup.py:202-245 and evolves one chain per MPI task).  One MCMC step of n chains = nmma_walk_propose -> a likelihood
 * launch on theta_dev -> nmma_walk_accept, all asynchronous on `stream`.  Random numbers: draw k of step s of the chain with key K
 * is the SplitMix64 counter hash of (K, s, k) (nmma_amd/sampler.py:counter_uniforms computes the same numbers on the host). */
#define NMMA_WALK_MAX_DIM 32
enum nmma_prior_kind {        /* bilby/core/prior/analytical.py, by the formula of ``rescale`` */
    NMMA_PRIOR_UNIFORM = 0,   /* a + u (b - a)                                  a = minimum, b = maximum */
    NMMA_PRIOR_SINE = 1,      /* arccos(cos a - u (cos a - cos b))                                        */
    NMMA_PRIOR_COSINE = 2,    /* arcsin(u (sin b - sin a) + sin a)                                        */
    NMMA_PRIOR_POWERLAW = 3,  /* (a^(1+alpha) + u (b^(1+alpha) - a^(1+alpha)))^(1/(1+alpha)); alpha = -1: a exp(u ln(b/a)) (LogUniform) */
    NMMA_PRIOR_GAUSSIAN = 4,  /* a + erfinv(2u - 1) sqrt(2) b                   a = mu, b = sigma         */
    NMMA_PRIOR_DELTA = 5,     /* a                                              a = peak                  */
    NMMA_PRIOR_TRUNC_GAUSSIAN = 6, /* a + sqrt(2) b erfinv(2 u alpha + c)       a = mu, b = sigma, alpha = normalisation =
                                 (erf((max - mu)/(sqrt(2) sigma)) - erf((min - mu)/(sqrt(2) sigma)))/2, c = erf((min - mu)/(sqrt(2) sigma))
                                 (bilby TruncatedGaussian / TruncatedNormal.rescale; priors/Sr2023.prior)  */
    NMMA_PRIOR_LOGNORMAL = 7, /* exp(a + sqrt(2 b^2) erfinv(2u - 1))            a = mu, b = sigma (LogNormal / LogGaussian) */
    NMMA_PRIOR_HALF_GAUSSIAN = 8 /* erfinv(u) sqrt(2) b                          b = sigma (HalfGaussian / HalfNormal)      */
};
enum nmma_boundary { NMMA_BOUNDARY_NONE = 0, NMMA_BOUNDARY_PERIODIC = 1, NMMA_BOUNDARY_REFLECTIVE = 2 };
typedef struct nmma_walk_prior {
    int32_t kind;      /* enum nmma_prior_kind */
    int32_t boundary;  /* enum nmma_boundary: how a proposal that leaves [0, 1] in this dimension is folded back */
    double a, b, alpha, c;
} nmma_walk_prior;

/* prop = u + gamma (live[j] - live[i]) with the boundary conditions applied; inside[c] = the proposal lies in the unit cube;
 * theta = the prior transform of prop (the chain's current theta v where the proposal fell outside).  priors: HOST array [ndim]. */
int32_t nmma_walk_propose(const nmma_walk_prior* priors, int32_t ndim, const double* live_dev, int64_t n_live, const double* u_dev,
                          const double* v_dev, const uint64_t* key_dev, int64_t n, uint64_t step, double* prop_dev, double* theta_dev,
                          int32_t* inside_dev, int32_t device, void* stream);
/* accept where inside and logl_prop > loglstar: u <- prop, v <- theta, logl <- logl_prop; counts_dev[n][4] += {accept, reject,
 * outside-the-cube, likelihood calls}.  n_steps_dev (or NULL): chain c takes part only while step <= n_steps_dev[c] (chains queued
 * with different walk lengths share the launches). */
int32_t nmma_walk_accept(int32_t ndim, int64_t n, const double* prop_dev, const double* theta_dev, const int32_t* inside_dev,
                         const double* logl_prop_dev, const double* loglstar_dev, double* u_dev, double* v_dev, double* logl_dev,
                         int32_t* counts_dev, const int32_t* n_steps_dev, uint64_t step, int32_t device, void* stream);
/* nmma_walk_accept for the walk's step number `step` (1-based) followed by nmma_walk_propose for the next one (random-number step
 * first_step + step) in ONE launch: an MCMC step of the queue is then this launch and the likelihood's. */
int32_t nmma_walk_step(const nmma_walk_prior* priors, int32_t ndim, const double* live_dev, int64_t n_live, const uint64_t* key_dev, int64_t n,
                       double* prop_dev, double* theta_dev, int32_t* inside_dev, const double* logl_prop_dev, const double* loglstar_dev,
                       double* u_dev, double* v_dev, double* logl_dev, int32_t* counts_dev, const int32_t* n_steps_dev, uint64_t step,
                       uint64_t first_step, int32_t device, void* stream);
/* The accept step of AcceptanceTrackingRWalk ("rwalk", core/mpi_setup.py:234-245): chains with active_dev[c] != 0 are accepted /
 * rejected as in nmma_walk_accept, then act_dev[c] (the autocorrelation estimate from the running acceptance ratio, bilby's
 * estimate_nmcmc with safety 1, smoothed over tau calls with old_act; old_act < 0: none) and active_dev[c] = (step < nact * act and
 * accept + reject <= maxmcmc) are updated.  Start with act = +inf, active = 1. */
int32_t nmma_walk_accept_rwalk(int32_t ndim, int64_t n, const double* prop_dev, const double* theta_dev, const int32_t* inside_dev,
                               const double* logl_prop_dev, const double* loglstar_dev, double* u_dev, double* v_dev, double* logl_dev,
                               int32_t* counts_dev, double* act_dev, int32_t* active_dev, uint64_t step, double nact, int32_t maxmcmc,
                               double tau, double old_act, int32_t device, void* stream);
/* nmma_walk_accept_rwalk for step `step` (1-based; the random-number step of its proposal) followed by nmma_walk_propose for
 * step + 1 in ONE launch. */
int32_t nmma_walk_step_rwalk(const nmma_walk_prior* priors, int32_t ndim, const double* live_dev, int64_t n_live, const uint64_t* key_dev,
                             int64_t n, double* prop_dev, double* theta_dev, int32_t* inside_dev, const double* logl_prop_dev,
                             const double* loglstar_dev, double* u_dev, double* v_dev, double* logl_dev, int32_t* counts_dev, double* act_dev,
                             int32_t* active_dev, uint64_t step, double nact, int32_t maxmcmc, double tau, double old_act, int32_t device,
                             void* stream);
/* theta = prior transform of u[n][ndim] (start points, fresh prior draws). */
int32_t nmma_walk_rescale(const nmma_walk_prior* priors, int32_t ndim, const double* u_dev, int64_t n, double* theta_dev, int32_t device,
                          void* stream);

/* ---- Constraint priors on the device (nmma/core/base.py:51-82: `evaluate_constraints` on the CONVERTED sample, the floor where the
 * product of the Constraint priors' prob is 0).  The conversion chain of a likelihood (core/conversion.py, em/model.py:272-286,
 * bilby's mass conversions) is elementwise arithmetic on the sampled columns, so the host lowers "derived quantity, then
 * minimum < value < maximum" into a small postfix program (nmma_amd/core/constraints.py traces the reference-shaped conversion
 * functions) that a kernel evaluates per row: no device-to-host copy of theta on a batched call, and the device walk can honour
 * a constrained prior set.  NaN fails every comparison (bilby's Constraint.prob does the same). */
enum nmma_con_opcode {
    NMMA_CON_PUSH_COL = 0,    /* push theta[row][col]                                   */
    NMMA_CON_PUSH_CONST = 1,  /* push value                                             */
    NMMA_CON_ADD = 2, NMMA_CON_SUB = 3, NMMA_CON_MUL = 4, NMMA_CON_DIV = 5, NMMA_CON_POW = 6,      /* (a, b) -> a op b         */
    NMMA_CON_MIN = 7, NMMA_CON_MAX = 8,
    NMMA_CON_NEG = 9, NMMA_CON_ABS = 10, NMMA_CON_SQRT = 11, NMMA_CON_LOG10 = 12, NMMA_CON_LOG = 13, NMMA_CON_EXP = 14,
    NMMA_CON_SIN = 15, NMMA_CON_COS = 16, NMMA_CON_ACOS = 17, NMMA_CON_ASIN = 18, NMMA_CON_SIGN = 19,
    NMMA_CON_CHECK_GT = 20,   /* ok &= top > value   (the value stays on the stack)     */
    NMMA_CON_CHECK_LT = 21    /* ok &= top < value   (pops)                             */
};
#define NMMA_CON_MAX_OPS 256
#define NMMA_CON_MAX_STACK 16
typedef struct nmma_con_op {
    int32_t op;      /* enum nmma_con_opcode */
    int32_t col;     /* NMMA_CON_PUSH_COL    */
    double value;    /* NMMA_CON_PUSH_CONST, NMMA_CON_CHECK_* */
} nmma_con_op;
typedef struct nmma_con_program nmma_con_program;
/* ops: HOST array, copied to `device`; the program is checked (stack depth, columns < n_cols) before anything is launched. */
int32_t nmma_con_create(const nmma_con_op* ops, int32_t n_ops, int32_t n_cols, int32_t device, nmma_con_program** out);
void nmma_con_destroy(nmma_con_program* p);
/* logl_dev[b] = NMMA_LOGL_FLOOR where row b of theta_dev[B][ld] fails a check (in place; asynchronous on `stream`). */
int32_t nmma_con_floor(const nmma_con_program* p, const double* theta_dev, int64_t B, int64_t ld, double* logl_dev, void* stream);

/* ---- One queue of the nested sampler in ONE call: `pool.map(sample, queue)` of core/mpi_setup.py:282-303, :339 for the
 * fixed-length ensemble walk ("acceptance-walk", :221-232) over an EM likelihood handle.  Host arrays in, host arrays out; inside:
 * one packed upload, prior transform of the start points, then per MCMC step the likelihood launch and accept + next proposal
 * (nmma_em_loglike, nmma_walk_step: the very launches of the per-step entry points, so the chains are bit-identical), a fresh
 * prior draw for every chain that never moved (bilby: "Unable to find a new point using walk"), one packed download.  The
 * workspace keeps its device and pinned buffers between queues. */
typedef struct nmma_walk_ws nmma_walk_ws;
int32_t nmma_walk_ws_create(int32_t device, nmma_walk_ws** out);
void nmma_walk_ws_destroy(nmma_walk_ws* ws);
/* The MCMC step as ONE launch: the likelihood kernel's workgroup owns the chains whose log L it has just summed, so the accept of
 * step `step` and the proposal of step `step + 1` (nmma_walk_step) can run in its epilogue and leave the tile's theta rows ready for
 * the next launch.  `wf_dev`: a DEVICE copy of this struct (pointers to the chains' state, as for nmma_walk_step; theta is the
 * launch's own theta_dev, which the kernel then WRITES).  last != 0: accept only (the walk's final step).  n_con_ops: the host's copy
 * of wf_dev->n_con_ops (> 0: the chains' Constraint program is evaluated in the step, on an fp64 stack in LDS).  Returns 2 --
 * nothing launched -- when the handle's task flavour has no fused instantiation (for constrained sets: the general and the dense
 * lean task), with more than 16 sampled dimensions or more than 4096 chains (the caller then takes nmma_em_loglike + nmma_walk_step);
 * the chains are bit-identical either way. */
typedef struct nmma_walk_fuse {
    nmma_walk_prior priors[NMMA_WALK_MAX_DIM];
    int32_t ndim, n_con_ops;
    const double* live; int64_t n_live;
    const uint64_t* key;
    double* prop; int32_t* inside;
    const double* loglstar;
    double* u; double* v; double* logl; int32_t* counts;
    const int32_t* n_steps;           /* or NULL */
    const nmma_con_op* con_ops;       /* device array (nmma_con_program) or NULL */
    uint64_t first_step;
} nmma_walk_fuse;
int32_t nmma_em_loglike_walk(nmma_em_handle* h, double* theta_dev, int64_t B, int64_t ld, double* out_dev, const nmma_walk_fuse* wf_dev,
                             uint64_t step, int32_t last, int32_t n_con_ops, void* stream);

typedef struct nmma_walk_queue {
    const nmma_walk_prior* priors;   /* [ndim] */
    int32_t ndim;
    int32_t walks;                   /* steps per chain when walks_per_chain is NULL */
    const double* live;              /* [n_live][ndim] unit-cube live points (>= 3)  */
    int64_t n_live;
    const double* u0;                /* [n][ndim] start points (unit cube)           */
    const double* loglstar;          /* [n] likelihood bound of every chain          */
    const uint64_t* key;             /* [n] chain keys                               */
    const int32_t* walks_per_chain;  /* [n] or NULL                                  */
    int64_t n;
    uint64_t first_step;             /* random-number step of the first proposal (1) */
    const nmma_con_program* constraints;   /* or NULL: rows failing a check count as evaluated and get NMMA_LOGL_FLOOR   */
    double* u;                       /* out [n][ndim]                                */
    double* v;                       /* out [n][ndim] theta of u                     */
    double* logl;                    /* out [n]                                      */
    int32_t* counts;                 /* out [n][4] accept, reject, outside the cube, likelihood calls                  */
    double gpu_ms;                   /* out: upload .. download complete, HIP events */
    double* records_dev;             /* NULL, or a DEVICE buffer [n][2 ndim + 3]: row c = u | v | logl | counts (the four int32 in the bit
                                      * patterns of two doubles).  The records are then packed there on `stream` INSTEAD of being downloaded
                                      * (u, v, logl, counts may be NULL; end only waits and checks the handle) -- the send buffer of a queue
                                      * sharded over ranks, whose records are all-gathered on the device (RCCL over xGMI) and downloaded
                                      * once: nmma_amd/parallel.py:ShardedQueue, the reference's chains-over-ranks split of
                                      * core/mpi_setup.py:651-667, :679-683 */
} nmma_walk_queue;
int32_t nmma_em_walk_queue(nmma_em_handle* h, nmma_walk_ws* ws, nmma_walk_queue* q, void* stream);
/* The same call in two halves, for a queue SHARDED over several devices from one host thread (the reference spreads the chains of a
 * queue over its MPI ranks, core/mpi_setup.py:651-667, :679-683; chains are independent and the live set is read-only, so the queue
 * shards like a batch -- counter-based random numbers make a chain's path independent of the shard it lands in): begin packs, uploads
 * and enqueues the whole step loop and the download on `stream` of the workspace's device and returns without waiting; end waits,
 * unpacks into q's output arrays (the same record that was begun) and checks the handle.  One queue in flight per workspace. */
int32_t nmma_em_walk_queue_begin(nmma_em_handle* h, nmma_walk_ws* ws, const nmma_walk_queue* q, void* stream);
int32_t nmma_em_walk_queue_end(nmma_em_handle* h, nmma_walk_ws* ws, nmma_walk_queue* q);

/* MultiMessengerLikelihood.sub_log_likelihood for a batch (joint/joint_likelihood.py:62-67): out_dev[b] = sum_k parts[k][b] in
 * messenger order, NMMA_LOGL_FLOOR where the sum is not finite or a messenger already returned the floor.
 * parts_dev: HOST array of n_parts (1..8) device pointers to [B] doubles. */
int32_t nmma_logl_sum_floor(const double* const* parts_dev, int32_t n_parts, int64_t B, double* out_dev, int32_t device, void* stream);

/* Surrogate output only: coeff_dev[B][M][NC] fp32 (lightcurve_generation.py:198). */
int32_t nmma_em_coefficients(nmma_em_handle* h, const double* theta_dev, int64_t B, int64_t ld,
                             float* coeff_dev, void* stream);

/* Run-time options of a handle, for A/B measurements and tests (each also has an environment variable that is read once, at
 * nmma_em_create): "walk_fuse" (NMMA_WALK_NO_FUSE), "walk_split" (NMMA_WALK_NO_SPLIT): 0 / 1; "lc_group" (NMMA_LC_GROUP): 0 = by batch
 * size, 16 / 32 / 64 lanes per sample of the likelihood-from-curves kernel; "stack2_fixup" (NMMA_STACK2_NO_FIXUP): 0 skips the
 * re-evaluation launches of nmma_em_loglike_stack2 (measurement only).  Unknown names fail. */
int32_t nmma_em_set_option(nmma_em_handle* h, const char* name, int32_t value);

/* Introspection used by bench.py / tests. */
int32_t nmma_em_n_sample_times(const nmma_em_handle* h);
int32_t nmma_em_device(const nmma_em_handle* h);            /* the HIP device the handle's tables live on (-1: null handle) */
int64_t nmma_em_flops_per_eval(const nmma_em_handle* h);   /* SURVEY.md section 8d figure */
int32_t nmma_em_last_launch_geometry(const nmma_em_handle* h, int32_t* grid_x, int32_t* grid_y,
                                     int32_t* block, int32_t* tile_samples, int32_t* lds_bytes);

/* Synchronise the handle's device and report asynchronous failures of earlier launches (the hand-off
 * watchdog of the log-likelihood kernel).  No reference counterpart: the reference is synchronous. */
int32_t nmma_em_check(nmma_em_handle* h);

/* Diagnostics: run nmma_em_loglike once and return 512 shader-clock stamps taken inside
 * workgroup 0: entries 2k, 2k+1 = MFMA role (wave 0) around the MLP of work item k; 64, 65 = prologue of
 * the likelihood role; 66+2k, 67+2k = stage Q of the first task of item k; for task t < 24 (fast mode):
 * 16+t = claim time, 40+t = claiming wave, 104+t = completion time; 96..101 = stages of the last item's
 * first task.  Synchronous. */
int32_t nmma_em_debug_timeline(nmma_em_handle* h, const double* theta_dev, int64_t B, int64_t ld,
                               double* out_dev, int64_t* stamps_host);

/* HIP-event timing of the dominant kernel on the launch stream: between begin and end
 * every nmma_em_loglike call brackets its fused per-filter kernel with two events. */
int32_t nmma_em_profile_begin(nmma_em_handle* h, int32_t max_launches);
int32_t nmma_em_profile_end(nmma_em_handle* h, double* fused_ms_total, double* combine_ms_total,
                            int32_t* n_launches);

#ifdef __cplusplus
}
#endif
#endif /* NMMA_HIP_H */
