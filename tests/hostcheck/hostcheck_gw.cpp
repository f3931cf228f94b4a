// TEST INFRASTRUCTURE: host build (g++) of nmma_amd/csrc/gw_math.h -- the very source the HIP kernels of the
// gravitational-wave leg inline -- so the waveform set-up, the per-bin evaluation and the detector projection can be
// checked on a CPU against oracle/gw_waveform_oracle.py.  Never loaded by the product path.
#include "../../nmma_amd/csrc/gw_math.h"

using namespace nmma::gw;

static GwParams params_from(const double* p) {
    GwParams q;
    q.mass_1 = p[0]; q.mass_2 = p[1]; q.chi_1 = p[2]; q.chi_2 = p[3]; q.lambda_1 = p[4]; q.lambda_2 = p[5];
    q.luminosity_distance = p[6]; q.theta_jn = p[7]; q.phase = p[8]; q.ra = p[9]; q.dec = p[10]; q.psi = p[11]; q.geocent_time = p[12];
    return q;
}

extern "C" {
int hc_gw_source_doubles() { return (int)(sizeof(GwSource) / sizeof(double)); }

// params[13] -> source record (as doubles); det = n_ifo x (tensor[9], vertex[3])
void hc_gw_setup(const double* params, double f_ref, int tidal, const double* det, int n_ifo, double start_time, double gmst_ref_time,
                 double gmst_ref, double gmst_rate, double* out) {
    GwSource S{};
    const GwParams q = params_from(params);
    setup_source(q, f_ref, tidal != 0, S);
    for (int i = 0; i < n_ifo && S.valid != 0.0; ++i) {
        GwDetector D;
        for (int k = 0; k < 9; ++k) D.tensor[k] = det[12 * i + k];
        for (int k = 0; k < 3; ++k) D.vertex[k] = det[12 * i + 9 + k];
        project_source(q, D, i, start_time, gmst_ref_time, gmst_ref, gmst_rate, 2.0, S);
    }
    const double* s = reinterpret_cast<const double*>(&S);
    for (int k = 0; k < hc_gw_source_doubles(); ++k) out[k] = s[k];
}

void hc_gw_eval(const double* source, const double* f, int n, double* amp, double* phase_over_pi) {
    const GwSource& S = *reinterpret_cast<const GwSource*>(source);
    for (int i = 0; i < n; ++i) eval_bin(S, make_bin(f[i]), amp[i], phase_over_pi[i]);
}

void hc_gw_projection(const double* source, int n_ifo, double* k_re, double* k_im, double* dt) {
    const GwSource& S = *reinterpret_cast<const GwSource*>(source);
    for (int i = 0; i < n_ifo; ++i) { k_re[i] = S.k_re[i]; k_im[i] = S.k_im[i]; dt[i] = S.dt[i]; }
}

double hc_gw_ln_i0(double x) { return ln_bessel_i0(x); }
}
