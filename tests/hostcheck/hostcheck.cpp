// TEST INFRASTRUCTURE: host build (g++) of nmma_amd/csrc/em_math.h -- the very source the
// HIP kernels inline -- so its branchy fp64 numerics can be checked on a CPU against
// numpy / scipy.  Never loaded by the product path.
#include "../../nmma_amd/csrc/em_math.h"

extern "C" {
double hc_interp_np(double x, const double* xp, const double* fp, int n, double left, double right) {
    return nmma::interp_np(x, xp, fp, n, left, right);
}
double hc_lerp_np(double x, double x0, double x1, double y0, double y1) { return nmma::lerp_np(x, x0, x1, y0, y1); }
double hc_ndtr(double x) { return nmma::ndtr(x); }
double hc_log_ndtr(double x) { return nmma::log_ndtr(x); }
double hc_log_gauss_mass_neginf(double b) { return nmma::log_gauss_mass_neginf(b); }
double hc_upper_limit_term_tab(double m, double est, double sigma_sys) { return nmma::upper_limit_term_tab(m, est, sigma_sys, nmma::kLogPhiTab); }
double hc_log_gauss_mass_tab(double b) { return nmma::log_gauss_mass_tab(b, nmma::kLogPhiTab); }
double hc_detection_term_tab(double m, double est, double sigma, double log_sigma, double lim) {
    return nmma::detection_term_tab(m, est, sigma, log_sigma, lim, nmma::kLogPhiTab);
}
double hc_detection_term(double m, double est, double sigma, double log_sigma, double lim) {
    return nmma::detection_term(m, est, sigma, log_sigma, lim);
}
double hc_upper_limit_term(double m, double est, double sigma_sys) { return nmma::upper_limit_term(m, est, sigma_sys); }
double hc_apply_slot(int col, int op, double value, const double* row) {
    nmma_slot s{col, op, value};
    return nmma::apply_slot(s, row);
}
double hc_distance_modulus(double d) { return nmma::distance_modulus(d); }
double hc_redshift_correction(double z) { return nmma::redshift_correction(z); }
double hc_extinction_mag(int law, double coeff, double zp1, double ebv) { return nmma::extinction_mag(law, coeff, zp1, ebv); }
}
