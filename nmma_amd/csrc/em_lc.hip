// em_lc.hip -- instantiations of em_lc_loglike<G, NM, SD, SA> and its launcher (em_lc.h): groups of 16 / 32 lanes per sample
// (the wave-per-sample forms: em_lc64.hip).
#define NMMA_LC_INSTANTIATE
#include "em_lc.h"

namespace nmma {

#define NMMA_LC_DEFINE(G, NM, SD, SA) template int NMMA_LC_SIGNATURE(G, NM, SD, SA);
NMMA_LC_VARIANTS_SUBWAVE(NMMA_LC_DEFINE)
#undef NMMA_LC_DEFINE

}  // namespace nmma

#ifdef NMMA_DBG_LC_STAMPS
extern "C" int32_t nmma_dbg_lc_stamps(unsigned long long* out64) {
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(nmma::g_lc_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : 1;
}
#endif
