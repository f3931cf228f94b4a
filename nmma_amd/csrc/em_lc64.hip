// em_lc64.hip -- instantiations of em_lc_loglike<64, NM, SD, SA> and its launcher (em_lc.h): a wave per sample, with and without the
// photometry / the curves staged in LDS (the sub-wave groups: em_lc.hip).
#define NMMA_LC_INSTANTIATE
#include "em_lc.h"

namespace nmma {

#define NMMA_LC_DEFINE(G, NM, SD, SA) template int NMMA_LC_SIGNATURE(G, NM, SD, SA);
NMMA_LC_VARIANTS_WAVE(NMMA_LC_DEFINE)
#undef NMMA_LC_DEFINE

}  // namespace nmma
