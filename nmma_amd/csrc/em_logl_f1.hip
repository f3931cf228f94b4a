// em_logl_f1.hip -- instantiations of em_logl (em_logl.h): the plain lean task (FASTM 1: BASELINE config 2).
#include "em_logl.h"

namespace nmma {

#ifdef NMMA_DEV_HEADLINE_ONLY      // development builds (tools/build_variant.sh): one flavour at 16-sample tiles, NP <= 4
#ifndef NMMA_DEV_FASTM
#define NMMA_DEV_FASTM 1
#endif
NMMA_LOGL_INSTANCE(1, 1, NMMA_DEV_FASTM ? 8 : 4, NMMA_DEV_FASTM, 0);
#if NMMA_DEV_FASTM == 1
NMMA_LOGL_INSTANCE(1, 1, 8, 1, 8);
#endif
#else
NMMA_LOGL_FLAVOUR(8, 1);
#endif

}  // namespace nmma
