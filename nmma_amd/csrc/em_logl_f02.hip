// em_logl_f02.hip -- instantiations of em_logl (em_logl.h): the generic item phase (FASTM 0, 4 MFMA-role waves) and the extended task (FASTM 2).
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_FLAVOUR(4, 0);
NMMA_LOGL_FLAVOUR(8, 2);
#endif

}  // namespace nmma
