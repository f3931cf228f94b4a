// em_logl_f6.hip -- instantiations of em_logl (em_logl.h): the dense lean task (FASTM 6: BASELINE config 4).
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_FLAVOUR(8, 6);
#endif

}  // namespace nmma
