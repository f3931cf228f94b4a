// em_logl_f4.hip -- instantiations of em_logl (em_logl.h): the lean task with extras on unequally spaced sample_times (FASTM 4).
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_FLAVOUR(8, 4);
#endif

}  // namespace nmma
