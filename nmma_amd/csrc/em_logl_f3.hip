// em_logl_f3.hip -- instantiations of em_logl (em_logl.h): the lean task with extras on equally spaced sample_times (FASTM 3).
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_FLAVOUR(8, 3);
#endif

}  // namespace nmma
