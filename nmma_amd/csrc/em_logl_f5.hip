// em_logl_f5.hip -- instantiations of em_logl (em_logl.h): the general lean task (FASTM 5: averaged bands, time-node systematics, finite limits).
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_FLAVOUR(8, 5);
#endif

}  // namespace nmma
