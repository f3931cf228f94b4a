// em_logl_w2.hip -- instantiations of em_logl (em_logl.h): the fused MCMC step of the lean task on unequally spaced sample_times and of the dense lean task (FASTM 4, 6).
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_WALK(4);
NMMA_LOGL_WALK(6);
#endif

}  // namespace nmma
