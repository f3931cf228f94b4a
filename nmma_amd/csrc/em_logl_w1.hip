// em_logl_w1.hip -- instantiations of em_logl (em_logl.h): the fused MCMC step (8 / 16 lanes per chain) of the plain lean task and of the lean task with extras on equally spaced sample_times (FASTM 1, 3).
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_WALK(1);
NMMA_LOGL_WALK(3);
#endif

}  // namespace nmma
