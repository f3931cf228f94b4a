// em_logl_w3.hip -- instantiations of em_logl (em_logl.h): the fused MCMC step of the general lean task (FASTM 5).
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_WALK(5);
#endif

}  // namespace nmma
