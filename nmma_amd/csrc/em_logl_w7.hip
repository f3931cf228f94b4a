// em_logl_w7.hip -- em_logl instantiations: the fused MCMC step on 32-sample tiles (lean task with extras on unequally spaced grids)
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_WALK2(4);
#endif

}  // namespace nmma
