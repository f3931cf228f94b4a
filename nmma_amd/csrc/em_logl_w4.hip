// em_logl_w4.hip -- em_logl instantiations: the fused MCMC step on 32-sample tiles (plain lean task, lean task with extras on equally spaced grids)
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_WALK2(1);
NMMA_LOGL_WALK2(3);
#endif

}  // namespace nmma
