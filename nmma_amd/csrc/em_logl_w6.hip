// em_logl_w6.hip -- em_logl instantiations: the fused MCMC step on 32-sample tiles (dense lean task)
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_WALK2(6);
#endif

}  // namespace nmma
