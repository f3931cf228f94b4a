// em_logl_wc3.hip -- em_logl instantiations: the fused MCMC step with a Constraint program on 32-sample tiles (FASTM 1, 3)
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_WALK2_CON(1);
NMMA_LOGL_WALK2_CON(3);
#endif

}  // namespace nmma
