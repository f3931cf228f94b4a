// em_logl_wc4.hip -- em_logl instantiations: the fused MCMC step with a Constraint program on 32-sample tiles (FASTM 4)
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_WALK2_CON(4);
#endif

}  // namespace nmma
