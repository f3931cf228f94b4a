// em_logl_wc2.hip -- instantiations of em_logl (em_logl.h): the fused MCMC step WITH the chains' Constraint program (FASTM 4).
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_WALK_CON(4);
#endif

}  // namespace nmma
