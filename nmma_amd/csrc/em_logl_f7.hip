// em_logl_f7.hip -- instantiations of em_logl (em_logl.h): the lean task of a combined model (FASTM 7: the second transient's curves
// as an operand, flux sum on the two bracket nodes of every datum -- nmma_em_loglike_stack2).
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_FLAVOUR(8, 7);
#endif

}  // namespace nmma
