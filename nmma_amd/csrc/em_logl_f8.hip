// em_logl_f8.hip -- instantiations of em_logl (em_logl.h): the lean task of a combined model on unequally spaced sample_times
// (FASTM 8: as FASTM 7 -- the second transient's curves as an operand, nmma_em_loglike_stack2 -- with FASTM 4's bracket search).
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_FLAVOUR(8, 8);
#endif

}  // namespace nmma
