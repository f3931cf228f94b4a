// em_logl_f9.hip -- instantiations of em_logl (em_logl.h): the dense task's constant-systematics, equally-spaced variant alone in its
// kernel (FASTM 9: BASELINE config 4 -- with the other variants inlined next to it that variant ran 3 % slower).
#include "em_logl.h"

namespace nmma {

#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_FLAVOUR(8, 9);
#endif

}  // namespace nmma
