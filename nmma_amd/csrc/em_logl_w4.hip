// em_logl with the MCMC step fused in, 32-sample tiles (queues beyond 4096 chains): 16 lanes per chain for the plain lean task and the lean
// tasks with extras (instantiation set: nmma_em_loglike_walk)
#include "em_logl.h"
namespace nmma {
#ifndef NMMA_DEV_HEADLINE_ONLY
NMMA_LOGL_WALK2(1);
NMMA_LOGL_WALK2(3);
NMMA_LOGL_WALK2(4);
#endif
}  // namespace nmma
