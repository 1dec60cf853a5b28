// ndt_fitness.hip — GetFitnessScore (placeholder until the octree-equivalent search lands in this round).
#include "ndt_internal.h"

int ndt_fitness(lsa_ndt *h, float *score) {
  (void)score;
  return lsa_fail(h->ctx, LSA_ERR_INVALID, "GetFitnessScore: not implemented yet");
}
