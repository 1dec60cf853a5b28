// icp.hip — lsa_icp_* (placeholder until the ICP plugin lands in this round).
#include "lsa_common.h"

struct lsa_icp { lsa_context *ctx; };

extern "C" int lsa_icp_create(lsa_icp **out, lsa_context *ctx, float, int, int) {
  if (out) *out = nullptr;
  return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_icp_create: not implemented yet");
}
extern "C" void lsa_icp_destroy(lsa_icp *) {}
extern "C" int lsa_icp_set_input_target(lsa_icp *, const void *, size_t, size_t) { return LSA_ERR_INVALID; }
extern "C" int lsa_icp_set_input_target_device(lsa_icp *, const float *, size_t) { return LSA_ERR_INVALID; }
extern "C" int lsa_icp_scan_match(lsa_icp *, const void *, size_t, size_t, const float *, float *) { return LSA_ERR_INVALID; }
extern "C" int lsa_icp_scan_match_device(lsa_icp *, const float *, size_t, const float *, float *) { return LSA_ERR_INVALID; }
extern "C" int lsa_icp_get_fitness_score(lsa_icp *, float *) { return LSA_ERR_INVALID; }
