// rccl_native.cpp — lsa_rccl_*: the per-pass exchange step (one in-place all-reduce of 32 doubles) through RCCL called
// directly from the library, for C++ hosts that do not want to write the lsa_allreduce_fn callback themselves.
// RCCL is NOT a link-time dependency: librccl is dlopen()ed the first time one of these entry points is used (the copy
// already mapped into the process wins — under PyTorch that is torch's own librccl.so.1 — else the ROCm one), so the
// single-GPU product has no RCCL requirement at all.
#include <dlfcn.h>

#include <mutex>

#include "lsa_common.h"

namespace {
// the slice of rccl.h this file needs (ROCm 7.2 rccl/rccl.h:40-43, :187, :220, :260, :339, :448, :467)
struct NcclUniqueId { char internal[128]; };
typedef struct ncclComm *NcclComm;
typedef int NcclResult;  // ncclSuccess == 0
constexpr int kNcclSum = 0, kNcclDouble = 8;

struct Api {
  void *lib = nullptr;
  NcclResult (*GetUniqueId)(NcclUniqueId *) = nullptr;
  NcclResult (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
  NcclResult (*CommDestroy)(NcclComm) = nullptr;
  NcclResult (*AllReduce)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  NcclResult (*CommCount)(const NcclComm, int *) = nullptr;
  const char *(*GetErrorString)(NcclResult) = nullptr;
  std::string err;
};

Api &api() {
  static Api a;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
      if ((a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;  // already in the process (e.g. torch's bundled copy)
    for (const char *n : names) {
      if (a.lib) break;
      a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!a.lib) { a.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
    a.CommCount = (decltype(a.CommCount))dlsym(a.lib, "ncclCommCount");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce) a.err = "librccl lacks an expected symbol";
  });
  return a;
}

const char *nccl_err(Api &a, NcclResult r) { return a.GetErrorString ? a.GetErrorString(r) : "rccl error"; }
}  // namespace

struct lsa_rccl_comm {
  lsa_context *ctx = nullptr;
  NcclComm comm = nullptr;
  int rank = 0, world = 1;
};

extern "C" int lsa_rccl_get_unique_id(void *id128) {
  if (!id128) return LSA_ERR_INVALID;
  Api &a = api();
  if (!a.err.empty()) { lsa_set_global_error("%s", a.err.c_str()); return LSA_ERR_COLLECTIVE; }
  NcclUniqueId id;
  NcclResult r = a.GetUniqueId(&id);
  if (r != 0) { lsa_set_global_error("ncclGetUniqueId: %s", nccl_err(a, r)); return LSA_ERR_COLLECTIVE; }
  memcpy(id128, id.internal, sizeof(id.internal));
  return LSA_OK;
}

extern "C" int lsa_rccl_comm_create(lsa_rccl_comm **out, lsa_context *ctx, const void *id128, int rank, int world) {
  if (!out || !ctx || !id128 || world < 1 || rank < 0 || rank >= world) { lsa_set_global_error("lsa_rccl_comm_create: bad argument"); return LSA_ERR_INVALID; }
  *out = nullptr;
  Api &a = api();
  if (!a.err.empty()) return lsa_fail(ctx, LSA_ERR_COLLECTIVE, "%s", a.err.c_str());
  LSA_HIP(ctx, lsa_enter(ctx));
  NcclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  NcclComm comm = nullptr;
  NcclResult r = a.CommInitRank(&comm, world, id, rank);  // collective over the `world` ranks
  if (r != 0) return lsa_fail(ctx, LSA_ERR_COLLECTIVE, "ncclCommInitRank(rank %d of %d): %s", rank, world, nccl_err(a, r));
  lsa_rccl_comm *c = new lsa_rccl_comm();
  c->ctx = ctx; c->comm = comm; c->rank = rank; c->world = world;
  *out = c;
  return LSA_OK;
}

extern "C" void lsa_rccl_comm_destroy(lsa_rccl_comm *c) {
  if (!c) return;
  (void)lsa_enter(c->ctx);
  (void)hipStreamSynchronize(c->ctx->stream);
  Api &a = api();
  if (a.CommDestroy && c->comm) (void)a.CommDestroy(c->comm);
  delete c;
}

// ncclCommCount of the communicator: how many ranks the exchange step really spans (bench.py prints it on every N > 1 line)
extern "C" int lsa_rccl_comm_count(lsa_rccl_comm *c, int *count) {
  if (!c || !count) return LSA_ERR_INVALID;
  Api &a = api();
  if (!a.CommCount) { *count = c->world; return LSA_OK; }
  NcclResult r = a.CommCount(c->comm, count);
  if (r != 0) return lsa_fail(c->ctx, LSA_ERR_COLLECTIVE, "ncclCommCount: %s", nccl_err(a, r));
  return LSA_OK;
}

// the lsa_allreduce_fn the library installs for lsa_ndt_set_allreduce_rccl
static int rccl_allreduce_cb(void *user, void *device_buf, size_t n_doubles, void *hip_stream) {
  lsa_rccl_comm *c = (lsa_rccl_comm *)user;
  Api &a = api();
  NcclResult r = a.AllReduce(device_buf, device_buf, n_doubles, kNcclDouble, kNcclSum, c->comm, (hipStream_t)hip_stream);
  if (r != 0) { (void)lsa_fail(c->ctx, LSA_ERR_COLLECTIVE, "ncclAllReduce: %s", nccl_err(a, r)); return 1; }
  return 0;
}

extern "C" int lsa_ndt_set_allreduce_rccl(lsa_ndt *ndt, lsa_rccl_comm *comm) {
  if (!ndt) return LSA_ERR_INVALID;
  if (!comm) return lsa_ndt_set_allreduce(ndt, nullptr, nullptr);
  return lsa_ndt_set_allreduce(ndt, rccl_allreduce_cb, comm);
}

extern "C" int lsa_icp_set_allreduce_rccl(lsa_icp *icp, lsa_rccl_comm *comm) {
  if (!icp) return LSA_ERR_INVALID;
  if (!comm) return lsa_icp_set_allreduce(icp, nullptr, nullptr);
  return lsa_icp_set_allreduce(icp, rccl_allreduce_cb, comm);
}

// in-place sum of n doubles in device memory over the communicator, on the context's stream (no host sync): the same
// call the NDT passes make, exposed for self-tests and for reducing other per-rank blocks (ICP's 32 doubles)
extern "C" int lsa_rccl_allreduce_f64(lsa_rccl_comm *c, void *d_buf, size_t n) {
  if (!c || !d_buf) return LSA_ERR_INVALID;
  LSA_HIP(c->ctx, lsa_enter(c->ctx));
  return rccl_allreduce_cb(c, d_buf, n, (void *)c->ctx->stream) == 0 ? LSA_OK : LSA_ERR_COLLECTIVE;
}
