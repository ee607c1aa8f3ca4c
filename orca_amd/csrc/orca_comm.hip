// orca_comm.hip - the RCCL communicator of the sharded Encoder: RCCL is dlopen'ed at first use, an RCCL already in the process (PyTorch-ROCm's) is shared
// Part of liborca_hip.so (include/orca_hip.h is the ABI; orca_internal.h what the units share).
#include "orca_internal.h"

#include <dlfcn.h>
#include <link.h>

#include <mutex>

// ---------------------------------------------------------------------------
// multi-GPU exchange: RCCL, resolved at run time
// ---------------------------------------------------------------------------
struct Id128 { char internal[ORCA_COMM_ID_BYTES]; };   // = ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
namespace {
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, /* ncclUniqueId by value */ Id128, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
}  // namespace

static int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
  const char* n = info->dlpi_name;
  if (n && strstr(n, "librccl.so")) { *static_cast<std::string*>(out) = n; return 1; }
  return 0;
}

static std::string g_rccl_err;      // why RCCL could not be loaded (written once, under the call_once below)
static RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);      // PyTorch-ROCm brings its own RCCL: share it
    void* h = nullptr;
    if (!loaded.empty()) h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      const char* e = dlerror();       // once: the call clears the state
      g_rccl_err = e ? e : "librccl.so not found";
      return;
    }
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) {
      g_rccl_err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
      return;
    }
    api.lib = h;
  });
  return api.lib ? &api : nullptr;
}

struct orca_comm {
  void* comm = nullptr;   // ncclComm_t
  int nranks = 1, rank = 0, device = 0;
};

static int rccl_fail(RcclApi* r, const char* what, int rc) {
  return fail(ORCA_EHIP, "%s failed: %s (RCCL result %d)", what, r && r->GetErrorString ? r->GetErrorString(rc) : "?", rc);
}

extern "C" int orca_comm_unique_id(void* id128_host) {
  if (!id128_host) return fail(ORCA_EINVAL, "orca_comm_unique_id: NULL argument");
  RcclApi* r = rccl_api();
  if (!r) return fail(ORCA_ENODEV, "RCCL (librccl.so) could not be loaded: %s", g_rccl_err.c_str());
  Id128 id;
  const int rc = r->GetUniqueId(&id);
  if (rc != 0) return rccl_fail(r, "ncclGetUniqueId", rc);
  memcpy(id128_host, &id, sizeof id);
  return ORCA_OK;
}

extern "C" int orca_comm_init_rank(orca_ctx* ctx, int nranks, int rank, const void* id128_host, orca_comm** out) {
  if (!ctx || !id128_host || !out) return fail(ORCA_EINVAL, "orca_comm_init_rank: NULL argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(ORCA_EINVAL, "orca_comm_init_rank: rank %d of %d", rank, nranks);
  RcclApi* r = rccl_api();
  if (!r) return fail(ORCA_ENODEV, "RCCL (librccl.so) could not be loaded: %s", g_rccl_err.c_str());
  HIPCHECK(hipSetDevice(ctx->device));
  Id128 id;
  memcpy(&id, id128_host, sizeof id);
  orca_comm* c = new orca_comm();
  c->nranks = nranks; c->rank = rank; c->device = ctx->device;
  const int rc = r->CommInitRank(&c->comm, nranks, id, rank);
  if (rc != 0) { delete c; return rccl_fail(r, "ncclCommInitRank", rc); }
  *out = c;
  return ORCA_OK;
}

extern "C" int orca_comm_destroy(orca_comm* comm) {
  if (!comm) return ORCA_OK;
  RcclApi* r = rccl_api();
  (void)hipSetDevice(comm->device);
  if (r && comm->comm) (void)r->CommDestroy(comm->comm);
  delete comm;
  return ORCA_OK;
}

extern "C" int orca_allgather(orca_ctx* ctx, orca_comm* comm, const float* send, float* recv, size_t count) {
  if (!ctx || !comm || !send || !recv) return fail(ORCA_EINVAL, "orca_allgather: NULL argument");
  if (comm->device != ctx->device) return fail(ORCA_EINVAL, "orca_allgather: communicator lives on device %d, context on %d", comm->device, ctx->device);
  if (count == 0) return ORCA_OK;
  RcclApi* r = rccl_api();
  if (!r) return fail(ORCA_ENODEV, "RCCL is not loaded");
  HIPCHECK(hipSetDevice(ctx->device));
  const int rc = r->AllGather(send, recv, count, /* ncclFloat32 */ 7, comm->comm, ctx->stream);
  if (rc != 0) return rccl_fail(r, "ncclAllGather", rc);
  return ORCA_OK;
}

