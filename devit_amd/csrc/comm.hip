// RCCL through the C ABI: the gradient all-reduce of the data-parallel step (distill_sub.py:333, the reducer inside
// DistributedDataParallel) for hosts that do not go through torch.distributed.  One communicator per process (one
// process per GPU); the library is bound at run time (dlopen), so libdevit_hip.so itself has no link-time dependency
// on RCCL and a process that never calls devit_comm_* never loads it -- or shares the copy PyTorch has loaded.
#include <dlfcn.h>
#include <string.h>

#include <rccl/rccl.h>

#include "devit_common.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
  ncclResult_t (*init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*destroy)(ncclComm_t) = nullptr;
  const char* (*error_string)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

bool bind_rccl() {
  if (g_rccl.lib) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* n : names)            // a copy some other library of the process (PyTorch) already loaded wins
    if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!lib)
    for (const char* n : names)
      if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!lib) return false;
  Rccl r;
  r.lib = lib;
  r.get_unique_id = (decltype(r.get_unique_id))dlsym(lib, "ncclGetUniqueId");
  r.init_rank = (decltype(r.init_rank))dlsym(lib, "ncclCommInitRank");
  r.all_reduce = (decltype(r.all_reduce))dlsym(lib, "ncclAllReduce");
  r.destroy = (decltype(r.destroy))dlsym(lib, "ncclCommDestroy");
  r.error_string = (decltype(r.error_string))dlsym(lib, "ncclGetErrorString");
  if (!r.get_unique_id || !r.init_rank || !r.all_reduce || !r.destroy) return false;
  g_rccl = r;
  return true;
}

const char* rccl_error(ncclResult_t e) { return g_rccl.error_string ? g_rccl.error_string(e) : "RCCL error"; }

}  // namespace

static_assert(sizeof(ncclUniqueId) == DEVIT_COMM_ID_BYTES, "devit_hip.h: DEVIT_COMM_ID_BYTES");

extern "C" int devit_comm_unique_id(void* id) {
  DEVIT_CHECK(id, DEVIT_ERR_ARG, "devit_comm_unique_id: null pointer");
  DEVIT_CHECK(bind_rccl(), DEVIT_ERR_DEVICE, "devit_comm_unique_id: cannot load librccl.so (%s)", dlerror());
  ncclUniqueId u;
  const ncclResult_t e = g_rccl.get_unique_id(&u);
  DEVIT_CHECK(e == ncclSuccess, DEVIT_ERR_LAUNCH, "devit_comm_unique_id: %s", rccl_error(e));
  memcpy(id, &u, sizeof(u));
  return DEVIT_OK;
}

extern "C" int devit_comm_init(const void* id, int rank, int world, void** comm) {
  DEVIT_CHECK(id && comm && world >= 1 && rank >= 0 && rank < world, DEVIT_ERR_ARG,
              "devit_comm_init: bad arguments (rank %d of %d)", rank, world);
  DEVIT_CHECK(bind_rccl(), DEVIT_ERR_DEVICE, "devit_comm_init: cannot load librccl.so (%s)", dlerror());
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t c = nullptr;
  const ncclResult_t e = g_rccl.init_rank(&c, world, u, rank);   // the calling thread's current HIP device
  DEVIT_CHECK(e == ncclSuccess, DEVIT_ERR_LAUNCH, "devit_comm_init: %s", rccl_error(e));
  *comm = (void*)c;
  return DEVIT_OK;
}

extern "C" int devit_comm_allreduce_f32(void* comm, float* buf, size_t count, void* stream) {
  DEVIT_CHECK(comm && buf && g_rccl.lib, DEVIT_ERR_ARG, "devit_comm_allreduce_f32: null pointer / no communicator");
  if (count == 0) return DEVIT_OK;
  const ncclResult_t e = g_rccl.all_reduce(buf, buf, count, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
  DEVIT_CHECK(e == ncclSuccess, DEVIT_ERR_LAUNCH, "devit_comm_allreduce_f32: %s", rccl_error(e));
  return DEVIT_OK;
}

extern "C" int devit_comm_destroy(void* comm) {
  if (!comm) return DEVIT_OK;
  DEVIT_CHECK(g_rccl.lib, DEVIT_ERR_ARG, "devit_comm_destroy: no communicator was ever created");
  const ncclResult_t e = g_rccl.destroy((ncclComm_t)comm);
  DEVIT_CHECK(e == ncclSuccess, DEVIT_ERR_LAUNCH, "devit_comm_destroy: %s", rccl_error(e));
  return DEVIT_OK;
}
