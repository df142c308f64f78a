// RCCL helper of the data-parallel trainer: ONE all-reduce of the flat gradient buffer per step, broadcast of the flat
// parameter / BatchNorm buffers at start-up.  Replaces torch DDP's bucketed NCCL all-reduce + buffer broadcast
// (yogo/train.py:155-159: init_process_group("nccl") + DistributedDataParallel) with direct calls into librccl over xGMI.
//
// librccl.so is opened lazily with dlopen (the library has no load-time dependency on it: single-GPU users never touch it).
// The unique id (128 bytes, from yogo_comm_unique_id on rank 0) travels to the other ranks through whatever rendez-vous the
// host has -- yogo_amd/train.py uses the torch.distributed store.
#include "common.h"
#include <dlfcn.h>

namespace {

typedef struct { char internal[128]; } rccl_unique_id;   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm_t;
typedef int (*fn_get_unique_id)(rccl_unique_id*);
typedef int (*fn_comm_init_rank)(rccl_comm_t*, int, rccl_unique_id, int);
typedef int (*fn_comm_destroy)(rccl_comm_t);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t);
typedef int (*fn_broadcast)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_broadcast broadcast = nullptr;
  fn_error_string error_string = nullptr;
};

Rccl g_rccl;

bool rccl_load() {
  if (g_rccl.handle) return true;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    yogo_set_error("comm: librccl.so not found (%s)", dlerror());
    return false;
  }
  Rccl r;
  r.handle = h;
  r.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  r.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
  r.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  r.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
  r.broadcast = (fn_broadcast)dlsym(h, "ncclBroadcast");
  r.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.broadcast) {
    yogo_set_error("comm: librccl.so lacks an expected symbol");
    return false;
  }
  g_rccl = r;
  return true;
}

int rccl_fail(const char* what, int rc) {
  yogo_set_error("comm: %s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(rc) : "?");
  return YOGO_ERR_HIP;
}

constexpr int RCCL_SUM = 0, RCCL_UINT8 = 1, RCCL_FLOAT32 = 7;

}  // namespace

extern "C" int yogo_comm_unique_id_bytes(void) { return 128; }

// HOST buffer of yogo_comm_unique_id_bytes() bytes; call on rank 0, hand the bytes to every rank
extern "C" int yogo_comm_unique_id(void* id_out) {
  YOGO_CHECK_ARG(id_out, "comm_unique_id: null pointer");
  if (!rccl_load()) return YOGO_ERR_HIP;
  rccl_unique_id id;
  if (int rc = g_rccl.get_unique_id(&id)) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id_out, id.internal, 128);
  return YOGO_OK;
}

// collective over all ranks (one process per GPU, the calling thread's current device); *comm_out is an opaque handle
extern "C" int yogo_comm_init(int rank, int world, const void* unique_id, void** comm_out) {
  YOGO_CHECK_ARG(unique_id && comm_out && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments");
  if (!rccl_load()) return YOGO_ERR_HIP;
  rccl_unique_id id;
  memcpy(id.internal, unique_id, 128);
  rccl_comm_t c = nullptr;
  if (int rc = g_rccl.comm_init_rank(&c, world, id, rank)) return rccl_fail("ncclCommInitRank", rc);
  *comm_out = c;
  return YOGO_OK;
}

// in-place SUM of `count` floats over all ranks, enqueued on `stream` (the 1 / world scale is folded into yogo_adamw_step)
extern "C" int yogo_comm_allreduce_flat(void* comm, float* buf, size_t count, hipStream_t stream) {
  YOGO_CHECK_ARG(comm && buf, "comm_allreduce_flat: null pointer");
  if (count == 0) return YOGO_OK;
  if (!rccl_load()) return YOGO_ERR_HIP;   // (a handle from another copy of the library: never call through a null entry)
  if (int rc = g_rccl.all_reduce(buf, buf, count, RCCL_FLOAT32, RCCL_SUM, comm, stream)) return rccl_fail("ncclAllReduce", rc);
  return YOGO_OK;
}

// `bytes` bytes of rank `root` to every rank, in place, enqueued on `stream`
extern "C" int yogo_comm_broadcast_flat(void* comm, void* buf, size_t bytes, int root, hipStream_t stream) {
  YOGO_CHECK_ARG(comm && buf && root >= 0, "comm_broadcast_flat: bad arguments");
  if (bytes == 0) return YOGO_OK;
  if (!rccl_load()) return YOGO_ERR_HIP;
  if (int rc = g_rccl.broadcast(buf, buf, bytes, RCCL_UINT8, root, comm, stream)) return rccl_fail("ncclBroadcast", rc);
  return YOGO_OK;
}

extern "C" int yogo_comm_destroy(void* comm) {
  if (!comm) return YOGO_OK;
  if (!rccl_load()) return YOGO_ERR_HIP;
  if (int rc = g_rccl.comm_destroy(comm)) return rccl_fail("ncclCommDestroy", rc);
  return YOGO_OK;
}
