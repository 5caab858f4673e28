// RCCL glue: the only collective on this path is the sum-all-reduce of ONE
// float64 (the local sum of log_prob) per NLL evaluation.
// librccl is dlopen'ed on first use so that libnf_hip.so loads on boxes without it
// and a single-GPU process never touches it.
#include "nf_internal.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

namespace {

// the subset of the RCCL ABI used here (rccl.h: ncclUniqueId is 128 opaque bytes,
// ncclFloat64 = 8, ncclSum = 0, ncclSuccess = 0)
struct UniqueId { char internal[NF_COMM_ID_BYTES]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
  void* lib = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  GetErrorStringFn get_error_string = nullptr;
};

Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return NF_OK;
  // The ROCm installation's RCCL first, by path: a process that has also imported torch already
  // holds torch's bundled librccl (same soname, built against another HIP runtime), and a bare
  // soname would hand that copy back.  NF_RCCL_LIBRARY overrides.
  const char* names[] = {getenv("NF_RCCL_LIBRARY"), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
  void* lib = nullptr;
  for (const char* n : names) {
    if (!n || !*n) continue;
    lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (lib) break;
  }
  if (!lib) {
    nf::set_error("cannot dlopen librccl: %s", dlerror());
    return NF_ERR_COMM;
  }
  Rccl r;
  r.lib = lib;
  r.get_unique_id = (GetUniqueIdFn)dlsym(lib, "ncclGetUniqueId");
  r.comm_init_rank = (CommInitRankFn)dlsym(lib, "ncclCommInitRank");
  r.comm_destroy = (CommDestroyFn)dlsym(lib, "ncclCommDestroy");
  r.all_reduce = (AllReduceFn)dlsym(lib, "ncclAllReduce");
  r.get_error_string = (GetErrorStringFn)dlsym(lib, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce) {
    nf::set_error("librccl lacks an expected symbol");
    dlclose(lib);
    return NF_ERR_COMM;
  }
  g_rccl = r;
  return NF_OK;
}

int rccl_fail(int code, const char* what) {
  nf::set_error("RCCL error %d (%s) in %s", code,
                g_rccl.get_error_string ? g_rccl.get_error_string(code) : "?", what);
  return NF_ERR_COMM;
}

}  // namespace

struct nf_comm {
  Comm comm = nullptr;
  int world = 1;
  int rank = 0;
};

namespace nf {

int comm_allreduce_sum_f64(nf_comm* c, double* dev_values, size_t count, hipStream_t stream) {
  if (!c || !c->comm) { set_error("communicator is not initialised"); return NF_ERR_COMM; }
  const int rc = g_rccl.all_reduce(dev_values, dev_values, count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, c->comm, stream);
  if (rc != 0) return rccl_fail(rc, "ncclAllReduce");
  return NF_OK;
}

int comm_allreduce_sum_f32(nf_comm* c, float* dev_values, size_t count, hipStream_t stream) {
  if (!c || !c->comm) { set_error("communicator is not initialised"); return NF_ERR_COMM; }
  const int rc = g_rccl.all_reduce(dev_values, dev_values, count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, stream);
  if (rc != 0) return rccl_fail(rc, "ncclAllReduce");
  return NF_OK;
}

}  // namespace nf

extern "C" {

int nf_comm_preflight(void) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipFree(nullptr);  // forces the context: a device this process cannot use fails here
  if (e != hipSuccess) {
    nf::set_error("no usable HIP device before communicator set-up: %s", hipGetErrorString(e));
    return NF_ERR_NO_DEVICE;
  }
  return load_rccl();
}

int nf_comm_unique_id(unsigned char id[NF_COMM_ID_BYTES]) {
  if (!id) { nf::set_error("id is NULL"); return NF_ERR_INVALID; }
  NF_TRY(load_rccl());
  UniqueId u;
  const int rc = g_rccl.get_unique_id(&u);
  if (rc != 0) return rccl_fail(rc, "ncclGetUniqueId");
  memcpy(id, u.internal, NF_COMM_ID_BYTES);
  return NF_OK;
}

int nf_comm_init_rank(nf_comm** out_comm, int world_size, int rank,
                      const unsigned char id[NF_COMM_ID_BYTES]) {
  if (!out_comm || !id) { nf::set_error("bad argument"); return NF_ERR_INVALID; }
  *out_comm = nullptr;
  if (world_size < 1 || rank < 0 || rank >= world_size) {
    nf::set_error("rank %d outside world of %d", rank, world_size);
    return NF_ERR_INVALID;
  }
  NF_TRY(load_rccl());
  UniqueId u;
  memcpy(u.internal, id, NF_COMM_ID_BYTES);
  nf_comm* c = new nf_comm();
  c->world = world_size;
  c->rank = rank;
  const int rc = g_rccl.comm_init_rank(&c->comm, world_size, u, rank);
  if (rc != 0) { delete c; return rccl_fail(rc, "ncclCommInitRank"); }
  *out_comm = c;
  return NF_OK;
}

int nf_comm_destroy(nf_comm* c) {
  if (!c) return NF_OK;
  int rc = 0;
  if (c->comm) rc = g_rccl.comm_destroy(c->comm);
  delete c;
  if (rc != 0) return rccl_fail(rc, "ncclCommDestroy");
  return NF_OK;
}

}  // extern "C"
