// comm.hip — the path's ONLY exchange, for a C / C++ host: independent windows / loop-closure candidates are
// sharded one per GPU (one process per GPU, one lvf_ctx each) and each rank contributes fixed-size records
// (score, relative_o_c[7], candidate id — src/lvio_fusion/src/relocator.cpp:196-206) to ONE all-gather over RCCL (xGMI: pure latency).
// The reference is C++; with this entry point it shards Relocator::CorrectLoop's candidate loop without any Python in the process
// (lvio_fusion_amd/relocalize.py is the same logic over torch.distributed, used by bench.py and the gloo CPU tests).
// librccl is opened at first use (dlopen) so the hot-path library itself carries no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and enums only

#include "lvf_internal.hpp"

struct lvf_comm {
  lvf_ctx* ctx = nullptr;
  int world = 1, rank = 0;
  ncclComm_t comm = nullptr;          // null: single-rank communicator without RCCL
  lvf::DevBuf<double> send, recv;
};

namespace lvf {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};
static Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      x.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (x.h) break;
    }
    if (!x.h) { const char* e = dlerror(); x.error = std::string("librccl could not be opened: ") + (e ? e : "?"); return x; }      // (dlerror() clears its state: one call)
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.h, "ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.h, "ncclCommInitRank"));
    x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.h, "ncclAllGather"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.h, "ncclCommDestroy"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.h, "ncclGetErrorString"));
    if (!x.GetUniqueId || !x.CommInitRank || !x.AllGather || !x.CommDestroy || !x.GetErrorString) x.error = "librccl lacks a required symbol";
    return x;
  }();
  return r;
}
static int rccl_fail(ncclResult_t e, const char* what) {
  set_error("RCCL error %d (%s) in %s", (int)e, rccl().GetErrorString ? rccl().GetErrorString(e) : "?", what);
  return LVF_ERR_HIP;
}
#define LVF_RCCL(call)                                              \
  do {                                                              \
    ncclResult_t e__ = (call);                                      \
    if (e__ != ncclSuccess) return ::lvf::rccl_fail(e__, #call);    \
  } while (0)

}  // namespace lvf

using namespace lvf;

extern "C" {

int lvf_comm_get_unique_id(void* id128) {
  LVF_REQUIRE(id128, "lvf_comm_get_unique_id: null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "lvf.h documents a 128-byte id");
  if (!rccl().error.empty()) { set_error("%s", rccl().error.c_str()); return LVF_ERR_STATE; }
  ncclUniqueId id;
  LVF_RCCL(rccl().GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  return LVF_OK;
}

int lvf_comm_create(lvf_ctx* ctx, int world_size, int rank, const void* id128, lvf_comm** out) {
  LVF_REQUIRE(ctx && out, "lvf_comm_create: null argument");
  LVF_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "lvf_comm_create: rank %d / world size %d", rank, world_size);
  LVF_REQUIRE(id128 || world_size == 1, "lvf_comm_create: a %d-rank communicator needs the id of lvf_comm_get_unique_id", world_size);
  LVF_TRY(lvf::enter(ctx));
  auto* c = new lvf_comm();
  c->ctx = ctx; c->world = world_size; c->rank = rank;
  if (id128) {
    if (!rccl().error.empty()) { delete c; set_error("%s", rccl().error.c_str()); return LVF_ERR_STATE; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclResult_t e = rccl().CommInitRank(&c->comm, world_size, id, rank);
    if (e != ncclSuccess) { delete c; return rccl_fail(e, "ncclCommInitRank"); }
  }
  *out = c;
  return LVF_OK;
}

int lvf_comm_destroy(lvf_comm* c) {
  if (!c) return LVF_OK;
  if (c->comm) {
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    (void)rccl().CommDestroy(c->comm);
  }
  delete c;
  return LVF_OK;
}
int lvf_comm_world_size(const lvf_comm* c) { return c ? c->world : -1; }
int lvf_comm_rank(const lvf_comm* c) { return c ? c->rank : -1; }

int lvf_comm_allgather(lvf_comm* c, const double* send, int n_doubles, double* recv) {
  LVF_REQUIRE(c && recv, "lvf_comm_allgather: null argument");
  LVF_REQUIRE(n_doubles >= 0 && (n_doubles == 0 || send), "lvf_comm_allgather: bad send buffer");
  if (n_doubles == 0) return LVF_OK;
  if (!c->comm) { std::memcpy(recv, send, (size_t)n_doubles * 8); return LVF_OK; }     // single rank, no RCCL
  LVF_TRY(lvf::enter(c->ctx));
  hipStream_t s = c->ctx->stream;
  LVF_TRY(c->send.assign(send, (size_t)n_doubles, s));
  LVF_TRY(c->recv.ensure((size_t)n_doubles * c->world));
  LVF_RCCL(rccl().AllGather(c->send.p, c->recv.p, (size_t)n_doubles, ncclFloat64, c->comm, s));
  LVF_HIP(hipMemcpyAsync(recv, c->recv.p, (size_t)n_doubles * c->world * 8, hipMemcpyDeviceToHost, s));
  LVF_HIP(hipStreamSynchronize(s));
  return LVF_OK;
}

}  // extern "C"
