// RCCL data-parallel exchange at the C boundary ("tg_allreduce_*,
// communicator created from a ncclUniqueId broadcast by rank 0").
//
// Replaces, for a host that does not run torch.distributed: DistributedDataParallel's
// gradient all-reduce (codes/models/base_model.py:130-136), the SyncBatchNorm statistics
// exchange (:133) and dist.all_reduce of the adaptive-D scalars
// (codes/models/vsrgan_model.py:166-173).  One process per GPU; the unique id travels over
// whatever channel the host already has (the reference's launcher uses env:// TCP,
// codes/utils/dist_utils.py:8-24).
//
// RCCL is bound at RUN time with dlopen -- first the copy the process has already mapped
// (PyTorch ships its own librccl.so; two RCCL copies in one process would each spin up their
// own proxy threads), then the system one.  libtecogan_hip.so therefore has no link-time
// dependency on RCCL: single-GPU users never load it, and a missing RCCL is a loud TG_E_HIP
// from tg_comm_* -- never a silent single-rank fallback.
#include <dlfcn.h>

#include <new>

#include "tg_common.h"

namespace {

typedef struct { char internal[TG_COMM_ID_BYTES]; } rccl_unique_id;   // ncclUniqueId (128 bytes)
typedef void* rccl_comm_t;
enum { RCCL_SUM = 0, RCCL_FLOAT32 = 7 };                               // ncclSum, ncclFloat32

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(rccl_unique_id*) = nullptr;
  int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(rccl_comm_t, int*) = nullptr;        // optional: what the LIBRARY says the communicator spans
  int (*CommUserRank)(rccl_comm_t, int*) = nullptr;     // optional
  const char* origin = "";
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.handle ? &r : nullptr;
  tried = true;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* nm : names)   // a copy that is already mapped (PyTorch's) wins
    if (!r.handle && (r.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) r.origin = "already mapped";
  const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* nm : paths)
    if (!r.handle && (r.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) r.origin = nm;
  if (!r.handle) return nullptr;
  auto sym = [&](const char* s) { return dlsym(r.handle, s); };
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
  r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.AllGather) {
    r.handle = nullptr;
    return nullptr;
  }
  return &r;
}

int rccl_fail(const char* what, int code) {
  Rccl* r = rccl();
  tg::set_error("%s: RCCL error %d (%s)", what, code,
                (r && r->GetErrorString) ? r->GetErrorString(code) : "?");
  return TG_E_HIP;
}

}  // namespace

struct tg_comm {
  rccl_comm_t comm;
  int world, rank;
};

#define TG_NEED_RCCL(r)                                                                     \
  Rccl* r = rccl();                                                                         \
  TG_REQUIRE(r, TG_E_HIP, "RCCL not found (dlopen of librccl.so / its nccl* symbols failed); " \
                          "the multi-GPU exchange has no fallback")

extern "C" int tg_comm_get_unique_id(uint8_t id[TG_COMM_ID_BYTES]) {
  TG_REQUIRE(id, TG_E_ARG, "comm_get_unique_id: null pointer");
  TG_NEED_RCCL(r);
  rccl_unique_id u;
  int rc = r->GetUniqueId(&u);
  if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
  for (int i = 0; i < TG_COMM_ID_BYTES; ++i) id[i] = (uint8_t)u.internal[i];
  return TG_OK;
}

extern "C" int tg_comm_init_rank(const uint8_t id[TG_COMM_ID_BYTES], int world, int rank,
                                 tg_comm** out) {
  TG_REQUIRE(id && out, TG_E_ARG, "comm_init_rank: null pointer");
  TG_REQUIRE(world >= 1 && rank >= 0 && rank < world, TG_E_ARG, "comm_init_rank: rank %d of %d",
             rank, world);
  TG_NEED_RCCL(r);
  rccl_unique_id u;
  for (int i = 0; i < TG_COMM_ID_BYTES; ++i) u.internal[i] = (char)id[i];
  tg_comm* c = new (std::nothrow) tg_comm();
  TG_REQUIRE(c, TG_E_ARG, "comm_init_rank: out of host memory");
  int rc = r->CommInitRank(&c->comm, world, u, rank);   // collective: blocks until all ranks call
  if (rc != 0) { delete c; return rccl_fail("ncclCommInitRank", rc); }
  c->world = world; c->rank = rank;
  *out = c;
  return TG_OK;
}

extern "C" int tg_comm_destroy(tg_comm* c) {
  if (!c) return TG_OK;
  Rccl* r = rccl();
  int rc = r ? r->CommDestroy(c->comm) : 0;
  delete c;
  return rc == 0 ? TG_OK : rccl_fail("ncclCommDestroy", rc);
}

// What RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank) -- NOT the numbers the caller passed
// to tg_comm_init_rank: the first multi-GPU run checks that the library agrees with the launcher (bench.py: ranks_seen).
extern "C" int tg_comm_query(const tg_comm* c, int* ranks_seen, int* rank_seen) {
  TG_REQUIRE(c && ranks_seen && rank_seen, TG_E_ARG, "comm_query: null pointer");
  TG_NEED_RCCL(r);
  TG_REQUIRE(r->CommCount && r->CommUserRank, TG_E_HIP, "comm_query: this librccl has no ncclCommCount / ncclCommUserRank");
  int rc = r->CommCount(c->comm, ranks_seen);
  if (rc != 0) return rccl_fail("ncclCommCount", rc);
  rc = r->CommUserRank(c->comm, rank_seen);
  return rc == 0 ? TG_OK : rccl_fail("ncclCommUserRank", rc);
}

extern "C" int tg_comm_world(const tg_comm* c) { return c ? c->world : 0; }
extern "C" int tg_comm_rank(const tg_comm* c) { return c ? c->rank : -1; }
extern "C" const char* tg_comm_library_origin(void) { Rccl* r = rccl(); return r ? r->origin : ""; }

extern "C" int tg_allreduce_sum_f32(tg_comm* c, float* buf, int64_t count, tg_stream_t stream) {
  TG_REQUIRE(c && buf && count > 0, TG_E_ARG, "allreduce_sum_f32: bad argument");
  TG_NEED_RCCL(r);
  int rc = r->AllReduce(buf, buf, (size_t)count, RCCL_FLOAT32, RCCL_SUM, c->comm, (hipStream_t)stream);
  return rc == 0 ? TG_OK : rccl_fail("ncclAllReduce", rc);
}

extern "C" int tg_allgather_f32(tg_comm* c, const float* send, float* recv, int64_t count_per_rank,
                                tg_stream_t stream) {
  TG_REQUIRE(c && send && recv && count_per_rank > 0, TG_E_ARG, "allgather_f32: bad argument");
  TG_NEED_RCCL(r);
  int rc = r->AllGather(send, recv, (size_t)count_per_rank, RCCL_FLOAT32, c->comm, (hipStream_t)stream);
  return rc == 0 ? TG_OK : rccl_fail("ncclAllGather", rc);
}
