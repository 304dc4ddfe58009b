// mia_comm.hip -- the transports behind a sharded mia_hip_iterate (include/mia_hip.h "several GPUs").
//
// mia_hip_iterate speaks to the other ranks through a table of two collectives (mia_hip_collectives: all-gather of
// bytes, in-place all-reduce of int32 by sum or maximum), nothing else.  Two tables are made here:
//   * RCCL over xGMI (mia_comm_rccl_table): librccl is opened on first use and looked up by name, so libmia_hip.so has
//     no link-time dependency on it and a single-GPU run never loads it;
//   * an in-process loopback (mia_hip_loopback_*): W contexts of ONE process, one host thread each, on one GPU (or on
//     GPUs with peer access) -- the ranks meet at a host barrier and copy each other's device buffers.  It is how the
//     sharded path is tested on a box with a single GPU (RCCL refuses two ranks on one device), and how `mia_hip -g`
//     runs its threaded driver there.
// The reference has no counterpart: src/ is single-threaded.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only
#include <dlfcn.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/mia_hip.h"
#include "mia_comm.h"

namespace {

// ---- RCCL ----------------------------------------------------------------------------------------------------------
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  RcclApi() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) { lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) return;
    GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    CommAbort = (decltype(CommAbort))dlsym(lib, "ncclCommAbort");
    CommCount = (decltype(CommCount))dlsym(lib, "ncclCommCount");
    CommUserRank = (decltype(CommUserRank))dlsym(lib, "ncclCommUserRank");
    AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    ok = GetUniqueId && CommInitRank && CommDestroy && AllReduce && AllGather;
  }
};
// a function-local static: initialised once, thread-safe (several host threads call mia_hip_comm_init at the same moment)
RcclApi* rccl_api() {
  static RcclApi api;
  return api.ok ? &api : nullptr;
}

struct RcclRank {
  ncclComm_t comm = nullptr;
  RcclApi* api = nullptr;
  std::string err;
};
int rccl_fail(RcclRank* r, const char* what, ncclResult_t e) {
  r->err = std::string(what) + ": " + (r->api->GetErrorString ? r->api->GetErrorString(e) : "RCCL error");
  return MIA_HIP_ERR_DEVICE;
}
int rccl_all_gather(void* user, const void* send, void* recv, size_t bytes, void* stream) {
  RcclRank* r = static_cast<RcclRank*>(user);
  const ncclResult_t e = r->api->AllGather(send, recv, bytes, ncclInt8, r->comm, static_cast<hipStream_t>(stream));
  return e == ncclSuccess ? MIA_HIP_OK : rccl_fail(r, "ncclAllGather", e);
}
int rccl_all_reduce(void* user, void* buf, size_t count, int op, void* stream) {
  RcclRank* r = static_cast<RcclRank*>(user);
  const ncclResult_t e = r->api->AllReduce(buf, buf, count, ncclInt32, op == MIA_HIP_OP_MAX ? ncclMax : ncclSum, r->comm, static_cast<hipStream_t>(stream));
  return e == ncclSuccess ? MIA_HIP_OK : rccl_fail(r, "ncclAllReduce", e);
}
int rccl_query(void* user, int32_t* n_ranks, int32_t* rank) {
  RcclRank* r = static_cast<RcclRank*>(user);
  int c = -1, u = -1;
  if (!r->api->CommCount || !r->api->CommUserRank) return MIA_HIP_ERR_DEVICE;
  if (r->api->CommCount(r->comm, &c) != ncclSuccess || r->api->CommUserRank(r->comm, &u) != ncclSuccess) return MIA_HIP_ERR_DEVICE;
  if (n_ranks) *n_ranks = c;
  if (rank) *rank = u;
  return MIA_HIP_OK;
}
void rccl_abort(void* user) {
  RcclRank* r = static_cast<RcclRank*>(user);
  if (r->comm && r->api->CommAbort) { (void)r->api->CommAbort(r->comm); r->comm = nullptr; }
}
void rccl_destroy(void* user) {
  RcclRank* r = static_cast<RcclRank*>(user);
  if (r->comm) (void)r->api->CommDestroy(r->comm);
  delete r;
}
const char* rccl_error(void* user) { return static_cast<RcclRank*>(user)->err.c_str(); }

// ---- loopback --------------------------------------------------------------------------------------------------------
__global__ void k_loop_reduce(const int32_t* __restrict__ all, int W, size_t count, int op, int32_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    int32_t v = all[i];
    for (int r = 1; r < W; r++) {
      const int32_t x = all[(size_t)r * count + i];
      v = op == MIA_HIP_OP_MAX ? (x > v ? x : v) : v + x;
    }
    out[i] = v;
  }
}

struct LoopGroup {
  int W = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0, attached = 0;
  uint64_t gen = 0;
  bool aborted = false;
  std::vector<const void*> ptr;     // what each rank shows the others in the current collective
  std::vector<size_t> size;         // ... and how much of it: every rank must bring the same amount
  double timeout_s = 120.0;
  // every rank arrives, or the group is aborted (a rank failed and will never come), or the wait times out
  bool barrier() {
    std::unique_lock<std::mutex> lk(m);
    if (aborted) return false;
    const uint64_t g = gen;
    if (++arrived == W) { arrived = 0; gen++; cv.notify_all(); return true; }
    const bool in_time = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return gen != g || aborted; });
    if (!in_time) { aborted = true; cv.notify_all(); return false; }
    return !aborted;
  }
};
struct LoopRank {
  LoopGroup* g = nullptr;
  int rank = 0;
  int32_t* scratch = nullptr;
  size_t scratch_words = 0;
  std::string err;
};
// a rank that fails takes the group down with it: the others' barriers return instead of waiting for a rank that will not come
int loop_fail(LoopRank* r, const std::string& what) {
  r->err = "loopback: " + what;
  std::lock_guard<std::mutex> lk(r->g->m);
  r->g->aborted = true;
  r->g->cv.notify_all();
  return MIA_HIP_ERR_DEVICE;
}

// show `p` (n bytes) to the other ranks once everything queued on `stream` has happened; true if all ranks brought n bytes
int loop_publish(LoopRank* r, const void* p, size_t n, hipStream_t stream) {
  LoopGroup* g = r->g;
  if (hipStreamSynchronize(stream) != hipSuccess) return loop_fail(r, "stream error before the exchange");
  { std::lock_guard<std::mutex> lk(g->m); g->ptr[(size_t)r->rank] = p; g->size[(size_t)r->rank] = n; }
  if (!g->barrier()) { r->err = "loopback: a rank left the group (error on that rank) or did not arrive within the time limit"; return MIA_HIP_ERR_STATE; }
  for (int k = 0; k < g->W; k++)
    if (g->size[(size_t)k] != n) { r->err = "loopback: the ranks disagree on the size of a collective"; return MIA_HIP_ERR_ARG; }   // (every rank sees the same sizes: all of them return)
  return MIA_HIP_OK;
}
int loop_all_gather(void* user, const void* send, void* recv, size_t bytes, void* stream_v) {
  LoopRank* r = static_cast<LoopRank*>(user);
  LoopGroup* g = r->g;
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (int rc = loop_publish(r, send, bytes, stream)) return rc;
  for (int k = 0; k < g->W; k++)
    if (bytes && hipMemcpyAsync(static_cast<char*>(recv) + (size_t)k * bytes, g->ptr[(size_t)k], bytes, hipMemcpyDefault, stream) != hipSuccess)
      return loop_fail(r, "copy from a peer's buffer");
  if (hipStreamSynchronize(stream) != hipSuccess) return loop_fail(r, "stream error in all_gather");
  if (!g->barrier()) { r->err = "loopback: a rank left the group"; return MIA_HIP_ERR_STATE; }     // nobody reuses its send buffer before all have read it
  return MIA_HIP_OK;
}
int loop_all_reduce(void* user, void* buf, size_t count, int op, void* stream_v) {
  LoopRank* r = static_cast<LoopRank*>(user);
  LoopGroup* g = r->g;
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (int rc = loop_publish(r, buf, count * 4, stream)) return rc;
  const size_t need = count * (size_t)g->W;
  if (need > r->scratch_words) {
    if (r->scratch) (void)hipFree(r->scratch);
    r->scratch = nullptr; r->scratch_words = 0;
    if (hipMalloc((void**)&r->scratch, need * 4 + 64) != hipSuccess) return loop_fail(r, "hipMalloc of the reduction scratch");
    r->scratch_words = need;
  }
  for (int k = 0; k < g->W; k++)
    if (count && hipMemcpyAsync(r->scratch + (size_t)k * count, g->ptr[(size_t)k], count * 4, hipMemcpyDefault, stream) != hipSuccess)
      return loop_fail(r, "copy from a peer's buffer");
  if (hipStreamSynchronize(stream) != hipSuccess) return loop_fail(r, "stream error in all_reduce");
  if (!g->barrier()) { r->err = "loopback: a rank left the group"; return MIA_HIP_ERR_STATE; }     // every rank has read every input: now they may be overwritten
  if (count) {
    const unsigned grid = (unsigned)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_loop_reduce, dim3(grid), dim3(256), 0, stream, (const int32_t*)r->scratch, g->W, count, op, static_cast<int32_t*>(buf));
    if (hipGetLastError() != hipSuccess) { r->err = "loopback: reduction kernel launch"; return MIA_HIP_ERR_DEVICE; }
  }
  return MIA_HIP_OK;
}
int loop_query(void* user, int32_t* n_ranks, int32_t* rank) {
  LoopRank* r = static_cast<LoopRank*>(user);
  if (n_ranks) *n_ranks = r->g->W;
  if (rank) *rank = r->rank;
  return MIA_HIP_OK;
}
void loop_abort(void* user) {
  LoopGroup* g = static_cast<LoopRank*>(user)->g;
  std::lock_guard<std::mutex> lk(g->m);
  g->aborted = true;
  g->cv.notify_all();
}
void loop_destroy(void* user) {
  LoopRank* r = static_cast<LoopRank*>(user);
  if (r->scratch) (void)hipFree(r->scratch);
  delete r;
}
const char* loop_error(void* user) { return static_cast<LoopRank*>(user)->err.c_str(); }

}  // namespace

int mia_comm_rccl_table(const void* id128, int32_t n_ranks, int32_t rank, mia_hip_collectives* out, std::string* err) {
  RcclApi* api = rccl_api();
  if (!api) { if (err) *err = "librccl.so.1 (RCCL) could not be opened"; return MIA_HIP_ERR_DEVICE; }
  RcclRank* r = new (std::nothrow) RcclRank;
  if (!r) return MIA_HIP_ERR_NOMEM;
  r->api = api;
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  const ncclResult_t e = api->CommInitRank(&r->comm, n_ranks, id, rank);
  if (e != ncclSuccess) {
    if (err) *err = std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(e) : "RCCL error");
    delete r;
    return MIA_HIP_ERR_DEVICE;
  }
  memset(out, 0, sizeof *out);
  out->user = r; out->n_ranks = n_ranks; out->rank = rank;
  out->all_gather = rccl_all_gather; out->all_reduce_i32 = rccl_all_reduce; out->query = rccl_query;
  out->abort = rccl_abort; out->destroy = rccl_destroy; out->error = rccl_error;
  out->name = "rccl";
  return MIA_HIP_OK;
}

extern "C" int mia_hip_comm_unique_id(void* id128) {
  static_assert(MIA_HIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id handed between the ranks is ncclUniqueId");
  if (!id128) return MIA_HIP_ERR_ARG;
  RcclApi* api = rccl_api();
  if (!api) return MIA_HIP_ERR_DEVICE;
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return MIA_HIP_ERR_DEVICE;
  memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return MIA_HIP_OK;
}

extern "C" int mia_hip_loopback_create(int32_t n_ranks, void** group) {
  if (!group || n_ranks < 1 || n_ranks > 256) return MIA_HIP_ERR_ARG;
  LoopGroup* g = new (std::nothrow) LoopGroup;
  if (!g) return MIA_HIP_ERR_NOMEM;
  g->W = n_ranks;
  g->ptr.assign((size_t)n_ranks, nullptr);
  g->size.assign((size_t)n_ranks, 0);
  if (const char* t = getenv("MIA_HIP_LOOPBACK_TIMEOUT")) { const double s = atof(t); if (s > 0) g->timeout_s = s; }
  *group = g;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_loopback_table(void* group, int32_t rank, mia_hip_collectives* out) {
  LoopGroup* g = static_cast<LoopGroup*>(group);
  if (!g || !out || rank < 0 || rank >= g->W) return MIA_HIP_ERR_ARG;
  LoopRank* r = new (std::nothrow) LoopRank;
  if (!r) return MIA_HIP_ERR_NOMEM;
  r->g = g; r->rank = rank;
  memset(out, 0, sizeof *out);
  out->user = r; out->n_ranks = g->W; out->rank = rank;
  out->all_gather = loop_all_gather; out->all_reduce_i32 = loop_all_reduce; out->query = loop_query;
  out->abort = loop_abort; out->destroy = loop_destroy; out->error = loop_error;
  out->name = "loopback";
  return MIA_HIP_OK;
}

extern "C" void mia_hip_loopback_destroy(void* group) { delete static_cast<LoopGroup*>(group); }
