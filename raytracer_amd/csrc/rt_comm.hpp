// rt_comm.hpp — part of the ONE translation unit rt355.hip (included there, after rt_ctx; not a header for anyone else).
// RCCL behind the C ABI: the communicator of a context and the two collectives of the sparse tile exchange (the counts of
// every rank, the gather of the occupied tiles to the root as grouped ncclSend / ncclRecv).  librccl is bound at run time
// (dlopen / dlsym), so librt355.so carries no link-time dependency on it.
#pragma once

#include <dlfcn.h>

namespace {

// The slice of RCCL's C API this file uses (rccl.h of ROCm 7: ncclResult_t and ncclDataType_t are ints, ncclComm_t an
// opaque pointer, ncclUniqueId 128 opaque bytes passed BY VALUE to ncclCommInitRank; stable since NCCL 2.7).
struct RcclUniqueId { char internal[RT_COMM_ID_BYTES]; };
constexpr int kRcclSuccess = 0, kRcclUint8 = 1, kRcclUint32 = 3;
struct RcclApi {
  void* handle = nullptr;
  int version = 0;
  int (*GetVersion)(int*) = nullptr;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

// Binds librccl once per process.  Order: the copy already mapped into the process (RTLD_NOLOAD: torch's, when the host
// is raytracer_amd — two RCCLs in one process would each bring their own kernels and IPC state), RT355_RCCL_LIB, the
// loader's search path, ROCm's directory.
// (`why`: the caller's own string — the reason when nullptr comes back)
const RcclApi* rccl_api(std::string& why_not) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle) return &g_rccl;
  std::string& error = why_not;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so"})
    if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
  if (!h) {
    const char* env = getenv("RT355_RCCL_LIB");
    if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
  }
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
    if (!h) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    const char* why = dlerror();
    error = std::string("librccl.so.1 could not be loaded (") + (why ? why : "no reason given") +
                   "); set RT355_RCCL_LIB to its path";
    return nullptr;
  }
  RcclApi api;
  api.handle = h;
  auto bind = [&](auto& fn, const char* name) {
    fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, name));
    if (!fn && error.empty()) error = std::string("librccl has no symbol ") + name;
    return fn != nullptr;
  };
  error.clear();
  bool ok = true;     // (every symbol is looked up: the message names the first one that is missing)
  ok = bind(api.GetVersion, "ncclGetVersion") && ok;
  ok = bind(api.GetUniqueId, "ncclGetUniqueId") && ok;
  ok = bind(api.CommInitRank, "ncclCommInitRank") && ok;
  ok = bind(api.CommDestroy, "ncclCommDestroy") && ok;
  ok = bind(api.GetErrorString, "ncclGetErrorString") && ok;
  ok = bind(api.AllGather, "ncclAllGather") && ok;
  ok = bind(api.Send, "ncclSend") && ok;
  ok = bind(api.Recv, "ncclRecv") && ok;
  ok = bind(api.GroupStart, "ncclGroupStart") && ok;
  ok = bind(api.GroupEnd, "ncclGroupEnd") && ok;
  if (!ok) return nullptr;
  (void)api.GetVersion(&api.version);
  g_rccl = api;
  return &g_rccl;
}

#define RT_RCCL(ctx, api, call)                                                                                     \
  do {                                                                                                              \
    const int r_ = (call);                                                                                          \
    if (r_ != kRcclSuccess) return fail(ctx, RT_EDEVICE, "%s failed: %s", #call, (api)->GetErrorString(r_));        \
  } while (0)

}  // namespace

extern "C" {

int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]) {
  if (!id) return fail(nullptr, RT_EINVAL, "rt_comm_unique_id: id is NULL");
  std::string why;
  const RcclApi* api = rccl_api(why);
  if (!api) return fail(nullptr, RT_EDEVICE, "rt_comm_unique_id: %s", why.c_str());
  RcclUniqueId u;
  RT_RCCL(nullptr, api, api->GetUniqueId(&u));
  memcpy(id, u.internal, RT_COMM_ID_BYTES);
  return RT_OK;
}

int rt_comm_init(rt_ctx* ctx, const uint8_t id[RT_COMM_ID_BYTES], int n_ranks, int rank) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks)
    return fail(ctx, RT_EINVAL, "rt_comm_init: bad argument (id %p, n_ranks %d, rank %d)", (const void*)id, n_ranks, rank);
  if (ctx->comm) return fail(ctx, RT_ESTATE, "rt_comm_init: the context already holds a communicator (rt_comm_destroy first)");
  std::string why;
  const RcclApi* api = rccl_api(why);
  if (!api) return fail(ctx, RT_EDEVICE, "rt_comm_init: %s", why.c_str());
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RcclUniqueId u;
  memcpy(u.internal, id, RT_COMM_ID_BYTES);
  void* comm = nullptr;
  RT_RCCL(ctx, api, api->CommInitRank(&comm, n_ranks, u, rank));
  ctx->comm = comm; ctx->comm_ranks = n_ranks; ctx->comm_rank = rank;
  return RT_OK;
}

int rt_comm_destroy(rt_ctx* ctx) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (!ctx->comm) return RT_OK;
  std::string why;
  const RcclApi* api = rccl_api(why);
  void* comm = ctx->comm;
  ctx->comm = nullptr; ctx->comm_ranks = 0; ctx->comm_rank = -1;
  if (!api) return fail(ctx, RT_EDEVICE, "rt_comm_destroy: %s", why.c_str());
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_RCCL(ctx, api, api->CommDestroy(comm));
  return RT_OK;
}

int rt_comm_info(const rt_ctx* ctx, int* n_ranks, int* rank, int* rccl_version) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (n_ranks) *n_ranks = ctx->comm ? ctx->comm_ranks : 0;
  if (rank) *rank = ctx->comm ? ctx->comm_rank : -1;
  if (rccl_version) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    *rccl_version = g_rccl.handle ? g_rccl.version : 0;
  }
  return RT_OK;
}

int rt_comm_all_gather_counts(rt_ctx* ctx, const uint32_t* count_dev, uint32_t* counts_all_dev, void* hip_stream) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (!ctx->comm) return fail(ctx, RT_ESTATE, "rt_comm_all_gather_counts: no communicator (rt_comm_init first)");
  if (!count_dev || !counts_all_dev) return fail(ctx, RT_EINVAL, "rt_comm_all_gather_counts: NULL buffer");
  std::string why;
  const RcclApi* api = rccl_api(why);
  if (!api) return fail(ctx, RT_EDEVICE, "rt_comm_all_gather_counts: %s", why.c_str());
  RT_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream;
  RT_RCCL(ctx, api, api->AllGather(count_dev, counts_all_dev, 1, kRcclUint32, ctx->comm, s));
  return RT_OK;
}

int rt_comm_gather(rt_ctx* ctx, const void* send, int64_t send_bytes, void* gathered, const int64_t* recv_bytes,
                   int64_t rank_stride_bytes, int root, void* hip_stream) {
  if (!ctx) return fail(nullptr, RT_EINVAL, "ctx is NULL");
  if (!ctx->comm) return fail(ctx, RT_ESTATE, "rt_comm_gather: no communicator (rt_comm_init first)");
  const int n = ctx->comm_ranks, me = ctx->comm_rank;
  if (root < 0 || root >= n || send_bytes < 0 || (send_bytes > 0 && !send))
    return fail(ctx, RT_EINVAL, "rt_comm_gather: bad argument (root %d of %d ranks, %lld bytes to send from %p)", root, n,
                (long long)send_bytes, send);
  if (me == root) {
    // everything is checked before the first call is posted: a group that one rank leaves half-posted hangs the others
    if (!recv_bytes || !gathered || rank_stride_bytes < 0)
      return fail(ctx, RT_EINVAL, "rt_comm_gather: the root needs recv_bytes, gathered and a stride");
    for (int r = 0; r < n; ++r)
      if (recv_bytes[r] < 0 || recv_bytes[r] > rank_stride_bytes)
        return fail(ctx, RT_EINVAL, "rt_comm_gather: recv_bytes[%d] = %lld does not fit the stride of %lld bytes", r,
                    (long long)recv_bytes[r], (long long)rank_stride_bytes);
    if (recv_bytes[me] != send_bytes)
      return fail(ctx, RT_EINVAL, "rt_comm_gather: the root sends %lld bytes but expects %lld from itself",
                  (long long)send_bytes, (long long)recv_bytes[me]);
  }
  std::string why;
  const RcclApi* api = rccl_api(why);
  if (!api) return fail(ctx, RT_EDEVICE, "rt_comm_gather: %s", why.c_str());
  RT_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream;
  RT_RCCL(ctx, api, api->GroupStart());
  int rc = kRcclSuccess;
  if (me == root)
    for (int r = 0; r < n && rc == kRcclSuccess; ++r)
      if (recv_bytes[r] > 0)
        rc = api->Recv(static_cast<unsigned char*>(gathered) + size_t(r) * size_t(rank_stride_bytes), size_t(recv_bytes[r]),
                       kRcclUint8, r, ctx->comm, s);
  if (rc == kRcclSuccess && send_bytes > 0) rc = api->Send(send, size_t(send_bytes), kRcclUint8, root, ctx->comm, s);
  const int rc_end = api->GroupEnd();     // (always closed: an open group would swallow every later call of the thread)
  if (rc != kRcclSuccess) return fail(ctx, RT_EDEVICE, "rt_comm_gather: ncclSend / ncclRecv failed: %s", api->GetErrorString(rc));
  if (rc_end != kRcclSuccess) return fail(ctx, RT_EDEVICE, "rt_comm_gather: ncclGroupEnd failed: %s", api->GetErrorString(rc_end));
  return RT_OK;
}

}  // extern "C"
