// Data-parallel gradient exchange inside the C ABI: all-reduce(sum) of fp32 gradient buckets over RCCL (xGMI), issued on the
// context's stream -- or on a communication stream beside the rest of the backward pass -- and capturable into the step's
// hipGraph.  Replaces the reference's in-graph towers (cifar10/gan_resnet.py:529-546: tf.split of every fed array over the
// devices; :697,786: tf.add_n(costs) / len(DEVICES), i.e. the gradient of the MEAN of the tower costs): one process per GPU holds
// one tower, the sum over the ranks is taken here and the 1/world factor is applied by the optimiser kernel (grad_scale).
//
// librccl is bound at run time (dlopen): a single-GPU process never loads it, and a process that already carries an RCCL (PyTorch
// links its own copy) shares that instance instead of bringing a second one into the address space.
#include <dlfcn.h>

#include <cstdlib>
#include "common.h"

namespace {

// the few RCCL declarations this file needs (rccl.h: ncclUniqueId, ncclDataType_t, ncclRedOp_t)
struct RcclUniqueId { char internal[RCGAN_COMM_ID_BYTES]; };
typedef int rccl_result_t;                 // ncclSuccess == 0
const int kRcclFloat32 = 7, kRcclSum = 0;  // ncclFloat32, ncclSum

struct RcclApi {
  void* lib = nullptr;
  rccl_result_t (*GetUniqueId)(RcclUniqueId*) = nullptr;
  rccl_result_t (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
  rccl_result_t (*CommDestroy)(void*) = nullptr;
  rccl_result_t (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  rccl_result_t (*GroupStart)() = nullptr;
  rccl_result_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(rccl_result_t) = nullptr;
  std::string why;
};

RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return &api;
  tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (int pass = 0; pass < 2 && !api.lib; ++pass)          // pass 0: an instance the process already carries
    for (const char* n : names) {
      api.lib = dlopen(n, RTLD_NOW | (pass == 0 ? RTLD_NOLOAD : 0));
      if (api.lib) break;
    }
  if (!api.lib) { api.why = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found"); return &api; }
  bool ok = true;
  auto sym = [&](const char* n) { void* p = dlsym(api.lib, n); if (!p) { ok = false; api.why = std::string("missing symbol ") + n; } return p; };
  api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
  api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
  api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
  api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  if (!ok) api.lib = nullptr;
  return &api;
}

#define RC_RCCL(ctx, api, expr)                                                                             \
  do {                                                                                                      \
    rccl_result_t _r = (expr);                                                                              \
    if (_r != 0) RC_FAIL(ctx, RCGAN_ERCCL, "%s -> %s", #expr, (api)->GetErrorString ? (api)->GetErrorString(_r) : "?"); \
  } while (0)

// the test double's "sum over the ranks": every rank holds what this rank holds
__global__ void scale_inplace_kernel(size_t count, float* p, float s) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) p[i] *= s;
}

int ensure_comm_stream(rcgan_ctx* ctx) {
  if (ctx->comm_stream) return RCGAN_OK;
  RC_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
  RC_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_fork, hipEventDisableTiming));
  RC_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_join, hipEventDisableTiming));
  return RCGAN_OK;
}

int allreduce_on(rcgan_ctx* ctx, hipStream_t stream, float* const* bufs, const size_t* counts, int n) {
  RC_REQUIRE(ctx, ctx->comm != nullptr || ctx->comm_stub, "no communicator: call rcgan_comm_init first");
  for (int i = 0; i < n; ++i) RC_REQUIRE(ctx, bufs[i] != nullptr || counts[i] == 0, "null bucket %d", i);
  if (ctx->comm_stub) {
    // fault injection for the tests of the host side's fallback: a communicator whose all-reduce cannot be recorded into a graph
    if (ctx->capturing && getenv("RCGAN_COMM_STUB_FAIL_IN_CAPTURE")) RC_FAIL(ctx, RCGAN_ERCCL, "test double: all-reduce refused inside a capture");
    for (int i = 0; i < n; ++i) {
      if (counts[i] == 0) continue;
      size_t blocks = (counts[i] + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(scale_inplace_kernel, dim3((int)blocks), dim3(256), 0, stream, counts[i], bufs[i], (float)ctx->comm_world);
      RC_LAUNCH_CHECK(ctx);
    }
    return RCGAN_OK;
  }
  RcclApi* api = rccl_api();
  if (n > 1) RC_RCCL(ctx, api, api->GroupStart());
  for (int i = 0; i < n; ++i) {
    if (counts[i] == 0) continue;
    rccl_result_t r = api->AllReduce(bufs[i], bufs[i], counts[i], kRcclFloat32, kRcclSum, ctx->comm, stream);
    if (r != 0) {
      if (n > 1) (void)api->GroupEnd();
      RC_FAIL(ctx, RCGAN_ERCCL, "ncclAllReduce(bucket %d, %zu floats) -> %s", i, counts[i], api->GetErrorString(r));
    }
  }
  if (n > 1) RC_RCCL(ctx, api, api->GroupEnd());
  return RCGAN_OK;
}

}  // namespace

extern "C" {

int rcgan_comm_unique_id(void* id_out) {
  if (!id_out) return RCGAN_EINVALID_ARG;
  RcclApi* api = rccl_api();
  if (!api->lib) return RCGAN_ERCCL;
  RcclUniqueId id;
  if (api->GetUniqueId(&id) != 0) return RCGAN_ERCCL;
  memcpy(id_out, &id, sizeof(id));
  return RCGAN_OK;
}

int rcgan_comm_init(rcgan_ctx* ctx, const void* id, int world, int rank) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  RC_REQUIRE(ctx, id != nullptr && world >= 1 && rank >= 0 && rank < world, "world %d rank %d", world, rank);
  RC_REQUIRE(ctx, ctx->comm == nullptr && !ctx->comm_stub, "communicator already initialised");
  RcclApi* api = rccl_api();
  if (!api->lib) RC_FAIL(ctx, RCGAN_ERCCL, "RCCL is not available: %s", api->why.c_str());
  RC_HIP(ctx, hipSetDevice(ctx->device));
  RcclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  void* comm = nullptr;
  RC_RCCL(ctx, api, api->CommInitRank(&comm, world, uid, rank));
  ctx->comm = comm; ctx->comm_world = world; ctx->comm_rank = rank;
  return ensure_comm_stream(ctx);
}

int rcgan_comm_init_stub(rcgan_ctx* ctx, int world) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  RC_REQUIRE(ctx, world >= 1, "world %d", world);
  RC_REQUIRE(ctx, ctx->comm == nullptr && !ctx->comm_stub, "communicator already initialised");
  ctx->comm_stub = true; ctx->comm_world = world; ctx->comm_rank = 0;
  return ensure_comm_stream(ctx);
}

int rcgan_comm_destroy(rcgan_ctx* ctx) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  if (ctx->comm) {
    RcclApi* api = rccl_api();
    if (api->lib) (void)api->CommDestroy(ctx->comm);
    ctx->comm = nullptr;
  }
  ctx->comm_stub = false; ctx->comm_world = 1; ctx->comm_rank = 0; ctx->comm_pending = false;
  if (ctx->comm_fork) { (void)hipEventDestroy(ctx->comm_fork); ctx->comm_fork = nullptr; }
  if (ctx->comm_join) { (void)hipEventDestroy(ctx->comm_join); ctx->comm_join = nullptr; }
  if (ctx->comm_stream) { (void)hipStreamDestroy(ctx->comm_stream); ctx->comm_stream = nullptr; }
  return RCGAN_OK;
}

int rcgan_comm_world(rcgan_ctx* ctx) { return ctx ? ctx->comm_world : 0; }

int rcgan_allreduce_sum(rcgan_ctx* ctx, float* buf, size_t count) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  return allreduce_on(ctx, ctx->stream, &buf, &count, 1);
}

int rcgan_allreduce_sum_buckets(rcgan_ctx* ctx, int n, float* const* bufs, const size_t* counts) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  RC_REQUIRE(ctx, n >= 0 && (n == 0 || (bufs && counts)), "bad bucket list");
  return allreduce_on(ctx, ctx->stream, bufs, counts, n);
}

int rcgan_allreduce_sum_async(rcgan_ctx* ctx, float* buf, size_t count) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  RC_REQUIRE(ctx, ctx->comm_stream != nullptr, "no communicator: call rcgan_comm_init first");
  RC_REQUIRE(ctx, !ctx->on_side, "asynchronous buckets are issued from the main stream");
  // fork: the bucket is final on the main stream here; everything launched on the main stream from now on runs beside it
  RC_HIP(ctx, hipEventRecord(ctx->comm_fork, ctx->stream));
  RC_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->comm_fork, 0));
  int rc = allreduce_on(ctx, ctx->comm_stream, &buf, &count, 1);
  // (recorded even after a failure: a capturing main stream must see the forked stream join again)
  if (hipEventRecord(ctx->comm_join, ctx->comm_stream) == hipSuccess) ctx->comm_pending = true;
  return rc;
}

int rcgan_allreduce_join(rcgan_ctx* ctx) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  if (!ctx->comm_pending) return RCGAN_OK;
  RC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->comm_join, 0));
  ctx->comm_pending = false;
  return RCGAN_OK;
}

}  // extern "C"
