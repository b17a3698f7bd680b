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
const int kRcclFloat32 = 7, kRcclBfloat16 = 9, kRcclSum = 0;  // ncclFloat32, ncclBfloat16, ncclSum

struct RcclApi {
  void* lib = nullptr;
  rccl_result_t (*GetUniqueId)(RcclUniqueId*) = nullptr;
  rccl_result_t (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
  rccl_result_t (*CommDestroy)(void*) = nullptr;
  rccl_result_t (*CommCount)(void*, int*) = nullptr;
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
  // RCGAN_RCCL_LIB names the library instead (a site build of RCCL; the tests point it at a file that does not exist)
  const char* override_name = getenv("RCGAN_RCCL_LIB");
  const char* names[] = {override_name && *override_name ? override_name : "librccl.so.1", "librccl.so"};
  const int n_names = override_name && *override_name ? 1 : 2;
  for (int pass = 0; pass < 2 && !api.lib; ++pass)          // pass 0: an instance the process already carries
    for (int i = 0; i < n_names; ++i) {
      api.lib = dlopen(names[i], RTLD_NOW | (pass == 0 ? RTLD_NOLOAD : 0));
      if (api.lib) break;
    }
  if (!api.lib) {
    const char* e = dlerror();                               // one call: dlerror() clears the message it returns
    api.why = std::string("dlopen(") + names[0] + "): " + (e ? e : "not found");
    return &api;
  }
  bool ok = true;
  auto sym = [&](const char* n) { void* p = dlsym(api.lib, n); if (!p) { ok = false; api.why = std::string("missing symbol ") + n; } return p; };
  api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
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

// the test double's link model: one thread watches the constant-rate wall clock until `ticks` have passed
__global__ void stub_wait_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// bf16 buckets: fp32 -> bf16 (nearest even; inf / nan kept) and back, all buckets of a group in one launch each (blockIdx.y = bucket)
struct Bf16Buckets {
  float* f32[8];
  unsigned short* b16[8];
  size_t count[8];
};
__global__ void buckets_to_bf16_kernel(Bf16Buckets b) {
  const float* src = b.f32[blockIdx.y];
  unsigned short* dst = b.b16[blockIdx.y];
  const size_t n = b.count[blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned u = __float_as_uint(src[i]);
    dst[i] = (u & 0x7fffffffu) > 0x7f800000u ? (unsigned short)((u >> 16) | 0x40u)      // nan stays nan
                                             : (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
}
__global__ void buckets_from_bf16_kernel(Bf16Buckets b, float scale) {
  float* dst = b.f32[blockIdx.y];
  const unsigned short* src = b.b16[blockIdx.y];
  const size_t n = b.count[blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = __uint_as_float((unsigned)src[i] << 16) * scale;
}

int stub_wait(rcgan_ctx* ctx, hipStream_t stream, size_t bytes) {
  if (ctx->stub_bus_gbps <= 0.0 && ctx->stub_latency_us <= 0.0) return RCGAN_OK;
  if (ctx->wall_clock_khz == 0) {
    int khz = 0;
    RC_HIP(ctx, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    ctx->wall_clock_khz = khz > 0 ? khz : 100000;
  }
  const int N = ctx->comm_world;
  double us = ctx->stub_latency_us;
  if (ctx->stub_bus_gbps > 0.0 && N > 1) us += 2.0 * (N - 1) / N * (double)bytes / (ctx->stub_bus_gbps * 1e3);
  hipLaunchKernelGGL(stub_wait_kernel, dim3(1), dim3(1), 0, stream, (long long)(us * 1e-3 * ctx->wall_clock_khz));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// one all-reduce group on `stream`: fp32 buckets in place, or (b16 != nullptr) their bf16 images in b16[i]
int allreduce_on(rcgan_ctx* ctx, hipStream_t stream, float* const* bufs, const size_t* counts, int n, unsigned short* const* b16 = nullptr) {
  RC_REQUIRE(ctx, ctx->comm != nullptr || ctx->comm_stub, "no communicator: call rcgan_comm_init first");
  size_t bytes = 0;
  for (int i = 0; i < n; ++i) {
    RC_REQUIRE(ctx, bufs[i] != nullptr || counts[i] == 0, "null bucket %d", i);
    bytes += counts[i] * (b16 ? 2 : 4);
  }
  ProfScope ps(ctx, RCGAN_PROF_ALLREDUCE, (double)bytes, -1.0, stream);
  if (ctx->comm_stub) {
    // fault injection for the tests of the host side's fallback: a communicator whose all-reduce cannot be recorded into a graph
    if (ctx->capturing && getenv("RCGAN_COMM_STUB_FAIL_IN_CAPTURE")) RC_FAIL(ctx, RCGAN_ERCCL, "test double: all-reduce refused inside a capture");
    if (!b16)          // (bf16 buckets: the widening launch applies the factor -- world * x is what N ranks holding x sum to)
      for (int i = 0; i < n; ++i) {
        if (counts[i] == 0) continue;
        size_t blocks = (counts[i] + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(scale_inplace_kernel, dim3((int)blocks), dim3(256), 0, stream, counts[i], bufs[i], (float)ctx->comm_world);
        RC_LAUNCH_CHECK(ctx);
      }
    return stub_wait(ctx, stream, bytes);
  }
  RcclApi* api = rccl_api();
  if (n > 1) RC_RCCL(ctx, api, api->GroupStart());
  for (int i = 0; i < n; ++i) {
    if (counts[i] == 0) continue;
    void* p = b16 ? (void*)b16[i] : (void*)bufs[i];
    rccl_result_t r = api->AllReduce(p, p, counts[i], b16 ? kRcclBfloat16 : kRcclFloat32, kRcclSum, ctx->comm, stream);
    if (r != 0) {
      if (n > 1) (void)api->GroupEnd();
      RC_FAIL(ctx, RCGAN_ERCCL, "ncclAllReduce(bucket %d, %zu %s) -> %s", i, counts[i], b16 ? "bf16" : "floats", api->GetErrorString(r));
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
  ctx->stub_bus_gbps = 0.0; ctx->stub_latency_us = 0.0;
  if (ctx->comm_fork) { (void)hipEventDestroy(ctx->comm_fork); ctx->comm_fork = nullptr; }
  if (ctx->comm_join) { (void)hipEventDestroy(ctx->comm_join); ctx->comm_join = nullptr; }
  if (ctx->comm_stream) { (void)hipStreamDestroy(ctx->comm_stream); ctx->comm_stream = nullptr; }
  return RCGAN_OK;
}

int rcgan_comm_world(rcgan_ctx* ctx) { return ctx ? ctx->comm_world : 0; }

const char* rcgan_comm_load_error(void) { return rccl_api()->why.c_str(); }

int rcgan_comm_count(rcgan_ctx* ctx, int* ranks) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  RC_REQUIRE(ctx, ranks != nullptr, "null argument");
  RC_REQUIRE(ctx, ctx->comm != nullptr || ctx->comm_stub, "no communicator: call rcgan_comm_init first");
  if (ctx->comm_stub) { *ranks = ctx->comm_world; return RCGAN_OK; }
  RcclApi* api = rccl_api();
  RC_RCCL(ctx, api, api->CommCount(ctx->comm, ranks));
  return RCGAN_OK;
}

int rcgan_comm_stub_model(rcgan_ctx* ctx, double bus_gbps, double latency_us) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  RC_REQUIRE(ctx, ctx->comm_stub, "the cost model belongs to the test double (rcgan_comm_init_stub)");
  RC_REQUIRE(ctx, bus_gbps >= 0.0 && latency_us >= 0.0, "bus %g GB/s, latency %g us", bus_gbps, latency_us);
  ctx->stub_bus_gbps = bus_gbps; ctx->stub_latency_us = latency_us;
  return RCGAN_OK;
}

size_t rcgan_allreduce_bf16_scratch_bytes(int n, const size_t* counts) {
  size_t b = 0;
  for (int i = 0; i < n && counts; ++i) b += (counts[i] * 2 + 255) / 256 * 256;
  return b;
}

int rcgan_allreduce_sum_bf16_buckets(rcgan_ctx* ctx, int n, float* const* bufs, const size_t* counts, void* scratch16, size_t scratch_bytes) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  RC_REQUIRE(ctx, n >= 0 && n <= 8 && (n == 0 || (bufs && counts && scratch16)), "bad bucket list (at most 8 buckets)");
  RC_REQUIRE(ctx, ctx->comm != nullptr || ctx->comm_stub, "no communicator: call rcgan_comm_init first");
  if (n == 0) return RCGAN_OK;
  const size_t need = rcgan_allreduce_bf16_scratch_bytes(n, counts);
  if (scratch_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, scratch_bytes);
  Bf16Buckets b = {};
  size_t off = 0, most = 0;
  for (int i = 0; i < n; ++i) {
    RC_REQUIRE(ctx, bufs[i] != nullptr || counts[i] == 0, "null bucket %d", i);
    b.f32[i] = bufs[i]; b.b16[i] = (unsigned short*)((char*)scratch16 + off); b.count[i] = counts[i];
    off += (counts[i] * 2 + 255) / 256 * 256;
    if (counts[i] > most) most = counts[i];
  }
  if (most == 0) return RCGAN_OK;
  size_t blocks = (most + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(buckets_to_bf16_kernel, dim3((int)blocks, n), dim3(256), 0, ctx->stream, b);
  RC_LAUNCH_CHECK(ctx);
  int rc = allreduce_on(ctx, ctx->stream, bufs, counts, n, b.b16);
  if (rc != RCGAN_OK) return rc;
  hipLaunchKernelGGL(buckets_from_bf16_kernel, dim3((int)blocks, n), dim3(256), 0, ctx->stream, b, ctx->comm_stub ? (float)ctx->comm_world : 1.0f);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

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
