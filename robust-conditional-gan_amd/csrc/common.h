// Shared helpers for the gfx950 kernels of librcgan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/rcgan_hip.h"

struct rcgan_ctx {
  int device;
  hipStream_t stream;
  std::string err;
  hipEvent_t events[64];
  bool event_made[64];
  std::vector<hipGraphExec_t> graphs;
  bool capturing;
  void* devtmp;        // small persistent device scratch (descriptor tables for batched launches)
  size_t devtmp_bytes;
  // per-kernel HIP-event profiling (bench.py roofline leg): every launch of kernel `prof_which` is bracketed
  int prof_which;
  std::vector<hipEvent_t> prof_ev;
  double prof_flops, prof_flops_exec;      // algorithmic (the reference's formulation) / executed by the kernels
  int prof_bn_in = 0;                      // launches of the section that also applied a batch norm to their staged input
  // fork/join onto a second stream (rcgan_side_begin/end/join): lets an independent kernel pair -- a layer's filter
  // gradient and its data gradient -- share the chip when neither fills it.  Capturable (event fork/join).
  hipStream_t main_stream, side_stream;
  hipEvent_t fork_ev, join_ev;
  bool on_side;
  // gradient (loss) scaling of 16-bit activations (rcgan_set_grad_scale): every gradient the loss kernels emit is multiplied by
  // gscale_host * (gscale_dev ? *gscale_dev : 1); the loss VALUES they accumulate stay unscaled
  float gscale_host;
  const float* gscale_dev;
  // data-parallel gradient exchange (comm.hip): an RCCL communicator (or the single-process test double), its own stream for
  // buckets exchanged beside the rest of the backward pass, fork / join events (capturable)
  void* comm;          // ncclComm_t
  int comm_world, comm_rank;
  bool comm_stub;      // rcgan_comm_init_stub: "every rank holds what this rank holds" -> sum = world * x
  hipStream_t comm_stream;
  hipEvent_t comm_fork, comm_join;
  bool comm_pending;   // an asynchronous bucket has not been joined yet
  // cost model of the test double (rcgan_comm_stub_model): an all-reduce group of `bytes` occupies its stream for
  // latency + 2 (world - 1) / world * bytes / bus bandwidth; 0 / 0 = free (the schedule's own cost only)
  double stub_bus_gbps, stub_latency_us;
  int wall_clock_khz;  // hipDeviceAttributeWallClockRate (the constant-rate counter the wait kernel reads)
  // grow-only device scratch of the two-step narrow data gradient (conv_direct.hip: per-pixel tap products, then col2im); (re)allocated
  // outside graph capture only -- the first call of every step is eager
  void* narrow_ws;
  size_t narrow_ws_bytes;
  void* splitr_ws;     // ... and of the split-reduction forward / data-gradient GEMMs (partial tiles)
  size_t splitr_ws_bytes;
  std::vector<void*> retired_ws;   // outgrown scratch buffers: captured graphs may still address them, freed with the context
  void* dbg_stamps;    // rcgan_debug_stamps
  // deferred parameter gradients of the projection head (head_rider.h): 0 = nothing pending, 1 = dE GEMM + parameter sums,
  // 2 = parameter sums; the argument block is loss.hip's
  int head_stage;
  alignas(8) unsigned char head_blob[640];
  int num_cus;         // compute units of the device (grid size of the persistent kernels)
  void* zero_page;     // 36 KiB of device memory: bytes [0,256) stay zero (halo source of the LDS-DMA kernels);
                       // bytes [1024,4096) are self-resetting arrival counters of the "last workgroup finishes" kernels
  unsigned* counters() const { return (unsigned*)((char*)zero_page + 1024); }
  // bytes [4096, 4096+32768): 8192 more self-resetting arrival counters (two-level reductions: one per cluster + one per launch)
  unsigned* tree_counters() const { return (unsigned*)((char*)zero_page + 4096); }
  // bytes [36864, 36864+65536): arrival counters of the batch-norm column reductions, ONE PER 128-BYTE LINE (RC_LINE_STRIDE words
  // apart): counters that share a line are served one after the other by the memory-side atomic unit
  unsigned* line_counters() const { return (unsigned*)((char*)zero_page + 36864); }
  // bytes [102400, 102400 + 1 MiB): 8192 line counters of the two-level (tree) reductions, one per cluster + one per launch
  unsigned* tree_line_counters() const { return (unsigned*)((char*)zero_page + 102400); }
};
#define RC_ZERO_PAGE_BYTES (102400 + 8192 * 128)
#define RC_LINE_STRIDE 32      // words between two line counters
#define RC_LCOUNTER_BN 0       // lines [0,32)
#define RC_LCOUNTER_BNSEG 32   // lines [32,288)
#define RC_COUNTER_BN 0        // [0,32): one per 64-channel column block of the batch-norm reductions
#define RC_COUNTER_WGRAD 32    // [32,..): filter-gradient finish
#define RC_COUNTER_HEAD 500    // loss partials of the fused projection head
#define RC_COUNTER_INPUTS 501  // the step-input rider of the filter preparation (step_inputs.h)
#define RC_COUNTER_BNSEG 512   // [512,768): segmented forward batch norm, one per (segment, 64-channel column block)

// grow-only scratch (*buf, *cap) of at least `need` bytes; never inside a capture, and an outgrown buffer stays allocated (a graph
// captured earlier replays launches that address it)
static inline hipError_t ctx_grow_scratch(rcgan_ctx* c, void** buf, size_t* cap, size_t need) {
  if (*cap >= need) return hipSuccess;
  if (c->capturing) return hipErrorStreamCaptureUnsupported;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, need);
  if (e != hipSuccess) return e;
  if (*buf) c->retired_ws.push_back(*buf);
  *buf = p; *cap = need;
  return hipSuccess;
}

// brackets one launch with events when profiling is armed for kernel id `which`
struct ProfScope {
  rcgan_ctx* c; bool on;
  hipStream_t s;
  ProfScope(rcgan_ctx* ctx, int which, double flops, double executed = -1.0, hipStream_t on_stream = nullptr)
      : c(ctx), on(ctx->prof_which == which && !ctx->capturing), s(on_stream ? on_stream : ctx->stream) {
    if (!on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(e, s);
    c->prof_ev.push_back(e);
    c->prof_flops += flops;
    c->prof_flops_exec += executed < 0.0 ? flops : executed;
  }
  ~ProfScope() {
    if (!on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    c->prof_ev.push_back(e);
  }
};

#define RC_FAIL(ctx, code, ...)                         \
  do {                                                  \
    char _b[512];                                       \
    snprintf(_b, sizeof(_b), __VA_ARGS__);              \
    (ctx)->err = std::string(__func__) + ": " + _b;     \
    return (code);                                      \
  } while (0)

#define RC_HIP(ctx, expr)                                                           \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) RC_FAIL(ctx, RCGAN_EHIP, "%s -> %s", #expr, hipGetErrorString(_e)); \
  } while (0)

#define RC_LAUNCH_CHECK(ctx)                                                        \
  do {                                                                              \
    hipError_t _e = hipGetLastError();                                              \
    if (_e != hipSuccess) RC_FAIL(ctx, RCGAN_EHIP, "launch -> %s", hipGetErrorString(_e)); \
  } while (0)

#define RC_REQUIRE(ctx, cond, ...)                                  \
  do {                                                              \
    if (!(cond)) RC_FAIL(ctx, RCGAN_EINVALID_ARG, __VA_ARGS__);     \
  } while (0)

// ---------------------------------------------------------------------------------------------
// 16-bit storage helpers.  One source tree, two builds: bf16 (default) or IEEE fp16 (-DRCGAN_HALF_FP16=1).  Every
// kernel goes through these two conversions, the MFMA wrapper and the constants in mfma_util.h, so the format is a
// compile-time property of the library; the type keeps its historical name.  Both round to nearest even.
// ---------------------------------------------------------------------------------------------
#ifndef RCGAN_HALF_FP16
#define RCGAN_HALF_FP16 0
#endif
#define RCGAN_H16 (RCGAN_HALF_FP16 ? RCGAN_F16 : RCGAN_BF16)
typedef uint16_t bf16_t;

#if RCGAN_HALF_FP16
__host__ __device__ inline float bf16_to_f32(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__host__ __device__ inline bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }   // v_cvt_f16_f32
#else
__host__ __device__ inline float bf16_to_f32(bf16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  f = __uint_as_float(u);
#else
  memcpy(&f, &u, 4);
#endif
  return f;
}

__host__ __device__ inline bf16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(bf16_t, (__bf16)f);      // v_cvt_pk_bf16_f32: round-to-nearest-even, quiet NaN
#else
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
#endif
}
#endif

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

static inline size_t dtype_size(int dtype) { return dtype == RCGAN_H16 ? 2 : 4; }

// dispatch a templated launcher over the activation dtype
#define RC_DISPATCH_DTYPE(ctx, dtype, ...)                              \
  do {                                                                  \
    if ((dtype) == RCGAN_F32) { typedef float T; __VA_ARGS__; }         \
    else if ((dtype) == RCGAN_H16) { typedef bf16_t T; __VA_ARGS__; }   \
    else RC_FAIL(ctx, RCGAN_EINVALID_ARG, "bad dtype %d", (int)(dtype)); \
  } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread
__device__ __forceinline__ float block_sum256(float v, float* red /* >= 4 floats of LDS */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0] + red[1] + red[2] + red[3];
  return r;
}

__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case RCGAN_ACT_RELU: return x > 0.f ? x : 0.f;
    case RCGAN_ACT_LRELU: return fmaxf(x, 0.2f * x);
    case RCGAN_ACT_TANH: return tanhf(x);
    case RCGAN_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    default: return x;
  }
}

// derivative factor: for relu/lrelu `s` is the pre-activation input (or the output: same sign);
// for tanh/sigmoid `s` is the OUTPUT y.
__device__ __forceinline__ float act_grad(int act, float s) {
  switch (act) {
    case RCGAN_ACT_RELU: return s > 0.f ? 1.f : 0.f;
    case RCGAN_ACT_LRELU: return s > 0.f ? 1.f : 0.2f;
    case RCGAN_ACT_TANH: return 1.f - s * s;
    case RCGAN_ACT_SIGMOID: return s * (1.f - s);
    default: return 1.f;
  }
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// TF SAME padding (tf.nn.conv2d padding='SAME')
static inline void same_pad(int in, int k, int s, int* out, int* before) {
  int o = (in + s - 1) / s;
  int tot = (o - 1) * s + k - in;
  if (tot < 0) tot = 0;
  *out = o;
  *before = tot / 2;
}
