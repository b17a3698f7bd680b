// The 8x8 stage of the discriminator as ONE launch: D.Block.3 .. D.Block.6 (gan_resnet.py:275-328, 398-404), four identity-
// shortcut residual blocks = eight 3x3 convolutions 128 -> 128 on 8 x 8 pixels, forward or data gradient.
//
// Why: one of these convolutions is 1.2 GFLOP per 64 images.  As its own launch it is a 64 x 64-tile kernel with a
// K-split over 16 wavefronts that takes 8-9 us (launch, pipeline fill, partial-sum exchange, drain) for ~1 us of matrix
// work per CU -- 8 launches forward and 8 backward in every discriminator pass, six passes per iteration.  There is no
// norm layer in D, so an image never meets another image: a workgroup carries ONE image through all eight layers.
//   * activations never leave the CU: 64 pixels x 128 channels (16 KB) live in LDS as a zero-padded 10 x 10 image, one
//     256-byte row per pixel (3x3 taps = row offsets, no bounds tests), 16-byte slots XOR-swizzled with the pixel index
//     (a fragment read takes the same slot of 16 different pixels); two such images alternate between layers
//   * 4 wavefronts, one per SIMD, 32 output channels x all 64 pixels each: 8 accumulator tiles; the FILTERS of a whole layer
//     sit in the wavefront's registers (below)
//   * the epilogue keeps the residual operand in registers (same lane = same (pixel, channel) in every layer), writes the
//     layer's output to HBM once (the backward pass and the filter gradients need it) and the next layer's input to LDS
//   * one raw barrier per layer; nothing waits for HBM stores
// Forward layer 2b (conv1 of block b): h = conv(relu(x)) + bias;      layer 2b+1: x' = x + conv(relu(h)) + bias
// Backward runs the layers in reverse with the rotated filters: dh = dgrad2(dy) * (h > 0);  dx = dy + dgrad1(dh) * (x > 0).
//
// History.  Round 2 streamed the filter rows through per-wavefront LDS-DMA rings (three taps = 24 KiB in flight): 79 us against
// 70 us for the eight launches -- measured and rejected.  Round 3 found why: (1) a lane-per-row fetch (64 bytes from each of 16
// rows 2304 bytes apart per instruction) crawls -- the same bytes as contiguous KiB blocks arrive ~15x faster -- hence the
// fragment-major copy of the filters; (2) one wavefront per SIMD needs a WHOLE LAYER of filter loads in flight to cover the L2
// round trip, which only registers can hold -- hence the 288-register ring with loads issued from asm.
#include <type_traits>

#include "conv_mfma.h"
#include "head_rider.h"
#include "mfma_util.h"

namespace {

constexpr int TR_C = 128;                 // channels
constexpr int TR_K = 9 * TR_C;            // reduction length of one layer
constexpr int TR_ROW = TR_C * 2;          // bytes of one padded pixel (all channels)
constexpr int TR_BUF = 100 * TR_ROW;      // one zero-padded 10 x 10 activation image
constexpr int TR_LAYERS = 8;

struct TrunkArgs {
  const bf16_t* x0;                       // [n][64][128]: trunk input (forward) / gradient of the trunk output (backward)
  const bf16_t* w[TR_LAYERS];             // the layers' filters in execution order, fragment-major (trunk_fragments_kernel)
  const float* bias[TR_LAYERS];           // forward (null: none)
  const bf16_t* mask[TR_LAYERS];          // backward: saved forward activation whose sign gates the layer's output
  bf16_t* out[TR_LAYERS];                 // [n][64][128] per layer
  int backward;
  unsigned long long* stamps;             // diagnostics (rcgan_debug_stamps): 24 s_memtime stamps per workgroup, normally null
  // rider: the projection head's deferred dE = dlogit^T feat (head_rider.h) on workgroups n .. n + cdiv(d, 16) - 1 of the backward
  // launch -- the stage's own workgroups occupy one CU each, at n = 128 half the chip is free beside them
  int n;
  SmallGemmArgs gemm;
  // the discriminator's pooled features (gan_resnet.py:405-407: relu, mean over the 64 pixels) at the stage's boundary:
  //   forward : feat != null -> feat[n][128] = mean_p relu(out[7][n][p][:]) of the stored 16-bit values, from the last layer's epilogue
  //   backward: feat != null -> the incoming gradient is built in the prologue from dfeat = feat[n][128] and xlast = the stage's
  //             stored output: dy[p][c] = xlast[p][c] > 0 ? dfeat[c] / 64 : 0, rounded to 16 bits (x0 unused)
  // -- what the projection head's own pooling (loss.hip, HeadArgs::x) computes, without its 2 + 2 memory round trips
  float* feat;
  const bf16_t* xlast;
  bf16_t* dy_out;                         // backward with feat: the formed gradient [n][64][128] (the last layer's filter gradient reads it)
};

__device__ __forceinline__ void trunk_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the same stage with the FILTERS IN REGISTERS.  What bounded the kernel above was its filter stream: 295 KB per layer
// and workgroup through per-wavefront LDS rings with three taps (24 KiB) in flight -- 4 us per layer against 2 us of matrix work
// -- and, without the stream, the LDS-read -> MFMA latency of one wavefront per SIMD.  Here a wavefront keeps the fragments of a
// WHOLE layer for its 32 output channels in registers (36 K-steps x 2 channel tiles x 16 bytes per lane = 288 registers of the
// 512 a single wavefront per SIMD may use) and re-issues every fragment's load for the NEXT layer the moment the MFMAs of the
// current layer have consumed it: each load has a full layer (36 K-steps, ~2 us) to come back from L2, 147 KB in flight per
// CU, and no filter byte passes through LDS, whose port is left to the pixel fragments (4 x 16 B per lane and K-step, read one
// K-step ahead of their MFMAs).  The compiler keeps the loads in program order; its own counted vmcnt waits do the rest.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TRW_LDS = 2 * TR_BUF + TR_LAYERS * TR_C * 4;      // two activation images + biases
static_assert((SG_AS_FLOATS + SG_RED_FLOATS) * 4 <= TRW_LDS, "the riding small GEMM needs more LDS than the stage");
#ifndef TRW_ABLATE
#define TRW_ABLATE 0      // diagnostics only (scripts/probes): 1 = no pixel-fragment reads after the first two K-steps, 2 = no filter reloads
#endif
constexpr int TRW_ASTEPS = 32;            // K-steps whose filter fragments live in AccVGPRs (2 x 4 registers each: 256); the other 4 in VGPRs

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

#if RCGAN_HALF_FP16
#define TRW_MFMA "v_mfma_f32_16x16x32_f16"
#else
#define TRW_MFMA "v_mfma_f32_16x16x32_bf16"
#endif

// The compiler must not know these loads: it would sink them next to their use, or spill what it cannot colour (a plain C++ array
// of 72 fragments came out with 85 spills and a vmcnt(0) behind every load).  Issued from asm into a register class chosen here,
// waited for by hand (trw_wait), consumed by asm MFMAs: nothing between a load and its use a layer later is visible to it.
// (OFF: immediate byte offset: the second channel tile of a K-step sits 1 KiB behind the first)
template <int OFF> __device__ __forceinline__ void trw_load_a(i32x4_t& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(dst) : "v"(p), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void trw_load_v(i32x4_t& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void trw_load2_v(uint2& dst, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
__device__ __forceinline__ void trw_mfma_a(f32x4_t& acc, const i32x4_t& w, const bf16x8_t& x) {
  asm volatile(TRW_MFMA " %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
}
// first K-step of a layer: C = 0 as an inline constant instead of a VALU zero-fill of the accumulators -- the compiler cannot see these
// MFMAs, so it cannot insert the wait states "VALU write -> MFMA SrcC read" needs (conv_rf.hip lost a tile to exactly that)
__device__ __forceinline__ void trw_mfma_a0(f32x4_t& acc, const i32x4_t& w, const bf16x8_t& x) {
  asm volatile(TRW_MFMA " %0, %1, %2, 0" : "=v"(acc) : "a"(w), "v"(x));
}
__device__ __forceinline__ void trw_mfma_v(f32x4_t& acc, const i32x4_t& w, const bf16x8_t& x) {
  asm volatile(TRW_MFMA " %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
}
// Memory operations retire in order.  Between the load of a fragment and its use one layer later this wavefront issues the other
// 70 fragment loads, the 16 output stores of the epilogue and (backward) 8 mask loads: always more than 63 younger operations, so
// "at most 63 outstanding" implies the fragment has landed -- and leaves the 63 youngest (~23 K-steps of filters) in flight.
__device__ __forceinline__ void trw_wait() { asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); }

// rows [128][1152] of up to 2 * TR_LAYERS filters (rcgan_conv_prepare layout) -> fragment-major [layer][wavefront 4][K-step 36][channel tile 2]
// [lane 64][8]: lane (r = l & 15, kc = l >> 4) of (wavefront w, K-step s, tile ct) owns row w*32 + ct*16 + r, elements s*32 + kc*8 ..+8.
// One thread per 16-byte fragment piece; the writes of a wavefront are one contiguous KiB.
struct TrunkFragArgs { const bf16_t* w[2 * TR_LAYERS]; bf16_t* out; };
__global__ __launch_bounds__(256) void trunk_fragments_kernel(TrunkFragArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;                  // [layer][w][s][ct][lane]
  const int lane = i & 63, ct = (i >> 6) & 1, rest = i >> 7;     // rest = (layer * 4 + w) * 36 + s
  const int s = rest % 36, lw = rest / 36, w = lw & 3, layer = lw >> 2;
  const int r = lane & 15, kc = lane >> 4;
  const uint4 v = *(const uint4*)(a.w[layer] + (long)(w * 32 + ct * 16 + r) * TR_K + s * 32 + kc * 8);
  *(uint4*)(a.out + (long)i * 8) = v;
}

template <int I, int N, typename F> __device__ __forceinline__ void trw_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); trw_for<I + 1, N>(f); }
}

template <bool BWD>
__global__ __launch_bounds__(256) void conv_trunk_rw_kernel(TrunkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (BWD && __builtin_expect((int)blockIdx.x >= a.n, 0)) {          // rider workgroups (see TrunkArgs)
    small_gemm_body(a.gemm, (int)blockIdx.x - a.n, (float*)smem, (float*)smem + SG_AS_FLOATS);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kc = lane >> 4;
  const long img = blockIdx.x;
  float* const bias_s = (float*)(smem + 2 * TR_BUF);             // [layer][128]
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (a.stamps && tid == 0) a.stamps[(long)blockIdx.x * 24 + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);

  // this lane's filter rows: channel wave*32 + ct*16 + r, reduction offset kc*8 (MFMA A operand); K-step s = tap * 4 + quarter
  // starts s * 32 elements into the row
  // The filters arrive FRAGMENT-MAJOR (trunk_fragments_kernel): the 16 bytes lane l feeds the MFMA of (wavefront, K-step s,
  // channel tile ct) sit at [layer][wavefront][s][ct][l] -- one load instruction reads ONE contiguous KiB.  (From the row-major
  // prepared layout the same instruction took 64 bytes from each of 16 rows 2304 bytes apart and the kernel spent 6 us per layer
  // waiting for them: 80 us against 35 us for the loads alone, scripts/bench_trunk.py.)
  const long wlane = (long)wave * (36 * 1024) + lane * 8;          // elements; K-step s: + s * 1024, channel tile 1: + 512
  i32x4_t WA[TRW_ASTEPS][2], WV[36 - TRW_ASTEPS][2];
  auto wload = [&](auto ic, const bf16_t* wl) __attribute__((always_inline)) {
    constexpr int s = decltype(ic)::value;
    const bf16_t* const p = wl + s * 1024;
    if constexpr (s < TRW_ASTEPS) { trw_load_a<0>(WA[s][0], p); trw_load_a<1024>(WA[s][1], p); }
    else { trw_load_v<0>(WV[s - TRW_ASTEPS][0], p); trw_load_v<1024>(WV[s - TRW_ASTEPS][1], p); }
  };
  {
    const bf16_t* const wl = a.w[0] + wlane;
    trw_for<0, 36>([&](auto ic) __attribute__((always_inline)) { wload(ic, wl); });
  }

  for (int i = tid; i < 2 * TR_BUF / 16; i += 256) ((uint4*)smem)[i] = make_uint4(0u, 0u, 0u, 0u);
  if (!BWD)
    for (int i = tid; i < TR_LAYERS * TR_C; i += 256) {
      const float* bp = a.bias[i >> 7];
      bias_s[i] = bp ? bp[i & 127] : 0.f;
    }
  int pp0[4];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    const int p = pt * 16 + r;
    pp0[pt] = (p >> 3) * 10 + (p & 7);
  }
  const int co4[2] = {wave * 32 + 4 * kc, wave * 32 + 16 + 4 * kc};

  // residual operand (block input / block output gradient): the stored 16-bit values, four channels per register pair
  uint2 R[2][4];
  const long gbase = img * 64 * TR_C;
  __syncthreads();                        // the zero fill is complete before the first interior write
  const bool pooled_in = BWD && a.feat != nullptr;        // (workgroup-uniform)
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pooled_in) {
      g4 = *(const float4*)(a.feat + img * TR_C + co4[ct]);
      const float inv = 1.f / 64.f;
      g4.x *= inv; g4.y *= inv; g4.z *= inv; g4.w *= inv;
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const int p = pt * 16 + r;
      uint2 v;
      if (pooled_in) {
        const uint2 xl = *(const uint2*)(a.xlast + gbase + (long)p * TR_C + co4[ct]);
        v.x = pack_h16x2(h16_lo(xl.x) > 0.f ? g4.x : 0.f, h16_hi(xl.x) > 0.f ? g4.y : 0.f);
        v.y = pack_h16x2(h16_lo(xl.y) > 0.f ? g4.z : 0.f, h16_hi(xl.y) > 0.f ? g4.w : 0.f);
        *(uint2*)(a.dy_out + gbase + (long)p * TR_C + co4[ct]) = v;
      } else {
        v = *(const uint2*)(a.x0 + gbase + (long)p * TR_C + co4[ct]);
      }
      R[ct][pt] = v;
      uint2 nx = v;
      if (!BWD) { nx.x = relu_bf16x2(v.x); nx.y = relu_bf16x2(v.y); }
      const int ppi = pp0[pt] + 11;
      const int slot = co4[ct] >> 3;
      *(uint2*)(smem + ppi * TR_ROW + ((slot ^ (ppi & 15)) << 4) + (kc & 1) * 8) = nx;
    }
  }
  trunk_barrier();
  stamp(1);

#pragma unroll 1
  for (int L = 0; L < TR_LAYERS; ++L) {
    const unsigned char* bin = smem + (L & 1) * TR_BUF;
    unsigned char* bout = smem + ((L + 1) & 1) * TR_BUF;
    // the filters of the next layer replace this layer's as they are consumed (past the last layer: a harmless re-read of its own,
    // so that the count of operations in flight is the same in every layer)
    const bf16_t* const wnext = a.w[L + 1 < TR_LAYERS ? L + 1 : L] + wlane;
    f32x4_t acc[2][4];                    // written (not accumulated) by the first K-step
    float4 b4[2];
    uint2 mk[2][4];
    if (!BWD) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) b4[ct] = *(const float4*)(bias_s + L * TR_C + co4[ct]);
    } else {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) trw_load2_v(mk[ct][pt], a.mask[L] + gbase + (long)(pt * 16 + r) * TR_C + co4[ct]);
    }

    // pixel fragments of K-step s: tap (kh, kw) = row offset kh*10 + kw of the padded image, 32-channel quarter cq.
    // (ppl: pp0 made opaque once per layer -- left loop-invariant, the compiler hoists all 144 fragment addresses of a layer out
    // of the layer loop and spills them; recomputed next to the read they cost three VALU instructions each under the MFMAs)
    int ppl[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) { ppl[pt] = pp0[pt]; asm volatile("" : "+v"(ppl[pt])); }
    auto xread = [&](int s, bf16x8_t (&xf)[4]) __attribute__((always_inline)) {
      const int t = s >> 2, cq = s & 3;
      const int tapoff = (t / 3) * 10 + (t % 3);
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const int pp = ppl[pt] + tapoff;
        xf[pt] = *(const bf16x8_t*)(bin + pp * TR_ROW + (((cq * 4 + kc) ^ (pp & 15)) << 4));
      }
    };
    bf16x8_t xf[2][4];
    xread(0, xf[0]);
    trw_for<0, 36>([&](auto ic) __attribute__((always_inline)) {
      constexpr int s = decltype(ic)::value;
#if TRW_ABLATE & 1
      if constexpr (s + 1 < 2) xread(s + 1, xf[(s + 1) & 1]);
#else
      if constexpr (s + 1 < 36) xread(s + 1, xf[(s + 1) & 1]);       // one K-step ahead of its MFMAs
#endif
      trw_wait();
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
          if constexpr (s == 0) trw_mfma_a0(acc[ct][pt], WA[s][ct], xf[s & 1][pt]);
          else if constexpr (s < TRW_ASTEPS) trw_mfma_a(acc[ct][pt], WA[s][ct], xf[s & 1][pt]);
          else trw_mfma_v(acc[ct][pt], WV[s - TRW_ASTEPS][ct], xf[s & 1][pt]);
        }
#if !(TRW_ABLATE & 2)
      wload(ic, wnext);
#endif
    });

    stamp(2 + 2 * L);
    // ---- epilogue: bias / mask, residual on odd layers, 16-bit rounding; output to HBM, next layer's operand to LDS ----
    const bool odd = L & 1;
    bf16_t* outp = a.out[L] + gbase;
    if (BWD) trw_wait();                  // the masks were requested in front of this layer's 72 fragment loads
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        float v[4] = {acc[ct][pt][0], acc[ct][pt][1], acc[ct][pt][2], acc[ct][pt][3]};
        if (!BWD) {
          v[0] += b4[ct].x; v[1] += b4[ct].y; v[2] += b4[ct].z; v[3] += b4[ct].w;
        } else {
          const uint2 m = mk[ct][pt];
          if (!(h16_lo(m.x) > 0.f)) v[0] = 0.f;
          if (!(h16_hi(m.x) > 0.f)) v[1] = 0.f;
          if (!(h16_lo(m.y) > 0.f)) v[2] = 0.f;
          if (!(h16_hi(m.y) > 0.f)) v[3] = 0.f;
        }
        if (odd) { v[0] += h16_lo(R[ct][pt].x); v[1] += h16_hi(R[ct][pt].x); v[2] += h16_lo(R[ct][pt].y); v[3] += h16_hi(R[ct][pt].y); }
        uint2 pk;
        pk.x = pack_h16x2(v[0], v[1]);
        pk.y = pack_h16x2(v[2], v[3]);
        const int p = pt * 16 + r;
        *(uint2*)(outp + (long)p * TR_C + co4[ct]) = pk;
        if (odd) R[ct][pt] = pk;
        if (!BWD) { pk.x = relu_bf16x2(pk.x); pk.y = relu_bf16x2(pk.y); }
        const int ppi = pp0[pt] + 11;
        const int slot = co4[ct] >> 3;
        *(uint2*)(bout + ppi * TR_ROW + ((slot ^ (ppi & 15)) << 4) + (kc & 1) * 8) = pk;
      }
    trunk_barrier();
    stamp(3 + 2 * L);
  }
  if (!BWD && a.feat != nullptr) {
    // the pooled features: the last layer left relu(out[7]) in LDS as "the next layer's input" (image (TR_LAYERS & 1), interior pixels
    // of the padded 10 x 10 grid, 16-byte slots swizzled with the pixel index).  Thread (channel pair cp = tid & 63, pixel quarter
    // q = tid >> 6) sums 16 pixels; the quarters meet in LDS (the other image is dead).  Kept out of the layer loop on purpose.
    const unsigned char* fin = smem + (TR_LAYERS & 1) * TR_BUF;
    float* part = (float*)(smem + ((TR_LAYERS + 1) & 1) * TR_BUF);       // [4][128]
    const int cp = tid & 63, q = tid >> 6;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int p = q * 16 + i, pp = ((p >> 3) + 1) * 10 + (p & 7) + 1;
      const uint32_t v = *(const uint32_t*)(fin + pp * TR_ROW + ((((2 * cp) >> 3) ^ (pp & 15)) << 4) + ((2 * cp) & 7) * 2);
      s0 += h16_lo(v); s1 += h16_hi(v);
    }
    part[q * 128 + 2 * cp] = s0;
    part[q * 128 + 2 * cp + 1] = s1;
    __syncthreads();
    if (tid < 128)
      a.feat[img * TR_C + tid] = ((part[tid] + part[128 + tid]) + (part[256 + tid] + part[384 + tid])) * (1.f / 64.f);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the last layer's (unused) fragment loads must not outlive the registers
}

extern "C" {

// bytes of the fragment-major copy rcgan_dtrunk_prepare writes: the eight filters in forward order, then their data-gradient
// (rotated) forms in backward order
size_t rcgan_dtrunk_fragment_bytes(void) { return (size_t)2 * TR_LAYERS * TR_C * TR_K * sizeof(bf16_t); }

// prepared[i]: the prepared filter of layer i in FORWARD execution order (rcgan_conv_prepare layout: forward rows first,
// data-gradient rows behind them).  One launch; the result serves the forward and the backward pass of the same weights.
int rcgan_dtrunk_prepare(rcgan_ctx* ctx, const void* const* prepared, void* frag) {
  RC_REQUIRE(ctx, prepared && frag, "null argument");
  const size_t elems = (size_t)9 * TR_C * TR_C;
  TrunkFragArgs f;
  f.out = (bf16_t*)frag;
  for (int i = 0; i < TR_LAYERS; ++i) {
    RC_REQUIRE(ctx, prepared[i], "layer %d: null pointer", i);
    f.w[i] = (const bf16_t*)prepared[i];                                           // forward pass: layers first to last
    f.w[TR_LAYERS + i] = (const bf16_t*)prepared[TR_LAYERS - 1 - i] + elems;       // backward pass: last to first, rotated rows
  }
  hipLaunchKernelGGL(trunk_fragments_kernel, dim3(2 * TR_LAYERS * 4 * 36 * 2 * 64 / 256), dim3(256), 0, ctx->stream, f);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// x0 / outs / masks: [n][8][8][128] in the library's 16-bit activation format; frag: what rcgan_dtrunk_prepare wrote for the
// stage's CURRENT weights.
int rcgan_dtrunk(rcgan_ctx* ctx, int n, int backward, const void* x0, const void* frag, const float* const* bias,
                 const void* const* masks, void* const* outs) {
  return rcgan_dtrunk_pooled(ctx, n, backward, x0, frag, bias, masks, outs, nullptr, nullptr, nullptr);
}

// The same stage with the discriminator's relu + spatial mean at its boundary (TrunkArgs::feat).  Forward: feat (optional) receives
// the pooled features [n][128] fp32.  Backward: with feat (= the gradient of the pooled features) and xlast (= the stage's stored
// forward output outs[7]) the incoming gradient is formed inside the launch, written to dy_out, and x0 is not read (pass NULL).
int rcgan_dtrunk_pooled(rcgan_ctx* ctx, int n, int backward, const void* x0, const void* frag, const float* const* bias,
                        const void* const* masks, void* const* outs, float* feat, const void* xlast, void* dy_out) {
  RC_REQUIRE(ctx, n >= 1 && frag && outs, "bad arguments");
  RC_REQUIRE(ctx, ((backward & 1) && feat) ? (xlast != nullptr && dy_out != nullptr) : x0 != nullptr,
             "the stage needs its input (or, backward, feat + xlast + dy_out)");
  RC_REQUIRE(ctx, (backward & 1) ? masks != nullptr : true, "the backward pass needs the saved activations");
  TrunkArgs a;
  a.x0 = (const bf16_t*)x0;
  a.backward = backward & 1;
  a.stamps = (unsigned long long*)ctx->dbg_stamps;
  a.n = n;
  a.feat = feat;
  a.xlast = (const bf16_t*)xlast;
  a.dy_out = (bf16_t*)dy_out;
  a.gemm = SmallGemmArgs{};
  const size_t elems = (size_t)9 * TR_C * TR_C;
  for (int i = 0; i < TR_LAYERS; ++i) {
    RC_REQUIRE(ctx, outs[i], "layer %d: null pointer", i);
    a.w[i] = (const bf16_t*)frag + (size_t)((backward & 1) ? TR_LAYERS + i : i) * elems;
    a.bias[i] = (!(backward & 1) && bias) ? bias[i] : nullptr;
    a.mask[i] = (backward & 1) ? (const bf16_t*)masks[i] : nullptr;
    RC_REQUIRE(ctx, !(backward & 1) || a.mask[i], "layer %d: null mask", i);
    a.out[i] = (bf16_t*)outs[i];
  }
  static bool attr_set = false;
  if (!attr_set) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_trunk_rw_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRW_LDS));
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_trunk_rw_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRW_LDS));
    attr_set = true;
  }
  // the head's dE GEMM is taken (head_stage 1 -> 2) only here, behind every check that can fail: a rider marked as carried by a
  // launch that never happened would leave the parameter sums reading a dE nobody wrote
  const int riders = ((backward & 1) && head_take_gemm(ctx, &a.gemm)) ? cdiv(a.gemm.d, 16) : 0;
  if (backward & 1) hipLaunchKernelGGL((conv_trunk_rw_kernel<true>), dim3(n + riders), dim3(256), TRW_LDS, ctx->stream, a);
  else hipLaunchKernelGGL((conv_trunk_rw_kernel<false>), dim3(n), dim3(256), TRW_LDS, ctx->stream, a);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

}  // extern "C"
