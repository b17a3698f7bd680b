// The 8x8 stage of the discriminator as ONE launch: D.Block.3 .. D.Block.6 (gan_resnet.py:275-328, 398-404), four identity-
// shortcut residual blocks = eight 3x3 convolutions 128 -> 128 on 8 x 8 pixels, forward or data gradient.
//
// Why: one of these convolutions is 1.2 GFLOP per 64 images.  As its own launch it is a 64 x 64-tile kernel with a
// K-split over 16 wavefronts that takes 9-10 us (launch, pipeline fill, partial-sum exchange, drain) for ~1 us of matrix
// work per CU -- 8 launches forward and 8 backward in every discriminator pass, six passes per iteration.  There is no
// norm layer in D, so an image never meets another image: a workgroup can carry ONE image through all eight layers.
//   * activations never leave the CU: 64 pixels x 128 channels (16 KB) live in LDS as a zero-padded 10 x 10 image, one
//     256-byte row per pixel (3x3 taps = row offsets, no bounds tests), 16-byte slots XOR-swizzled with the pixel index
//     (a fragment read takes the same slot of 16 different pixels); two such images alternate between layers
//   * 4 wavefronts, one per SIMD, 32 output channels x all 64 pixels each: 8 accumulator tiles.  The filter rows are
//     private to a wavefront: each lane LDS-DMAs exactly the 16 bytes it will feed the MFMA with (lane-linear 1-KiB
//     deposits, read back with a conflict-free ds_read_b128 at lane*16), three taps (24 KiB per wavefront) ahead of their
//     use, in one continuous stream across the eight layers.  (Plain loads were sunk next to their use and waited for with
//     vmcnt(0) -- one L2 round trip per tap; asm loads into VGPRs got copied by the compiler before they had landed.)
//   * the epilogue keeps the residual operand in registers (same lane = same (pixel, channel) in every layer), writes the
//     layer's output to HBM once (the backward pass and the filter gradients need it) and the next layer's input to LDS
//   * one raw barrier per layer; nothing waits for HBM stores
// Forward layer 2b (conv1 of block b): h = conv(relu(x)) + bias;      layer 2b+1: x' = x + conv(relu(h)) + bias
// Backward runs the layers in reverse with the rotated filters: dh = dgrad2(dy) * (h > 0);  dx = dy + dgrad1(dh) * (x > 0).
#include "conv_mfma.h"
#include "mfma_util.h"

namespace {

constexpr int TR_C = 128;                 // channels
constexpr int TR_K = 9 * TR_C;            // reduction length of one layer
constexpr int TR_ROW = TR_C * 2;          // bytes of one padded pixel (all channels)
constexpr int TR_BUF = 100 * TR_ROW;      // one zero-padded 10 x 10 activation image
constexpr int TR_LAYERS = 8;
constexpr int TR_TAPB = 8 * 1024;         // filter bytes of one tap for one wavefront (32 channels x 128 x 2 B)
constexpr int TR_RING = 3;                // taps in flight per wavefront
constexpr int TR_LDS = 2 * TR_BUF + 4 * TR_RING * TR_TAPB + TR_LAYERS * TR_C * 4;   // images + filter rings + biases

struct TrunkArgs {
  const bf16_t* x0;                       // [n][64][128]: trunk input (forward) / gradient of the trunk output (backward)
  const bf16_t* w[TR_LAYERS];             // filter rows [128][1152] of the layers in execution order
  const float* bias[TR_LAYERS];           // forward (null: none)
  const bf16_t* mask[TR_LAYERS];          // backward: saved forward activation whose sign gates the layer's output
  bf16_t* out[TR_LAYERS];                 // [n][64][128] per layer
  int backward;
};

__device__ __forceinline__ void trunk_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the 16 DMA pieces of the two younger taps may stay in flight (anything else this wavefront issued later only makes the
// wait conservative: vmcnt counts every outstanding memory operation)
__device__ __forceinline__ void wait_ring() {
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace

// ABL (diagnostics, wrong results): 1 = no filter DMA after the prologue, 2 = pixel fragments read once per tap instead of per
// 32-channel quarter, 3 = no MFMAs, 4 = no filter fragment reads from LDS (one fragment reused)
template <bool BWD, int ABL>
__global__ __launch_bounds__(256) void conv_trunk_kernel(TrunkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kc = lane >> 4;
  const long img = blockIdx.x;

  for (int i = tid; i < 2 * TR_BUF / 16; i += 256) ((uint4*)smem)[i] = make_uint4(0u, 0u, 0u, 0u);
  if (!BWD)
    for (int i = tid; i < TR_LAYERS * TR_C; i += 256) {
      const float* bp = a.bias[i >> 7];
      ((float*)(smem + 2 * TR_BUF + 4 * TR_RING * TR_TAPB))[i] = bp ? bp[i & 127] : 0.f;
    }

  // pixel tile pt: pixel p = pt*16 + r = (oh, ow); padded index of the pixel a tap (kh, kw) reads = pp0 + kh*10 + kw
  int pp0[4];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    const int p = pt * 16 + r;
    pp0[pt] = (p >> 3) * 10 + (p & 7);
  }
  // accumulator / epilogue layout: lane holds pixel p (same r), channels co4 .. co4+3 of channel tile ct
  const int co4[2] = {wave * 32 + 4 * kc, wave * 32 + 16 + 4 * kc};
  // this lane's filter rows: channel wave*32 + ct*16 + r, reduction offset kc*8 (MFMA A operand)
  const long wrow[2] = {(long)(wave * 32 + r) * TR_K + kc * 8, (long)(wave * 32 + 16 + r) * TR_K + kc * 8};

  // filter ring of this wavefront: [slot][32-channel quarter of the tap][channel tile] x 1 KiB (lane l owns bytes l*16..)
  unsigned char* const ringp = smem + 2 * TR_BUF + wave * (TR_RING * TR_TAPB);
  const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ringp;
  float* const bias_s = (float*)(smem + 2 * TR_BUF + 4 * TR_RING * TR_TAPB);      // [layer][128]
  auto load_tap = [&](int slot, int F) __attribute__((always_inline)) {
    const int L = F / 9, t = F - L * 9;
    const bf16_t* base = a.w[L] + t * TR_C;
#pragma unroll
    for (int cq = 0; cq < 4; ++cq)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) glds16_asm(base + wrow[ct] + cq * 32, ring_lds + slot * TR_TAPB + (cq * 2 + ct) * 1024);
  };
  load_tap(0, 0);
  load_tap(1, 1);
  load_tap(2, 2);

  // residual operand (block input / block output gradient), bf16-rounded values held as floats
  float R[2][4][4];
  const long gbase = img * 64 * TR_C;
  unsigned char* buf0 = smem;
  __syncthreads();                        // the zero fill is complete before the first interior write
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const int p = pt * 16 + r;
      const uint2 v = *(const uint2*)(a.x0 + gbase + (long)p * TR_C + co4[ct]);
      R[ct][pt][0] = h16_lo(v.x); R[ct][pt][1] = h16_hi(v.x); R[ct][pt][2] = h16_lo(v.y); R[ct][pt][3] = h16_hi(v.y);
      uint2 nx = v;
      if (!BWD) { nx.x = relu_bf16x2(v.x); nx.y = relu_bf16x2(v.y); }
      const int ppi = pp0[pt] + 11;
      const int slot = co4[ct] >> 3;
      *(uint2*)(buf0 + ppi * TR_ROW + ((slot ^ (ppi & 15)) << 4) + (kc & 1) * 8) = nx;
    }
  trunk_barrier();

#pragma unroll 1
  for (int L = 0; L < TR_LAYERS; ++L) {
    const unsigned char* bin = smem + (L & 1) * TR_BUF;
    unsigned char* bout = smem + ((L + 1) & 1) * TR_BUF;
    f32x4_t acc[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // epilogue operands requested up front (they arrive under the K loop)
    float4 b4[2];
    uint2 mk[2][4];
    if (!BWD) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) b4[ct] = *(const float4*)(bias_s + L * TR_C + co4[ct]);
    } else {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) mk[ct][pt] = *(const uint2*)(a.mask[L] + gbase + (long)(pt * 16 + r) * TR_C + co4[ct]);
    }

#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {    // ring slot = kw (the flat tap index advances by 3 per kh row)
        const int tapoff = kh * 10 + kw;
        int rowb[4], sw[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) { const int pp = pp0[pt] + tapoff; rowb[pt] = pp * TR_ROW; sw[pt] = pp & 15; }
        if (ABL != 1) wait_ring();        // this tap's eight fragments have landed
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
          bf16x8_t xf[4];
#pragma unroll
          for (int pt = 0; pt < 4; ++pt) xf[pt] = *(const bf16x8_t*)(bin + rowb[pt] + ((((ABL == 2 ? 0 : cq) * 4 + kc) ^ sw[pt]) << 4));
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            const bf16x8_t wf = *(const bf16x8_t*)(ringp + (ABL == 4 ? 0 : kw * TR_TAPB + (cq * 2 + ct) * 1024) + lane * 16);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
              if (ABL == 3) { acc[ct][pt][0] += __builtin_bit_cast(float, (int)wf[0] + (int)xf[pt][1]); }
              else acc[ct][pt] = mfma16(wf, xf[pt], acc[ct][pt]);
            }
          }
        }
        // refill the slot with the tap three ahead (past the last layer: a harmless re-read of the last tap, so that the
        // count of loads in flight is the same everywhere); the MFMAs above have read the slot
        __builtin_amdgcn_sched_barrier(0);
        if (ABL != 1) load_tap(kw, min(L * 9 + kh * 3 + kw + 3, TR_LAYERS * 9 - 1));
      }
    }

    // ---- epilogue: bias / mask, residual on odd layers, 16-bit rounding; output to HBM, next layer's operand to LDS ----
    const bool odd = L & 1;
    bf16_t* outp = a.out[L] + gbase;
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
        if (odd) { v[0] += R[ct][pt][0]; v[1] += R[ct][pt][1]; v[2] += R[ct][pt][2]; v[3] += R[ct][pt][3]; }
        uint2 pk;
        pk.x = pack_h16x2(v[0], v[1]);
        pk.y = pack_h16x2(v[2], v[3]);
        const int p = pt * 16 + r;
        *(uint2*)(outp + (long)p * TR_C + co4[ct]) = pk;
        if (odd) { R[ct][pt][0] = h16_lo(pk.x); R[ct][pt][1] = h16_hi(pk.x); R[ct][pt][2] = h16_lo(pk.y); R[ct][pt][3] = h16_hi(pk.y); }
        if (!BWD) { pk.x = relu_bf16x2(pk.x); pk.y = relu_bf16x2(pk.y); }
        const int ppi = pp0[pt] + 11;
        const int slot = co4[ct] >> 3;
        *(uint2*)(bout + ppi * TR_ROW + ((slot ^ (ppi & 15)) << 4) + (kc & 1) * 8) = pk;
      }
    trunk_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

extern "C" {

// x0 / outs / masks: [n][8][8][128] in the library's 16-bit activation format; prepared[i]: the prepared filter of layer i
// (rcgan_conv_prepare layout: forward rows first, data-gradient rows behind them).
int rcgan_dtrunk(rcgan_ctx* ctx, int n, int backward, const void* x0, const void* const* prepared, const float* const* bias,
                 const void* const* masks, void* const* outs) {
  RC_REQUIRE(ctx, n >= 1 && x0 && prepared && outs, "bad arguments");
  RC_REQUIRE(ctx, (backward & 1) ? masks != nullptr : true, "the backward pass needs the saved activations");
  TrunkArgs a;
  a.x0 = (const bf16_t*)x0;
  a.backward = backward & 1;
  const size_t elems = (size_t)9 * TR_C * TR_C;
  for (int i = 0; i < TR_LAYERS; ++i) {
    RC_REQUIRE(ctx, prepared[i] && outs[i], "layer %d: null pointer", i);
    a.w[i] = (const bf16_t*)prepared[i] + ((backward & 1) ? elems : 0);
    a.bias[i] = (!(backward & 1) && bias) ? bias[i] : nullptr;
    a.mask[i] = (backward & 1) ? (const bf16_t*)masks[i] : nullptr;
    RC_REQUIRE(ctx, !(backward & 1) || a.mask[i], "layer %d: null mask", i);
    a.out[i] = (bf16_t*)outs[i];
  }
  static bool attr_set = false;
  const size_t lds = TR_LDS;
  if (!attr_set) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_trunk_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_trunk_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_trunk_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_trunk_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_trunk_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_trunk_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int abl = (backward >> 4) & 7;      // diagnostics only (scripts/bench_trunk.py)
  if (backward & 1) hipLaunchKernelGGL((conv_trunk_kernel<true, 0>), dim3(n), dim3(256), lds, ctx->stream, a);
  else if (abl == 1) hipLaunchKernelGGL((conv_trunk_kernel<false, 1>), dim3(n), dim3(256), lds, ctx->stream, a);
  else if (abl == 2) hipLaunchKernelGGL((conv_trunk_kernel<false, 2>), dim3(n), dim3(256), lds, ctx->stream, a);
  else if (abl == 3) hipLaunchKernelGGL((conv_trunk_kernel<false, 3>), dim3(n), dim3(256), lds, ctx->stream, a);
  else if (abl == 4) hipLaunchKernelGGL((conv_trunk_kernel<false, 4>), dim3(n), dim3(256), lds, ctx->stream, a);
  else hipLaunchKernelGGL((conv_trunk_kernel<false, 0>), dim3(n), dim3(256), lds, ctx->stream, a);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

}  // extern "C"
