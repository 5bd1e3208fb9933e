// mask_head.hip -- CondInst-style dynamic mask head, fused.
//
// Reference op chain (projects/SeqFormer/seqformer/models/segmentation_condInst.py):
//   :425-456  build, per instance, a 10-channel map = relative coordinates (2) + the frame's 8
//             mask features, materialised for ALL instances ([1, n*10, H*W]: 46 MB per 360p
//             frame at n=300);
//   :404-422  three grouped 1x1 convolutions (groups = n): 10->8, ReLU, 8->8, ReLU, 8->1, with the
//             169 per-instance parameters split as [w0(80) w1(64) w2(8) b0(8) b1(8) b2(1)] (:614-637);
//   :640-662  "aligned bilinear" x2 up-sampling (replicate pad, align_corners interpolate, pad, crop).
// Here it is one kernel: nothing but the [n, 2H, 2W] logits is ever written.
//
// Mapping: one wave per (instance, strip of 63 columns x R rows).  The instance is wave-uniform:
// its 169 parameters are fetched with three vector loads, kept across the lanes and broadcast
// into SGPRs with v_readlane as the three layers (v_pk_fma_f32 chains) consume them; the 8 features of a pixel are 8 coalesced loads; relative
// coordinates are computed, never stored.  The up-sampling needs each pixel's left / upper
// neighbours: the strip's rows (and one halo row above) are all held in registers, the upper
// neighbour is the previous row's register and the left neighbour comes from the lane below with
// one DPP move; lane 0 and the first row are a one-pixel halo (computed, not stored), so no LDS
// and no barrier is used at all.
// out[2y  ][2x] = (a+b+c+d)/4   out[2y  ][2x+1] = (b+d)/2      a = in[y-1][x-1]  b = in[y-1][x]
// out[2y+1][2x] = (c+d)/2       out[2y+1][2x+1] = d            c = in[y  ][x-1]  d = in[y  ][x]
// with indices clamped at 0 -- the closed form of the reference's pad/interpolate/pad/crop.
// Bound: the output write (n * 4HW * 4 B; 18.4 MB per 360p frame at n = 300) -> HBM roofline.
#include "vnx_common.h"

namespace vnx {

constexpr int kMhChannels = 8;    // mask feature channels (hidden_dim / 32)
constexpr int kMhHidden = 8;      // dynamic_mask_channels
constexpr int kMhParams = (kMhChannels + 2) * kMhHidden + kMhHidden * kMhHidden + kMhHidden + kMhHidden + kMhHidden + 1;
constexpr int kMhStripW = 63;     // stored columns per wave (64 lanes - 1 halo lane)
constexpr int kMhStripH = 5;      // stored rows per wave (+1 halo row, all held in registers)
static_assert(kMhParams == 169, "parameter vector layout");

typedef float float2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float lane_below(float v) {
  // lane l receives lane l-1 (wave_shr:1); lane 0 keeps its own value, which is never used
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v),
                                                                0x138, 0xF, 0xF, false));
}

__global__ void __launch_bounds__(256)
dynamic_mask_head_kernel(const float* __restrict__ feats, const float* __restrict__ ref,
                         const float* __restrict__ params, const int* __restrict__ inst_image,
                         float* __restrict__ out, int H, int W, int n_inst, int stride,
                         int strips_x, int strips_y) {
  const int lane = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t wave_id = int64_t(blockIdx.x) * 4 + wave_in_block;
  const int strips = strips_x * strips_y;
  const int j = int(wave_id / strips);  // instance
  if (j >= n_inst) return;
  const int s = int(wave_id - int64_t(j) * strips);
  const int sy = s / strips_x, sx = s - sy * strips_x;
  const int x = sx * kMhStripW - 1 + lane;          // lane 0 is the left halo
  const int xc = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
  const int y0 = sy * kMhStripH;

  // The instance's 169 parameters are wave-uniform: scalar loads, SGPR operands of the packed
  // FMAs.  Alternatives measured on MI355X at 300 instances x 48x80 (tools/mask_head_scaling.py):
  // lane-parked copy + v_readlane 24.0 us (617 readlanes + 221 s_nops per wave), LDS broadcast
  // reads 31.6 us (the scheduler hoists all 169 reads: 256 VGPRs or spills), this form 24-26 us.
  // The kernel is VALU-issue bound (~1300 issued instructions per wave at ~4 clk each), not
  // memory bound; the HBM roofline (18.4 MB of logits, ~3 us) is 8x away.
  const float* P = params + int64_t(j) * kMhParams;
  auto param = [&](int k) -> float { return P[k]; };
  constexpr int W0 = 0, W1 = 80, W2 = 144, B0 = 152, B1 = 160, B2 = 168;
  const float refx = ref[2 * j], refy = ref[2 * j + 1];
  const int img = inst_image[j];
  const float* F = feats + int64_t(img) * kMhChannels * H * W;
  const float half = float(stride / 2);
  const float relx = refx - (float(xc * stride) + half);

  float* O = out + int64_t(j) * (2 * H) * (2 * W);

  // All rows of the strip (plus the halo row above it) live in registers, so every one of the
  // 169 parameters is fetched (scalar load) exactly once per wave and the 8 x (R+1) feature
  // loads are all in flight together.  Rows are processed in PAIRS held as float2 so the layer
  // arithmetic compiles to v_pk_fma_f32 (two FMAs per issued instruction): the kernel is
  // VALU-issue bound (PMC: SQ_ACTIVE_INST_VALU ~ 80 % of the kernel), not memory bound.
  constexpr int R1 = kMhStripH + 1;        // rows computed
  static_assert(R1 % 2 == 0, "rows are processed in pairs");
  constexpr int RP = R1 / 2;
  float2_t x0[RP][kMhChannels + 2];
#pragma unroll
  for (int p = 0; p < RP; ++p) {
    float v[2][kMhChannels + 2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int y = y0 + 2 * p + h - 1;
      const int yc = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
      v[h][0] = relx;
      v[h][1] = refy - (float(yc * stride) + half);
#pragma unroll
      for (int c = 0; c < kMhChannels; ++c) v[h][2 + c] = F[(int64_t(c) * H + yc) * W + xc];
    }
#pragma unroll
    for (int i = 0; i < kMhChannels + 2; ++i) x0[p][i] = float2_t{v[0][i], v[1][i]};
  }
  const float2_t zero2 = {0.f, 0.f};
  float2_t x1[RP][kMhHidden];
#pragma unroll
  for (int o = 0; o < kMhHidden; ++o) {
    float w[kMhChannels + 2];
#pragma unroll
    for (int i = 0; i < kMhChannels + 2; ++i) w[i] = param(W0 + o * (kMhChannels + 2) + i);
    const float bias = param(B0 + o);
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      float2_t a = {bias, bias};
#pragma unroll
      for (int i = 0; i < kMhChannels + 2; ++i) a = __builtin_elementwise_fma(float2_t{w[i], w[i]}, x0[p][i], a);
      x1[p][o] = __builtin_elementwise_max(a, zero2);
    }
  }
  float2_t x2[RP][kMhHidden];
#pragma unroll
  for (int o = 0; o < kMhHidden; ++o) {
    float w[kMhHidden];
#pragma unroll
    for (int i = 0; i < kMhHidden; ++i) w[i] = param(W1 + o * kMhHidden + i);
    const float bias = param(B1 + o);
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      float2_t a = {bias, bias};
#pragma unroll
      for (int i = 0; i < kMhHidden; ++i) a = __builtin_elementwise_fma(float2_t{w[i], w[i]}, x1[p][i], a);
      x2[p][o] = __builtin_elementwise_max(a, zero2);
    }
  }
  float logit[R1];
  {
    float w[kMhHidden];
#pragma unroll
    for (int i = 0; i < kMhHidden; ++i) w[i] = param(W2 + i);
    const float bias = param(B2);
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      float2_t a = {bias, bias};
#pragma unroll
      for (int i = 0; i < kMhHidden; ++i) a = __builtin_elementwise_fma(float2_t{w[i], w[i]}, x2[p][i], a);
      logit[2 * p] = a.x;
      logit[2 * p + 1] = a.y;
    }
  }
#pragma unroll
  for (int r = 1; r < R1; ++r) {
    const int y = y0 + r - 1;
    const float cur = logit[r], prev = logit[r - 1];
    const float left = lane_below(cur), prev_left = lane_below(prev);
    if (y < H && lane > 0 && x < W) {
      const float2_t top = {0.25f * ((prev_left + prev) + (left + cur)), 0.5f * (prev + cur)};
      const float2_t bot = {0.5f * (left + cur), cur};
      *reinterpret_cast<float2_t*>(O + int64_t(2 * y) * (2 * W) + 2 * x) = top;
      *reinterpret_cast<float2_t*>(O + int64_t(2 * y + 1) * (2 * W) + 2 * x) = bot;
    }
  }
}

}  // namespace vnx

using namespace vnx;

extern "C" int vnx_dynamic_mask_head_forward(int dtype, const void* mask_feats,
                                             const void* reference_points, const void* params,
                                             const int32_t* inst_image, void* out, int num_images,
                                             int channels, int height, int width, int num_insts,
                                             int num_params, int stride, void* hip_stream) {
  if (dtype != VNX_F32) {
    set_error("vnx_dynamic_mask_head_forward: only f32 is built (got dtype %d)", dtype);
    return VNX_ERR_UNSUPPORTED;
  }
  if (channels != kMhChannels || num_params != kMhParams) {
    set_error("vnx_dynamic_mask_head_forward: built for %d feature channels / %d parameters "
              "(hidden_dim 256, dynamic_mask_channels 8, rel_coord); got %d / %d",
              kMhChannels, kMhParams, channels, num_params);
    return VNX_ERR_UNSUPPORTED;
  }
  if (num_images < 0 || height < 0 || width < 0 || num_insts < 0 || stride <= 0) {
    set_error("vnx_dynamic_mask_head_forward: bad sizes");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  if (num_insts == 0 || height == 0 || width == 0) return VNX_OK;
  if (!mask_feats || !reference_points || !params || !inst_image || !out) {
    set_error("vnx_dynamic_mask_head_forward: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const int strips_x = (width + kMhStripW - 1) / kMhStripW;
  const int strips_y = (height + kMhStripH - 1) / kMhStripH;
  const int64_t waves = int64_t(num_insts) * strips_x * strips_y;
  const int64_t blocks = (waves + 3) / 4;
  if (blocks >= (int64_t(1) << 31)) {
    set_error("vnx_dynamic_mask_head_forward: %lld workgroups exceed the grid limit", (long long)blocks);
    return VNX_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(dynamic_mask_head_kernel, dim3(uint32_t(blocks)), dim3(256), 0,
                     (hipStream_t)hip_stream, (const float*)mask_feats, (const float*)reference_points,
                     (const float*)params, (const int*)inst_image, (float*)out, height, width,
                     num_insts, stride, strips_x, strips_y);
  return check_launch("dynamic_mask_head");
}
