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
// Forward (dynamic_mask_head_runs_kernel): a wave takes a run of whole image rows of one instance in row-major order --
// every lane a pixel, 64 consecutive pixels per register -- and walks it in chunks of up to three register pairs.  The
// instance is wave-uniform: its 169 parameters stream through the scalar file in groups of two output channels and feed
// v_pk_fma_f32 chains; the 8 features of a pixel are 8 coalesced loads, prefetched a chunk ahead; relative coordinates are
// computed, never stored.  The up-sampling needs each pixel's left / upper neighbours: the wave keeps its logits (and the
// last row of the chunk before) in its own piece of LDS and reads them there; a run that does not start at the top of
// the frame computes the row above it first.  (Rounds 1 - 3: strips of 63 columns x 5 rows per wave, neighbours by DPP --
// kept in the development build, variant 799.)
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
#ifdef VNX_DEV_VARIANTS
#ifndef VNX_MH_ROWS
#define VNX_MH_ROWS 5
#endif
constexpr int kMhStripH = VNX_MH_ROWS;   // strip kernel: stored rows per wave (+1 halo row, all held in registers)
#endif
static_assert(kMhParams == 169, "parameter vector layout");

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t sgpr2_t __attribute__((ext_vector_type(2)));
typedef unsigned int store2_t __attribute__((__vector_size__(2 * sizeof(unsigned int))));
typedef uint32_t sgpr4_t __attribute__((ext_vector_type(4)));
typedef uint32_t sgpr8_t __attribute__((ext_vector_type(8)));
typedef uint32_t sgpr16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float lane_below(float v) {
  // lane l receives lane l-1 (wave_shr:1); lane 0 keeps its own value, which is never used
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v),
                                                                0x138, 0xF, 0xF, false));
}

// d += w * x for a row pair, w = the low (HI = false) or high half of an SGPR pair broadcast to both rows.
// Written as asm: given `float2{w, w}` hipcc copies every weight into a fresh SGPR pair and, three row pairs
// later, spills those copies to VGPR lanes (536 v_readlane + 262 v_writelane beside 440 FMAs per wave).
template <bool HI>
__device__ __forceinline__ void pk_fma_bcast(float2_t& d, uint64_t w_pair, float2_t x) {
  if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(d) : "s"(w_pair), "v"(x));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(d) : "s"(w_pair), "v"(x));
}

// d = w * x + c: the first FMA of an accumulator takes the bias from a register pair shared by all row pairs of an
// output channel -- one move per channel instead of one per (channel, row pair): 51 -> 17 moves per wave.
__device__ __forceinline__ float2_t pk_fma_first(uint64_t w_pair, float2_t x, float2_t c) {
  float2_t d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "s"(w_pair), "v"(x), "v"(c));
  return d;
}

// ReLU of a row pair.  asm: `__builtin_elementwise_max(a, 0)` on a value that comes out of an asm statement makes hipcc
// canonicalise it first (v_max_f32 x, x, x before v_max_f32 x, 0, x) -- 96 of a wave's 868 vector instructions.
__device__ __forceinline__ float2_t relu2(float2_t a) {
  float2_t r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r.x) : "v"(a.x));
  asm("v_max_f32 %0, 0, %1" : "=v"(r.y) : "v"(a.y));
  return r;
}

// The three layers (10 -> 8, ReLU, 8 -> 8, ReLU, 8 -> 1) of a wave's pixels.  x0[p][i] = input channel i (relative x, relative
// y, the 8 features) of register pair p; logit[2p], logit[2p + 1] = the pair's outputs.  P = the instance's 169 parameters.
//
// The parameters are wave-uniform: scalar loads, SGPR operands of the packed FMAs (v_pk_fma_f32 takes one SGPR pair and
// broadcasts either half with op_sel).  What decides the speed is HOW MANY of them are live: left to itself hipcc hoists
// all eleven s_load_dwordx16 to the top, 169 values meet ~100 SGPRs, and the rest lives in VGPR lanes -- 655 v_readlane +
// 371 v_writelane beside 432 v_pk_fma_f32 per wave (27 us per 360p frame at 300 instances).  Here the parameters arrive
// in groups of two output channels (asm s_load, at most two groups in flight, waited for by hand), so none of them ever
// leaves the scalar file: 14.6 us.  Other forms measured on MI355X (tools/time_heads.py): the parameters parked across the
// lanes of three VGPRs and read back pair by pair with v_readlane where consumed (no scalar loads inside the layers, 169
// more instructions per wave) 15.4 us; LDS broadcast reads 31.6 us.
// after_layer1(): called once the inputs have been consumed (the runs kernel issues its next feature loads there).
template <int RP, typename Mid>
__device__ __forceinline__ void mask_logits(const float* P, const float2_t (&x0)[RP][kMhChannels + 2], float (&logit)[2 * RP],
                                            Mid&& after_layer1) {
  constexpr int W0 = 0, W1 = 80, W2 = 144, B0 = 152, B1 = 160, B2 = 168;
  // a group of parameters in the scalar file: 20 (or 16) weights of two output channels + their two biases
  struct Group { sgpr16_t w; sgpr4_t w2; sgpr2_t b; };
  auto fetch20 = [&](Group& g, int w_at, int b_at) {   // 20 consecutive weights, 2 consecutive biases
    asm volatile("s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx4 %1, %3, %5\n\ts_load_dwordx2 %2, %3, %6"
                 : "=&s"(g.w), "=&s"(g.w2), "=&s"(g.b)
                 : "s"(P), "n"(w_at * 4), "n"(w_at * 4 + 64), "n"(b_at * 4)
                 : "memory");
  };
  auto fetch16 = [&](Group& g, int w_at, int b_at) {   // 16 consecutive weights, 2 consecutive biases
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx2 %1, %2, %4"
                 : "=&s"(g.w), "=&s"(g.b)
                 : "s"(P), "n"(w_at * 4), "n"(b_at * 4)
                 : "memory");
  };
  auto landed = [&](Group& g) {   // everything issued so far is in its registers (the loads are invisible to hipcc's own counting)
#if defined(VNX_MH_ABL) && (VNX_MH_ABL & 8)     // timing ablation: the parameter groups are not waited for
    asm volatile("" : "+s"(g.w), "+s"(g.w2), "+s"(g.b)::"memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(g.w), "+s"(g.w2), "+s"(g.b)::"memory");
#endif
  };
  auto wpair = [](const Group& g, int k) -> uint64_t {   // weights 2k, 2k + 1 of the group as one SGPR pair
    return k < 8 ? (uint64_t(g.w[2 * k + 1]) << 32) | g.w[2 * k] : (uint64_t(g.w2[2 * (k - 8) + 1]) << 32) | g.w2[2 * (k - 8)];
  };

  float2_t x1[RP][kMhHidden];
  {
    Group ga, gb;
    ga.w2 = sgpr4_t{0u, 0u, 0u, 0u}; gb.w2 = ga.w2;
#define VNX_L1(G, o0)                                                                                          \
    {   /* the group's two output channels together: 2 x RP independent accumulators -- a v_pk_fma_f32 that reads the */ \
        /* result of one less than four instructions before it costs a wait state (hipcc fills in s_nop)            */ \
      float2_t a[2][RP];                                                                                         \
      float2_t bias2[2];                                                                                         \
      _Pragma("unroll") for (int oo = 0; oo < 2; ++oo) {                                                         \
        const float bias = __uint_as_float(G.b[oo]);                                                             \
        bias2[oo] = float2_t{bias, bias};                                                                        \
      }                                                                                                          \
      _Pragma("unroll") for (int i = 0; i < kMhChannels + 2; i += 2) {                                           \
        _Pragma("unroll") for (int oo = 0; oo < 2; ++oo) {                                                       \
          const uint64_t w2 = wpair(G, (oo * (kMhChannels + 2) + i) / 2);                                        \
          _Pragma("unroll") for (int p = 0; p < RP; ++p) {                                                       \
            if (i == 0) a[oo][p] = pk_fma_first(w2, x0[p][i], bias2[oo]);                                        \
            else pk_fma_bcast<false>(a[oo][p], w2, x0[p][i]);                                                    \
          }                                                                                                      \
        }                                                                                                        \
        _Pragma("unroll") for (int oo = 0; oo < 2; ++oo) {                                                       \
          const uint64_t w2 = wpair(G, (oo * (kMhChannels + 2) + i) / 2);                                        \
          _Pragma("unroll") for (int p = 0; p < RP; ++p) pk_fma_bcast<true>(a[oo][p], w2, x0[p][i + 1]);         \
        }                                                                                                        \
      }                                                                                                          \
      _Pragma("unroll") for (int oo = 0; oo < 2; ++oo) {                                                         \
        _Pragma("unroll") for (int p = 0; p < RP; ++p) x1[p][(o0) + oo] = relu2(a[oo][p]);                       \
      }                                                                                                          \
    }
    // (sched_barrier: without it the machine scheduler sinks every FMA block below the last fetch -- the asm
    //  statements only order each other -- and all 169 values are live at once again)
#define VNX_FENCE __builtin_amdgcn_sched_barrier(0);
    fetch20(ga, W0, B0);
    landed(ga); fetch20(gb, W0 + 20, B0 + 2); VNX_FENCE
    VNX_L1(ga, 0) VNX_FENCE
    landed(gb); fetch20(ga, W0 + 40, B0 + 4); VNX_FENCE
    VNX_L1(gb, 2) VNX_FENCE
    landed(ga); fetch20(gb, W0 + 60, B0 + 6); VNX_FENCE
    VNX_L1(ga, 4) VNX_FENCE
    landed(gb); VNX_FENCE
    VNX_L1(gb, 6) VNX_FENCE
#undef VNX_L1
  }
  after_layer1();
  __builtin_amdgcn_sched_barrier(0);
  // Layer two, with layer three accumulated as its channels appear (logit = b3 + sum_k w3[k] relu(x2[k]) in the order
  // k = 0 .. 7 either way): only two channels of x2 are ever live, which leaves the registers of the inputs free for the
  // loads issued in after_layer1().
  {
    Group ga, gb;
    ga.w2 = sgpr4_t{0u, 0u, 0u, 0u}; gb.w2 = ga.w2;
    sgpr8_t w3; uint32_t b3;
    float2_t l3[RP];
#define VNX_L2(G, o0)                                                                                          \
    {                                                                                                            \
      float2_t a[2][RP];                                                                                         \
      float2_t bias2[2];                                                                                         \
      _Pragma("unroll") for (int oo = 0; oo < 2; ++oo) {                                                         \
        const float bias = __uint_as_float(G.b[oo]);                                                             \
        bias2[oo] = float2_t{bias, bias};                                                                        \
      }                                                                                                          \
      _Pragma("unroll") for (int i = 0; i < kMhHidden; i += 2) {                                                 \
        _Pragma("unroll") for (int oo = 0; oo < 2; ++oo) {                                                       \
          const uint64_t w2 = wpair(G, (oo * kMhHidden + i) / 2);                                                \
          _Pragma("unroll") for (int p = 0; p < RP; ++p) {                                                       \
            if (i == 0) a[oo][p] = pk_fma_first(w2, x1[p][i], bias2[oo]);                                        \
            else pk_fma_bcast<false>(a[oo][p], w2, x1[p][i]);                                                    \
          }                                                                                                      \
        }                                                                                                        \
        _Pragma("unroll") for (int oo = 0; oo < 2; ++oo) {                                                       \
          const uint64_t w2 = wpair(G, (oo * kMhHidden + i) / 2);                                                \
          _Pragma("unroll") for (int p = 0; p < RP; ++p) pk_fma_bcast<true>(a[oo][p], w2, x1[p][i + 1]);         \
        }                                                                                                        \
      }                                                                                                          \
      const uint64_t w3pair = (uint64_t(w3[(o0) + 1]) << 32) | w3[(o0)];                                         \
      _Pragma("unroll") for (int p = 0; p < RP; ++p) {                                                           \
        const float2_t xa = relu2(a[0][p]);                                                                      \
        if ((o0) == 0) l3[p] = pk_fma_first(w3pair, xa, bias3);                                                  \
        else pk_fma_bcast<false>(l3[p], w3pair, xa);                                                             \
      }                                                                                                          \
      _Pragma("unroll") for (int p = 0; p < RP; ++p) pk_fma_bcast<true>(l3[p], w3pair, relu2(a[1][p]));          \
    }
#define VNX_FENCE __builtin_amdgcn_sched_barrier(0);
    fetch16(ga, W1, B1);
    // the last layer's 8 weights + bias (W2 .. B2 are NOT contiguous: 144..151 and 168)
    asm volatile("s_load_dwordx8 %0, %2, %3\n\ts_load_dword %1, %2, %4" : "=&s"(w3), "=&s"(b3) : "s"(P), "n"(W2 * 4), "n"(B2 * 4) : "memory");
    landed(ga);
    asm volatile("" : "+s"(w3), "+s"(b3)::"memory");      // (landed() waited for everything issued)
    fetch16(gb, W1 + 16, B1 + 2); VNX_FENCE
    const float bias3f = __uint_as_float(b3);
    const float2_t bias3 = {bias3f, bias3f};
    VNX_L2(ga, 0) VNX_FENCE
    landed(gb); fetch16(ga, W1 + 32, B1 + 4); VNX_FENCE
    VNX_L2(gb, 2) VNX_FENCE
    landed(ga); fetch16(gb, W1 + 48, B1 + 6); VNX_FENCE
    VNX_L2(ga, 4) VNX_FENCE
    landed(gb); VNX_FENCE
    VNX_L2(gb, 6) VNX_FENCE
#undef VNX_L2
#undef VNX_FENCE
#pragma unroll
    for (int p = 0; p < RP; ++p) { logit[2 * p] = l3[p].x; logit[2 * p + 1] = l3[p].y; }
  }
}

#ifdef VNX_DEV_VARIANTS     // the strip kernel of rounds 1 - 3: A/B reference of the development build (variant 799)
// HALVES = 1: the wave is one strip of 63 columns (+ halo lane 0).  HALVES = 2: two strips of 31 columns
// (+ halo lanes 0 and 32) stacked vertically -- lanes 32..63 take the kMhStripH rows below those of lanes
// 0..31.  The launcher picks whichever wastes fewer lanes on the frame's width: at 80 columns (360p) one
// 63-wide strip pair leaves 46 of 126 lanes idle, three 31-wide ones 13 of 93.
template <int HALVES>
__global__ void __launch_bounds__(256)      // (256, 5) -- 96 VGPRs, 4 500 waves of a 360p frame resident at once -- spills 13 registers: 16.5 vs 14.6 us
dynamic_mask_head_kernel(const float* __restrict__ feats, const float* __restrict__ ref,
                         const float* __restrict__ params, const int* __restrict__ inst_image,
                         float* __restrict__ out, int H, int W, int n_inst, int stride,
                         int strips_x, int strips_y) {
  const int lane = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t wave_id = blockIdx.x * 4u + uint32_t(wave_in_block);     // the launcher keeps this below 2^31
  const uint32_t strips = uint32_t(strips_x * strips_y);
  const int j = int(wave_id / strips);  // instance
  if (j >= n_inst) return;
  const int s = int(wave_id - uint32_t(j) * strips);
  const int sy = s / strips_x, sx = s - sy * strips_x;
  constexpr int kLanes = 64 / HALVES, kCols = kLanes - 1;
  const int lane_in = lane & (kLanes - 1), half_id = lane / kLanes;
  const int x = sx * kCols - 1 + lane_in;           // lane 0 of a strip is its left halo
  const int xc = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
  const int y0 = (sy * HALVES + half_id) * kMhStripH;

  const float* P = params + int64_t(j) * kMhParams;
  const float refx = ref[2 * j], refy = ref[2 * j + 1];
  const int img = inst_image[j];
  // the frame's 8 feature planes through one buffer descriptor: per-lane offset = the pixel, scalar offset =
  // the plane -- 8 scalars instead of a 64-bit base address per (plane, row) pair (96 SGPRs, spilled)
  const __amdgpu_buffer_rsrc_t fsrc = uniform_rsrc(feats + int64_t(img) * kMhChannels * H * W,
                                                   uint32_t(kMhChannels) * uint32_t(H * W) * 4u);
  const uint32_t plane_bytes = uint32_t(H * W) * 4u;
  const float half = float(stride / 2);
  const float relx = refx - (float(xc * stride) + half);

  float* O = out + int64_t(j) * (2 * H) * (2 * W);

  // All rows of the strip (plus the halo row above it) live in registers, so every parameter is fetched
  // exactly once per wave and the 8 x (R+1) feature loads are all in flight together.  Rows are processed
  // in PAIRS held as float2 so the layer arithmetic is v_pk_fma_f32 (two FMAs per issued instruction).
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
      const uint32_t pix = uint32_t(yc * W + xc) * 4u;
#pragma unroll
      for (int c = 0; c < kMhChannels; ++c)
        v[h][2 + c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(fsrc, int(pix), int(uint32_t(c) * plane_bytes), 0));
    }
#pragma unroll
    for (int i = 0; i < kMhChannels + 2; ++i) x0[p][i] = float2_t{v[0][i], v[1][i]};
  }
  float logit[R1];
  mask_logits<RP>(P, x0, logit, [] {});
  // Stores: a lane owns 2 x 2 outputs, 8 bytes in each of two rows.  (Measured, round 2: without the stores
  // the kernel takes 10.5 of its 14.6 us at 360p; pairing neighbouring lanes by column parity so that each
  // writes one dwordx4 instead of two dwordx2 -- 4 shuffles per row -- gave 14.65 us, no gain: the cost is
  // the 18 MB of write traffic not the store instruction count.)
#pragma unroll
  for (int r = 1; r < R1; ++r) {
    const int y = y0 + r - 1;
    const float cur = logit[r], prev = logit[r - 1];
    const float left = lane_below(cur), prev_left = lane_below(prev);
    if (y < H && lane_in > 0 && x < W) {
      const float2_t top = {0.25f * ((prev_left + prev) + (left + cur)), 0.5f * (prev + cur)};
      const float2_t bot = {0.5f * (left + cur), cur};
      // write-once output streamed past the caches (`nt`): 14.6 -> 12.4 us at 360p, cold (the 18 MB no longer
      // evict the features and parameters the other waves are reading, and the lines need no allocation)
      __builtin_nontemporal_store(top, reinterpret_cast<float2_t*>(O + int64_t(2 * y) * (2 * W) + 2 * x));
      __builtin_nontemporal_store(bot, reinterpret_cast<float2_t*>(O + int64_t(2 * y + 1) * (2 * W) + 2 * x));
    }
  }
}


#endif  // VNX_DEV_VARIANTS

// ------------------------------------------------------------------------------ forward, row-major runs
// The strip kernel above spends lanes and rows on geometry: at W = 80 three 31-column half-strips use 80 of 93 lanes, one
// row in six is a halo, the last strip row of a 48-row frame is 8 of 10 rows -- 69 % of the computed pixels are stored.
// Here a wave takes a RUN of whole image rows of one instance in row-major order: register r of a chunk holds pixels
// f .. f + 63 of the flattened H x W frame -- every lane a pixel whatever W is, every feature load 256 contiguous bytes --
// and walks down its run in chunks of up to three register pairs.  The x2 up-sampling needs the logits of the pixel to
// the left and of the row above: the wave keeps the logits of its chunk, preceded by the last `halo` (>= W + 1) of the
// chunk before, in its own piece of LDS (a wave's LDS operations execute in order: no barrier) and reads the three
// neighbours at f - 1, f - W, f - W - 1 from there.  A run that does not start at the top of the frame computes the row
// above it first: W pixels of halo per run instead of one row in six.
// Everything around the layers is kept off the vector ALU where the hardware offers another place for it (the layers
// themselves run at the VALU's issue rate; with 49 further vector instructions per register the rest was as long again):
//   * LDS addresses are the lane's offset + an immediate; what depends on x, y is a select between per-wave constants;
//   * feature loads: the frame's eight planes are one descriptor, the pixel in the vector offset (+ an immediate per
//     register), the plane in the scalar offset (the range check takes it off the buffer's size): a pixel past the frame
//     reads the next plane or, in the last one, zero -- nothing is clamped, nothing past the tensor is touched;
//   * stores: the descriptor covers exactly the run's own output rows, offsets are relative to them -- the halo's lanes
//     come out negative, the lanes past the run's end beyond the range, and the hardware drops both.
// The three gradient buffers of the backward zero-filled by a slice of threads each (grid-stride; scalar stores up to the first
// 16-byte boundary of `a` -- a caller's pointer, a view with a storage offset is only 4-byte aligned -- 16-byte
// stores for its body, scalar stores for its tail and for the two small buffers).
__device__ __forceinline__ void zero3_slice(float* __restrict__ a, size_t na, float* __restrict__ b, size_t nb,
                                            float* __restrict__ c, size_t nc, size_t t0, size_t stride) {
  size_t head = (size_t(16) - (reinterpret_cast<uintptr_t>(a) & 15u)) & 15u;
  head = head / 4 < na ? head / 4 : na;                    // floats before the boundary (the pointer is 4-byte aligned)
  const size_t na4 = (na - head) / 4;
  float4_t* a4 = reinterpret_cast<float4_t*>(a + head);
  for (size_t i = t0; i < head; i += stride) a[i] = 0.f;
  for (size_t i = t0; i < na4; i += stride) a4[i] = float4_t{0.f, 0.f, 0.f, 0.f};
  for (size_t i = head + na4 * 4 + t0; i < na; i += stride) a[i] = 0.f;
  for (size_t i = t0; i < nb; i += stride) b[i] = 0.f;
  for (size_t i = t0; i < nc; i += stride) c[i] = 0.f;
}

// ZERO = 1 (the training forward, vnx_dynamic_mask_head_forward_train): before anything else every thread of the grid
// zero-fills its slice of the backward's three gradient buffers (za / zb / zc) -- the zero-fill the backward otherwise
// launches for itself (4.8 us of the 35-us training forward + backward; round 6).  ZERO = 0: the same kernel without it.
template <int ZERO>
__global__ void __launch_bounds__(256)
dynamic_mask_head_runs_kernel(const float* __restrict__ feats, const float* __restrict__ ref,
                              const float* __restrict__ params, const int* __restrict__ inst_image,
                              float* __restrict__ out, int H, int W, int n_inst, int stride, int runs, int halo,
                              float* __restrict__ za, size_t zna, float* __restrict__ zb, size_t znb,
                              float* __restrict__ zc, size_t znc) {
  extern __shared__ float mh_lds[];
  if constexpr (ZERO != 0)
    zero3_slice(za, zna, zb, znb, zc, znc, size_t(blockIdx.x) * blockDim.x + threadIdx.x, size_t(gridDim.x) * blockDim.x);
  const int lane = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t wave_id = blockIdx.x * 4u + uint32_t(wave_in_block);     // the launcher keeps this below 2^31
  const int j = int(wave_id / uint32_t(runs));  // instance
  if (j >= n_inst) return;
  const int k = int(wave_id) - j * runs;
  const int ya = (k * H) / runs, yb = ((k + 1) * H) / runs;               // the run's rows [ya, yb)
  if (ya >= yb) return;
  const int f0 = ya * W, f1 = yb * W;                                     // its pixels
  const int start = ya > 0 ? f0 - W : 0;                                  // first computed pixel: the row above the run
  const int total = f1 - start;

  const float* P = params + int64_t(j) * kMhParams;
  const float refx = ref[2 * j], refy = ref[2 * j + 1];
  // one frame in the call (inference: 300 instances of one frame): every instance belongs to it, and the features need
  // not wait for a trip to memory to learn that (the launcher passes no index array)
  const int img = inst_image ? inst_image[j] : 0;
  // (requesting the instance's 676 parameter bytes here, beside the frame index, so that the layers' own scalar loads hit
  //  the cache: 9.9 against 9.7 us at 360p, 6.6 against 6.1 us at the training shape -- not kept)
  const uint32_t plane_bytes = uint32_t(H * W) * 4u;
  const __amdgpu_buffer_rsrc_t fsrc = uniform_rsrc(feats + int64_t(img) * kMhChannels * H * W,
                                                   uint32_t(kMhChannels) * plane_bytes);
  const int half = stride / 2;
  // the run's own output rows: [2 ya, 2 yb) x 2W
  const __amdgpu_buffer_rsrc_t osrc = uniform_rsrc(out + int64_t(j) * (2 * H) * (2 * W) + int64_t(4) * f0,
                                                   uint32_t(f1 - f0) * 16u);

  // the lane's pixel in the next register to be filled: start is a multiple of W
  const int ly = lane / W;
  int y = start / W + ly, x = lane - ly * W;
  const int q64 = 64 / W, r64 = 64 - q64 * W;                             // a register further on: 64 pixels
  int done = 0;                                                           // pixels of the run already computed

  // the features of six registers (one full chunk) from run pixel `first` on, all in flight together
  const uint32_t lane4 = uint32_t(lane) * 4u;
  float2_t feat[3][kMhChannels];       // [register pair][plane]: .x / .y = the pair's two registers, as the layers take them
  auto issue = [&](int first) {
    const uint32_t pix = uint32_t(start + first) * 4u + lane4;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int c = 0; c < kMhChannels; ++c) {
#if defined(VNX_MH_ABL) && (VNX_MH_ABL & 4)     // timing ablation: no feature loads
          const float f = float(pix) * 1e-6f + float(c);
#else
          const float f = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(fsrc, int(pix + 256u * (2 * p + h)),
                                                                               int(uint32_t(c) * plane_bytes), 0));
#endif
          if (h == 0) feat[p][c].x = f; else feat[p][c].y = f;
        }
      }
    }
  };
  issue(0);

  // the wave's LDS: [halo entries of the chunk before][384 of this chunk]
  char* const lds = reinterpret_cast<char*>(mh_lds) + size_t(wave_in_block) * size_t(halo + 384) * 4u;
  char* const at_d = lds + (uint32_t(halo) * 4u + lane4);                  // this lane's entry in register 0 of the chunk
  char* const at_c = at_d - 4, * const at_b = at_d - W * 4, * const at_a = at_b - 4;
  const int neg_lane16 = -16 * lane;

  auto chunk = [&](auto tag) {
    constexpr int RP = decltype(tag)::value, R = 2 * RP;
    int xs[R], ys[R];
    float2_t x0[RP][kMhChannels + 2];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      float v[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = 2 * p + h;
        xs[r] = x; ys[r] = y;
        // exactly the reference's arithmetic: location = x * stride + stride // 2 (an integer), one subtraction
        v[h][0] = refx - float(__mul24(x, stride) + half);   // (24-bit multiply: full rate)
        v[h][1] = refy - float(__mul24(y, stride) + half);
        x += r64; y += q64;
        if (x >= W) { x -= W; ++y; }
      }
      x0[p][0] = float2_t{v[0][0], v[1][0]};
      x0[p][1] = float2_t{v[0][1], v[1][1]};
#pragma unroll
      for (int c = 0; c < kMhChannels; ++c) x0[p][2 + c] = feat[p][c];
    }
    float logit[R];
    const int next = done + 64 * R;
    // the next chunk's features leave once this chunk's are consumed (their registers are free): layers two and three
    // and the stores below cover the latency
#if defined(VNX_MH_ABL) && (VNX_MH_ABL & 2)     // timing ablation: no layers
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      float2_t t = x0[p][0];
#pragma unroll
      for (int c = 1; c < kMhChannels + 2; ++c) t += x0[p][c];
      logit[2 * p] = t.x; logit[2 * p + 1] = t.y;
    }
    if (next < total) issue(next);
#else
    mask_logits<RP>(P, x0, logit, [&] { if (next < total) issue(next); });
#endif
#pragma unroll
    for (int r = 0; r < R; ++r) *reinterpret_cast<float*>(at_d + 256 * r) = logit[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // out[2y][2x] = (a+b+c+d)/4, out[2y][2x+1] = (b+d)/2, out[2y+1][2x] = (c+d)/2, out[2y+1][2x+1] = d with
    // a = in[y-1][x-1], b = in[y-1][x], c = in[y][x-1], d = in[y][x], indices clamped at 0.  All LDS reads of the chunk
    // first (no branch: a lane outside the run's own pixels reads something and stores nowhere).
    float na[R], nb[R], nc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool inner = xs[r] > 0, below = ys[r] > 0;
      const char* pc = inner ? at_c : at_d;
      nc[r] = *reinterpret_cast<const float*>(pc + 256 * r);
      nb[r] = *reinterpret_cast<const float*>((below ? at_b : at_d) + 256 * r);
      na[r] = *reinterpret_cast<const float*>((below ? (inner ? at_a : at_b) : pc) + 256 * r);
    }
    const int chunk16 = (start + done - f0) * 16;     // byte offset of the chunk's first pixel's output, in the run's rows
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float d = logit[r];
      const float2_t top = {0.25f * ((na[r] + nb[r]) + (nc[r] + d)), 0.5f * (nb[r] + d)};
      const float2_t bot = {0.5f * (nc[r] + d), d};
      // byte offset of out[2y][2x] in the run's rows: 4 * (4 (y - ya) W + 2 x) = 16 (f - f0) - 8 x
      const int off = (chunk16 + 1024 * r) - ((xs[r] << 3) + neg_lane16);
#if defined(VNX_MH_ABL) && (VNX_MH_ABL & 1)     // timing ablation: no stores (but for results no run produces)
      if (top.x != 1234.5f) continue;
#endif
      // write-once output streamed past the caches (`nt`)
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(store2_t, top), osrc, off, 0, 2);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(store2_t, bot), osrc, off + 8 * W, 0, 2);
    }
    done = next;
    if (RP == 3 && done < total) {
      // the last `halo` logits move to the front (64 at a time, all read before any is written: the two ranges overlap
      // when halo > 384)
      float t[6];
      const int regs = halo >> 6;
      for (int g = 0; g < regs; g += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
          if (g + u < regs) t[u] = *reinterpret_cast<const float*>(lds + lane4 + (384 + 64 * (g + u)) * 4);
#pragma unroll
        for (int u = 0; u < 6; ++u)
          if (g + u < regs) *reinterpret_cast<float*>(lds + lane4 + 64 * (g + u) * 4) = t[u];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };
  while (done < total) {
    const int left = total - done;                    // wave-uniform
    if (left > 256) chunk(std::integral_constant<int, 3>{});
    else if (left > 128) chunk(std::integral_constant<int, 2>{});
    else chunk(std::integral_constant<int, 1>{});
  }
}

// Runs per instance: the kernel is bound by vector-ALU issue, so its time is that of the most loaded SIMD --
// ceil(waves / 1024) waves (1 024 SIMDs; workgroups of four waves spread over a CU's four) of ceil(pixels / 128) register
// pairs each.  Among the run counts that fill two, three or four whole rounds of the chip (one wave per SIMD leaves its
// stalls uncovered: 12.2 us against 10.7 with two at 360p, 300 instances) the one with the smallest product; on a tie the
// one with fewer runs (fewer halo rows).  `forced` > 0: that many runs (development build).
static int mask_head_runs(int n, int H, int W, int forced) {
  if (forced > 0) return forced < H ? forced : H;
  int best = 1; long long best_cost = -1;
  for (int cap = 2; cap <= 4; ++cap) {
    long long runs = (1024LL * cap) / n;
    if (runs < 1) runs = 1;
    if (runs > H) runs = H;
    const long long rows = (H + runs - 1) / runs;
    const long long px = rows * W + (runs > 1 ? W : 0);
    const long long pairs = (px + 127) / 128;
    const long long per_simd = (n * runs + 1023) / 1024;
    const long long cost = per_simd * pairs;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = int(runs); }
  }
  return best;
}

// ------------------------------------------------------------------------------ backward
// Training path (forward_mask_head_train, segmentation_condInst.py:354-401: the matched
// instances of every frame, for each of the 6 decoder layers).  Autograd's chain for the
// reference is: transpose of the x2 up-sampling, three grouped-conv backward-data and three
// backward-weight launches with groups = n, the ReLU masks, and the sum over the repeated
// feature map.  Here one kernel: a wave owns (instance, 63 columns x kMbRows rows), recomputes
// the two hidden layers of its pixels from the 8 features (nothing was saved by the forward),
// back-propagates, keeps the 169 parameter gradients + 2 reference-point gradients as
// lane-local sums over its rows, reduces them across the wave with DPP once, and adds them to
// grad_params / grad_ref; the feature gradients of all instances of a frame meet in grad_feats
// through coalesced fp32 atomics.  Outputs are zero-filled by the C entry point (memset nodes,
// graph-capturable).  Lane 63 is a one-pixel halo on the right: the transpose of the
// up-sampling needs the upstream gradient around pixel (y, x+1) and (y+1, x+1).
#ifndef VNX_MB_ROWS
#define VNX_MB_ROWS 8
#endif
constexpr int kMbRows = VNX_MB_ROWS;
#ifndef VNX_MB_GROUP
#define VNX_MB_GROUP 4
#endif
constexpr int kMbGroup = VNX_MB_GROUP;   // waves (consecutive instances) per workgroup of the backward

struct UpAdj { float d, c, b, a; };  // what a pixel's 2x2 output block sends to in[y][x], in[y][x-1], in[y-1][x], in[y-1][x-1]

__device__ __forceinline__ float lane_above(float v) {
  // lane l receives lane l+1 (wave_shl:1); lane 63's result is never used
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v),
                                                                0x130, 0xF, 0xF, false));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_fold(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}

// packed FMA forms of the backward kernel (two FMAs per issued instruction):
//   pk_fma_sv:  d += (scalar pair of weights) * (pair of inputs)
//   pk_fma_sb:  d += (scalar pair of weights) * (one gradient, the low / high half of g, for both lanes)
//   pk_fma_bv:  d += (one gradient, the low / high half of g, for both lanes) * (pair of inputs)
__device__ __forceinline__ void pk_fma_sv(float2_t& d, uint64_t w_pair, float2_t x) {
  asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d) : "s"(w_pair), "v"(x));
}
template <bool HI>
__device__ __forceinline__ void pk_fma_sb(float2_t& d, uint64_t w_pair, float2_t g) {
  if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(d) : "s"(w_pair), "v"(g));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(d) : "s"(w_pair), "v"(g));
}
template <bool HI>
__device__ __forceinline__ void pk_fma_bv(float2_t& d, float2_t g, float2_t x) {
  if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(d) : "v"(g), "v"(x));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(d) : "v"(g), "v"(x));
}

// Wave sums of MANY registers at once ("transposed" reduction).  Summing one register over the 64 lanes takes six shuffle +
// add steps whatever is done; but after the first step only half of the lanes carry anything new, so two registers can
// share one: fold(A, B) = a register whose one half holds A's pair sums and whose other half holds B's.  N registers
// become N / 2, N / 4, ... and finally one, in which lane l holds the complete sum of register tree_index(l): about 2 N
// operations instead of 6 N (+ 2 N to move each total to its lane).  The six folds, by the distance of the lanes they add
// (each checked lane by lane on the hardware: tools/fold_probe.hip):
//   32, 16: v_permlane32_swap / v_permlane16_swap (gfx950) exchange half of A with half of B in place, one add;
//    8, 4:  two DPP adds, each writing only lanes it has a source for and whose quad its bank_mask names
//           (row_shr:8 / row_shl:8; row_shr:4 on quads 1, 3 / row_shl:4 on quads 0, 2 -- a "bank" is a quad of a row);
//    2, 1:  no write mask is finer than a quad: two selects on the lane's bit (the half to keep, the half to send) and one
//           DPP add (quad_perm).
// A fold with no partner (N not a power of two) is the first of its two operations.
// asm: hipcc does not know that these statements are DPP / lane-swap instructions and adds no wait states after the
// vector instruction that produced their source (the probe read unshuffled values without them): s_nop 1 in front.
__device__ __forceinline__ float fold32(float a, float b) {     // lanes 0..31: a[l] + a[l + 32]; lanes 32..63: b[l - 32] + b[l]
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));       // a = [a.lo | b.lo], b = [a.hi | b.hi]
  return a + b;
}
__device__ __forceinline__ float fold16(float a, float b) {     // rows 0, 2 (of 16 lanes): a's rows 0 + 1, 2 + 3; rows 1, 3: b's
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));       // a = [a0 b0 a2 b2], b = [a1 b1 a3 b3]
  return a + b;
}
__device__ __forceinline__ float fold8(float a) {               // lanes 8..15 of a row: a[l - 8] + a[l]
  float r = a;
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ float fold8(float a, float b) {      // ... and lanes 0..7: b[l + 8] + b[l]
  float r = fold8(a);
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(b));
  return r;
}
__device__ __forceinline__ float fold4(float a) {               // quads 1, 3 of a row: a[l - 4] + a[l]
  float r = a;
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ float fold4(float a, float b) {      // ... and quads 0, 2: b[l + 4] + b[l]
  float r = fold4(a);
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(r) : "v"(b));
  return r;
}
__device__ __forceinline__ float quad_swap2(float v) {          // v[l ^ 2]
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));   // quad_perm:[2,3,0,1]
}
__device__ __forceinline__ float quad_swap1(float v) {          // v[l ^ 1]
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));   // quad_perm:[1,0,3,2]
}
// the sums of r[0 .. N), N <= 64: lane l of the result holds the one of r[tree_index(l)] (if that is below N)
template <int N>
__device__ __forceinline__ float wave_sums(float (&r)[N], int lane) {
  static_assert(N >= 1 && N <= 64, "one tree");
  constexpr int n1 = (N + 1) / 2, n2 = (n1 + 1) / 2, n3 = (n2 + 1) / 2, n4 = (n3 + 1) / 2, n5 = (n4 + 1) / 2;
  const float zero = 0.f;
#pragma unroll
  for (int m = 0; m < n1; ++m) r[m] = fold32(r[2 * m], 2 * m + 1 < N ? r[2 * m + 1] : zero);
#pragma unroll
  for (int m = 0; m < n2; ++m) r[m] = fold16(r[2 * m], 2 * m + 1 < n1 ? r[2 * m + 1] : zero);
#pragma unroll
  for (int m = 0; m < n3; ++m) r[m] = 2 * m + 1 < n2 ? fold8(r[2 * m], r[2 * m + 1]) : fold8(r[2 * m]);
#pragma unroll
  for (int m = 0; m < n4; ++m) r[m] = 2 * m + 1 < n3 ? fold4(r[2 * m], r[2 * m + 1]) : fold4(r[2 * m]);
  // (written as keep + swap(send): a select between two shuffled sums invites the compiler to shuffle the selected
  //  value instead, which mixes the two registers)
  const bool bit1 = (lane & 2) != 0, bit0 = (lane & 1) != 0;
#pragma unroll
  for (int m = 0; m < n5; ++m) {
    if (2 * m + 1 < n4) {
      const float a = r[2 * m], b = r[2 * m + 1];
      r[m] = (bit1 ? b : a) + quad_swap2(bit1 ? a : b);
    } else {
      r[m] = r[2 * m] + quad_swap2(r[2 * m]);
    }
  }
  if (n5 > 1) return (bit0 ? r[1] : r[0]) + quad_swap1(bit0 ? r[0] : r[1]);
  return r[0] + quad_swap1(r[0]);
}
// which register's sum lane l holds: bit k of the index = the side taken at the fold of distance 32 >> k
// (32, 16: the upper half / the odd rows hold the second register; 8, 4: the LOWER lanes do; 2, 1: the lane's bit)
__device__ __forceinline__ int tree_index(int l) {
  return ((l >> 5) & 1) | (((l >> 4) & 1) << 1) | ((((l >> 3) & 1) ^ 1) << 2) | ((((l >> 2) & 1) ^ 1) << 3) |
         (((l >> 1) & 1) << 4) | ((l & 1) << 5);
}

// PART: the 171 lane-local accumulators + the activations of a pixel were 268 VGPRs -- ONE wave per SIMD, and the
// training shape (120 instances x 12 strips) is 1.4 waves per SIMD, each a 6 200-instruction dependent sequence: 44.7 us.
// Two waves per strip instead: part 0 keeps the first layer's weight gradients (80) and sends the feature gradients,
// part 1 everything else (64 + 8 + 8 + 1 of layers two and three, the first layer's bias, the reference point); both
// recompute the forward and the data side of the backward chain (+50 % arithmetic), each fits 168 VGPRs: three waves
// per SIMD, the whole training shape resident at once.
template <int PART>
__device__ __forceinline__ void
mask_head_bwd_part(const int block, const float* __restrict__ feats, const float* __restrict__ ref,
                   const float* __restrict__ params, const int* __restrict__ inst_image,
                   const float* __restrict__ grad_out, float* __restrict__ grad_feats,
                   float* __restrict__ grad_ref, float* __restrict__ grad_params,
                   int H, int W, int n_inst, int stride, int strips_x, int strips_y, float* __restrict__ pool,
                   int* __restrict__ pool_img) {
  // `block` = (group of kMbGroup consecutive instances, strip): the wave's instance is group * kMbGroup + its index.
  // pool: the workgroup's LDS [2][kMbGroup][8 planes][64 lanes] through which part 0's waves add up their feature
  // gradients when they all belong to one frame (see the kernel below); null for part 1.
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int strips = strips_x * strips_y;
  const int q = block / strips;  // group
  const int j = q * kMbGroup + wave;  // instance
  const int s = block - q * strips;
  const int sy = s / strips_x, sx = s - sy * strips_x;
  const int x = sx * kMhStripW + lane;               // lane 63 is the right halo
  const int xc = x > W - 1 ? W - 1 : x;
  const int y0 = sy * kMbRows;
  const bool col_ok = x < W;
  const bool owner = col_ok && lane < kMhStripW;

  constexpr bool kPools = PART == 0 && kMbGroup > 1;
  bool pooled = false;
  int pool_frame = 0;
  if constexpr (kPools) {
    // publish the frame of every wave's instance and agree on whether the group is one frame's
    if (lane == 0) pool_img[wave] = j < n_inst ? inst_image[j] : -1;
    __syncthreads();
    pooled = true;
    pool_frame = pool_img[0];                        // wave 0's instance exists if any of the group's does
#pragma unroll
    for (int w = 1; w < kMbGroup; ++w) pooled = pooled && (pool_img[w] == pool_frame || pool_img[w] < 0);
  }
  // Row r's feature gradients of the group's waves -> one atomic per pixel and plane: every wave stores its eight values
  // in its slab (plain LDS stores: ds_add_f32 runs at 0.4 lanes per clock on this part), one barrier, then
  // wave w sums plane w (w + kMbGroup, ...) over the slabs and sends it.  Two sets of slabs alternate by the row's parity,
  // so a wave may write row r + 1 while another still reads row r.
  auto exchange = [&](int r, int y, const float (&v)[kMhChannels]) {
    float* const set = pool + (r & 1) * (kMbGroup * kMhChannels * 64);
#pragma unroll
    for (int c = 0; c < kMhChannels; ++c) set[(wave * kMhChannels + c) * 64 + lane] = v[c];
    __syncthreads();
    float* const GFq = grad_feats + int64_t(pool_frame) * kMhChannels * H * W;
    for (int c = wave; c < kMhChannels; c += kMbGroup) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kMbGroup; ++w) t += set[(w * kMhChannels + c) * 64 + lane];
      if (owner) atomic_add(GFq + (int64_t(c) * H + y) * W + x, t);
    }
  };
  if (j >= n_inst) {
    if constexpr (kPools) {
      if (pooled) {           // no instance: zeros into the exchange, but this wave's share of the sums is real
        const float zeros[kMhChannels] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < kMbRows && y0 + r < H; ++r) exchange(r, y0 + r, zeros);
      }
    }
    return;
  }

  const float* P = params + int64_t(j) * kMhParams;
  constexpr int W0 = 0, W1 = 80, W2 = 144, B0 = 152, B1 = 160;
  const float refx = ref[2 * j], refy = ref[2 * j + 1];
  const int img = inst_image[j];
  const float* F = feats + int64_t(img) * kMhChannels * H * W;
  float* GF = grad_feats + int64_t(img) * kMhChannels * H * W;
  const float* G = grad_out + int64_t(j) * (2 * H) * (2 * W);
  const float half = float(stride / 2);
  const float relx = refx - (float(xc * stride) + half);

  auto up_adj = [&](int y) -> UpAdj {
    float2_t top = {0.f, 0.f}, bot = {0.f, 0.f};
#if defined(VNX_MB_ABL) && (VNX_MB_ABL & 8)      // timing ablation: no upstream-gradient loads
    if (y < H && col_ok && relx == 1234.5f) {
#else
    if (y < H && col_ok) {
#endif
      top = *reinterpret_cast<const float2_t*>(G + int64_t(2 * y) * (2 * W) + 2 * x);
      bot = *reinterpret_cast<const float2_t*>(G + int64_t(2 * y + 1) * (2 * W) + 2 * x);
    }
    const float q = 0.25f * top.x;
    return UpAdj{q + 0.5f * (top.y + bot.x) + bot.y, q + 0.5f * bot.x, q + 0.5f * top.y, q};
  };

  constexpr bool kHas0 = PART != 1, kHas1 = PART != 0;        // part 0: first-layer weights + features; part 1: the rest
  constexpr int kA = kHas0 ? kMhHidden : 1, kB = kHas1 ? kMhHidden : 1;     // the other part's arrays shrink to nothing
  constexpr int kP0 = (kMhChannels + 2) / 2, kP1 = kMhHidden / 2;           // pairs per row of W0 / W1
  float2_t acc_w0[kA][kP0], acc_w1[kB][kP1], acc_w2[kB > 1 ? kP1 : 1];
  float acc_b0[kB], acc_b1[kB], acc_b2 = 0.f, acc_rx = 0.f, acc_ry = 0.f;
  const float2_t zero2 = {0.f, 0.f};
#pragma unroll
  for (int o = 0; o < kA; ++o) {
#pragma unroll
    for (int k = 0; k < kP0; ++k) acc_w0[o][k] = zero2;
  }
#pragma unroll
  for (int o = 0; o < kB; ++o) {
#pragma unroll
    for (int k = 0; k < kP1; ++k) acc_w1[o][k] = zero2;
    acc_b0[o] = 0.f; acc_b1[o] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < (kB > 1 ? kP1 : 1); ++k) acc_w2[k] = zero2;

  // Parameters stream through the scalar file as in the forward kernel: groups of the 20 (16) weights of two output
  // channels (+ their biases), fetched one group ahead (s_load into ga / gb alternately, s_waitcnt by hand -- the loads are
  // invisible to hipcc's own counting).  Left to the compiler all 169 stayed live across the row loop, 150 of them
  // spilled into VGPR lanes: ~240 v_readlane per pixel row next to ~330 FMAs.  A pair of neighbouring weights is one
  // 64-bit scalar operand of a packed FMA over a pair of inputs (forward, data side of the backward); the weight side
  // multiplies a broadcast gradient with input pairs.
  struct Group { sgpr16_t w; sgpr4_t w2; sgpr2_t b; };
  auto fetch20 = [&](Group& g, int w_at, int b_at) {   // 20 consecutive weights, 2 consecutive biases
    asm volatile("s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx4 %1, %3, %5\n\ts_load_dwordx2 %2, %3, %6"
                 : "=&s"(g.w), "=&s"(g.w2), "=&s"(g.b)
                 : "s"(P), "n"(w_at * 4), "n"(w_at * 4 + 64), "n"(b_at * 4)
                 : "memory");
  };
  auto fetch16 = [&](Group& g, int w_at, int b_at) {   // 16 consecutive weights, 2 consecutive biases
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx2 %1, %2, %4"
                 : "=&s"(g.w), "=&s"(g.b)
                 : "s"(P), "n"(w_at * 4), "n"(b_at * 4)
                 : "memory");
  };
  auto fetch8 = [&](Group& g, int w_at) {              // the last layer's 8 weights
    sgpr8_t t;
    asm volatile("s_load_dwordx8 %0, %1, %2" : "=&s"(t) : "s"(P), "n"(w_at * 4) : "memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) g.w[q] = t[q];
  };
  auto landed = [&](Group& g) {
#if defined(VNX_MB_ABL) && (VNX_MB_ABL & 2)      // timing ablation: the parameter groups are not waited for
    asm volatile("" : "+s"(g.w), "+s"(g.w2), "+s"(g.b)::"memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(g.w), "+s"(g.w2), "+s"(g.b)::"memory");
#endif
  };
  auto wpair = [](const Group& g, int k) -> uint64_t {   // weights 2k, 2k + 1 of the group as one SGPR pair
    return k < 8 ? (uint64_t(g.w[2 * k + 1]) << 32) | g.w[2 * k] : (uint64_t(g.w2[2 * (k - 8) + 1]) << 32) | g.w2[2 * (k - 8)];
  };
#define VNX_FENCE __builtin_amdgcn_sched_barrier(0);

  UpAdj cur = up_adj(y0);
#pragma unroll 1
  for (int r = 0; r < kMbRows; ++r) {
    const int y = y0 + r;
    if (y >= H) break;
    Group ga, gb;
    ga.w2 = sgpr4_t{0u, 0u, 0u, 0u}; gb.w2 = ga.w2;
    fetch20(ga, W0, B0);
    const UpAdj nxt = up_adj(y + 1);  // zeros below the map
    // d(loss)/d(logit[y][x]) before up-sampling: the transpose of
    //   out[2y][2x] = (a+b+c+d)/4, out[2y][2x+1] = (b+d)/2, out[2y+1][2x] = (c+d)/2, out[2y+1][2x+1] = d
    // with a, b, c the clamped upper-left / upper / left neighbours (clamping folds the
    // first row's b, a and the first column's c, a back onto the pixel itself)
    float gl = cur.d + lane_above(cur.c) + nxt.b + lane_above(nxt.a);
    if (x == 0) gl += cur.c + nxt.a;
    if (y == 0) gl += cur.b + lane_above(cur.a);
    if (x == 0 && y == 0) gl += cur.a;
    if (!owner) gl = 0.f;   // halo lane / columns past the map contribute nothing
    cur = nxt;

    // inputs of the first layer as pairs: (rel_x, rel_y), (f0, f1), ...
    float2_t x0[kP0];
    x0[0] = float2_t{relx, refy - (float(y * stride) + half)};
#pragma unroll
    for (int c = 0; c < kMhChannels; c += 2)
#if defined(VNX_MB_ABL) && (VNX_MB_ABL & 4)      // timing ablation: no feature loads
      x0[1 + c / 2] = float2_t{relx * float(c), refy * float(y)};
#else
      x0[1 + c / 2] = float2_t{F[(int64_t(c) * H + y) * W + xc], F[(int64_t(c + 1) * H + y) * W + xc]};
#endif

    // ---- forward, recomputed: a dot product is a chain of packed FMAs over input pairs + one add of the halves
    float2_t x1[kP1], x2[kP1];
#define VNX_FWD(G, NP, X, OUT, o0)                                                                      \
    _Pragma("unroll") for (int oo = 0; oo < 2; ++oo) {                                                    \
      float2_t a = float2_t{__uint_as_float(G.b[oo]), 0.f};                                               \
      _Pragma("unroll") for (int k = 0; k < NP; ++k) pk_fma_sv(a, wpair(G, oo * NP + k), X[k]);           \
      const float v = fmaxf(a.x + a.y, 0.f);                                                              \
      if (oo == 0) OUT[(o0) / 2].x = v; else OUT[(o0) / 2].y = v;                                         \
    }
    landed(ga); fetch20(gb, W0 + 20, B0 + 2); VNX_FENCE
    VNX_FWD(ga, kP0, x0, x1, 0) VNX_FENCE
    landed(gb); fetch20(ga, W0 + 40, B0 + 4); VNX_FENCE
    VNX_FWD(gb, kP0, x0, x1, 2) VNX_FENCE
    landed(ga); fetch20(gb, W0 + 60, B0 + 6); VNX_FENCE
    VNX_FWD(ga, kP0, x0, x1, 4) VNX_FENCE
    landed(gb); fetch16(ga, W1, B1); VNX_FENCE
    VNX_FWD(gb, kP0, x0, x1, 6) VNX_FENCE
    landed(ga); fetch16(gb, W1 + 16, B1 + 2); VNX_FENCE
    VNX_FWD(ga, kP1, x1, x2, 0) VNX_FENCE
    landed(gb); fetch16(ga, W1 + 32, B1 + 4); VNX_FENCE
    VNX_FWD(gb, kP1, x1, x2, 2) VNX_FENCE
    landed(ga); fetch16(gb, W1 + 48, B1 + 6); VNX_FENCE
    VNX_FWD(ga, kP1, x1, x2, 4) VNX_FENCE
    landed(gb); fetch8(ga, W2); VNX_FENCE
    VNX_FWD(gb, kP1, x1, x2, 6) VNX_FENCE
#undef VNX_FWD
    // ---- layer 3: logit = w2 . x2 + b2
    landed(ga); fetch16(gb, W1, B1); VNX_FENCE
    float2_t g2[kP1];
    const float2_t gl2 = {gl, gl};
    if constexpr (kHas1) {
      acc_b2 += gl;
#pragma unroll
      for (int k = 0; k < kP1; ++k) acc_w2[k] = gl2 * x2[k] + acc_w2[k];
    }
#pragma unroll
    for (int k = 0; k < kP1; ++k) {
      g2[k].x = x2[k].x > 0.f ? __uint_as_float(ga.w[2 * k]) * gl : 0.f;
      g2[k].y = x2[k].y > 0.f ? __uint_as_float(ga.w[2 * k + 1]) * gl : 0.f;
    }
    VNX_FENCE
    // ---- layer 2: weight side (part 1): acc_w1[o] += g2[o] * x1;  data side (both): g1 += W1[o] * g2[o]
    float2_t g1[kP1];
#pragma unroll
    for (int k = 0; k < kP1; ++k) g1[k] = zero2;
#define VNX_BWD2(G, o0)                                                                                 \
    {                                                                                                     \
      if constexpr (kHas1) { acc_b1[(o0)] += g2[(o0) / 2].x; acc_b1[(o0) + 1] += g2[(o0) / 2].y; }        \
      _Pragma("unroll") for (int k = 0; k < kP1; ++k) {                                                   \
        if constexpr (kHas1) {                                                                            \
          pk_fma_bv<false>(acc_w1[(o0)][k], g2[(o0) / 2], x1[k]);                                         \
          pk_fma_bv<true>(acc_w1[(o0) + 1][k], g2[(o0) / 2], x1[k]);                                      \
        }                                                                                                 \
        pk_fma_sb<false>(g1[k], wpair(G, k), g2[(o0) / 2]);                                               \
        pk_fma_sb<true>(g1[k], wpair(G, kP1 + k), g2[(o0) / 2]);                                          \
      }                                                                                                   \
    }
    landed(gb); fetch16(ga, W1 + 16, B1 + 2); VNX_FENCE
    VNX_BWD2(gb, 0) VNX_FENCE
    landed(ga); fetch16(gb, W1 + 32, B1 + 4); VNX_FENCE
    VNX_BWD2(ga, 2) VNX_FENCE
    landed(gb); fetch16(ga, W1 + 48, B1 + 6); VNX_FENCE
    VNX_BWD2(gb, 4) VNX_FENCE
    landed(ga); fetch20(gb, W0, B0); VNX_FENCE
    VNX_BWD2(ga, 6) VNX_FENCE
#undef VNX_BWD2
#pragma unroll
    for (int k = 0; k < kP1; ++k) {
      g1[k].x = x1[k].x > 0.f ? g1[k].x : 0.f;
      g1[k].y = x1[k].y > 0.f ? g1[k].y : 0.f;
    }
    // ---- layer 1: weight side (part 0): acc_w0[o] += g1[o] * x0;  data side: features (part 0), reference point (part 1)
    float2_t g0[kP0];
#pragma unroll
    for (int k = 0; k < kP0; ++k) g0[k] = zero2;
#define VNX_BWD1(G, o0)                                                                                 \
    {                                                                                                     \
      if constexpr (kHas1) { acc_b0[(o0)] += g1[(o0) / 2].x; acc_b0[(o0) + 1] += g1[(o0) / 2].y; }        \
      _Pragma("unroll") for (int k = 0; k < kP0; ++k) {                                                   \
        if constexpr (kHas0) {                                                                            \
          pk_fma_bv<false>(acc_w0[(o0)][k], g1[(o0) / 2], x0[k]);                                         \
          pk_fma_bv<true>(acc_w0[(o0) + 1][k], g1[(o0) / 2], x0[k]);                                      \
        }                                                                                                 \
        if ((kHas0 && k > 0) || (kHas1 && k == 0)) {                                                      \
          pk_fma_sb<false>(g0[k], wpair(G, k), g1[(o0) / 2]);                                             \
          pk_fma_sb<true>(g0[k], wpair(G, kP0 + k), g1[(o0) / 2]);                                        \
        }                                                                                                 \
      }                                                                                                   \
    }
    landed(gb); fetch20(ga, W0 + 20, B0 + 2); VNX_FENCE
    VNX_BWD1(gb, 0) VNX_FENCE
    landed(ga); fetch20(gb, W0 + 40, B0 + 4); VNX_FENCE
    VNX_BWD1(ga, 2) VNX_FENCE
    landed(gb); fetch20(ga, W0 + 60, B0 + 6); VNX_FENCE
    VNX_BWD1(gb, 4) VNX_FENCE
    landed(ga); VNX_FENCE
    VNX_BWD1(ga, 6) VNX_FENCE
#undef VNX_BWD1
    if constexpr (kHas1) { acc_rx += g0[0].x; acc_ry += g0[0].y; }     // rel = ref - pixel centre
    if constexpr (kHas0) {
      bool direct = true;
      if constexpr (kPools) {
        if (pooled) {
          float v[kMhChannels];
#pragma unroll
          for (int c = 0; c < kMhChannels; c += 2) {
            v[c] = owner ? g0[1 + c / 2].x : 0.f;
            v[c + 1] = owner ? g0[1 + c / 2].y : 0.f;
          }
          exchange(r, y, v);
          direct = false;
        }
      }
      if (direct && owner) {
#pragma unroll
        for (int c = 0; c < kMhChannels; c += 2) {
          atomic_add(GF + (int64_t(c) * H + y) * W + x, g0[1 + c / 2].x);
          atomic_add(GF + (int64_t(c + 1) * H + y) * W + x, g0[1 + c / 2].y);
        }
      }
    }
  }
#undef VNX_FENCE

  // wave sums (wave_sums above): the accumulators in the order of the parameters they belong to, 64 to a tree; the lane
  // holding a total adds it to its parameter's gradient
  float* GP = grad_params + int64_t(j) * kMhParams;
  const int ti = tree_index(lane);
  if constexpr (kHas0) {
    // part 0: the first layer's 80 weights = parameters 0 .. 79
    float t0[64], t1[16];
#pragma unroll
    for (int f = 0; f < 80; ++f) {
      const float2_t a = acc_w0[f / (kMhChannels + 2)][(f % (kMhChannels + 2)) / 2];
      const float v = (f & 1) ? a.y : a.x;
      if (f < 64) t0[f] = v; else t1[f - 64] = v;
    }
    const float s0 = wave_sums(t0, lane), s1 = wave_sums(t1, lane);
    atomic_add(GP + ti, s0);
    if (ti < 16) atomic_add(GP + 64 + ti, s1);
  }
  if constexpr (kHas1) {
    // part 1: the second layer's 64 weights = parameters 80 .. 143; then w2 (144 .. 151), b0, b1, b2 (152 .. 168) and
    // the reference point (x, y) as slots 169, 170
    float t0[64], t1[27];
#pragma unroll
    for (int f = 0; f < 64; ++f) {
      const float2_t a = acc_w1[f / kMhHidden][(f % kMhHidden) / 2];
      t0[f] = (f & 1) ? a.y : a.x;
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      t1[f] = (f & 1) ? acc_w2[f / 2].y : acc_w2[f / 2].x;
      t1[8 + f] = acc_b0[f];
      t1[16 + f] = acc_b1[f];
    }
    t1[24] = acc_b2; t1[25] = acc_rx; t1[26] = acc_ry;
    const float s0 = wave_sums(t0, lane), s1 = wave_sums(t1, lane);
    atomic_add(GP + W1 + ti, s0);
    if (ti < 25) atomic_add(GP + W2 + ti, s1);
    else if (ti < 27) atomic_add(grad_ref + 2 * j + (ti - 25), s1);
  }
}

// zero-fill of the three gradient buffers of the backward, one launch (grid-stride, scalar stores: the small ones are not
// 16-byte multiples)
__global__ void __launch_bounds__(256)
zero3_kernel(float* __restrict__ a, size_t na, float* __restrict__ b, size_t nb, float* __restrict__ c, size_t nc) {
  zero3_slice(a, na, b, nb, c, nc, size_t(blockIdx.x) * blockDim.x + threadIdx.x, size_t(gridDim.x) * blockDim.x);
}

// one launch, the two parts of a strip in neighbouring workgroups (they read the same features and upstream gradients)
// One launch; a workgroup = kMbGroup waves = the same strip and part for kMbGroup consecutive instances, the two parts of a
// (group, strip) in neighbouring workgroups (they read the same features and upstream gradients).
// Why groups: the feature gradients of all instances of a frame add into the same [8, H, W] planes, and the chip does
// 322 G fp32 atomics per second (tools/atomic_bench.hip): 120 instances x 12 strips x 8 rows x 8 planes x 63 lanes = 5.8 M
// of them are 18 us on their own, under a kernel whose arithmetic is 15 us (without them the forward + backward graph ran
// 35.2 instead of 40.2 us).  Consecutive instances are the same frame's (the callers pass per-frame counts), so the waves of
// a group add into LDS first and the workgroup issues one global atomic per pixel and plane; a group that straddles two
// frames falls back to per-wave atomics.
constexpr int kMbParts = 2;
__global__ void __launch_bounds__(64 * kMbGroup) __attribute__((amdgpu_waves_per_eu(3, 3)))
dynamic_mask_head_bwd_kernel(const float* __restrict__ feats, const float* __restrict__ ref,
                             const float* __restrict__ params, const int* __restrict__ inst_image,
                             const float* __restrict__ grad_out, float* __restrict__ grad_feats,
                             float* __restrict__ grad_ref, float* __restrict__ grad_params,
                             int H, int W, int n_inst, int stride, int strips_x, int strips_y) {
  __shared__ float pool[kMbGroup > 1 ? 2 * kMbGroup * kMhChannels * 64 : 1];
  __shared__ int pool_img[kMbGroup];
  const int block = int(blockIdx.x >> 1);
  if (blockIdx.x & 1)
    mask_head_bwd_part<1>(block, feats, ref, params, inst_image, grad_out, grad_feats, grad_ref, grad_params, H, W, n_inst,
                          stride, strips_x, strips_y, nullptr, nullptr);
  else
    mask_head_bwd_part<0>(block, feats, ref, params, inst_image, grad_out, grad_feats, grad_ref, grad_params, H, W, n_inst,
                          stride, strips_x, strips_y, pool, pool_img);
}

}  // namespace vnx

using namespace vnx;

// forward; zero != nullptr: the training forward -- grad_feats / grad_ref / grad_params zero-filled by the same launch
struct MaskHeadZero { void* grad_feats; void* grad_ref; void* grad_params; };

static int mask_head_forward_impl(int dtype, const void* mask_feats,
                                  const void* reference_points, const void* params,
                                  const int32_t* inst_image, void* out, int num_images,
                                  int channels, int height, int width, int num_insts,
                                  int num_params, int stride, const MaskHeadZero* zero, void* hip_stream) {
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
  const size_t zn0 = size_t(num_images) * kMhChannels * size_t(height) * size_t(width), zn1 = size_t(num_insts) * 2,
               zn2 = size_t(num_insts) * kMhParams;
  if (zero && ((zn0 && !zero->grad_feats) || (num_insts && (!zero->grad_ref || !zero->grad_params)))) {
    set_error("vnx_dynamic_mask_head_forward_train: null gradient buffer");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const int variant = kernel_variant();
  if (zero && (num_insts == 0 || height == 0 || width == 0 || variant == 799)) {
    // no forward launch to fold the zero-fill into (or the development build's strip kernel): its own launch
    if (zn0 + zn1 + zn2) {
      size_t blocks = ((zn0 + zn1 + zn2) / 4 + 255) / 256;
      blocks = blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks);
      hipLaunchKernelGGL(zero3_kernel, dim3(uint32_t(blocks)), dim3(256), 0, (hipStream_t)hip_stream, (float*)zero->grad_feats, zn0,
                         (float*)zero->grad_ref, zn1, (float*)zero->grad_params, zn2);
      const int zs = check_launch("mask_head_fwd zero");
      if (zs != VNX_OK) return zs;
    }
    zero = nullptr;
  }
  if (num_insts == 0 || height == 0 || width == 0) return VNX_OK;
  if (!mask_feats || !reference_points || !params || !inst_image || !out) {
    set_error("vnx_dynamic_mask_head_forward: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  if (variant != 799) {                      // (development build: 799 = the strip kernel, 700 + r = r runs per instance)
    const int runs = mask_head_runs(num_insts, height, width, variant > 700 && variant < 799 ? variant - 700 : 0);
    const int halo = (width + 1 + 63) / 64 * 64;      // logits kept from the chunk before: the row above + the pixel to the left
    const int64_t waves = int64_t(num_insts) * runs;
    const size_t lds_bytes = size_t(4) * size_t(halo + 384) * sizeof(float);
    if (waves >= (int64_t(1) << 31) || lds_bytes > 64 * 1024 || int64_t(height) * width >= (int64_t(1) << 26)) {   // 32-bit byte offsets into a frame's output
      set_error("vnx_dynamic_mask_head_forward: %lld waves / frame %d x %d exceed the kernel's limits", (long long)waves, height, width);
      return VNX_ERR_UNSUPPORTED;
    }
    if (zero)
      hipLaunchKernelGGL(dynamic_mask_head_runs_kernel<1>, dim3(uint32_t((waves + 3) / 4)), dim3(256), lds_bytes,
                         (hipStream_t)hip_stream, (const float*)mask_feats, (const float*)reference_points,
                         (const float*)params, num_images == 1 ? (const int*)nullptr : (const int*)inst_image, (float*)out,
                         height, width, num_insts, stride, runs, halo, (float*)zero->grad_feats, zn0, (float*)zero->grad_ref, zn1,
                         (float*)zero->grad_params, zn2);
    else
      hipLaunchKernelGGL(dynamic_mask_head_runs_kernel<0>, dim3(uint32_t((waves + 3) / 4)), dim3(256), lds_bytes,
                         (hipStream_t)hip_stream, (const float*)mask_feats, (const float*)reference_points,
                         (const float*)params, num_images == 1 ? (const int*)nullptr : (const int*)inst_image, (float*)out,
                         height, width, num_insts, stride, runs, halo, (float*)nullptr, size_t(0), (float*)nullptr, size_t(0),
                         (float*)nullptr, size_t(0));
    return check_launch("dynamic_mask_head");
  }
#ifdef VNX_DEV_VARIANTS
  // lanes spent per useful pixel by the two strip shapes (see the kernel): choose the tighter one
  const int sx1 = (width + 62) / 63, sy1 = (height + kMhStripH - 1) / kMhStripH;
  const int sx2 = (width + 30) / 31, sy2 = (height + 2 * kMhStripH - 1) / (2 * kMhStripH);
  const bool halves = int64_t(sx2) * sy2 < int64_t(sx1) * sy1;
  const int strips_x = halves ? sx2 : sx1, strips_y = halves ? sy2 : sy1;
  const int64_t waves = int64_t(num_insts) * strips_x * strips_y;
  const int64_t blocks = (waves + 3) / 4;
  if (waves >= (int64_t(1) << 31)) {
    set_error("vnx_dynamic_mask_head_forward: %lld waves exceed the grid limit", (long long)waves);
    return VNX_ERR_UNSUPPORTED;
  }
  if (halves)
    hipLaunchKernelGGL(dynamic_mask_head_kernel<2>, dim3(uint32_t(blocks)), dim3(256), 0,
                       (hipStream_t)hip_stream, (const float*)mask_feats, (const float*)reference_points,
                       (const float*)params, (const int*)inst_image, (float*)out, height, width,
                       num_insts, stride, strips_x, strips_y);
  else
    hipLaunchKernelGGL(dynamic_mask_head_kernel<1>, dim3(uint32_t(blocks)), dim3(256), 0,
                       (hipStream_t)hip_stream, (const float*)mask_feats, (const float*)reference_points,
                       (const float*)params, (const int*)inst_image, (float*)out, height, width,
                       num_insts, stride, strips_x, strips_y);
  return check_launch("dynamic_mask_head");
#else
  return VNX_ERR_UNSUPPORTED;   // not reached: the product build has no variants
#endif
}

extern "C" int vnx_dynamic_mask_head_forward(int dtype, const void* mask_feats,
                                             const void* reference_points, const void* params,
                                             const int32_t* inst_image, void* out, int num_images,
                                             int channels, int height, int width, int num_insts,
                                             int num_params, int stride, void* hip_stream) {
  return mask_head_forward_impl(dtype, mask_feats, reference_points, params, inst_image, out, num_images, channels, height, width,
                                num_insts, num_params, stride, nullptr, hip_stream);
}

extern "C" int vnx_dynamic_mask_head_forward_train(int dtype, const void* mask_feats,
                                                   const void* reference_points, const void* params,
                                                   const int32_t* inst_image, void* out, void* grad_feats, void* grad_ref,
                                                   void* grad_params, int num_images, int channels, int height, int width,
                                                   int num_insts, int num_params, int stride, void* hip_stream) {
  const MaskHeadZero zero{grad_feats, grad_ref, grad_params};
  return mask_head_forward_impl(dtype, mask_feats, reference_points, params, inst_image, out, num_images, channels, height, width,
                                num_insts, num_params, stride, &zero, hip_stream);
}

static int mask_head_backward_impl(int dtype, const void* mask_feats, const void* reference_points,
                                   const void* params, const int32_t* inst_image,
                                   const void* grad_out, void* grad_feats, void* grad_ref,
                                   void* grad_params, int num_images, int channels, int height,
                                   int width, int num_insts, int num_params, int stride, bool zero_first,
                                   void* hip_stream) {
  if (dtype != VNX_F32) {
    set_error("vnx_dynamic_mask_head_backward: only f32 is built (got dtype %d)", dtype);
    return VNX_ERR_UNSUPPORTED;
  }
  if (channels != kMhChannels || num_params != kMhParams) {
    set_error("vnx_dynamic_mask_head_backward: built for %d feature channels / %d parameters; got %d / %d",
              kMhChannels, kMhParams, channels, num_params);
    return VNX_ERR_UNSUPPORTED;
  }
  if (num_images < 0 || height < 0 || width < 0 || num_insts < 0 || stride <= 0) {
    set_error("vnx_dynamic_mask_head_backward: bad sizes");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  const size_t feat_bytes = size_t(num_images) * kMhChannels * height * width * sizeof(float);
  if ((feat_bytes && !grad_feats) || (num_insts && (!grad_ref || !grad_params))) {
    set_error("vnx_dynamic_mask_head_backward: null output pointer");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  // every output element is defined on return: zero-fill, then accumulate.  ONE launch for the three buffers (three memset
  // nodes were ~3 us of the 42 us training-shape forward + backward)
  {
    const size_t n0 = feat_bytes / 4, n1 = size_t(num_insts) * 2, n2 = size_t(num_insts) * kMhParams;
    const size_t n = n0 + n1 + n2;
    if (n && zero_first) {
      size_t blocks = (n / 4 + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      if (blocks < 1) blocks = 1;
      hipLaunchKernelGGL(zero3_kernel, dim3(uint32_t(blocks)), dim3(256), 0, stream, (float*)grad_feats, n0, (float*)grad_ref, n1,
                         (float*)grad_params, n2);
      const int zs = check_launch("mask_head_bwd zero");
      if (zs != VNX_OK) return zs;
    }
  }
  if (num_insts == 0 || height == 0 || width == 0) return VNX_OK;
  if (!mask_feats || !reference_points || !params || !inst_image || !grad_out) {
    set_error("vnx_dynamic_mask_head_backward: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const int strips_x = (width + kMhStripW - 1) / kMhStripW;
  const int strips_y = (height + kMbRows - 1) / kMbRows;
  const int64_t groups = (int64_t(num_insts) + kMbGroup - 1) / kMbGroup;
  const int64_t blocks = groups * strips_x * strips_y * kMbParts;                  // two parts per (group, strip)
  if (blocks >= (int64_t(1) << 31)) {
    set_error("vnx_dynamic_mask_head_backward: %lld workgroups exceed the grid limit", (long long)blocks);
    return VNX_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(dynamic_mask_head_bwd_kernel, dim3(uint32_t(blocks)), dim3(64 * kMbGroup), 0, stream,
                     (const float*)mask_feats, (const float*)reference_points, (const float*)params,
                     (const int*)inst_image, (const float*)grad_out, (float*)grad_feats, (float*)grad_ref,
                     (float*)grad_params, height, width, num_insts, stride, strips_x, strips_y);
  return check_launch("dynamic_mask_head_bwd");
}

extern "C" int vnx_dynamic_mask_head_backward(int dtype, const void* mask_feats, const void* reference_points,
                                              const void* params, const int32_t* inst_image,
                                              const void* grad_out, void* grad_feats, void* grad_ref,
                                              void* grad_params, int num_images, int channels, int height,
                                              int width, int num_insts, int num_params, int stride,
                                              void* hip_stream) {
  return mask_head_backward_impl(dtype, mask_feats, reference_points, params, inst_image, grad_out, grad_feats, grad_ref, grad_params,
                                 num_images, channels, height, width, num_insts, num_params, stride, true, hip_stream);
}

extern "C" int vnx_dynamic_mask_head_backward_zeroed(int dtype, const void* mask_feats, const void* reference_points,
                                                     const void* params, const int32_t* inst_image,
                                                     const void* grad_out, void* grad_feats, void* grad_ref,
                                                     void* grad_params, int num_images, int channels, int height,
                                                     int width, int num_insts, int num_params, int stride,
                                                     void* hip_stream) {
  return mask_head_backward_impl(dtype, mask_feats, reference_points, params, inst_image, grad_out, grad_feats, grad_ref, grad_params,
                                 num_images, channels, height, width, num_insts, num_params, stride, false, hip_stream);
}
