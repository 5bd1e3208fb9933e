// msda_d32.hip -- multi-scale deformable attention tuned for the production head
// geometry of SeqFormer / IDOL: 32 channels per head (d_model 256 / 8 heads).
//
// Machine mapping (gfx950, wave64):
//  * One pixel of one head is a 32-channel row: 128 B in fp32, 64 B in bf16/f16.
//    Eight lanes cover it with one 16-B (8-B) load each, so a wave issues eight
//    row gathers per load instruction and every request is a whole cache line.
//  * blockIdx.x % heads selects the head.  The dispatcher hands consecutive
//    workgroups to consecutive XCDs, so with 8 heads each XCD's private 4 MiB L2
//    only ever sees one head's 1/8 slice of `value` instead of all of it (an
//    affinity for speed only; nothing depends on the placement).
//  * Phase 1 (set-up): the wave's lanes each take one (query, level, point)
//    sample, read its location and weight, and compute the four tap byte offsets
//    and the four bilinear weights (pre-multiplied by the attention weight) ONCE,
//    then park them in LDS.  The reference recomputes this in each of the 32
//    channel threads of a head (ms_deform_im2col_cuda.cuh:272-296).
//  * Phase 2 (gather): eight-lane groups walk their samples, broadcast-read the
//    32-B tap record from LDS and issue four buffer loads per sample.  Taps that
//    fall outside the map carry an offset beyond the buffer descriptor's range:
//    the hardware returns zeros for them, which is exactly the reference's
//    zero-padding rule (cuh:55-78), with no branch and no memory traffic.
//  * QPW (queries per wave) trades per-wave work for wave count: the 64 lanes
//    are QPW queries x (8/QPW) sample groups x 8 channel lanes; sample groups
//    are summed with cross-lane exchanges at the end.
#include "vnx_common.h"
#include "msda_d32_gvdirect_body.h"

namespace vnx {

constexpr uint32_t kTapOutside = 0x80000000u;  // byte offset > any num_records the host admits
// The backward keeps ELEMENT offsets (one record addresses both `value` and the fp32 gradient
// image); x2 or x4 of this marker is still out of range and does not wrap 32 bits.
constexpr uint32_t kTapOutsideElem = 0x20000000u;

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// 4 consecutive channels of one tap as fp32
template <typename TV>
__device__ __forceinline__ float4_t load_tap(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off);

template <>
#ifndef VNX_TAP_AUX      // cache policy of the fp32 row gathers (A/B: 1 = sc0, 2 = nt, 16 = sc1; 0 = default).  Measured, forward, cold:
                         // headline 8.6 / 8.9 / 10.9 / 9.0 us for default / sc0 / nt / sc1, encoder-360p 54.2 / 54.3 / 87.4 / 58.3 us
#define VNX_TAP_AUX 0
#endif
__device__ __forceinline__ float4_t load_tap<float>(__amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
  uint4_t r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, int(off), 0, VNX_TAP_AUX);
  float4_t v;
  v.x = __uint_as_float(r.x); v.y = __uint_as_float(r.y);
  v.z = __uint_as_float(r.z); v.w = __uint_as_float(r.w);
  return v;
}
template <>
__device__ __forceinline__ float4_t load_tap<bf16_t>(__amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
  uint2_t r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, int(off), 0, 0);
  float4_t v;
  v.x = __uint_as_float(r.x << 16); v.y = __uint_as_float(r.x & 0xffff0000u);
  v.z = __uint_as_float(r.y << 16); v.w = __uint_as_float(r.y & 0xffff0000u);
  return v;
}
__device__ __forceinline__ float half_bits_to_float(uint32_t bits16) {
  return float(__builtin_bit_cast(_Float16, uint16_t(bits16)));
}
template <>
__device__ __forceinline__ float4_t load_tap<f16_t>(__amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
  uint2_t r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, int(off), 0, 0);
  float4_t v;
  v.x = half_bits_to_float(r.x & 0xffffu); v.y = half_bits_to_float(r.x >> 16);
  v.z = half_bits_to_float(r.y & 0xffffu); v.w = half_bits_to_float(r.y >> 16);
  return v;
}

template <typename TV>
__device__ __forceinline__ void store_row4(TV* p, float4_t v);
template <>
__device__ __forceinline__ void store_row4<float>(float* p, float4_t v) {
  __builtin_nontemporal_store(v, reinterpret_cast<float4_t*>(p));     // write-once output: see nt_load
}
template <>
__device__ __forceinline__ void store_row4<bf16_t>(bf16_t* p, float4_t v) {
  uint2_t r;
  r.x = f32x2_to_bf16x2(v.x, v.y);
  r.y = f32x2_to_bf16x2(v.z, v.w);
  __builtin_nontemporal_store(r, reinterpret_cast<uint2_t*>(p));
}
__device__ __forceinline__ uint32_t float_to_half_bits(float f) {
  return uint32_t(__builtin_bit_cast(uint16_t, _Float16(f)));
}
template <>
__device__ __forceinline__ void store_row4<f16_t>(f16_t* p, float4_t v) {
  uint2_t r;
  r.x = float_to_half_bits(v.x) | (float_to_half_bits(v.y) << 16);
  r.y = float_to_half_bits(v.z) | (float_to_half_bits(v.w) << 16);
  __builtin_nontemporal_store(r, reinterpret_cast<uint2_t*>(p));
}

// 16-bit rows, 4 lanes x 16 B (LPR == 4): one lane holds 8 consecutive channels = 4 dwords of two elements each
template <typename TV>
__device__ __forceinline__ void unpack8(uint4_t r, float4_t& lo, float4_t& hi);
template <>
__device__ __forceinline__ void unpack8<bf16_t>(uint4_t r, float4_t& lo, float4_t& hi) {
  lo = float4_t{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
  hi = float4_t{__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u), __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
}
template <>
__device__ __forceinline__ void unpack8<f16_t>(uint4_t r, float4_t& lo, float4_t& hi) {
  lo = float4_t{half_bits_to_float(r.x & 0xffffu), half_bits_to_float(r.x >> 16), half_bits_to_float(r.y & 0xffffu), half_bits_to_float(r.y >> 16)};
  hi = float4_t{half_bits_to_float(r.z & 0xffffu), half_bits_to_float(r.z >> 16), half_bits_to_float(r.w & 0xffffu), half_bits_to_float(r.w >> 16)};
}
template <>
__device__ __forceinline__ void unpack8<float>(uint4_t, float4_t&, float4_t&) {}      // never instantiated with LPR == 4
template <typename TV>
__device__ __forceinline__ uint4_t pack8(float4_t lo, float4_t hi);
template <>
__device__ __forceinline__ uint4_t pack8<bf16_t>(float4_t lo, float4_t hi) {
  return uint4_t{f32x2_to_bf16x2(lo.x, lo.y), f32x2_to_bf16x2(lo.z, lo.w), f32x2_to_bf16x2(hi.x, hi.y), f32x2_to_bf16x2(hi.z, hi.w)};
}
template <>
__device__ __forceinline__ uint4_t pack8<f16_t>(float4_t lo, float4_t hi) {
  return uint4_t{float_to_half_bits(lo.x) | (float_to_half_bits(lo.y) << 16), float_to_half_bits(lo.z) | (float_to_half_bits(lo.w) << 16),
                 float_to_half_bits(hi.x) | (float_to_half_bits(hi.y) << 16), float_to_half_bits(hi.z) | (float_to_half_bits(hi.w) << 16)};
}
template <>
__device__ __forceinline__ uint4_t pack8<float>(float4_t, float4_t) { return uint4_t{0u, 0u, 0u, 0u}; }

// -----------------------------------------------------------------------------
// forward
// -----------------------------------------------------------------------------
// LP = levels*points (template value 0 = use the runtime value, no unrolling).
// a / b for 0 <= a < 2^24, 0 < b < 2^24 without the integer-division sequence (one reciprocal,
// one multiply, an exact remainder check)
__device__ __forceinline__ int small_div(int a, int b) {
  int q = int(float(a) * __frcp_rn(float(b)));
  int r = a - q * b;
  if (r < 0) { --q; r += b; }
  if (r >= b) ++q;
  return q;
}
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {     // v_pk_min_u16
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b)));
}
// rows per unit of the grad_value kernels for a level of n pixels = gv_level_split(n, units_min, rows_max).rpu
// (vnx_common.h), here without the integer-division sequence (one lane per sample evaluates it)
__device__ __forceinline__ int gv_rows_per_unit(int n, int units_min, int rows_max) {
  int units = (n + rows_max - 1) / rows_max;
  if (units < units_min) units = units_min;
  if (units > n) units = n;
  return units > 0 ? small_div(n + units - 1, units) : 1;
}

// -----------------------------------------------------------------------------
// fused prologue
// -----------------------------------------------------------------------------
// The module computes  attention = softmax(Linear(q))  over the L*P samples of a (query, head)
// and  location = reference + offsets / (W_l, H_l)   (2-d references,
// projects/IDOL/idol/models/ops/modules/ms_deform_attn.py:102-105) or
// reference_xy + offsets / P * reference_wh * 0.5   (4-d, :106-108) in separate kernels and
// hands the op two tensors that exist only to be read once.  With FUSED the kernels take the
// two Linear outputs and the reference points instead: `loc` = raw offsets, `attn` = raw
// logits, and the softmax (a 16-lane row per (query, head): L*P == 16) and the location
// arithmetic happen in the one-lane-per-sample phase that decoded them anyway.
// `nt` (non-temporal) loads / stores.  Measured on MI355X, cold inputs (tools/kbench.hip, round 2):
//   * write-once outputs stored with `nt` (forward output, grad_value): decoder-360p backward 33.8 -> 31.4 us,
//     forward 9.14 -> 9.01, encoder forward 69 -> 65 us -- adopted everywhere;
//   * read-once locations / attention weights loaded with `nt`: decoder forward 9.09 -> 8.54 us (B = 10: 16.1 ->
//     15.1), but the ENCODER shape loses (65 -> 72 us), and the grad_loc kernel gains nothing at either shape
//     (nor do nt loads of grad_output / nt stores of its two gradients); cache-warm the decoder forward loses
//     too (6.1 -> 6.9 us: an nt line does not stay for the next replay of the same input -- an artefact of
//     replaying one input).  So only the forward does it, only in its small-call configuration (one wave per
//     workgroup, <= 4 096 rows), and at compile time: the same choice made by a run-time flag cost 0.3 us.
template <typename T> __device__ __forceinline__ T nt_load(const T* p) { return __builtin_nontemporal_load(p); }

// One lane's sample: its level's size and first pixel, its location and its attention weight -- FOUR requests issued
// together, one wait (fp32 locations).  Written as asm because the compiler sinks ordinary loads to their first use: the
// level start and the attention weight are used only inside the in-bounds branch, so they went out after the location had
// arrived -- the T = 5 forward ran locations -> (level start, attention weight) -> gathers, three dependent round trips
// instead of two, the grad_loc kernel locations -> shapes -> level start -> gathers (ISA, round 5).  The two geometry loads
// hit L2 and are requested first: a wave's loads return in order, the cold location load sets the pace.
template <bool NT>
__device__ __forceinline__ void load_sample_f32(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                                                const float* __restrict__ loc, const float* __restrict__ attn, int l, int64_t wi,
                                                int& H, int& W, int& start, float& x, float& y, float& a) {
  typedef int i4_t __attribute__((ext_vector_type(4)));
  typedef int i2_t __attribute__((ext_vector_type(2)));
  i4_t hw;
  i2_t st;
  float2_t xy;
  if constexpr (NT)
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx2 %1, %5, off\n\t"
                 "global_load_dwordx2 %2, %6, off nt\n\tglobal_load_dword %3, %7, off nt\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(hw), "=&v"(st), "=&v"(xy), "=&v"(a)
                 : "v"(shapes + 2 * l), "v"(lsi + l), "v"(loc + 2 * wi), "v"(attn + wi)
                 : "memory");
  else
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx2 %1, %5, off\n\t"
                 "global_load_dwordx2 %2, %6, off\n\tglobal_load_dword %3, %7, off\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(hw), "=&v"(st), "=&v"(xy), "=&v"(a)
                 : "v"(shapes + 2 * l), "v"(lsi + l), "v"(loc + 2 * wi), "v"(attn + wi)
                 : "memory");
  // (the wait is part of the SAME statement: between two statements the compiler would take the destination registers for
  //  valid and might copy or spill them before the data has landed -- ADVICE r5; nothing was overlapped between them anyway)
  H = hw.x; W = hw.z; start = st.x;      // int64 entries, narrowed as the reference does (cuh:276-277)
  x = xy.x; y = xy.y;
}

__device__ __forceinline__ float row16_max(float v) {
#pragma unroll
  for (int k = 1; k < 16; k <<= 1) v = fmaxf(v, __shfl_xor(v, k, 16));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
#pragma unroll
  for (int k = 1; k < 16; k <<= 1) v += __shfl_xor(v, k, 16);
  return v;
}

// -> location (x, y) in [0, 1] units, softmax weight a, and d(location)/d(offset) (sx, sy).
// Called by all 16 lanes of a (query, head) row together (`valid` is uniform over the row).
template <typename TL>
__device__ __forceinline__ void fused_decode(const TL* __restrict__ raw_off, const TL* __restrict__ raw_logit,
                                             const FusedArgs& fa, int64_t wi, int b, int q, int l, int H, int W,
                                             const MsdaDims& d, bool valid, float& x, float& y, float& a,
                                             float& sx, float& sy) {
  const float lg = valid ? to_acc(raw_logit[wi]) : 0.f;
  const float ex = expf(lg - row16_max(lg));
  a = ex / row16_sum(ex);
  x = y = sx = sy = 0.f;
  if (valid) {
    const int64_t r = ((int64_t(b / fa.ref_div) * d.Lq + q) * d.L + l) * fa.ref_dim;
    const float ox = to_acc(raw_off[2 * wi]), oy = to_acc(raw_off[2 * wi + 1]);
    if (fa.ref_dim == 2) {
      x = fused_ref<TL>(fa, r) + ox / float(W);
      y = fused_ref<TL>(fa, r + 1) + oy / float(H);
      sx = 1.f / float(W);
      sy = 1.f / float(H);
    } else {
      const float rw = fused_ref<TL>(fa, r + 2), rh = fused_ref<TL>(fa, r + 3), np = float(d.P);
      x = fused_ref<TL>(fa, r) + ox / np * rw * 0.5f;
      y = fused_ref<TL>(fa, r + 1) + oy / np * rh * 0.5f;
      sx = rw * 0.5f / np;
      sy = rh * 0.5f / np;
    }
  }
}

// VNX_FWD_WPE / VNX_K1_WPE: amdgpu_waves_per_eu hint (second __launch_bounds__ argument); the
// register-allocation and scheduling heuristics follow it, and the measured best is built in.
// VNX_FWD_BATCH: samples whose 4 row loads each are issued before the first use (4: 16 loads in flight)
#ifndef VNX_FWD_BATCH
#define VNX_FWD_BATCH 4
#endif
#ifndef VNX_K1_BATCH
#define VNX_K1_BATCH 4
#endif
#ifndef VNX_K1_BATCH_F32      // the 8-lanes-per-row fp32 path of the one-wave configuration (need not divide the samples of a group)
#define VNX_K1_BATCH_F32 VNX_K1_BATCH
#endif
// tile boxes of the tile-fed grad_value path per WAVE (QPW = 4 queries on large calls) instead of per workgroup (16): a
// tile's box is 4 + 12 pixels wide instead of 16 + 12, so fewer tiles meet a block (encoder backward, cold: 175.2 -> 169.1 us
// at 360p, 654.8 -> 625.8 us at 720p B = 5), this kernel loses its LDS step, the grad_value kernel scans four times the
// words (1 275 per level at 360p: two load rounds instead of one).  VNX_TILE_PER_WAVE=0: the per-workgroup form.
#ifndef VNX_TILE_PER_WAVE
#define VNX_TILE_PER_WAVE 1
#endif
constexpr bool kTilePerWave = VNX_TILE_PER_WAVE != 0;
// fp32 rows of the grad_loc kernel as 4 lanes x 32 B (A/B macro: 0 = never, 1 = large-call configurations, 2 = all)
#ifndef VNX_K1_F32_LPR4_MODE
#define VNX_K1_F32_LPR4_MODE 0
#endif
#define VNX_K1_F32_LPR4(WPB_) (VNX_K1_F32_LPR4_MODE == 2 || (VNX_K1_F32_LPR4_MODE == 1 && (WPB_) > 1))
#ifndef VNX_K1_BATCH_LARGE
#define VNX_K1_BATCH_LARGE 2
#endif
// Timing ablations of the grad_loc kernel (A/B builds of the development library only; wrong results by construction), a
// bit mask: 1 = no phase 3 (combination + gradient stores), 2 = no tile boxes, 4 = phase 2 without the 8-lane reductions
// and result writes (row loads + dots only), 8 = no zeroing of the query-split levels' rows.
#ifndef VNX_K1_ABL
#define VNX_K1_ABL 0
#endif
// Phase 3's stores: 0 = three plain 4-byte stores per sample (rounds 1-3); 1 = (x, y) as one 8-byte store; 2 = that, and
// both gradients non-temporal.  Measured (round 4, cold, grad_loc kernel alone / whole backward): encoder-360p 75.5 / 170.7 us
// (0), 75.6 / 171.2 (1), 64.4 / 165.0 (2); headline 11.4 / 26.6 -> 11.2 / 26.0; encoder-720p B = 5 623 -> 620.  The gradients
// are read much later, by the caller: streamed past the caches they no longer evict the rows this kernel gathers.  (Round 2
// tried `nt` on the sample RECORDS as well, which the grad_value kernel reads right away: that lost; boxes and records stay plain.)
#ifndef VNX_K1_P3
#define VNX_K1_P3 2
#endif
#ifndef VNX_FWD_WPE
#define VNX_FWD_WPE 0
#endif
#ifndef VNX_K1_WPE
#define VNX_K1_WPE 0
#endif
#if VNX_FWD_WPE > 0
#define VNX_FWD_BOUNDS(n) __launch_bounds__(n, VNX_FWD_WPE)
#else
#define VNX_FWD_BOUNDS(n) __launch_bounds__(n)
#endif
#if VNX_K1_WPE > 0
#define VNX_K1_BOUNDS(n) __launch_bounds__(n, VNX_K1_WPE)
#else
#define VNX_K1_BOUNDS(n) __launch_bounds__(n)
#endif
// PF: stream this wave's share of the head's rows towards L2 while the locations are in flight
// (see "phase 0" below); built for the one-pass case (QPW * L*P <= 64), unfused.
constexpr int kPfSteps = 3;       // LDS-DMA instructions per wave, 32 rows (2 x 64-B halves) each
// LPR = lanes per row.  8: a row is 8 x 16 B (fp32) or 8 x 8 B (16-bit values: the round-1 map, as many load
// instructions and L1 cycles as fp32 -- bf16 bought nothing).  4: 16-bit rows as 4 x 16 B -- a wave instruction
// covers 16 rows instead of 8, a sample's four taps cost a lane 4 loads of 16 B instead of 8 of 8 B (L*P == 16 only).
template <typename TV, typename TL, int QPW, int WPB, int LP_T, bool FUSED = false, bool PF = false, int LPR = 8>
__global__ void VNX_FWD_BOUNDS(64 * WPB)
msda_fwd_d32_kernel(const TV* __restrict__ value, const int64_t* __restrict__ shapes,
                    const int64_t* __restrict__ lsi, const TL* __restrict__ loc,
                    const TL* __restrict__ attn, TV* __restrict__ out, MsdaDims d,
                    int tiles_per_batch, int prefetch_rows, unsigned long long* stamps, FusedArgs fa) {
  static_assert(!FUSED || LP_T == 16, "the fused prologue is built for L*P == 16");
  static_assert(LPR == 8 || (LPR == 4 && sizeof(TV) == 2 && LP_T == 16 && !PF), "4 lanes per row: 16-bit values, L*P == 16");
  stamp_begin(stamps);
  constexpr int D = 32;
  constexpr int PG = (64 / LPR) / QPW;   // sample groups per query
  constexpr int kRowBytes = D * int(sizeof(TV));
  constexpr int kLaneBytes = kRowBytes / LPR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int LP = LP_T > 0 ? LP_T : d.L * d.P;
  const int p_shift = __builtin_ctz(uint32_t(d.P) | 0x10000u);      // log2(P) when P is a power of two (L * P == 16)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int head_rot = (prefetch_rows >> 16) & 0xf;   // A/B record (variants 61..68): which head runs on which XCD
  prefetch_rows &= 0xffff;
  const int tile = blockIdx.x / d.M;
  const int b = tile / tiles_per_batch;
  // Which head runs where: blockIdx % heads = the XCD (observed placement), and the head handled there
  // rotates with the batch element.  Each XCD's L2 still holds one head-slice per batch element, but no
  // XCD is the only requester of one address class: with `value` rows of head h at byte offset 128 h of
  // every 1-KiB pixel record, the rows at offset 384 (mod 1 KiB) are served ~35 % slower than the other
  // seven classes when all eight stream at once from one XCD each (measured: that head's workgroups end
  // at 8.6 us mean / 11.1 max, all others at 6.3 / 7.9; the slow class follows the ADDRESS when the tensor
  // is shifted by 128 B, not the XCD; alone it is as fast as the rest).  Spread over the XCDs the same
  // rows cost the T=5 call 9.1 instead of 11.1 us (B = 8: 11.5 instead of 14.2).  head_rot 14 = fixed map.
  const int m = (blockIdx.x % d.M + (head_rot == 14 ? 0 : head_rot == 15 || head_rot == 0 ? b : head_rot)) % d.M;
  const int q0 = (tile - b * tiles_per_batch) * (QPW * WPB) + wave * QPW;

  // per-wave LDS: [QPW][LP+1] tap offsets (uint4) then [QPW][LP+1] tap weights (float4)
  const int ent = QPW * (LP + 1);
  uint4_t* s_off = reinterpret_cast<uint4_t*>(smem) + size_t(wave) * 2 * ent;
  float4_t* s_wt = reinterpret_cast<float4_t*>(s_off + ent);

  const int pixel_bytes = d.M * kRowBytes;

  const TV* head_base = value + (int64_t(b) * d.S * d.M + m) * D;
  const uint32_t head_bytes = uint32_t((int64_t(d.S) * d.M - m) * kRowBytes);
  const __amdgpu_buffer_rsrc_t rsrc = uniform_rsrc(head_base, head_bytes);

  // ---- phase 0 (optional): stream this wave's share of the head's rows towards L2 --------
  // A decoder call with scattered locations on a small map reads nearly every row of `value`
  // (3.8 taps per row at the T=5 360p shape), but only learns WHICH rows after a first HBM round
  // trip for the locations -- and then every wave of the grid asks for its taps at the same moment.
  // Cold, the kernel is the sum of those two phases (10.9 us = 5.6 us cache-warm + 25 MB / 4.7 TB/s),
  // not their maximum.  With PF the wave first issues its location loads, then touches both 64-B
  // halves of its share of the head's rows in address order with `buffer_load_dword ... lds` (no
  // VGPR destination, nothing ever waits for the data; it lands in a 256-B dump behind the tap
  // records), and only then waits for the locations: HBM streams `value` into L2 while the
  // locations travel.  The loads are unconditional and counted (kPfSteps), so the compiler's own
  // s_waitcnt for the locations is vmcnt(kPfSteps), not vmcnt(0).
  static_assert(!PF || (!FUSED && LP_T > 0 && QPW * LP_T <= 64), "PF: one-pass, unfused configurations only");
  float pf_x = 0.f, pf_y = 0.f, pf_a = 0.f;
  if constexpr (PF) {
    const int qi = lane / LP_T, p = lane - qi * LP_T;
    const int q = q0 + qi < d.Lq ? q0 + qi : d.Lq - 1;         // clamped: the loads below are unconditional
    const int64_t wi = ((int64_t(b) * d.Lq + q) * d.M + m) * LP_T + p;
    // hipcc sinks ordinary loads from `const __restrict__` memory below anything, compiler barriers
    // included; the location loads therefore go out as asm (issued HERE), and are waited for by hand
    // below: vmcnt(kPfSteps) = "everything older than the stream has landed".
    static_assert(sizeof(TL) == 4, "PF: fp32 locations");
    float2_t pf_xy;
    asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dword %1, %3, off"
                 : "=&v"(pf_xy), "=&v"(pf_a)
                 : "v"(loc + 2 * wi), "v"(attn + wi)
                 : "memory");
    const int first = ((tile - b * tiles_per_batch) * WPB + wave) * prefetch_rows;
    unsigned char* dump = smem + size_t(WPB) * 2 * ent * 16 + size_t(wave) * 256;
#pragma unroll
    for (int i = 0; i < kPfSteps; ++i) {
      const int r = i * 32 + (lane >> 1);
      const int row = first + r;
      const uint32_t off = (r < prefetch_rows && row < d.S)
                               ? uint32_t(row) * uint32_t(pixel_bytes) + uint32_t(lane & 1) * (kRowBytes / 2)
                               : kTapOutside;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dump, 4, off, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(pf_xy), "+v"(pf_a) : "n"(kPfSteps) : "memory");
    pf_x = pf_xy.x; pf_y = pf_xy.y;
  }

  // ---- phase 1: one (query, sample) pair per lane and step --------------------
  const int pairs = QPW * LP;
  constexpr bool kOneRequest = !PF && !FUSED && sizeof(TL) == 4 && LP_T > 0 && QPW * LP_T <= 64;      // see load_sample_f32
  for (int e = lane; e < pairs; e += 64) {
    const int qi = e / LP, p = e - qi * LP;
    const int q = q0 + qi;
    uint4_t o4 = {kTapOutside, kTapOutside, kTapOutside, kTapOutside};
    float4_t w4 = {0.f, 0.f, 0.f, 0.f};
    const int l = LP_T == 16 ? (p >> p_shift) : p / d.P;       // L * P == 16: P is a power of two
    const int64_t wi = ((int64_t(b) * d.Lq + q) * d.M + m) * LP + p;
    float x = 0.f, y = 0.f, a = 0.f;
    if constexpr (FUSED) {
      float sx, sy;
      const bool valid = q < d.Lq;
      fused_decode<TL>(loc, attn, fa, wi, b, q, l, valid ? int(shapes[2 * l]) : 1, valid ? int(shapes[2 * l + 1]) : 1,
                       d, valid, x, y, a, sx, sy);
    }
    if (q < d.Lq) {
      if constexpr (PF) {
        x = pf_x; y = pf_y; a = pf_a;
      } else if constexpr (!FUSED) {
        if constexpr (kOneRequest) {
          // (filled below, together with the level geometry: load_sample_f32)
        } else if constexpr (sizeof(TL) == 4) {
          if constexpr (WPB == 1) {
            x = to_acc(nt_load(loc + 2 * wi)); y = to_acc(nt_load(loc + 2 * wi + 1));
            a = to_acc(nt_load(attn + wi));
          } else {
            x = to_acc(loc[2 * wi]); y = to_acc(loc[2 * wi + 1]);
            a = to_acc(attn[wi]);
          }
        } else {
          x = to_acc(loc[2 * wi]); y = to_acc(loc[2 * wi + 1]);
          a = to_acc(attn[wi]);
        }
      }
      int H, W, start;
      if constexpr (kOneRequest) {      // one (query, sample) per lane, fp32 locations: geometry, location and weight in one request
        load_sample_f32<WPB == 1>(shapes, lsi, reinterpret_cast<const float*>(loc), reinterpret_cast<const float*>(attn), l, wi,
                                  H, W, start, x, y, a);      // (WPB == 1: the small-call configuration, `nt` loads, see nt_load)
      } else {
        H = int(shapes[2 * l]); W = int(shapes[2 * l + 1]);
        start = int(lsi[l]);
      }
      const float h = y * float(H) - 0.5f, w = x * float(W) - 0.5f;
      if (h > -1.f && w > -1.f && h < float(H) && w < float(W)) {
        const float hf = floorf(h), wf = floorf(w);
        const int h0 = int(hf), w0 = int(wf);
        const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw = 1.f - lw;
        const bool top = h0 >= 0, bot = h0 + 1 <= H - 1, lef = w0 >= 0, rig = w0 + 1 <= W - 1;
        const uint32_t pb = uint32_t(pixel_bytes);
        // 24-bit multiplies (full rate; v_mul_lo_u32 issues at a quarter of it): level sides and the pixel index stay
        // below 2^23 (msda_d32_fwd_supported), the low 32 bits of the signed 48-bit product are the value mod 2^32
        const uint32_t o00 = uint32_t(__mul24(start + __mul24(h0, W) + w0, int(pb)));  // mod 2^32 on purpose
        const uint32_t row_b = __umul24(uint32_t(W), pb);
        o4.x = (top && lef) ? o00 : kTapOutside;
        o4.y = (top && rig) ? o00 + pb : kTapOutside;
        o4.z = (bot && lef) ? o00 + row_b : kTapOutside;
        o4.w = (bot && rig) ? o00 + row_b + pb : kTapOutside;
        w4.x = a * (hh * hw); w4.y = a * (hh * lw); w4.z = a * (lh * hw); w4.w = a * (lh * lw);
      }
    }
    s_off[qi * (LP + 1) + p] = o4;
    s_wt[qi * (LP + 1) + p] = w4;
  }
  if (WPB > 1) __syncthreads(); else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- phase 2: gather ----------------------------------------------------------
  const int ch = lane & (LPR - 1);
  const int qi = (lane / LPR) % QPW;
  const int pg = lane / (LPR * QPW);
  const int q = q0 + qi;

  const int per_group = LP / PG;  // host guarantees LP % PG == 0
  const uint4_t* g_off = s_off + qi * (LP + 1) + pg * per_group;
  const float4_t* g_wt = s_wt + qi * (LP + 1) + pg * per_group;
  const uint32_t lane_off = uint32_t(ch * kLaneBytes);

  if constexpr (LPR == 4) {
    // 16-bit rows, 8 channels per lane: the taps stay packed (4 dwords) until they are used
    constexpr int kPer = LP_T / PG;
    constexpr int kWant = WPB == 1 ? VNX_FWD_BATCH : 2;
    constexpr int kBatch = kPer < kWant ? kPer : kWant;
    static_assert(kPer % kBatch == 0, "whole batches");
    float4_t acc_lo = {0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
#pragma unroll
    for (int i0 = 0; i0 < kPer; i0 += kBatch) {
      uint4_t o[kBatch];
      float4_t w[kBatch];
      uint4_t raw[kBatch][4];
#pragma unroll
      for (int j = 0; j < kBatch; ++j) { o[j] = g_off[i0 + j]; w[j] = g_wt[i0 + j]; }
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        raw[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, int(o[j].x + lane_off), 0, 0);
        raw[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, int(o[j].y + lane_off), 0, 0);
        raw[j][2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, int(o[j].z + lane_off), 0, 0);
        raw[j][3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, int(o[j].w + lane_off), 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        const float wt[4] = {w[j].x, w[j].y, w[j].z, w[j].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float4_t lo, hi;
          unpack8<TV>(raw[j][t], lo, hi);
          acc_lo += wt[t] * lo;
          acc_hi += wt[t] * hi;
        }
      }
    }
#pragma unroll
    for (int off = LPR * QPW; off < 64; off <<= 1) {
      acc_lo.x += __shfl_xor(acc_lo.x, off, 64); acc_lo.y += __shfl_xor(acc_lo.y, off, 64);
      acc_lo.z += __shfl_xor(acc_lo.z, off, 64); acc_lo.w += __shfl_xor(acc_lo.w, off, 64);
      acc_hi.x += __shfl_xor(acc_hi.x, off, 64); acc_hi.y += __shfl_xor(acc_hi.y, off, 64);
      acc_hi.z += __shfl_xor(acc_hi.z, off, 64); acc_hi.w += __shfl_xor(acc_hi.w, off, 64);
    }
    if (pg == 0 && q < d.Lq) {
      TV* o = out + ((int64_t(b) * d.Lq + q) * d.M + m) * D + ch * 8;
      __builtin_nontemporal_store(pack8<TV>(acc_lo, acc_hi), reinterpret_cast<uint4_t*>(o));
    }
    stamp_end(stamps);
    return;
  }

  float4_t acc = {0.f, 0.f, 0.f, 0.f};
  if constexpr (LP_T > 0 && QPW <= 4) {
    // The gathers are latency-bound (two dependent HBM round trips per wave: locations, then
    // taps), so put a whole batch of 4 samples = 16 row loads in flight before the first use.
    constexpr int kPer = LP_T / PG;
    // 4 samples on the small calls (one wave per workgroup, 3 waves per SIMD: latency is everything); 2 on the large
    // ones, where 55 instead of 95 VGPRs = 8 instead of 5 waves per SIMD matter more (encoder 360p: 59.9 -> 54.8 us)
    constexpr int kWant = WPB == 1 ? VNX_FWD_BATCH : 2;
    constexpr int kBatch = kPer < kWant ? kPer : kWant;
#pragma unroll
    for (int i0 = 0; i0 < kPer; i0 += kBatch) {
      uint4_t o[kBatch];
      float4_t w[kBatch];
      float4_t v[kBatch][4];
#pragma unroll
      for (int j = 0; j < kBatch; ++j) { o[j] = g_off[i0 + j]; w[j] = g_wt[i0 + j]; }
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        v[j][0] = load_tap<TV>(rsrc, o[j].x + lane_off);
        v[j][1] = load_tap<TV>(rsrc, o[j].y + lane_off);
        v[j][2] = load_tap<TV>(rsrc, o[j].z + lane_off);
        v[j][3] = load_tap<TV>(rsrc, o[j].w + lane_off);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        acc += w[j].x * v[j][0];
        acc += w[j].y * v[j][1];
        acc += w[j].z * v[j][2];
        acc += w[j].w * v[j][3];
      }
    }
  } else {
    // 8 queries per wave: the compiler's own schedule (8 loads in flight, 63 VGPRs, 8 waves per
    // SIMD) measures better than wider batches on query-rich calls
    constexpr int kUnroll = LP_T > 0 ? LP_T / PG : 1;
#pragma unroll kUnroll
    for (int i = 0; i < per_group; ++i) {
      const uint4_t o = g_off[i];
      const float4_t w = g_wt[i];
      const float4_t v0 = load_tap<TV>(rsrc, o.x + lane_off);
      const float4_t v1 = load_tap<TV>(rsrc, o.y + lane_off);
      const float4_t v2 = load_tap<TV>(rsrc, o.z + lane_off);
      const float4_t v3 = load_tap<TV>(rsrc, o.w + lane_off);
      acc += w.x * v0;
      acc += w.y * v1;
      acc += w.z * v2;
      acc += w.w * v3;
    }
  }

  // sum the sample groups of a query (lanes that differ only in pg)
#pragma unroll
  for (int off = 8 * QPW; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off, 64);
    acc.y += __shfl_xor(acc.y, off, 64);
    acc.z += __shfl_xor(acc.z, off, 64);
    acc.w += __shfl_xor(acc.w, off, 64);
  }
  if (pg == 0 && q < d.Lq) {
    TV* o = out + ((int64_t(b) * d.Lq + q) * d.M + m) * D + ch * 4;
    store_row4<TV>(o, acc);
  }
  stamp_end(stamps);
}

// ---- forward of calls with MANY queries (the encoders'): the coarse levels staged whole in LDS --------------------------------
// The gather kernel above is bound by what a CU's vector L1 delivers -- 64 B per clock; a sample is four 128-byte rows: 1.67 GB
// at encoder-360p, 48.5 us at 2.1 GHz of a 55-us launch -- while the LDS, which delivers twice that, sits
// idle.  The two LDS-staged forwards of rounds 2-3 (tools/experiments/msda_tile) staged data-dependent WINDOWS of every level and
// paid for them in instructions (bounding boxes, far taps, window bookkeeping: 5.8-7.3 vector instructions per sample against
// 4.9).  This kernel stages no window: every level of the pyramid receives the same number of taps, but the two coarsest
// ones are 6 % of the pixels -- 300 rows of one head, 38 KB in fp32, at 360p -- so a workgroup copies them WHOLE into LDS
// once and then walks many query tiles of its (batch, head): the taps of the staged levels are `ds_read_b128` at an offset the
// decode computes exactly as it computes the global one (zero padding = one zero row), the taps of the fine levels are the
// gathers of the kernel above.  No boxes, no far path, no per-tile staging, the same instruction count per sample -- and half
// of the bytes leave the L1's queue for a pipe that was unused.
// Which levels are staged is decided on the device (the host knows S, not the level sizes): the longest suffix of the packed
// levels that fits kSlabRowsCap rows; none (unpacked levels, or a pyramid whose coarsest level is larger): every tap is a gather.
// Workgroup = 8 waves, a wave = 8 queries x 8 lanes (one 8-lane set walks all 16 samples of its query: no cross-set sum);
// LDS = slab + zero row + the waves' sample records: 76 KB, two workgroups per CU.  fp32 values, 4 levels x 4 points.
constexpr int kSlabWaves = 8, kSlabQpw = 8, kSlabRowsCap = 320;
constexpr int kSlabEnt = kSlabQpw * 17;                                             // records per wave: [8 queries][16 + 1]
constexpr size_t slab_rec_bytes(int waves) { return size_t(waves) * 2 * kSlabEnt * 16; }      // offsets (uint4) + weights (float4)
constexpr size_t slab_lds_bytes(int waves, int cap128) { return size_t(cap128 + 1) * 128 + slab_rec_bytes(waves) + 64; }
constexpr size_t kSlabRecBytes = slab_rec_bytes(kSlabWaves);
constexpr size_t kSlabLdsBytes = slab_lds_bytes(kSlabWaves, kSlabRowsCap);
// Round 6, the LARGE slab for 16-bit values at 720p: there the two coarsest levels are 1 160 rows of 64 bytes = 74 KB -- with
// them staged half of the taps leave the L1's queue instead of a quarter (the small slab holds level 3 only; in fp32 the two
// levels are 148 KB and do not fit beside anything).  One workgroup of SIXTEEN waves per CU (the same four waves per SIMD as
// two workgroups of eight): slab 77 KB + records 70 KB = 147 KB.
constexpr int kSlabWavesL = 16, kSlabRowsCapL = 600;      // 600 x 128 B = 1 200 rows of 64 B (+ the zero row)
static_assert(slab_lds_bytes(kSlabWavesL, kSlabRowsCapL) <= 160 * 1024, "the large slab's workgroup fits a CU's LDS");

// a staged row's four channels of this lane, as fp32 (fp32 rows: 16 bytes; 16-bit rows: 8)
template <typename TV> __device__ __forceinline__ float4_t slab_tap(const unsigned char* p);
template <> __device__ __forceinline__ float4_t slab_tap<float>(const unsigned char* p) { return *reinterpret_cast<const float4_t*>(p); }
template <> __device__ __forceinline__ float4_t slab_tap<bf16_t>(const unsigned char* p) {
  const uint2_t r = *reinterpret_cast<const uint2_t*>(p);
  return float4_t{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}
template <> __device__ __forceinline__ float4_t slab_tap<f16_t>(const unsigned char* p) {
  const uint2_t r = *reinterpret_cast<const uint2_t*>(p);
  return float4_t{half_bits_to_float(r.x & 0xffffu), half_bits_to_float(r.x >> 16), half_bits_to_float(r.y & 0xffffu), half_bits_to_float(r.y >> 16)};
}

// Round 6: TV = the value's (and the output's) element type.  16-bit rows are 64 bytes, so the slab region of the fp32 layout
// (41 KB) holds 641 of them -- the same two coarsest levels at 360p (300 rows, 19 KB) and level 3 at 720p -- a staged tap is a
// `ds_read_b64`, a gathered one the 8-byte load of the gather kernel's 8-lane map.  Until round 6 calls with 16-bit values (every
// encoder layer of a model under bf16 autocast) kept the gather kernel.
template <typename TV, typename TL, bool FUSED, int WAVES = kSlabWaves, int CAP128 = kSlabRowsCap>
__global__ void __launch_bounds__(64 * WAVES, 4)      // 4 waves per SIMD (two workgroups of 8 waves per CU, or one of 16), <= 128 VGPRs
msda_fwd_slab_kernel(const TV* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                     const TL* __restrict__ loc, const TL* __restrict__ attn, TV* __restrict__ out, MsdaDims d,
                     int parts, unsigned long long* stamps, FusedArgs fa) {
  stamp_begin(stamps);
  constexpr int D = 32, LP = 16, kRowBytes = D * int(sizeof(TV)), kPieces = kRowBytes / 16, kLaneBytes = kRowBytes / 8;
  constexpr int kRowsCap = (CAP128 + 1) * 128 / kRowBytes - 1;      // rows the slab region holds (+ the zero row)
  constexpr int kSlabWaves = WAVES;                                 // (shadows the namespace's: the workgroup's waves)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* slab = smem;                                                        // [rows + 1][row bytes]
  uint4_t* rec_base = reinterpret_cast<uint4_t*>(smem + size_t(CAP128 + 1) * 128);
  int* s_lvl = reinterpret_cast<int*>(smem + size_t(CAP128 + 1) * 128 + slab_rec_bytes(WAVES));      // [4][4]: H, W, start, -
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef VNX_SLAB_BATCH_MAJOR
#define VNX_SLAB_BATCH_MAJOR 1
#endif
  // batch-major: the workgroups resident at one time (64 per XCD) belong to one or two batch elements, whose rows of the XCD's
  // head then stay in its 4-MB L2 (2.5 MB per (batch, head) at 720p; with all five batch elements in flight the kernel fetched
  // 929 MB for 100 MB of values -- PMC, profiles/r05_shapes_kernel_avg_us.json)
  const int x = blockIdx.x % d.M, rest = blockIdx.x / d.M;
  const int b = VNX_SLAB_BATCH_MAJOR ? rest / parts : rest % d.B, part = VNX_SLAB_BATCH_MAJOR ? rest % parts : rest / d.B;
  const int m = (x + b) % d.M;      // head <-> XCD map rotating with the batch element, as the gather kernel

  // ---- the level table; which levels are staged ----
  if (tid < 4) {
    s_lvl[4 * tid] = int(shapes[2 * tid]); s_lvl[4 * tid + 1] = int(shapes[2 * tid + 1]); s_lvl[4 * tid + 2] = int(lsi[tid]);
  }
  __syncthreads();
  int first_staged = 4, slab_first = d.S;      // levels [first_staged, 4) live in LDS, rows [slab_first, S) of the map
  {
    bool packed = true;
    int running = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) { packed = packed && s_lvl[4 * l + 2] == running; running += s_lvl[4 * l] * s_lvl[4 * l + 1]; }
    packed = packed && running == d.S;
    if (packed) {
#pragma unroll
      for (int l = 3; l >= 0; --l)
        if (d.S - s_lvl[4 * l + 2] <= kRowsCap) { first_staged = l; slab_first = s_lvl[4 * l + 2]; }
    }
  }
  first_staged = __builtin_amdgcn_readfirstlane(first_staged);
  slab_first = __builtin_amdgcn_readfirstlane(slab_first);
  const int n_slab = d.S - slab_first;
  {
    const TV* src = value + ((int64_t(b) * d.S + slab_first) * d.M + m) * D;
    for (int i = tid; i < n_slab * kPieces; i += 64 * kSlabWaves)      // 16-byte pieces
      *reinterpret_cast<uint4_t*>(slab + (i / kPieces) * kRowBytes + (i % kPieces) * 16) =
          *reinterpret_cast<const uint4_t*>(src + int64_t(i / kPieces) * d.M * D + (i % kPieces) * (16 / int(sizeof(TV))));
    if (tid < kPieces) *reinterpret_cast<uint4_t*>(slab + n_slab * kRowBytes + tid * 16) = uint4_t{0u, 0u, 0u, 0u};
  }
  __syncthreads();
  const uint32_t zero_row = uint32_t(n_slab) * kRowBytes;

  uint4_t* s_off = rec_base + size_t(wave) * 2 * kSlabEnt;
  float4_t* s_wt = reinterpret_cast<float4_t*>(s_off + kSlabEnt);
  const int pixel_bytes = d.M * kRowBytes;
  const TV* head_base = value + (int64_t(b) * d.S * d.M + m) * D;
  const __amdgpu_buffer_rsrc_t rsrc = uniform_rsrc(head_base, uint32_t((int64_t(d.S) * d.M - m) * kRowBytes));
  const int ch = lane & 7, qi2 = lane >> 3;
  const uint32_t lane_off = uint32_t(ch * kLaneBytes);

  const int n_tiles = (d.Lq + kSlabWaves * kSlabQpw - 1) / (kSlabWaves * kSlabQpw);
  // The locations and weights of a tile's two decode passes are REQUESTED one tile ahead (plain calls, fp32: six registers) --
  // after the previous tile's decode, before its gathers -- so that a tile starts with its samples in hand instead of with a
  // trip to memory: the kernel is bound by its waves' waits (four per SIMD), not by a pipe.  (asm: the compiler would sink the
  // loads to their use; a wave's loads return in order, so the waits it generates for the gathers behind them stay valid.)
  constexpr bool kAhead = !FUSED && sizeof(TL) == 4;
  float2_t pre_xy[2] = {{0.f, 0.f}, {0.f, 0.f}};
  float pre_a[2] = {0.f, 0.f};
  auto request_samples = [&](int tile) {
    if constexpr (kAhead) {
      const int tq0 = (tile * kSlabWaves + wave) * kSlabQpw;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = 64 * k + lane, q = tq0 + (e >> 4);
        const int qc = q < d.Lq ? q : d.Lq - 1;      // clamped: the loads are unconditional
        const int64_t wi = ((int64_t(b) * d.Lq + qc) * d.M + m) * LP + (e & 15);
        asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dword %1, %3, off"
                     : "=&v"(pre_xy[k]), "=&v"(pre_a[k])
                     : "v"(reinterpret_cast<const float*>(loc) + 2 * wi), "v"(reinterpret_cast<const float*>(attn) + wi)
                     : "memory");
      }
    }
  };
  if (part < n_tiles) request_samples(part);
  for (int t = part; t < n_tiles; t += parts) {
    const int q0 = (t * kSlabWaves + wave) * kSlabQpw;
    if constexpr (kAhead)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre_xy[0]), "+v"(pre_xy[1]), "+v"(pre_a[0]), "+v"(pre_a[1]) : : "memory");
    // ---- phase 1: one (query, sample) per lane and step (cuh:253-298) ----
#pragma unroll
    for (int e0 = 0; e0 < kSlabQpw * LP; e0 += 64) {
      const int e = e0 + lane, qi = e >> 4, p = e & 15, l = p >> 2;
      const int q = q0 + qi;
      uint4_t o4 = {kTapOutside, kTapOutside, kTapOutside, kTapOutside};
      float4_t w4 = {0.f, 0.f, 0.f, 0.f};
      const bool staged = l >= first_staged;
      if (staged) o4 = uint4_t{zero_row, zero_row, zero_row, zero_row};
      const int64_t wi = ((int64_t(b) * d.Lq + q) * d.M + m) * LP + p;
      const int H = s_lvl[4 * l], W = s_lvl[4 * l + 1], start = s_lvl[4 * l + 2];
      float sx = 0.f, sy = 0.f, a = 0.f;
      if constexpr (FUSED) {
        float gx, gy;
        fused_decode<TL>(loc, attn, fa, wi, b, q, l, q < d.Lq ? H : 1, q < d.Lq ? W : 1, d, q < d.Lq, sx, sy, a, gx, gy);
      } else if constexpr (kAhead) {
        sx = pre_xy[e0 >> 6].x; sy = pre_xy[e0 >> 6].y; a = pre_a[e0 >> 6];
      } else if (q < d.Lq) {
        sx = to_acc(loc[2 * wi]); sy = to_acc(loc[2 * wi + 1]); a = to_acc(attn[wi]);
      }
      if (q < d.Lq) {
        const float h = sy * float(H) - 0.5f, w = sx * float(W) - 0.5f;
        if (h > -1.f && w > -1.f && h < float(H) && w < float(W)) {
          const float hf = floorf(h), wf = floorf(w);
          const int h0 = int(hf), w0 = int(wf);
          const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw = 1.f - lw;
          const bool top = h0 >= 0, bot = h0 + 1 <= H - 1, lef = w0 >= 0, rig = w0 + 1 <= W - 1;
          // a pixel's row: byte offset into the head's rows of `value` (gathers) or into the slab (staged levels)
          const uint32_t pb = staged ? uint32_t(kRowBytes) : uint32_t(pixel_bytes);
          const uint32_t none = staged ? zero_row : kTapOutside;
          const uint32_t o00 = uint32_t(__mul24((staged ? start - slab_first : start) + __mul24(h0, W) + w0, int(pb)));  // mod 2^32 on purpose
          const uint32_t row_b = __umul24(uint32_t(W), pb);
          o4.x = (top && lef) ? o00 : none;
          o4.y = (top && rig) ? o00 + pb : none;
          o4.z = (bot && lef) ? o00 + row_b : none;
          o4.w = (bot && rig) ? o00 + row_b + pb : none;
          w4.x = a * (hh * hw); w4.y = a * (hh * lw); w4.z = a * (lh * hw); w4.w = a * (lh * lw);
        }
      }
      s_off[qi * 17 + p] = o4;
      s_wt[qi * 17 + p] = w4;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (t + parts < n_tiles) request_samples(t + parts);      // uniform

    // ---- phase 2: an 8-lane set walks the 16 samples of its query, level by level: gathers or LDS reads ----
    // (Counters at encoder-360p, profiles/r05_backward_pmc.csv: 3.1 vector instructions per sample -- the gather kernel has 4.9 --
    //  46 % of the issue slots, LDS 45 % busy with 17 % conflicts, 53 % of the wave cycles waiting.  Measured on top of this
    //  form and dropped: the staged level's samples read and used while the gathered level's rows are in flight (44.9 against
    //  42.6 us: the second path costs the first its registers), eight steps of two samples with the next step's rows requested
    //  before this step's are used (spills), the rows of a sample's left / right taps ordered by row parity so that the two
    //  sets the LDS serves together never meet in a bank (43.2: the conflicts are not what the kernel waits for); and the
    //  gather kernel's lane map -- 4 queries per wave, the lower half of a wave gathering levels 0, 1 while the upper half reads
    //  levels 2, 3 from LDS, 12 waves per workgroup, six per SIMD -- 61.6 us: each kind of load is issued with half of its
    //  lanes masked.)
    const uint4_t* g_off = s_off + qi2 * 17;
    const float4_t* g_wt = s_wt + qi2 * 17;
    if constexpr (sizeof(TV) == 2) {
      // 16-bit rows: a row is 4 lanes x 16 bytes (8 channels per lane), so the 8-lane set of a query is two 4-lane halves that
      // take points {0, 1} and {2, 3} of every level: 8 loads of 16 bytes per lane and level instead of 16 of 8 bytes -- the
      // gather kernel's map for 16-bit rows (sixteen rows per wave instruction).  With 8 lanes x 8 bytes this kernel LOST to the
      // gather kernel on 16-bit values (encoder-360p bf16, cold: 43.9 against 39.5 us unfused, 56.0 against 50.5 fused).
      const int sub = (lane >> 2) & 1, c4 = lane & 3;
      const uint32_t off16 = uint32_t(c4 * 16);
      float4_t acc_lo = {0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
#pragma unroll 1
      for (int l = 0; l < 4; ++l) {
        uint4_t o[2];
        float4_t w[2];
        uint4_t raw[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) { o[j] = g_off[4 * l + 2 * sub + j]; w[j] = g_wt[4 * l + 2 * sub + j]; }
        if (l < first_staged) {      // uniform
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            raw[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, int(o[j].x + off16), 0, 0);
            raw[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, int(o[j].y + off16), 0, 0);
            raw[j][2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, int(o[j].z + off16), 0, 0);
            raw[j][3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, int(o[j].w + off16), 0, 0);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            raw[j][0] = *reinterpret_cast<const uint4_t*>(slab + o[j].x + off16);
            raw[j][1] = *reinterpret_cast<const uint4_t*>(slab + o[j].y + off16);
            raw[j][2] = *reinterpret_cast<const uint4_t*>(slab + o[j].z + off16);
            raw[j][3] = *reinterpret_cast<const uint4_t*>(slab + o[j].w + off16);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float wt[4] = {w[j].x, w[j].y, w[j].z, w[j].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float4_t lo, hi;
            unpack8<TV>(raw[j][k], lo, hi);
            acc_lo += wt[k] * lo;
            acc_hi += wt[k] * hi;
          }
        }
      }
      // the two halves of the set meet (lane ^ 4)
      acc_lo.x += __shfl_xor(acc_lo.x, 4, 64); acc_lo.y += __shfl_xor(acc_lo.y, 4, 64);
      acc_lo.z += __shfl_xor(acc_lo.z, 4, 64); acc_lo.w += __shfl_xor(acc_lo.w, 4, 64);
      acc_hi.x += __shfl_xor(acc_hi.x, 4, 64); acc_hi.y += __shfl_xor(acc_hi.y, 4, 64);
      acc_hi.z += __shfl_xor(acc_hi.z, 4, 64); acc_hi.w += __shfl_xor(acc_hi.w, 4, 64);
      const int q = q0 + qi2;
      if (q < d.Lq && sub == 0)
        *reinterpret_cast<uint4_t*>(out + ((int64_t(b) * d.Lq + q) * d.M + m) * D + c4 * 8) = pack8<TV>(acc_lo, acc_hi);
    } else {
    float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {      // (not unrolled: one level's sixteen rows in registers at a time)
      uint4_t o[4];
      float4_t w[4];
      float4_t v[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] = g_off[4 * l + j]; w[j] = g_wt[4 * l + j]; }
      if (l < first_staged) {      // uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j][0] = load_tap<TV>(rsrc, o[j].x + lane_off);
          v[j][1] = load_tap<TV>(rsrc, o[j].y + lane_off);
          v[j][2] = load_tap<TV>(rsrc, o[j].z + lane_off);
          v[j][3] = load_tap<TV>(rsrc, o[j].w + lane_off);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j][0] = slab_tap<TV>(slab + o[j].x + lane_off);
          v[j][1] = slab_tap<TV>(slab + o[j].y + lane_off);
          v[j][2] = slab_tap<TV>(slab + o[j].z + lane_off);
          v[j][3] = slab_tap<TV>(slab + o[j].w + lane_off);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc += w[j].x * v[j][0];
        acc += w[j].y * v[j][1];
        acc += w[j].z * v[j][2];
        acc += w[j].w * v[j][3];
      }
    }
    const int q = q0 + qi2;
    if (q < d.Lq) store_row4<TV>(out + ((int64_t(b) * d.Lq + q) * d.M + m) * D + ch * 4, acc);
    }
    __builtin_amdgcn_wave_barrier();      // this tile's records have been read: the next decode may overwrite them
  }
  stamp_end(stamps);
}

// the slab kernel takes a call when it is built for it (fp32 values, 4 levels x 4 points) and the call has enough queries per
// (batch, head) for a workgroup's slab copy to pay (an encoder's; development build: variant 730 forces, 731 forbids)
static bool use_slab_forward(int vdt, int ldt, const MsdaDims& d, int variant) {
  if (ldt != VNX_F32 || d.L != 4 || d.P != 4 || d.D != 32) return false;
  if (vdt != VNX_F32 && vdt != VNX_BF16 && vdt != VNX_F16) return false;      // (16-bit values: round 6)
  if (int64_t(d.S) * d.M * 128 >= (int64_t(1) << 31) || d.S >= (1 << 23)) return false;
  if (variant == 731) return false;
  return variant == 730 || d.Lq >= 2048;
}

// Workgroups per (batch, head): whole ROUNDS of resident workgroups (two per CU = 512) -- a partly filled last round leaves
// CUs idle for a whole workgroup's life (560 workgroups at encoder-360p: 54.6 us against 42.6 with 480) -- and as many rounds
// (1, 2, 4, ...) as bring a workgroup to twelve tiles or fewer: every workgroup copies the slab once, but on the largest
// calls shorter workgroups pack the launch's tail better (encoder-720p B = 5: one round = 25 tiles each 247 us, four rounds
// 231-234; B = 2, 9.6 tiles each: 85 us with one round, 87 with two; encoder-360p at B = 10 -- two clips -- 13.3 tiles in one
// round 88.5 us, 6.7 in two rounds 83.4, the gather kernel 110.9; VNX_SLAB_TILES_MAX 8 / 12 / 16 measured: 12).
static int slab_parts(const MsdaDims& d, int n_tiles, int resident = 512) {
  const int per_round = resident / (d.B * d.M) > 0 ? resident / (d.B * d.M) : 1;
  int parts = per_round;
#ifndef VNX_SLAB_TILES_MAX
#define VNX_SLAB_TILES_MAX 12
#endif
  while (parts < n_tiles && (n_tiles + parts - 1) / parts > VNX_SLAB_TILES_MAX) parts *= 2;
  return parts > n_tiles ? n_tiles : parts;
}

template <typename TV, typename TL, int WAVES, int CAP128>
static int launch_fwd_slab_cfg(const void* value, const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                               void* out, const MsdaDims& d, const FusedArgs* fa, hipStream_t stream) {
  constexpr size_t lds = slab_lds_bytes(WAVES, CAP128);
  const int n_tiles = (d.Lq + WAVES * kSlabQpw - 1) / (WAVES * kSlabQpw);
  const int parts = slab_parts(d, n_tiles, lds * 2 <= 160 * 1024 ? 512 : 256);
  const int64_t blocks = int64_t(parts) * d.B * d.M;
  static thread_local int raised_on[2] = {-1, -1};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int which = fa != nullptr;
  if (raised_on[which] != dev) {
    const void* fn = fa ? reinterpret_cast<const void*>(&msda_fwd_slab_kernel<TV, TL, true, WAVES, CAP128>)
                        : reinterpret_cast<const void*>(&msda_fwd_slab_kernel<TV, TL, false, WAVES, CAP128>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess)
      return check_launch("msda_fwd_slab (LDS limit)");
    raised_on[which] = dev;
  }
  if (fa)
    hipLaunchKernelGGL((msda_fwd_slab_kernel<TV, TL, true, WAVES, CAP128>), dim3(uint32_t(blocks)), dim3(64 * WAVES), lds, stream,
                       (const TV*)value, shapes, lsi, (const TL*)loc, (const TL*)attn, (TV*)out, d, parts,
                       take_stamp_region(kStampFwd, blocks), *fa);
  else
    hipLaunchKernelGGL((msda_fwd_slab_kernel<TV, TL, false, WAVES, CAP128>), dim3(uint32_t(blocks)), dim3(64 * WAVES), lds, stream,
                       (const TV*)value, shapes, lsi, (const TL*)loc, (const TL*)attn, (TV*)out, d, parts,
                       take_stamp_region(kStampFwd, blocks), FusedArgs{});
  return check_launch("msda_fwd_slab");
}

// 16-bit values, the large slab: built, exact, measured (round 6, kbench cold, encoder-720p bf16, small / large slab: B = 5
// 167.4 / 168.0 us unfused -- warm 135 / 147 --, 200.8 / 207.2 fused; B = 2 65.1 / 61.9; forced at 360p 39.7 / 41.5; the gather
// kernel 177.6 / 68.8) -- and NOT the product path: with half of the taps staged instead of a quarter the call is no faster.
// With 64-byte rows the gathered half was not bound by the L1's delivery to begin with (the same finding as at 360p,
// DESIGN.md section 3.1g), and a CU's one 16-wave workgroup copies 77 KB per 6-13 tiles where two 8-wave ones copy 15 KB each.
// Development build: variant 737 takes it (tests/test_msda_slab.py holds it to the oracle), anything else the small slab.
static bool use_large_slab(const MsdaDims&, int variant) { return variant == 737; }

template <typename TV, typename TL>
static int launch_fwd_slab(const void* value, const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                           void* out, const MsdaDims& d, const FusedArgs* fa, hipStream_t stream) {
  if constexpr (sizeof(TV) == 2) {
    if (use_large_slab(d, kernel_variant()))
      return launch_fwd_slab_cfg<TV, TL, kSlabWavesL, kSlabRowsCapL>(value, shapes, lsi, loc, attn, out, d, fa, stream);
  }
  return launch_fwd_slab_cfg<TV, TL, kSlabWaves, kSlabRowsCap>(value, shapes, lsi, loc, attn, out, d, fa, stream);
}

struct FwdCfg { int qpw; int wpb; };

static FwdCfg pick_fwd_cfg(const MsdaDims& d, int variant) {
  // variant: 0 auto; 2..5 force QPW = 8,4,2,1 with 4 waves/block; 12..15 the same with 1 wave/block
  const int LP = d.L * d.P;
  FwdCfg c{8, 4};
  if (variant >= 20 && variant < 60) variant = (variant - 20) % 20;  // prefetch on/off wrappers
  if (variant >= 60 && variant < 69) variant = 13;
  if (variant == 69) variant = 0;                                    // automatic configuration, 8-lane map for 16-bit rows
  if (variant >= 2 && variant <= 5) c = FwdCfg{8 >> (variant - 2), 4};
  else if (variant >= 12 && variant <= 15) c = FwdCfg{8 >> (variant - 12), 1};
  else {
    // measured on MI355X (tools/time_variants.py): 4 queries per wave wins from the T=5 decoder
    // call (1 500 rows) to the 360p encoder call (25 500 rows); one query per 8-lane group only
    // pays on very large calls, where occupancy matters more than per-wave parallelism
    const int64_t rows = int64_t(d.B) * d.Lq;
    if (rows <= 4096) c = FwdCfg{4, 1};
    else if (rows <= 262144) c = FwdCfg{4, 4};
    else c = FwdCfg{8, 4};
  }
  while (c.qpw < 8 && (LP % (8 / c.qpw)) != 0) c.qpw <<= 1;
  return c;
}

template <typename TV, typename TL, int QPW, int WPB>
static int launch_fwd_cfg(const void* value, const int64_t* shapes, const int64_t* lsi,
                          const void* loc, const void* attn, void* out, const MsdaDims& d,
                          int variant, hipStream_t stream) {
  const int LP = d.L * d.P;
  const int tiles_per_batch = (d.Lq + QPW * WPB - 1) / (QPW * WPB);
  // phase 0 pays when the call has at least ~2 taps per row of the map (see the kernel);
  // variants 20..39 force it on for A/B runs, 40..59 force it off
  const bool dense = int64_t(d.Lq) * LP * 4 >= 2 * int64_t(d.S);
  // Measured (tools/time_variants.py, T=5 decoder call): cold 10.5 vs 11.1 us with uniform
  // locations, but 10.9 vs 9.3 us with model-like ones and 6.7 vs 5.7 us cache-warm -> off by default.
  (void)dense;
  bool want_prefetch = (variant >= 20 && variant < 40);
  if constexpr (!(QPW * 16 <= 64)) want_prefetch = false;
  if (LP != 16 || sizeof(TL) != 4 || d.L > 4) want_prefetch = false;
  int prefetch_rows = want_prefetch ? (d.S + tiles_per_batch * WPB - 1) / (tiles_per_batch * WPB) : 0;
  if (prefetch_rows > 32 * kPfSteps) prefetch_rows = 32 * kPfSteps;     // a larger share is streamed in part
  if (variant >= 60 && variant < 68) prefetch_rows |= (variant - 60) << 16;
  if (variant == 68) prefetch_rows |= 14 << 16;      // the fixed head -> XCD map (A/B record)
  const int64_t blocks = int64_t(d.B) * tiles_per_batch * d.M;
  if (blocks >= (int64_t(1) << 31)) {
    set_error("msda_forward: %lld workgroups exceed the grid limit", (long long)blocks);
    return VNX_ERR_UNSUPPORTED;
  }
  const size_t lds = size_t(WPB) * 2 * QPW * (LP + 1) * 16 + (want_prefetch ? size_t(WPB) * 256 : 0);
  if constexpr (QPW * 16 <= 64 && sizeof(TL) == 4) {
    if (want_prefetch) {
      hipLaunchKernelGGL((msda_fwd_d32_kernel<TV, TL, QPW, WPB, 16, false, true>), dim3(uint32_t(blocks)),
                         dim3(64 * WPB), lds, stream, (const TV*)value, shapes, lsi,
                         (const TL*)loc, (const TL*)attn, (TV*)out, d, tiles_per_batch, prefetch_rows,
                         take_stamp_region(kStampFwd, blocks), FusedArgs{});
      return check_launch("msda_fwd_d32_pf");
    }
  }
  // 16-bit values: rows as 4 lanes x 16 B (variant 69 keeps the 8 x 8 B map for A/B runs)
  constexpr int kLpr = sizeof(TV) == 2 ? 4 : 8;
  if (LP == 16 && kLpr == 4 && variant != 69)
    hipLaunchKernelGGL((msda_fwd_d32_kernel<TV, TL, QPW, WPB, 16, false, false, kLpr>), dim3(uint32_t(blocks)),
                       dim3(64 * WPB), lds, stream, (const TV*)value, shapes, lsi,
                       (const TL*)loc, (const TL*)attn, (TV*)out, d, tiles_per_batch, prefetch_rows,
                       take_stamp_region(kStampFwd, blocks), FusedArgs{});
  else if (LP == 16)
    hipLaunchKernelGGL((msda_fwd_d32_kernel<TV, TL, QPW, WPB, 16>), dim3(uint32_t(blocks)),
                       dim3(64 * WPB), lds, stream, (const TV*)value, shapes, lsi,
                       (const TL*)loc, (const TL*)attn, (TV*)out, d, tiles_per_batch, prefetch_rows,
                       take_stamp_region(kStampFwd, blocks), FusedArgs{});
  else
    hipLaunchKernelGGL((msda_fwd_d32_kernel<TV, TL, QPW, WPB, 0>), dim3(uint32_t(blocks)),
                       dim3(64 * WPB), lds, stream, (const TV*)value, shapes, lsi,
                       (const TL*)loc, (const TL*)attn, (TV*)out, d, tiles_per_batch, prefetch_rows,
                       take_stamp_region(kStampFwd, blocks), FusedArgs{});
  return check_launch("msda_fwd_d32");
}

template <typename TV, typename TL>
static int launch_fwd(const void* value, const int64_t* shapes, const int64_t* lsi,
                      const void* loc, const void* attn, void* out, const MsdaDims& d,
                      int variant, hipStream_t stream) {
  const FwdCfg c = pick_fwd_cfg(d, variant);
#define VNX_CASE(Q, W)                                                                       \
  if (c.qpw == Q && c.wpb == W)                                                              \
    return launch_fwd_cfg<TV, TL, Q, W>(value, shapes, lsi, loc, attn, out, d, variant, stream);
  VNX_CASE(8, 4) VNX_CASE(4, 4) VNX_CASE(2, 4) VNX_CASE(1, 4)
  VNX_CASE(8, 1) VNX_CASE(4, 1) VNX_CASE(2, 1) VNX_CASE(1, 1)
  VNX_CASE(4, 2)
#undef VNX_CASE
  set_error("msda_forward: no kernel for qpw=%d wpb=%d", c.qpw, c.wpb);
  return VNX_ERR_UNSUPPORTED;
}

bool msda_d32_fwd_supported(int vdt, int ldt, const MsdaDims& d) {
  if (d.D != 32) return false;
  if (vdt == VNX_F64) return false;
  if (vdt == VNX_F32 && ldt != VNX_F32) return false;
  const int LP = d.L * d.P;
  if (LP > 64) return false;                                    // LDS record budget per wave
  if (int64_t(d.S) * d.M * 32 * elem_size(vdt) >= (int64_t(1) << 31)) return false;  // descriptor range
  if (d.S >= (1 << 23) || d.M * 128 >= (1 << 23)) return false;                        // 24-bit index multiplies
  return true;
}


int msda_forward_d32(int vdt, int ldt, const void* value, const int64_t* shapes,
                     const int64_t* lsi, const void* loc, const void* attn, void* out, MsdaDims d,
                     int variant, hipStream_t stream) {
  if (use_slab_forward(vdt, ldt, d, variant)) {
    if (vdt == VNX_F32) return launch_fwd_slab<float, float>(value, shapes, lsi, loc, attn, out, d, nullptr, stream);
    if (vdt == VNX_BF16) return launch_fwd_slab<bf16_t, float>(value, shapes, lsi, loc, attn, out, d, nullptr, stream);
    return launch_fwd_slab<f16_t, float>(value, shapes, lsi, loc, attn, out, d, nullptr, stream);
  }
  if (vdt == VNX_F32) return launch_fwd<float, float>(value, shapes, lsi, loc, attn, out, d, variant, stream);
  if (vdt == VNX_BF16 && ldt == VNX_F32) return launch_fwd<bf16_t, float>(value, shapes, lsi, loc, attn, out, d, variant, stream);
  if (vdt == VNX_BF16 && ldt == VNX_BF16) return launch_fwd<bf16_t, bf16_t>(value, shapes, lsi, loc, attn, out, d, variant, stream);
  if (vdt == VNX_F16 && ldt == VNX_F32) return launch_fwd<f16_t, float>(value, shapes, lsi, loc, attn, out, d, variant, stream);
  if (vdt == VNX_F16 && ldt == VNX_F16) return launch_fwd<f16_t, f16_t>(value, shapes, lsi, loc, attn, out, d, variant, stream);
  set_error("msda_forward_d32: unsupported dtype pair (%d, %d)", vdt, ldt);
  return VNX_ERR_INVALID_ARGUMENT;
}

// -----------------------------------------------------------------------------
// backward
// -----------------------------------------------------------------------------
// 8-lane sum that leaves the total in every lane of the group: two quad
// permutes and a half-row mirror, all DPP (no LDS traffic).
__device__ __forceinline__ float group8_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  return v;
}

// The same for four values at once, as v_add_f32_dpp (one instruction per step): left to the compiler the four chains are
// vectorised into v_pk_add_f32, which has no DPP form, so every step becomes v_mov_b32_dpp + half a packed add.  The four
// chains are interleaved, so a DPP read follows the write of its register by three instructions (>= the 2 wait states the
// hardware wants between a VALU write and a DPP read); the leading s_nop covers the first step.
__device__ __forceinline__ void group8_sum4(float4_t& v) {
  float a = v.x, b = v.y, c = v.z, d = v.w;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  v = float4_t{a, b, c, d};
}

// Four values summed over the 8 lanes of a set in EIGHT DPP adds instead of twelve: the half-row mirror step first, with
// bank masks -- the even quad of a set keeps (a, b), the odd quad (c, d), each adding the other quad's copy -- leaves two
// values per lane, which two quad permutes finish.  Afterwards every lane of the even quad holds (sum a, sum b) in
// (t0, t1), every lane of the odd quad (sum c, sum d).
__device__ __forceinline__ void group8_sum4_split(const float4_t v, float& t0, float& t1) {
  const float a = v.x, b = v.y, c = v.z, d = v.w;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %1, %3, %3 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
      : "=&v"(t0), "=&v"(t1)
      : "v"(a), "v"(b), "v"(c), "v"(d));
}

// the 4-lane form (rows of 16-bit values as 4 lanes x 16 B): two quad permutes
__device__ __forceinline__ void group4_sum4(float4_t& v) {
  float a = v.x, b = v.y, c = v.z, d = v.w;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  v = float4_t{a, b, c, d};
}

template <typename TV>
__device__ __forceinline__ float4_t load_row4(const TV* p);
template <>
__device__ __forceinline__ float4_t load_row4<float>(const float* p) {
  return *reinterpret_cast<const float4_t*>(p);
}
template <>
__device__ __forceinline__ float4_t load_row4<bf16_t>(const bf16_t* p) {
  const uint2_t r = *reinterpret_cast<const uint2_t*>(p);
  float4_t v;
  v.x = __uint_as_float(r.x << 16); v.y = __uint_as_float(r.x & 0xffff0000u);
  v.z = __uint_as_float(r.y << 16); v.w = __uint_as_float(r.y & 0xffff0000u);
  return v;
}
template <>
__device__ __forceinline__ float4_t load_row4<f16_t>(const f16_t* p) {
  const uint2_t r = *reinterpret_cast<const uint2_t*>(p);
  float4_t v;
  v.x = half_bits_to_float(r.x & 0xffffu); v.y = half_bits_to_float(r.x >> 16);
  v.z = half_bits_to_float(r.y & 0xffffu); v.w = half_bits_to_float(r.y >> 16);
  return v;
}

template <typename TL>
__device__ __forceinline__ void store_loc(TL* p, float v) { *p = from_acc<TL>(v); }

// Same lane map as the forward (QPW queries x 8/QPW sample groups x 8 channel
// lanes) with three phases per wave:
//   1. one lane per (query, sample): geometry -> LDS record {4 tap offsets | lh, lw, attn, -}
//   2. 8-lane groups walk their samples: 4 tap loads, 16 hardware fp32 atomics into
//      the grad_value image (taps outside the map carry an out-of-range offset, which
//      the buffer unit drops for atomics just as it zero-fills loads), channel partial
//      sums of the three scalar gradients reduced over the 8 lanes with DPP;
//   3. the phase-1 lane of each sample scales by (W, H) and writes grad_loc /
//      grad_attn coalesced -- once, no atomics (the reference's block reductions,
//      cuh:376-394, become 9 DPP adds).
// grad_value accumulates in fp32 (`gv`: grad_value itself for fp32, the workspace
// image for 16-bit values), laid out [B,S,M,32] fp32.
// The kernel's body as a device function of the workgroup's index `vblock`, the wave's index inside it `wave` and the
// workgroup's LDS: msda_bwd_d32_kernel below is this and nothing else; msda_bwd_pair_kernel runs the one-wave-per-workgroup
// configuration in every WAVE of its grad_loc workgroups (vblock = that wave's own workgroup index, wave = 0, its own LDS slice).
template <typename TV, typename TL, int QPW, int WPB, int LP_T, bool ATOMICS, bool FUSED = false, int LPR = 8, int KB = 0>
__device__ __forceinline__ void
msda_bwd_d32_body(const TV* __restrict__ value, const int64_t* __restrict__ shapes,
                  const int64_t* __restrict__ lsi, const TL* __restrict__ loc,
                  const TL* __restrict__ attn, const TV* __restrict__ grad_out,
                  float* __restrict__ gv, TL* __restrict__ grad_loc, TL* __restrict__ grad_attn,
                  const MsdaDims& d, int tiles_per_batch, uint4_t* __restrict__ sample_records,
                  uint32_t* __restrict__ sample_units, uint32_t* __restrict__ tile_summary, int units_min,
                  unsigned long long* stamps, const FusedArgs& fa, const uint32_t vblock, const int wave,
                  unsigned char* __restrict__ smem) {
  static_assert(!FUSED || (LP_T == 16 && !ATOMICS), "the fused prologue is built for L*P == 16, record-fed grad_value");
  static_assert(LPR == 8 || (LPR == 4 && LP_T == 16 && !ATOMICS), "4 lanes per row: L*P == 16, owner-computes grad_value");
  stamp_begin(stamps);
  constexpr int D = 32;
  constexpr int PG = (64 / LPR) / QPW;
  constexpr int kRowBytes = D * int(sizeof(TV));
  constexpr int kLaneBytes = kRowBytes / LPR;

  const int LP = LP_T > 0 ? LP_T : d.L * d.P;
  const int p_shift = __builtin_ctz(uint32_t(d.P) | 0x10000u);      // log2(P) when P is a power of two (L * P == 16)
  const int lane = threadIdx.x & 63;
  const int tile = vblock / d.M;
  const int b = tile / tiles_per_batch;
  const int m = (vblock % d.M + b) % d.M;      // head <-> XCD map rotates with the batch element (see the forward)
  const int q0 = (tile - b * tiles_per_batch) * (QPW * WPB) + wave * QPW;

  // per-wave LDS: tap offsets, geometry, results
  const int ent = QPW * (LP + 1);
  uint4_t* s_off = reinterpret_cast<uint4_t*>(smem) + size_t(wave) * 3 * ent;
  float4_t* s_geo = reinterpret_cast<float4_t*>(s_off + ent);
  float4_t* s_res = s_geo + ent;

  const uint32_t pixel_elems = uint32_t(d.M * D);

  // ---- phase 0 (record-fed grad_value path with >= 1 024 queries only): zero the grad_value rows of the query-split levels
  //      (gv_query_splits) of this (batch, head), where that path's pieces meet through fp32 atomics: the tiles of a batch
  //      element share the rows, eight rows per wave and step.  (The tile-fed path -- every encoder call -- has no atomics
  //      and no zeroing since round 4: its pieces store partial rows, msda_d32_gvtiles.hip; this phase cost 6.7 us of
  //      the 81-us kernel at encoder-360p.) ---------------------------------------------------------------
  if constexpr (!ATOMICS) {      // (fp32 values: rows of grad_value itself; 16-bit values: rows of the fp32 split image)
    if (!(VNX_K1_ABL & 8) && fa.qsplit_zero != nullptr && sample_units != nullptr && d.Lq >= 1024 &&
        levels_packed(shapes, lsi, d.L, d.S)) {
      const int t_in_b = tile - b * tiles_per_batch;
      for (int l = 0; l < d.L; ++l) {
        const int Hz = int(shapes[2 * l]), Wz = int(shapes[2 * l + 1]), n = Hz * Wz;
        if (gv_query_splits(gv_level_units(Hz, Wz, units_min, false), d.Lq, d.P, true, d.B * d.M) > 1) {
          float* rows = fa.qsplit_zero + ((int64_t(b) * d.S + int(lsi[l])) * d.M + m) * D;
          for (int r = (t_in_b * WPB + wave) * 8 + (lane >> 3); r < n; r += tiles_per_batch * WPB * 8)
            *reinterpret_cast<float4_t*>(rows + int64_t(r) * d.M * D + (lane & 7) * 4) = float4_t{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  }

  // ---- phase 1 ------------------------------------------------------------------
  const int pairs = QPW * LP;
  int keep_H = 1, keep_W = 1;      // level size of this lane's sample: reused by phase 3 when pairs <= 64
  // tile mode of the grad_value path (msda_d32_gvtiles.hip; L*P == 16, P == 4): instead of a record and a unit range per
  // SAMPLE, this workgroup leaves per level the BOUNDING BOX of the pixels its queries' samples touch, two words:
  //   x_lo | (0xffff - x_hi) << 16   and   y_lo | (0xffff - y_hi) << 16
  // -- the form a packed 16-bit minimum reduces; 0xffffffff = no taps.  (Until late in round 3 it was the range of
  // strip units, which cost two divisions per sample here and tied this kernel to the unit geometry.)
  uint32_t tile_kx = 0xffffffffu, tile_ky = 0xffffffffu;
  for (int e = lane; e < pairs; e += 64) {
    const int qi = e / LP, p = e - qi * LP;
    const int q = q0 + qi;
    uint4_t o4 = {kTapOutsideElem, kTapOutsideElem, kTapOutsideElem, kTapOutsideElem};
    float4_t g4 = {0.f, 0.f, 0.f, 0.f};
    const int l = LP_T == 16 ? (p >> p_shift) : p / d.P;       // L * P == 16: P is a power of two
    const int64_t wi = ((int64_t(b) * d.Lq + q) * d.M + m) * LP + p;
    float x = 0.f, y = 0.f, a = 0.f;
    if constexpr (FUSED) {
      float sx, sy;
      const bool valid = q < d.Lq;
      fused_decode<TL>(loc, attn, fa, wi, b, q, l, valid ? int(shapes[2 * l]) : 1, valid ? int(shapes[2 * l + 1]) : 1,
                       d, valid, x, y, a, sx, sy);
      g4.w = a;   // the softmax weight of every sample, in or out of the map (softmax backward, phase 3)
    }
    if (q < d.Lq) {
      int H = 1, W = 1, start = 0;
      constexpr bool kOneRequest = !FUSED && sizeof(TL) == 4 && LP_T > 0 && QPW * LP_T <= 64;      // see load_sample_f32
      if constexpr (kOneRequest) {      // (WPB == 1, the small-call configuration: `nt` loads as in the forward)
        load_sample_f32<WPB == 1>(shapes, lsi, reinterpret_cast<const float*>(loc), reinterpret_cast<const float*>(attn), l, wi,
                                  H, W, start, x, y, a);
      } else if constexpr (!FUSED) {
        if constexpr (WPB == 1 && sizeof(TL) == 4) {
          x = to_acc(nt_load(loc + 2 * wi)); y = to_acc(nt_load(loc + 2 * wi + 1));
          a = to_acc(nt_load(attn + wi));
        } else {
          x = to_acc(loc[2 * wi]); y = to_acc(loc[2 * wi + 1]);
          a = to_acc(attn[wi]);
        }
      }
      if constexpr (LP_T == 16 && !ATOMICS) {
        if (fa.tile_loc != nullptr) {      // what the tile-fed grad_value kernel decodes again (fp32: the same bits), laid out
          // [batch][head][level][query][point]: that kernel walks the queries of one (batch, head, level)
          const int64_t ci = ((int64_t(b) * d.M + m) * d.L + l) * (int64_t(d.Lq) * d.P) + int64_t(q) * d.P + (p - l * d.P);
          *reinterpret_cast<float2_t*>(fa.tile_loc + 2 * ci) = float2_t{x, y};
          fa.tile_attn[ci] = a;
        }
      }
      if constexpr (!kOneRequest) { H = int(shapes[2 * l]); W = int(shapes[2 * l + 1]); start = int(lsi[l]); }
      keep_H = H; keep_W = W;
      const float h = y * float(H) - 0.5f, w = x * float(W) - 0.5f;
      uint4_t record = {0xffffffffu, 0u, 0u, 0u};  // sample outside the map: no taps
      if (h > -1.f && w > -1.f && h < float(H) && w < float(W)) {
        const float hf = floorf(h), wf = floorf(w);
        const int h0 = int(hf), w0 = int(wf);
        const bool top = h0 >= 0, bot = h0 + 1 <= H - 1, lef = w0 >= 0, rig = w0 + 1 <= W - 1;
        // element (not byte) offsets: the same record addresses `value` (x sizeof(TV))
        // and the fp32 gradient image (x 4)
        const uint32_t o00 = uint32_t(__mul24(start + __mul24(h0, W) + w0, int(pixel_elems)));  // mod 2^32 on purpose (24-bit multiplies: see the forward)
        const uint32_t row_e = __umul24(uint32_t(W), pixel_elems);
        o4.x = (top && lef) ? o00 : kTapOutsideElem;
        o4.y = (top && rig) ? o00 + pixel_elems : kTapOutsideElem;
        o4.z = (bot && lef) ? o00 + row_e : kTapOutsideElem;
        o4.w = (bot && rig) ? o00 + row_e + pixel_elems : kTapOutsideElem;
        g4.x = h - hf; g4.y = w - wf; g4.z = a;
        record = uint4_t{gv_pack_corner(h0, w0, H, W), __float_as_uint(g4.x),
                         __float_as_uint(g4.y), __float_as_uint(a)};
        if constexpr (LP_T == 16 && !ATOMICS) {
          if (!(VNX_K1_ABL & 2) && tile_summary != nullptr) {
            // coordinates saturate at 0xfffe, which the grad_value kernel reads as "or beyond" (a level side of 65 535+)
            const uint32_t x_lo = min(uint32_t(lef ? w0 : w0 + 1), 0xfffeu), x_hi = min(uint32_t(rig ? w0 + 1 : w0), 0xfffeu);
            const uint32_t y_lo = min(uint32_t(top ? h0 : h0 + 1), 0xfffeu), y_hi = min(uint32_t(bot ? h0 + 1 : h0), 0xfffeu);
            tile_kx = pk_min_u16(tile_kx, x_lo | ((0xffffu - x_hi) << 16));
            tile_ky = pk_min_u16(tile_ky, y_lo | ((0xffffu - y_hi) << 16));
          }
        }
      }
      // the geometry, kept for the grad_value kernel (msda_d32_gvrec.hip):
      // [batch][head][level][query*points], 16 B per sample -- plus, in 4 B, the range of that
      // kernel's units (pixel ranges of `rows per unit` rows) the sample's corners fall into
      if (sample_records != nullptr) {
        const int64_t ri = ((int64_t(b) * d.M + m) * d.L + l) * (int64_t(d.Lq) * d.P) + int64_t(q) * d.P + (p - l * d.P);
        // (nt stores of the records / unit tags / gradients: this kernel 93.6 -> 73 us at the encoder shape, the
        //  grad_value kernel that reads them correspondingly slower -- 211.7 vs 213-215 us per backward; not kept)
        sample_records[ri] = record;
        if (sample_units != nullptr) {
          uint32_t uu = 0xffffffffu;
          if (record.x != 0xffffffffu) {
            int h0, w0;
            gv_unpack_corner(record.x, H, W, h0, w0);
            const bool top = h0 >= 0, bot = h0 + 1 <= H - 1, lef = w0 >= 0, rig = w0 + 1 <= W - 1;
            const int p00 = h0 * W + w0;
            const int lo = (top && lef) ? p00 : (top && rig) ? p00 + 1 : (bot && lef) ? p00 + W : p00 + W + 1;
            const int hi = (bot && rig) ? p00 + W + 1 : (bot && lef) ? p00 + W : (top && rig) ? p00 + 1 : p00;
            const int rpu = gv_rows_per_unit(H * W, units_min, kGvRowsMax);
            uu = uint32_t(small_div(lo, rpu)) | (uint32_t(small_div(hi, rpu)) << 16);
          }
          sample_units[ri] = uu;
        }
      }
    }
    s_off[qi * (LP + 1) + p] = o4;
    s_geo[qi * (LP + 1) + p] = g4;
  }
  uint2_t* s_tile = reinterpret_cast<uint2_t*>(smem + size_t(WPB) * 3 * ent * 16);    // [WPB][4] (tile mode)
  uint2_t* tile_words = reinterpret_cast<uint2_t*>(tile_summary);                       // [b][head][level][tile]
  if constexpr (LP_T == 16 && !ATOMICS) {
    if (!(VNX_K1_ABL & 2) && tile_summary != nullptr) {       // uniform.  A lane's samples all have level (lane & 15) >> 2 (P == 4):
      // minimum over the 4 points (quad) and over the wave's queries (lane bits 4, 5)
      tile_kx = pk_min_u16(tile_kx, uint32_t(__builtin_amdgcn_update_dpp(int(tile_kx), int(tile_kx), 0xB1, 0xF, 0xF, true)));
      tile_ky = pk_min_u16(tile_ky, uint32_t(__builtin_amdgcn_update_dpp(int(tile_ky), int(tile_ky), 0xB1, 0xF, 0xF, true)));
      tile_kx = pk_min_u16(tile_kx, uint32_t(__builtin_amdgcn_update_dpp(int(tile_kx), int(tile_kx), 0x4E, 0xF, 0xF, true)));
      tile_ky = pk_min_u16(tile_ky, uint32_t(__builtin_amdgcn_update_dpp(int(tile_ky), int(tile_ky), 0x4E, 0xF, 0xF, true)));
      tile_kx = pk_min_u16(tile_kx, uint32_t(__shfl_xor(int(tile_kx), 16, 64)));
      tile_ky = pk_min_u16(tile_ky, uint32_t(__shfl_xor(int(tile_ky), 16, 64)));
      tile_kx = pk_min_u16(tile_kx, uint32_t(__shfl_xor(int(tile_kx), 32, 64)));
      tile_ky = pk_min_u16(tile_ky, uint32_t(__shfl_xor(int(tile_ky), 32, 64)));
      if ((lane & 3) == 0 && lane < 16) {
        if (kTilePerWave) {      // one box per WAVE (QPW queries): finer selection for the grad_value kernel, no LDS step here
          const int wt = (tile - b * tiles_per_batch) * WPB + wave, wn = (d.Lq + QPW - 1) / QPW;
          if (wt < wn) tile_words[((int64_t(b) * d.M + m) * d.L + (lane >> 2)) * wn + wt] = uint2_t{tile_kx, tile_ky};
        } else if (WPB > 1) s_tile[wave * 4 + (lane >> 2)] = uint2_t{tile_kx, tile_ky};
        else tile_words[((int64_t(b) * d.M + m) * d.L + (lane >> 2)) * tiles_per_batch + (tile - b * tiles_per_batch)] = uint2_t{tile_kx, tile_ky};
      }
    }
  }
  if (WPB > 1) __syncthreads(); else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if constexpr (LP_T == 16 && !ATOMICS && WPB > 1 && !kTilePerWave) {
    if (tile_summary != nullptr && threadIdx.x < 4) {
      uint2_t k = s_tile[threadIdx.x];
#pragma unroll
      for (int w2 = 1; w2 < WPB; ++w2) {
        const uint2_t o = s_tile[w2 * 4 + threadIdx.x];
        k.x = pk_min_u16(k.x, o.x); k.y = pk_min_u16(k.y, o.y);
      }
      tile_words[((int64_t(b) * d.M + m) * d.L + int(threadIdx.x)) * tiles_per_batch + (tile - b * tiles_per_batch)] = k;
    }
  }

  // ---- phase 2 ------------------------------------------------------------------
  const int ch = lane & (LPR - 1);
  const int qi = (lane / LPR) % QPW;
  const int pg = lane / (LPR * QPW);
  const int q = q0 + qi;

  const int64_t head_elem = (int64_t(b) * d.S * d.M + m) * D;
  const uint32_t head_elems = uint32_t((int64_t(d.S) * d.M - m) * D);
  const TV* vbase = value + head_elem;
  float* gbase = gv + head_elem;
  const __amdgpu_buffer_rsrc_t vsrc = uniform_rsrc(vbase, head_elems * uint32_t(sizeof(TV)));
  const __amdgpu_buffer_rsrc_t gsrc = uniform_rsrc(gbase, head_elems * 4u);

  const int per_group = LP / PG;
  const uint4_t* g_off = s_off + qi * (LP + 1) + pg * per_group;
  const float4_t* g_geo = s_geo + qi * (LP + 1) + pg * per_group;
  float4_t* g_res = s_res + qi * (LP + 1) + pg * per_group;

  if constexpr (LPR == 4) {
    // a row as 4 lanes x 8 channels (16-bit values: one 16-B load per tap; fp32: two): the four dots of a sample are
    // reduced over 4 lanes (two quad permutes) instead of 8, and a wave instruction covers 16 samples.  fp32 (round 3):
    // this kernel is bound by vector issue on large calls; per 8 samples 8 x 8 lanes cost 4 loads + 12 packed dot
    // instructions + 8 DPP adds, 16 x 4 lanes cost per 16 samples 8 loads + 20 + 8.
    constexpr bool k16 = sizeof(TV) == 2;
    float4_t t_lo = {0.f, 0.f, 0.f, 0.f}, t_hi = t_lo;
    if (q < d.Lq) {
      const TV* gp = grad_out + ((int64_t(b) * d.Lq + q) * d.M + m) * D + ch * 8;
      if constexpr (k16) {
        unpack8<TV>(*reinterpret_cast<const uint4_t*>(gp), t_lo, t_hi);
      } else {
        t_lo = *reinterpret_cast<const float4_t*>(gp);
        t_hi = *reinterpret_cast<const float4_t*>(gp + 4);
      }
    }
    const float2_t ta = {t_lo.x, t_lo.y}, tb = {t_lo.z, t_lo.w}, tc = {t_hi.x, t_hi.y}, td = {t_hi.z, t_hi.w};
    auto dot8f = [&](const float4_t lo, const float4_t hi) {
      float2_t a2 = ta * float2_t{lo.x, lo.y};
      a2 = tb * float2_t{lo.z, lo.w} + a2;
      a2 = tc * float2_t{hi.x, hi.y} + a2;
      a2 = td * float2_t{hi.z, hi.w} + a2;
      return a2.x + a2.y;
    };
    auto dot8 = [&](const uint4_t raw) {
      float4_t lo, hi;
      unpack8<TV>(raw, lo, hi);
      return dot8f(lo, hi);
    };
    constexpr int kPer = LP_T / PG;
    constexpr int kWant = KB > 0 ? KB : k16 ? (WPB == 1 ? VNX_K1_BATCH : VNX_K1_BATCH_LARGE) : (WPB == 1 ? 2 : 1);   // fp32: 8 registers per tap
    constexpr int kBatch = kPer < kWant ? kPer : kWant;
    static_assert(kPer % kBatch == 0, "whole batches");
    const uint32_t lane_off = uint32_t(ch * kLaneBytes);
#pragma unroll
    for (int i0 = 0; i0 < kPer; i0 += kBatch) {
      uint4_t o[kBatch];
#pragma unroll
      for (int j = 0; j < kBatch; ++j) o[j] = g_off[i0 + j];
      if constexpr (k16) {
        uint4_t raw[kBatch][4];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          raw[j][0] = __builtin_amdgcn_raw_buffer_load_b128(vsrc, int(o[j].x * 2u + lane_off), 0, 0);
          raw[j][1] = __builtin_amdgcn_raw_buffer_load_b128(vsrc, int(o[j].y * 2u + lane_off), 0, 0);
          raw[j][2] = __builtin_amdgcn_raw_buffer_load_b128(vsrc, int(o[j].z * 2u + lane_off), 0, 0);
          raw[j][3] = __builtin_amdgcn_raw_buffer_load_b128(vsrc, int(o[j].w * 2u + lane_off), 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          float4_t dd = {dot8(raw[j][0]), dot8(raw[j][1]), dot8(raw[j][2]), dot8(raw[j][3])};
          group4_sum4(dd);
          if (ch == 0) g_res[i0 + j] = dd;
        }
      } else {
        uint4_t lo[kBatch][4], hi[kBatch][4];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          const uint32_t ob[4] = {o[j].x * 4u + lane_off, o[j].y * 4u + lane_off, o[j].z * 4u + lane_off, o[j].w * 4u + lane_off};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            lo[j][k] = __builtin_amdgcn_raw_buffer_load_b128(vsrc, int(ob[k]), 0, 0);
            hi[j][k] = __builtin_amdgcn_raw_buffer_load_b128(vsrc, int(ob[k]), 16, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          float4_t dd;
          dd.x = dot8f(__builtin_bit_cast(float4_t, lo[j][0]), __builtin_bit_cast(float4_t, hi[j][0]));
          dd.y = dot8f(__builtin_bit_cast(float4_t, lo[j][1]), __builtin_bit_cast(float4_t, hi[j][1]));
          dd.z = dot8f(__builtin_bit_cast(float4_t, lo[j][2]), __builtin_bit_cast(float4_t, hi[j][2]));
          dd.w = dot8f(__builtin_bit_cast(float4_t, lo[j][3]), __builtin_bit_cast(float4_t, hi[j][3]));
          group4_sum4(dd);
          if (ch == 0) g_res[i0 + j] = dd;
        }
      }
    }
  }

  float4_t top = {0.f, 0.f, 0.f, 0.f};
  if constexpr (LPR == 8) {
    // (`nt` for this row, read once: 27.5-27.8 vs 27.2-27.5 us per headline backward -- not kept)
    if (q < d.Lq) top = load_row4<TV>(grad_out + ((int64_t(b) * d.Lq + q) * d.M + m) * D + ch * 4);
  }

  // one sample: the three scalar gradients from its four taps (+ the grad_value atomics of the general path)
  auto one_sample = [&](int i, const uint4_t o, const float4_t geo, const float4_t v1, const float4_t v2,
                        const float4_t v3, const float4_t v4) {
    const float lh = geo.x, lw = geo.y, a = geo.z;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const float4_t tg = top * a;
    if (ATOMICS) {
      const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
      const float4_t g1 = tg * w1, g2 = tg * w2, g3 = tg * w3, g4 = tg * w4;
      const uint32_t b1 = o.x * 4u + ch * 16u, b2 = o.y * 4u + ch * 16u;
      const uint32_t b3 = o.z * 4u + ch * 16u, b4 = o.w * 4u + ch * 16u;
#define VNX_ATOM4(g, bo)                                                               \
      __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(g.x, gsrc, int(bo), 0, 0);         \
      __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(g.y, gsrc, int(bo + 4), 0, 0);     \
      __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(g.z, gsrc, int(bo + 8), 0, 0);     \
      __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(g.w, gsrc, int(bo + 12), 0, 0);
      VNX_ATOM4(g1, b1) VNX_ATOM4(g2, b2) VNX_ATOM4(g3, b3) VNX_ATOM4(g4, b4)
#undef VNX_ATOM4
    }
    // per-channel: val = bilinear tap, gh / gw = d(val)/d(h, w)   (cuh:123-151)
    const float4_t dh = hw * (v3 - v1) + lw * (v4 - v2);
    const float4_t dw = hh * (v2 - v1) + lh * (v4 - v3);
    const float4_t val = hh * (hw * v1 + lw * v2) + lh * (hw * v3 + lw * v4);
    float s_ga = top.x * val.x + top.y * val.y + top.z * val.z + top.w * val.w;
    float s_gx = tg.x * dw.x + tg.y * dw.y + tg.z * dw.z + tg.w * dw.w;
    float s_gy = tg.x * dh.x + tg.y * dh.y + tg.z * dh.z + tg.w * dh.w;
    s_ga = group8_sum(s_ga);
    s_gx = group8_sum(s_gx);
    s_gy = group8_sum(s_gy);
    if (ch == 0) {
      float4_t r = {s_gx, s_gy, s_ga, 0.f};
      g_res[i] = r;
    }
  };
  auto tap = [&](uint32_t elem_off) { return load_tap<TV>(vsrc, elem_off * uint32_t(sizeof(TV)) + ch * kLaneBytes); };
  if constexpr (LPR == 4) {
    // done above
  } else if constexpr (LP_T > 0 && !ATOMICS) {
    // As in the forward: the row loads of a whole batch of samples are issued before the first use.  (Left
    // to the compiler this loop waited for four loads at a time, eight dependent memory round trips per
    // wave at the decoder shape: 10.9 us per workgroup against the forward's 6.2.)
    // Batch: 4 samples (16 loads, 118 VGPRs) for the small, latency-bound calls (one wave per workgroup);
    // 2 for the large ones (4 waves per workgroup, rows > 4096), which are bound by how many waves fit:
    // encoder shape 106 vs 112 us.
    constexpr int kPer = LP_T / PG;
    constexpr int kWant = KB > 0 ? KB : WPB == 1 ? VNX_K1_BATCH_F32 : VNX_K1_BATCH_LARGE;
    constexpr int kBatch = kPer < kWant ? kPer : kWant;
    // Round 2, late: this kernel is bound by vector-instruction issue on large calls (PMC, encoder 360p: 46.6 M wave
    // instructions x 4 clk / 1 024 SIMDs = 96 us = its duration), and most of them were the per-channel bilinear
    // algebra of every sample (val, d val / dh, d val / dw: 56 operations, then three dots and three 8-lane
    // reductions).  By linearity all three gradients are combinations of FOUR dots d_k = <grad_out row, tap row k>:
    //   grad_attn = hh (hw d1 + lw d2) + lh (hw d3 + lw d4),  grad_w = a (hh (d2 - d1) + lh (d4 - d3)),
    //   grad_h = a (hw (d3 - d1) + lw (d4 - d2))
    // so phase 2 only takes the dots (3 packed instructions per tap) and reduces them; the combinations are left to
    // phase 3, where ONE lane per sample does them for 64 samples at once instead of every lane for its group's sample.
    typedef float float2_t __attribute__((ext_vector_type(2)));
    const float2_t t_lo = {top.x, top.y}, t_hi = {top.z, top.w};
    auto dot = [&](const float4_t v) {
      float2_t acc = t_lo * float2_t{v.x, v.y};
      acc = t_hi * float2_t{v.z, v.w} + acc;
      return acc.x + acc.y;
    };
#pragma unroll
    for (int i0 = 0; i0 < kPer; i0 += kBatch) {      // (kBatch need not divide kPer: the last batch is shorter)
      uint4_t o[kBatch];
      float4_t v[kBatch][4];
#pragma unroll
      for (int j = 0; j < kBatch; ++j)
        if (i0 + j < kPer) o[j] = g_off[i0 + j];
#pragma unroll
      for (int j = 0; j < kBatch; ++j)
        if (i0 + j < kPer) {
          v[j][0] = tap(o[j].x); v[j][1] = tap(o[j].y); v[j][2] = tap(o[j].z); v[j][3] = tap(o[j].w);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        if (i0 + j >= kPer) continue;
        const float4_t dd = {dot(v[j][0]), dot(v[j][1]), dot(v[j][2]), dot(v[j][3])};
        if (VNX_K1_ABL & 4) {
          if (dd.x + dd.y + dd.z + dd.w == 12345.678f) g_res[i0 + j] = dd;
          continue;
        }
        float t0, t1;
        group8_sum4_split(dd, t0, t1);        // lanes 0..3 of the set: (d1, d2); lanes 4..7: (d3, d4)
        if ((ch & 3) == 0) reinterpret_cast<float2_t*>(g_res + i0 + j)[ch >> 2] = float2_t{t0, t1};
      }
    }
  } else {
    for (int i = 0; i < per_group; ++i) {
      const uint4_t o = g_off[i];
      one_sample(i, o, g_geo[i], tap(o.x), tap(o.y), tap(o.z), tap(o.w));
    }
  }
  if (WPB > 1) __syncthreads(); else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- phase 3 ------------------------------------------------------------------
  for (int e = lane; e < ((VNX_K1_ABL & 1) ? 0 : pairs); e += 64) {
    const int qi3 = e / LP, p = e - qi3 * LP;
    const int q3 = q0 + qi3;
    if (q3 < d.Lq) {
      const int l = LP_T == 16 ? (p >> p_shift) : p / d.P;       // L * P == 16: P is a power of two
      const int64_t wi = ((int64_t(b) * d.Lq + q3) * d.M + m) * LP + p;
      float4_t r = s_res[qi3 * (LP + 1) + p];
      if constexpr (LP_T > 0 && !ATOMICS) {   // the four dots of phase 2 -> (grad_w, grad_h, grad_attn), see there
        const float4_t gq = s_geo[qi3 * (LP + 1) + p];
        const float lh = gq.x, lw = gq.y, a = gq.z, hh = 1.f - lh, hw = 1.f - lw;
        const float d1 = r.x, d2 = r.y, d3 = r.z, d4 = r.w;
        r.x = a * (hh * (d2 - d1) + lh * (d4 - d3));
        r.y = a * (hw * (d3 - d1) + lw * (d4 - d2));
        r.z = hh * (hw * d1 + lw * d2) + lh * (hw * d3 + lw * d4);
      }
      // (a vector load of the level size here is one more dependent memory round trip at the end of the wave)
      const float Hf = pairs <= 64 ? float(keep_H) : float(int(shapes[2 * l]));
      const float Wf = pairs <= 64 ? float(keep_W) : float(int(shapes[2 * l + 1]));
      if constexpr (!FUSED) {
#if VNX_K1_P3 >= 1      // A/B: the (x, y) pair of a sample as one 8-byte store (fp32 gradients), 2: both gradients `nt`
        if constexpr (sizeof(TL) == 4) {
          const float2_t gxy = {Wf * r.x, Hf * r.y};      // cuh:157-158
          if (VNX_K1_P3 >= 2) {
            __builtin_nontemporal_store(gxy, reinterpret_cast<float2_t*>(grad_loc + 2 * wi));
            __builtin_nontemporal_store(r.z, reinterpret_cast<float*>(grad_attn + wi));
          } else {
            *reinterpret_cast<float2_t*>(grad_loc + 2 * wi) = gxy;
            *reinterpret_cast<float*>(grad_attn + wi) = r.z;      // cuh:156
          }
        } else
#endif
        {
        store_loc<TL>(grad_loc + 2 * wi, Wf * r.x);       // cuh:157
        store_loc<TL>(grad_loc + 2 * wi + 1, Hf * r.y);   // cuh:158
        store_loc<TL>(grad_attn + wi, r.z);               // cuh:156
        }
      } else {
        // chain rule through the prologue: d loc / d offset is a per-sample scale, the softmax
        // backward  g_logit = a (g_a - sum_j a_j g_a_j)  is a 16-lane row sum, and 2-d reference
        // points collect the location gradients of their level over points (lanes) and heads
        const float gx = Wf * r.x, gy = Hf * r.y, a = s_geo[qi3 * (LP + 1) + p].w;
        const float dot = row16_sum(a * r.z);
        float sx, sy;
        if (fa.ref_dim == 2) {
          sx = 1.f / Wf; sy = 1.f / Hf;
        } else {
          const int64_t rf = ((int64_t(b / fa.ref_div) * d.Lq + q3) * d.L + l) * 4;
          sx = fused_ref<TL>(fa, rf + 2) * 0.5f / float(d.P); sy = fused_ref<TL>(fa, rf + 3) * 0.5f / float(d.P);
        }
#if VNX_K1_P3 >= 2 && !defined(VNX_K1_P3F_PLAIN)      // as in the unfused branch: one 8-byte store per (x, y), both gradients `nt`
        if constexpr (sizeof(TL) == 4) {
          __builtin_nontemporal_store(float2_t{gx * sx, gy * sy}, reinterpret_cast<float2_t*>(grad_loc + 2 * wi));
          __builtin_nontemporal_store(a * (r.z - dot), reinterpret_cast<float*>(grad_attn + wi));
        } else
#endif
        {
        store_loc<TL>(grad_loc + 2 * wi, gx * sx);
        store_loc<TL>(grad_loc + 2 * wi + 1, gy * sy);
        store_loc<TL>(grad_attn + wi, a * (r.z - dot));
        }
        if (fa.grad_reference != nullptr) {
          float rx = gx, ry = gy;
          for (int k = 1; k < d.P; k <<= 1) { rx += __shfl_xor(rx, k, 16); ry += __shfl_xor(ry, k, 16); }
          if (p - l * d.P == 0) {
            float* gr = fa.grad_reference + ((int64_t(b) * d.Lq + q3) * d.L + l) * 2;
            atomic_add(gr, rx);
            atomic_add(gr + 1, ry);
          }
        }
      }
    }
  }
  stamp_end(stamps);
}

template <typename TV, typename TL, int QPW, int WPB, int LP_T, bool ATOMICS, bool FUSED = false, int LPR = 8>
__global__ void VNX_K1_BOUNDS(64 * WPB)
msda_bwd_d32_kernel(const TV* __restrict__ value, const int64_t* __restrict__ shapes,
                    const int64_t* __restrict__ lsi, const TL* __restrict__ loc,
                    const TL* __restrict__ attn, const TV* __restrict__ grad_out,
                    float* __restrict__ gv, TL* __restrict__ grad_loc, TL* __restrict__ grad_attn,
                    MsdaDims d, int tiles_per_batch, uint4_t* __restrict__ sample_records,
                    uint32_t* __restrict__ sample_units, uint32_t* __restrict__ tile_summary, int units_min,
                    unsigned long long* stamps, FusedArgs fa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  msda_bwd_d32_body<TV, TL, QPW, WPB, LP_T, ATOMICS, FUSED, LPR>(value, shapes, lsi, loc, attn, grad_out, gv, grad_loc, grad_attn, d,
                                                                 tiles_per_batch, sample_records, sample_units, tile_summary,
                                                                 units_min, stamps, fa, blockIdx.x,
                                                                 __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)), smem);
}

template <typename TV, typename TL, int QPW, int WPB>
static int launch_bwd_cfg(const void* value, const int64_t* shapes, const int64_t* lsi,
                          const void* loc, const void* attn, const void* grad_out, void* gv,
                          void* grad_loc, void* grad_attn, const MsdaDims& d, bool atomics,
                          void* records, void* tile_summary, float* tile_copy, hipStream_t stream) {
  // records / tile mode: `gv` is not an accumulation image but the fp32 target of the query-split levels' atomics --
  // grad_value itself for fp32 values, the fp32 split image for 16-bit ones (or null) -- whose rows this kernel zeroes
  void* qsplit_zero = (!atomics && records != nullptr) ? gv : nullptr;
  if (tile_summary != nullptr && (d.L * d.P != 16 || d.P != 4 || atomics)) {
    set_error("msda_backward: tile mode needs 4 levels x 4 points");
    return VNX_ERR_UNSUPPORTED;
  }
  const int LP = d.L * d.P;
  const int tiles_per_batch = (d.Lq + QPW * WPB - 1) / (QPW * WPB);
  const int64_t blocks = int64_t(d.B) * tiles_per_batch * d.M;
  if (blocks >= (int64_t(1) << 31)) {
    set_error("msda_backward: %lld workgroups exceed the grid limit", (long long)blocks);
    return VNX_ERR_UNSUPPORTED;
  }
  const size_t lds = size_t(WPB) * 3 * QPW * (LP + 1) * 16 + 128;    // + [WPB][4] tile word pairs (tile mode)
  // the unit ranges sit behind the records in the workspace (gv_unit_ids_offset); only the P == 4
  // grad_value kernel reads them
  void* unit_ids = (records != nullptr && d.P == 4) ? (void*)((char*)records + gv_unit_ids_offset(d)) : nullptr;
  const int units_min = gv_units_min(d, tile_summary != nullptr, kernel_variant());
  constexpr int kLpr = (sizeof(TV) == 2 || VNX_K1_F32_LPR4(WPB)) ? 4 : 8;    // 16-bit rows as 4 lanes x 16 B; fp32: VNX_K1_F32_LPR4
#define VNX_BWD_ARGS (const TV*)value, shapes, lsi, (const TL*)loc, (const TL*)attn, (const TV*)grad_out, (float*)gv, (TL*)grad_loc, \
                     (TL*)grad_attn, d, tiles_per_batch, (uint4_t*)records, (uint32_t*)unit_ids,                \
                     (uint32_t*)tile_summary, units_min,                                                          \
                     take_stamp_region(kStampGradLoc, blocks),                                                   \
                     FusedArgs{nullptr, nullptr, 0, 0, (float*)qsplit_zero, tile_copy,                            \
                               tile_copy ? tile_copy + 2 * (int64_t(d.B) * d.Lq * d.M * d.L * d.P) : nullptr}
#define VNX_LAUNCH(LPT, AT)                                                                     \
  hipLaunchKernelGGL((msda_bwd_d32_kernel<TV, TL, QPW, WPB, LPT, AT, false, ((LPT) == 16 && !(AT)) ? kLpr : 8>), dim3(uint32_t(blocks)),   \
                     dim3(64 * WPB), lds, stream, VNX_BWD_ARGS)
  if (!atomics) { if (LP == 16) VNX_LAUNCH(16, false); else VNX_LAUNCH(0, false); }
  else { if (LP == 16) VNX_LAUNCH(16, true); else VNX_LAUNCH(0, true); }
#undef VNX_LAUNCH
#undef VNX_BWD_ARGS
  return check_launch("msda_bwd_d32");
}

template <typename TV, typename TL>
static int launch_bwd(const void* value, const int64_t* shapes, const int64_t* lsi,
                      const void* loc, const void* attn, const void* grad_out, void* gv,
                      void* grad_loc, void* grad_attn, const MsdaDims& d, int variant, void* records,
                      void* tile_summary, float* tile_copy, hipStream_t stream) {
  // variant 100+v: ablation without the grad_value atomics (timing only, wrong grad_value)
  const bool atomics = variant < 100;
  const FwdCfg c = pick_fwd_cfg(d, atomics ? variant : variant - 100);
#define VNX_CASE(Q, W)                                                                       \
  if (c.qpw == Q && c.wpb == W)                                                              \
    return launch_bwd_cfg<TV, TL, Q, W>(value, shapes, lsi, loc, attn, grad_out, gv, grad_loc, \
                                        grad_attn, d, atomics, records, tile_summary, tile_copy, stream);
  VNX_CASE(8, 4) VNX_CASE(4, 4) VNX_CASE(2, 4) VNX_CASE(1, 4)
  VNX_CASE(8, 1) VNX_CASE(4, 1) VNX_CASE(2, 1) VNX_CASE(1, 1)
  VNX_CASE(4, 2)
#undef VNX_CASE
  set_error("msda_backward: no kernel for qpw=%d wpb=%d", c.qpw, c.wpb);
  return VNX_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------------------------
// The backward of a call below 1 024 queries as ONE launch (round 5).  Its two halves -- grad_value by the self-decoding
// kernel (msda_d32_gvdirect_body.h), grad_loc / grad_attn by the per-query gather kernel above -- read the same inputs and
// write disjoint outputs, so they can share the GPU; but two kernels on one stream run strictly one after the other, a fork
// onto a second stream costs more than it gains on this runtime and hipExtAnyOrderLaunch is not honoured on gfx9 (DESIGN.md
// section 3.3e).  Here some workgroups of the grid are grad_value units and the others run the grad_loc kernel's one-wave
// configuration in each of their eight waves: wave w of grad_loc workgroup (g, x) takes the virtual workgroup (8 g + w) M + x
// -- the same head class x for all eight, so the head <-> XCD map of both kernels is kept.  Registers and LDS are the
// maximum of the two roles: two workgroups per CU either way.
constexpr int kPairWaves = 8;
// samples whose tap loads a grad_loc wave has in flight: three CUs' worth of workgroups need <= 80 VGPRs (the stand-alone kernel
// takes 4 = 16 loads: 92 VGPRs; fp32 3 + 3 + 2: 76; 16-bit rows, 4 samples per group: 2 + 2)
#ifndef VNX_PAIR_BATCH_F32
#define VNX_PAIR_BATCH_F32 3
#endif
template <typename TV> constexpr int kPairBatch = sizeof(TV) == 2 ? 2 : VNX_PAIR_BATCH_F32;
template <int QPW> constexpr int kPairGlWaveLds = 3 * QPW * 17 * 16 + 128;      // the one-wave configuration's LDS (L*P == 16)

template <typename TV, typename TL, int QPW, int LPR>
#ifndef VNX_PAIR_WGS_PER_CU
#define VNX_PAIR_WGS_PER_CU 3
#endif
__global__ void __launch_bounds__(64 * kPairWaves, VNX_PAIR_WGS_PER_CU * kPairWaves / 4)
msda_bwd_pair_kernel(const TV* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                     const TL* __restrict__ loc, const TL* __restrict__ attn, const TV* __restrict__ grad_out,
                     TV* __restrict__ grad_value, TL* __restrict__ grad_loc, TL* __restrict__ grad_attn, MsdaDims d,
                     int tiles_per_batch, int ut, int rows_max, uint32_t gv_groups, uint32_t gl_groups, uint32_t gl_blocks, int order,
                     unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  stamp_begin(stamps);
  // Roles by GROUPS of M consecutive workgroups (one per head class / XCD).  order 1 (what the library uses): all grad_loc
  // groups first -- their waves are bound by the latency of two cold gather rounds, so they start at once and the grad_value
  // units fill the CUs as they drain; 0: the grad_value groups first; 2: alternating while both last.  Measured (kbench, cold,
  // backward alone / fwd+bwd step; two launches 22.7 / 28.8 us): order 1 21.4 / 26.8, order 0 24.1 / 26.6, order 2 24.8 / 31.4
  // at the T = 5 decoder call; B = 10 39.5 / 49.7 -> 36.2 / 47.5 (0: 40.5 / 46.9, 2: 45.4 / 55.7); 720p 41.5 -> 35.7 (0: 40.7, 2: 38.2).
  // (Also measured and dropped: the units of the coarse levels -- the longest -- in front of the grad_loc groups, 21.9 / 29.4.)
  const uint32_t M = uint32_t(d.M), G = blockIdx.x / M, x = blockIdx.x - G * M;
  bool is_gv;
  uint32_t g;      // the group's index within its role
  // (Round 6, on the bench's step: only the coarse levels' units ahead of the grad_loc groups and the level-0 units behind them --
  //  2 / 4 / 6 / 8 unit groups per batch element in front: 6.78 / 6.96 / 6.41 / 6.01 Gpoints/s against 7.72 with all of them in front.)
  if (order == 0) { is_gv = G < gv_groups; g = is_gv ? G : G - gv_groups; }
  else if (order == 1) { is_gv = G >= gl_groups; g = is_gv ? G - gl_groups : G; }
  else {
    const uint32_t n_min = gv_groups < gl_groups ? gv_groups : gl_groups;
    if (G < 2 * n_min) { is_gv = (G & 1u) == 0u; g = G >> 1; }
    else { is_gv = gv_groups > gl_groups; g = G - n_min; }
  }
  if (is_gv) {      // uniform over the workgroup
    rec::msda_bwd_gv_direct_body<TV, TL, 4>(shapes, lsi, loc, attn, grad_out, grad_value, d, ut, rows_max, 0, nullptr, g * M + x, smem);
    stamp_end(stamps);
    return;
  }
  const uint32_t w = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
  const uint32_t vb = (g * kPairWaves + w) * M + x;
  if (vb < gl_blocks)      // uniform over the wave; the body has no workgroup barrier in this configuration
    msda_bwd_d32_body<TV, TL, QPW, 1, 16, false, false, LPR, kPairBatch<TV>>(value, shapes, lsi, loc, attn, grad_out, nullptr, grad_loc, grad_attn, d,
                                                            tiles_per_batch, nullptr, nullptr, nullptr, 0, nullptr, FusedArgs{}, vb, 0,
                                                            smem + w * kPairGlWaveLds<QPW>);
  stamp_end(stamps);
}

int msda_gvdirect_units_bound(const MsdaDims& d, int ut, int rows);

// -> true when the call is one the paired kernel is built for (what the decoders of both models present: fp32 locations, fp32
// or 16-bit values, L*P == 16 with 4 points, the one-wave configuration of the grad_loc kernel)
bool msda_backward_pair_supported(int vdt, int ldt, const MsdaDims& d) {
  if (ldt != VNX_F32 || (vdt != VNX_F32 && vdt != VNX_BF16 && vdt != VNX_F16)) return false;
  if (d.P != 4 || d.L * d.P != 16 || d.Lq >= 1024) return false;
  const FwdCfg c = pick_fwd_cfg(d, 0);
  return c.qpw == 4 && c.wpb == 1;
}

template <typename TV>
static int launch_pair(const void* value, const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                       const void* grad_out, void* grad_value, void* grad_loc, void* grad_attn, const MsdaDims& d, int order,
                       hipStream_t stream) {      // order: 0 / 1 / 2 as the kernel's, < 0 = chosen here
  constexpr int QPW = 4;
  constexpr int kLpr = sizeof(TV) == 2 ? 4 : 8;      // 16-bit rows as 4 lanes x 16 B (as the stand-alone launcher)
  const int tiles_per_batch = (d.Lq + QPW - 1) / QPW;
  const int64_t gl_blocks = int64_t(d.B) * tiles_per_batch * d.M;
  const int64_t gl_groups = (gl_blocks + int64_t(kPairWaves) * d.M - 1) / (int64_t(kPairWaves) * d.M);      // of M workgroups each
  const int rows = gvd_rows_max(d.S);
#ifdef VNX_PAIR_UT      // A/B: units per level at least, forced
  const int ut = VNX_PAIR_UT;
#else
  const int ut = gvd_units_min(d.S, d.L, d.B * d.M, rows, gl_groups * d.M);      // (the small levels are cut only if BOTH roles then fit about one round)
#endif
  const int64_t gv_blocks = ((int64_t(d.B) * msda_gvdirect_units_bound(d, ut, rows) + 1) & ~int64_t(1)) * d.M;
  // Role order (see the kernel): with everything resident in ONE round of three workgroups per CU the grad_value groups go
  // first -- their units are the longest workgroups (T = 5 call, 735 workgroups: 19.5 us against 20.7 with the grad_loc groups
  // first); with more rounds the grad_loc groups (B = 10: 33.4 against 39.2 us).  order < 0: this choice.
  if (order < 0)
    order = int64_t(d.B) * d.M * gvd_units_estimate(d.S, d.L, rows) + gl_groups * d.M <= VNX_GVD_ONE_ROUND ? 0 : 1;
  const int64_t blocks = gv_blocks + gl_groups * d.M;
  if (blocks >= (int64_t(1) << 31)) {
    set_error("msda_backward: %lld workgroups exceed the grid limit", (long long)blocks);
    return VNX_ERR_UNSUPPORTED;
  }
  constexpr size_t kLds = rec::kGvdLdsBytes > size_t(kPairWaves) * kPairGlWaveLds<QPW> ? rec::kGvdLdsBytes
                                                                                           : size_t(kPairWaves) * kPairGlWaveLds<QPW>;
  int dev = 0;
  (void)hipGetDevice(&dev);
  static thread_local int raised_on = -1;      // more than 64 KiB of LDS per workgroup: the limit is raised once per device
  if (kLds > 64 * 1024 && raised_on != dev) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&msda_bwd_pair_kernel<TV, float, QPW, kLpr>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, int(kLds)) != hipSuccess)
      return check_launch("msda_bwd_pair (LDS limit)");
    raised_on = dev;
  }
  hipLaunchKernelGGL((msda_bwd_pair_kernel<TV, float, QPW, kLpr>), dim3(uint32_t(blocks)), dim3(64 * kPairWaves), kLds, stream,
                     (const TV*)value, shapes, lsi, (const float*)loc, (const float*)attn, (const TV*)grad_out,
                     (TV*)grad_value, (float*)grad_loc, (float*)grad_attn, d, tiles_per_batch, ut, rows, uint32_t(gv_blocks / d.M),
                     uint32_t(gl_groups), uint32_t(gl_blocks), order, take_stamp_region(kStampGradPair, blocks));
  return check_launch("msda_bwd_pair");
}

int msda_backward_pair_d32(int vdt, const void* value, const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                           const void* grad_out, void* grad_value, void* grad_loc, void* grad_attn, MsdaDims d, int order,
                           hipStream_t stream) {
#define VNX_ARGS value, shapes, lsi, loc, attn, grad_out, grad_value, grad_loc, grad_attn, d, order, stream
  if (vdt == VNX_F32) return launch_pair<float>(VNX_ARGS);
  if (vdt == VNX_BF16) return launch_pair<bf16_t>(VNX_ARGS);
  if (vdt == VNX_F16) return launch_pair<f16_t>(VNX_ARGS);
#undef VNX_ARGS
  set_error("msda_backward_pair_d32: unsupported value dtype %d", vdt);
  return VNX_ERR_INVALID_ARGUMENT;
}

// queries per workgroup of the grad_loc kernel = per tile word of the tile mode (the launcher's own choice)
// ---- grad_loc / grad_attn of calls with many queries, the coarse levels staged whole in LDS: the backward twin of
// msda_fwd_slab_kernel (see there).  Same workgroup shape, slab and grid; a wave decodes 8 queries (two passes of 4 = two tile
// boxes for the tile-fed grad_value kernel, msda_d32_gvtiles.hip), an 8-lane set then takes the four dots <grad_out row, tap
// row> of each of its query's 16 samples -- rows of the staged levels from LDS, the others gathered -- and the decode's lane
// of each sample combines them (cuh:123-158) and stores the gradients once, non-temporal.  The dots of a sample overwrite its
// tap offsets in the wave's records (they are in registers by then), so the LDS budget is the forward's.
// FUSED (round 6; what the models' encoder layers call): `loc` / `attn` are the raw offsets and attention logits of the fused
// prologue (fused_decode: reference point + scaled offset, softmax over a (query, head)'s 16 logits), the decoded locations
// and weights are left for the tile-fed grad_value kernel in fa.tile_loc / fa.tile_attn, and phase 3 applies the chain rule
// through the prologue as the gather kernel's FUSED instantiation does (msda_bwd_d32_body).
template <bool FUSED>
__global__ void __launch_bounds__(64 * kSlabWaves, 2 * kSlabWaves / 4)
msda_bwd_slab_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                     const float* __restrict__ loc, const float* __restrict__ attn, const float* __restrict__ grad_out,
                     float* __restrict__ grad_loc, float* __restrict__ grad_attn, uint32_t* __restrict__ tile_summary, MsdaDims d,
                     int parts, unsigned long long* stamps, FusedArgs fa) {
  stamp_begin(stamps);
  constexpr int D = 32, LP = 16, kRowBytes = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* slab = smem;
  uint4_t* rec_base = reinterpret_cast<uint4_t*>(smem + size_t(kSlabRowsCap + 1) * kRowBytes);
  int* s_lvl = reinterpret_cast<int*>(smem + size_t(kSlabRowsCap + 1) * kRowBytes + kSlabRecBytes);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = blockIdx.x % d.M, rest = blockIdx.x / d.M, b = rest / parts, part = rest % parts;      // batch-major: see the forward
  const int m = (x + b) % d.M;
  if (tid < 4) {
    s_lvl[4 * tid] = int(shapes[2 * tid]); s_lvl[4 * tid + 1] = int(shapes[2 * tid + 1]); s_lvl[4 * tid + 2] = int(lsi[tid]);
  }
  __syncthreads();
  int first_staged = 4, slab_first = d.S;
  {
    bool packed = true;
    int running = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) { packed = packed && s_lvl[4 * l + 2] == running; running += s_lvl[4 * l] * s_lvl[4 * l + 1]; }
    packed = packed && running == d.S;
    if (packed) {
#pragma unroll
      for (int l = 3; l >= 0; --l)
        if (d.S - s_lvl[4 * l + 2] <= kSlabRowsCap) { first_staged = l; slab_first = s_lvl[4 * l + 2]; }
    }
  }
  first_staged = __builtin_amdgcn_readfirstlane(first_staged);
  slab_first = __builtin_amdgcn_readfirstlane(slab_first);
  const int n_slab = d.S - slab_first;
  {
    const float* src = value + ((int64_t(b) * d.S + slab_first) * d.M + m) * D;
    for (int i = tid; i < n_slab * 8; i += 64 * kSlabWaves)
      *reinterpret_cast<float4_t*>(slab + (i >> 3) * kRowBytes + (i & 7) * 16) =
          *reinterpret_cast<const float4_t*>(src + int64_t(i >> 3) * d.M * D + (i & 7) * 4);
    if (tid < 8) *reinterpret_cast<float4_t*>(slab + n_slab * kRowBytes + tid * 16) = float4_t{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  const uint32_t zero_row = uint32_t(n_slab) * kRowBytes;

  uint4_t* s_off = rec_base + size_t(wave) * 2 * kSlabEnt;            // tap offsets, then the four dots of the sample
  float4_t* s_geo = reinterpret_cast<float4_t*>(s_off + kSlabEnt);    // lh, lw, attention weight
  const int pixel_bytes = d.M * kRowBytes;
  const float* head_base = value + (int64_t(b) * d.S * d.M + m) * D;
  const __amdgpu_buffer_rsrc_t rsrc = uniform_rsrc(head_base, uint32_t((int64_t(d.S) * d.M - m) * kRowBytes));
  const int ch = lane & 7, qi2 = lane >> 3;
  const uint32_t lane_off = uint32_t(ch * 16);
  uint2_t* tile_words = reinterpret_cast<uint2_t*>(tile_summary);      // [b][head][level][4-query tile]
  const int wn = (d.Lq + 3) / 4;

  const int n_tiles = (d.Lq + kSlabWaves * kSlabQpw - 1) / (kSlabWaves * kSlabQpw);
  for (int t = part; t < n_tiles; t += parts) {
    const int q0 = (t * kSlabWaves + wave) * kSlabQpw;
    // ---- phase 1: decode; the bounding box of each 4-query tile's taps, per level ----
#pragma unroll 1
    for (int e0 = 0; e0 < kSlabQpw * LP; e0 += 64) {
      const int e = e0 + lane, qi = e >> 4, p = e & 15, l = p >> 2;
      const int q = q0 + qi;
      const bool staged = l >= first_staged;
      const uint32_t none = staged ? zero_row : kTapOutside;
      uint4_t o4 = {none, none, none, none};
      float4_t g4 = {0.f, 0.f, 0.f, 0.f};
      uint32_t tile_kx = 0xffffffffu, tile_ky = 0xffffffffu;
      const int64_t wi = ((int64_t(b) * d.Lq + q) * d.M + m) * LP + p;
      float sx = 0.f, sy = 0.f, a = 0.f;
      if constexpr (FUSED) {      // (all 16 lanes of a (query, head) row together: q is uniform over them)
        const bool valid = q < d.Lq;
        float dx, dy;
        fused_decode<float>(loc, attn, fa, wi, b, q, l, valid ? s_lvl[4 * l] : 1, valid ? s_lvl[4 * l + 1] : 1, d, valid, sx, sy, a, dx, dy);
        g4.w = a;      // the softmax weight of every sample, in or out of the map (softmax backward, phase 3)
        if (valid && fa.tile_loc != nullptr) {      // [batch][head][level][query][point]: what the tile-fed grad_value kernel walks
          const int64_t ci = ((int64_t(b) * d.M + m) * 4 + l) * (int64_t(d.Lq) * 4) + int64_t(q) * 4 + (p & 3);
          *reinterpret_cast<float2_t*>(fa.tile_loc + 2 * ci) = float2_t{sx, sy};
          fa.tile_attn[ci] = a;
        }
      }
      if (q < d.Lq) {
        const int H = s_lvl[4 * l], W = s_lvl[4 * l + 1], start = s_lvl[4 * l + 2];
        if constexpr (!FUSED) { sx = loc[2 * wi]; sy = loc[2 * wi + 1]; a = attn[wi]; }
        const float h = sy * float(H) - 0.5f, w = sx * float(W) - 0.5f;
        if (h > -1.f && w > -1.f && h < float(H) && w < float(W)) {
          const float hf = floorf(h), wf = floorf(w);
          const int h0 = int(hf), w0 = int(wf);
          const bool top = h0 >= 0, bot = h0 + 1 <= H - 1, lef = w0 >= 0, rig = w0 + 1 <= W - 1;
          const uint32_t pb = staged ? uint32_t(kRowBytes) : uint32_t(pixel_bytes);
          const uint32_t o00 = uint32_t(__mul24((staged ? start - slab_first : start) + __mul24(h0, W) + w0, int(pb)));
          const uint32_t row_b = __umul24(uint32_t(W), pb);
          o4.x = (top && lef) ? o00 : none;
          o4.y = (top && rig) ? o00 + pb : none;
          o4.z = (bot && lef) ? o00 + row_b : none;
          o4.w = (bot && rig) ? o00 + row_b + pb : none;
          g4.x = h - hf; g4.y = w - wf; g4.z = a;
          if (tile_summary != nullptr) {      // as msda_bwd_d32_body: coordinates saturate at 0xfffe
            const uint32_t x_lo = min(uint32_t(lef ? w0 : w0 + 1), 0xfffeu), x_hi = min(uint32_t(rig ? w0 + 1 : w0), 0xfffeu);
            const uint32_t y_lo = min(uint32_t(top ? h0 : h0 + 1), 0xfffeu), y_hi = min(uint32_t(bot ? h0 + 1 : h0), 0xfffeu);
            tile_kx = x_lo | ((0xffffu - x_hi) << 16);
            tile_ky = y_lo | ((0xffffu - y_hi) << 16);
          }
        }
      }
      s_off[qi * 17 + p] = o4;
      s_geo[qi * 17 + p] = g4;
      if (tile_summary != nullptr) {      // uniform.  Minimum over a level's 4 points (quad) and the pass's 4 queries (lane bits 4, 5)
        tile_kx = pk_min_u16(tile_kx, uint32_t(__builtin_amdgcn_update_dpp(int(tile_kx), int(tile_kx), 0xB1, 0xF, 0xF, true)));
        tile_ky = pk_min_u16(tile_ky, uint32_t(__builtin_amdgcn_update_dpp(int(tile_ky), int(tile_ky), 0xB1, 0xF, 0xF, true)));
        tile_kx = pk_min_u16(tile_kx, uint32_t(__builtin_amdgcn_update_dpp(int(tile_kx), int(tile_kx), 0x4E, 0xF, 0xF, true)));
        tile_ky = pk_min_u16(tile_ky, uint32_t(__builtin_amdgcn_update_dpp(int(tile_ky), int(tile_ky), 0x4E, 0xF, 0xF, true)));
        tile_kx = pk_min_u16(tile_kx, uint32_t(__shfl_xor(int(tile_kx), 16, 64)));
        tile_ky = pk_min_u16(tile_ky, uint32_t(__shfl_xor(int(tile_ky), 16, 64)));
        tile_kx = pk_min_u16(tile_kx, uint32_t(__shfl_xor(int(tile_kx), 32, 64)));
        tile_ky = pk_min_u16(tile_ky, uint32_t(__shfl_xor(int(tile_ky), 32, 64)));
        const int wt = (q0 >> 2) + (e0 >> 6);      // this pass's 4-query tile
        if ((lane & 3) == 0 && lane < 16 && wt < wn)
          tile_words[((int64_t(b) * d.M + m) * 4 + (lane >> 2)) * wn + wt] = uint2_t{tile_kx, tile_ky};
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- phase 2: the four dots of every sample ----
    {
      uint4_t* g_off = s_off + qi2 * 17;
      const int q = q0 + qi2;
      float4_t top = {0.f, 0.f, 0.f, 0.f};
      if (q < d.Lq) top = *reinterpret_cast<const float4_t*>(grad_out + ((int64_t(b) * d.Lq + q) * d.M + m) * D + ch * 4);
      const float2_t t_lo = {top.x, top.y}, t_hi = {top.z, top.w};
      auto dot = [&](const float4_t v) {
        float2_t acc = t_lo * float2_t{v.x, v.y};
        acc = t_hi * float2_t{v.z, v.w} + acc;
        return acc.x + acc.y;
      };
#pragma unroll 1
      for (int l = 0; l < 4; ++l) {
        uint4_t o[4];
        float4_t v[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = g_off[4 * l + j];
        if (l < first_staged) {      // uniform
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j][0] = load_tap<float>(rsrc, o[j].x + lane_off);
            v[j][1] = load_tap<float>(rsrc, o[j].y + lane_off);
            v[j][2] = load_tap<float>(rsrc, o[j].z + lane_off);
            v[j][3] = load_tap<float>(rsrc, o[j].w + lane_off);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j][0] = *reinterpret_cast<const float4_t*>(slab + o[j].x + lane_off);
            v[j][1] = *reinterpret_cast<const float4_t*>(slab + o[j].y + lane_off);
            v[j][2] = *reinterpret_cast<const float4_t*>(slab + o[j].z + lane_off);
            v[j][3] = *reinterpret_cast<const float4_t*>(slab + o[j].w + lane_off);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4_t dd = {dot(v[j][0]), dot(v[j][1]), dot(v[j][2]), dot(v[j][3])};
          float t0, t1;
          group8_sum4_split(dd, t0, t1);        // lanes 0..3 of the set: (d1, d2); lanes 4..7: (d3, d4)
          if ((ch & 3) == 0) reinterpret_cast<float2_t*>(g_off + 4 * l + j)[ch >> 2] = float2_t{t0, t1};
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- phase 3: the decode's lane of a sample combines its four dots (cuh:123-158) ----
#pragma unroll 1
    for (int e0 = 0; e0 < kSlabQpw * LP; e0 += 64) {
      const int e = e0 + lane, qi = e >> 4, p = e & 15, l = p >> 2;
      const int q = q0 + qi;
      if (q < d.Lq) {
        const int64_t wi = ((int64_t(b) * d.Lq + q) * d.M + m) * LP + p;
        const float4_t r = __builtin_bit_cast(float4_t, s_off[qi * 17 + p]);
        const float4_t gq = s_geo[qi * 17 + p];
        const float lh = gq.x, lw = gq.y, a = gq.z, hh = 1.f - lh, hw = 1.f - lw;
        const float d1 = r.x, d2 = r.y, d3 = r.z, d4 = r.w;
        const float gw = a * (hh * (d2 - d1) + lh * (d4 - d3));
        const float gh = a * (hw * (d3 - d1) + lw * (d4 - d2));
        const float ga = hh * (hw * d1 + lw * d2) + lh * (hw * d3 + lw * d4);
        const float Hf = float(s_lvl[4 * l]), Wf = float(s_lvl[4 * l + 1]);
        if constexpr (!FUSED) {
          __builtin_nontemporal_store(float2_t{Wf * gw, Hf * gh}, reinterpret_cast<float2_t*>(grad_loc + 2 * wi));
          __builtin_nontemporal_store(ga, grad_attn + wi);
        } else {
          // chain rule through the prologue (as msda_bwd_d32_body): d loc / d offset is a per-sample scale, the softmax backward
          // g_logit = a (g_a - sum_j a_j g_a_j) a 16-lane row sum, and 2-d reference points collect the location gradients of
          // their level over points (lanes) and heads (atomics)
          const float gx = Wf * gw, gy = Hf * gh, aw = gq.w;
          const float dot = row16_sum(aw * ga);
          float kx, ky;
          if (fa.ref_dim == 2) {
            kx = 1.f / Wf; ky = 1.f / Hf;
          } else {
            const float* rf = static_cast<const float*>(fa.reference) + ((int64_t(b / fa.ref_div) * d.Lq + q) * d.L + l) * 4;
            kx = rf[2] * 0.5f / 4.f; ky = rf[3] * 0.5f / 4.f;
          }
          __builtin_nontemporal_store(float2_t{gx * kx, gy * ky}, reinterpret_cast<float2_t*>(grad_loc + 2 * wi));
          __builtin_nontemporal_store(aw * (ga - dot), grad_attn + wi);
          if (fa.grad_reference != nullptr) {
            float rx = gx, ry = gy;
            rx += __shfl_xor(rx, 1, 16); ry += __shfl_xor(ry, 1, 16);
            rx += __shfl_xor(rx, 2, 16); ry += __shfl_xor(ry, 2, 16);
            if ((p & 3) == 0) {
              float* gr = fa.grad_reference + ((int64_t(b) * d.Lq + q) * d.L + l) * 2;
              atomic_add(gr, rx);
              atomic_add(gr + 1, ry);
            }
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  stamp_end(stamps);
}

template <bool FUSED>
static int launch_bwd_slab(const void* value, const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                           const void* grad_out, void* grad_loc, void* grad_attn, void* tile_summary, const MsdaDims& d,
                           const FusedArgs& fa, hipStream_t stream) {
  const int n_tiles = (d.Lq + kSlabWaves * kSlabQpw - 1) / (kSlabWaves * kSlabQpw);
  const int parts = slab_parts(d, n_tiles);
  const int64_t blocks = int64_t(parts) * d.B * d.M;
  static thread_local int raised_on = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (raised_on != dev) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&msda_bwd_slab_kernel<FUSED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            int(kSlabLdsBytes)) != hipSuccess)
      return check_launch("msda_bwd_slab (LDS limit)");
    raised_on = dev;
  }
  hipLaunchKernelGGL(msda_bwd_slab_kernel<FUSED>, dim3(uint32_t(blocks)), dim3(64 * kSlabWaves), kSlabLdsBytes, stream,
                     (const float*)value, shapes, lsi, (const float*)loc, (const float*)attn, (const float*)grad_out,
                     (float*)grad_loc, (float*)grad_attn, (uint32_t*)tile_summary, d, parts,
                     take_stamp_region(kStampGradLoc, blocks), fa);
  return check_launch(FUSED ? "msda_bwd_slab_fused" : "msda_bwd_slab");
}

int msda_bwd_tile_queries(const MsdaDims& d, int variant) {
  // the grad_loc launcher's variant decoding: < 100 as is, 100..199 (grad_loc only, timing) minus 100, others automatic
  const FwdCfg c = pick_fwd_cfg(d, variant < 100 ? variant : (variant < 200 ? variant - 100 : 0));
  return kTilePerWave ? c.qpw : c.qpw * c.wpb;
}

bool msda_d32_bwd_supported(int vdt, int ldt, const MsdaDims& d) {
  if (!msda_d32_fwd_supported(vdt, ldt, d)) return false;
  // element offsets stay below the out-of-range marker (2^29 elements)
  return int64_t(d.S) * d.M * 32 < int64_t(kTapOutsideElem);
}

int msda_backward_d32(int vdt, int ldt, const void* value, const int64_t* shapes,
                      const int64_t* lsi, const void* loc, const void* attn,
                      const void* grad_out, void* gv, void* grad_loc, void* grad_attn, MsdaDims d,
                      int variant, void* records, void* tile_summary, float* tile_copy, hipStream_t stream) {
  // the 16-bit row limit of the sample records (h0+1, w0+1 packed into one word)
#define VNX_ARGS value, shapes, lsi, loc, attn, grad_out, gv, grad_loc, grad_attn, d, variant, records, tile_summary, tile_copy, stream
  // tile-fed calls (the encoders'), fp32: the coarse levels staged in LDS (msda_bwd_slab_kernel; its boxes are per 4 queries,
  // as the automatic configuration's).  The launcher's variant decoding: 100..199 = grad_loc only (timing), others automatic.
  // Development build: 730 forces, 731 / 733 forbid (733: this kernel only, the forward keeps its slab).
  {
    const int v = variant >= 100 && variant < 200 ? variant - 100 : (variant < 100 ? variant : 0);
    const int kv = kernel_variant();
    if (vdt == VNX_F32 && ldt == VNX_F32 && v == 0 && gv == nullptr && records == nullptr && tile_copy == nullptr &&
        tile_summary != nullptr && kv != 733 && msda_bwd_tile_queries(d, 0) == 4 && use_slab_forward(vdt, ldt, d, kv))
      return launch_bwd_slab<false>(value, shapes, lsi, loc, attn, grad_out, grad_loc, grad_attn, tile_summary, d, FusedArgs{}, stream);
  }
  if (vdt == VNX_F32) return launch_bwd<float, float>(VNX_ARGS);
  if (vdt == VNX_BF16 && ldt == VNX_F32) return launch_bwd<bf16_t, float>(VNX_ARGS);
  if (vdt == VNX_BF16 && ldt == VNX_BF16) return launch_bwd<bf16_t, bf16_t>(VNX_ARGS);
  if (vdt == VNX_F16 && ldt == VNX_F32) return launch_bwd<f16_t, float>(VNX_ARGS);
  if (vdt == VNX_F16 && ldt == VNX_F16) return launch_bwd<f16_t, f16_t>(VNX_ARGS);
#undef VNX_ARGS
  set_error("msda_backward_d32: unsupported dtype pair (%d, %d)", vdt, ldt);
  return VNX_ERR_INVALID_ARGUMENT;
}

// ---- fused prologue: launchers (L*P == 16, the three auto-selected configurations) -----------
template <typename TV, typename TL, int QPW, int WPB>
static int launch_fwd_fused_cfg(const void* value, const int64_t* shapes, const int64_t* lsi, const void* raw_off,
                                const void* raw_logit, void* out, const MsdaDims& d, const FusedArgs& fa,
                                hipStream_t stream) {
  const int tiles_per_batch = (d.Lq + QPW * WPB - 1) / (QPW * WPB);
  const int64_t blocks = int64_t(d.B) * tiles_per_batch * d.M;
  if (blocks >= (int64_t(1) << 31)) {
    set_error("msda_fused_forward: %lld workgroups exceed the grid limit", (long long)blocks);
    return VNX_ERR_UNSUPPORTED;
  }
  const size_t lds = size_t(WPB) * 2 * QPW * 17 * 16;
  constexpr int kLpr = sizeof(TV) == 2 ? 4 : 8;
  hipLaunchKernelGGL((msda_fwd_d32_kernel<TV, TL, QPW, WPB, 16, true, false, kLpr>), dim3(uint32_t(blocks)), dim3(64 * WPB), lds,
                     stream, (const TV*)value, shapes, lsi, (const TL*)raw_off, (const TL*)raw_logit, (TV*)out, d,
                     tiles_per_batch, 0, take_stamp_region(kStampFwd, blocks), fa);
  return check_launch("msda_fwd_d32_fused");
}

template <typename TV, typename TL, int QPW, int WPB>
static int launch_bwd_fused_cfg(const void* value, const int64_t* shapes, const int64_t* lsi, const void* raw_off,
                                const void* raw_logit, const void* grad_out, void* grad_off, void* grad_logit,
                                const MsdaDims& d, void* records, void* tile_summary, const FusedArgs& fa,
                                hipStream_t stream) {
  const int tiles_per_batch = (d.Lq + QPW * WPB - 1) / (QPW * WPB);
  const int64_t blocks = int64_t(d.B) * tiles_per_batch * d.M;
  if (blocks >= (int64_t(1) << 31)) {
    set_error("msda_fused_backward: %lld workgroups exceed the grid limit", (long long)blocks);
    return VNX_ERR_UNSUPPORTED;
  }
  const size_t lds = size_t(WPB) * 3 * QPW * 17 * 16 + 128;
  void* unit_ids = (records != nullptr && d.P == 4) ? (void*)((char*)records + gv_unit_ids_offset(d)) : nullptr;
  const int units_min = gv_units_min(d, tile_summary != nullptr, kernel_variant());
  constexpr int kLpr = (sizeof(TV) == 2 || VNX_K1_F32_LPR4(WPB)) ? 4 : 8;
  hipLaunchKernelGGL((msda_bwd_d32_kernel<TV, TL, QPW, WPB, 16, false, true, kLpr>), dim3(uint32_t(blocks)), dim3(64 * WPB),
                     lds, stream, (const TV*)value, shapes, lsi, (const TL*)raw_off, (const TL*)raw_logit,
                     (const TV*)grad_out, (float*)nullptr, (TL*)grad_off, (TL*)grad_logit, d, tiles_per_batch,
                     (uint4_t*)records, (uint32_t*)unit_ids, (uint32_t*)tile_summary, units_min,
                     take_stamp_region(kStampGradLoc, blocks), fa);
  return check_launch("msda_bwd_d32_fused");
}

template <typename TV, typename TL>
static int fused_dispatch(bool backward, const void* value, const int64_t* shapes, const int64_t* lsi,
                          const void* raw_off, const void* raw_logit, const void* grad_out, void* out_or_grad_off,
                          void* grad_logit, const MsdaDims& d, void* records, void* tile_summary, const FusedArgs& fa,
                          hipStream_t stream) {
  // encoder calls: the coarse levels staged in LDS (msda_fwd_slab_kernel; 16-bit values, and 16-bit offsets / logits: round 6)
  if (!backward && use_slab_forward(sizeof(TV) == 4 ? VNX_F32 : VNX_BF16, VNX_F32, d, kernel_variant()))
    return launch_fwd_slab<TV, TL>(value, shapes, lsi, raw_off, raw_logit, out_or_grad_off, d, &fa, stream);
  if constexpr (sizeof(TV) == 4 && sizeof(TL) == 4) {
    // ... their backward, tile-fed grad_value: msda_bwd_slab_kernel<true> was built and measured in round 6 (boxes per 4 queries, as
    // the automatic configuration's) and is NOT the product path -- kbench cold, fused backward, slab / gather form of the grad_loc
    // half: encoder-360p B = 5 175.2 / 171.0 us, B = 10 356.6 / 322.5, 720p B = 5 660.7 / 632.0, B = 2 287.0 / 286.0.  The fused
    // form decodes (exp, two 16-lane reductions per sample) and writes the 12-byte decoded copy of every sample on top of the
    // unfused kernel's work, at four waves per SIMD (128 VGPRs, 32 B of scratch): the slab's few per cent (66.0 against 69.7 us
    // unfused) do not survive it.  Development build only: variant 734.
    if (backward && records == nullptr && tile_summary != nullptr && fa.tile_loc != nullptr && fa.qsplit_zero == nullptr &&
        kernel_variant() == 734 && msda_bwd_tile_queries(d, 0) == 4 && use_slab_forward(VNX_F32, VNX_F32, d, 0))
      return launch_bwd_slab<true>(value, shapes, lsi, raw_off, raw_logit, grad_out, out_or_grad_off, grad_logit, tile_summary, d, fa, stream);
  }
  const FwdCfg c = pick_fwd_cfg(d, 0);
#define VNX_CASE(Q, W)                                                                                   \
  if (c.qpw == Q && c.wpb == W)                                                                          \
    return backward ? launch_bwd_fused_cfg<TV, TL, Q, W>(value, shapes, lsi, raw_off, raw_logit, grad_out, \
                                                         out_or_grad_off, grad_logit, d, records, tile_summary, fa, stream) \
                    : launch_fwd_fused_cfg<TV, TL, Q, W>(value, shapes, lsi, raw_off, raw_logit,           \
                                                         out_or_grad_off, d, fa, stream);
  VNX_CASE(4, 1) VNX_CASE(4, 4) VNX_CASE(8, 4)
#undef VNX_CASE
  set_error("msda_fused: no kernel for qpw=%d wpb=%d", c.qpw, c.wpb);
  return VNX_ERR_UNSUPPORTED;
}

bool msda_d32_fused_supported(int vdt, int ldt, const MsdaDims& d) {
  if (!msda_d32_bwd_supported(vdt, ldt, d) || d.L * d.P != 16) return false;
  return (vdt == VNX_F32 && ldt == VNX_F32) || (vdt == VNX_BF16 && (ldt == VNX_F32 || ldt == VNX_BF16));
}

int msda_fused_d32(bool backward, int vdt, int ldt, const void* value, const int64_t* shapes, const int64_t* lsi,
                   const void* raw_off, const void* raw_logit, const void* grad_out, void* out_or_grad_off,
                   void* grad_logit, MsdaDims d, void* records, const void* reference, float* grad_reference,
                   int ref_dim, int ref_div, void* grad_value_f32, void* tile_summary, float* tile_loc, float* tile_attn,
                   hipStream_t stream, int ref_f32) {
  // grad_value_f32: the fp32 target of the query-split levels' atomics (grad_value itself for fp32 values, the split
  // image for 16-bit ones), or null.  Backward: either `records` (record-fed grad_value kernel) or tile_summary +
  // tile_loc + tile_attn (tile-fed one: queries per tile = msda_bwd_tile_queries(d, 0))
  const FusedArgs fa{reference, grad_reference, ref_dim, ref_div, backward ? static_cast<float*>(grad_value_f32) : nullptr,
                     backward ? tile_loc : nullptr, backward ? tile_attn : nullptr, ref_f32};
#define VNX_ARGS backward, value, shapes, lsi, raw_off, raw_logit, grad_out, out_or_grad_off, grad_logit, d, records, tile_summary, fa, stream
  if (vdt == VNX_F32) return fused_dispatch<float, float>(VNX_ARGS);
  if (vdt == VNX_BF16 && ldt == VNX_F32) return fused_dispatch<bf16_t, float>(VNX_ARGS);
  if (vdt == VNX_BF16 && ldt == VNX_BF16) return fused_dispatch<bf16_t, bf16_t>(VNX_ARGS);
#undef VNX_ARGS
  set_error("msda_fused: unsupported dtype pair (%d, %d)", vdt, ldt);
  return VNX_ERR_UNSUPPORTED;
}

}  // namespace vnx
