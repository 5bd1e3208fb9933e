// vnx_common.h -- shared device/host helpers for libvnext_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/vnext_hip.h"
#include <atomic>

namespace vnx {

constexpr int kWave = 64;  // CDNA wavefront

// ---- element types ---------------------------------------------------------
struct bf16_t { uint16_t bits; };
struct f16_t { _Float16 v; };

template <typename T> struct acc_of { using type = float; };
template <> struct acc_of<double> { using type = double; };
template <typename T> using acc_t = typename acc_of<T>::type;

__device__ __forceinline__ float to_acc(float x) { return x; }
__device__ __forceinline__ double to_acc(double x) { return x; }
__device__ __forceinline__ float to_acc(bf16_t x) { return __uint_as_float(uint32_t(x.bits) << 16); }
__device__ __forceinline__ float to_acc(f16_t x) { return float(x.v); }

// fp32 -> bf16, round to nearest even: gfx950 has the instruction (v_cvt_pk_bf16_f32, two values per issue; the compiler
// emits it for this vector conversion).  Until round 3 this was integer arithmetic -- ~6 instructions per value, a
// third more vector instructions in the bf16 grad_value kernel than in the fp32 one (PMC: 14.1 M vs 11.0 M per launch at
// decoder-720p), which is why 16-bit values were SLOWER there.
typedef __bf16 vnx_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float vnx_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {       // lo in bits 0..15
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(vnx_f32x2_t{lo, hi}, vnx_bf16x2_t));
}
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) { return uint16_t(f32x2_to_bf16x2(f, 0.f)); }

// four consecutive channels of a row as fp32, whatever the storage type (fp32: 16 bytes, bf16 / f16: 8) -- the element-wise kernels
// of the layer stack (add_norm.hip, ffn_act.hip) compute in fp32 and round once on the way out
typedef float vnx_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t vnx_u2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ vnx_f4 row4_load(const T* p);
template <> __device__ __forceinline__ vnx_f4 row4_load<float>(const float* p) { return *reinterpret_cast<const vnx_f4*>(p); }
template <> __device__ __forceinline__ vnx_f4 row4_load<bf16_t>(const bf16_t* p) {
  const vnx_u2 r = *reinterpret_cast<const vnx_u2*>(p);
  return vnx_f4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}
template <> __device__ __forceinline__ vnx_f4 row4_load<f16_t>(const f16_t* p) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 r = *reinterpret_cast<const h4*>(p);
  return vnx_f4{float(r.x), float(r.y), float(r.z), float(r.w)};
}
template <typename T> __device__ __forceinline__ void row4_store(T* p, vnx_f4 v);
template <> __device__ __forceinline__ void row4_store<float>(float* p, vnx_f4 v) { *reinterpret_cast<vnx_f4*>(p) = v; }
template <typename T> __device__ __forceinline__ void row4_store_nt(T* p, vnx_f4 v);
template <> __device__ __forceinline__ void row4_store_nt<float>(float* p, vnx_f4 v) { __builtin_nontemporal_store(v, reinterpret_cast<vnx_f4*>(p)); }

template <typename T> __device__ __forceinline__ T from_acc(acc_t<T> x);
template <> __device__ __forceinline__ float from_acc<float>(float x) { return x; }
template <> __device__ __forceinline__ double from_acc<double>(double x) { return x; }
template <> __device__ __forceinline__ bf16_t from_acc<bf16_t>(float x) { return bf16_t{f32_to_bf16_bits(x)}; }
template <> __device__ __forceinline__ f16_t from_acc<f16_t>(float x) { return f16_t{_Float16(x)}; }
template <> __device__ __forceinline__ void row4_store<bf16_t>(bf16_t* p, vnx_f4 v) {
  *reinterpret_cast<vnx_u2*>(p) = vnx_u2{f32x2_to_bf16x2(v.x, v.y), f32x2_to_bf16x2(v.z, v.w)};
}
template <> __device__ __forceinline__ void row4_store<f16_t>(f16_t* p, vnx_f4 v) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  *reinterpret_cast<h4*>(p) = h4{_Float16(v.x), _Float16(v.y), _Float16(v.z), _Float16(v.w)};
}
template <> __device__ __forceinline__ void row4_store_nt<f16_t>(f16_t* p, vnx_f4 v) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(h4{_Float16(v.x), _Float16(v.y), _Float16(v.z), _Float16(v.w)}, reinterpret_cast<h4*>(p));
}
template <> __device__ __forceinline__ void row4_store_nt<bf16_t>(bf16_t* p, vnx_f4 v) {
  __builtin_nontemporal_store(vnx_u2{f32x2_to_bf16x2(v.x, v.y), f32x2_to_bf16x2(v.z, v.w)}, reinterpret_cast<vnx_u2*>(p));
}

// hardware floating-point atomics (global_atomic_add_f32 / _f64), agent scope
__device__ __forceinline__ void atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// True when the levels tile [0, S) back to back in order (level_start_index[l] is the
// running sum of H*W and the total is spatial_size).
__device__ __forceinline__ bool levels_packed(const int64_t* shapes, const int64_t* lsi, int L, int S) {
  int64_t running = 0;
  bool ok = true;
  for (int l = 0; l < L; ++l) {
    ok = ok && (lsi[l] == running);
    running += shapes[2 * l] * shapes[2 * l + 1];
  }
  return ok && running == S;
}

__device__ __forceinline__ float floor_acc(float x) { return floorf(x); }
__device__ __forceinline__ double floor_acc(double x) { return floor(x); }

// ---- host side ---------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Kernel-span stamps (measurement aid behind bench.py's roofline).  While a stamp buffer is armed
// (vnx_debug_arm_stamps) every launch of a tuned MSDA kernel is handed a region of 2 x gridDim
// 64-bit slots; each workgroup leaves {its start, its last wave's end} there in constant-rate
// wall-clock ticks (s_memrealtime).  Host side: span = max(end) - min(start) = the kernel's own
// duration on the device, without the inter-kernel gap that event timing around back-to-back
// launches includes.  Per-workgroup slots: no same-address atomics across workgroups.
// nullptr (the normal case) = no stamps.
enum StampKernel { kStampFwd = 1, kStampGradLoc = 2, kStampGradValue = 3, kStampGradPair = 4 };
unsigned long long* take_stamp_region(int kernel, long long blocks);
__device__ __forceinline__ void stamp_begin(unsigned long long* s) {
  if (s && threadIdx.x == 0) s[2 * size_t(blockIdx.x)] = (unsigned long long)wall_clock64();
}
__device__ __forceinline__ void stamp_end(unsigned long long* s) {
  if (s && (threadIdx.x & 63) == 0) atomicMax(s + 2 * size_t(blockIdx.x) + 1, (unsigned long long)wall_clock64());
}
// Kernel variant of this call: 0 = automatic.  The PRODUCT library has no other value -- the forced configurations, the
// archived kernels and the timing ablations (which return wrong results by construction) exist only in the development
// build (-DVNX_DEV_VARIANTS, libvnext_hip_dev.so, include/vnext_hip_dev.h), where vnx_set_kernel_variant sets a
// process-wide value that every entry point reads once.
#ifdef VNX_DEV_VARIANTS
extern std::atomic<int> g_kernel_variant;
inline int kernel_variant() { return g_kernel_variant.load(std::memory_order_relaxed); }
#else
constexpr int kernel_variant() { return 0; }
#endif

struct MsdaDims {
  int B, S, M, D, L, Lq, P;
};

// grad_value units: every level is split into at least this many.  Shared by the grad_loc kernel
// (which tags every sample with the units it touches) and the grad_value kernels.  2 (development
// build, variants 200+x: x).  `variant`: the value the entry point read (kernel_variant()).
int gv_units_min(const MsdaDims& d, bool tiles, int variant);
// Backward workspace of the record-fed path: [16-B sample records | 256-B aligned | 4-B unit ranges]
inline size_t gv_unit_ids_offset(const MsdaDims& d) {
  const size_t n = size_t(d.B) * d.M * d.L * d.Lq * d.P;
  return (16 * n + 255) & ~size_t(255);
}


// ---- buffer descriptors ---------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  // wave-uniform inputs only (callers pass readfirstlane'd values)
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, int(bytes), 0x00020000);
}

// Descriptor over [base, base+bytes) whose words the compiler can prove wave-uniform
// (readfirstlane of both pointer halves; through uint32_t so nothing sign-extends).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, uint32_t bytes) {
  const uint32_t lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uintptr_t(base)))));
  const uint32_t hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uintptr_t(base) >> 32))));
  const uint32_t n = uint32_t(__builtin_amdgcn_readfirstlane(int(bytes)));
  return make_rsrc(reinterpret_cast<const void*>(uintptr_t(lo) | (uintptr_t(hi) << 32)), n);
}

// The fused prologue's extra arguments (msda_d32.hip, msda_d32_tile.hip)
struct FusedArgs {
  const void* reference;   // [B / ref_div, Lq, L, ref_dim], same element type as the offsets
  float* grad_reference;   // [B, Lq, L, 2] fp32, zero-filled, accumulated over heads; or null
  int ref_dim;             // 2 or 4
  int ref_div;             // consecutive batch elements sharing one reference row (frames of a clip)
  float* qsplit_zero;      // backward, fp32 grad_value: rows of the query-split levels are zeroed here (or null)
  float* tile_loc;         // backward, tile mode of grad_value (msda_d32_gvtiles.hip): the decoded locations [B,Lq,M,L,P,2]
  float* tile_attn;        //   and softmax weights [B,Lq,M,L,P] are left here in fp32 for that kernel (or null)
  int ref_f32;             // the reference points are fp32 whatever the offsets' type (round 6: under autocast the Linears emit
                           // 16-bit offsets / logits while the reference points stay fp32 -- VNX_MSDA_REF_F32 of the C ABI)
};
// element i of the reference points as fp32
template <typename TL>
__device__ __forceinline__ float fused_ref(const FusedArgs& fa, int64_t i) {
  return fa.ref_f32 ? static_cast<const float*>(fa.reference)[i] : to_acc(static_cast<const TL*>(fa.reference)[i]);
}

// Query split of the grad_value units (msda_d32_gvrec.hip).  A level of at most two row-units (the coarse
// levels: 240 and 60 pixels at 360p) receives taps from EVERY query, so each of its units sorts one chunk
// per 128 queries -- forty at the encoder shape (Lq = 5100), against a dozen for a fine-level unit: the
// kernel's critical path (a level-3 unit ran ~100 us of the 190).  From 1024 queries up such a level is
// therefore also split by query range into pieces of about ten chunks, each its own workgroup, and the
// pieces meet in grad_value through fp32 atomics (1.5 M dwords per 360p encoder call) on rows the
// grad_loc kernel zeroed.  NOT at the decoder shape: there the kernel lasts 15 us, the contended,
// L2-bypassing atomics cost each piece 12-25 us, and the backward went from 32 to 49 us.
// fp32 grad_value and 4 points per level only (the selection kernel); 1 = no split.
#ifndef VNX_QS_COARSE
#define VNX_QS_COARSE 4         // query pieces of a coarse level (<= 2 row-units) at most, from 32 (batch, head) pairs up
#endif
#ifndef VNX_QS_COARSE_SMALL
#define VNX_QS_COARSE_SMALL 16  // ... with fewer pairs (the grid does not fill the chip: more, shorter pieces)
#endif
#ifndef VNX_QS_MID
#define VNX_QS_MID 2
#endif
#ifndef VNX_QS_CHUNKS
#define VNX_QS_CHUNKS 10      // a piece is given about this many chunks of 128 queries
#endif
// batch_heads = B x M: with fewer than 32 (batch, head) pairs the grid leaves workgroup slots empty, and the middle
// levels are split further: 8 pieces instead of 2 (round 2: 4 -- encoder backward at B = 2 123 -> 109 us at 360p, 450 ->
// 412 us at 720p; round 3, tile-fed path: 8 pieces 322 -> 288 us at 720p B = 2, 234 -> 192 us at B = 1, 360p unchanged; 16
// pieces 326 us); at B = 5 (a full grid) 4 pieces measured slower than 2 (168.8 -> 191.7 us at 360p, 625 -> 637 us at 720p).
// The unit split of a level of n pixels, the ONE definition shared by the grad_loc kernel (which zeroes the rows of
// query-split levels and tags every sample with the units it touches), the grad_value kernels' level tables and the
// launcher's grid bound: at most rows_max rows per unit (kGvRowsMax / kGvTileRowsMax by grad_value path), at least units_min units, then normalised so that no unit is
// empty (units = ceil(n / rows_per_unit)).  (Round 2 had the zeroing phase use the count BEFORE normalisation: with
// units_min = 5 a 16-pixel level gave 5 there and 4 in the grad_value kernel -- different sides of the query-split
// threshold, atomics onto rows nobody had zeroed.  Latent at the default units_min = 2)
// The corner (h0, w0) of a sample's 2 x 2 footprint, h0 in [-1, H - 1], w0 in [-1, W - 1], in one word of the 16-byte
// sample record (grad_loc kernel -> record-fed grad_value kernel): two 16-bit fields while both sides of the level are
// below 65 535 pixels, the flat index over a (H + 1) x (W + 1) grid otherwise (one integer division per sample to
// unpack -- levels that large do not occur in the models, but a 1 x 70 000 level used to come back silently wrong).
// 0xffffffff = sample outside the map (never produced by either form).
__device__ __forceinline__ bool gv_corner_is_flat(int H, int W) { return H >= 0xffff || W >= 0xffff; }
__device__ __forceinline__ uint32_t gv_pack_corner(int h0, int w0, int H, int W) {
  return gv_corner_is_flat(H, W) ? uint32_t(h0 + 1) * uint32_t(W + 1) + uint32_t(w0 + 1)
                                 : (uint32_t(h0 + 1) << 16) | uint32_t(w0 + 1);
}
__device__ __forceinline__ void gv_unpack_corner(uint32_t v, int H, int W, int& h0, int& w0) {
  if (gv_corner_is_flat(H, W)) {
    const uint32_t h1 = v / uint32_t(W + 1);
    h0 = int(h1) - 1; w0 = int(v - h1 * uint32_t(W + 1)) - 1;
  } else {
    h0 = int(v >> 16) - 1; w0 = int(v & 0xffffu) - 1;
  }
}

// Rows per unit: 320 for the record-fed kernel (5 rows per 8-lane group, 72 VGPRs, 3 units per CU; 19 units per (batch,
// head) at 360p = 760 workgroups, one round of the 768 resident), 256 for the tile-fed kernel (4 rows per group: 62 VGPRs
// and 39 KB of LDS, FOUR units per CU -- encoder-360p backward 190.5 -> 176.8 us, 720p B = 2 341 -> 325 us; at 720p B = 5
// the thinner strips cost 2 %: 718 -> 732 us).
constexpr int kGvRowsMax = 320;
constexpr int kGvTileRowsMax = 256;
struct GvSplit { int units, rpu; };
__host__ __device__ inline GvSplit gv_level_split(int n, int units_min, int rows_max) {
  GvSplit s{0, 1};
  if (n > 0) {
    int units = (n + rows_max - 1) / rows_max;
    if (units < units_min) units = units_min;
    if (units > n) units = n;
    s.rpu = (n + units - 1) / units;
    s.units = (n + s.rpu - 1) / s.rpu;
  }
  return s;
}

// Units of the self-decoding grad_value kernel (msda_d32_gvdirect.hip; calls below 1 024 queries -- the decoders').  A unit is
// a pixel range of one level and reads + decodes the level's samples of ALL queries itself, so what the kernel fetches grows
// with the number of units (every one of them stages all grad_out rows of its head): units are LARGE -- up to VNX_GVD_ROWS
// rows, 10 per (batch, head) at 360p -- and hold no rows in registers: after one sort an 8-lane group walks its rows one
// after the other and stores each at once.  Every level receives the same number of taps (4 x points x queries) whatever
// its size, so a row of a coarse level (80 taps at the 60-pixel level of a 300-query call) is spread over 1 << gshift
// adjacent groups whose partial sums meet in registers; `ut` (gvd_units_min) cuts such a level in two when the grid has room.
#ifndef VNX_GVD_QC
#define VNX_GVD_QC 304            // queries staged per pass (all 300 of a decoder call): 19 KiB of grad_out row halves in LDS
#endif
#ifndef VNX_GVD_ROWS
#define VNX_GVD_ROWS 768          // rows of a unit at most (what the kernel's LDS holds); the launchers choose <= this per call (gvd_rows_max)
#endif
#ifndef VNX_GVD_ROWS_LARGE
#define VNX_GVD_ROWS_LARGE 640    // ... for pyramids above VNX_GVD_SMALL_S pixels
#endif
#ifndef VNX_GVD_SMALL_S
#define VNX_GVD_SMALL_S 8192
#endif
#ifndef VNX_GVD_ONE_ROUND
#define VNX_GVD_ONE_ROUND 832     // workgroups resident at once -- 3 per CU x 256 CUs = 768 -- plus a twelfth: the T = 5 decoder call with
                                  // its small levels cut in two is 440 + 375 = 815 workgroups, and in the bench's fwd + bwd step that
                                  // is the best split measured (round 6, the product library built each way, same box: 7.73-7.76
                                  // Gpoints/s against 7.45-7.47 uncut -- the uncut call's longest units, whole coarse levels, end the
                                  // launch alone on their CUs and hold the next launch back -- 6.80 with three units per level, 6.45
                                  // with four; backward-only sequences prefer the uncut call, 19.5 against 21.5 us)
#endif
struct GvdSplit { int units, rpu, gshift; };
// Units per level at least.  A level that fits one unit (the 240- and 60-pixel levels at 360p) receives as many taps as a
// whole fine level, so its unit is the call's longest: when the grid has room for it in ONE round of resident workgroups such
// a level is cut in two (T = 5 decoder call: 10 -> 12 units per (batch, head), 480 workgroups, kernel 14.8 -> 13.6 us; at
// B = 10 the grid takes two rounds either way and the extra units cost 0.6 us: there 1).
// Rows per unit at most, chosen per call (round 6; measured with three workgroups per CU, paired backward, cold): 768 rows at
// the 360p pyramid (S = 5 100: 9 units per (batch, head) instead of 10 -- T = 5 call 21.9 -> 20.8 us with the grad_loc groups
// first, 19.5 with the grad_value groups first once everything is resident in one round; B = 10 35.4 -> 33.1), 640 at the
// 720p pyramid (S = 19 560: 35.1 against 40.3 us with 768 -- there a unit's time is its rows' stores).
__host__ __device__ inline int gvd_rows_max(int S) {
  const int r = S <= VNX_GVD_SMALL_S ? VNX_GVD_ROWS : VNX_GVD_ROWS_LARGE;
  return r < VNX_GVD_ROWS ? r : VNX_GVD_ROWS;
}
// Units per (batch, head) the launchers expect (an estimate for tuning choices: the host knows S and L, not the level sizes;
// pyramids whose levels shrink four-fold need S / rows + L - 1)
__host__ __device__ inline int gvd_units_estimate(int S, int L, int rows) { return S / rows + (L > 1 ? L - 1 : 1); }
__host__ __device__ inline int gvd_units_min(int S, int L, int batch_heads, int rows, int64_t other_workgroups = 0) {
  // (the host knows S and L, not the level sizes: S / ROWS + L bounds the units of the 640-row split; the levels that gain a
  //  unit are the one or two smallest -- a heuristic for a tuning choice, any value is correct)
  //  `other_workgroups`: what shares the round with the units -- the grad_loc workgroups of the paired kernel, msda_d32.hip)
  return int64_t(batch_heads) * (gvd_units_estimate(S, L, rows) + 2) + other_workgroups <= VNX_GVD_ONE_ROUND ? 2 : 1;
}
__host__ __device__ inline GvdSplit gvd_level_split(int n, int ut, int Lq, int P, int rows_max) {
  GvdSplit s{0, 1, 0};
  if (n <= 0) return s;
  int units = (n + rows_max - 1) / rows_max;
  if (units < ut) units = ut;
  if (units > n) units = n;
  s.rpu = (n + units - 1) / units;
  s.units = (n + s.rpu - 1) / s.rpu;
  // taps a row expects: 4 x points x (queries of a pass) / pixels of the level; from 16 up a row takes 2 groups, 32: 4, 64: 8
  const int64_t taps = int64_t(4) * P * (Lq < VNX_GVD_QC ? Lq : VNX_GVD_QC);
  while (s.gshift < 3 && taps >= (int64_t(16) << s.gshift) * n) ++s.gshift;
  return s;
}

// Units of the tile-fed grad_value kernel (msda_d32_gvtiles.hip): RECTANGLES of a level, at most rows_max pixels each.
// A narrow level (W < 32) is cut into bands of whole image rows; a wider one also into columns of about 32 pixels --
// blocks of 32 x 8: a band of a 160-pixel-wide level would be 1.6 image rows thin while samples reach +-6 rows, i.e.
// 8.5 bands' worth of query tiles would hit every band of the 720p level 0 where a block sees 4.7.  Measured (encoder
// backward, B = 5, cold): round 3 -- blocks from 128 pixels of width up 732 -> 691 us at 720p, from 64 pixels up 651 us and
// 176.7 -> 174.6 us at 360p; block width 16 instead of 32: 699 us.  Round 4 (pieces meeting in partial rows): blocks from
// 32 pixels of width up -- the 40-pixel levels (24 x 40 at 360p, 23 x 40 at 720p) become four 20 x 12 blocks instead of four
// 40 x 6 bands -- 163.5 -> 153.4 us at 360p, 611 -> 594 us at 720p B = 5, 272.7 -> 266.7 us at B = 2; from 16 pixels up:
// the same (a 20-pixel level fits one unit either way); block width 24: 154.8 / 590 / 262.5 us, 16: 162 / 617 / 412 us.
// Very flat levels (H < 8) get wider blocks so that the unit count stays ~ n / rows_max.  At least units_min units per
// level when it has the rows.
// Unit u of a level: block (u % nbx, u / nbx), pixels [bx * bw, ..) x [by * bh, ..), clipped to the level.
#ifndef VNX_GV_BLOCK_MINW
#define VNX_GV_BLOCK_MINW 32
#endif
#ifndef VNX_GV_BLOCK_W
#define VNX_GV_BLOCK_W 32
#endif
struct GvGrid { int nbx, nby, bw, bh; };
__host__ __device__ inline GvGrid gv_level_grid(int H, int W, int units_min, int rows_max) {
  GvGrid g{1, 1, 1, 1};
  if (H <= 0 || W <= 0) { g.nbx = g.nby = 0; return g; }
  if (W < VNX_GV_BLOCK_MINW && W <= rows_max) {
    g.nbx = 1; g.bw = W;
  } else {
    const int target = H >= rows_max / VNX_GV_BLOCK_W ? VNX_GV_BLOCK_W : rows_max / H;       // flat levels: wider blocks
    g.nbx = (W + target - 1) / target;
    g.bw = (W + g.nbx - 1) / g.nbx;
    g.nbx = (W + g.bw - 1) / g.bw;
  }
  g.bh = rows_max / g.bw;
  if (g.bh < 1) g.bh = 1;
  if (g.bh > H) g.bh = H;
  g.nby = (H + g.bh - 1) / g.bh;
  if (g.nbx * g.nby < units_min && g.nby < H) {           // small levels: at least units_min bands
    int want = (units_min + g.nbx - 1) / g.nbx;
    if (want > H) want = H;
    g.bh = (H + want - 1) / want;
    g.nby = (H + g.bh - 1) / g.bh;
  }
  return g;
}

// units of one level on either grad_value path: what decides its query split (gv_query_splits) -- shared by the grad_loc
// kernel (zeroing), the grad_value kernels and the 16-bit convert pass
__host__ __device__ inline int gv_level_units(int H, int W, int units_min, bool tiles) {
  if (tiles) {
    const GvGrid g = gv_level_grid(H, W, units_min, kGvTileRowsMax);
    return g.nbx * g.nby;
  }
  return gv_level_split(H * W, units_min, kGvRowsMax).units;
}

__host__ __device__ inline int gv_query_splits(int row_units, int Lq, int P, bool f32, int batch_heads) {
  if (!f32 || P != 4 || row_units > 4 || Lq < 1024) return 1;
  const int chunks = (Lq + 127) / 128;
  const int qs = (chunks + VNX_QS_CHUNKS - 1) / VNX_QS_CHUNKS;
  // round 4, pieces meeting in partial rows (no atomics): coarse cap 8 -> 4 from 32 pairs up (encoder-720p B = 5 611 -> 600 us;
  // 360p has 4 pieces either way), 8 -> 16 below (720p B = 2 272.7 -> 263.9 us; 16 at B = 5: 628 us, 4 at B = 2: 332 us)
  const int mid = batch_heads < 32 ? 4 * VNX_QS_MID : VNX_QS_MID;
  const int coarse = batch_heads < 32 ? VNX_QS_COARSE_SMALL : VNX_QS_COARSE;
  const int cap = row_units > 2 ? mid : coarse;          // middle levels: 3-4 row-units (960 pixels at 360p)
  return qs < 1 ? 1 : (qs > cap ? cap : qs);
}

// Tile-fed grad_value path: the query pieces of a split level do NOT meet through atomics (rounds 2-3 did: 4 M fp32 atomics
// per T=5 360p encoder call, issued 8 lanes x 16-B stride per row = the slow address pattern of tools/atomic_bench.hip,
// 80 G dwords/s: 44 us of the 87-us kernel, plus 7 us in the grad_loc kernel to zero the rows first -- found by timing
// ablations in round 4).  Every piece stores its rows, plain 16-B stores, into its own slab of fp32 PARTIAL rows in the
// workspace, and a small finishing kernel adds the pieces of each row in a fixed order and writes grad_value once (in its
// own dtype: the fp32 "split image" + convert pass of 16-bit values is gone too).  Layout per (batch, head): for every
// split level l in order, qs_l pieces x n_l pixels x 32 floats, starting at row  sum_{l' < l} qs_l' n_l'.
// A split level has at most 4 units of at most kGvTileRowsMax pixels (gv_query_splits), at most this many pieces:
__host__ __device__ inline int gv_split_pieces_max(int batch_heads) {          // = the caps of gv_query_splits
  const int mid = batch_heads < 32 ? 4 * VNX_QS_MID : VNX_QS_MID;
  const int coarse = batch_heads < 32 ? VNX_QS_COARSE_SMALL : VNX_QS_COARSE;
  return mid > coarse ? mid : coarse;
}
// A split level is either coarse (<= 2 units: <= 2 x 256 pixels, `coarse` pieces) or middle (3-4 units: <= 4 x 256 pixels,
// `mid` pieces), never both: a level contributes at most max(512 coarse, 1024 mid) rows, and all of them at most S x the
// larger piece count.  (Until round 4 the bound was min(S, 1024 L) x max(mid, coarse): twice this, 134 MB at B = 2 for the
// ~25 MB the pieces write, allocated by every backward call -- ADVICE r4.)
static_assert(VNX_QS_MID <= VNX_QS_COARSE && 4 * VNX_QS_MID <= VNX_QS_COARSE_SMALL,
              "msda_gvtiles_units_bound counts the extra workgroups of the split levels with the COARSE cap");
__host__ __device__ inline int64_t gv_partial_rows_bound(int S, int L, int batch_heads) {      // partial rows per (batch, head), an upper bound
  const int64_t mid = batch_heads < 32 ? 4 * VNX_QS_MID : VNX_QS_MID;
  const int64_t coarse = batch_heads < 32 ? VNX_QS_COARSE_SMALL : VNX_QS_COARSE;
  const int64_t per_level_c = int64_t(2) * kGvTileRowsMax * coarse, per_level_m = int64_t(4) * kGvTileRowsMax * mid;
  const int64_t by_levels = int64_t(L) * (per_level_c > per_level_m ? per_level_c : per_level_m);
  const int64_t by_pixels = int64_t(S) * (coarse > mid ? coarse : mid);
  return by_levels < by_pixels ? by_levels : by_pixels;
}

inline int elem_size(int dtype) {
  switch (dtype) {
    case VNX_F32: return 4;
    case VNX_F64: return 8;
    case VNX_BF16: return 2;
    case VNX_F16: return 2;
    default: return 0;
  }
}

}  // namespace vnx
