// msda_d32_tile.hip -- multi-scale deformable attention forward for the ENCODER shape (the query
// is a pixel of the pyramid, Lq == S): spatially tiled, with the bilinear neighbourhoods staged in LDS.
//
// Why: in an encoder call every pixel samples 8 heads x 4 levels x 4 points a few pixels around
// itself (ops/modules/ms_deform_attn.py:65-73 initialises the offsets to (k+1) steps along the head's
// direction; deformable_transformer.py:183-196 puts the reference point on the pixel centre).  The
// per-query gather kernel (msda_d32.hip) reads 4 x 128 B per sample from L2 whatever the locality:
// 1.67 GB of L2->CU traffic per T=5 360p call, bound by the ~35 TB/s the L2s deliver (60 us).  Here a
// workgroup owns one CELL of the pyramid -- the pixels of ALL levels whose centres fall into one 8x8
// block of level-0 pixels (64 + 16 + 4 + 1 queries on a dyadic pyramid) -- for one (batch, head), and
// per level:
//   1. decodes its samples once (one lane per (query, point)), reduces their bounding box over the
//      workgroup (packed 16-bit min/max, DPP) and picks a window of <= 320 pixel rows around it;
//   2. streams that window of `value` (rows of one head: 128 B every 1 KiB) into LDS with
//      `buffer_load_dwordx4 ... lds` -- no VGPR round trip; pixels outside the map are out-of-range
//      for the buffer descriptor and arrive as zeros, which is the reference's zero padding
//      (ms_deform_im2col_cuda.cuh:55-78);
//   3. gathers from LDS: an 8-lane set reads one 128-B row per `ds_read_b128`.  The sets are paired
//      (0,3) (1,2) (4,7) (5,6) -- the lane groups the LDS services together -- and a pair takes the
//      LEFT and RIGHT tap of the same sample: adjacent rows sit in opposite halves of the 64 banks,
//      so every read is conflict-free by construction (random row pairs would collide half the time).
// Samples whose taps leave the window ("far": a learned offset can be anything) are fetched from
// global memory by the same lanes, so the result is exact for any input; the window only decides speed.
// When the levels are not packed or do not add up to Lq the same kernel runs on linear blocks of 64
// queries (still correct, no locality to exploit).  fp32, 32-channel heads, 4 levels x 4 points.
#include "vnx_common.h"

namespace vnx {

namespace tile {

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));
typedef short short2_t __attribute__((ext_vector_type(2)));

constexpr int kWaves = 8, kThreads = 64 * kWaves;
constexpr int kL = 4, kP = 4;           // levels x points this kernel is built for
constexpr int kCell = 8;                // cell edge, in pixels of level 0
constexpr int kNq = 96;                 // queries per pass (a dyadic cell has 85)
constexpr int kGroups = kNq / (4 * kWaves);   // 4-query groups per wave and pass: 3
constexpr int kSets = kThreads / 8;     // 8-lane sets = rows staged per step: 64
// Two window buffers, levels alternate (0, 2 -> A; 1, 3 -> B): level l + 1 is staged while level l is
// gathered.  Sized for the encoder's neighbourhoods (cell edge 8 / 2^l pixels + the ~9-pixel spread of
// the sampling pattern + 1): 16 x 16 at the finest level, 12 x 10 at the next.
constexpr int kRowsA = 256, kRowsB = 128;
constexpr int kAllocB = ((kRowsB + kSets - 1) / kSets) * kSets;   // the last staging step may run over
static_assert(kRowsA % kSets == 0, "buffer A must hold whole staging steps (buffer B follows it)");
constexpr uint32_t kFar = 0x80000000u;
constexpr uint32_t kTapOutside = 0x80000000u;

// LDS layout (bytes)
constexpr int kOffZero = 0;                                   // one zero row (samples outside the map)
constexpr int kOffWinA = 128;
constexpr int kOffWinB = kOffWinA + kRowsA * 128;
constexpr int kOffRec = kOffWinB + kAllocB * 128;            // [2 buffers][2 sides][kNq * kP] x 16 B
constexpr int kRecStride = kNq * kP + 2;                     // records per (buffer, side); + 32 B: the left and the
                                                             // right record of a query sit 8 banks apart
constexpr int kOffQtab = kOffRec + 4 * kRecStride * 16;      // [kNq] int: global query index
constexpr int kOffBox = kOffQtab + kNq * 4;                  // [kL][kWaves][2] packed (x | y << 16) min, max
constexpr int kOffStat = kOffBox + kL * kWaves * 2 * 4;      // [kNq][2] softmax max, sum (fused)
constexpr int kLdsBytes = kOffStat + kNq * 8;
static_assert(kLdsBytes <= 76 * 1024, "two workgroups per CU (an 81 KB version ran one per CU)");

// a / b for 0 <= a < 2^24, 0 < b < 2^24: one reciprocal, one multiply, an exact remainder check
__device__ __forceinline__ int sdiv(int a, int b) {
  int q = int(float(a) * __frcp_rn(float(b)));
  int r = a - q * b;
  if (r < 0) { --q; r += b; }
  if (r >= b) ++q;
  return q;
}
__device__ __forceinline__ int sdiv_u(int a, int b) { return __builtin_amdgcn_readfirstlane(sdiv(a, b)); }
__device__ __forceinline__ int lane_value(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// first pixel of level extent `n` whose centre lies in cell c of `cells` equal cells of [0, 1)
__device__ __forceinline__ int cell_lo(int c, int cells, int n) {
  const int num = 2 * c * n - cells;          // (2x + 1) * cells >= 2 c n
  int x = num <= 0 ? 0 : sdiv(num + 2 * cells - 1, 2 * cells);
  x = x < n ? x : n;
  return c <= 0 ? 0 : (c >= cells ? n : x);
}

// minimum over the wave, valid in lane 63: six v_min_i32 with DPP operands (lanes without a source read
// the identity)
__device__ __forceinline__ int wave_min(int v) {
#define VNX_STEP(ctrl, rmask) v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, ctrl, rmask, 0xF, false));
  VNX_STEP(0x111, 0xF) VNX_STEP(0x112, 0xF) VNX_STEP(0x114, 0xF) VNX_STEP(0x118, 0xF)
  VNX_STEP(0x142, 0xA) VNX_STEP(0x143, 0xC)
#undef VNX_STEP
  return v;
}
// saturating: a map wider than 32 K pixels only makes the window choice meaningless, never the result
__device__ __forceinline__ uint32_t pk16(int x, int y) {
  x = x < -32768 ? -32768 : (x > 32767 ? 32767 : x);
  y = y < -32768 ? -32768 : (y > 32767 ? 32767 : y);
  return (uint32_t(x) & 0xffffu) | (uint32_t(y) << 16);
}
__device__ __forceinline__ int pk_x(uint32_t v) { return int(short(v & 0xffffu)); }
__device__ __forceinline__ int pk_y(uint32_t v) { return int(short(v >> 16)); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b)));
}

__device__ __forceinline__ float4_t gload(__amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
  const uint4_t r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, int(off), 0, 0);
  return float4_t{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
}
__device__ __forceinline__ float2_t gload2(__amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
  const uint2_t r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, int(off), 0, 0);
  return float2_t{__uint_as_float(r.x), __uint_as_float(r.y)};
}
__device__ __forceinline__ float gload1(__amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, int(off), 0, 0));
}
__device__ __forceinline__ float4_t lds4(uint32_t byte_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  return *reinterpret_cast<const float4_t*>(smem + byte_off);
}

// development aid (variants 701 / 702): shader-clock stamps of the first / second item of every workgroup
__device__ unsigned long long g_tile_stamps[2048 * 16];
#define VNX_TSTAMP(k)                                                                      \
  do {                                                                                     \
    if (debug && tid == 0 && item == wg + wgs * (debug - 1) && blockIdx.x < 2048)           \
      g_tile_stamps[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter();                  \
  } while (0)

struct Window { int x0, y0, w, h; };   // pixels [x0, x0 + w) x [y0, y0 + h); may include column / row -1 and W / H

template <bool FUSED>
__global__ void __launch_bounds__(kThreads, 4)
msda_fwd_tile_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                     const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                     const float* __restrict__ attn, float* __restrict__ out, MsdaDims d, FusedArgs fa, int debug) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* const qtab = reinterpret_cast<int*>(smem + kOffQtab);
  uint32_t* const box = reinterpret_cast<uint32_t*>(smem + kOffBox);
  float2_t* const stat = reinterpret_cast<float2_t*>(smem + kOffStat);
  uint4_t* const recs = reinterpret_cast<uint4_t*>(smem + kOffRec);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd_slot = blockIdx.x % d.M;
  const int wg = blockIdx.x / d.M, wgs = gridDim.x / d.M;

  // ---- level table (wave-uniform) ------------------------------------------------------------
  int Hs[kL], Ws[kL], St[kL];
  bool pyramid = true;
  {
    int running = 0;
#pragma unroll
    for (int l = 0; l < kL; ++l) {
      Hs[l] = int(shapes[2 * l]); Ws[l] = int(shapes[2 * l + 1]); St[l] = int(lsi[l]);
      pyramid = pyramid && St[l] == running && Hs[l] > 0 && Ws[l] > 0 && Hs[l] < 32000 && Ws[l] < 32000;
      running += Hs[l] * Ws[l];
    }
    pyramid = pyramid && running == d.Lq && running == d.S;
  }
  const int CX = pyramid ? sdiv_u(Ws[0] + kCell - 1, kCell) : 1, CY = pyramid ? sdiv_u(Hs[0] + kCell - 1, kCell) : 1;
  const int cells = pyramid ? CX * CY : sdiv_u(d.Lq + 63, 64);
  const int n_items = d.B * cells;
  // this lane's level (lanes 0..3 work out per-level quantities side by side; the rest follow lane 3)
  const int my_l = lane < kL ? lane : kL - 1;
  const int myH = my_l == 0 ? Hs[0] : my_l == 1 ? Hs[1] : my_l == 2 ? Hs[2] : Hs[3];
  const int myW = my_l == 0 ? Ws[0] : my_l == 1 ? Ws[1] : my_l == 2 ? Ws[2] : Ws[3];

  if (tid < 8) reinterpret_cast<float4_t*>(smem + kOffZero)[tid] = float4_t{0.f, 0.f, 0.f, 0.f};

  const uint32_t pixel_bytes = uint32_t(d.M) * 128u;
  const int ch = lane & 7, set = lane >> 3;
  const int side = (set >> 1) & 1;
  const int pair = ((set & 4) ? 2 : 0) + ((set & 1) ^ side);

  for (int item = wg; item < n_items; item += wgs) {
    const int b = sdiv_u(item, cells), cell = item - b * cells;
    const int m = xcd_slot;     // fixed head <-> XCD map here: rotating it with the batch element (msda_d32.hip) measured 70 vs 65.5 us
    VNX_TSTAMP(0);
    // ---- the cell's queries: lane l works out level l ------------------------------------------
    int xa[kL], ya[kL], nx[kL], nl[kL];
    int nq = 0;
    if (pyramid) {
      const int cy = sdiv_u(cell, CX), cx = cell - cy * CX;
      const int vxa = cell_lo(cx, CX, myW), vya = cell_lo(cy, CY, myH);
      const int vnx = cell_lo(cx + 1, CX, myW) - vxa, vny = cell_lo(cy + 1, CY, myH) - vya;
#pragma unroll
      for (int l = 0; l < kL; ++l) {
        xa[l] = lane_value(vxa, l); ya[l] = lane_value(vya, l);
        nx[l] = lane_value(vnx, l);
        nl[l] = nx[l] * lane_value(vny, l);
        nq += nl[l];
      }
    } else {
#pragma unroll
      for (int l = 0; l < kL; ++l) { xa[l] = ya[l] = 0; nx[l] = 1; nl[l] = 0; }
      nq = d.Lq - cell * 64 < 64 ? d.Lq - cell * 64 : 64;
    }
    const __amdgpu_buffer_rsrc_t rsrc =
        uniform_rsrc(value + (int64_t(b) * d.S * d.M + m) * 32, uint32_t((int64_t(d.S) * d.M - m) * 128));
    // sampling_loc / attn_weight (or raw offsets / logits) and the output rows of this (batch, head)
    const __amdgpu_buffer_rsrc_t loc_rsrc =
        uniform_rsrc(loc + (int64_t(b) * d.Lq * d.M + m) * 32, uint32_t((int64_t(d.Lq) * d.M - m) * 128));
    const __amdgpu_buffer_rsrc_t attn_rsrc =
        uniform_rsrc(attn + (int64_t(b) * d.Lq * d.M + m) * 16, uint32_t((int64_t(d.Lq) * d.M - m) * 64));
    float* const out_head = out + (int64_t(b) * d.Lq * d.M + m) * 32;

    for (int q0 = 0; q0 < nq; q0 += kNq) {
      const int nqp = nq - q0 < kNq ? nq - q0 : kNq;
      __syncthreads();   // the previous pass is done with qtab / records / windows
      if (tid < nqp) {
        int i = q0 + tid, q;
        if (pyramid) {
          int l = 0;
#pragma unroll
          for (int k = 0; k < kL - 1; ++k)
            if (l == k && i >= nl[k]) { i -= nl[k]; l = k + 1; }
          int nxl = nx[0], xal = xa[0], yal = ya[0], Wl = Ws[0], stl = St[0];
#pragma unroll
          for (int k = 1; k < kL; ++k)
            if (l == k) { nxl = nx[k]; xal = xa[k]; yal = ya[k]; Wl = Ws[k]; stl = St[k]; }
          const int yy = sdiv(i, nxl), xx = i - yy * nxl;
          q = stl + (yal + yy) * Wl + xal + xx;
        } else {
          q = cell * 64 + i;
        }
        qtab[tid] = q;
        if constexpr (FUSED) {   // softmax statistics of this (query, head): 16 logits = 64 contiguous bytes
          const uint32_t lo = uint32_t(q) * uint32_t(d.M) * 64u;
          const float4_t a0 = gload(attn_rsrc, lo), a1 = gload(attn_rsrc, lo + 16), a2 = gload(attn_rsrc, lo + 32),
                         a3 = gload(attn_rsrc, lo + 48);
          float mx = fmaxf(fmaxf(fmaxf(a0.x, a0.y), fmaxf(a0.z, a0.w)), fmaxf(fmaxf(a1.x, a1.y), fmaxf(a1.z, a1.w)));
          mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(a2.x, a2.y), fmaxf(a2.z, a2.w)), fmaxf(fmaxf(a3.x, a3.y), fmaxf(a3.z, a3.w))));
          float s = 0.f;
          s += expf(a0.x - mx); s += expf(a0.y - mx); s += expf(a0.z - mx); s += expf(a0.w - mx);
          s += expf(a1.x - mx); s += expf(a1.y - mx); s += expf(a1.z - mx); s += expf(a1.w - mx);
          s += expf(a2.x - mx); s += expf(a2.y - mx); s += expf(a2.z - mx); s += expf(a2.w - mx);
          s += expf(a3.x - mx); s += expf(a3.y - mx); s += expf(a3.z - mx); s += expf(a3.w - mx);
          stat[tid] = float2_t{mx, s};
        }
      }
      __syncthreads();
      VNX_TSTAMP(1);

      // ---- decode all levels: thread j owns point k = j & 3 of query i = j >> 2 --------------------
      const int qi = tid >> 2;
      const bool live = qi < nqp;
      float sat[kL], sab[kL], slw[kL];   // attention x (1 - lh), attention x lh, lw
      int sh0[kL], sw0[kL];              // sh0 == kNone: the sample is outside the map (or the lane idle)
      constexpr int kNone = -0x40000000;
      {
        const int q = live ? qtab[qi] : 0;
        const uint32_t srow = (uint32_t(q) * uint32_t(d.M)) * 16u + uint32_t(tid & 3);   // sample index within (b, head)
        float2_t st2 = {0.f, 1.f};
        if constexpr (FUSED) { if (live) st2 = stat[qi]; }
        float2_t xy[kL];
        float aw[kL];
        float4_t ref[kL];
#pragma unroll
        for (int l = 0; l < kL; ++l) {
          const uint32_t wi = live ? srow + l * kP : kTapOutside / 8;     // idle lanes: out of range -> zeros
          xy[l] = gload2(loc_rsrc, wi * 8u);
          aw[l] = gload1(attn_rsrc, wi * 4u);
          ref[l] = float4_t{0.f, 0.f, 0.f, 0.f};
          if constexpr (FUSED) {
            if (live) {
              const float* rf = static_cast<const float*>(fa.reference) +
                                ((int64_t(b / fa.ref_div) * d.Lq + q) * kL + l) * fa.ref_dim;
              if (fa.ref_dim == 2) {
                const float2_t r2 = *reinterpret_cast<const float2_t*>(rf);
                ref[l].x = r2.x; ref[l].y = r2.y;
              } else {
                ref[l] = *reinterpret_cast<const float4_t*>(rf);
              }
            }
          }
        }
#pragma unroll
        for (int l = 0; l < kL; ++l) {
          const float Hf = float(Hs[l]), Wf = float(Ws[l]);
          float x = xy[l].x, y = xy[l].y, a = aw[l];
          if constexpr (FUSED) {
            a = expf(a - st2.x) / st2.y;
            if (fa.ref_dim == 2) {
              x = ref[l].x + x / Wf;
              y = ref[l].y + y / Hf;
            } else {
              x = ref[l].x + x / float(kP) * ref[l].z * 0.5f;
              y = ref[l].y + y / float(kP) * ref[l].w * 0.5f;
            }
          }
          const float h = y * Hf - 0.5f, w = x * Wf - 0.5f;
          const bool in = live && h > -1.f && w > -1.f && h < Hf && w < Wf;   // cuh:288
          const float hf = floorf(h), wf = floorf(w);
          sh0[l] = in ? int(hf) : kNone; sw0[l] = in ? int(wf) : 0;
          const float lh = h - hf;
          sat[l] = a * (1.f - lh); sab[l] = a * lh; slw[l] = w - wf;
          // tap bounding box of the wave: x0, y0, -x1, -y1 as four independent minima
          const int big = 0x7fffffff;
          const int mx0 = wave_min(in ? sw0[l] : big), my0 = wave_min(in ? sh0[l] : big);
          const int mx1 = wave_min(in ? -(sw0[l] + 1) : big), my1 = wave_min(in ? -(sh0[l] + 1) : big);
          if (lane == 63) {
            box[(l * kWaves + wave) * 2] = pk16(mx0, my0);
            box[(l * kWaves + wave) * 2 + 1] = pk16(mx1 == big ? -32768 : -mx1, my1 == big ? -32768 : -my1);
          }
        }
      }
      VNX_TSTAMP(2);
      __syncthreads();
      VNX_TSTAMP(3);

      // ---- the four windows: lane l works out level l, then everyone reads them back -------------
      Window wn[kL];
      {
        uint32_t mn = box[my_l * kWaves * 2], mx = box[my_l * kWaves * 2 + 1];
#pragma unroll
        for (int w2 = 1; w2 < kWaves; ++w2) {
          mn = pk_min(mn, box[(my_l * kWaves + w2) * 2]); mx = pk_max(mx, box[(my_l * kWaves + w2) * 2 + 1]);
        }
        const int cap = (my_l & 1) ? kRowsB : kRowsA;
        const int x0 = pk_x(mn), y0 = pk_y(mn), x1 = pk_x(mx), y1 = pk_y(mx);
        int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
        if (bw <= 0 || bh <= 0) { bw = 0; bh = 0; }
        int wx0 = x0, wy0 = y0, ww = bw, wh = bh;
        if (bw * bh > cap) {   // a box of the same aspect around the centre
          const float sc = sqrtf(float(cap) / (float(bw) * float(bh)));
          int w2 = int(float(bw) * sc + 0.5f);
          w2 = w2 < 2 ? 2 : (w2 > bw ? bw : w2);
          if (w2 > cap / 2) w2 = cap / 2;
          int h2 = sdiv(cap, w2);
          if (h2 > bh) { h2 = bh; const int w3 = sdiv(cap, h2); w2 = w3 < bw ? w3 : bw; }
          wx0 = x0 + (bw - w2) / 2; wy0 = y0 + (bh - h2) / 2; ww = w2; wh = h2;
        }
#pragma unroll
        for (int l = 0; l < kL; ++l) {
          wn[l].x0 = lane_value(wx0, l); wn[l].y0 = lane_value(wy0, l);
          wn[l].w = lane_value(ww, l); wn[l].h = lane_value(wh, l);
        }
      }

      // stage level l into its buffer and leave its records; nothing is waited for here
      auto stage = [&](int l) {
        const Window w = wn[l];
        const int n_rows = w.w * w.h;
        const int Hl = Hs[l], Wl = Ws[l], stl = St[l];
        const int base_off = (l & 1) ? kOffWinB : kOffWinA;
        const uint32_t inv = w.w > 0 ? (65536u + uint32_t(w.w) - 1u) / uint32_t(w.w) : 0u;   // exact for r < n_rows <= 240
        for (int r0 = 0; r0 < n_rows; r0 += kSets) {
          const int r = r0 + (tid >> 3);
          const int yy = int((uint32_t(r) * inv) >> 16), xx = r - yy * w.w;
          const int py = w.y0 + yy, px = w.x0 + xx;
          const bool ok = r < n_rows && py >= 0 && py < Hl && px >= 0 && px < Wl;
          const uint32_t off = ok ? uint32_t(stl + py * Wl + px) * pixel_bytes + uint32_t(ch) * 16u : kTapOutside;
          unsigned char* dst = smem + base_off + (r0 + wave * 8) * 128;   // + lane * 16 by the hardware
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
        }
        if (live) {
          // one record per side: {offset of the top tap, of the bottom tap, their weights}; taps of a sample
          // outside the map read the zero row; taps outside the window: {kFar | x + 1, h0 + 1, weights}
          uint4_t rl = {uint32_t(kOffZero), uint32_t(kOffZero), 0u, 0u}, rr = rl;
          if (sh0[l] != kNone) {
            const int rx = sw0[l] - w.x0, ry = sh0[l] - w.y0;
            const bool inside = rx >= 0 && ry >= 0 && rx + 1 < w.w && ry + 1 < w.h;
            const uint32_t top = uint32_t(base_off + (ry * w.w + rx) * 128), bot = top + uint32_t(w.w) * 128u;
            const float lw = slw[l], hw = 1.f - lw;
            rl.x = inside ? top : (kFar | uint32_t(sw0[l] + 1));
            rl.y = inside ? bot : uint32_t(sh0[l] + 1);
            rr.x = inside ? top + 128u : (kFar | uint32_t(sw0[l] + 2));
            rr.y = inside ? bot + 128u : uint32_t(sh0[l] + 1);
            rl.z = __float_as_uint(sat[l] * hw); rl.w = __float_as_uint(sab[l] * hw);
            rr.z = __float_as_uint(sat[l] * lw); rr.w = __float_as_uint(sab[l] * lw);
          }
          recs[((l & 1) * 2 + 0) * kRecStride + tid] = rl;
          recs[((l & 1) * 2 + 1) * kRecStride + tid] = rr;
        }
      };

      float4_t acc[kGroups];
#pragma unroll
      for (int g = 0; g < kGroups; ++g) acc[g] = float4_t{0.f, 0.f, 0.f, 0.f};

      auto gather = [&](int l) {
        const int Hl = Hs[l], Wl = Ws[l], stl = St[l];
        const uint4_t* rs = recs + ((l & 1) * 2 + side) * kRecStride;
        const uint32_t lane_off = uint32_t(ch) * 16u;
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
          if ((g * kWaves + wave) * 4 < nqp) {               // wave-uniform
            const int i = (g * kWaves + wave) * 4 + pair;    // this pair's query (a pair past the end re-reads the
            const int ic = i < nqp ? i : nqp - 1;            // last query; its sums are never stored)
            uint4_t rc[kP];
#pragma unroll
            for (int k = 0; k < kP; ++k) rc[k] = rs[ic * kP + k];
            const uint32_t any = rc[0].x | rc[1].x | rc[2].x | rc[3].x;
            if (__builtin_expect(__ballot((any & kFar) != 0u) == 0ull, 1)) {
              // every tap of these four samples is in the window: eight row reads in flight, then the FMAs
              float4_t vt[kP], vb[kP];
#pragma unroll
              for (int k = 0; k < kP; ++k) { vt[k] = lds4(rc[k].x + lane_off); vb[k] = lds4(rc[k].y + lane_off); }
#pragma unroll
              for (int k = 0; k < kP; ++k) {
                acc[g] += __uint_as_float(rc[k].z) * vt[k];
                acc[g] += __uint_as_float(rc[k].w) * vb[k];
              }
            } else {
              // some taps are outside the window: those lanes fetch their rows from global memory, the
              // others from LDS; all sixteen reads are issued before the first use
              float4_t vt[kP], vb[kP];
#pragma unroll
              for (int k = 0; k < kP; ++k) {
                const uint4_t rec = rc[k];
                if (rec.x & kFar) {
                  const int x = int(rec.x & 0x7fffffffu) - 1, h0 = int(rec.y) - 1;
                  const bool okx = x >= 0 && x <= Wl - 1;
                  const uint32_t o0 = uint32_t(stl + h0 * Wl + x) * pixel_bytes + lane_off;
                  vt[k] = gload(rsrc, (okx && h0 >= 0) ? o0 : kTapOutside);
                  vb[k] = gload(rsrc, (okx && h0 + 1 <= Hl - 1) ? o0 + uint32_t(Wl) * pixel_bytes : kTapOutside);
                } else {
                  vt[k] = lds4(rec.x + lane_off); vb[k] = lds4(rec.y + lane_off);
                }
              }
#pragma unroll
              for (int k = 0; k < kP; ++k) {
                acc[g] += __uint_as_float(rc[k].z) * vt[k];
                acc[g] += __uint_as_float(rc[k].w) * vb[k];
              }
            }
          }
        }
      };

      // ---- pipeline: level l + 1 lands in the other buffer while level l is gathered -------------
      VNX_TSTAMP(4);
      stage(0);
      stage(1);
      VNX_TSTAMP(5);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      VNX_TSTAMP(6);
      gather(0);
      VNX_TSTAMP(7);
      __syncthreads();            // buffer A and its records are free
      stage(2);
      VNX_TSTAMP(8);
      gather(1);
      VNX_TSTAMP(9);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();            // level 2 landed; buffer B free
      stage(3);
      VNX_TSTAMP(10);
      gather(2);
      VNX_TSTAMP(11);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      VNX_TSTAMP(12);
      gather(3);
      VNX_TSTAMP(13);

      // ---- left + right halves of each pair, then one 128-B row per (query, head) ---------------
#pragma unroll
      for (int g = 0; g < kGroups; ++g) {
        if ((g * kWaves + wave) * 4 < nqp) {
          float4_t o = acc[g];
          o.x += __shfl_xor(o.x, 24, 64); o.y += __shfl_xor(o.y, 24, 64);
          o.z += __shfl_xor(o.z, 24, 64); o.w += __shfl_xor(o.w, 24, 64);
          const int i = (g * kWaves + wave) * 4 + pair;
          if (side == 0 && i < nqp) {
            const int q = qtab[i];
            __builtin_nontemporal_store(o, reinterpret_cast<float4_t*>(out_head + (uint32_t(q) * uint32_t(d.M)) * 32u + ch * 4));
          }
        }
      }
      VNX_TSTAMP(14);
    }
  }
}

}  // namespace tile

bool msda_tile_fwd_supported(int vdt, int ldt, const MsdaDims& d) {
  if (vdt != VNX_F32 || ldt != VNX_F32) return false;
  if (d.D != 32 || d.L != tile::kL || d.P != tile::kP) return false;
  if (d.S > 32766) return false;      // pixel coordinates travel in 15 / 16-bit fields: no level side can exceed S
  return int64_t(d.S) * d.M * 128 < (int64_t(1) << 31) && int64_t(d.Lq) * d.M * 128 < (int64_t(1) << 31);
}

static int g_num_cu = 0;
static int num_cu() {
  if (g_num_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    g_num_cu = n;
  }
  return g_num_cu;
}

// fa == nullptr: sampling_loc / attn_weight; else the fused prologue (raw offsets / logits + reference points)
int msda_forward_tile(const void* value, const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                      void* out, MsdaDims d, const FusedArgs* fa, int debug, hipStream_t stream) {
  // Work items are (batch, cell); the cell grid lives on the device (spatial_shapes is device memory), so
  // the grid is sized from an estimate -- S / 85 cells on a dyadic pyramid, Lq / 64 linear blocks otherwise
  // -- and every workgroup strides over the real item list.  Sized so that the resident workgroups
  // (2 per CU) each take the same number of items when the estimate is right.
  const int64_t est_cells = d.Lq == d.S ? (int64_t(d.S) + 84) / 85 : (int64_t(d.Lq) + 63) / 64;
  const int64_t est_items = int64_t(d.B) * (est_cells > 0 ? est_cells : 1);
  const int64_t cap = int64_t(num_cu()) * 2 / d.M > 0 ? int64_t(num_cu()) * 2 / d.M : 1;
  const int64_t rounds = (est_items + cap - 1) / cap;
  const int64_t per_head = (est_items + rounds - 1) / rounds;
  const dim3 grid(uint32_t(per_head * d.M));
  static bool lds_opt_in = false;     // > 64 KiB of dynamic LDS needs the attribute once per process
  if (!lds_opt_in) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&tile::msda_fwd_tile_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, tile::kLdsBytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&tile::msda_fwd_tile_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, tile::kLdsBytes) != hipSuccess) {
      set_error("msda_fwd_tile: cannot opt in to %d bytes of LDS: %s", tile::kLdsBytes, hipGetErrorString(hipGetLastError()));
      return VNX_ERR_LAUNCH;
    }
    lds_opt_in = true;
  }
  if (fa)
    hipLaunchKernelGGL((tile::msda_fwd_tile_kernel<true>), grid, dim3(tile::kThreads), tile::kLdsBytes, stream,
                       (const float*)value, shapes, lsi, (const float*)loc, (const float*)attn, (float*)out, d, *fa, debug);
  else
    hipLaunchKernelGGL((tile::msda_fwd_tile_kernel<false>), grid, dim3(tile::kThreads), tile::kLdsBytes, stream,
                       (const float*)value, shapes, lsi, (const float*)loc, (const float*)attn, (float*)out, d, FusedArgs{}, debug);
  return check_launch("msda_fwd_tile");
}

// development aid, not part of the public header
extern "C" int vnx_debug_read_tile_stamps(unsigned long long* host, int n) {
  return int(hipMemcpyFromSymbol(host, HIP_SYMBOL(tile::g_tile_stamps), sizeof(unsigned long long) * size_t(n)));
}

}  // namespace vnx
