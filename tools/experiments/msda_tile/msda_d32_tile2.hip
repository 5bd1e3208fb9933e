// msda_d32_tile2.hip -- multi-scale deformable attention forward for the ENCODER shape (queries = pixels of
// the pyramid, Lq == S), second LDS-staged form: larger cells, one sample per 8-lane set, next item prefetched.
//
// The first form (msda_d32_tile.hip) proved the data path -- windows of `value` streamed into LDS with
// `buffer_load ... lds`, taps read with ds_read_b128, zero LDS bank conflicts -- and lost to the L2 gather kernel
// anyway: 5.8 wave-instructions per sample for 1.0 of packed FMA, two thirds of them bookkeeping that an 85-query cell
// cannot amortise (PMC: VALU 70 % busy, LDS 27 %).  This form spends its instructions differently:
//   * a cell is 16 x 8 pixels of level 0 (170 queries on a dyadic pyramid) handled by ONE workgroup of 1 024 threads
//     per CU with 138 KB of LDS: per-item set-up, window search and barriers are paid once per 2 720 samples instead of
//     once per 1 360, and a window of 512 rows holds the taps of all eight head directions;
//   * thread t decodes the (query, point) pair t for all four levels (as before), and the SAME pair is gathered by one
//     8-lane set: ONE 16-byte record per sample {LDS address of the top-left tap, a (1 - lh), a lh, lw}; the set reads
//     it once, forms the four weights with three packed multiplies and reads the four taps from top-left + {0, 128,
//     pitch, pitch + 128} bytes -- 13 vector instructions and 5 LDS reads per sample (before: two records per sample,
//     two sets per sample);
//   * the four points of a query sit in four adjacent sets: their sums meet through one DPP rotate and one swizzle;
//   * the locations / weights of the NEXT item are requested while this item is gathered (one workgroup per CU has
//     nobody else to hide that round trip behind).
// Taps outside the window ("far") are fetched from global memory by the same lanes, samples outside the map read a
// zero region: exact for any input, the window only decides speed.  Unpacked levels or Lq != S run on linear blocks
// of queries.  fp32, 32-channel heads, 4 levels x 4 points.  Reference semantics: ms_deform_im2col_cuda.cuh:33-84,
// 237-299; why an encoder call is local: ops/modules/ms_deform_attn.py:65-73, deformable_transformer.py:183-196.
#include "vnx_common.h"

#include <type_traits>

namespace vnx {

namespace tile2 {

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));
typedef short short2_t __attribute__((ext_vector_type(2)));

constexpr int kWaves = 16, kThreads = 64 * kWaves;
constexpr int kL = 4, kP = 4;
constexpr int kCellW = 16, kCellH = 8;         // cell, in pixels of level 0
constexpr int kNq = 192;                       // queries per pass (a dyadic cell has 170)
constexpr int kPairs = kNq * kP;               // (query, point) pairs per pass: 768, one decoding thread each
constexpr int kSets = kThreads / 8;            // 128 eight-lane sets
constexpr int kRounds = kPairs / kSets;        // pairs per set and pass: 6
static_assert(kPairs % kSets == 0 && kPairs <= kThreads, "one thread per pair, whole rounds of sets");
constexpr int kRowsA = 512, kRowsB = 288;      // window buffers: levels 0, 2 -> A; 1, 3 -> B
constexpr int kMaxPitch = 32;                  // window width in pixels (the zero region spans one pitch)
constexpr uint32_t kFar = 0x80000000u;
constexpr uint32_t kTapOutside = 0x80000000u;

// LDS layout (bytes)
constexpr int kOffZero = 0;                                        // zeros: [0, kMaxPitch * 128 + 256)
constexpr int kOffWinA = kMaxPitch * 128 + 256;                    // also: the tap boxes of the decode phase
constexpr int kOffWinB = kOffWinA + kRowsA * 128;
constexpr int kOffRec = kOffWinB + kRowsB * 128;                   // [kL][kPairs] x 16 B: every level's records
constexpr int kOffQtab = kOffRec + kL * kPairs * 16;               // [2][kNq] int: global query index
constexpr int kOffMeta = kOffQtab + 2 * kNq * 4;                   // [2][kL + 1][8] int: cell geometry (see cell_meta)
constexpr int kOffWin = kOffMeta + 2 * (kL + 1) * 8 * 4;           // [kL][4] int: the windows
constexpr int kLdsBytes = kOffWin + kL * 4 * 4;
static_assert(kLdsBytes <= 160 * 1024, "one workgroup per CU");
static_assert(kL * kThreads * 8 <= kRowsA * 128, "the tap boxes fit window buffer A");
static_assert(kRowsA % 8 == 0 && kRowsB % 8 == 0, "a DMA step of 8 rows never leaves its buffer");

__device__ __forceinline__ int sdiv(int a, int b) {       // a / b for 0 <= a < 2^24, 0 < b < 2^24
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

__device__ __forceinline__ uint32_t pk16(int x, int y) {      // saturating
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
// packed signed 16-bit minimum / maximum over the wave, valid in lane 63 (lanes without a DPP source read the identity)
__device__ __forceinline__ uint32_t wave_pk_min(uint32_t v) {
#define VNX_STEP(ctrl, rmask) v = pk_min(v, uint32_t(__builtin_amdgcn_update_dpp(0x7fff7fff, int(v), ctrl, rmask, 0xF, false)));
  VNX_STEP(0x111, 0xF) VNX_STEP(0x112, 0xF) VNX_STEP(0x114, 0xF) VNX_STEP(0x118, 0xF) VNX_STEP(0x142, 0xA) VNX_STEP(0x143, 0xC)
#undef VNX_STEP
  return v;
}
__device__ __forceinline__ uint32_t wave_pk_max(uint32_t v) {
#define VNX_STEP(ctrl, rmask) v = pk_max(v, uint32_t(__builtin_amdgcn_update_dpp(int(0x80008000u), int(v), ctrl, rmask, 0xF, false)));
  VNX_STEP(0x111, 0xF) VNX_STEP(0x112, 0xF) VNX_STEP(0x114, 0xF) VNX_STEP(0x118, 0xF) VNX_STEP(0x142, 0xA) VNX_STEP(0x143, 0xC)
#undef VNX_STEP
  return v;
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

// the value of the lane 8 places away inside its row of 16 (the partner 8-lane set): DPP row_ror:8
__device__ __forceinline__ float ror8(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));
}

struct Window { int x0, y0, w, h; };   // pixels [x0, x0 + w) x [y0, y0 + h); may include column / row -1 and W / H

__global__ void __launch_bounds__(kThreads)
msda_fwd_tile2_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                      const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                      const float* __restrict__ attn, float* __restrict__ out, MsdaDims d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* const qtab = reinterpret_cast<int*>(smem + kOffQtab);
  int* const meta = reinterpret_cast<int*>(smem + kOffMeta);
  int* const win = reinterpret_cast<int*>(smem + kOffWin);
  uint2_t* const boxes = reinterpret_cast<uint2_t*>(smem + kOffWinA);     // [kL][kThreads] {packed min, packed max}
  uint4_t* const recs = reinterpret_cast<uint4_t*>(smem + kOffRec);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = blockIdx.x % d.M;                  // fixed head <-> XCD map (as the first tiled form)
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
  const int CX = pyramid ? sdiv_u(Ws[0] + kCellW - 1, kCellW) : 1, CY = pyramid ? sdiv_u(Hs[0] + kCellH - 1, kCellH) : 1;
  const int cells = pyramid ? CX * CY : sdiv_u(d.Lq + kNq - 1, kNq);
  const int n_items = d.B * cells;

  for (int i = tid; i < kOffWinA / 16; i += kThreads) reinterpret_cast<float4_t*>(smem)[i] = float4_t{0.f, 0.f, 0.f, 0.f};

  const uint32_t pixel_bytes = uint32_t(d.M) * 128u;
  const int ch = lane & 7;
  const uint32_t lane_off = uint32_t(ch) * 16u;
  const int gset = wave * 8 + (lane >> 3);         // this lane's 8-lane set among the workgroup's 128

  // Per-item work that is uniform over the workgroup is done by ONE wave and left in LDS (every wave doing it for itself
  // was a third of the kernel's vector instructions: PMC, first version of this file).
  // cell_meta (wave 0 only; lanes 0..3 = levels): the cell of `item` -> meta[buf]:
  //   [l][0..5] = {first column, first row, columns, queries of the cell before this level, level width, level start}
  //   [kL][0..4] = {queries of the cell, batch element, queries of level 0, of levels 0..1, of levels 0..2}
  auto cell_meta = [&](int item, int buf) __attribute__((always_inline)) {
    const int b = sdiv_u(item, cells), cell = item - b * cells;
    int* mt = meta + buf * (kL + 1) * 8;
    if (pyramid) {
      // this lane's level: geometry straight from memory (a select chain over Hs[] / Ws[] / St[] is turned into an indexed
      // load of a scratch copy of the arrays by this compiler)
      const int my_l = lane < kL ? lane : kL - 1;
      const int myH = int(shapes[2 * my_l]), myW = int(shapes[2 * my_l + 1]), mySt = int(lsi[my_l]);
      const int cy = sdiv_u(cell, CX), cx = cell - cy * CX;
      const int vxa = cell_lo(cx, CX, myW), vya = cell_lo(cy, CY, myH);
      const int vnx = cell_lo(cx + 1, CX, myW) - vxa;
      const int vnl = vnx * (cell_lo(cy + 1, CY, myH) - vya);
      const int n0 = lane_value(vnl, 0), n1 = lane_value(vnl, 1), n2 = lane_value(vnl, 2), n3 = lane_value(vnl, 3);
      const int pre = (my_l > 0 ? n0 : 0) + (my_l > 1 ? n1 : 0) + (my_l > 2 ? n2 : 0);
      if (lane < kL) {
        mt[lane * 8 + 0] = vxa; mt[lane * 8 + 1] = vya; mt[lane * 8 + 2] = vnx; mt[lane * 8 + 3] = pre;
        mt[lane * 8 + 4] = myW; mt[lane * 8 + 5] = mySt;
      }
      if (lane == 0) {
        mt[kL * 8 + 0] = n0 + n1 + n2 + n3; mt[kL * 8 + 1] = b;
        mt[kL * 8 + 2] = n0; mt[kL * 8 + 3] = n0 + n1; mt[kL * 8 + 4] = n0 + n1 + n2;
      }
    } else if (lane == 0) {
      mt[kL * 8 + 0] = d.Lq - cell * kNq < kNq ? d.Lq - cell * kNq : kNq;
      mt[kL * 8 + 1] = b;
    }
  };
  // queries [q0, q0 + kNq) of the cell described by meta[buf] -> qtab[buf] (every thread)
  auto fill_qtab = [&](int item, int q0, int nqp, int buf) __attribute__((always_inline)) {
    if (tid < nqp) {
      const int* mt = meta + buf * (kL + 1) * 8;
      int i = q0 + tid, q;
      if (pyramid) {
        const int l = int(i >= mt[kL * 8 + 2]) + int(i >= mt[kL * 8 + 3]) + int(i >= mt[kL * 8 + 4]);
        const int4 g = *reinterpret_cast<const int4*>(mt + l * 8);       // {xa, ya, nx, pre}
        const int2 ws = *reinterpret_cast<const int2*>(mt + l * 8 + 4);   // {W, start}
        i -= g.w;
        const int yy = sdiv(i, g.z), xx = i - yy * g.z;
        q = ws.y + (g.y + yy) * ws.x + g.x + xx;
      } else {
        const int b = mt[kL * 8 + 1];
        q = (item - b * cells) * kNq + i;
      }
      qtab[buf * kNq + tid] = q;
    }
  };
  // raw locations / weights of pair t (all levels) of the pass whose queries are in qtab[buf]
  float2_t xy[kL];
  float aw[kL];
  auto load_raw = [&](int b, int nqp, int buf) __attribute__((always_inline)) {
    const bool live = tid < nqp * kP;
    const int q = live ? qtab[buf * kNq + (tid >> 2)] : 0;
    const __amdgpu_buffer_rsrc_t loc_rsrc =
        uniform_rsrc(loc + (int64_t(b) * d.Lq * d.M + m) * 32, uint32_t((int64_t(d.Lq) * d.M - m) * 128));
    const __amdgpu_buffer_rsrc_t attn_rsrc =
        uniform_rsrc(attn + (int64_t(b) * d.Lq * d.M + m) * 16, uint32_t((int64_t(d.Lq) * d.M - m) * 64));
    const uint32_t srow = (uint32_t(q) * uint32_t(d.M)) * 16u + uint32_t(tid & 3);   // sample index within (b, head)
#pragma unroll
    for (int l = 0; l < kL; ++l) {
      const uint32_t wi = live ? srow + l * kP : kTapOutside / 8;     // idle lanes: out of range -> zeros
      xy[l] = gload2(loc_rsrc, wi * 8u);
      aw[l] = gload1(attn_rsrc, wi * 4u);
    }
  };

  int qbuf = 0;
  bool have_raw = false;          // meta[qbuf], qtab[qbuf] and xy / aw already describe the upcoming item (prefetched)
  for (int item = wg; item < n_items; item += wgs) {
    if (!have_raw) {
      __syncthreads();            // the previous pass is done with meta / qtab (and with everything else)
      if (wave == 0) cell_meta(item, qbuf);
      __syncthreads();
    }
    const int nq = __builtin_amdgcn_readfirstlane(meta[qbuf * (kL + 1) * 8 + kL * 8 + 0]);
    const int b = __builtin_amdgcn_readfirstlane(meta[qbuf * (kL + 1) * 8 + kL * 8 + 1]);
    const __amdgpu_buffer_rsrc_t rsrc =
        uniform_rsrc(value + (int64_t(b) * d.S * d.M + m) * 32, uint32_t((int64_t(d.S) * d.M - m) * 128));
    float* const out_head = out + (int64_t(b) * d.Lq * d.M + m) * 32;

    for (int q0 = 0; q0 < nq; q0 += kNq) {
      const int nqp = nq - q0 < kNq ? nq - q0 : kNq;
      const int npairs = nqp * kP;
      if (!have_raw) {
        if (q0 > 0) __syncthreads();      // the previous pass of this cell is done with qtab[qbuf]
        fill_qtab(item, q0, nqp, qbuf);
        __syncthreads();
        load_raw(b, nqp, qbuf);
      }
      have_raw = false;

      // ---- decode all levels: thread t owns pair t = 4 * query + point ------------------------------
      const bool live = tid < npairs;
      float sat[kL], sab[kL], slw[kL];   // attention x (1 - lh), attention x lh, lw
      int sh0[kL], sw0[kL];              // sh0 == kNone: the sample is outside the map (or the thread idle)
      constexpr int kNone = -0x40000000;
#pragma unroll
      for (int l = 0; l < kL; ++l) {
        const float Hf = float(Hs[l]), Wf = float(Ws[l]);
        const float x = xy[l].x, y = xy[l].y, a = aw[l];
        const float h = y * Hf - 0.5f, w = x * Wf - 0.5f;                       // cuh:285-286
        const bool in = live && h > -1.f && w > -1.f && h < Hf && w < Wf;       // cuh:288
        const float hf = floorf(h), wf = floorf(w);
        sh0[l] = in ? int(hf) : kNone; sw0[l] = in ? int(wf) : 0;
        const float lh = h - hf;
        sat[l] = a * (1.f - lh); sab[l] = a * lh; slw[l] = w - wf;
        // this sample's tap box, packed (x, y): top-left tap, bottom-right tap; reduced by the window waves below
        boxes[l * kThreads + tid] = uint2_t{in ? pk16(sw0[l], sh0[l]) : 0x7fff7fffu, in ? pk16(sw0[l] + 1, sh0[l] + 1) : 0x80008000u};
      }

      // ---- the item after this one: its cell (wave 0), then (below) its queries and raw samples -------------------
      const bool last_pass = q0 + kNq >= nq;
      const bool has_next = last_pass && item + wgs < n_items;
      if (has_next && wave == 0) cell_meta(item + wgs, qbuf ^ 1);
      __syncthreads();     // boxes and the next cell are visible; the previous gather is over

      // ---- windows: wave l reduces level l's boxes and picks the window; the others wait at the barrier ----------
      if (wave < kL) {
        uint32_t mn = 0x7fff7fffu, mx = 0x80008000u;
#pragma unroll
        for (int j = 0; j < kThreads / 64; ++j) {
          const uint2_t e = boxes[wave * kThreads + j * 64 + lane];
          mn = pk_min(mn, e.x); mx = pk_max(mx, e.y);
        }
        mn = wave_pk_min(mn); mx = wave_pk_max(mx);
        if (lane == 63) {
          const int cap = (wave & 1) ? kRowsB : kRowsA;
          const int x0 = pk_x(mn), y0 = pk_y(mn), x1 = pk_x(mx), y1 = pk_y(mx);
          int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
          if (bw <= 0 || bh <= 0) { bw = 0; bh = 0; }
          int wx0 = x0, wy0 = y0, ww = bw, wh = bh;
          if (ww > kMaxPitch) { wx0 = x0 + (bw - kMaxPitch) / 2; ww = kMaxPitch; }     // the middle columns
          if (ww > 0 && ww * wh > cap) {                                             // the middle rows
            const int h2 = sdiv(cap, ww);
            wy0 = y0 + (bh - h2) / 2; wh = h2;
          }
          win[wave * 4 + 0] = wx0; win[wave * 4 + 1] = wy0; win[wave * 4 + 2] = ww; win[wave * 4 + 3] = wh;
        }
      }
      int nb = 0, n_nqp = 0;
      bool prefetch = false;
      if (has_next) {
        const int* mt = meta + (qbuf ^ 1) * (kL + 1) * 8;
        n_nqp = __builtin_amdgcn_readfirstlane(mt[kL * 8 + 0]);
        nb = __builtin_amdgcn_readfirstlane(mt[kL * 8 + 1]);
        prefetch = n_nqp <= kNq;      // (a larger cell is walked in passes by the unprefetched path)
        if (prefetch) fill_qtab(item + wgs, 0, n_nqp, qbuf ^ 1);
      }
      __syncthreads();     // windows and the next item's queries are visible

      Window wn[kL];
#pragma unroll
      for (int l = 0; l < kL; ++l) {
        const int4 w4 = *reinterpret_cast<const int4*>(win + l * 4);
        wn[l].x0 = __builtin_amdgcn_readfirstlane(w4.x); wn[l].y0 = __builtin_amdgcn_readfirstlane(w4.y);
        wn[l].w = __builtin_amdgcn_readfirstlane(w4.z); wn[l].h = __builtin_amdgcn_readfirstlane(w4.w);
      }

      // stage level l into its buffer; nothing is waited for here
      auto stage = [&](int l) __attribute__((always_inline)) {
        const Window w = wn[l];
        const int n_rows = w.w * w.h;
        const int Hl = Hs[l], Wl = Ws[l], stl = St[l];
        const int base_off = (l & 1) ? kOffWinB : kOffWinA;
        // (r * inv) >> 16 == r / w for r < 640 and w <= 32: the error term r * (inv * w - 65536) stays below 65536
        const uint32_t inv = w.w > 0 ? (65536u + uint32_t(w.w) - 1u) / uint32_t(w.w) : 0u;
        for (int r0 = wave * 8; r0 < n_rows; r0 += kSets) {
          const int r = r0 + (lane >> 3);
          const int yy = int((uint32_t(r) * inv) >> 16), xx = r - yy * w.w;
          const int py = w.y0 + yy, px = w.x0 + xx;
          const bool ok = r < n_rows && py >= 0 && py < Hl && px >= 0 && px < Wl;
          const uint32_t off = ok ? uint32_t(stl + py * Wl + px) * pixel_bytes + lane_off : kTapOutside;
          unsigned char* dst = smem + base_off + r0 * 128;   // + lane * 16 by the hardware
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
        }
      };

      // the first two windows start to travel while the records are worked out
      stage(0);
      stage(1);

      // ---- records of all four levels: {LDS address of the top-left tap, a (1 - lh), a lh, lw}; a sample outside the
      //      map reads the zero region with zero weights; taps outside the window: {kFar | (h0 + 1) << 15 | (w0 + 1), ...}
      if (live) {
#pragma unroll
        for (int l = 0; l < kL; ++l) {
          const int base_off = (l & 1) ? kOffWinB : kOffWinA;
          uint4_t rc = {uint32_t(kOffZero), 0u, 0u, 0u};
          if (sh0[l] != kNone) {
            const int rx = sw0[l] - wn[l].x0, ry = sh0[l] - wn[l].y0;
            const bool inside = rx >= 0 && ry >= 0 && rx + 1 < wn[l].w && ry + 1 < wn[l].h;
            rc.x = inside ? uint32_t(base_off + (ry * wn[l].w + rx) * 128) : (kFar | (uint32_t(sh0[l] + 1) << 15) | uint32_t(sw0[l] + 1));
            rc.y = __float_as_uint(sat[l]); rc.z = __float_as_uint(sab[l]); rc.w = __float_as_uint(slw[l]);
          }
          recs[l * kPairs + tid] = rc;
        }
      }

      float4_t acc[kRounds];
#pragma unroll
      for (int r = 0; r < kRounds; ++r) acc[r] = float4_t{0.f, 0.f, 0.f, 0.f};

      // rounds [R0, R0 + N) of level l for this set: the records first, then -- when no sample of the wave is far --
      // 4 N tap reads in flight and the FMAs, straight-line (the far path keeps its global loads, and the vmcnt waits
      // they need, to itself: merged with the LDS path they made every FMA wait for the window DMA in flight)
      auto gather_block = [&](int l, auto R0c, auto Nc) __attribute__((always_inline)) {
        constexpr int R0 = decltype(R0c)::value, N = decltype(Nc)::value;
        if (R0 * kSets + wave * 8 >= npairs) return;               // wave-uniform
        const int Hl = Hs[l], Wl = Ws[l], stl = St[l];
        const uint4_t* rs = recs + l * kPairs;
        const uint32_t pitch_bytes = uint32_t(wn[l].w) * 128u;
        uint4_t rc[N];
        uint32_t any = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const int p = (R0 + j) * kSets + gset;       // a set past the end re-reads the last pair; its sum is never stored
          rc[j] = rs[p < npairs ? p : npairs - 1];
          any |= rc[j].x;
        }
        if (__builtin_expect(__ballot((any & kFar) != 0u) == 0ull, 1)) {
          float4_t v[N][4];
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const uint32_t a0 = rc[j].x + lane_off, a1 = a0 + pitch_bytes;
            v[j][0] = lds4(a0); v[j][1] = lds4(a0 + 128u); v[j][2] = lds4(a1); v[j][3] = lds4(a1 + 128u);
          }
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const float at = __uint_as_float(rc[j].y), ab = __uint_as_float(rc[j].z), lw = __uint_as_float(rc[j].w), hw = 1.f - lw;
            acc[R0 + j] += (at * hw) * v[j][0];
            acc[R0 + j] += (at * lw) * v[j][1];
            acc[R0 + j] += (ab * hw) * v[j][2];
            acc[R0 + j] += (ab * lw) * v[j][3];
          }
        } else {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const float at = __uint_as_float(rc[j].y), ab = __uint_as_float(rc[j].z), lw = __uint_as_float(rc[j].w), hw = 1.f - lw;
            float4_t v00, v01, v10, v11;
            if (rc[j].x & kFar) {          // this sample's taps come from global memory
              const int w0 = int(rc[j].x & 0x7fffu) - 1, h0 = int((rc[j].x >> 15) & 0xffffu) - 1;
              const bool lef = w0 >= 0, rig = w0 + 1 <= Wl - 1, top = h0 >= 0, bot = h0 + 1 <= Hl - 1;
              const uint32_t o00 = uint32_t(stl + h0 * Wl + w0) * pixel_bytes + lane_off;
              v00 = gload(rsrc, (top && lef) ? o00 : kTapOutside);
              v01 = gload(rsrc, (top && rig) ? o00 + pixel_bytes : kTapOutside);
              v10 = gload(rsrc, (bot && lef) ? o00 + uint32_t(Wl) * pixel_bytes : kTapOutside);
              v11 = gload(rsrc, (bot && rig) ? o00 + uint32_t(Wl + 1) * pixel_bytes : kTapOutside);
            } else {
              const uint32_t a0 = rc[j].x + lane_off, a1 = a0 + pitch_bytes;
              v00 = lds4(a0); v01 = lds4(a0 + 128u); v10 = lds4(a1); v11 = lds4(a1 + 128u);
            }
            acc[R0 + j] += (at * hw) * v00;
            acc[R0 + j] += (at * lw) * v01;
            acc[R0 + j] += (ab * hw) * v10;
            acc[R0 + j] += (ab * lw) * v11;
          }
        }
      };
      auto gather = [&](int l) __attribute__((always_inline)) {
        using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>;
        using I4 = std::integral_constant<int, 4>;
        static_assert(kRounds == 6, "three blocks of two rounds (three rounds per block: 91 spilled VGPRs)");
        gather_block(l, I0{}, I2{});
        gather_block(l, I2{}, I2{});
        gather_block(l, I4{}, I2{});
      };

      // ---- pipeline: level l + 1 lands in the other buffer while level l is gathered -------------
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (prefetch) { load_raw(nb, n_nqp, qbuf ^ 1); have_raw = true; }     // in flight during the gathers
      gather(0);
      __syncthreads();            // buffer A is free
      stage(2);
      gather(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();            // level 2 landed; buffer B free
      stage(3);
      gather(2);
      // the rows this lane will store: read the query indices BEFORE the last barrier -- after it a faster wave may
      // already be filling this qtab buffer with the queries of the item after the next (it is `qbuf ^ 1` by then)
      int qst[kRounds];
#pragma unroll
      for (int r = 0; r < kRounds; ++r) {
        const int p = r * kSets + gset;
        qst[r] = ((gset & 3) == 0 && p < npairs) ? qtab[qbuf * kNq + (p >> 2)] : -1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      gather(3);

      // ---- the four points of a query sit in four adjacent sets: sum them, one 128-B row per (query, head) ----
#pragma unroll
      for (int r = 0; r < kRounds; ++r) {
        if (r * kSets + wave * 8 < npairs) {
          // (scalars, not o.x ...: __builtin_bit_cast of a vector COMPONENT reads component 0 with this compiler)
          float ox = acc[r].x, oy = acc[r].y, oz = acc[r].z, ow = acc[r].w;
          ox += ror8(ox); oy += ror8(oy); oz += ror8(oz); ow += ror8(ow);
          float4_t o;
          o.x = ox + __shfl_xor(ox, 16, 64); o.y = oy + __shfl_xor(oy, 16, 64);
          o.z = oz + __shfl_xor(oz, 16, 64); o.w = ow + __shfl_xor(ow, 16, 64);
          if (qst[r] >= 0)
            __builtin_nontemporal_store(o, reinterpret_cast<float4_t*>(out_head + (uint32_t(qst[r]) * uint32_t(d.M)) * 32u + ch * 4));
        }
      }
      if (have_raw) qbuf ^= 1;
    }
  }
}

}  // namespace tile2

bool msda_tile2_fwd_supported(int vdt, int ldt, const MsdaDims& d) {
  if (vdt != VNX_F32 || ldt != VNX_F32) return false;
  if (d.D != 32 || d.L != tile2::kL || d.P != tile2::kP) return false;
  if (d.S > 32766) return false;      // pixel coordinates travel in 15 / 16-bit fields: no level side can exceed S
  return int64_t(d.S) * d.M * 128 < (int64_t(1) << 31) && int64_t(d.Lq) * d.M * 128 < (int64_t(1) << 31);
}

static int g_num_cu2 = 0;
static int num_cu2() {
  if (g_num_cu2 == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    g_num_cu2 = n;
  }
  return g_num_cu2;
}

int msda_forward_tile2(const void* value, const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                       void* out, MsdaDims d, hipStream_t stream) {
  // Work items are (batch, cell); the cell grid lives on the device, so the grid is sized from an estimate (S / 170
  // cells on a dyadic pyramid, Lq / 192 linear blocks otherwise) and every workgroup strides over the real item list:
  // one workgroup per CU, each taking the same number of items when the estimate is right.
  const int64_t est_cells = d.Lq == d.S ? (int64_t(d.S) + 169) / 170 : (int64_t(d.Lq) + tile2::kNq - 1) / tile2::kNq;
  const int64_t est_items = int64_t(d.B) * (est_cells > 0 ? est_cells : 1);
  const int64_t cap = int64_t(num_cu2()) / d.M > 0 ? int64_t(num_cu2()) / d.M : 1;
  const int64_t rounds = (est_items + cap - 1) / cap;
  const int64_t per_head = (est_items + rounds - 1) / rounds;
  const dim3 grid(uint32_t(per_head * d.M));
  static bool lds_opt_in = false;     // > 64 KiB of dynamic LDS needs the attribute once per process
  if (!lds_opt_in) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&tile2::msda_fwd_tile2_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, tile2::kLdsBytes) != hipSuccess) {
      set_error("msda_fwd_tile2: cannot opt in to %d bytes of LDS: %s", tile2::kLdsBytes, hipGetErrorString(hipGetLastError()));
      return VNX_ERR_LAUNCH;
    }
    lds_opt_in = true;
  }
  hipLaunchKernelGGL(tile2::msda_fwd_tile2_kernel, grid, dim3(tile2::kThreads), tile2::kLdsBytes, stream,
                     (const float*)value, shapes, lsi, (const float*)loc, (const float*)attn, (float*)out, d);
  return check_launch("msda_fwd_tile2");
}

}  // namespace vnx
