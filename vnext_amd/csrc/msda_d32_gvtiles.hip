// msda_d32_gvtiles.hip -- grad_value, owner-computes, for calls with MANY queries (the encoder's: the queries
// are the pixels of the pyramid, Lq = S), fed by per-tile summaries instead of per-sample records.
//
// The record-fed kernel (msda_d32_gvrec.hip) has the grad_loc kernel leave 16 B of geometry + 4 B of unit range for
// EVERY sample (65 MB written per T=5 360p encoder call, 250 MB at 720p) and every unit scan the 4-B words of ALL of
// its level's samples in selection windows of 2 048 -- ten windows of three barriers and a dependent load each before
// the first tap is applied.  At the decoder shape (300 scattered queries) that selection is what makes a unit cheap;
// at the encoder shape it is a third of the kernel, and the records are 40 % of the backward's memory traffic.
//
// Here selection works on TILES of queries: a wave of the grad_loc kernel handles T consecutive queries (T = 4
// on large calls; until late in round 3 the tile was the workgroup's 16) of one (batch, head), and consecutive queries of
// an encoder call are neighbouring pixels whose samples land next to each other.  That wave reduces, per level, the
// bounding box of the pixels its samples' corners touch and leaves TWO 4-byte words per (batch, head, level, tile): 1 275
// word pairs per level at 360p where
// there were 20 400 tags.  A unit of this kernel -- a rectangle of <= 256 pixels of one level: a band of whole image
// rows of a narrow level, a block of about 32 x 8 pixels of a wide one (gv_level_grid, vnx_common.h) --
//   1. reads its level's tile words (one load round per 512 tiles), keeps the tiles whose box meets its rectangle
//      (wave ballots + one 32-entry scan: ascending order);
//   2. walks the kept tiles 128 queries (8 tiles) per chunk: one sample per thread -- its location and weight read
//      from the op's own inputs, the geometry recomputed (15 VALU instructions; the predecessor of the record kernel
//      was slow because every unit recomputed ALL samples of its level, not because of these) -- the tiles' grad_out
//      rows staged in LDS (consecutive queries: the rows of a tile are one strided run), then the same counting sort
//      by destination row and register accumulation as the record-fed kernel;
//   3. writes its rows once -- or, for the pieces of a query-split level (gv_query_splits, split by tile range here), stores
//      them into the piece's slab of fp32 partial rows; gv_split_finish_kernel adds the pieces (no atomics, no zeroing:
//      gv_partial_rows_bound in vnx_common.h has the measurement that retired the atomics).
// No records, no tags, no windows.  Levels must be packed (checked on the device).  Reference semantics:
// ms_deform_im2col_cuda.cuh:87-159 (the scatter this replaces), :253-298 (index decode).
#include "msda_gv_common.h"

namespace vnx {
namespace rec {

#ifndef VNX_TILE_ROUNDS
#define VNX_TILE_ROUNDS 2
#endif
#ifndef VNX_TILE_UNITS_PER_CU
#define VNX_TILE_UNITS_PER_CU 4
#endif
// Timing ablations (A/B builds of the development library only; wrong grad_value by construction): 1 = no chunks at all
// (what set-up + tile selection + the final store cost), 2 = chunks without taps (staging, prefetch and decode only: no
// rank atomics, no scatter, nothing to apply), 3 = everything but the apply loop.
#ifndef VNX_GVT_ABL
#define VNX_GVT_ABL 0
#endif
// Workgroup order: 0 = the batch element minor (all B batch elements of a unit on neighbouring workgroups), 1 = major
// (gv_decode_block_bm).  Round 6, measured both ways (kbench cold + FETCH_SIZE): major fetches less -- 619 against 722 MB per 720p
// B = 5 launch, 68.5 against 95.3 at 360p (counter units as reported) -- and is SLOWER: encoder-360p backward 185.5 against
// 150.9 us, 720p B = 5 716.5 against 575.9, B = 2 308.8 against 257.5, bf16 165.8 against 130.8.  The kernel is bound by issue and
// LDS, not by what it fetches, and with one batch element resident per XCD its ~90 units all read the same tile words and the
// same stretch of decoded samples at the same time.  Minor stays.
#ifndef VNX_GVT_BATCH_MAJOR
#define VNX_GVT_BATCH_MAJOR 0
#endif
constexpr int kTileRowsMax = kGvTileRowsMax;      // 256 rows per unit: 4 per 8-lane group
constexpr int kTileRounds = VNX_TILE_ROUNDS;
constexpr int kTileWin = kTileRounds * kThreads;  // tile words examined per selection round: 1 024
constexpr int kTileParts = kTileRounds * kWaves;  // (round, wave) pieces per selection
constexpr int kTileLevels = 4;                    // the tile words exist for 4 levels x 4 points only (msda_d32.hip, tile mode)
constexpr size_t kTilesLdsBytes = size_t(kQcMax) * 128 + size_t(kThreads) * 32 + size_t(kTileRowsMax) * 12 + 16 +
                                  8 * kTileLevels * 4 + size_t(kTileWin) * 2 + size_t(kTileParts) * 8 + 16;
static_assert(kTilesLdsBytes * VNX_TILE_UNITS_PER_CU <= 160 * 1024, "the units that share a CU must fit its LDS");

template <typename TV, typename TL>
__global__ void __launch_bounds__(kThreads, VNX_TILE_UNITS_PER_CU * kWaves / 4)
msda_bwd_gv_tiles_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                         const TL* __restrict__ loc, const TL* __restrict__ attn,
                         const uint2_t* __restrict__ summaries, const TV* __restrict__ grad_out,
                         TV* __restrict__ grad_value, MsdaDims d, int units_min, int tile_shift, int n_tiles,
                         float* __restrict__ partials, int compact, int units_pb) {
  constexpr int D = 32, P = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4_t* grows = reinterpret_cast<float4_t*>(smem);                       // [128][8] grad_out rows
  uint2_t* list = reinterpret_cast<uint2_t*>(grows + kQcMax * 8);            // [4*threads] taps
  uint32_t* cnt2 = reinterpret_cast<uint32_t*>(list + 4 * kThreads);         // [2][rows]
  uint32_t* offs = cnt2 + 2 * kTileRowsMax;                                      // [rows]
  uint32_t* alloc = offs + kTileRowsMax;                                         // [4]
  int* meta = reinterpret_cast<int*>(alloc + 4);                             // [8*L]
  uint16_t* hit = reinterpret_cast<uint16_t*>(meta + 8 * kTileLevels);       // [kTileWin] window-relative tile
  uint32_t* part_s = reinterpret_cast<uint32_t*>(hit + kTileWin);            // [32] kept tiles per piece
  uint32_t* pre_s = part_s + kTileParts;                                     // [32] exclusive prefixes
  uint32_t* tot = pre_s + kTileParts;                                        // [1] kept tiles of the round

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // dispatch order = batch element, then cost order (units numbered from the last level back); head <-> XCD map rotating
  // with the batch element: gv_decode_block_bm (msda_gv_common.h)
  int unit, b, m;
#if VNX_GVT_BATCH_MAJOR
  if (kGvPair16 && sizeof(TV) == 2 && (d.M & 1) == 0) gv_decode_block_bm<true>(blockIdx.x, d.M, units_pb, unit, b, m);
  else gv_decode_block_bm<false>(blockIdx.x, d.M, units_pb, unit, b, m);
  if (b >= d.B) return;      // uniform: the padding workgroup of an odd grid
#else
  {
    int rest;
    if (kGvPair16 && sizeof(TV) == 2 && (d.M & 1) == 0) gv_decode_block<true>(blockIdx.x, d.M, d.B, rest, b, m);
    else gv_decode_block<false>(blockIdx.x, d.M, d.B, rest, b, m);
    unit = rest / d.B;
  }
#endif

  if (tid < d.L) {     // level table: {H, W, start, workgroups, query pieces, block width, blocks per row, block height}
    const int H = int(shapes[2 * tid]), W = int(shapes[2 * tid + 1]);
    const GvGrid g = gv_level_grid(H, W, units_min, kTileRowsMax);
    const int qs = gv_query_splits(g.nbx * g.nby, d.Lq, P, true, d.B * d.M);
    int* mt = meta + 8 * tid;
    mt[0] = H; mt[1] = W; mt[2] = int(lsi[tid]); mt[3] = g.nbx * g.nby * qs; mt[4] = qs; mt[5] = g.bw; mt[6] = g.nbx; mt[7] = g.bh;
  }
  for (int i = tid; i < 2 * kTileRowsMax; i += kThreads) cnt2[i] = 0;
  if (tid == 0) alloc[0] = 0;
  __syncthreads();

  // this workgroup's unit: pixels [x0, x1) x [y0, y1) of level lvl (local row = (y - y0) * pitch + x - x0, pitch = the
  // level's block width), possibly one of `qsplit` query pieces of it
  int lvl = -1, Hl = 0, Wl = 0, start = 0, qsplit = 1, qpiece = 0, x0 = 0, x1 = 0, y0 = 0, y1 = 0, pitch = 1;
  int pbase = 0;                               // first partial row of this level's pieces (gv_partial_rows_bound)
  {
    int running = 0;
    bool packed = true;
    int u = unit;
    for (int l = 0; l < d.L; ++l) {
      packed = packed && (meta[8 * l + 2] == running);
      running += meta[8 * l] * meta[8 * l + 1];
    }
    for (int l = d.L - 1; l >= 0; --l) {       // units are numbered from the last (coarsest) level back
      const int* mt = meta + 8 * l;
      const int total = mt[3];
      if (lvl < 0) {
        if (u < total) {             // a level's workgroups: unit major, query piece minor
          lvl = l; Hl = mt[0]; Wl = mt[1]; start = mt[2]; qsplit = mt[4]; pitch = mt[5];
          const int nbx = mt[6], bh = mt[7];
          const int u_lvl = u / qsplit;
          qpiece = u - u_lvl * qsplit;
          const int by = u_lvl / nbx, bx = u_lvl - by * nbx;
          x0 = bx * pitch; x1 = x0 + pitch < Wl ? x0 + pitch : Wl;
          y0 = by * bh; y1 = y0 + bh < Hl ? y0 + bh : Hl;
        } else {
          u -= total;
        }
      }
    }
    packed = packed && (running == d.S);
    if (!packed || lvl < 0) return;  // uniform over the workgroup
    for (int l = 0; l < lvl; ++l)
      if (meta[8 * l + 4] > 1) pbase += meta[8 * l + 4] * meta[8 * l] * meta[8 * l + 1];
  }
  // uniform over the workgroup, but it came through LDS: scalarise (SGPRs, see opaque())
  pbase = __builtin_amdgcn_readfirstlane(pbase);
  lvl = __builtin_amdgcn_readfirstlane(lvl);
  Hl = __builtin_amdgcn_readfirstlane(Hl); Wl = __builtin_amdgcn_readfirstlane(Wl); start = __builtin_amdgcn_readfirstlane(start);
  qsplit = __builtin_amdgcn_readfirstlane(qsplit); qpiece = __builtin_amdgcn_readfirstlane(qpiece);
  pitch = __builtin_amdgcn_readfirstlane(pitch);
  x0 = __builtin_amdgcn_readfirstlane(x0); x1 = __builtin_amdgcn_readfirstlane(x1);
  y0 = __builtin_amdgcn_readfirstlane(y0); y1 = __builtin_amdgcn_readfirstlane(y1);
  const int rows = pitch * (y1 - y0);
  constexpr int kRpg = (kTileRowsMax + kGroups - 1) / kGroups;
  float4_t racc[kRpg];
#pragma unroll
  for (int k = 0; k < kRpg; ++k) racc[k] = float4_t{0.f, 0.f, 0.f, 0.f};

  const int LP = d.L * P;
  const uint2_t* summ = summaries + ((int64_t(b) * d.M + m) * d.L + lvl) * int64_t(n_tiles);
  // sample (q, head m, level lvl, point k) of this batch element: index  base + q * (M * LP) + k  into attn, twice that
  // into loc; q * M * LP * 2 < 2^32 (msda_d32_gvtiles_supported)
  // compact: loc / attn are the grad_loc kernel's copies laid out [batch][head][level][query][point] -- a chunk of 128
  // queries is 4 KB + 2 KB of consecutive bytes.  In the op's own layout the 32 + 16 bytes of a (query, head, level)
  // sit in two cache lines of their own per query: 2 GB fetched per 720p B = 5 launch (PMC), the kernel bandwidth-bound.
  const int64_t s_base = compact ? ((int64_t(b) * d.M + m) * d.L + lvl) * int64_t(d.Lq) * P
                                 : (int64_t(b) * d.Lq * d.M + m) * LP + lvl * P;
  const TL* attn_bm = attn + s_base;
  const TL* loc_bm = loc + 2 * s_base;
  const uint32_t s_stride = compact ? uint32_t(P) : uint32_t(d.M) * uint32_t(LP);
  const TV* go_head = grad_out + (int64_t(b) * d.Lq * d.M + m) * D;
  const uint32_t q_stride = uint32_t(d.M) * uint32_t(D);
  // byte ranges of this batch element's slices (msda_d32_gvtiles_supported keeps them below 2^31)
  constexpr uint32_t kOutOfRange = 0x80000000u;
  const uint32_t n_samp = compact ? uint32_t(d.Lq) * P : uint32_t(d.Lq) * uint32_t(d.M) * uint32_t(LP);   // from s_base on, at most
  const __amdgpu_buffer_rsrc_t loc_src = uniform_rsrc(loc_bm, n_samp * 2u * uint32_t(sizeof(TL)));
  const __amdgpu_buffer_rsrc_t attn_src = uniform_rsrc(attn_bm, n_samp * uint32_t(sizeof(TL)));
  const __amdgpu_buffer_rsrc_t go_src = uniform_rsrc(go_head, uint32_t(d.Lq) * uint32_t(d.M) * uint32_t(D) * uint32_t(sizeof(TV)));
  const float Hf = float(Hl), Wf = float(Wl);
  const int tile_mask = (1 << tile_shift) - 1;
  const int tpc_shift = 7 - tile_shift;              // tiles per chunk = 128 >> tile_shift
  const int dr[4] = {0, 1, pitch, pitch + 1};

  int gchunk = 0;                                    // parity of the double-buffered row counters

  // the tiles this workgroup takes: all of them, or one piece of a query-split level
  const int t_per = (n_tiles + qsplit - 1) / qsplit;
  const int t_lo = qpiece * t_per, t_hi = (qpiece + 1) * t_per < n_tiles ? (qpiece + 1) * t_per : n_tiles;
  for (int win0 = t_lo; win0 < t_hi; win0 += kTileWin) {
    const int n_w = t_hi - win0 < kTileWin ? t_hi - win0 : kTileWin;
    const int tw = opaque(tid);
    // ---- selection: which tiles of this round touch my rows -----------------------------------
    unsigned long long bal[kTileRounds];
    uint32_t hitbits = 0;
#pragma unroll
    for (int r = 0; r < kTileRounds; ++r) {
      const int ti = r * kThreads + tw;
      // words = x_lo | (0xffff - x_hi) << 16, y_lo | (0xffff - y_hi) << 16: the box of the pixels the tile's samples touch
      // (what packed 16-bit minima reduce; coordinates saturate at 0xfffe = "or beyond"); 0xffffffff = no taps
      const uint2_t v = ti < n_w ? summ[win0 + ti] : uint2_t{0xffffffffu, 0xffffffffu};
      const int xl = int(v.x & 0xffffu), yl = int(v.y & 0xffffu);
      int xh = 0xffff - int(v.x >> 16), yh = 0xffff - int(v.y >> 16);
      xh = xh >= 0xfffe ? 0x7fffffff : xh; yh = yh >= 0xfffe ? 0x7fffffff : yh;
      const bool h = v.x != 0xffffffffu && xl < x1 && xh >= x0 && yl < y1 && yh >= y0;
      const unsigned long long bh = __ballot(h);
      bal[r] = bh;
      hitbits |= uint32_t(h) << r;
      if (lane == 0) part_s[r * kWaves + wave] = uint32_t(__popcll(bh));
    }
    __syncthreads();
    if (tid < 64) {                                   // one wave scans the 32 (round, wave) pieces
      const uint32_t ns = tid < kTileParts ? part_s[tid] : 0u;
      const uint32_t is = wave_inclusive_scan(ns);
      if (tid < kTileParts) pre_s[tid] = is - ns;
      if (tid == kTileParts - 1) tot[0] = is;
    }
    __syncthreads();
    const int n_hit = int(tot[0]);
    if (n_hit == 0) continue;                         // uniform: no tile of this round lands here
#pragma unroll
    for (int r = 0; r < kTileRounds; ++r) {
      if (hitbits & (1u << r)) {
        const uint32_t lo32 = __builtin_amdgcn_mbcnt_lo(uint32_t(bal[r]), 0u);
        const uint32_t pos = pre_s[r * kWaves + wave] + __builtin_amdgcn_mbcnt_hi(uint32_t(bal[r] >> 32), lo32);
        hit[pos] = uint16_t(r * kThreads + tw);
      }
    }
    __syncthreads();

    // ---- chunks of 128 queries (whole tiles) over the kept tiles ---------------------------------
    const int n_chunks = (VNX_GVT_ABL == 1 || VNX_GVT_ABL == 4 || VNX_GVT_ABL == 5) ? 0 : ((n_hit << tile_shift) + kQcMax - 1) / kQcMax;
    float nx = 0.f, ny = 0.f, na = 0.f;
    bool nvalid = false;
    float4_t pg0 = {0.f, 0.f, 0.f, 0.f}, pg1 = pg0;
    // query of slot s (0..127) of chunk c, or -1
    auto slot_query = [&](int c, int s) {
      const int ht = (c << tpc_shift) + (s >> tile_shift);
      if (ht >= n_hit) return -1;
      const int q = ((win0 + int(hit[ht])) << tile_shift) + (s & tile_mask);
      return q < d.Lq ? q : -1;
    };
    // Through buffer descriptors with 32-bit byte offsets (a slot without a query reads out of range = zeros, no branch):
    // the global_load form spent 28 of the chunk's ~200 vector instructions on 64-bit address arithmetic.
    auto prefetch = [&](int c) {
      const int tq = opaque(tid);
      const int q = slot_query(c, tq >> 2);
      nvalid = q >= 0;
      const uint32_t si = nvalid ? __umul24(uint32_t(q), s_stride) + uint32_t(tq & 3) : kOutOfRange;
      if constexpr (sizeof(TL) == 4) {
        const uint2_t xy = __builtin_bit_cast(uint2_t, __builtin_amdgcn_raw_buffer_load_b64(loc_src, int(si * 8u), 0, 0));
        nx = __uint_as_float(xy.x); ny = __uint_as_float(xy.y);
        na = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(attn_src, int(si * 4u), 0, 0));
      } else {
        const uint32_t xy = __builtin_amdgcn_raw_buffer_load_b32(loc_src, int(si * 4u), 0, 0);
        const uint32_t a16 = __builtin_amdgcn_raw_buffer_load_b16(attn_src, int(si * 2u), 0, 0);
        nx = to_acc(__builtin_bit_cast(TL, uint16_t(xy & 0xffffu))); ny = to_acc(__builtin_bit_cast(TL, uint16_t(xy >> 16)));
        na = to_acc(__builtin_bit_cast(TL, uint16_t(a16)));
      }
      const int g0 = opaque(tid), g1 = g0 + kThreads;
      const int qa = slot_query(c, g0 >> 3), qb = slot_query(c, g1 >> 3);
      const uint32_t oa = qa >= 0 ? (__umul24(uint32_t(qa), q_stride) + uint32_t(g0 & 7) * 4u) * uint32_t(sizeof(TV)) : kOutOfRange;
      const uint32_t ob = qb >= 0 ? (__umul24(uint32_t(qb), q_stride) + uint32_t(g1 & 7) * 4u) * uint32_t(sizeof(TV)) : kOutOfRange;
      pg0 = load4_buf<TV>(go_src, oa);
      pg1 = load4_buf<TV>(go_src, ob);
    };
    prefetch(0);
    for (int c = 0; c < n_chunks; ++c, ++gchunk) {
      uint32_t* cnt = cnt2 + (gchunk & 1) * kTileRowsMax;
      uint32_t* cnt_next = cnt2 + ((gchunk + 1) & 1) * kTileRowsMax;
      const float x = nx, y = ny, a = na;
      const bool valid = nvalid;
      const int tc = opaque(tid);
      const uint32_t slot = uint32_t(tc) >> 2;
      const int grp = tc >> 3, ch4 = tc & 7;
      grows[tc] = pg0;
      grows[tc + kThreads] = pg1;
      if (c + 1 < n_chunks) prefetch(c + 1);
      uint32_t mask = 0;
      int row00 = 0;
      float wt[4] = {0.f, 0.f, 0.f, 0.f};
      uint32_t rank[4] = {0u, 0u, 0u, 0u};
      if (valid) {
        const float h = y * Hf - 0.5f, w = x * Wf - 0.5f;                     // cuh:285-286
        if (h > -1.f && w > -1.f && h < Hf && w < Wf) {                        // cuh:288
          const float hf = floorf(h), wf = floorf(w);
          const int h0 = int(hf), w0 = int(wf);
          const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw = 1.f - lw;
          // a corner counts if it is inside the unit's rectangle (which lies inside the map)
          const int lx = w0 - x0, ly = h0 - y0, bwx = x1 - x0, bhy = y1 - y0;
          const bool cl = lx >= 0 && lx < bwx, cr = lx + 1 >= 0 && lx + 1 < bwx;
          const bool rt = ly >= 0 && ly < bhy, rb = ly + 1 >= 0 && ly + 1 < bhy;
          mask = uint32_t(rt && cl) | (uint32_t(rt && cr) << 1) | (uint32_t(rb && cl) << 2) | (uint32_t(rb && cr) << 3);
          row00 = ly * pitch + lx;
          wt[0] = a * (hh * hw); wt[1] = a * (hh * lw); wt[2] = a * (lh * hw); wt[3] = a * (lh * lw);
        }
      }
      if (VNX_GVT_ABL == 2) mask = (wt[0] + wt[1] + wt[2] + wt[3] == 12345.f) ? mask : 0u;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (mask & (1u << t))
          rank[t] = __hip_atomic_fetch_add(cnt + row00 + dr[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __syncthreads();
      {
        const uint32_t my_cnt = tid < rows ? cnt[tid] : 0u;
        const uint32_t incl = wave_inclusive_scan(my_cnt);
        const uint32_t wave_total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
        uint32_t base = 0;
        if (lane == 0 && wave_total != 0)
          base = __hip_atomic_fetch_add(alloc, wave_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
        if (tid < rows) offs[tid] = base + incl - my_cnt;
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (mask & (1u << t)) list[offs[row00 + dr[t]] + rank[t]] = uint2_t{slot, __float_as_uint(wt[t])};
      __syncthreads();
      if (tid == 0) alloc[0] = 0;
      uint32_t rn[kRpg], ro[kRpg];
#pragma unroll
      for (int k = 0; k < kRpg; ++k) {
        const int row = grp + k * kGroups;
        rn[k] = row < rows ? cnt[row] : 0u;
        ro[k] = row < rows ? offs[row] : 0u;
        if (row < rows) cnt_next[row] = 0;
        if (VNX_GVT_ABL == 3) rn[k] = rn[k] == 0x7fffffffu ? 1u : 0u;
      }
      const float4_t* g4 = grows + ch4;
#pragma unroll
      for (int k = 0; k < kRpg; ++k) {
        const uint32_t n = rn[k];
        if (n == 0) continue;
        const uint2_t* seg = list + ro[k];
        float4_t a1 = {0.f, 0.f, 0.f, 0.f};
        uint32_t i = 0;
        for (; i + 2 <= n; i += 2) {
          const uint2_t e0 = seg[i], e1 = seg[i + 1];
          const float4_t x0 = g4[e0.x * 8], x1 = g4[e1.x * 8];
          racc[k] += __uint_as_float(e0.y) * x0;
          a1 += __uint_as_float(e1.y) * x1;
        }
        if (i < n) {
          const uint2_t e = seg[i];
          a1 += __uint_as_float(e.y) * g4[e.x * 8];
        }
        racc[k] += a1;
      }
      __syncthreads();
    }
  }

  // local row -> pixel (y0 + row / pitch, x0 + row % pitch); the columns past x1 of an edge block hold nothing
  const int te = opaque(tid);
  const int grp = te >> 3, ch4 = te & 7;
  const uint32_t inv = (65536u + uint32_t(pitch) - 1u) / uint32_t(pitch);      // row / pitch exactly for row < 256
  const int64_t level_elem = ((int64_t(b) * d.S + start) * d.M + m) * D;
  if (VNX_GVT_ABL == 5) return;                      // (ablation: no final store at all)
  if (VNX_GVT_ABL == 4 && qsplit > 1) return;        // (ablation: no store of the query pieces' partial rows)
  if (qsplit > 1) {   // a piece of a query-split level: its rows go to its own slab of fp32 partial rows (plain stores, read
                      // back right away by gv_split_finish_kernel, which adds the pieces and writes grad_value)
    float* part = partials + ((int64_t(b) * d.M + m) * gv_partial_rows_bound(d.S, d.L, d.B * d.M) + pbase + int64_t(qpiece) * (Hl * Wl)) * D;
#pragma unroll
    for (int k = 0; k < kRpg; ++k) {
      const int row = grp + k * kGroups;
      const int ry = int((uint32_t(row) * inv) >> 16), rx = row - ry * pitch;
      if (row < rows && rx < x1 - x0)
        *reinterpret_cast<float4_t*>(part + uint32_t((y0 + ry) * Wl + x0 + rx) * uint32_t(D) + ch4 * 4) = racc[k];
    }
    return;
  }
  TV* out = grad_value + level_elem;
#pragma unroll
  for (int k = 0; k < kRpg; ++k) {
    const int row = grp + k * kGroups;
    const int ry = int((uint32_t(row) * inv) >> 16), rx = row - ry * pitch;
    if (row < rows && rx < x1 - x0)
      store4<TV>(out + __umul24(uint32_t((y0 + ry) * Wl + x0 + rx), q_stride) + ch4 * 4, racc[k]);
  }
}


// The pieces of the query-split levels -> grad_value.  One thread per (batch, head, split pixel, 4 channels): adds the
// level's qs partial rows in piece order (a fixed order: these rows' sums do not depend on scheduling) and writes the row
// of grad_value, in its dtype, exactly once.  The level table is the grad_value kernel's (same gv_level_grid /
// gv_query_splits arguments); nothing to do, on the device, when no level is split or the levels are not packed.
template <typename TV>
__global__ void __launch_bounds__(256)
gv_split_finish_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                       const float* __restrict__ partials, TV* __restrict__ grad_value, MsdaDims d, int units_min) {
  constexpr int D = 32, P = 4;
  if (!levels_packed(shapes, lsi, d.L, d.S)) return;
  __shared__ int s_start[kTileLevels], s_n[kTileLevels], s_qs[kTileLevels], s_before[kTileLevels + 1], s_base[kTileLevels];
  if (int(threadIdx.x) < d.L) {
    const int l = threadIdx.x;
    const int H = int(shapes[2 * l]), W = int(shapes[2 * l + 1]);
    const GvGrid g = gv_level_grid(H, W, units_min, kTileRowsMax);
    const int qs = gv_query_splits(g.nbx * g.nby, d.Lq, P, true, d.B * d.M);
    s_start[l] = int(lsi[l]); s_qs[l] = qs; s_n[l] = qs > 1 ? H * W : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int before = 0, base = 0;
    for (int l = 0; l < d.L; ++l) { s_before[l] = before; s_base[l] = base; before += s_n[l]; base += s_qs[l] * s_n[l]; }
    s_before[d.L] = before;
  }
  __syncthreads();
  const int split_rows = s_before[d.L];                            // split pixels per (batch, head)
  const int per_bm = split_rows * 8;                               // 16-B pieces per (batch, head): blockIdx.y
  const int bm = int(blockIdx.y);
  const int b = bm / d.M, m = bm - b * d.M;
  const float* slab = partials + int64_t(bm) * gv_partial_rows_bound(d.S, d.L, d.B * d.M) * D;
  TV* gv_bm = grad_value + (int64_t(b) * d.S * d.M + m) * D;
  for (int i = int(blockIdx.x) * int(blockDim.x) + int(threadIdx.x); i < per_bm; i += int(gridDim.x) * int(blockDim.x)) {
    const int r = i >> 3, ch4 = i & 7;
    int l = 0;
    while (l + 1 < d.L && r >= s_before[l + 1]) ++l;               // (unsplit levels: s_before[l + 1] == s_before[l])
    const int px = r - s_before[l], n_l = s_n[l], qs = s_qs[l];
    const float* src = slab + (int64_t(s_base[l] + px) * D + ch4 * 4);
    float4_t acc = *reinterpret_cast<const float4_t*>(src);
    for (int j = 1; j < qs; ++j) acc += *reinterpret_cast<const float4_t*>(src + int64_t(j) * n_l * D);
    store4<TV>(gv_bm + int64_t(s_start[l] + px) * d.M * D + ch4 * 4, acc);
  }
}

}  // namespace rec


// bytes of the tile words: [batch][head][level][tile] x 2
size_t msda_gvtiles_summary_bytes(const MsdaDims& d, int tile_queries) {
  const size_t n_tiles = (size_t(d.Lq) + tile_queries - 1) / tile_queries;
  return size_t(8) * size_t(d.B) * d.M * d.L * n_tiles;
}

// bytes of the query pieces' partial rows (fp32), all (batch, head) slabs
size_t msda_gvtiles_partial_bytes(const MsdaDims& d) {
  return size_t(d.B) * size_t(d.M) * size_t(gv_partial_rows_bound(d.S, d.L, d.B * d.M)) * 32 * sizeof(float);
}

// Workgroups per (batch, head): the host knows S, not the level shapes.  gv_level_grid: a narrow level (W <= 63: bands
// of floor(256 / W) >= 4 rows) has at most n / 193 + 1 units; a wider one at most (W / bw + 1)(H / bh + 1) blocks with
// bw > 21.3, bw * bh >= 224 and H >= 8, i.e. n / 224 + n / 170 + n / 512 + 1 <= n / 81 + 1; a flat wide one (H < 8)
// n / 128 + 1.  Exhaustively (every H <= 1 200 x W <= 20 000, units_min 1 / 2 / 16: tools/check_units_bound.py) a level
// has at most 3 n / 256 + units_min units, so 3 * ceil(S / 256) + (units_min + 2) per level bounds them -- a unit past the
// grid would lose its rows; tests/test_units_bound.py holds the kernel's grid to the launcher's bound through
// vnx_debug_gvtiles_units on random pyramids -- and
// 4 * (pieces - 1) for the query pieces of levels of at most four units.  Workgroups past the real count exit at once;
// they are the LAST of the grid (units are numbered from the coarsest level back) and overlap the real ones: forcing the
// exact count (42 per (batch, head) at 360p, the bound is 124) changes nothing, 300 costs 4 us (tools/r3_call28.sh).
int msda_gvtiles_units_bound(const MsdaDims& d, int units_min) {
#ifdef VNX_TILES_BOUND_FORCE      // A/B build: what the workgroups past the real unit count cost (valid for ONE shape only)
  return VNX_TILES_BOUND_FORCE;
#endif
  const int qs = gv_query_splits(1, d.Lq, d.P, true, d.B * d.M);
  return d.L * (units_min + 2) + 3 * ((d.S + rec::kTileRowsMax - 1) / rec::kTileRowsMax) + d.L * 4 * (qs - 1);
}

bool msda_d32_gvtiles_supported(int vdt, int ldt, const MsdaDims& d) {
  if (d.D != 32 || d.P != 4 || vdt == VNX_F64) return false;
  if (vdt == VNX_F32 && ldt != VNX_F32) return false;
  if (d.L != rec::kTileLevels) return false;
  // 24-bit stride multiplies: query index, heads x channels and heads x samples below 2^24; element offsets below 2^32
  if (d.Lq >= (1 << 24) || d.M * 32 >= (1 << 24) || d.M * d.L * 4 >= (1 << 24)) return false;
  if (int64_t(d.Lq) * d.M * 32 >= (int64_t(1) << 32) || int64_t(d.Lq) * d.M * d.L * 8 >= (int64_t(1) << 32)) return false;
  // byte offsets into one batch element's locations (8 B per sample) and grad_out rows: 32-bit buffer offsets below 2^31
  if (int64_t(d.Lq) * d.M * d.L * d.P * 8 >= (int64_t(1) << 31) || int64_t(d.Lq) * d.M * 32 * 4 >= (int64_t(1) << 31)) return false;
  const int64_t blocks = int64_t(d.B) * d.M * msda_gvtiles_units_bound(d, 16);
  return blocks < (int64_t(1) << 31);
}

template <typename TV, typename TL>
static int launch_gvtiles(const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                          const void* summaries, const void* grad_out, void* grad_value, const MsdaDims& d,
                          int units_min, int tile_queries, float* partials, int compact, hipStream_t stream) {
  int tile_shift = 0;
  while ((1 << tile_shift) < tile_queries) ++tile_shift;
  if ((1 << tile_shift) != tile_queries || tile_queries > rec::kQcMax) {
    set_error("msda_backward_gvtiles: tile of %d queries (a power of two <= %d is required)", tile_queries, rec::kQcMax);
    return VNX_ERR_UNSUPPORTED;
  }
  const int n_tiles = (d.Lq + tile_queries - 1) / tile_queries;
  const int units_pb = msda_gvtiles_units_bound(d, units_min);
  const int64_t blocks = ((int64_t(d.B) * units_pb + 1) & ~int64_t(1)) * d.M;   // even: gv_decode_block
  hipLaunchKernelGGL((rec::msda_bwd_gv_tiles_kernel<TV, TL>), dim3(uint32_t(blocks)), dim3(rec::kThreads),
                     rec::kTilesLdsBytes, stream, shapes, lsi, (const TL*)loc, (const TL*)attn,
                     (const rec::uint2_t*)summaries, (const TV*)grad_out, (TV*)grad_value, d, units_min, tile_shift, n_tiles,
                     partials, compact, units_pb);
  int st = check_launch("msda_bwd_gv_tiles");
  if (st != VNX_OK) return st;
  // the pieces of the query-split levels -> grad_value (sized by the bound on split pixels; idle threads leave at once)
  // grid: x = a share of the (batch, head)'s split pixels (grid-stride), y = (batch, head)
  const int64_t px_bound = gv_partial_rows_bound(d.S, d.L, d.B * d.M) / gv_split_pieces_max(d.B * d.M);
  int64_t fx = (px_bound * 8 + 255) / 256;
  fx = fx < 1 ? 1 : (fx > 16 ? 16 : fx);
  if (int64_t(d.B) * d.M > 65535) {
    set_error("msda_backward_gvtiles: batch x heads = %lld exceeds the grid limit of the finishing kernel", (long long)(int64_t(d.B) * d.M));
    return VNX_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL((rec::gv_split_finish_kernel<TV>), dim3(uint32_t(fx), uint32_t(d.B * d.M)), dim3(256), 0, stream, shapes, lsi,
                     (const float*)partials, (TV*)grad_value, d, units_min);
  return check_launch("msda_gv_split_finish");
}

// grad_value from the op's inputs and the tile words (two launches: the units, then the pieces of the query-split levels);
// a no-op on the device when the levels are not packed.  partials: msda_gvtiles_partial_bytes(d) bytes of scratch.
int msda_backward_gvtiles_d32(int vdt, int ldt, const int64_t* shapes, const int64_t* lsi, const void* loc,
                              const void* attn, const void* summaries, const void* grad_out, void* grad_value,
                              MsdaDims d, int tile_queries, float* partials, bool compact, hipStream_t stream) {
  const int units_min = gv_units_min(d, true, kernel_variant());
#define VNX_ARGS shapes, lsi, loc, attn, summaries, grad_out, grad_value, d, units_min, tile_queries, partials, int(compact), stream
  if (vdt == VNX_F32) return launch_gvtiles<float, float>(VNX_ARGS);
  if (vdt == VNX_BF16 && ldt == VNX_F32) return launch_gvtiles<bf16_t, float>(VNX_ARGS);
  if (vdt == VNX_BF16 && ldt == VNX_BF16) return launch_gvtiles<bf16_t, bf16_t>(VNX_ARGS);
  if (vdt == VNX_F16 && ldt == VNX_F32) return launch_gvtiles<f16_t, float>(VNX_ARGS);
  if (vdt == VNX_F16 && ldt == VNX_F16) return launch_gvtiles<f16_t, f16_t>(VNX_ARGS);
#undef VNX_ARGS
  set_error("msda_backward_gvtiles_d32: unsupported dtype pair (%d, %d)", vdt, ldt);
  return VNX_ERR_INVALID_ARGUMENT;
}

}  // namespace vnx
