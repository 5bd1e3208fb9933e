// msda_d32_gvrec.hip -- grad_value, owner-computes, fed by the sample records the
// grad_loc/grad_attn kernel leaves behind.
//
// msda_d32_gv.hip has every unit (a pixel range of one level for one (batch, head)) recompute the
// bilinear geometry of ALL of its level's samples, 12x redundantly for the finest 360p level, and
// stage every chunk's grad_out rows in LDS whether or not a tap lands in the unit.  Phase
// timestamps and PMC counters showed that bookkeeping, not the accumulation, to be two thirds of
// its time (~400 issued instructions per thread and chunk on 16 waves per CU).
//
// Here the per-query kernel (msda_bwd_d32_kernel, which computes every sample's geometry anyway)
// writes one 16-byte record per sample {(h0+1)<<16 | (w0+1), lh, lw, attn} into a workspace laid
// out [batch][head][level][query*points] -- contiguous for each unit's scan -- and this kernel
//   * reads its level's records with coalesced 16-B loads (no location / weight loads, no floor,
//     no integer division, no range test in floating point),
//   * ranks the taps that land in its pixel range by destination row (integer LDS atomics),
//     turns the row counts into segment offsets with a DPP wave scan, scatters {query, weight},
//   * stages the chunk's grad_out rows of this head in LDS (reading them per tap straight from L2
//     instead measured 19 us for a single round of units: one serialised memory latency per row),
//   * lets 8-lane groups (16 B per lane = one 32-channel row) sum each row's segment in registers
//     and update the LDS slab row once,
//   * writes the slab with 16-B stores.  No floating-point atomics, no zero-fill pass, no fp32
//     image for 16-bit tensors; every grad_value row has exactly one owner.
// Levels must be packed (checked on the device; see msda_d32_gv.hip / capi.hip).
#include "msda_gv_common.h"

namespace vnx {
namespace rec {

// slab 40 K + grad_out rows 16 K + tap list 16 K + 3 x 320 counters/offsets + allocator + level
// table = 76.8 KiB -> two units per CU.
constexpr size_t kLdsBytes = size_t(kRowsMax) * 128 + size_t(kQcMax) * 128 + size_t(kThreads) * 32 +
                             size_t(kRowsMax) * 12 + 16 + 4 * kLevelsMax * 4;

// Development aid: per-workgroup phase timestamps, written when debug != 0: shader-clock
// ticks (s_memtime; variant 408) or constant-rate wall-clock ticks (s_memrealtime; variant 412,
// what bench.py uses for this kernel's span: slots 0 and 12 of every workgroup).  Unlike the
// other two tuned kernels this one takes no stamp-region argument: at its 80-VGPR budget one
// more kernel argument costs 20 spilled registers.
__device__ unsigned long long g_rec_stamps[4096 * 16];
#define VNX_STAMP(k)                                                              \
  do {                                                                            \
    if (debug && tid == 0 && blockIdx.x < 4096)                                   \
      g_rec_stamps[blockIdx.x * 16 + (k)] =                                        \
          debug == 5 ? (unsigned long long)wall_clock64() : __builtin_readcyclecounter(); \
  } while (0)

// More stamps inside the selection kernel cost it ~20 spilled registers: diagnostic builds only (-DVNX_SEL_STAMPS)
#ifdef VNX_SEL_STAMPS
#define VNX_SEL_STAMP(k) VNX_STAMP(k)
#else
#define VNX_SEL_STAMP(k) do { } while (0)
#endif

// RS (register slab): the rows a group owns accumulate in its registers (4 VGPRs per row)
// instead of a 40 KiB LDS slab -- no slab zero-fill, no read-modify-write in the apply phase, and
// the LDS left (staged rows + tap list + counters, 37 KiB) lets 3 units share a CU at <= 80 VGPRs
// (two taps in flight per row instead of four).  Measured on MI355X (decoder 360p, B=5, cold):
// 38.9 us per backward against 43.4 us for the LDS slab at 2 units per CU (variant 420); 4 units
// per CU needs <= 64 VGPRs and spills 41 registers: 49.9 us.
template <typename TV, int P_T, int RS>   // RS = 0: LDS slab, 2 units per CU; 3 | 4: register slab, RS units per CU
__global__ void __launch_bounds__(kThreads, (RS ? RS : 2) * kWaves / 4)  // 2nd argument = waves per SIMD
msda_bwd_gv_rec_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                       const uint4_t* __restrict__ records, const TV* __restrict__ grad_out,
                       TV* __restrict__ grad_value, MsdaDims d, int units_min, int units_bound, int debug) {
  constexpr int D = 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4_t* slab = reinterpret_cast<float4_t*>(smem);                        // [rows][8]   (!RS)
  float4_t* grows = slab + (RS ? 0 : kRowsMax * 8);                          // [qc][8] grad_out rows
  uint2_t* list = reinterpret_cast<uint2_t*>(grows + kQcMax * 8);            // [4*threads] taps
  uint32_t* cnt2 = reinterpret_cast<uint32_t*>(list + 4 * kThreads);         // [2][rows]
  uint32_t* offs = cnt2 + 2 * kRowsMax;                                      // [rows]
  uint32_t* alloc = offs + kRowsMax;                                         // [4]
  int* meta = reinterpret_cast<int*>(alloc + 4);                             // [4*L]

  const int P = P_T > 0 ? P_T : d.P;
  const int tid = threadIdx.x, lane = tid & 63;
  // Dispatch order = cost order: the unit index is the slow axis and units are numbered from the LAST
  // level back (below), so the coarse levels' units -- three sort chunks each at the decoder shape against
  // one for a fine-level unit, 12-15 us against 6-7 -- start first instead of last (they used to start up
  // to 7 us into the kernel and set its end: 20 us for 8.5 us of mean work per workgroup).
  // (The coarse levels' units are also split by query range from 1024 queries up: gv_query_splits.)
  // Tried: a per-(level, window) table of the unit range its samples touch, filled by the grad_loc kernel with
  // atomicMin, so that a unit skips empty selection windows.  The 400 K same-address atomics took the
  // grad_loc kernel from 108 to 700 us at the encoder shape (12 to 160 at the decoder shape) and the skipped
  // windows bought this kernel 5 us of 130: a window that selects nothing costs a tag load and a ballot.
  const int rest = blockIdx.x / d.M;
  const int unit = rest / d.B;
  const int b = rest - unit * d.B;
  const int m = (blockIdx.x % d.M + b) % d.M;      // head <-> XCD map rotates with the batch element (msda_d32.hip)
  VNX_STAMP(0);

  // level table: lane l works out level l's unit split once (two integer divisions per level)
  if (tid < d.L) {
    const int H = int(shapes[2 * tid]), W = int(shapes[2 * tid + 1]);
    const int n = H * W;
    const GvSplit sp = gv_level_split(n, units_min, kRowsMax);
    const int units = sp.units, rpu = sp.rpu;
    meta[4 * tid] = H; meta[4 * tid + 1] = W; meta[4 * tid + 2] = int(lsi[tid]);
    meta[4 * tid + 3] = units | (rpu << 12);
  }
  for (int i = tid; i < 2 * kRowsMax; i += kThreads) cnt2[i] = 0;
  if (tid == 0) alloc[0] = 0;
  __syncthreads();
  VNX_STAMP(1);

  int lvl = -1, r0 = 0, r1 = 0, Hl = 0, Wl = 0, start = 0;
  {
    int running = 0;
    bool packed = true;
    int u = unit;
    for (int l = 0; l < d.L; ++l) {
      packed = packed && (meta[4 * l + 2] == running);
      running += meta[4 * l] * meta[4 * l + 1];
    }
    for (int l = d.L - 1; l >= 0; --l) {       // units are numbered from the last (coarsest) level back
      const int H = meta[4 * l], W = meta[4 * l + 1], st = meta[4 * l + 2], ur = meta[4 * l + 3];
      const int n = H * W, units = ur & 0xfff, rpu = ur >> 12;
      if (lvl < 0) {
        if (u < units) {
          lvl = l; Hl = H; Wl = W; start = st;
          r0 = u * rpu;
          r1 = r0 + rpu < n ? r0 + rpu : n;
        } else {
          u -= units;
        }
      }
    }
    packed = packed && (running == d.S);
    if (!packed || lvl < 0) return;  // uniform over the workgroup
  }
  // uniform over the workgroup, but it came through LDS: scalarise (SGPRs instead of VGPRs, see opaque())
  lvl = __builtin_amdgcn_readfirstlane(lvl); r0 = __builtin_amdgcn_readfirstlane(r0); r1 = __builtin_amdgcn_readfirstlane(r1);
  Hl = __builtin_amdgcn_readfirstlane(Hl); Wl = __builtin_amdgcn_readfirstlane(Wl); start = __builtin_amdgcn_readfirstlane(start);
  const int rows = r1 - r0;
  if (!RS)
    for (int i = tid; i < rows * 8; i += kThreads) slab[i] = float4_t{0.f, 0.f, 0.f, 0.f};
  constexpr int kRpgAll = (kRowsMax + kGroups - 1) / kGroups;
  float4_t racc[kRpgAll];
#pragma unroll
  for (int k = 0; k < kRpgAll; ++k) racc[k] = float4_t{0.f, 0.f, 0.f, 0.f};

  const int n_samples = d.Lq * P;
  const int qc_ = kThreads / P < kQcMax ? kThreads / P : kQcMax;
  const int n_chunks = (d.Lq + qc_ - 1) / qc_;
  const uint4_t* my_recs = records + ((int64_t(b) * d.M + m) * d.L + lvl) * n_samples;
  const TV* go_head = grad_out + (int64_t(b) * d.Lq * d.M + m) * D;
  const int64_t q_stride = int64_t(d.M) * D;
  const uint4_t none = {0xffffffffu, 0u, 0u, 0u};
  uint4_t next = (tid < qc_ * P && tid < n_samples) ? my_recs[tid] : none;
  // a chunk = qc queries = qc*P <= 512 samples; its grad_out rows are staged in LDS (two 16-B
  // pieces per thread), loaded one chunk ahead like the records
  const int qc = kThreads / P < kQcMax ? kThreads / P : kQcMax;
  const int g0 = tid, g1 = tid + kThreads;
  float4_t pg0 = {0.f, 0.f, 0.f, 0.f}, pg1 = pg0;
  auto prefetch_rows = [&](int chunk) {
    const int qa = chunk * qc + (g0 >> 3), qb = chunk * qc + (g1 >> 3);
    if ((g0 >> 3) < qc && qa < d.Lq) pg0 = load4<TV>(go_head + int64_t(qa) * q_stride + (g0 & 7) * 4);
    if ((g1 >> 3) < qc && qb < d.Lq) pg1 = load4<TV>(go_head + int64_t(qb) * q_stride + (g1 & 7) * 4);
  };
  prefetch_rows(0);
  VNX_STAMP(2);

  const int grp = tid >> 3, ch4 = tid & 7;
  const int dr[4] = {0, 1, Wl, Wl + 1};
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    uint32_t* cnt = cnt2 + (chunk & 1) * kRowsMax;
    uint32_t* cnt_next = cnt2 + ((chunk + 1) & 1) * kRowsMax;
    if (chunk == 0) VNX_STAMP(3);
    const uint4_t r = next;
    const int s = chunk * qc * P + tid;               // sample index within (b, m, level)
    if ((g0 >> 3) < qc) grows[g0] = pg0;
    if ((g1 >> 3) < qc) grows[g1] = pg1;
    if (chunk + 1 < n_chunks) {  // the next chunk's loads: in flight while this one is sorted and applied
      const int s2 = s + qc * P;
      next = (tid < qc * P && s2 < n_samples) ? my_recs[s2] : none;
      prefetch_rows(chunk + 1);
    }
    uint32_t mask = 0;
    int row00 = 0;
    float wt[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t rank[4] = {0u, 0u, 0u, 0u};
    if (r.x != 0xffffffffu && tid < qc * P) {
      int h0, w0;
      gv_unpack_corner(r.x, Hl, Wl, h0, w0);
      const float lh = __uint_as_float(r.y), lw = __uint_as_float(r.z), a = __uint_as_float(r.w);
      const float hh = 1.f - lh, hw = 1.f - lw;
      const bool top = h0 >= 0, bot = h0 + 1 <= Hl - 1, lef = w0 >= 0, rig = w0 + 1 <= Wl - 1;
      const int p00 = h0 * Wl + w0;
      mask = (uint32_t(top && lef && p00 >= r0 && p00 < r1)) |
             (uint32_t(top && rig && p00 + 1 >= r0 && p00 + 1 < r1) << 1) |
             (uint32_t(bot && lef && p00 + Wl >= r0 && p00 + Wl < r1) << 2) |
             (uint32_t(bot && rig && p00 + Wl + 1 >= r0 && p00 + Wl + 1 < r1) << 3);
      row00 = p00 - r0;
      wt[0] = a * (hh * hw); wt[1] = a * (hh * lw); wt[2] = a * (lh * hw); wt[3] = a * (lh * lw);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (mask & (1u << t))
        rank[t] = __hip_atomic_fetch_add(cnt + row00 + dr[t], 1u, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP);
    if (chunk == 0) VNX_STAMP(4);
    __syncthreads();
    if (chunk == 0) VNX_STAMP(5);

    // row counts -> segment offsets: DPP wave scan + one LDS allocation per wave
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
    if (chunk == 0) VNX_STAMP(6);

    // scatter {query, weight} into the row segments
    {
      const uint32_t slot = uint32_t(tid) / uint32_t(P);  // query slot inside the chunk
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (mask & (1u << t)) list[offs[row00 + dr[t]] + rank[t]] = uint2_t{slot, __float_as_uint(wt[t])};
    }
    __syncthreads();
    if (chunk == 0) VNX_STAMP(7);

    // 8-lane groups own rows: sum the row's segment in registers, one slab update
    if (tid == 0) alloc[0] = 0;
    constexpr int kRpg = (kRowsMax + kGroups - 1) / kGroups;
    uint32_t rn[kRpg], ro[kRpg];
#pragma unroll
    for (int k = 0; k < kRpg; ++k) {
      const int row = grp + k * kGroups;
      rn[k] = row < rows ? cnt[row] : 0u;
      ro[k] = row < rows ? offs[row] : 0u;
      if (row < rows) cnt_next[row] = 0;
    }
    // One row at a time, four taps in flight within the row.  (Walking the group's rows
    // side by side -- one tap of each row per step -- was measured slower: 70 us total backward
    // against 42 us, because every group then iterates to the longest row of the wave.)
    const float4_t* g4 = grows + ch4;
#pragma unroll
    for (int k = 0; k < kRpg; ++k) {
      const uint32_t n = rn[k];
      if (n == 0) continue;
      const int row = grp + k * kGroups;
      const uint2_t* seg = list + ro[k];
      uint32_t i = 0;
      if (RS) {  // register diet: accumulate straight into the row's registers, two taps in flight
        float4_t a1 = {0.f, 0.f, 0.f, 0.f};
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
        continue;
      }
      float4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
      for (; i + 4 <= n; i += 4) {
        const uint2_t e0 = seg[i], e1 = seg[i + 1], e2 = seg[i + 2], e3 = seg[i + 3];
        const float4_t x0 = g4[e0.x * 8], x1 = g4[e1.x * 8], x2 = g4[e2.x * 8], x3 = g4[e3.x * 8];
        a0 += __uint_as_float(e0.y) * x0;
        a1 += __uint_as_float(e1.y) * x1;
        a2 += __uint_as_float(e2.y) * x2;
        a3 += __uint_as_float(e3.y) * x3;
      }
      for (; i < n; ++i) {
        const uint2_t e = seg[i];
        a0 += __uint_as_float(e.y) * g4[e.x * 8];
      }
      slab[row * 8 + ch4] += (a0 + a1) + (a2 + a3);
    }
    if (chunk == 0) VNX_STAMP(8);
    __syncthreads();
    if (chunk == 0) VNX_STAMP(9);
    if (chunk == 1) VNX_STAMP(10);
  }
  VNX_STAMP(11);

  TV* out = grad_value + ((int64_t(b) * d.S + start + r0) * d.M + m) * D;
  if (RS) {
#pragma unroll
    for (int k = 0; k < kRpgAll; ++k) {
      const int row = grp + k * kGroups;
      if (row < rows) store4<TV>(out + int64_t(row) * d.M * D + ch4 * 4, racc[k]);
    }
  } else {
    for (int i = tid; i < rows * 8; i += kThreads) {
      const int row = i >> 3, c4 = i & 7;
      store4<TV>(out + int64_t(row) * d.M * D + c4 * 4, slab[i]);
    }
  }
  VNX_STAMP(12);
}


// ---------------------------------------------------------------------------------------------
// Per-unit sample selection (P == 4).
//
// The kernel above makes every unit of a level scan ALL of that level's samples, chunk by chunk,
// although a fine-level unit receives a small share of them: 1/12 at the finest 360p level, and at
// the encoder shape (Lq = S) 40 chunks are sorted for taps that four would hold.  The grad_loc
// kernel therefore also leaves, per sample, the range of units its four corners touch
// (`unit_lo | unit_hi << 16`, 0xffffffff = no taps; same [batch][head][level][query*points] layout),
// and a unit first COMPACTS: it reads those 4-byte words for a window of 2048 samples (512
// queries), keeps the samples whose range contains it (wave ballots + one 32-entry scan: their
// ascending order is kept, so every sum stays deterministic), numbers the distinct queries among
// them, and only then runs the sort / apply chunks -- over the kept samples, 128 distinct queries
// (<= 512 samples) per chunk, staging only those queries' grad_out rows.
// Tried (round 2, late): 176 instead of 128 distinct queries per chunk (22 KiB of staged rows: what three units per CU
// leave of the LDS; 184 no longer fits: 33.6 us), a chunk whose queries keep more than 512 samples being sorted and
// applied in several rounds over the same staged rows -- the coarse half-level units of the decoder shape keep ~280
// distinct queries, i.e. two chunks instead of three on what the phase stamps call the critical path.  Same box, A/B:
// decoder backward 27.6 vs 27.9-28.2 us, but the encoder shape LOSES even with the kernel templated back to 128 for
// large calls (225.5 vs 217.9 us at 360p, 492 vs 452 us at 720p B = 2: the step loop no longer unrolls into the old
// schedule).  Not kept.
// Timing ablations of the selection kernel (A/B builds of the development library only; wrong grad_value by construction):
// 1 = no chunks (set-up + selection + final store), 2 = chunks without taps (no rank atomics / scatter / apply), 3 = no apply,
// 4 = no selection either (returns after the level table: launch + set-up)
#ifndef VNX_SEL_ABL
#define VNX_SEL_ABL 0
#endif
constexpr int kWin = 2048;                        // samples per selection window
constexpr int kWinQueries = kWin / 4;
constexpr int kSelRounds = kWin / kThreads;       // 4
constexpr int kSelParts = kSelRounds * kWaves;    // 32 (round, wave) pieces per window
constexpr size_t kSelLdsBytes = size_t(kQcMax) * 128 + size_t(kThreads) * 32 + size_t(kRowsMax) * 12 + 16 +
                                4 * kLevelsMax * 4 + size_t(kWin) * 4 + size_t(kWinQueries) * 2 +
                                size_t(kSelParts) * 16 + 64;

template <typename TV, bool NT>
__global__ void __launch_bounds__(kThreads, VNX_SEL_UNITS_PER_CU * kWaves / 4)
msda_bwd_gv_sel_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                       const uint4_t* __restrict__ records, const uint32_t* __restrict__ unit_ids,
                       const TV* __restrict__ grad_out, TV* __restrict__ grad_value, MsdaDims d, int units_min,
                       float* __restrict__ split_image, int debug) {
  constexpr int D = 32, P = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4_t* grows = reinterpret_cast<float4_t*>(smem);                       // [128][8] grad_out rows
  uint2_t* list = reinterpret_cast<uint2_t*>(grows + kQcMax * 8);            // [4*threads] taps
  uint32_t* cnt2 = reinterpret_cast<uint32_t*>(list + 4 * kThreads);         // [2][rows]
  uint32_t* offs = cnt2 + 2 * kRowsMax;                                      // [rows]
  uint32_t* alloc = offs + kRowsMax;                                         // [4]
  int* meta = reinterpret_cast<int*>(alloc + 4);                             // [4*L]
  uint16_t* sel_id = reinterpret_cast<uint16_t*>(meta + 4 * kLevelsMax);     // [kWin] window-relative sample
  uint16_t* sel_qr = sel_id + kWin;                                          // [kWin] rank of its query
  uint16_t* selq = sel_qr + kWin;                                            // [kWin/4] window-relative query
  uint32_t* part_s = reinterpret_cast<uint32_t*>(selq + kWinQueries);        // [32] kept samples per piece
  uint32_t* part_q = part_s + kSelParts;                                     // [32] new queries per piece
  uint32_t* pre_s = part_q + kSelParts;                                      // [32] exclusive prefixes
  uint32_t* pre_q = pre_s + kSelParts;
  uint32_t* cs = pre_q + kSelParts;                                          // [6] chunk starts, [8..9] totals

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Dispatch order = cost order: the unit index is the slow axis and units are numbered from the LAST
  // level back (below), so the coarse levels' units -- three sort chunks each at the decoder shape against
  // one for a fine-level unit, 12-15 us against 6-7 -- start first instead of last (they used to start up
  // to 7 us into the kernel and set its end: 20 us for 8.5 us of mean work per workgroup).
  int rest, b, m;                                  // head <-> XCD map: gv_decode_block (msda_gv_common.h)
  if (kGvPair16 && sizeof(TV) == 2 && (d.M & 1) == 0) gv_decode_block<true>(blockIdx.x, d.M, d.B, rest, b, m);
  else gv_decode_block<false>(blockIdx.x, d.M, d.B, rest, b, m);
  const int unit = rest / d.B;
  VNX_STAMP(0);

  if (tid < d.L) {
    const int H = int(shapes[2 * tid]), W = int(shapes[2 * tid + 1]);
    const int n = H * W;
    const GvSplit sp = gv_level_split(n, units_min, kRowsMax);
    const int units = sp.units, rpu = sp.rpu;
    const int qs = gv_query_splits(units, d.Lq, P, sizeof(TV) == 4 || split_image != nullptr, d.B * d.M);
    meta[4 * tid] = H; meta[4 * tid + 1] = W; meta[4 * tid + 2] = int(lsi[tid]);
    meta[4 * tid + 3] = units | (rpu << 12) | (qs << 24);     // units <= 4000, rpu <= 320, qs <= 8
  }
  for (int i = tid; i < 2 * kRowsMax; i += kThreads) cnt2[i] = 0;
  if (tid == 0) alloc[0] = 0;
  __syncthreads();
  VNX_SEL_STAMP(1);

  int lvl = -1, r0 = 0, r1 = 0, Hl = 0, Wl = 0, start = 0, u_lvl = 0, qsplit = 1, qpiece = 0, lvl_units = 0;
  {
    int running = 0;
    bool packed = true;
    int u = unit;
    for (int l = 0; l < d.L; ++l) {
      packed = packed && (meta[4 * l + 2] == running);
      running += meta[4 * l] * meta[4 * l + 1];
    }
    for (int l = d.L - 1; l >= 0; --l) {       // units are numbered from the last (coarsest) level back
      const int H = meta[4 * l], W = meta[4 * l + 1], st = meta[4 * l + 2], ur = meta[4 * l + 3];
      const int n = H * W, units = ur & 0xfff, rpu = (ur >> 12) & 0xfff, qs = ur >> 24;
      if (lvl < 0) {
        if (u < units * qs) {        // a level's workgroups: row-unit major, query piece minor
          lvl = l; Hl = H; Wl = W; start = st; qsplit = qs; lvl_units = units;
          u_lvl = u / qs; qpiece = u - u_lvl * qs;
          r0 = u_lvl * rpu;
          r1 = r0 + rpu < n ? r0 + rpu : n;
        } else {
          u -= units * qs;
        }
      }
    }
    packed = packed && (running == d.S);
    if (!packed || lvl < 0) return;  // uniform over the workgroup
  }
  // the unit's geometry is uniform over the workgroup but came through LDS: say so, and it lives in SGPRs
  lvl = __builtin_amdgcn_readfirstlane(lvl); r0 = __builtin_amdgcn_readfirstlane(r0); r1 = __builtin_amdgcn_readfirstlane(r1);
  Hl = __builtin_amdgcn_readfirstlane(Hl); Wl = __builtin_amdgcn_readfirstlane(Wl); start = __builtin_amdgcn_readfirstlane(start);
  u_lvl = __builtin_amdgcn_readfirstlane(u_lvl); qsplit = __builtin_amdgcn_readfirstlane(qsplit);
  qpiece = __builtin_amdgcn_readfirstlane(qpiece);
  // A level of at most two units (the coarse ones: 240 and 60 pixels at 360p): more than half of ALL samples of the
  // level land in a unit, so selecting them buys nothing and costs the unit its longest latency chain (tag loads, ballots,
  // scan: 3.4 us of the 12-15 us a coarse unit of the T=5 decoder call takes -- the kernel's critical path).  Such a unit
  // takes every sample of a window in order: identity tables, no loads, no ballots, no scan.
  if (VNX_SEL_ABL == 4) return;
  const bool dense = __builtin_amdgcn_readfirstlane(lvl_units) <= 2;
  const int rows = r1 - r0;
  constexpr int kRpg = (kRowsMax + kGroups - 1) / kGroups;
  float4_t racc[kRpg];
#pragma unroll
  for (int k = 0; k < kRpg; ++k) racc[k] = float4_t{0.f, 0.f, 0.f, 0.f};

  const int n_samples = d.Lq * P;
  const int64_t level_base = ((int64_t(b) * d.M + m) * d.L + lvl) * n_samples;
  const uint4_t* my_recs = records + level_base;
  const uint32_t* my_uids = unit_ids + level_base;
  const TV* go_head = grad_out + (int64_t(b) * d.Lq * d.M + m) * D;
  // element strides as 24-bit multiplies (v_mul_u32_u24, full rate; the 64-bit products the compiler made of them were
  // three quarter-rate v_mul_lo / v_mul_hi each): queries and rows < 2^24, heads x 32 channels < 2^24, products < 2^32
  // (the launcher refuses larger calls)
  const uint32_t q_stride = uint32_t(d.M) * uint32_t(D);
  const uint4_t none = {0xffffffffu, 0u, 0u, 0u};
  // (g0 = tid, g1 = tid + kThreads: the two 16-B pieces of staged rows a thread moves; grp = tid >> 3, ch4 = tid & 7: the
  //  8-lane group and its 16-B channel piece -- all derived where used from an opaque copy of tid, see opaque())

  const int dr[4] = {0, 1, Wl, Wl + 1};

  int gchunk = 0;                                    // parity of the double-buffered row counters

  // the queries this workgroup takes: all of them, or one piece of a query-split level
  const int q_per = (d.Lq + qsplit - 1) / qsplit;
  const int s_lo = qpiece * q_per * P, s_hi = (qpiece + 1) * q_per * P < n_samples ? (qpiece + 1) * q_per * P : n_samples;
  for (int win0 = s_lo; win0 < s_hi; win0 += kWin) {
    const int n_w = s_hi - win0 < kWin ? s_hi - win0 : kWin;
    const int tw = opaque(tid);
    int n_sel, n_q;
    if (dense) {
#pragma unroll
      for (int r = 0; r < kSelRounds; ++r) {
        const int sidx = r * kThreads + tw;
        if (sidx < n_w) { sel_id[sidx] = uint16_t(sidx); sel_qr[sidx] = uint16_t(sidx >> 2); }
      }
      if (4 * tw < n_w) {              // kThreads == kWinQueries: one query of the window per thread
        selq[tw] = uint16_t(tw);
        if ((tw & (kQcMax - 1)) == 0) cs[tw / kQcMax] = uint32_t(4 * tw);
      }
      n_sel = n_w; n_q = (n_w + 3) >> 2;
    } else {
    // ---- selection: which samples of this window touch my rows -----------------------------
    unsigned long long bal[kSelRounds], balf[kSelRounds];
    uint32_t hitbits = 0;
#pragma unroll
    for (int r = 0; r < kSelRounds; ++r) {
      const int sidx = r * kThreads + tw;
      // NT: tags and records through `nt` loads when the whole grid is resident at once (the headline decoder call: 760
      // workgroups): every unit reads them at the same moment, once, on its own CU -- headline backward 27.3 -> 26.8 us.  On
      // grids of several rounds (decoder-720p, B = 10) the later units then miss in L2: 45 -> 52 us; plain loads there.
      const uint32_t v = sidx < n_w ? (NT ? __builtin_nontemporal_load(my_uids + win0 + sidx) : my_uids[win0 + sidx]) : 0xffffffffu;
      const bool hit = int(v & 0xffffu) <= u_lvl && u_lvl <= int(v >> 16) && v != 0xffffffffu;
      const unsigned long long bh = __ballot(hit);
      const uint32_t quad = uint32_t(bh >> (tw & 60)) & 0xfu;          // my query's four samples
      const bool first = hit && (quad & ((1u << (tw & 3)) - 1u)) == 0u;
      const unsigned long long bf = __ballot(first);
      bal[r] = bh; balf[r] = bf;
      hitbits |= uint32_t(hit) << r | uint32_t(first) << (8 + r);
      if (lane == 0) { part_s[r * kWaves + wave] = uint32_t(__popcll(bh)); part_q[r * kWaves + wave] = uint32_t(__popcll(bf)); }
    }
    if (win0 == 0) VNX_SEL_STAMP(2);
    __syncthreads();
    if (win0 == 0) VNX_SEL_STAMP(3);
    if (tid < 64) {                                   // one wave scans the 32 (round, wave) pieces
      const uint32_t ns = tid < kSelParts ? part_s[tid] : 0u, nq = tid < kSelParts ? part_q[tid] : 0u;
      const uint32_t is = wave_inclusive_scan(ns), iq = wave_inclusive_scan(nq);
      if (tid < kSelParts) { pre_s[tid] = is - ns; pre_q[tid] = iq - nq; }
      if (tid == kSelParts - 1) { cs[8] = is; cs[9] = iq; }
    }
    __syncthreads();
    n_sel = int(cs[8]); n_q = int(cs[9]);
    if (n_sel == 0) continue;                         // uniform: nothing of this window lands here
#pragma unroll
    for (int r = 0; r < kSelRounds; ++r) {
      if (hitbits & (1u << r)) {
        const uint32_t lo_mask_lo = __builtin_amdgcn_mbcnt_lo(uint32_t(bal[r]), 0u);
        const uint32_t pos = pre_s[r * kWaves + wave] + __builtin_amdgcn_mbcnt_hi(uint32_t(bal[r] >> 32), lo_mask_lo);
        const uint32_t qlo = __builtin_amdgcn_mbcnt_lo(uint32_t(balf[r]), 0u);
        const bool first = hitbits & (1u << (8 + r));
        // firsts at lanes <= mine: those strictly below, plus my own flag
        const uint32_t qr = pre_q[r * kWaves + wave] + __builtin_amdgcn_mbcnt_hi(uint32_t(balf[r] >> 32), qlo) +
                            (first ? 1u : 0u) - 1u;
        sel_id[pos] = uint16_t(r * kThreads + tw);
        sel_qr[pos] = uint16_t(qr);
        if (first) {
          selq[qr] = uint16_t((r * kThreads + tw) >> 2);
          if ((qr & uint32_t(kQcMax - 1)) == 0u) cs[qr / kQcMax] = pos;
        }
      }
    }
    }   // !dense
    const int n_chunks = VNX_SEL_ABL == 1 ? 0 : (n_q + kQcMax - 1) / kQcMax;
    if (tid == 0) cs[n_chunks] = uint32_t(n_sel);
    __syncthreads();
    if (win0 == 0) VNX_SEL_STAMP(4);

    // ---- chunks of <= 128 distinct queries over the kept samples -----------------------------
    const int q_win = win0 >> 2;
    uint4_t next = none;
    int next_slot = 0;
    float4_t pg0 = {0.f, 0.f, 0.f, 0.f}, pg1 = pg0;
    auto prefetch = [&](int c) {
      const int i = int(cs[c]) + tid;
      if (i < int(cs[c + 1])) {
        next = NT ? __builtin_nontemporal_load(my_recs + win0 + int(sel_id[i])) : my_recs[win0 + int(sel_id[i])];
        next_slot = int(sel_qr[i]) - c * kQcMax;
      } else {
        next = none;
      }
      const int g0 = opaque(tid), g1 = g0 + kThreads;
      const int qa = c * kQcMax + (g0 >> 3), qb = c * kQcMax + (g1 >> 3);
      if (qa < n_q) pg0 = load4<TV>(go_head + __umul24(uint32_t(q_win) + selq[qa], q_stride) + (g0 & 7) * 4);
      if (qb < n_q) pg1 = load4<TV>(go_head + __umul24(uint32_t(q_win) + selq[qb], q_stride) + (g1 & 7) * 4);
    };
    prefetch(0);
    for (int c = 0; c < n_chunks; ++c, ++gchunk) {
      uint32_t* cnt = cnt2 + (gchunk & 1) * kRowsMax;
      uint32_t* cnt_next = cnt2 + ((gchunk + 1) & 1) * kRowsMax;
      const uint4_t r = next;
      const uint32_t slot = uint32_t(next_slot);
      const int tc = opaque(tid);
      const int grp = tc >> 3, ch4 = tc & 7;
      grows[tc] = pg0;
      grows[tc + kThreads] = pg1;
      if (win0 == 0 && c == 0) VNX_SEL_STAMP(5);
      if (c + 1 < n_chunks) prefetch(c + 1);
      uint32_t mask = 0;
      int row00 = 0;
      float wt[4] = {0.f, 0.f, 0.f, 0.f};
      uint32_t rank[4] = {0u, 0u, 0u, 0u};
      if (r.x != 0xffffffffu) {
        int h0, w0;
        gv_unpack_corner(r.x, Hl, Wl, h0, w0);
        const float lh = __uint_as_float(r.y), lw = __uint_as_float(r.z), a = __uint_as_float(r.w);
        const float hh = 1.f - lh, hw = 1.f - lw;
        const bool top = h0 >= 0, bot = h0 + 1 <= Hl - 1, lef = w0 >= 0, rig = w0 + 1 <= Wl - 1;
        const int p00 = h0 * Wl + w0;
        mask = (uint32_t(top && lef && p00 >= r0 && p00 < r1)) |
               (uint32_t(top && rig && p00 + 1 >= r0 && p00 + 1 < r1) << 1) |
               (uint32_t(bot && lef && p00 + Wl >= r0 && p00 + Wl < r1) << 2) |
               (uint32_t(bot && rig && p00 + Wl + 1 >= r0 && p00 + Wl + 1 < r1) << 3);
        row00 = p00 - r0;
        wt[0] = a * (hh * hw); wt[1] = a * (hh * lw); wt[2] = a * (lh * hw); wt[3] = a * (lh * lw);
      }
      if (VNX_SEL_ABL == 2) mask = (wt[0] + wt[1] + wt[2] + wt[3] == 12345.f) ? mask : 0u;
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
      if (win0 == 0 && c == 0) VNX_SEL_STAMP(6);
      if (tid == 0) alloc[0] = 0;
      uint32_t rn[kRpg], ro[kRpg];
#pragma unroll
      for (int k = 0; k < kRpg; ++k) {
        const int row = grp + k * kGroups;
        rn[k] = row < rows ? cnt[row] : 0u;
        ro[k] = row < rows ? offs[row] : 0u;
        if (row < rows) cnt_next[row] = 0;
        if (VNX_SEL_ABL == 3) rn[k] = rn[k] == 0x7fffffffu ? 1u : 0u;
      }
      const float4_t* g4 = grows + ch4;
#pragma unroll
      for (int k = 0; k < kRpg; ++k) {
        const uint32_t n = rn[k];
        if (n == 0) continue;
        const uint2_t* seg = list + ro[k];
        float4_t a1 = {0.f, 0.f, 0.f, 0.f};
        uint32_t i = 0;
        // (Four entries per step instead of two -- half the dependent LDS round trips of a dense row: measured twice in
        // round 2, first with the 39 spills it caused (38.7 vs 31.3 us), then spill-free after the kernel had been
        // slimmed to 72 VGPRs: 30.0 vs 29.5 us at the decoder shape, 230 vs 229 at the encoder -- no gain either way.)
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
      if (win0 == 0 && c == 0) VNX_SEL_STAMP(7);
      __syncthreads();
      if (win0 == 0 && c == 0) VNX_SEL_STAMP(8);
    }
  }
  VNX_SEL_STAMP(11);

  TV* out = grad_value + ((int64_t(b) * d.S + start + r0) * d.M + m) * D;
  const int te = opaque(tid);
  const int grp = te >> 3, ch4 = te & 7;
  {
    if (qsplit > 1) {   // pieces of a query-split level meet through fp32 atomics (rows zeroed by the grad_loc kernel);
                        // a group's 8 lanes x 4 dwords = one row's 32 consecutive dwords per instruction group.
                        // 16-bit values: onto the fp32 split image, converted by split_levels_convert_kernel
      float* out32 = (sizeof(TV) == 4 ? reinterpret_cast<float*>(grad_value) : split_image) +
                     ((int64_t(b) * d.S + start + r0) * d.M + m) * D;
#pragma unroll
      for (int k = 0; k < kRpg; ++k) {
        const int row = grp + k * kGroups;
        if (row < rows) {
          float* p = out32 + __umul24(uint32_t(row), q_stride) + ch4 * 4;
          atomic_add(p, racc[k].x); atomic_add(p + 1, racc[k].y); atomic_add(p + 2, racc[k].z); atomic_add(p + 3, racc[k].w);
        }
      }
      VNX_STAMP(12);
      return;
    }
  }
#pragma unroll
  for (int k = 0; k < kRpg; ++k) {
    const int row = grp + k * kGroups;
    if (row < rows) store4<TV>(out + __umul24(uint32_t(row), q_stride) + ch4 * 4, racc[k]);
  }
  VNX_STAMP(12);
}

}  // namespace rec

int msda_gvrec_units_bound(const MsdaDims& d, int units_min, int rows_max) {
  // row-units of all levels, plus the extra pieces of the query-split levels (at most four row-units each)
  const int qs = gv_query_splits(1, d.Lq, d.P, true, d.B * d.M);
  return d.L * (units_min + 1) + (d.S + rows_max - 1) / rows_max + d.L * 4 * (qs - 1);
}

size_t msda_gvrec_record_bytes(const MsdaDims& d) {   // records + (aligned) unit ranges
  return gv_unit_ids_offset(d) + size_t(4) * size_t(d.B) * d.M * d.L * d.Lq * d.P;
}

bool msda_d32_gvrec_supported(int vdt, int ldt, const MsdaDims& d) {
  if (d.D != 32 || vdt == VNX_F64) return false;
  if (vdt == VNX_F32 && ldt != VNX_F32) return false;
  if (d.L > rec::kLevelsMax) return false;
  if (d.S > rec::kRowsMax * 4000) return false;     // units and rows/unit share one word
  if (int64_t(d.Lq) * d.P >= (int64_t(1) << 31)) return false;
  // 24-bit stride multiplies in the selection kernel: query index and heads x channels below 2^24, offsets below 2^32
  if (d.Lq >= (1 << 24) || d.M * 32 >= (1 << 24) || int64_t(d.Lq) * d.M * 32 >= (int64_t(1) << 32)) return false;
  const int64_t blocks = int64_t(d.B) * d.M * msda_gvrec_units_bound(d, 16, rec::kRowsMax);
  return blocks < (int64_t(1) << 31);
}

template <typename TV>
static int launch_gvrec(const int64_t* shapes, const int64_t* lsi, const void* records,
                        const void* grad_out, void* grad_value, const MsdaDims& d, int units_min,
                        int debug, int mode, float* split_image, hipStream_t stream) {
  // mode 0: per-unit sample selection (P == 4); 3: register slab, every unit scans its level;
  // 1: the LDS-slab form (variants 425 / 420 select the last two)
  const int units_bound = msda_gvrec_units_bound(d, units_min, rec::kRowsMax);
  const int64_t blocks = ((int64_t(d.B) * units_bound + 1) & ~int64_t(1)) * d.M;     // (unit, batch) pairs: even (gv_decode_block)
  if (mode == 0 && d.P == 4 && d.L <= rec::kLevelsMax) {
    const uint32_t* unit_ids = reinterpret_cast<const uint32_t*>((const char*)records + gv_unit_ids_offset(d));
    // one round of resident workgroups (3 per CU x 256 CUs = 768; the bound counts ~1.5x the real units)?
    if (blocks <= 1200)
      hipLaunchKernelGGL((rec::msda_bwd_gv_sel_kernel<TV, true>), dim3(uint32_t(blocks)), dim3(rec::kThreads),
                         rec::kSelLdsBytes, stream, shapes, lsi, (const rec::uint4_t*)records, unit_ids,
                         (const TV*)grad_out, (TV*)grad_value, d, units_min, split_image, debug);
    else
      hipLaunchKernelGGL((rec::msda_bwd_gv_sel_kernel<TV, false>), dim3(uint32_t(blocks)), dim3(rec::kThreads),
                         rec::kSelLdsBytes, stream, shapes, lsi, (const rec::uint4_t*)records, unit_ids,
                         (const TV*)grad_out, (TV*)grad_value, d, units_min, split_image, debug);
    return check_launch("msda_bwd_gv_sel");
  }
#define VNX_LAUNCH(PT, RS)                                                                             \
  hipLaunchKernelGGL((rec::msda_bwd_gv_rec_kernel<TV, PT, RS>), dim3(uint32_t(blocks)), dim3(rec::kThreads), \
                     rec::kLdsBytes - (RS ? size_t(rec::kRowsMax) * 128 : 0), stream, shapes, lsi,       \
                     (const rec::uint4_t*)records, (const TV*)grad_out, (TV*)grad_value, d, units_min,  \
                     units_bound, debug)
  if (mode != 1) { if (d.P == 4) VNX_LAUNCH(4, 3); else VNX_LAUNCH(0, 3); }
  else { if (d.P == 4) VNX_LAUNCH(4, 0); else VNX_LAUNCH(0, 0); }
#undef VNX_LAUNCH
  return check_launch("msda_bwd_gv_rec");
}

// grad_value from the sample records; a no-op on the device when the levels are not packed.
int msda_backward_gvrec_d32(int vdt, const int64_t* shapes, const int64_t* lsi, const void* records,
                            const void* grad_out, void* grad_value, MsdaDims d, int variant, float* split_image,
                            hipStream_t stream) {
  // every level is split into at least gv_units_min(d) units (2: 19 units per (b, head) at 360p = 760
  // workgroups <= the 768 resident at 3 per CU -- one round; 4: 960 workgroups, 39.2 vs 37.3 us)
  const int units_min = gv_units_min(d, false, kernel_variant());
  if (vdt == VNX_F32) return launch_gvrec<float>(shapes, lsi, records, grad_out, grad_value, d, units_min, (variant == 408 ? 1 : variant == 412 ? 5 : 0), (variant == 420 ? 1 : variant == 425 ? 3 : 0), split_image, stream);
  if (vdt == VNX_BF16) return launch_gvrec<bf16_t>(shapes, lsi, records, grad_out, grad_value, d, units_min, (variant == 408 ? 1 : variant == 412 ? 5 : 0), (variant == 420 ? 1 : variant == 425 ? 3 : 0), split_image, stream);
  if (vdt == VNX_F16) return launch_gvrec<f16_t>(shapes, lsi, records, grad_out, grad_value, d, units_min, (variant == 408 ? 1 : variant == 412 ? 5 : 0), (variant == 420 ? 1 : variant == 425 ? 3 : 0), split_image, stream);
  set_error("msda_backward_gvrec_d32: unsupported dtype %d", vdt);
  return VNX_ERR_INVALID_ARGUMENT;
}

// ---- 16-bit values: the rows of the query-split levels, accumulated in the fp32 split image, -> grad_value --------
// (fp atomics need an fp32 target; every other row of a 16-bit grad_value is written once by its owner.  Without the
//  split a coarse-level unit of a 16-bit encoder call sorted all 40 (360p) / 153 (720p) chunks of its level alone:
//  bf16 encoder backward 231 vs 191 us fp32 at 360p, 703 vs 339 us at 720p B = 2.)
template <typename TV>
__global__ void __launch_bounds__(256)
split_levels_convert_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                            const float* __restrict__ image, TV* __restrict__ grad_value, MsdaDims d, int units_min,
                            bool tiles) {
  if (!levels_packed(shapes, lsi, d.L, d.S)) return;
  // the split levels' pixel ranges, once per workgroup, and the running count of split pixels before each level: the
  // index space of the pass is ONLY those pixels (round 3 walked all B x S x M x 8 pieces and tested each: 11-14 us per
  // call for the 6 MB of rows it converts at 360p -- what made the bf16 encoder backward slower than the fp32 one)
  __shared__ int s_lo[rec::kLevelsMax], s_n[rec::kLevelsMax], s_before[rec::kLevelsMax + 1];
  if (int(threadIdx.x) < d.L) {
    const int l = threadIdx.x;
    const int st = int(lsi[l]), Hc = int(shapes[2 * l]), Wc = int(shapes[2 * l + 1]), n = Hc * Wc;
    const bool split = gv_query_splits(gv_level_units(Hc, Wc, units_min, tiles), d.Lq, d.P, true, d.B * d.M) > 1;
    s_lo[l] = st; s_n[l] = split ? n : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int l = 0; l < d.L; ++l) { s_before[l] = run; run += s_n[l]; }
    s_before[d.L] = run;
  }
  __syncthreads();
  const int split_rows = s_before[d.L];                  // split pixels per batch element
  const int row_pieces = d.M * 8;                        // 16-B pieces per pixel
  const int64_t per_batch = int64_t(split_rows) * row_pieces;
  const int64_t n4 = int64_t(d.B) * per_batch;           // pieces of 4 channels to convert
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += int64_t(gridDim.x) * blockDim.x) {
    const int b = int(i / per_batch);
    const int64_t in_b = i - int64_t(b) * per_batch;
    const int r = int(in_b / row_pieces), piece = int(in_b - int64_t(r) * row_pieces);
    int l = 0;
    while (l + 1 < d.L && r >= s_before[l + 1]) ++l;     // (levels without split pixels have s_before[l + 1] == s_before[l])
    const int64_t e = ((int64_t(b) * d.S + s_lo[l] + (r - s_before[l])) * row_pieces + piece) * 4;
    rec::store4<TV>(grad_value + e, *reinterpret_cast<const rec::float4_t*>(image + e));
  }
}

int msda_split_levels_convert(int vdt, const int64_t* shapes, const int64_t* lsi, const float* image, void* grad_value,
                              MsdaDims d, bool tiles, hipStream_t stream) {      // tiles: the grad_value path that ran
  const int units_min = gv_units_min(d, tiles, kernel_variant());
  const int64_t n4 = int64_t(d.B) * d.S * d.M * 8;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  if (vdt == VNX_BF16)
    hipLaunchKernelGGL((split_levels_convert_kernel<bf16_t>), dim3(uint32_t(blocks)), dim3(256), 0, stream, shapes, lsi, image,
                       (bf16_t*)grad_value, d, units_min, tiles);
  else if (vdt == VNX_F16)
    hipLaunchKernelGGL((split_levels_convert_kernel<f16_t>), dim3(uint32_t(blocks)), dim3(256), 0, stream, shapes, lsi, image,
                       (f16_t*)grad_value, d, units_min, tiles);
  else
    return VNX_OK;
  return check_launch("split_levels_convert");
}

// development build only (include/vnext_hip_dev.h): the phase stamps variants 408 / 412 leave
#ifdef VNX_DEV_VARIANTS
extern "C" int vnx_debug_read_rec_stamps(unsigned long long* host, int n) {
  return int(hipMemcpyFromSymbol(host, HIP_SYMBOL(rec::g_rec_stamps), sizeof(unsigned long long) * size_t(n)));
}
#endif

}  // namespace vnx
