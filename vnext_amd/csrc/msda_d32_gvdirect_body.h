// msda_d32_gvdirect_body.h -- the body of msda_bwd_gv_direct_kernel (see msda_d32_gvdirect.hip for what it does and why),
// as a device function so that two kernels can run it: the stand-alone grad_value kernel and the paired backward kernel of
// msda_d32.hip.
#pragma once
#include <type_traits>

#include "msda_gv_common.h"

namespace vnx {
namespace rec {

#ifndef VNX_GVD_UNITS_PER_CU
#define VNX_GVD_UNITS_PER_CU 3
#endif
// Channel parts a unit's rows are staged and walked in: 1 = whole 128-B rows (rounds 5's form: 80 KB of LDS, two units per
// CU), 2 = two passes of 16 channels over ONE sorted tap list (52 KB: three units per CU, see the walk below)
#ifndef VNX_GVD_PARTS
#define VNX_GVD_PARTS 2
#endif
#ifndef VNX_GVD_AUX
#define VNX_GVD_AUX 0             // cache policy of the location / weight loads (2 = nt)
#endif
// Timing ablations (A/B builds of the development library only; wrong grad_value by construction): 1 = return after the
// level table, 2 = no taps (loads, decode and barriers only), 3 = rows stored without their sums, 4 = no store
#ifndef VNX_GVD_ABL
#define VNX_GVD_ABL 0
#endif
#ifndef VNX_GVD_FULL_BARRIERS
#define VNX_GVD_FULL_BARRIERS 0   // A/B: __syncthreads() (waits for the rows in flight) where the kernel has LDS-only barriers
#endif


// Development aid (-DVNX_GVD_STAMPS, development library only): per-workgroup phase timestamps in constant-rate wall-clock
// ticks, 8 per workgroup -- 0 start, 1 level table read, 2 loads issued, 3 decoded + ranked (the loads have arrived),
// 4 first barrier, 5 offsets + second barrier, 6 scattered + rows staged + third barrier, 7 rows walked and stored.
#ifdef VNX_GVD_STAMPS
static __device__ unsigned long long g_gvd_stamps[4096 * 8];      // (per translation unit: the stand-alone kernel's is the one read back)
#define VNX_GVD_STAMP(k) do { if (tid == 0 && vblock < 4096) g_gvd_stamps[vblock * 8 + (k)] = (unsigned long long)wall_clock64(); } while (0)
#else
#define VNX_GVD_STAMP(k) do { } while (0)
#endif

constexpr int kGvdQc = VNX_GVD_QC;
constexpr int kGvdRows = VNX_GVD_ROWS;
constexpr int kGvdSamples = kGvdQc * 4;                                 // samples of a level per pass, at most
constexpr int kGvdSpt = (kGvdSamples + kThreads - 1) / kThreads;        // ... per thread
constexpr int kGvdCap = 4 * kGvdSamples;                                // taps per pass, at most: the sorted list holds them all
constexpr int kGvdParts = VNX_GVD_PARTS;                                // channel parts of a row (see VNX_GVD_PARTS)
constexpr int kGvdPartPieces = 8 / kGvdParts;                           // 16-B pieces (4 fp32 channels) of a row part
constexpr int kGvdRowPieces = (kGvdQc * kGvdPartPieces + kThreads - 1) / kThreads;   // pieces of a part's staged rows per thread and pass
// LDS of a unit: one part of the staged rows, the sorted tap list (weights + query slots), one word per row of the unit
// (taps | first tap << 16), the allocation word and the level table
constexpr size_t kGvdLdsBytes = size_t(kGvdQc) * 16 * kGvdPartPieces + size_t(kGvdCap) * 6 + size_t(kGvdRows) * 4 + 16 + 4 * kLevelsMax * 4;
static_assert(kGvdParts == 1 || kGvdParts == 2, "whole rows or halves");
static_assert(kGvdLdsBytes * VNX_GVD_UNITS_PER_CU <= 160 * 1024, "the units that share a CU must fit its LDS");
static_assert(kGvdCap < 65536 && kGvdQc < 65536, "tap ranks and query slots fit 16 bits");

// The kernel's body as a device function of the workgroup's index `vblock` and its LDS: msda_bwd_gv_direct_kernel below is this
// and nothing else; msda_bwd_pair_kernel (msda_d32.hip) runs it in the first workgroups of a grid whose other workgroups do the
// grad_loc kernel's work.
template <typename TV, typename TL, int P_T>
__device__ __forceinline__ void msda_bwd_gv_direct_body(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                                                        const TL* __restrict__ loc, const TL* __restrict__ attn,
                                                        const TV* __restrict__ grad_out, TV* __restrict__ grad_value,
                                                        const MsdaDims& d, int ut, int rows_max, int compact, unsigned long long* stamps,
                                                        uint32_t vblock, unsigned char* smem) {
  stamp_begin(stamps);
  constexpr int D = 32;
  float4_t* grows = reinterpret_cast<float4_t*>(smem);                       // [qc][8 / parts] grad_out rows of this head, one channel part
  float* l_wt = reinterpret_cast<float*>(grows + kGvdQc * kGvdPartPieces);   // [cap] tap weights, sorted by row
  uint32_t* cnt = reinterpret_cast<uint32_t*>(l_wt + kGvdCap);               // [rows] taps of the row (low half; < 2^16: kGvdCap), then | first tap << 16
  uint32_t* alloc = cnt + kGvdRows;                                          // [4]
  int* meta = reinterpret_cast<int*>(alloc + 4);                             // [4*L]
  uint16_t* l_slot = reinterpret_cast<uint16_t*>(meta + 4 * kLevelsMax);     // [cap] tap query slots, same order

  const int P = P_T > 0 ? P_T : d.P;
  const int tid = threadIdx.x, lane = tid & 63;
  // units are numbered from the last (coarsest) level back; head <-> XCD map rotating with the batch element: as in
  // msda_d32_gvrec.hip / msda_gv_common.h
  int rest, b, m;
  if (kGvPair16 && sizeof(TV) == 2 && (d.M & 1) == 0) gv_decode_block<true>(vblock, d.M, d.B, rest, b, m);
  else gv_decode_block<false>(vblock, d.M, d.B, rest, b, m);
  const int unit = rest / d.B;
  VNX_GVD_STAMP(0);

  // ---- the grad_out rows of this head, all queries of a pass: 16 B per thread and step, into registers, written to LDS just
  //      before the walk.  They are REQUESTED behind the pass's samples and stay in flight across the decode, the ranking and
  //      the two LDS-only barriers of the sort (lds_barrier).  The first forms of the round requested them first -- by LDS-DMA,
  //      before the level table -- and lost 1.3 us to three things the phase stamps showed: a wave's loads return in order,
  //      so nothing requested after the rows could be used before they had landed; __syncthreads() waits for every load in
  //      flight; and with an LDS-DMA pending the compiler makes every LDS access wait for it (it cannot tell the DMA's
  //      target from the access).  (The DMA form was kept for A/B runs through round 5; removed with the channel parts.)
  //      With two channel parts BOTH parts are requested here (the same 38 KB in flight); the second waits in registers
  //      while the first is walked. -----------------------------
  const TV* go_head = grad_out + (int64_t(b) * d.Lq * d.M + m) * D;
  const uint32_t q_stride = uint32_t(d.M) * uint32_t(D);
  constexpr uint32_t kOutOfRange = 0x80000000u;          // byte ranges stay below 2^31 (msda_d32_gvdirect_supported)
  const __amdgpu_buffer_rsrc_t go_src = uniform_rsrc(go_head, uint32_t(d.Lq) * q_stride * uint32_t(sizeof(TV)));
  int qc = kGvdSamples / P;          // queries per pass: what the staged rows and the sample slots of the threads hold
  qc = qc < kGvdQc ? qc : kGvdQc;
  float4_t pg[kGvdParts][kGvdRowPieces];
  auto request_rows = [&](int q_lo) {
    const int tg = opaque(tid);
#pragma unroll
    for (int part = 0; part < kGvdParts; ++part) {
#pragma unroll
      for (int i = 0; i < kGvdRowPieces; ++i) {
        const int g = i * kThreads + tg, ql = g / kGvdPartPieces;
        const bool ok = ql < qc && q_lo + ql < d.Lq;
        const uint32_t off = ok ? (__umul24(uint32_t(q_lo + ql), q_stride) + uint32_t(part * kGvdPartPieces + g % kGvdPartPieces) * 4u) * uint32_t(sizeof(TV))
                                : kOutOfRange;
        pg[part][i] = load4_buf<TV>(go_src, off);
      }
    }
  };
  auto stage_rows = [&](int part) {      // a part's pieces: registers -> LDS [query slot][piece]
    const int tg = opaque(tid);
#pragma unroll
    for (int i = 0; i < kGvdRowPieces; ++i)
      if ((i * kThreads + tg) / kGvdPartPieces < kGvdQc) grows[i * kThreads + tg] = pg[part][i];
  };
  int lvl = -1, r0 = 0, r1 = 0, Hl = 0, Wl = 0, start = 0, gshift = 0;
  bool packed = true;
  if (tid < d.L) {      // level table: lane l works out level l's unit split once
    // (Round 6, timing ablation: the table from compile-time constants instead of from memory -- what a host-side copy of the
    //  level sizes in the kernel arguments would save: 19.6-20.2 against 19.8 us at the T = 5 call, 33.2 against 33.2 at B = 10:
    //  nothing.  The kernel is bound by issue -- 65 % of the vector issue slots, profiles/r06_backward_pmc.csv -- not by this trip.)
    const int H = int(shapes[2 * tid]), W = int(shapes[2 * tid + 1]), first = int(lsi[tid]);     // (one round trip: both requested
    const GvdSplit sp = gvd_level_split(H * W, ut, d.Lq, P, rows_max);                                     //  before the divisions below)
    meta[4 * tid] = H; meta[4 * tid + 1] = W; meta[4 * tid + 2] = first;
    meta[4 * tid + 3] = sp.units | (sp.rpu << 18) | (sp.gshift << 29);      // units < 2^18 (S < 2^27), rpu <= 2047, gshift <= 3
  }
  for (int i = tid; i < kGvdRows; i += kThreads) cnt[i] = 0;
  if (tid == 0) alloc[0] = 0;
  lds_barrier();
  {
    int running = 0;
    int u = unit;
    for (int l = 0; l < d.L; ++l) {
      packed = packed && (meta[4 * l + 2] == running);
      running += meta[4 * l] * meta[4 * l + 1];
    }
    for (int l = d.L - 1; l >= 0; --l) {
      const int H = meta[4 * l], W = meta[4 * l + 1], st = meta[4 * l + 2];
      const uint32_t ur = uint32_t(meta[4 * l + 3]);
      const int n = H * W, units = int(ur & 0x3ffffu), rpu = int((ur >> 18) & 0x7ffu);
      if (lvl < 0) {
        if (u < units) {
          lvl = l; Hl = H; Wl = W; start = st; gshift = int(ur >> 29);
          r0 = u * rpu;
          r1 = r0 + rpu < n ? r0 + rpu : n;
        } else {
          u -= units;
        }
      }
    }
    packed = packed && (running == d.S);
  }
  if (!packed || lvl < 0) {        // uniform over the workgroup (unpacked levels: the general path does the call, capi.hip)
    return;
  }
  // uniform over the workgroup, but it came through LDS: scalarise (SGPRs instead of VGPRs, see opaque()).
  // (Tried: every lane reading the level sizes through the scalar cache and working the unit out in registers, no LDS -- the
  //  phase stamps showed why it buys nothing, 2.6 vs 2.3 us: the first barrier of the kernel waits for the row DMA requested
  //  above either way; what this phase costs is those 38 KB arriving, not the table.)
  lvl = __builtin_amdgcn_readfirstlane(lvl); r0 = __builtin_amdgcn_readfirstlane(r0); r1 = __builtin_amdgcn_readfirstlane(r1);
  Hl = __builtin_amdgcn_readfirstlane(Hl); Wl = __builtin_amdgcn_readfirstlane(Wl); start = __builtin_amdgcn_readfirstlane(start);
  gshift = __builtin_amdgcn_readfirstlane(gshift);
  if (VNX_GVD_ABL == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
  const int rows = r1 - r0;
  VNX_GVD_STAMP(1);

  const int LP = d.L * P;
  // sample (q, head m, level lvl, point k) of this batch element: element  q * (M * LP) + k  from s_base on in the op's own
  // layout; compact (the fused backward: its grad_loc kernel leaves the decoded locations and softmax weights laid out
  // [batch][head][level][query][point], msda_d32.hip): q * P + k -- consecutive bytes
  const int64_t s_base = compact ? ((int64_t(b) * d.M + m) * d.L + lvl) * int64_t(d.Lq) * P
                                 : (int64_t(b) * d.Lq * d.M + m) * LP + lvl * P;
  const TL* attn_bm = attn + s_base;
  const TL* loc_bm = loc + 2 * s_base;
  const uint32_t s_stride = compact ? uint32_t(P) : uint32_t(d.M) * uint32_t(LP);
  const uint32_t n_samp = uint32_t(d.Lq) * s_stride;      // from s_base on, at most
  const __amdgpu_buffer_rsrc_t loc_src = uniform_rsrc(loc_bm, n_samp * 2u * uint32_t(sizeof(TL)));
  const __amdgpu_buffer_rsrc_t attn_src = uniform_rsrc(attn_bm, n_samp * uint32_t(sizeof(TL)));
  const float Hf = float(Hl), Wf = float(Wl);
  const int dr[4] = {0, 1, Wl, Wl + 1};
  TV* out = grad_value + ((int64_t(b) * d.S + start + r0) * d.M + m) * D;

  const int n_pass = (d.Lq + qc - 1) / qc;
  const uint32_t gmask = (1u << gshift) - 1u;

  // One pass as a function of (pass, "the call has one pass"): the single-pass case -- every decoder call of the models -- is
  // compiled WITHOUT the loop around it: as a loop body the pass had 83 registers live across its sort (what the optimiser
  // hoists out of a loop stays in registers through all of it), 89 with the held sums below -- over the 80 that three
  // workgroups per CU allow.
  auto do_pass = [&](const int pass, auto single_tag) {
    constexpr bool kSingle = decltype(single_tag)::value;
    const int q_lo = pass * qc;
    // ---- this pass's samples: up to kGvdSpt per thread, slot j * 512 + tid -> (query slot, point) -------------------
    float sx[kGvdSpt], sy[kGvdSpt], sa[kGvdSpt];
    uint32_t svalid = 0;               // bit j: slot j holds a sample
    {
      const int tq = opaque(tid);
#pragma unroll
      for (int j = 0; j < kGvdSpt; ++j) {
        const uint32_t s = uint32_t(j * kThreads + tq);
        const uint32_t ql = P_T == 4 ? (s >> 2) : s / uint32_t(P);
        const uint32_t k = s - ql * uint32_t(P);
        const bool valid = int(ql) < qc && q_lo + int(ql) < d.Lq;
        svalid |= uint32_t(valid) << j;
        const uint32_t si = valid ? __umul24(uint32_t(q_lo) + ql, s_stride) + k : kOutOfRange;
        if constexpr (sizeof(TL) == 4) {
          const uint2_t xy = __builtin_bit_cast(uint2_t, __builtin_amdgcn_raw_buffer_load_b64(loc_src, int(si * 8u), 0, VNX_GVD_AUX));
          sx[j] = __uint_as_float(xy.x); sy[j] = __uint_as_float(xy.y);
          sa[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(attn_src, int(si * 4u), 0, VNX_GVD_AUX));
        } else {
          const uint32_t xy = __builtin_amdgcn_raw_buffer_load_b32(loc_src, int(si * 4u), 0, VNX_GVD_AUX);
          const uint32_t a16 = __builtin_amdgcn_raw_buffer_load_b16(attn_src, int(si * 2u), 0, VNX_GVD_AUX);
          sx[j] = to_acc(__builtin_bit_cast(TL, uint16_t(xy & 0xffffu))); sy[j] = to_acc(__builtin_bit_cast(TL, uint16_t(xy >> 16)));
          sa[j] = to_acc(__builtin_bit_cast(TL, uint16_t(a16)));
        }
      }
    }
    request_rows(q_lo);      // behind the samples: a wave's loads return in order, and the samples are needed first
    if (pass == 0) VNX_GVD_STAMP(2);

    // ---- decode (cuh:285-288, 38-45), rank every tap inside its row ------------------------------------------------------
    uint32_t mask = 0;                 // 4 bits per sample: which corners land in [r0, r1)
    int row00[kGvdSpt];
    uint32_t rank[kGvdSpt][2];         // two 16-bit ranks per word
#pragma unroll
    for (int j = 0; j < kGvdSpt; ++j) {
      row00[j] = 0; rank[j][0] = rank[j][1] = 0u;
      float lh = 0.f, lw = 0.f;
      uint32_t mj = 0;
      if (svalid & (1u << j)) {
        const float h = sy[j] * Hf - 0.5f, w = sx[j] * Wf - 0.5f;                 // cuh:285-286
        if (h > -1.f && w > -1.f && h < Hf && w < Wf) {                            // cuh:288
          const float hf = floorf(h), wf = floorf(w);
          const int h0 = int(hf), w0 = int(wf);
          lh = h - hf; lw = w - wf;
          const bool top = h0 >= 0, bot = h0 + 1 <= Hl - 1, lef = w0 >= 0, rig = w0 + 1 <= Wl - 1;
          const int p00 = h0 * Wl + w0;
          mj = (uint32_t(top && lef && p00 >= r0 && p00 < r1)) |
               (uint32_t(top && rig && p00 + 1 >= r0 && p00 + 1 < r1) << 1) |
               (uint32_t(bot && lef && p00 + Wl >= r0 && p00 + Wl < r1) << 2) |
               (uint32_t(bot && rig && p00 + Wl + 1 >= r0 && p00 + Wl + 1 < r1) << 3);
          row00[j] = p00 - r0;
        }
      }
      sy[j] = lh; sx[j] = lw;          // (sx, sy) now hold the fractions; sa the attention weight
      if (VNX_GVD_ABL == 2) mj = (lh + lw == 12345.f) ? mj : 0u;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (mj & (1u << t)) {
          const uint32_t r = __hip_atomic_fetch_add(cnt + row00[j] + dr[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          rank[j][t >> 1] |= r << ((t & 1) * 16);
        }
      mask |= mj << (4 * j);
    }
    if (pass == 0) VNX_GVD_STAMP(3);
    if (VNX_GVD_FULL_BARRIERS) __syncthreads(); else lds_barrier();      // (the rows stay in flight)
    if (pass == 0) VNX_GVD_STAMP(4);

    // ---- row counts -> segment offsets: DPP wave scan + one LDS allocation per wave and 512 rows ------------------------------
    for (int rb = 0; rb < rows; rb += kThreads) {      // uniform
      const int r = rb + tid;
      const uint32_t my_cnt = r < rows ? cnt[r] : 0u;      // (high half still zero: the ranks above were counted from it)
      const uint32_t incl = wave_inclusive_scan(my_cnt);
      const uint32_t wave_total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
      uint32_t base = 0;
      if (lane == 0 && wave_total != 0)
        base = __hip_atomic_fetch_add(alloc, wave_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
      if (r < rows) cnt[r] = my_cnt | ((base + incl - my_cnt) << 16);
    }
    if (VNX_GVD_FULL_BARRIERS) __syncthreads(); else lds_barrier();
    if (pass == 0) VNX_GVD_STAMP(5);

    // ---- scatter {query slot, weight} into the rows' segments --------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < kGvdSpt; ++j) {
      const uint32_t mj = (mask >> (4 * j)) & 0xfu;
      if (mj) {
        const float lh = sy[j], lw = sx[j], a = sa[j], hh = 1.f - lh, hw = 1.f - lw;
        const float wt[4] = {a * (hh * hw), a * (hh * lw), a * (lh * hw), a * (lh * lw)};     // cuh:115-152
        const uint32_t sj = uint32_t(j * kThreads + opaque(tid));
        const uint32_t slot = P_T == 4 ? (sj >> 2) : sj / uint32_t(P);       // the query slot of sample j (as in the loads)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (mj & (1u << t)) {
            const uint32_t pos = (cnt[row00[j] + dr[t]] >> 16) + ((rank[j][t >> 1] >> ((t & 1) * 16)) & 0xffffu);
            l_slot[pos] = uint16_t(slot);
            l_wt[pos] = wt[t];
          }
      }
    }
    stage_rows(0);
    __syncthreads();
    if (pass == 0) VNX_GVD_STAMP(6);
    if (tid == 0) alloc[0] = 0;

    // ---- lane groups walk the rows: slot = row * groups-per-row + sub; a row's segment summed in registers, stored at once.
    //      Round 5: a group is 4 lanes, each with 32 bytes of the row (bytes [16 j, 16 j + 16) and [64 + 16 j, ...): every load
    //      and store instruction moves whole 64-byte half rows): the walk is bound by the vector instructions of its PER-ROW
    //      work (a fine level has 1.25 taps per row), which all lanes of a group execute alike -- with 8 lanes x 16 bytes the
    //      kernel issued 999 vector instructions per wave, 450 of them here (profiles/r05_backward_pmc.csv); 4 lanes halved the
    //      row iterations of a wave: grad_value kernel 13.8 -> 13.4 us at the T = 5 decoder call, 25.4 -> 23.8 at B = 10,
    //      33.7 -> 32.5 at 720p (walk + store 4.5 -> 3.4 us per workgroup by the phase stamps).
    //      Round 6, two channel parts: the rows are staged and walked 16 channels at a time -- the SAME sorted list twice, the
    //      second part's pieces waiting in registers meanwhile -- so that a unit needs 52 KB of LDS instead of 80 and a CU holds
    //      three.  A group is then 2 lanes x 32 bytes of a 64-byte row part: a wave covers 32 rows per iteration, i.e. as many
    //      row iterations per wave over the two walks as one walk of whole rows. ----
    //      The first part's row sums WAIT IN REGISTERS (a group walks at most kIters rows) and leave with the second part's:
    //      stored when they were ready -- 64-byte halves of a cache line written microseconds apart -- the decoder-720p backward
    //      took 91 us against 36 (100 MB of rows; B = 10 at 360p 52.8 against 36.7): half lines do not merge on their way out.
    constexpr int kLpr = 4 / kGvdParts;                // lanes per row (part), two 16-B pieces each
    constexpr int kPieces = kGvdPartPieces / kLpr;
    constexpr int kGrp = kThreads / kLpr;              // groups per workgroup = slots per round
    // slots = rows << gshift <= max(kGvdRows, 2 * taps of a pass / 16)  (gvd_level_split: gshift > 0 only while taps >= (16 << gshift) * n / 2)
    constexpr int kSlotsMax = kGvdRows > kGvdCap / 8 ? kGvdRows : kGvdCap / 8;
    constexpr int kIters = (kSlotsMax + kGrp - 1) / kGrp;
    float4_t held[kGvdParts == 2 ? kIters : 1][kPieces];
    auto walk = [&](auto cp_tag) {
      constexpr int cp = decltype(cp_tag)::value;
      constexpr bool last = cp == kGvdParts - 1;
      const int ta = opaque(tid);
      const int grp = ta / kLpr, cl = ta % kLpr;
      const uint32_t step = gmask + 1u;
      // which 16-byte pieces of the row part a lane sums.  Whole rows: cl and cl + 4 (every instruction moves 64-byte half
      // rows).  Parts: the two groups of a lane QUAD (two rows) take their pieces in opposite order -- even group cl, cl + 2;
      // odd group cl + 2, cl -- so that ONE quad exchange of the second piece before the store leaves lanes 0..3 of the quad
      // with pieces 0..3 of the even group's row in one register and of the odd group's row in another: the stores then move
      // 64 contiguous bytes per quad, as with whole rows.  (Stored as they are summed -- 32 bytes per row and instruction,
      // four instructions per 128-byte line -- decoder-720p took 52 us against 35: without the stores this form is the faster,
      // 22.8 against 26.0 us.)
      const int gp = kGvdParts == 2 ? (grp & 1) : 0;
      const float4_t* g4[kPieces];
#pragma unroll
      for (int h = 0; h < kPieces; ++h) g4[h] = grows + cl + kLpr * (h ^ gp);
      const int n_slots = rows << gshift;
#pragma unroll
      for (int it = 0; it < kIters; ++it) {
        const int sb = it * kGrp;
        if (sb >= n_slots) break;                         // uniform
        const int slot = sb + grp;
        const int row = slot >> gshift;
        const uint32_t sub = uint32_t(slot) & gmask;      // the entries of the row this group takes: sub, sub + step, ...
        uint32_t n = 0, o = 0;
        if (slot < n_slots) {
          const uint32_t w = cnt[row];
          n = VNX_GVD_ABL == 3 ? 0u : (w & 0xffffu); o = w >> 16;
        }
        float4_t a0[kPieces];      // (one set of sums, two taps' rows in flight: a second set cost 8 registers the held sums need)
#pragma unroll
        for (int h = 0; h < kPieces; ++h) a0[h] = float4_t{0.f, 0.f, 0.f, 0.f};
        uint32_t i = sub;
        for (; i + step < n; i += 2 * step) {        // two taps in flight
          const uint32_t s0 = l_slot[o + i], s1 = l_slot[o + i + step];
          const float w0 = l_wt[o + i], w1 = l_wt[o + i + step];
          float4_t v0[kPieces], v1[kPieces];
#pragma unroll
          for (int h = 0; h < kPieces; ++h) { v0[h] = g4[h][s0 * kGvdPartPieces]; v1[h] = g4[h][s1 * kGvdPartPieces]; }
#pragma unroll
          for (int h = 0; h < kPieces; ++h) { a0[h] += w0 * v0[h]; a0[h] += w1 * v1[h]; }
        }
        if (i < n) {
          const uint32_t s0 = l_slot[o + i];
          const float w0 = l_wt[o + i];
#pragma unroll
          for (int h = 0; h < kPieces; ++h) a0[h] += w0 * g4[h][s0 * kGvdPartPieces];
        }
        // (two ROWS side by side, one tap of each per step -- measured slower: grad_value kernel 16.2 vs 13.5 us at the T = 5
        //  decoder call; the merged loop runs to the longer of the two rows)
        // a row spread over 1 << gshift groups (adjacent groups of one wave): every group ends with the row's sums.  (Parts: the
        // two groups of a quad hold their pieces in opposite order, so the first step adds the OTHER piece of the partner.)
#pragma unroll
        for (int sh = 0; sh < 3; ++sh)
          if (sh < gshift) {
            float4_t t[kPieces];
#pragma unroll
            for (int h = 0; h < kPieces; ++h) {
              const float4_t src = a0[(kGvdParts == 2 && sh == 0) ? (h ^ 1) : h];
              t[h].x = __shfl_xor(src.x, kLpr << sh, 64); t[h].y = __shfl_xor(src.y, kLpr << sh, 64);
              t[h].z = __shfl_xor(src.z, kLpr << sh, 64); t[h].w = __shfl_xor(src.w, kLpr << sh, 64);
            }
#pragma unroll
            for (int h = 0; h < kPieces; ++h) a0[h] += t[h];
          }
        if constexpr (!last) {
#pragma unroll
          for (int h = 0; h < kPieces; ++h) held[it][h] = a0[h];
          continue;
        }
        if (VNX_GVD_ABL == 4) continue;
        // several passes: the later ones add onto what the first stored (the same lanes wrote it: program order).  Non-temporal
        // stores in every case: written with plain stores the 26 MB of rows of a T = 5 call stay dirty in L2 until the
        // end-of-kernel write-back, which then takes 10 us (kernel 18.0 us; 7.8 without any store; 14.8 with `nt`) -- and a
        // branch that stores the same value plain on one side and `nt` on the other is merged by the compiler into the
        // plain form.  Other cache policies of the store (sc1 nt, sc0 sc1 nt: 13.3-13.4 us against 13.5; sc1, sc0 sc1
        // without nt: 14.4-14.5): within noise or worse, `nt` stays.
        if constexpr (kGvdParts == 2) {
          // (the row's word back to zero for the next pass's ranks: the groups that share the row read it in the same wave
          //  instruction above)
          if (!kSingle && sub == 0u && slot < n_slots) cnt[row] = 0u;
          const int slot_e = slot & ~1, slot_o = slot | 1;      // the quad's two groups
          const bool ok_e = (uint32_t(slot_e) & gmask) == 0u && slot_e < n_slots;
          const bool ok_o = (uint32_t(slot_o) & gmask) == 0u && slot_o < n_slots;
          const uint32_t lane_el = uint32_t(ta & 3) * 4u;
          TV* pe = out + __umul24(uint32_t(slot_e >> gshift), q_stride) + lane_el;
          TV* po = out + __umul24(uint32_t(slot_o >> gshift), q_stride) + lane_el;
          auto quad_rows = [&](const float4_t first, const float4_t second, float4_t& even_row, float4_t& odd_row) {
            // the partner group's second piece: quad_perm [2, 3, 0, 1].  (Inline assembly as the DPP sums of msda_d32.hip: written
            // with __builtin_amdgcn_update_dpp per component the compiler folded the moves into the selects below and all four
            // components came out as the first.)
            float sx, sy, sz, sw_;
            asm volatile(
                "s_nop 1\n\t"
                "v_mov_b32_dpp %0, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b32_dpp %1, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b32_dpp %2, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b32_dpp %3, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                : "=&v"(sx), "=&v"(sy), "=&v"(sz), "=&v"(sw_)
                : "v"(second.x), "v"(second.y), "v"(second.z), "v"(second.w));
            const float4_t sw = {sx, sy, sz, sw_};
            even_row = gp ? sw : first;
            odd_row = gp ? first : sw;
          };
          float4_t xe[2], xo[2];
          quad_rows(held[it][0], held[it][1], xe[0], xo[0]);
          quad_rows(a0[0], a0[1], xe[1], xo[1]);
          if (ok_e) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              if (!kSingle && pass > 0) xe[c] += load4<TV>(pe + c * 16);
              store4<TV>(pe + c * 16, xe[c]);
            }
          }
          if (ok_o) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              if (!kSingle && pass > 0) xo[c] += load4<TV>(po + c * 16);
              store4<TV>(po + c * 16, xo[c]);
            }
          }
        } else if (sub == 0u && slot < n_slots) {
          if (!kSingle) cnt[row] = 0u;
          TV* p = out + __umul24(uint32_t(row), q_stride) + cl * 4;
#pragma unroll
          for (int h = 0; h < kPieces; ++h) {
            if (!kSingle && pass > 0) a0[h] += load4<TV>(p + h * kLpr * 4);
            store4<TV>(p + h * kLpr * 4, a0[h]);
          }
        }
      }
    };
    walk(std::integral_constant<int, 0>{});
    if constexpr (kGvdParts == 2) {
      __syncthreads();      // every group is done with the first part's rows
      stage_rows(1);
      __syncthreads();
      walk(std::integral_constant<int, 1>{});
    }
    if (!kSingle && pass + 1 < n_pass) __syncthreads();       // the staged rows and the lists are rewritten next
  };
  if (n_pass == 1) {      // uniform
    do_pass(0, std::true_type{});
  } else {
    for (int pass = 0; pass < n_pass; ++pass) do_pass(pass, std::false_type{});
  }
  VNX_GVD_STAMP(7);
  stamp_end(stamps);
}


}  // namespace rec
}  // namespace vnx
