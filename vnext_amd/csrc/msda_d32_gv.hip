// msda_d32_gv.hip -- grad_value of multi-scale deformable attention without global
// atomics ("owner computes"), for 32-channel heads.
//
// Why: the reference scatters every tap with one fp32 atomicAdd per channel
// (ms_deform_im2col_cuda.cuh:125-152).  On MI355X global fp32 atomics retire at
// ~322 G dwords/s chip-wide whatever the address pattern or working set
// (tools/atomic_bench.hip; one dword per clock per L2 channel), which puts a floor
// of 76 us under the T=5 decoder call (24.6 M dword atomics) -- 6x its HBM time.
// LDS fp32 atomics are no way out either: ds_add_f32 measures ~170 clk per wave
// instruction (tools/lds_atomic_bench.hip), an order of magnitude slower than an LDS
// read + write pair.  So this kernel uses no floating-point atomics at all.
//
// Every element of grad_value has exactly one owner.  A workgroup ("unit") owns a
// contiguous range of pixels of ONE level for one (batch, head) and one HALF of the
// 32 channels, and keeps that slab [rows][16] in LDS (fp32).  Splitting the channels
// rather than the rows or the queries halves every LDS buffer without adding any
// floating-point work, so several units fit a CU and a decoder-sized call (300
// queries) is one pass with no chunk loop.  Per chunk of queries:
//   * the chunk's grad_out half-rows of this head go to LDS once (16-B loads), and
//     each thread computes the bilinear geometry of up to three samples of the
//     unit's level;
//   * the taps that land inside the unit's range are counting-sorted by destination
//     row: an integer LDS atomic gives each tap its rank inside its row, a wave scan
//     plus one atomic per wave turns the row counts into segment offsets, and each
//     tap record {query slot, weight*attn} is written to its slot (in windows of
//     kGvCap records when a chunk has more taps than the list holds);
//   * 4-lane groups (16 B per lane = the 16 channels) then own rows: a row's segment
//     is summed in registers -- the only serial chain is the FMA -- and added to the
//     slab row once;
//   * the next chunk's global loads are issued before the sort, so their latency
//     hides behind the LDS work.
// The slab is finally written once with 16-B stores: no zero-fill pass, no global
// atomic, no fp32 workspace / convert pass for 16-bit tensors, and nothing depends
// on the order workgroups run in.
//
// The unit table is derived ON DEVICE from spatial_shapes / level_start_index (the
// ABI hands these over as device tensors, ms_deform_attn_cuda.cu:67-68): each level
// is cut into max(units_min, ceil(n_l / ROWS_MAX)) ranges, so every level -- which
// receives the same number of samples -- gets at least units_min workgroups.  The
// scheme needs the levels packed back to back (level_start_index[l] == sum of the
// previous H*W and their total == spatial_size), which is how the reference builds
// them (deformable_transformer.py:97-106).  Every workgroup re-derives that
// predicate; when it fails the kernel does nothing and the general path (atomics
// into a zero-filled image) does the work -- see vnx_msda_backward in capi.hip.
#include "vnx_common.h"

namespace vnx {

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));

template <typename TV>
__device__ __forceinline__ float4_t gv_load4(const TV* p);
template <>
__device__ __forceinline__ float4_t gv_load4<float>(const float* p) {
  return *reinterpret_cast<const float4_t*>(p);
}
template <>
__device__ __forceinline__ float4_t gv_load4<bf16_t>(const bf16_t* p) {
  const uint2_t r = *reinterpret_cast<const uint2_t*>(p);
  float4_t v;
  v.x = __uint_as_float(r.x << 16); v.y = __uint_as_float(r.x & 0xffff0000u);
  v.z = __uint_as_float(r.y << 16); v.w = __uint_as_float(r.y & 0xffff0000u);
  return v;
}
template <>
__device__ __forceinline__ float4_t gv_load4<f16_t>(const f16_t* p) {
  const uint2_t r = *reinterpret_cast<const uint2_t*>(p);
  float4_t v;
  v.x = float(__builtin_bit_cast(_Float16, uint16_t(r.x & 0xffffu)));
  v.y = float(__builtin_bit_cast(_Float16, uint16_t(r.x >> 16)));
  v.z = float(__builtin_bit_cast(_Float16, uint16_t(r.y & 0xffffu)));
  v.w = float(__builtin_bit_cast(_Float16, uint16_t(r.y >> 16)));
  return v;
}

template <typename TV>
__device__ __forceinline__ void gv_store4(TV* p, float4_t v);
template <>
__device__ __forceinline__ void gv_store4<float>(float* p, float4_t v) {
  *reinterpret_cast<float4_t*>(p) = v;
}
template <>
__device__ __forceinline__ void gv_store4<bf16_t>(bf16_t* p, float4_t v) {
  uint2_t r;
  r.x = uint32_t(f32_to_bf16_bits(v.x)) | (uint32_t(f32_to_bf16_bits(v.y)) << 16);
  r.y = uint32_t(f32_to_bf16_bits(v.z)) | (uint32_t(f32_to_bf16_bits(v.w)) << 16);
  *reinterpret_cast<uint2_t*>(p) = r;
}
template <>
__device__ __forceinline__ void gv_store4<f16_t>(f16_t* p, float4_t v) {
  uint2_t r;
  r.x = uint32_t(__builtin_bit_cast(uint16_t, _Float16(v.x))) | (uint32_t(__builtin_bit_cast(uint16_t, _Float16(v.y))) << 16);
  r.y = uint32_t(__builtin_bit_cast(uint16_t, _Float16(v.z))) | (uint32_t(__builtin_bit_cast(uint16_t, _Float16(v.w))) << 16);
  *reinterpret_cast<uint2_t*>(p) = r;
}

constexpr int kGvWaves = 8;                 // 512 threads
constexpr int kGvThreads = 64 * kGvWaves;
constexpr int kGvHalf = 16;                 // channels per unit
constexpr int kGvGroups = kGvThreads / 4;   // 4-lane groups, 16 B per lane
constexpr int kGvSpt = 3;                   // samples per thread per chunk
constexpr int kGvRowsMax = 320;             // 20 KiB slab
constexpr int kGvRpg = (kGvRowsMax + kGvGroups - 1) / kGvGroups;  // rows per 4-lane group
constexpr int kGvQcMax = 304;               // queries per chunk (19 KiB of grad_out half-rows)
constexpr int kGvCap = 1280;                // tap records per window (10 KiB)
constexpr int kGvLevelsMax = 64;
// 20480 + 19456 + 10240 + 2*1280 (counters, offsets) + 16 + 1024 (level table) = 53776 B
// -> three units per CU by LDS (160 KiB / 3 = 54613 B).
constexpr size_t kGvLdsBytes = size_t(kGvRowsMax) * kGvHalf * 4 + size_t(kGvQcMax) * kGvHalf * 4 +
                               size_t(kGvCap) * 8 + size_t(kGvRowsMax) * 8 + 16 + 4 * kGvLevelsMax * 4;

// Development aid: per-workgroup phase timestamps (s_memtime), written when ablate & 8.
__device__ unsigned long long g_gv_stamps[4096 * 16];
#define VNX_STAMP(k)                                                         \
  do {                                                                       \
    if ((ablate & 8) && tid == 0 && blockIdx.x < 4096)                       \
      g_gv_stamps[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter();     \
  } while (0)

struct GvGeom {
  uint32_t mask;  // taps inside the unit's range
  int row00;      // first tap's row relative to the unit
  float w[4];
};

// Bilinear geometry of one sample against the unit's pixel range [r0, r1) of a Hl x Wl level
// (ms_deform_im2col_cuda.cuh:285-288 range test, :38-78 taps).
__device__ __forceinline__ GvGeom gv_geometry(float x, float y, float a, int Hl, int Wl, int r0, int r1) {
  GvGeom g;
  g.mask = 0; g.row00 = 0;
  g.w[0] = g.w[1] = g.w[2] = g.w[3] = 0.f;
  const float Hf = float(Hl), Wf = float(Wl);
  const float h = y * Hf - 0.5f, w = x * Wf - 0.5f;
  if (h > -1.f && w > -1.f && h < Hf && w < Wf) {
    const float hf = floorf(h), wf = floorf(w);
    const int h0 = int(hf), w0 = int(wf);
    const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw = 1.f - lw;
    const bool top = h0 >= 0, bot = h0 + 1 <= Hl - 1, lef = w0 >= 0, rig = w0 + 1 <= Wl - 1;
    const int p00 = h0 * Wl + w0;
    const int pa = p00, pb = p00 + 1, pc = p00 + Wl, pd = p00 + Wl + 1;
    g.mask = (uint32_t(top && lef && pa >= r0 && pa < r1)) |
             (uint32_t(top && rig && pb >= r0 && pb < r1) << 1) |
             (uint32_t(bot && lef && pc >= r0 && pc < r1) << 2) |
             (uint32_t(bot && rig && pd >= r0 && pd < r1) << 3);
    g.row00 = p00 - r0;
    g.w[0] = a * (hh * hw); g.w[1] = a * (hh * lw); g.w[2] = a * (lh * hw); g.w[3] = a * (lh * lw);
  }
  return g;
}

// P_T > 0: points per level known at compile time (index arithmetic without integer division).
template <typename TV, typename TL, int P_T>
__global__ void __launch_bounds__(kGvThreads, 4)
msda_bwd_gv_tile_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                        const TL* __restrict__ loc, const TL* __restrict__ attn,
                        const TV* __restrict__ grad_out, TV* __restrict__ grad_value, MsdaDims d,
                        int units_min, int units_bound, int qc, int ablate) {
  constexpr int D = 32;
  constexpr int H4 = kGvHalf / 4;  // float4 per half-row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4_t* slab = reinterpret_cast<float4_t*>(smem);                    // [rows][H4]
  float4_t* grows = slab + kGvRowsMax * H4;                              // [qc][H4]
  uint2_t* list = reinterpret_cast<uint2_t*>(grows + kGvQcMax * H4);     // [cap] tap records
  uint32_t* cnt = reinterpret_cast<uint32_t*>(list + kGvCap);            // [rows] taps per row
  uint32_t* offs = cnt + kGvRowsMax;                                     // [rows] segment starts
  uint32_t* alloc = offs + kGvRowsMax;                                   // [4] running total
  int* meta = reinterpret_cast<int*>(alloc + 4);                         // [4*L] W, start, units, rows/unit

  const int P = P_T > 0 ? P_T : d.P;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // blockIdx -> (channel half, head, unit, batch); head innermost but one for XCD affinity
  const int half = blockIdx.x & 1;
  const int m = (blockIdx.x >> 1) % d.M;
  const int rest = (blockIdx.x >> 1) / d.M;
  const int unit = rest % units_bound;
  const int b = rest / units_bound;
  VNX_STAMP(0);

  // ---- level table: lane l works out level l's unit split; shared through LDS --------------
  // meta[4l..] = {W_l | H_l << 16, start_l (or -1 when not packed), units_l, rows_per_unit_l}
  if (tid < d.L) {
    const int H = int(shapes[2 * tid]), W = int(shapes[2 * tid + 1]);
    const int st = int(lsi[tid]);
    const int n = H * W;
    int units = 0, rpu = 1;
    if (n > 0) {
      units = (n + kGvRowsMax - 1) / kGvRowsMax;
      if (units < units_min) units = units_min;
      if (units > n) units = n;
      rpu = (n + units - 1) / units;
      units = (n + rpu - 1) / rpu;
    }
    meta[4 * tid] = H; meta[4 * tid + 1] = W; meta[4 * tid + 2] = st;
    meta[4 * tid + 3] = units | (rpu << 12);  // units <= 2^12 is checked on the host
  }
  for (int i = tid; i < kGvRowsMax; i += kGvThreads) cnt[i] = 0;
  if (tid == 0) alloc[0] = 0;
  __syncthreads();
  VNX_STAMP(1);

  // ---- which (level, pixel range) is this unit? (no divisions here) ---------------------------
  int lvl = -1, r0 = 0, r1 = 0, Hl = 0, Wl = 0, start = 0;
  {
    int running = 0;
    bool packed = true;
    int u = unit;
    for (int l = 0; l < d.L; ++l) {
      const int H = meta[4 * l], W = meta[4 * l + 1], st = meta[4 * l + 2], ur = meta[4 * l + 3];
      const int n = H * W, units = ur & 0xfff, rpu = ur >> 12;
      packed = packed && (st == running);
      running += n;
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
  const int rows = r1 - r0;
  for (int i = tid; i < rows * H4; i += kGvThreads) slab[i] = float4_t{0.f, 0.f, 0.f, 0.f};

  const int LP = d.L * P;
  const int n_chunks = (d.Lq + qc - 1) / qc;
  const int n_samples = qc * P;  // per full chunk, <= kGvSpt * kGvThreads
  const int64_t q_stride = int64_t(d.M) * D;  // grad_out elements between queries
  const TV* go_head = grad_out + (int64_t(b) * d.Lq * d.M + m) * D + half * kGvHalf;
  const int64_t loc_head = (int64_t(b) * d.Lq * d.M + m) * LP + lvl * P;  // + q*M*LP + k

  // this thread's slots: float4 pieces of the chunk's [qc][H4] half-rows, and samples
  float4_t pg[kGvSpt];
  float px[kGvSpt], py[kGvSpt], pa[kGvSpt];
  auto prefetch = [&](int chunk) {
    const int q_base = chunk * qc;
#pragma unroll
    for (int j = 0; j < kGvSpt; ++j) {
      const int slot = tid + j * kGvThreads;
      const int q = q_base + (slot >> 2);
      pg[j] = float4_t{0.f, 0.f, 0.f, 0.f};
      if ((slot >> 2) < qc && q < d.Lq) pg[j] = gv_load4<TV>(go_head + int64_t(q) * q_stride + (slot & 3) * 4);
    }
#pragma unroll
    for (int j = 0; j < kGvSpt; ++j) {
      const int e = tid + j * kGvThreads;
      const int qs = e / P;
      const int q = q_base + qs;
      px[j] = -4.f; py[j] = -4.f; pa[j] = 0.f;  // fails the range test
      if (e < n_samples && q < d.Lq) {
        const int64_t wi = loc_head + int64_t(q) * d.M * LP + (e - qs * P);
        px[j] = to_acc(loc[2 * wi]); py[j] = to_acc(loc[2 * wi + 1]); pa[j] = to_acc(attn[wi]);
      }
    }
  };
  prefetch(0);
  VNX_STAMP(2);

  const int grp = tid >> 2, c4 = tid & 3;
  const int dr[4] = {0, 1, Wl, Wl + 1};
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    if (chunk == 0) VNX_STAMP(3);
    // ---- stage the chunk: grad_out half-rows -> LDS; geometry; rank each tap in its row -------
#pragma unroll
    for (int j = 0; j < kGvSpt; ++j) {
      const int slot = tid + j * kGvThreads;
      if ((slot >> 2) < qc) grows[slot] = pg[j];
    }
    // kept across the sort: the sample itself (12 B) and 4 x 16-bit ranks; the geometry is
    // recomputed at scatter time, which is cheaper than carrying it in registers
    float sx[kGvSpt], sy[kGvSpt], sa[kGvSpt];
    uint32_t rk01[kGvSpt], rk23[kGvSpt];
#pragma unroll
    for (int j = 0; j < kGvSpt; ++j) {
      sx[j] = px[j]; sy[j] = py[j]; sa[j] = pa[j];
      GvGeom g = gv_geometry(sx[j], sy[j], sa[j], Hl, Wl, r0, r1);
      if (ablate & 2) g.mask = 0;
      uint32_t rk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (g.mask & (1u << t))  // integer LDS atomics are fast (tools/lds_atomic_bench.hip)
          rk[t] = __hip_atomic_fetch_add(cnt + g.row00 + dr[t], 1u, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP);
      rk01[j] = rk[0] | (rk[1] << 16);
      rk23[j] = rk[2] | (rk[3] << 16);
    }
    if (chunk == 0) VNX_STAMP(8);
    // the next chunk's loads go out now; they land while this chunk is sorted and applied
    if (chunk + 1 < n_chunks) prefetch(chunk + 1);
    __syncthreads();
    if (chunk == 0) VNX_STAMP(9);

    // ---- row counts -> segment offsets: wave scan + one allocation per wave --------------------
    {
      const uint32_t my_cnt = tid < rows ? cnt[tid] : 0u;
      uint32_t incl = my_cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
      }
      const uint32_t wave_total = __shfl(incl, 63, 64);
      uint32_t base = 0;
      if (lane == 0 && wave_total != 0)
        base = __hip_atomic_fetch_add(alloc, wave_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
      if (tid < rows) offs[tid] = base + incl - my_cnt;
    }
    __syncthreads();
    if (chunk == 0) VNX_STAMP(10);
    const uint32_t total = alloc[0];

    for (uint32_t win = 0; win < total; win += kGvCap) {  // one window unless the chunk is tap-heavy
      // ---- scatter the taps of this window into their row segments ------------------------------
#pragma unroll
      for (int j = 0; j < kGvSpt; ++j) {
        GvGeom g = gv_geometry(sx[j], sy[j], sa[j], Hl, Wl, r0, r1);
        if (ablate & 2) g.mask = 0;
        const uint32_t qs = uint32_t((tid + j * kGvThreads) / P);
        const uint32_t rk[4] = {rk01[j] & 0xffffu, rk01[j] >> 16, rk23[j] & 0xffffu, rk23[j] >> 16};
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (g.mask & (1u << t)) {
            const uint32_t pos = offs[g.row00 + dr[t]] + rk[t] - win;
            if (pos < uint32_t(kGvCap)) list[pos] = uint2_t{qs, __float_as_uint(g.w[t])};
          }
      }
      __syncthreads();
      if (chunk == 0 && win == 0) VNX_STAMP(11);
      // ---- 4-lane groups own rows: sum the row's segment in registers, one slab update --------
      if (!(ablate & 1)) {
        uint32_t rn[kGvRpg], ro[kGvRpg];
#pragma unroll
        for (int k = 0; k < kGvRpg; ++k) {  // this group's rows: counts and offsets up front
          const int row = grp + k * kGvGroups;
          rn[k] = row < rows ? cnt[row] : 0u;
          ro[k] = row < rows ? offs[row] : 0u;
        }
#pragma unroll
        for (int k = 0; k < kGvRpg; ++k) {
          const int row = grp + k * kGvGroups;
          const uint32_t n = rn[k], o = ro[k];
          // the part of [o, o+n) inside [win, win+cap)
          const uint32_t lo = o > win ? o : win;
          const uint32_t hi = (o + n) < (win + kGvCap) ? (o + n) : (win + kGvCap);
          if (n == 0 || lo >= hi) continue;
          const uint2_t* seg = list + (lo - win);
          const uint32_t len = hi - lo;
          const float4_t* g4 = grows + c4;
          float4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
          uint32_t i = 0;
          for (; i + 4 <= len; i += 4) {  // four independent record -> row-read chains in flight
            const uint2_t e0 = seg[i], e1 = seg[i + 1], e2 = seg[i + 2], e3 = seg[i + 3];
            const float4_t x0 = g4[e0.x * H4], x1 = g4[e1.x * H4], x2 = g4[e2.x * H4], x3 = g4[e3.x * H4];
            a0 += __uint_as_float(e0.y) * x0;
            a1 += __uint_as_float(e1.y) * x1;
            a2 += __uint_as_float(e2.y) * x2;
            a3 += __uint_as_float(e3.y) * x3;
          }
          for (; i < len; ++i) {
            const uint2_t e = seg[i];
            a0 += __uint_as_float(e.y) * g4[e.x * H4];
          }
          slab[row * H4 + c4] += (a0 + a1) + (a2 + a3);
        }
      }
      __syncthreads();
      if (chunk == 0 && win == 0) VNX_STAMP(12);
    }
    if (chunk + 1 < n_chunks) {
      // reset the counters for the next chunk; the barrier orders its rank atomics after this
      if (tid < rows) cnt[tid] = 0;
      if (tid == 0) alloc[0] = 0;
      __syncthreads();
    }
  }
  VNX_STAMP(6);

  // ---- write the slab: one owner per element, 16 B per lane, 64-B half lines -----------------
  TV* out = grad_value + ((int64_t(b) * d.S + start + r0) * d.M + m) * D + half * kGvHalf;
  for (int i = tid; i < rows * H4; i += kGvThreads) {
    const int row = i >> 2, k4 = i & 3;
    gv_store4<TV>(out + int64_t(row) * d.M * D + k4 * 4, slab[i]);
  }
  VNX_STAMP(7);
}

int msda_gv_units_bound(const MsdaDims& d, int units_min) {
  return d.L * (units_min + 1) + (d.S + kGvRowsMax - 1) / kGvRowsMax;
}

bool msda_d32_gv_supported(int vdt, int ldt, const MsdaDims& d) {
  if (d.D != 32 || vdt == VNX_F64) return false;
  if (vdt == VNX_F32 && ldt != VNX_F32) return false;
  if (d.P > 64 || d.L > kGvLevelsMax) return false;  // >= 8 queries x P samples per chunk; level table in LDS
  if (d.S > kGvRowsMax * 4000) return false;         // units per level and rows per unit pack into one word
  const int64_t blocks = int64_t(d.B) * d.M * msda_gv_units_bound(d, 16) * 2;
  return blocks < (int64_t(1) << 31);
}

template <typename TV, typename TL>
static int launch_gv(const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                     const void* grad_out, void* grad_value, const MsdaDims& d, int units_min,
                     int ablate, hipStream_t stream) {
  const int units_bound = msda_gv_units_bound(d, units_min);
  const int64_t blocks = int64_t(d.B) * d.M * units_bound * 2;
  int qc = (kGvSpt * kGvThreads) / d.P;
  if (qc > kGvQcMax) qc = kGvQcMax;
#define VNX_LAUNCH(PT)                                                                             \
  hipLaunchKernelGGL((msda_bwd_gv_tile_kernel<TV, TL, PT>), dim3(uint32_t(blocks)),                \
                     dim3(kGvThreads), kGvLdsBytes, stream, shapes, lsi, (const TL*)loc,          \
                     (const TL*)attn, (const TV*)grad_out, (TV*)grad_value, d, units_min,          \
                     units_bound, qc, ablate)
  if (d.P == 4) VNX_LAUNCH(4); else VNX_LAUNCH(0);
#undef VNX_LAUNCH
  return check_launch("msda_bwd_gv_tile");
}

// grad_value for packed levels; a no-op on the device when the levels are not packed.
int msda_backward_gv_d32(int vdt, int ldt, const int64_t* shapes, const int64_t* lsi,
                         const void* loc, const void* attn, const void* grad_out, void* grad_value,
                         MsdaDims d, int variant, hipStream_t stream) {
  // every level receives Lq*P samples; give each at least this many owners
  int units_min = 4;
  if (variant >= 200 && variant < 300) units_min = variant - 200;
  if (units_min < 1) units_min = 1;
  if (units_min > 16) units_min = 16;
  // timing ablations only: 401 no accumulation, 402 no taps, 408 phase timestamps
  const int ablate = (variant == 401) ? 1 : (variant == 402) ? 2 : (variant == 408) ? 8 : 0;
#define VNX_ARGS shapes, lsi, loc, attn, grad_out, grad_value, d, units_min, ablate, stream
  if (vdt == VNX_F32) return launch_gv<float, float>(VNX_ARGS);
  if (vdt == VNX_BF16 && ldt == VNX_F32) return launch_gv<bf16_t, float>(VNX_ARGS);
  if (vdt == VNX_BF16 && ldt == VNX_BF16) return launch_gv<bf16_t, bf16_t>(VNX_ARGS);
  if (vdt == VNX_F16 && ldt == VNX_F32) return launch_gv<f16_t, float>(VNX_ARGS);
  if (vdt == VNX_F16 && ldt == VNX_F16) return launch_gv<f16_t, f16_t>(VNX_ARGS);
#undef VNX_ARGS
  set_error("msda_backward_gv_d32: unsupported dtype pair (%d, %d)", vdt, ldt);
  return VNX_ERR_INVALID_ARGUMENT;
}

// ---- helpers for the general (not packed) path: run only when NOT packed ---------------------
__global__ void __launch_bounds__(256)
zero_if_not_packed_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi, int L,
                          int S, float4_t* __restrict__ dst, int64_t n16, unsigned char* tail,
                          int tail_bytes) {
  if (levels_packed(shapes, lsi, L, S)) return;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16;
       i += int64_t(gridDim.x) * blockDim.x)
    dst[i] = float4_t{0.f, 0.f, 0.f, 0.f};
  if (blockIdx.x == 0 && int(threadIdx.x) < tail_bytes) tail[threadIdx.x] = 0;
}

int zero_if_not_packed(const int64_t* shapes, const int64_t* lsi, int L, int S, void* dst,
                       size_t bytes, hipStream_t stream) {
  const int64_t n16 = int64_t(bytes / 16);
  int64_t blocks = (n16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_if_not_packed_kernel, dim3(uint32_t(blocks)), dim3(256), 0, stream, shapes,
                     lsi, L, S, (float4_t*)dst, n16, (unsigned char*)dst + n16 * 16, int(bytes % 16));
  return check_launch("zero_if_not_packed");
}

// development aid, not part of the public header
extern "C" int vnx_debug_read_gv_stamps(unsigned long long* host, int n) {
  return int(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gv_stamps), sizeof(unsigned long long) * size_t(n)));
}

}  // namespace vnx
