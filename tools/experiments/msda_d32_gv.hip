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
// Every row of grad_value has exactly one owner.  A workgroup ("unit") owns a
// contiguous range of pixels of ONE level for one (batch, head) and keeps that slab
// [rows][32] in LDS (fp32).  It walks the queries in chunks:
//   * the chunk's grad_out rows of this head go to LDS once (coalesced 16-B loads),
//     and one thread per sample of the unit's level computes the bilinear geometry;
//   * the taps that land inside the unit's range are counting-sorted by destination
//     row: an integer LDS atomic gives each tap its rank inside its row, a block scan
//     turns the row counts into segment offsets, each tap record {query slot,
//     weight*attn} is written to its slot;
//   * 8-lane groups (16 B per lane = one 32-channel row) then own rows: a row's
//     segment is summed in registers -- the only serial chain is the FMA -- and added
//     to the slab row once;
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

constexpr int kGvWaves = 8;      // 512 threads
constexpr int kGvThreads = 64 * kGvWaves;
constexpr int kGvGroups = kGvThreads / 8;  // 8-lane groups
constexpr int kGvRowsMax = 336;  // 42 KiB slab
constexpr int kGvQcMax = 128;    // queries per chunk (16 KiB of grad_out rows)
constexpr int kGvSamplesMax = kGvThreads;  // one sample per thread per chunk
constexpr int kGvLevelsMax = 64;
// slab 42 K + rows 16 K + tap list 16 K + 3 x 336 counters/offsets + level table ~ 78.8 KiB
// -> two units per CU.
constexpr size_t kGvLdsBytes = size_t(kGvRowsMax) * 128 + size_t(kGvQcMax) * 128 +
                               size_t(kGvSamplesMax) * 32 + size_t(kGvRowsMax) * 12 + kGvWaves * 4 +
                               3 * kGvLevelsMax * 4;

// Inclusive prefix sum over the 64 lanes of a wave in 6 DPP adds (no LDS round trips):
// row_shr 1/2/4/8 scan each 16-lane row, row_bcast15 / row_bcast31 carry the row totals.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  int x = int(v);
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return uint32_t(x);
}

template <typename TV, typename TL>
__global__ void __launch_bounds__(kGvThreads)
msda_bwd_gv_tile_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                        const TL* __restrict__ loc, const TL* __restrict__ attn,
                        const TV* __restrict__ grad_out, TV* __restrict__ grad_value, MsdaDims d,
                        int units_min, int units_bound, int qc, int ablate) {
  constexpr int D = 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* slab = reinterpret_cast<float*>(smem);
  float* grows = slab + kGvRowsMax * D;                                   // [qc][32]
  uint2_t* list = reinterpret_cast<uint2_t*>(grows + kGvQcMax * D);       // [4*samples] taps
  uint32_t* cnt2 = reinterpret_cast<uint32_t*>(list + 4 * kGvSamplesMax); // [2][rows] taps per row
  uint32_t* offs = cnt2 + 2 * kGvRowsMax;                                 // [rows] segment starts
  uint32_t* wtot = offs + kGvRowsMax;                                     // [waves]
  int* meta = reinterpret_cast<int*>(wtot + kGvWaves);                    // [3*L] H, W, start

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int m = blockIdx.x % d.M;
  const int rest = blockIdx.x / d.M;
  const int unit = rest % units_bound;
  const int b = rest / units_bound;

  const int64_t row_stride = int64_t(d.M) * D;  // grad_out elements between queries
  const TV* go_head = grad_out + (int64_t(b) * d.Lq * d.M + m) * D;
  const int g0 = tid, g1 = tid + kGvThreads;    // this thread's float4 slots in [qc][8]

  // chunk 0's grad_out rows do not depend on the unit: request them before anything else
  float4_t pg0 = {0.f, 0.f, 0.f, 0.f}, pg1 = {0.f, 0.f, 0.f, 0.f};
  auto prefetch_rows = [&](int chunk) {
    const int q_base = chunk * qc;
    const int qa = q_base + (g0 >> 3), qb = q_base + (g1 >> 3);
    if ((g0 >> 3) < qc && qa < d.Lq) pg0 = gv_load4<TV>(go_head + int64_t(qa) * row_stride + (g0 & 7) * 4);
    if ((g1 >> 3) < qc && qb < d.Lq) pg1 = gv_load4<TV>(go_head + int64_t(qb) * row_stride + (g1 & 7) * 4);
  };
  prefetch_rows(0);

  // ---- level table: one round of vector loads, shared through LDS -------------------------
  if (tid < d.L) {
    meta[3 * tid] = int(shapes[2 * tid]);
    meta[3 * tid + 1] = int(shapes[2 * tid + 1]);
    meta[3 * tid + 2] = int(lsi[tid]);
  }
  for (int i = tid; i < 2 * kGvRowsMax; i += kGvThreads) cnt2[i] = 0;
  if (tid == 0) wtot[0] = 0;
  __syncthreads();

  // ---- which (level, pixel range) is this unit? ------------------------------------------------
  int lvl = -1, r0 = 0, r1 = 0, Hl = 0, Wl = 0, start = 0;
  {
    int running = 0;
    bool packed = true;
    int u = unit;
    for (int l = 0; l < d.L; ++l) {
      const int H = meta[3 * l], W = meta[3 * l + 1], st = meta[3 * l + 2];
      const int n = H * W;
      packed = packed && (st == running);
      running += n;
      if (lvl < 0 && n > 0) {
        int units = (n + kGvRowsMax - 1) / kGvRowsMax;
        if (units < units_min) units = units_min;
        if (units > n) units = n;
        const int rows_per_unit = (n + units - 1) / units;
        units = (n + rows_per_unit - 1) / rows_per_unit;
        if (u < units) {
          lvl = l; Hl = H; Wl = W; start = st;
          r0 = u * rows_per_unit;
          r1 = r0 + rows_per_unit < n ? r0 + rows_per_unit : n;
        } else {
          u -= units;
        }
      }
    }
    packed = packed && (running == d.S);
    if (!packed || lvl < 0) return;  // uniform over the workgroup
  }
  const int rows = r1 - r0;
  for (int i = tid; i < rows * (D / 4); i += kGvThreads)
    reinterpret_cast<float4_t*>(slab)[i] = float4_t{0.f, 0.f, 0.f, 0.f};

  const int LP = d.L * d.P;
  const float Hf = float(Hl), Wf = float(Wl);
  const int n_chunks = (d.Lq + qc - 1) / qc;
  const int sq = tid / d.P, sk = tid - sq * d.P;  // this thread's sample: query slot, point
  const bool has_sample = sq < qc;
  float px = 0.f, py = 0.f, pa = 0.f;
  auto prefetch_sample = [&](int chunk) {
    const int q = chunk * qc + sq;
    if (has_sample && q < d.Lq) {
      const int64_t wi = ((int64_t(b) * d.Lq + q) * d.M + m) * LP + lvl * d.P + sk;
      px = to_acc(loc[2 * wi]); py = to_acc(loc[2 * wi + 1]); pa = to_acc(attn[wi]);
    } else {
      px = -4.f; py = -4.f; pa = 0.f;  // fails the range test below
    }
  };
  prefetch_sample(0);

  const int grp = tid >> 3, ch4 = tid & 7;
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    uint32_t* cnt = cnt2 + (chunk & 1) * kGvRowsMax;        // this chunk's row counters
    uint32_t* cnt_next = cnt2 + ((chunk + 1) & 1) * kGvRowsMax;
    // ---- stage the chunk: grad_out rows -> LDS; geometry; rank each tap inside its row -------
    if ((g0 >> 3) < qc) reinterpret_cast<float4_t*>(grows)[g0] = pg0;
    if ((g1 >> 3) < qc) reinterpret_cast<float4_t*>(grows)[g1] = pg1;
    uint32_t mask = 0;
    int row00 = 0;
    float wt[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t rank[4] = {0u, 0u, 0u, 0u};
    const int dr[4] = {0, 1, Wl, Wl + 1};
    {
      const float h = py * Hf - 0.5f, w = px * Wf - 0.5f;
      if (h > -1.f && w > -1.f && h < Hf && w < Wf) {
        const float hf = floorf(h), wf = floorf(w);
        const int h0 = int(hf), w0i = int(wf);
        const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw = 1.f - lw;
        const bool top = h0 >= 0, bot = h0 + 1 <= Hl - 1, lef = w0i >= 0, rig = w0i + 1 <= Wl - 1;
        const int p00 = h0 * Wl + w0i;
        const int pa_ = p00, pb_ = p00 + 1, pc_ = p00 + Wl, pd_ = p00 + Wl + 1;
        mask = (uint32_t(top && lef && pa_ >= r0 && pa_ < r1)) |
               (uint32_t(top && rig && pb_ >= r0 && pb_ < r1) << 1) |
               (uint32_t(bot && lef && pc_ >= r0 && pc_ < r1) << 2) |
               (uint32_t(bot && rig && pd_ >= r0 && pd_ < r1) << 3);
        if (ablate == 2) mask = 0;
        row00 = p00 - r0;
        wt[0] = pa * (hh * hw); wt[1] = pa * (hh * lw); wt[2] = pa * (lh * hw); wt[3] = pa * (lh * lw);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (mask & (1u << t))  // integer LDS atomics are fast (tools/lds_atomic_bench.hip)
          rank[t] = __hip_atomic_fetch_add(cnt + row00 + dr[t], 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // the next chunk's loads go out now; they land while this chunk is sorted and applied
    if (chunk + 1 < n_chunks) { prefetch_rows(chunk + 1); prefetch_sample(chunk + 1); }
    __syncthreads();

    // ---- row counts -> segment offsets: DPP wave scan + one LDS allocation per wave --------
    // (segments need not be in row order, only disjoint)
    {
      const uint32_t my_cnt = tid < rows ? cnt[tid] : 0u;
      const uint32_t incl = wave_inclusive_scan(my_cnt);
      const uint32_t wave_total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
      uint32_t base = 0;
      if (lane == 0 && wave_total != 0)
        base = __hip_atomic_fetch_add(wtot, wave_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
      if (tid < rows) offs[tid] = base + incl - my_cnt;
    }
    __syncthreads();

    // ---- scatter the taps into their row segments ---------------------------------------------
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (mask & (1u << t))
        list[offs[row00 + dr[t]] + rank[t]] = uint2_t{uint32_t(sq), __float_as_uint(wt[t])};
    __syncthreads();

    // ---- 8-lane groups own rows: sum the row's segment in registers, one slab update ----------
    if (tid == 0) wtot[0] = 0;  // the allocator is idle from here to the next chunk's scan
    constexpr int kRpg = (kGvRowsMax + kGvGroups - 1) / kGvGroups;
    uint32_t rn[kRpg], ro[kRpg];
#pragma unroll
    for (int k = 0; k < kRpg; ++k) {  // this group's rows: counts and offsets in one batch
      const int row = grp + k * kGvGroups;
      rn[k] = (row < rows && ablate != 1) ? cnt[row] : 0u;
      ro[k] = row < rows ? offs[row] : 0u;
      if (row < rows) cnt_next[row] = 0;  // the other parity's counters are idle during this phase
    }
#pragma unroll
    for (int k = 0; k < kRpg; ++k) {
      const int row = grp + k * kGvGroups;
      const uint32_t n = rn[k];
      if (n == 0) continue;
      const uint2_t* seg = list + ro[k];
      const float4_t* g4 = reinterpret_cast<const float4_t*>(grows) + ch4;
      float4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
      uint32_t i = 0;
      for (; i + 4 <= n; i += 4) {  // four independent record -> row-read chains in flight
        const uint2_t r0_ = seg[i], r1_ = seg[i + 1], r2_ = seg[i + 2], r3_ = seg[i + 3];
        const float4_t x0 = g4[r0_.x * 8], x1 = g4[r1_.x * 8], x2 = g4[r2_.x * 8], x3 = g4[r3_.x * 8];
        a0 += __uint_as_float(r0_.y) * x0;
        a1 += __uint_as_float(r1_.y) * x1;
        a2 += __uint_as_float(r2_.y) * x2;
        a3 += __uint_as_float(r3_.y) * x3;
      }
      for (; i < n; ++i) {
        const uint2_t r = seg[i];
        a0 += __uint_as_float(r.y) * g4[r.x * 8];
      }
      reinterpret_cast<float4_t*>(slab)[row * 8 + ch4] += (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
  }

  // ---- write the slab: one owner per row, 16 B per lane, whole 128-B lines -----------------
  TV* out = grad_value + ((int64_t(b) * d.S + start + r0) * d.M + m) * D;
  for (int i = tid; i < rows * (D / 4); i += kGvThreads) {
    const int row = i >> 3, c4 = i & 7;
    const float4_t v = reinterpret_cast<const float4_t*>(slab)[i];
    gv_store4<TV>(out + int64_t(row) * d.M * D + c4 * 4, v);
  }
}

int msda_gv_units_bound(const MsdaDims& d, int units_min) {
  return d.L * (units_min + 1) + (d.S + kGvRowsMax - 1) / kGvRowsMax;
}

bool msda_d32_gv_supported(int vdt, int ldt, const MsdaDims& d) {
  if (d.D != 32 || vdt == VNX_F64) return false;
  if (vdt == VNX_F32 && ldt != VNX_F32) return false;
  if (d.P > 64 || d.L > kGvLevelsMax) return false;  // >= 8 queries x P samples per chunk; level table in LDS
  const int64_t blocks = int64_t(d.B) * d.M * msda_gv_units_bound(d, 16);
  return blocks < (int64_t(1) << 31);
}

template <typename TV, typename TL>
static int launch_gv(const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                     const void* grad_out, void* grad_value, const MsdaDims& d, int units_min,
                     int ablate, hipStream_t stream) {
  const int units_bound = msda_gv_units_bound(d, units_min);
  const int64_t blocks = int64_t(d.B) * d.M * units_bound;
  int qc = kGvSamplesMax / d.P;
  if (qc > kGvQcMax) qc = kGvQcMax;
  hipLaunchKernelGGL((msda_bwd_gv_tile_kernel<TV, TL>), dim3(uint32_t(blocks)), dim3(kGvThreads),
                     kGvLdsBytes, stream, shapes, lsi, (const TL*)loc, (const TL*)attn,
                     (const TV*)grad_out, (TV*)grad_value, d, units_min, units_bound, qc, ablate);
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
  const int ablate = (variant == 401) ? 1 : (variant == 402) ? 2 : 0;  // timing ablations only
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

}  // namespace vnx
