// msda_d32_gv.hip -- grad_value of multi-scale deformable attention without global
// atomics ("owner computes"), for 32-channel heads.
//
// Why: the reference scatters every tap with one fp32 atomicAdd per channel
// (ms_deform_im2col_cuda.cuh:125-152).  On MI355X global fp32 atomics retire at
// ~322 G dwords/s chip-wide whatever the address pattern or working set
// (tools/atomic_bench.hip; one dword per clock per L2 channel), which puts a floor
// of 76 us under the T=5 decoder call (24.6 M dword atomics) -- 6x its HBM time.
//
// Here every row of grad_value has exactly one owner.  A workgroup ("unit") owns a
// contiguous range of pixels of ONE level for one (batch, head) and keeps that slab
// [rows][32] in LDS (fp32).  It walks the queries in chunks:
//   * the chunk's grad_out rows of this head go to LDS once (coalesced 16-B loads),
//     and one thread per sample of the unit's level computes the bilinear geometry;
//     samples with a tap inside the unit's range append a 24-B record
//     {query slot | tap mask, first tap row, 4 weights*attn} to an LDS list;
//   * half-waves then take records: 32 lanes = the 32 channels; one LDS read of the
//     grad_out row, up to four ds_add_f32 into the slab (32 consecutive banks).
//   * the next chunk's global loads are issued before the records are processed, so
//     their latency hides behind the LDS work.
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
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));

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
constexpr int kGvRowsMax = 384;  // 48 KiB slab
constexpr int kGvQcMax = 128;    // queries per chunk (16 KiB of grad_out rows)
constexpr int kGvSamplesMax = kGvThreads;  // one sample per thread per chunk
// slab 48 K + rows 16 K + records 12 K + counters = 76 KiB + 16 B -> two units per CU

// A record = {query slot in the chunk | tap mask << 16, first tap's row relative to the unit
// (may be negative / beyond)} + 4 weights, kept as two arrays so both stay naturally aligned.

template <typename TV, typename TL>
__global__ void __launch_bounds__(kGvThreads)
msda_bwd_gv_tile_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                        const TL* __restrict__ loc, const TL* __restrict__ attn,
                        const TV* __restrict__ grad_out, TV* __restrict__ grad_value, MsdaDims d,
                        int units_min, int units_bound, int qc) {
  constexpr int D = 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* slab = reinterpret_cast<float*>(smem);
  float* grows = slab + kGvRowsMax * D;                                   // [qc][32]
  uint4_t* rec_w = reinterpret_cast<uint4_t*>(grows + kGvQcMax * D);      // [samples] 4 weights
  uint2_t* rec_h = reinterpret_cast<uint2_t*>(rec_w + kGvSamplesMax);     // [samples] slot|mask, row
  uint32_t* counters = reinterpret_cast<uint32_t*>(rec_h + kGvSamplesMax);  // [2]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int m = blockIdx.x % d.M;
  const int rest = blockIdx.x / d.M;
  const int unit = rest % units_bound;
  const int b = rest / units_bound;

  // ---- which (level, pixel range) is this unit? (all scalar) -----------------------
  int lvl = -1, r0 = 0, r1 = 0, Hl = 0, Wl = 0, start = 0;
  {
    int64_t running = 0;
    bool packed = true;
    int u = unit;
    for (int l = 0; l < d.L; ++l) {
      const int H = int(shapes[2 * l]), W = int(shapes[2 * l + 1]);
      const int n = H * W;
      packed = packed && (lsi[l] == running);
      running += n;
      if (lvl < 0 && n > 0) {
        int units = (n + kGvRowsMax - 1) / kGvRowsMax;
        if (units < units_min) units = units_min;
        if (units > n) units = n;
        const int rows_per_unit = (n + units - 1) / units;
        units = (n + rows_per_unit - 1) / rows_per_unit;
        if (u < units) {
          lvl = l; Hl = H; Wl = W; start = int(lsi[l]);
          r0 = u * rows_per_unit;
          r1 = r0 + rows_per_unit < n ? r0 + rows_per_unit : n;
        } else {
          u -= units;
        }
      }
    }
    packed = packed && (running == d.S);
    if (!packed || lvl < 0) return;
  }
  const int rows = r1 - r0;

  // ---- zero the slab and the counters ------------------------------------------------
  for (int i = tid; i < rows * (D / 4); i += kGvThreads)
    reinterpret_cast<float4_t*>(slab)[i] = float4_t{0.f, 0.f, 0.f, 0.f};
  if (tid < 2) counters[tid] = 0;
  __syncthreads();

  const int LP = d.L * d.P;
  const float Hf = float(Hl), Wf = float(Wl);
  const int n_chunks = (d.Lq + qc - 1) / qc;
  const int64_t row_stride = int64_t(d.M) * D;  // grad_out elements between queries
  const TV* go_head = grad_out + (int64_t(b) * d.Lq * d.M + m) * D;

  // this thread's sample in a chunk, and its two 16-B pieces of the chunk's grad_out rows
  const int sq = tid / d.P, sk = tid - sq * d.P;         // query slot, point
  const bool has_sample = sq < qc;
  const int g0 = tid, g1 = tid + kGvThreads;             // float4 index in [qc][8]

  // prefetch registers
  float px = 0.f, py = 0.f, pa = 0.f;
  float4_t pg0 = {0.f, 0.f, 0.f, 0.f}, pg1 = {0.f, 0.f, 0.f, 0.f};
  auto prefetch = [&](int chunk) {
    const int q_base = chunk * qc;
    const int q = q_base + sq;
    if (has_sample && q < d.Lq) {
      const int64_t wi = ((int64_t(b) * d.Lq + q) * d.M + m) * LP + lvl * d.P + sk;
      px = to_acc(loc[2 * wi]); py = to_acc(loc[2 * wi + 1]); pa = to_acc(attn[wi]);
    } else {
      px = -4.f; py = -4.f; pa = 0.f;  // fails the range test below
    }
    const int qa = q_base + (g0 >> 3), qb = q_base + (g1 >> 3);
    if ((g0 >> 3) < qc && qa < d.Lq) pg0 = gv_load4<TV>(go_head + int64_t(qa) * row_stride + (g0 & 7) * 4);
    if ((g1 >> 3) < qc && qb < d.Lq) pg1 = gv_load4<TV>(go_head + int64_t(qb) * row_stride + (g1 & 7) * 4);
  };
  prefetch(0);

  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    uint32_t* counter = counters + (chunk & 1);
    // ---- stage this chunk: grad_out rows -> LDS, geometry -> records -----------------------
    if ((g0 >> 3) < qc) reinterpret_cast<float4_t*>(grows)[g0] = pg0;
    if ((g1 >> 3) < qc) reinterpret_cast<float4_t*>(grows)[g1] = pg1;
    {
      const float h = py * Hf - 0.5f, w = px * Wf - 0.5f;
      bool mine = false;
      uint32_t mask = 0;
      int row00 = 0;
      float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
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
        mine = mask != 0;
        row00 = p00 - r0;
        w0 = pa * (hh * hw); w1 = pa * (hh * lw); w2 = pa * (lh * hw); w3 = pa * (lh * lw);
      }
      const unsigned long long ballot = __ballot(mine);
      if (ballot != 0) {
        const int n_mine = __builtin_popcountll(ballot);
        uint32_t base = 0;
        if (lane == 0) base = __hip_atomic_fetch_add(counter, uint32_t(n_mine), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP);
        base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
        if (mine) {
          const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi(uint32_t(ballot >> 32),
                                                                __builtin_amdgcn_mbcnt_lo(uint32_t(ballot), 0));
          rec_h[pos] = uint2_t{uint32_t(sq) | (mask << 16), uint32_t(row00)};
          rec_w[pos] = uint4_t{__float_as_uint(w0), __float_as_uint(w1), __float_as_uint(w2),
                               __float_as_uint(w3)};
        }
      }
    }
    // ---- next chunk's loads go out now; they land while the records are processed -----------
    if (chunk + 1 < n_chunks) prefetch(chunk + 1);
    __syncthreads();
    const int n_rec = int(*counter);
    if (tid == 0) counters[(chunk + 1) & 1] = 0;

    // ---- half-waves take records; 32 lanes = the 32 channels of the rows -----------------
    const int hw_id = tid >> 5, c = tid & 31;
    constexpr int kHalfWaves = kGvThreads / 32;
    for (int i = hw_id; i < n_rec; i += 2 * kHalfWaves) {
      // two records in flight per half-wave
      const int i2 = i + kHalfWaves;
      const bool second = i2 < n_rec;
      const int ib = second ? i2 : i;
      const uint2_t ha = rec_h[i], hb = rec_h[ib];
      const uint4_t wa = rec_w[i], wb = rec_w[ib];
      const float ga = grows[(ha.x & 0xffffu) * D + c];
      const float gb = grows[(hb.x & 0xffffu) * D + c];
      float* sa = slab + int(ha.y) * D + c;
      float* sb = slab + int(hb.y) * D + c;
      const uint32_t ma = ha.x >> 16, mb = second ? hb.x >> 16 : 0u;
      if (ma & 1u) unsafeAtomicAdd(sa, __uint_as_float(wa.x) * ga);
      if (ma & 2u) unsafeAtomicAdd(sa + D, __uint_as_float(wa.y) * ga);
      if (ma & 4u) unsafeAtomicAdd(sa + Wl * D, __uint_as_float(wa.z) * ga);
      if (ma & 8u) unsafeAtomicAdd(sa + (Wl + 1) * D, __uint_as_float(wa.w) * ga);
      if (mb & 1u) unsafeAtomicAdd(sb, __uint_as_float(wb.x) * gb);
      if (mb & 2u) unsafeAtomicAdd(sb + D, __uint_as_float(wb.y) * gb);
      if (mb & 4u) unsafeAtomicAdd(sb + Wl * D, __uint_as_float(wb.z) * gb);
      if (mb & 8u) unsafeAtomicAdd(sb + (Wl + 1) * D, __uint_as_float(wb.w) * gb);
    }
    __syncthreads();
  }

  // ---- write the slab: one owner per row, 16 B per lane, whole 128-B lines -----------------
  TV* out = grad_value + ((int64_t(b) * d.S + start + r0) * d.M + m) * D;
  for (int i = tid; i < rows * (D / 4); i += kGvThreads) {
    const int row = i >> 3, ch4 = i & 7;
    const float4_t v = reinterpret_cast<const float4_t*>(slab)[i];
    gv_store4<TV>(out + int64_t(row) * d.M * D + ch4 * 4, v);
  }
}

int msda_gv_units_bound(const MsdaDims& d, int units_min) {
  return d.L * (units_min + 1) + (d.S + kGvRowsMax - 1) / kGvRowsMax;
}

bool msda_d32_gv_supported(int vdt, int ldt, const MsdaDims& d) {
  if (d.D != 32 || vdt == VNX_F64) return false;
  if (vdt == VNX_F32 && ldt != VNX_F32) return false;
  if (d.P > 64) return false;  // a chunk holds at least 8 queries x P samples, one per thread
  const int64_t blocks = int64_t(d.B) * d.M * msda_gv_units_bound(d, 16);
  return blocks < (int64_t(1) << 31);
}

template <typename TV, typename TL>
static int launch_gv(const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                     const void* grad_out, void* grad_value, const MsdaDims& d, int units_min,
                     hipStream_t stream) {
  const int units_bound = msda_gv_units_bound(d, units_min);
  const int64_t blocks = int64_t(d.B) * d.M * units_bound;
  int qc = kGvSamplesMax / d.P;
  if (qc > kGvQcMax) qc = kGvQcMax;
  const size_t lds = size_t(kGvRowsMax) * 128 + size_t(kGvQcMax) * 128 + size_t(kGvSamplesMax) * 24 + 16;
  hipLaunchKernelGGL((msda_bwd_gv_tile_kernel<TV, TL>), dim3(uint32_t(blocks)), dim3(kGvThreads), lds,
                     stream, shapes, lsi, (const TL*)loc, (const TL*)attn, (const TV*)grad_out,
                     (TV*)grad_value, d, units_min, units_bound, qc);
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
#define VNX_ARGS shapes, lsi, loc, attn, grad_out, grad_value, d, units_min, stream
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
