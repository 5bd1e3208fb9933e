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
// contiguous range of pixels of ONE level for one (batch, head), keeps that slab
// [rows][32] in LDS (fp32), scans all samples of its level for that (batch, head),
// and for each tap landing in its range adds w*attn*grad_out[q, head, :] into the
// slab with LDS atomics (32 consecutive banks, conflict-free).  The slab is then
// written once with 16-B stores: no zero-fill pass, no global atomic, no fp32
// workspace / convert pass for 16-bit tensors, and the result does not depend on
// the order workgroups run in.
//
// The unit table is derived ON DEVICE from spatial_shapes / level_start_index (the
// ABI hands these over as device tensors, ms_deform_attn_cuda.cu:67-68): each level
// is cut into max(units_min, ceil(n_l / ROWS_MAX)) ranges, so every level -- which
// receives the same number of samples -- gets at least units_min workgroups.  The
// scheme needs the levels packed back to back (level_start_index[l] == sum of the
// previous H*W and their total == spatial_size), which is how the reference builds
// them (deformable_transformer.py:97-106).  Every workgroup re-derives that
// predicate; when it fails the kernel does nothing and the general path (atomics
// into a zero-filled image) does the work -- see msda_backward in capi.hip.
#include "vnx_common.h"

namespace vnx {

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float gv_to_float(float x) { return x; }
__device__ __forceinline__ float gv_to_float(bf16_t x) { return to_acc(x); }
__device__ __forceinline__ float gv_to_float(f16_t x) { return to_acc(x); }

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

constexpr int kRowBits = 10;  // slab rows per unit <= 1024

template <typename TV, typename TL, int WAVES, int ROWS_MAX>
__global__ void __launch_bounds__(64 * WAVES)
msda_bwd_gv_tile_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                        const TL* __restrict__ loc, const TL* __restrict__ attn,
                        const TV* __restrict__ grad_out, TV* __restrict__ grad_value, MsdaDims d,
                        int units_min, int units_bound) {
  static_assert(ROWS_MAX <= (1 << kRowBits), "row index must fit the record");
  constexpr int D = 32;
  constexpr int kList = 256;  // taps one wave can emit per 64-sample step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* slab = reinterpret_cast<float*>(smem);
  uint2_t* lists = reinterpret_cast<uint2_t*>(smem + size_t(ROWS_MAX) * D * 4);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
        int units = (n + ROWS_MAX - 1) / ROWS_MAX;
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

  // ---- zero the slab ------------------------------------------------------------------
  for (int i = threadIdx.x; i < rows * (D / 4); i += 64 * WAVES)
    reinterpret_cast<float4_t*>(slab)[i] = float4_t{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // ---- scan this level's samples of (b, m) ----------------------------------------------
  uint2_t* list = lists + wave * kList;
  const int LP = d.L * d.P;
  const int n_samples = d.Lq * d.P;
  const int half = lane >> 5, c = lane & 31;
  const float Hf = float(Hl), Wf = float(Wl);
  for (int base = wave * 64; base < n_samples; base += WAVES * 64) {
    const int e = base + lane;
    int rt[4] = {0, 0, 0, 0};
    float wt[4] = {0.f, 0.f, 0.f, 0.f};
    bool ok[4] = {false, false, false, false};
    int q = 0;
    if (e < n_samples) {
      q = e / d.P;
      const int k = e - q * d.P;
      const int64_t wi = ((int64_t(b) * d.Lq + q) * d.M + m) * LP + lvl * d.P + k;
      const float x = to_acc(loc[2 * wi]), y = to_acc(loc[2 * wi + 1]);
      const float a = to_acc(attn[wi]);
      const float h = y * Hf - 0.5f, w = x * Wf - 0.5f;
      if (h > -1.f && w > -1.f && h < Hf && w < Wf) {
        const float hf = floorf(h), wf = floorf(w);
        const int h0 = int(hf), w0 = int(wf);
        const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw = 1.f - lw;
        const bool top = h0 >= 0, bot = h0 + 1 <= Hl - 1, lef = w0 >= 0, rig = w0 + 1 <= Wl - 1;
        const int p00 = h0 * Wl + w0;
        rt[0] = p00; rt[1] = p00 + 1; rt[2] = p00 + Wl; rt[3] = p00 + Wl + 1;
        wt[0] = a * (hh * hw); wt[1] = a * (hh * lw); wt[2] = a * (lh * hw); wt[3] = a * (lh * lw);
        ok[0] = top && lef; ok[1] = top && rig; ok[2] = bot && lef; ok[3] = bot && rig;
      }
    }
    // compact the taps that land in [r0, r1) into this wave's list
    int cnt = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool mine = ok[t] && rt[t] >= r0 && rt[t] < r1;
      const unsigned long long mask = __ballot(mine);
      const int pos = cnt + __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32),
                                                      __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0));
      if (mine) list[pos] = uint2_t{(uint32_t(q) << kRowBits) | uint32_t(rt[t] - r0), __float_as_uint(wt[t])};
      cnt += __builtin_popcountll(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // two taps per step (one per half-wave); 32 lanes = the 32 channels of the tap's row
    const int64_t go_base = (int64_t(b) * d.Lq * d.M + m) * D + c;
    constexpr int UN = 4;
    for (int i = 0; i < cnt; i += 2 * UN) {
      uint2_t rec[UN];
      float g[UN];
#pragma unroll
      for (int j = 0; j < UN; ++j) {
        const int idx = i + 2 * j + half;
        rec[j] = idx < cnt ? list[idx] : uint2_t{0u, 0u};
      }
#pragma unroll
      for (int j = 0; j < UN; ++j) {
        const int idx = i + 2 * j + half;
        g[j] = idx < cnt ? gv_to_float(grad_out[go_base + int64_t(rec[j].x >> kRowBits) * d.M * D]) : 0.f;
      }
#pragma unroll
      for (int j = 0; j < UN; ++j) {
        const int idx = i + 2 * j + half;
        if (idx < cnt)
          unsafeAtomicAdd(&slab[(rec[j].x & ((1u << kRowBits) - 1u)) * D + c], __uint_as_float(rec[j].y) * g[j]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // ---- write the slab: one owner per row, 16 B per lane, whole 128-B lines -----------------
  TV* out = grad_value + ((int64_t(b) * d.S + start + r0) * d.M + m) * D;
  for (int i = threadIdx.x; i < rows * (D / 4); i += 64 * WAVES) {
    const int row = i >> 3, ch4 = i & 7;
    const float4_t v = reinterpret_cast<const float4_t*>(slab)[i];
    gv_store4<TV>(out + int64_t(row) * d.M * D + ch4 * 4, v);
  }
}

constexpr int kGvWaves = 8;
constexpr int kGvRowsMax = 480;  // 60 KiB slab + 16 KiB lists = 76 KiB -> two units per CU

int msda_gv_units_bound(const MsdaDims& d, int units_min) {
  return d.L * (units_min + 1) + (d.S + kGvRowsMax - 1) / kGvRowsMax;
}

bool msda_d32_gv_supported(int vdt, int ldt, const MsdaDims& d) {
  if (d.D != 32 || vdt == VNX_F64) return false;
  if (vdt == VNX_F32 && ldt != VNX_F32) return false;
  if (d.Lq >= (1 << (32 - kRowBits))) return false;  // query index shares a word with the row
  const int64_t blocks = int64_t(d.B) * d.M * msda_gv_units_bound(d, 16);
  return blocks < (int64_t(1) << 31);
}

template <typename TV, typename TL>
static int launch_gv(const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn,
                     const void* grad_out, void* grad_value, const MsdaDims& d, int units_min,
                     hipStream_t stream) {
  const int units_bound = msda_gv_units_bound(d, units_min);
  const int64_t blocks = int64_t(d.B) * d.M * units_bound;
  const size_t lds = size_t(kGvRowsMax) * 32 * 4 + size_t(kGvWaves) * 256 * 8;
  hipLaunchKernelGGL((msda_bwd_gv_tile_kernel<TV, TL, kGvWaves, kGvRowsMax>), dim3(uint32_t(blocks)),
                     dim3(64 * kGvWaves), lds, stream, shapes, lsi, (const TL*)loc, (const TL*)attn,
                     (const TV*)grad_out, (TV*)grad_value, d, units_min, units_bound);
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
