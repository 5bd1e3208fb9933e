// msda_d32.hip -- multi-scale deformable attention tuned for the production head
// geometry of SeqFormer / IDOL: 32 channels per head (d_model 256 / 8 heads).
//
// Machine mapping (gfx950, wave64):
//  * One pixel of one head is a 32-channel row: 128 B in fp32, 64 B in bf16/f16.
//    Eight lanes cover it with one 16-B (8-B) load each, so a wave issues eight
//    row gathers per load instruction and every request is a whole cache line.
//  * blockIdx.x % heads selects the head.  The dispatcher hands consecutive
//    workgroups to consecutive XCDs, so with 8 heads each XCD's private 4 MiB L2
//    only ever sees one head's 1/8 slice of `value` instead of all of it (an
//    affinity for speed only; nothing depends on the placement).
//  * Phase 1 (set-up): the wave's lanes each take one (query, level, point)
//    sample, read its location and weight, and compute the four tap byte offsets
//    and the four bilinear weights (pre-multiplied by the attention weight) ONCE,
//    then park them in LDS.  The reference recomputes this in each of the 32
//    channel threads of a head (ms_deform_im2col_cuda.cuh:272-296).
//  * Phase 2 (gather): eight-lane groups walk their samples, broadcast-read the
//    32-B tap record from LDS and issue four buffer loads per sample.  Taps that
//    fall outside the map carry an offset beyond the buffer descriptor's range:
//    the hardware returns zeros for them, which is exactly the reference's
//    zero-padding rule (cuh:55-78), with no branch and no memory traffic.
//  * QPW (queries per wave) trades per-wave work for wave count: the 64 lanes
//    are QPW queries x (8/QPW) sample groups x 8 channel lanes; sample groups
//    are summed with cross-lane exchanges at the end.
#include "vnx_common.h"

namespace vnx {

constexpr uint32_t kTapOutside = 0x80000000u;  // > any num_records the host admits

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  // wave-uniform inputs only (callers pass readfirstlane'd values)
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, int(bytes), 0x00020000);
}

// 4 consecutive channels of one tap as fp32
template <typename TV>
__device__ __forceinline__ float4_t load_tap(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off);

template <>
__device__ __forceinline__ float4_t load_tap<float>(__amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
  uint4_t r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, int(off), 0, 0);
  float4_t v;
  v.x = __uint_as_float(r.x); v.y = __uint_as_float(r.y);
  v.z = __uint_as_float(r.z); v.w = __uint_as_float(r.w);
  return v;
}
template <>
__device__ __forceinline__ float4_t load_tap<bf16_t>(__amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
  uint2_t r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, int(off), 0, 0);
  float4_t v;
  v.x = __uint_as_float(r.x << 16); v.y = __uint_as_float(r.x & 0xffff0000u);
  v.z = __uint_as_float(r.y << 16); v.w = __uint_as_float(r.y & 0xffff0000u);
  return v;
}
template <>
__device__ __forceinline__ float4_t load_tap<f16_t>(__amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
  uint2_t r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, int(off), 0, 0);
  typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
  half2_t lo = __builtin_bit_cast(half2_t, r.x), hi = __builtin_bit_cast(half2_t, r.y);
  float4_t v;
  v.x = float(lo.x); v.y = float(lo.y); v.z = float(hi.x); v.w = float(hi.y);
  return v;
}

template <typename TV>
__device__ __forceinline__ void store_row4(TV* p, float4_t v);
template <>
__device__ __forceinline__ void store_row4<float>(float* p, float4_t v) {
  *reinterpret_cast<float4_t*>(p) = v;
}
template <>
__device__ __forceinline__ void store_row4<bf16_t>(bf16_t* p, float4_t v) {
  uint2_t r;
  r.x = uint32_t(f32_to_bf16_bits(v.x)) | (uint32_t(f32_to_bf16_bits(v.y)) << 16);
  r.y = uint32_t(f32_to_bf16_bits(v.z)) | (uint32_t(f32_to_bf16_bits(v.w)) << 16);
  *reinterpret_cast<uint2_t*>(p) = r;
}
template <>
__device__ __forceinline__ void store_row4<f16_t>(f16_t* p, float4_t v) {
  typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
  half4_t h;
  h.x = _Float16(v.x); h.y = _Float16(v.y); h.z = _Float16(v.z); h.w = _Float16(v.w);
  *reinterpret_cast<half4_t*>(p) = h;
}

// -----------------------------------------------------------------------------
// forward
// -----------------------------------------------------------------------------
// LP = levels*points (template value 0 = use the runtime value, no unrolling).
template <typename TV, typename TL, int QPW, int WPB, int LP_T>
__global__ void __launch_bounds__(64 * WPB)
msda_fwd_d32_kernel(const TV* __restrict__ value, const int64_t* __restrict__ shapes,
                    const int64_t* __restrict__ lsi, const TL* __restrict__ loc,
                    const TL* __restrict__ attn, TV* __restrict__ out, MsdaDims d,
                    int tiles_per_batch) {
  constexpr int D = 32;
  constexpr int PG = 8 / QPW;            // sample groups per query
  constexpr int kRowBytes = D * int(sizeof(TV));
  constexpr int kLaneBytes = kRowBytes / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int LP = LP_T > 0 ? LP_T : d.L * d.P;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = blockIdx.x % d.M;
  const int tile = blockIdx.x / d.M;
  const int b = tile / tiles_per_batch;
  const int q0 = (tile - b * tiles_per_batch) * (QPW * WPB) + wave * QPW;

  // per-wave LDS: [QPW][LP+1] tap offsets (uint4) then [QPW][LP+1] tap weights (float4)
  const int ent = QPW * (LP + 1);
  uint4_t* s_off = reinterpret_cast<uint4_t*>(smem) + size_t(wave) * 2 * ent;
  float4_t* s_wt = reinterpret_cast<float4_t*>(s_off + ent);

  const int pixel_bytes = d.M * kRowBytes;

  // ---- phase 1: one (query, sample) pair per lane and step --------------------
  const int pairs = QPW * LP;
  for (int e = lane; e < pairs; e += 64) {
    const int qi = e / LP, p = e - qi * LP;
    const int q = q0 + qi;
    uint4_t o4 = {kTapOutside, kTapOutside, kTapOutside, kTapOutside};
    float4_t w4 = {0.f, 0.f, 0.f, 0.f};
    if (q < d.Lq) {
      const int l = p / d.P;
      const int64_t wi = ((int64_t(b) * d.Lq + q) * d.M + m) * LP + p;
      const float x = to_acc(loc[2 * wi]), y = to_acc(loc[2 * wi + 1]);
      const float a = to_acc(attn[wi]);
      const int H = int(shapes[2 * l]), W = int(shapes[2 * l + 1]);
      const int start = int(lsi[l]);
      const float h = y * float(H) - 0.5f, w = x * float(W) - 0.5f;
      if (h > -1.f && w > -1.f && h < float(H) && w < float(W)) {
        const float hf = floorf(h), wf = floorf(w);
        const int h0 = int(hf), w0 = int(wf);
        const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw = 1.f - lw;
        const bool top = h0 >= 0, bot = h0 + 1 <= H - 1, lef = w0 >= 0, rig = w0 + 1 <= W - 1;
        const uint32_t pb = uint32_t(pixel_bytes);
        const uint32_t o00 = uint32_t(start + h0 * W + w0) * pb;  // mod 2^32 on purpose
        o4.x = (top && lef) ? o00 : kTapOutside;
        o4.y = (top && rig) ? o00 + pb : kTapOutside;
        o4.z = (bot && lef) ? o00 + uint32_t(W) * pb : kTapOutside;
        o4.w = (bot && rig) ? o00 + uint32_t(W + 1) * pb : kTapOutside;
        w4.x = a * (hh * hw); w4.y = a * (hh * lw); w4.z = a * (lh * hw); w4.w = a * (lh * lw);
      }
    }
    s_off[qi * (LP + 1) + p] = o4;
    s_wt[qi * (LP + 1) + p] = w4;
  }
  if (WPB > 1) __syncthreads(); else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- phase 2: gather ----------------------------------------------------------
  const int ch = lane & 7;
  const int qi = (lane >> 3) % QPW;
  const int pg = lane / (8 * QPW);
  const int q = q0 + qi;

  const TV* head_base = value + (int64_t(b) * d.S * d.M + m) * D;
  const uint32_t head_bytes = uint32_t((int64_t(d.S) * d.M - m) * kRowBytes);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(uintptr_t(head_base)));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(uint32_t(uintptr_t(head_base) >> 32));
  const __amdgpu_buffer_rsrc_t rsrc =
      make_rsrc(reinterpret_cast<const void*>(uintptr_t(lo) | (uintptr_t(hi) << 32)),
                __builtin_amdgcn_readfirstlane(head_bytes));

  const int per_group = LP / PG;  // host guarantees LP % PG == 0
  const uint4_t* g_off = s_off + qi * (LP + 1) + pg * per_group;
  const float4_t* g_wt = s_wt + qi * (LP + 1) + pg * per_group;
  const uint32_t lane_off = uint32_t(ch * kLaneBytes);

  float4_t acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int kUnroll = LP_T > 0 ? LP_T / PG : 1;
#pragma unroll kUnroll
  for (int i = 0; i < per_group; ++i) {
    const uint4_t o = g_off[i];
    const float4_t w = g_wt[i];
    const float4_t v0 = load_tap<TV>(rsrc, o.x + lane_off);
    const float4_t v1 = load_tap<TV>(rsrc, o.y + lane_off);
    const float4_t v2 = load_tap<TV>(rsrc, o.z + lane_off);
    const float4_t v3 = load_tap<TV>(rsrc, o.w + lane_off);
    acc += w.x * v0;
    acc += w.y * v1;
    acc += w.z * v2;
    acc += w.w * v3;
  }

  // sum the sample groups of a query (lanes that differ only in pg)
#pragma unroll
  for (int off = 8 * QPW; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off, 64);
    acc.y += __shfl_xor(acc.y, off, 64);
    acc.z += __shfl_xor(acc.z, off, 64);
    acc.w += __shfl_xor(acc.w, off, 64);
  }
  if (pg == 0 && q < d.Lq) {
    TV* o = out + ((int64_t(b) * d.Lq + q) * d.M + m) * D + ch * 4;
    store_row4<TV>(o, acc);
  }
}

struct FwdCfg { int qpw; int wpb; };

static FwdCfg pick_fwd_cfg(const MsdaDims& d, int variant) {
  // variant: 0 auto; 2..5 force QPW = 8,4,2,1 with 4 waves/block; 12..15 the same with 1 wave/block
  const int LP = d.L * d.P;
  FwdCfg c{8, 4};
  if (variant >= 2 && variant <= 5) c = FwdCfg{8 >> (variant - 2), 4};
  else if (variant >= 12 && variant <= 15) c = FwdCfg{8 >> (variant - 12), 1};
  else {
    const int64_t rows = int64_t(d.B) * d.Lq;
    // few rows: spread each query over more lanes so all 256 CUs get waves
    if (rows * d.M <= 256 * 64) c = FwdCfg{2, 1};
    else if (rows * d.M <= 256 * 256) c = FwdCfg{4, 2};
    else c = FwdCfg{8, 4};
  }
  while (c.qpw < 8 && (LP % (8 / c.qpw)) != 0) c.qpw <<= 1;
  return c;
}

template <typename TV, typename TL, int QPW, int WPB>
static int launch_fwd_cfg(const void* value, const int64_t* shapes, const int64_t* lsi,
                          const void* loc, const void* attn, void* out, const MsdaDims& d,
                          hipStream_t stream) {
  const int LP = d.L * d.P;
  const int tiles_per_batch = (d.Lq + QPW * WPB - 1) / (QPW * WPB);
  const int64_t blocks = int64_t(d.B) * tiles_per_batch * d.M;
  if (blocks >= (int64_t(1) << 31)) {
    set_error("msda_forward: %lld workgroups exceed the grid limit", (long long)blocks);
    return VNX_ERR_UNSUPPORTED;
  }
  const size_t lds = size_t(WPB) * 2 * QPW * (LP + 1) * 16;
  if (LP == 16)
    hipLaunchKernelGGL((msda_fwd_d32_kernel<TV, TL, QPW, WPB, 16>), dim3(uint32_t(blocks)),
                       dim3(64 * WPB), lds, stream, (const TV*)value, shapes, lsi,
                       (const TL*)loc, (const TL*)attn, (TV*)out, d, tiles_per_batch);
  else
    hipLaunchKernelGGL((msda_fwd_d32_kernel<TV, TL, QPW, WPB, 0>), dim3(uint32_t(blocks)),
                       dim3(64 * WPB), lds, stream, (const TV*)value, shapes, lsi,
                       (const TL*)loc, (const TL*)attn, (TV*)out, d, tiles_per_batch);
  return check_launch("msda_fwd_d32");
}

template <typename TV, typename TL>
static int launch_fwd(const void* value, const int64_t* shapes, const int64_t* lsi,
                      const void* loc, const void* attn, void* out, const MsdaDims& d,
                      int variant, hipStream_t stream) {
  const FwdCfg c = pick_fwd_cfg(d, variant);
#define VNX_CASE(Q, W)                                                                       \
  if (c.qpw == Q && c.wpb == W)                                                              \
    return launch_fwd_cfg<TV, TL, Q, W>(value, shapes, lsi, loc, attn, out, d, stream);
  VNX_CASE(8, 4) VNX_CASE(4, 4) VNX_CASE(2, 4) VNX_CASE(1, 4)
  VNX_CASE(8, 1) VNX_CASE(4, 1) VNX_CASE(2, 1) VNX_CASE(1, 1)
  VNX_CASE(4, 2)
#undef VNX_CASE
  set_error("msda_forward: no kernel for qpw=%d wpb=%d", c.qpw, c.wpb);
  return VNX_ERR_UNSUPPORTED;
}

bool msda_d32_fwd_supported(int vdt, int ldt, const MsdaDims& d) {
  if (d.D != 32) return false;
  if (vdt == VNX_F64) return false;
  if (vdt == VNX_F32 && ldt != VNX_F32) return false;
  const int LP = d.L * d.P;
  if (LP > 64) return false;                                    // LDS record budget per wave
  if (int64_t(d.S) * d.M * 32 * elem_size(vdt) >= (int64_t(1) << 31)) return false;  // descriptor range
  return true;
}

bool msda_d32_bwd_supported(int, int, const MsdaDims&) { return false; }

int msda_forward_d32(int vdt, int ldt, const void* value, const int64_t* shapes,
                     const int64_t* lsi, const void* loc, const void* attn, void* out, MsdaDims d,
                     int variant, hipStream_t stream) {
  if (vdt == VNX_F32) return launch_fwd<float, float>(value, shapes, lsi, loc, attn, out, d, variant, stream);
  if (vdt == VNX_BF16 && ldt == VNX_F32) return launch_fwd<bf16_t, float>(value, shapes, lsi, loc, attn, out, d, variant, stream);
  if (vdt == VNX_BF16 && ldt == VNX_BF16) return launch_fwd<bf16_t, bf16_t>(value, shapes, lsi, loc, attn, out, d, variant, stream);
  if (vdt == VNX_F16 && ldt == VNX_F32) return launch_fwd<f16_t, float>(value, shapes, lsi, loc, attn, out, d, variant, stream);
  if (vdt == VNX_F16 && ldt == VNX_F16) return launch_fwd<f16_t, f16_t>(value, shapes, lsi, loc, attn, out, d, variant, stream);
  set_error("msda_forward_d32: unsupported dtype pair (%d, %d)", vdt, ldt);
  return VNX_ERR_INVALID_ARGUMENT;
}

int msda_backward_d32(int, int, const void*, const int64_t*, const int64_t*, const void*,
                      const void*, const void*, void*, void*, void*, MsdaDims, int, hipStream_t) {
  set_error("msda_backward_d32: not built");
  return VNX_ERR_UNSUPPORTED;
}

}  // namespace vnx
