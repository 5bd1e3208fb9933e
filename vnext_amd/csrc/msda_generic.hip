// msda_generic.hip -- shape-agnostic multi-scale deformable attention kernels.
//
// These cover every (heads, channels, levels, points) combination and every
// dtype the ABI accepts; the reference's own channel sweep
// (ops/test.py:85 -- D = 30, 32, 64, 71, 1025, 2048, 3096) runs through them.
// The tuned D=32 kernels live in msda_d32.hip.
//
// Semantics follow ms_deform_im2col_cuda.cuh:33-84 (forward taps), :87-159
// (backward taps) and :253-298 (index decode); the mapping onto the machine is
// ours: wave64 sub-groups own one (batch, query, head) row, channel partials
// are reduced with cross-lane shuffles instead of shared memory + a serial
// thread-0 loop (cuh:376-394), and the location/weight gradients are written
// once, non-atomically.
#include "vnx_common.h"

namespace vnx {

template <typename A>
struct Taps {
  bool inside;
  bool ok[4];
  int pix[4];  // pixel index inside the level, y*W+x
  A lh, lw, hh, hw;
};

template <typename A>
__device__ __forceinline__ Taps<A> make_taps(A x, A y, int H, int W) {
  Taps<A> t;
  const A h = y * A(H) - A(0.5);
  const A w = x * A(W) - A(0.5);
  t.inside = (h > A(-1)) && (w > A(-1)) && (h < A(H)) && (w < A(W));
  const int h0 = int(floor_acc(h)), w0 = int(floor_acc(w));
  const int h1 = h0 + 1, w1 = w0 + 1;
  t.lh = h - A(h0);
  t.lw = w - A(w0);
  t.hh = A(1) - t.lh;
  t.hw = A(1) - t.lw;
  t.ok[0] = t.inside && h0 >= 0 && w0 >= 0;
  t.ok[1] = t.inside && h0 >= 0 && w1 <= W - 1;
  t.ok[2] = t.inside && h1 <= H - 1 && w0 >= 0;
  t.ok[3] = t.inside && h1 <= H - 1 && w1 <= W - 1;
  t.pix[0] = h0 * W + w0;
  t.pix[1] = h0 * W + w1;
  t.pix[2] = h1 * W + w0;
  t.pix[3] = h1 * W + w1;
  return t;
}

// One thread per output element (b, q, m, c); consecutive lanes walk the channel
// axis so every tap is a contiguous read.
template <typename TV, typename TL>
__global__ void __launch_bounds__(256)
msda_fwd_generic_kernel(const TV* __restrict__ value, const int64_t* __restrict__ shapes,
                        const int64_t* __restrict__ lsi, const TL* __restrict__ loc,
                        const TL* __restrict__ attn, TV* __restrict__ out, MsdaDims d) {
  using A = acc_t<TV>;
  const int64_t total = int64_t(d.B) * d.Lq * d.M * d.D;
  const int64_t row_stride = int64_t(d.M) * d.D;
  for (int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += int64_t(gridDim.x) * blockDim.x) {
    const int c = int(idx % d.D);
    const int64_t rm = idx / d.D;  // (b*Lq+q)*M + m
    const int m = int(rm % d.M);
    const int64_t r = rm / d.M;
    const int b = int(r / d.Lq);
    const int64_t wbase = rm * d.L * d.P;
    A acc = A(0);
    for (int l = 0; l < d.L; ++l) {
      const int H = int(shapes[2 * l]), W = int(shapes[2 * l + 1]);
      const TV* vl = value + (int64_t(b) * d.S + lsi[l]) * row_stride + int64_t(m) * d.D + c;
      for (int k = 0; k < d.P; ++k) {
        const int64_t wi = wbase + l * d.P + k;
        const A x = A(to_acc(loc[2 * wi])), y = A(to_acc(loc[2 * wi + 1]));
        const Taps<A> t = make_taps<A>(x, y, H, W);
        if (!t.inside) continue;
        const A a = A(to_acc(attn[wi]));
        const A v1 = t.ok[0] ? A(to_acc(vl[int64_t(t.pix[0]) * row_stride])) : A(0);
        const A v2 = t.ok[1] ? A(to_acc(vl[int64_t(t.pix[1]) * row_stride])) : A(0);
        const A v3 = t.ok[2] ? A(to_acc(vl[int64_t(t.pix[2]) * row_stride])) : A(0);
        const A v4 = t.ok[3] ? A(to_acc(vl[int64_t(t.pix[3]) * row_stride])) : A(0);
        acc += a * (t.hh * t.hw * v1 + t.hh * t.lw * v2 + t.lh * t.hw * v3 + t.lh * t.lw * v4);
      }
    }
    out[idx] = from_acc<TV>(acc);
  }
}

// A sub-group of `width` lanes (power of two, <= 64) owns one (b, q, m) row and
// strides over its channels; `64/width` rows share a wave.
template <typename TV, typename TL, typename TG>
__global__ void __launch_bounds__(256)
msda_bwd_generic_kernel(const TV* __restrict__ value, const int64_t* __restrict__ shapes,
                        const int64_t* __restrict__ lsi, const TL* __restrict__ loc,
                        const TL* __restrict__ attn, const TV* __restrict__ grad_out,
                        TG* __restrict__ grad_value, TL* __restrict__ grad_loc,
                        TL* __restrict__ grad_attn, MsdaDims d, int width, int only_if_not_packed) {
  using A = acc_t<TV>;
  if (only_if_not_packed && levels_packed(shapes, lsi, d.L, d.S)) return;
  const int groups_per_block = blockDim.x / width;
  const int gib = threadIdx.x / width;
  const int lig = threadIdx.x % width;
  const int64_t rows = int64_t(d.B) * d.Lq * d.M;
  const int64_t row_stride = int64_t(d.M) * d.D;
  for (int64_t rm = int64_t(blockIdx.x) * groups_per_block + gib; rm < rows;
       rm += int64_t(gridDim.x) * groups_per_block) {
    const int m = int(rm % d.M);
    const int64_t r = rm / d.M;
    const int b = int(r / d.Lq);
    const int64_t wbase = rm * d.L * d.P;
    const TV* g = grad_out + rm * d.D;
    for (int l = 0; l < d.L; ++l) {
      const int H = int(shapes[2 * l]), W = int(shapes[2 * l + 1]);
      const int64_t lvl = (int64_t(b) * d.S + lsi[l]) * row_stride + int64_t(m) * d.D;
      for (int k = 0; k < d.P; ++k) {
        const int64_t wi = wbase + l * d.P + k;
        const A x = A(to_acc(loc[2 * wi])), y = A(to_acc(loc[2 * wi + 1]));
        const Taps<A> t = make_taps<A>(x, y, H, W);
        A s_ga = A(0), s_gx = A(0), s_gy = A(0);
        if (t.inside) {  // uniform across the sub-group
          const A a = A(to_acc(attn[wi]));
          const A w1 = t.hh * t.hw, w2 = t.hh * t.lw, w3 = t.lh * t.hw, w4 = t.lh * t.lw;
          for (int c = lig; c < d.D; c += width) {
            const A top = A(to_acc(g[c]));
            const A tg = top * a;
            A gh = A(0), gw = A(0), v1 = A(0), v2 = A(0), v3 = A(0), v4 = A(0);
            if (t.ok[0]) {
              const int64_t o = lvl + int64_t(t.pix[0]) * row_stride + c;
              v1 = A(to_acc(value[o])); gh -= t.hw * v1; gw -= t.hh * v1;
              atomic_add(grad_value + o, TG(w1 * tg));
            }
            if (t.ok[1]) {
              const int64_t o = lvl + int64_t(t.pix[1]) * row_stride + c;
              v2 = A(to_acc(value[o])); gh -= t.lw * v2; gw += t.hh * v2;
              atomic_add(grad_value + o, TG(w2 * tg));
            }
            if (t.ok[2]) {
              const int64_t o = lvl + int64_t(t.pix[2]) * row_stride + c;
              v3 = A(to_acc(value[o])); gh += t.hw * v3; gw -= t.lh * v3;
              atomic_add(grad_value + o, TG(w3 * tg));
            }
            if (t.ok[3]) {
              const int64_t o = lvl + int64_t(t.pix[3]) * row_stride + c;
              v4 = A(to_acc(value[o])); gh += t.lw * v4; gw += t.lh * v4;
              atomic_add(grad_value + o, TG(w4 * tg));
            }
            s_ga += top * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
            s_gx += A(W) * gw * tg;
            s_gy += A(H) * gh * tg;
          }
        }
        for (int off = width >> 1; off > 0; off >>= 1) {
          s_ga += __shfl_xor(s_ga, off, kWave);
          s_gx += __shfl_xor(s_gx, off, kWave);
          s_gy += __shfl_xor(s_gy, off, kWave);
        }
        if (lig == 0) {
          grad_attn[wi] = from_acc<TL>(acc_t<TL>(s_ga));
          grad_loc[2 * wi] = from_acc<TL>(acc_t<TL>(s_gx));
          grad_loc[2 * wi + 1] = from_acc<TL>(acc_t<TL>(s_gy));
        }
      }
    }
  }
}

// fp32 accumulation image -> 16-bit grad_value
template <typename TV>
__global__ void __launch_bounds__(256)
convert_f32_kernel(const float* __restrict__ src, TV* __restrict__ dst, int64_t n,
                   const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi, int L, int S) {
  // shapes != nullptr: run only when the levels are NOT packed (general-path tail)
  if (shapes != nullptr && levels_packed(shapes, lsi, L, S)) return;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += int64_t(gridDim.x) * blockDim.x)
    dst[i] = from_acc<TV>(src[i]);
}

static int grid_for(int64_t work_items, int per_block) {
  int64_t blocks = (work_items + per_block - 1) / per_block;
  const int64_t cap = 256 * 8 * 4;  // CUs x resident blocks, then grid-stride
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return int(blocks);
}

template <typename TV, typename TL>
static int launch_fwd_generic(const void* value, const int64_t* shapes, const int64_t* lsi,
                              const void* loc, const void* attn, void* out, MsdaDims d,
                              hipStream_t stream) {
  const int64_t total = int64_t(d.B) * d.Lq * d.M * d.D;
  hipLaunchKernelGGL((msda_fwd_generic_kernel<TV, TL>), dim3(grid_for(total, 256)), dim3(256), 0,
                     stream, (const TV*)value, shapes, lsi, (const TL*)loc, (const TL*)attn,
                     (TV*)out, d);
  return check_launch("msda_fwd_generic");
}

int msda_forward_generic(int vdt, int ldt, const void* value, const int64_t* shapes,
                         const int64_t* lsi, const void* loc, const void* attn, void* out,
                         MsdaDims d, hipStream_t stream) {
  if (vdt == VNX_F32) return launch_fwd_generic<float, float>(value, shapes, lsi, loc, attn, out, d, stream);
  if (vdt == VNX_F64) return launch_fwd_generic<double, double>(value, shapes, lsi, loc, attn, out, d, stream);
  if (vdt == VNX_BF16 && ldt == VNX_BF16) return launch_fwd_generic<bf16_t, bf16_t>(value, shapes, lsi, loc, attn, out, d, stream);
  if (vdt == VNX_BF16 && ldt == VNX_F32) return launch_fwd_generic<bf16_t, float>(value, shapes, lsi, loc, attn, out, d, stream);
  if (vdt == VNX_F16 && ldt == VNX_F16) return launch_fwd_generic<f16_t, f16_t>(value, shapes, lsi, loc, attn, out, d, stream);
  if (vdt == VNX_F16 && ldt == VNX_F32) return launch_fwd_generic<f16_t, float>(value, shapes, lsi, loc, attn, out, d, stream);
  set_error("msda_forward: unsupported dtype pair (%d, %d)", vdt, ldt);
  return VNX_ERR_INVALID_ARGUMENT;
}

template <typename TV, typename TL, typename TG>
static int launch_bwd_generic(const void* value, const int64_t* shapes, const int64_t* lsi,
                              const void* loc, const void* attn, const void* grad_out,
                              void* gv_acc, void* grad_loc, void* grad_attn, MsdaDims d,
                              int only_if_not_packed, hipStream_t stream) {
  int width = 1;
  while (width < d.D && width < kWave) width <<= 1;
  const int64_t rows = int64_t(d.B) * d.Lq * d.M;
  hipLaunchKernelGGL((msda_bwd_generic_kernel<TV, TL, TG>), dim3(grid_for(rows, 256 / width)),
                     dim3(256), 0, stream, (const TV*)value, shapes, lsi, (const TL*)loc,
                     (const TL*)attn, (const TV*)grad_out, (TG*)gv_acc, (TL*)grad_loc,
                     (TL*)grad_attn, d, width, only_if_not_packed);
  return check_launch("msda_bwd_generic");
}

template <typename TV>
static int launch_convert(const void* src, void* dst, int64_t n, const int64_t* shapes,
                          const int64_t* lsi, int L, int S, hipStream_t stream) {
  hipLaunchKernelGGL((convert_f32_kernel<TV>), dim3(grid_for(n, 256)), dim3(256), 0, stream,
                     (const float*)src, (TV*)dst, n, shapes, lsi, L, S);
  return check_launch("convert_f32");
}

// grad_value (or the fp32 workspace standing in for it) must already be zero.
int msda_backward_generic(int vdt, int ldt, const void* value, const int64_t* shapes,
                          const int64_t* lsi, const void* loc, const void* attn,
                          const void* grad_out, void* gv_acc, void* grad_loc, void* grad_attn,
                          MsdaDims d, int only_if_not_packed, hipStream_t stream) {
  if (vdt == VNX_F32) return launch_bwd_generic<float, float, float>(value, shapes, lsi, loc, attn, grad_out, gv_acc, grad_loc, grad_attn, d, only_if_not_packed, stream);
  if (vdt == VNX_F64) return launch_bwd_generic<double, double, double>(value, shapes, lsi, loc, attn, grad_out, gv_acc, grad_loc, grad_attn, d, only_if_not_packed, stream);
  if (vdt == VNX_BF16 && ldt == VNX_BF16) return launch_bwd_generic<bf16_t, bf16_t, float>(value, shapes, lsi, loc, attn, grad_out, gv_acc, grad_loc, grad_attn, d, only_if_not_packed, stream);
  if (vdt == VNX_BF16 && ldt == VNX_F32) return launch_bwd_generic<bf16_t, float, float>(value, shapes, lsi, loc, attn, grad_out, gv_acc, grad_loc, grad_attn, d, only_if_not_packed, stream);
  if (vdt == VNX_F16 && ldt == VNX_F16) return launch_bwd_generic<f16_t, f16_t, float>(value, shapes, lsi, loc, attn, grad_out, gv_acc, grad_loc, grad_attn, d, only_if_not_packed, stream);
  if (vdt == VNX_F16 && ldt == VNX_F32) return launch_bwd_generic<f16_t, float, float>(value, shapes, lsi, loc, attn, grad_out, gv_acc, grad_loc, grad_attn, d, only_if_not_packed, stream);
  set_error("msda_backward: unsupported dtype pair (%d, %d)", vdt, ldt);
  return VNX_ERR_INVALID_ARGUMENT;
}

// shapes == nullptr: unconditional; otherwise only when the levels are not packed
int convert_f32_to(int vdt, const void* src, void* dst, int64_t n, const int64_t* shapes,
                   const int64_t* lsi, int L, int S, hipStream_t stream) {
  if (vdt == VNX_BF16) return launch_convert<bf16_t>(src, dst, n, shapes, lsi, L, S, stream);
  if (vdt == VNX_F16) return launch_convert<f16_t>(src, dst, n, shapes, lsi, L, S, stream);
  set_error("convert_f32_to: dtype %d is not 16-bit", vdt);
  return VNX_ERR_INVALID_ARGUMENT;
}

// ---- helper of the general (not packed) path: zero-fill that runs only when the levels are NOT packed ----
typedef float zero_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
zero_if_not_packed_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi, int L,
                          int S, zero_f4* __restrict__ dst, int64_t n16, unsigned char* tail,
                          int tail_bytes) {
  if (levels_packed(shapes, lsi, L, S)) return;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16;
       i += int64_t(gridDim.x) * blockDim.x)
    dst[i] = zero_f4{0.f, 0.f, 0.f, 0.f};
  if (blockIdx.x == 0 && int(threadIdx.x) < tail_bytes) tail[threadIdx.x] = 0;
}

int zero_if_not_packed(const int64_t* shapes, const int64_t* lsi, int L, int S, void* dst,
                       size_t bytes, hipStream_t stream) {
  const int64_t n16 = int64_t(bytes / 16);
  int64_t blocks = (n16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_if_not_packed_kernel, dim3(uint32_t(blocks)), dim3(256), 0, stream, shapes,
                     lsi, L, S, (zero_f4*)dst, n16, (unsigned char*)dst + n16 * 16, int(bytes % 16));
  return check_launch("zero_if_not_packed");
}

}  // namespace vnx
