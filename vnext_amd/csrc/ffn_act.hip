// ffn_act.hip -- the middle of a transformer FFN, h -> dropout(relu(h + bias)), in ONE in-place pass, and its backward
// with the bias gradient in the same pass (the glue of the 6 + 6 layer stack; VERDICT r3
// item 8).
//
// Reference: `src2 = self.linear2(self.dropout2(self.activation(self.linear1(src))))`
// (projects/SeqFormer/seqformer/models/deformable_transformer.py:226-229 encoder layer, :330-338 decoder layer and its
// `_box` twin; IDOL's copy is the same).  ATen runs, on the [rows, d_ffn] hidden tensor -- 51 000 x 1 024 fp32 = 209 MB in
// the encoder of a two-clip training step -- relu (read + write), dropout (read + write + a byte mask), and in the
// backward masked_scale (read grad + mask, write), threshold_backward (read grad + activation, write) and the column
// sum for linear1's bias (read): 5.25 + 6.25 passes over the tensor.  Here: forward 1 pass in place (the GEMM leaves h
// without its bias; the bias is added here), backward 1 pass in place + the column sums:
//   * y = keep ? max(h + b, 0) / (1 - p) : 0, keep(element) = hash(seed, element) >= p 2^32 -- the hash of add_norm.hip;
//   * y > 0  <=>  the element passed the ReLU AND was kept, so the backward needs neither a mask nor the seed:
//     g_h = y > 0 ? g / (1 - p) : 0, read from the activation linear2's weight gradient needs anyway;
//   * grad_bias[c] = sum_rows g_h[row][c]: per-workgroup partial rows, then a fixed-order reduction (deterministic).
// Round 6: h, the gradient that comes in and the one that goes out may be bf16 (`dtype`; under torch.autocast(bfloat16) the
// GEMMs around this pass emit and take bf16); the bias, its gradient and the partial sums stay fp32, the arithmetic is fp32.
#include "vnx_common.h"

#include <algorithm>

namespace vnx {

typedef float ffn_f4 __attribute__((ext_vector_type(4)));

constexpr int kFaMaxBlocks = 2048;     // workgroups of the backward = partial rows of the bias gradient (8 waves per SIMD at 1 024 columns)
constexpr int kFaMaxCols = 4096;       // columns (d_ffn) at most: one float4 per thread, 1 024 threads

__device__ __forceinline__ uint32_t fa_hash(uint32_t idx, uint32_t seed_lo, uint32_t seed_hi) {      // = an_hash (add_norm.hip)
  uint32_t h = idx ^ seed_lo;
  h *= 0xcc9e2d51u; h = (h << 15) | (h >> 17); h *= 0x1b873593u;
  h ^= seed_hi;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ void fa_effective_seed(uint32_t& lo, uint32_t& hi, const unsigned long long* seed_device) {
  if (seed_device == nullptr) return;                                  // = an_effective_seed (add_norm.hip)
  const unsigned long long s = *seed_device;
  uint32_t a = uint32_t(s) * 0x9E3779B1u, b = (uint32_t(s >> 32) + 0x7F4A7C15u) * 0x85EBCA77u;
  a ^= a >> 15; b ^= b >> 13;
  lo ^= a * 0xC2B2AE3Du;
  hi ^= b * 0x27D4EB2Fu + a;
}

// in place over h [rows, cols]; thread = 4 consecutive columns (blockIdx.x * 256 + threadIdx.x), rows blockIdx.y,
// blockIdx.y + gridDim.y, ...: no division anywhere, the bias quad is loaded once per thread
template <typename T>
__global__ void __launch_bounds__(256)
bias_relu_dropout_fwd_kernel(T* __restrict__ h, const float* __restrict__ bias, const uint8_t* __restrict__ row_zero,
                             int64_t rows, int cols4, int relu, uint32_t threshold, float scale, uint32_t seed_lo,
                             uint32_t seed_hi, const unsigned long long* __restrict__ seed_device) {
  const int c4 = int(blockIdx.x) * 256 + int(threadIdx.x);
  if (c4 >= cols4) return;
  if (threshold != 0u) fa_effective_seed(seed_lo, seed_hi, seed_device);
  const ffn_f4 b = bias != nullptr ? *reinterpret_cast<const ffn_f4*>(bias + c4 * 4) : ffn_f4{0.f, 0.f, 0.f, 0.f};
  for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
    const int64_t i = row * cols4 + c4;                                // float4 index
    if (row_zero != nullptr && row_zero[row]) {                        // a padding row: zeros, whatever h holds (masked_fill)
      row4_store<T>(h + i * 4, ffn_f4{0.f, 0.f, 0.f, 0.f});
      continue;
    }
    ffn_f4 v = row4_load<T>(h + i * 4) + b;
    // (a comparison, not fmaxf: fmaxf(NaN, 0) is 0, ATen's relu(NaN) is NaN -- a diverging run must stay visible)
    if (relu) { v.x = v.x <= 0.f ? 0.f : v.x; v.y = v.y <= 0.f ? 0.f : v.y; v.z = v.z <= 0.f ? 0.f : v.z; v.w = v.w <= 0.f ? 0.f : v.w; }
    if (threshold != 0u) {
      const uint32_t base = uint32_t(i) * 4u;                          // element index mod 2^32 ...
      const uint32_t hi = seed_hi ^ uint32_t(uint64_t(i) >> 30);       // ... and what lies above it
      v.x = fa_hash(base + 0u, seed_lo, hi) >= threshold ? v.x * scale : 0.f;
      v.y = fa_hash(base + 1u, seed_lo, hi) >= threshold ? v.y * scale : 0.f;
      v.z = fa_hash(base + 2u, seed_lo, hi) >= threshold ? v.z * scale : 0.f;
      v.w = fa_hash(base + 3u, seed_lo, hi) >= threshold ? v.w * scale : 0.f;
    }
    row4_store<T>(h + i * 4, v);
  }
}

// g [rows, cols] -> out (out may be g itself: every element is read before it is written, by the same thread); thread t owns
// columns 4t..4t+3, a workgroup (cols / 4 threads) walks the rows blockIdx.x, blockIdx.x + gridDim.x, ...;
// partial[blockIdx.x][cols] = this workgroup's column sums
template <typename T>
__global__ void __launch_bounds__(1024)
bias_relu_dropout_bwd_kernel(const T* g, const T* __restrict__ y, const uint8_t* __restrict__ row_zero, T* out,
                             float* __restrict__ partial, int64_t rows, int cols, float scale) {
  const int c = int(threadIdx.x) * 4;
  ffn_f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const int64_t at = row * cols + c;
    ffn_f4 gv = row4_load<T>(g + at);
    if (row_zero != nullptr && row_zero[row]) gv = ffn_f4{0.f, 0.f, 0.f, 0.f};
    if (y != nullptr) {                         // ReLU (+ dropout): y > 0 <=> passed and kept
      // written as !(y <= 0), ATen's threshold_backward: a NaN activation lets its gradient through instead of
      // silently zeroing it (a diverging run must stay visible downstream)
      const ffn_f4 yv = row4_load<T>(y + at);
      gv.x = !(yv.x <= 0.f) ? gv.x * scale : 0.f;
      gv.y = !(yv.y <= 0.f) ? gv.y * scale : 0.f;
      gv.z = !(yv.z <= 0.f) ? gv.z * scale : 0.f;
      gv.w = !(yv.w <= 0.f) ? gv.w * scale : 0.f;
    }
    acc += gv;
    row4_store<T>(out + at, gv);
  }
  if (partial != nullptr) *reinterpret_cast<ffn_f4*>(partial + int64_t(blockIdx.x) * cols + c) = acc;
}

// grad_bias[c] = sum_b partial[b][c] in a fixed order: 32 row slices per column, then a tree over the slices
__global__ void __launch_bounds__(1024)
column_partial_sum_kernel(const float* __restrict__ partial, float* __restrict__ out, int blocks, int cols) {
  __shared__ float red[32][32];
  const int c = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int col = int(blockIdx.x) * 32 + c;
  float s0 = 0.f, s1 = 0.f;
  if (col < cols) {
    int b = slice;
    for (; b + 32 < blocks; b += 64) { s0 += partial[int64_t(b) * cols + col]; s1 += partial[int64_t(b + 32) * cols + col]; }
    if (b < blocks) s0 += partial[int64_t(b) * cols + col];
  }
  const float s = s0 + s1;
  red[slice][c] = s;
  __syncthreads();
#pragma unroll
  for (int step = 16; step > 0; step >>= 1) {
    if (slice < step) red[slice][c] += red[slice + step][c];
    __syncthreads();
  }
  if (slice == 0 && col < cols) out[col] = red[0][c];
}

static int fa_check(const char* who, int dtype, int64_t rows, int cols, float p) {
  if (dtype != VNX_F32 && dtype != VNX_BF16 && dtype != VNX_F16) { set_error("%s: f32, bf16 or f16 rows (got dtype %d)", who, dtype); return VNX_ERR_UNSUPPORTED; }
  if (cols <= 0 || (cols & 3) || cols > kFaMaxCols) {
    set_error("%s: built for rows of 4..%d channels, a multiple of 4 (got %d)", who, kFaMaxCols, cols);
    return VNX_ERR_UNSUPPORTED;
  }
  if (rows < 0 || rows >= (int64_t(1) << 40) || !(p >= 0.f && p < 1.f)) {
    set_error("%s: bad sizes rows=%lld p=%g", who, (long long)rows, double(p));
    return VNX_ERR_INVALID_ARGUMENT;
  }
  return VNX_OK;
}

static uint32_t fa_threshold(float p) {      // keep iff hash >= threshold:  P(drop) = threshold / 2^32
  const double t = double(p) * 4294967296.0;
  return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 0xffffffffu : uint32_t(t + 0.5));
}

}  // namespace vnx

using namespace vnx;

extern "C" size_t vnx_bias_relu_dropout_partial_bytes(int channels) {
  return size_t(kFaMaxBlocks) * size_t(channels > 0 ? channels : 0) * sizeof(float);
}

extern "C" int vnx_bias_relu_dropout_forward(int dtype, void* h, const void* bias, const unsigned char* row_zero, long long rows,
                                             int channels, int relu, float p, unsigned long long seed,
                                             const unsigned long long* seed_device, void* hip_stream) {
  if (int st = fa_check("vnx_bias_relu_dropout_forward", dtype, rows, channels, p)) return st;
  if (!relu && p > 0.f) {
    set_error("vnx_bias_relu_dropout_forward: dropout without the ReLU is not built (the backward reads the mask off y > 0)");
    return VNX_ERR_UNSUPPORTED;
  }
  if (rows == 0) return VNX_OK;
  if (!h) {
    set_error("vnx_bias_relu_dropout_forward: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const int cols4 = channels / 4;
  const dim3 grid(uint32_t((cols4 + 255) / 256), uint32_t(std::min<int64_t>(rows, 16384)));
#define VNX_FA_FWD(T)                                                                                                  \
  hipLaunchKernelGGL(bias_relu_dropout_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)hip_stream, (T*)h,              \
                     (const float*)bias, (const uint8_t*)row_zero, int64_t(rows), channels / 4, relu, fa_threshold(p), \
                     1.f / (1.f - p), uint32_t(seed), uint32_t(seed >> 32), seed_device)
  if (dtype == VNX_BF16) VNX_FA_FWD(bf16_t); else if (dtype == VNX_F16) VNX_FA_FWD(f16_t); else VNX_FA_FWD(float);
#undef VNX_FA_FWD
  return check_launch("bias_relu_dropout_fwd");
}

extern "C" int vnx_bias_relu_dropout_backward(int dtype, const void* grad, const void* y, const unsigned char* row_zero,
                                              void* grad_h, void* grad_bias, void* partial, long long rows, int channels,
                                              float p, void* hip_stream) {
  if (int st = fa_check("vnx_bias_relu_dropout_backward", dtype, rows, channels, p)) return st;
  if (grad_bias && !partial) {
    set_error("vnx_bias_relu_dropout_backward: grad_bias needs the partial scratch (vnx_bias_relu_dropout_partial_bytes)");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  int blocks = int(std::min<int64_t>(kFaMaxBlocks, rows));
  if (rows > 0) {
    if (!grad || !grad_h || (!y && p > 0.f)) {      // y == null: the forward ran without the ReLU
      set_error("vnx_bias_relu_dropout_backward: null pointer argument");
      return VNX_ERR_INVALID_ARGUMENT;
    }
#define VNX_FA_BWD(T)                                                                                                         \
    hipLaunchKernelGGL(bias_relu_dropout_bwd_kernel<T>, dim3(uint32_t(blocks)), dim3(channels / 4), 0, stream, (const T*)grad, \
                       (const T*)y, (const uint8_t*)row_zero, (T*)grad_h, grad_bias ? (float*)partial : (float*)nullptr,       \
                       int64_t(rows), channels, 1.f / (1.f - p))
    if (dtype == VNX_BF16) VNX_FA_BWD(bf16_t); else if (dtype == VNX_F16) VNX_FA_BWD(f16_t); else VNX_FA_BWD(float);
#undef VNX_FA_BWD
  } else {
    blocks = 0;
  }
  if (grad_bias)
    hipLaunchKernelGGL(column_partial_sum_kernel, dim3(uint32_t((channels + 31) / 32)), dim3(1024), 0, stream,
                       (const float*)partial, (float*)grad_bias, blocks, channels);
  return check_launch("bias_relu_dropout_bwd");
}
