// add_norm.hip -- y = LayerNorm(x + dropout(r)): the residual / dropout / normalisation chain that closes every
// sub-layer of the deformable transformer, in ONE pass (the training step; round-1 review
// item "fuse residual + dropout + LayerNorm chains").
//
// Reference (three ATen launches forward, four backward, per site; 30 sites per step):
//   projects/SeqFormer/seqformer/models/deformable_transformer.py:201-236  `src = src + self.dropout1(src2);
//   src = self.norm1(src)` ... (encoder layer), :286-385 (decoder layer: norm1 / norm2 / norm3 and the *_box copies);
//   projects/IDOL/idol/models/deformable_transformer.py: the same layers.
// Here: one wave per row of 256 channels (the models' hidden size; 64 lanes x float4), mean and variance by two wave
// reductions over registers (two-pass, as accurate as ATen's Welford), no LDS, no barrier.  The dropout mask is not
// stored: keep(element) = hash(seed, element index) >= p * 2^32, recomputed by the backward from the same seed.
// Forward traffic: x, r read, y and z = x + dropout(r) written (z is what the backward needs; r is then dead);
// backward: grad_y, z read, grad_x, grad_r written; the gamma / beta gradients leave as per-workgroup partial rows
// that a second, tiny kernel adds in a fixed order (deterministic, no atomics).
// Round 6: the BRANCH r (and its gradient) may be bf16 -- under torch.autocast(bfloat16) the Linear or attention output that
// arrives here is bf16 while the residual stream x, the LayerNorm and its output are fp32, exactly the eager chain's types
// (x + dropout(r) promotes; autocast runs layer_norm in fp32).  `dtype` of the C entry points names r's type; everything else
// is fp32.  (Until then the op fell to the eager chain under autocast: 3 + 4 launches per site, 30 sites per step.)
#include "vnx_common.h"

#include <algorithm>

namespace vnx {

typedef float float4_t __attribute__((ext_vector_type(4)));

constexpr int kAnC = 256;            // channels per row: one float4 per lane
constexpr int kAnWaves = 4;          // rows per workgroup pass
constexpr int kAnMaxBlocks = 1024;   // workgroups of the backward = rows of the partial-gradient buffer
constexpr int kAnParts = 3;          // partial rows per workgroup: gamma, beta, and the bias folded into r (ABI 11)
static_assert(kAnParts <= kAnWaves, "one wave per partial row");

// murmur3-style mix of (element index, 64-bit seed) -> 32 uniform bits
__device__ __forceinline__ uint32_t an_hash(uint32_t idx, uint32_t seed_lo, uint32_t seed_hi) {
  uint32_t h = idx ^ seed_lo;
  h *= 0xcc9e2d51u; h = (h << 15) | (h >> 17); h *= 0x1b873593u;
  h ^= seed_hi;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}

// Effective seed of a launch: the host's 64 bits, mixed with a 64-bit word read from DEVICE memory when the caller
// passes one.  A host integer is baked into a captured hipGraph -- every replay of a captured training step would drop
// the same elements -- whereas a device word bumped once per step (by a captured `add_`) gives every replay
// fresh masks; forward and backward of one step read the same word, so the backward still recomputes the mask.
__device__ __forceinline__ void an_effective_seed(uint32_t& lo, uint32_t& hi, const unsigned long long* seed_device) {
  if (seed_device == nullptr) return;
  const unsigned long long s = *seed_device;                       // wave-uniform (scalar load)
  uint32_t a = uint32_t(s) * 0x9E3779B1u, b = (uint32_t(s >> 32) + 0x7F4A7C15u) * 0x85EBCA77u;
  a ^= a >> 15; b ^= b >> 13;
  lo ^= a * 0xC2B2AE3Du;
  hi ^= b * 0x27D4EB2Fu + a;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// z = x + keep * r * scale for this lane's four channels of `row`
__device__ __forceinline__ float4_t an_residual(float4_t x, float4_t r, int64_t row, int lane, uint32_t threshold,
                                                float scale, uint32_t seed_lo, uint32_t seed_hi) {
  if (threshold == 0u) return x + r;
  const uint32_t base = uint32_t(row) * uint32_t(kAnC) + uint32_t(lane) * 4u;
  const uint32_t hi = seed_hi ^ uint32_t(uint64_t(row) >> 24);     // rows beyond 2^24 (4 G elements) still differ
  float4_t z = x;
  if (an_hash(base + 0u, seed_lo, hi) >= threshold) z.x += r.x * scale;
  if (an_hash(base + 1u, seed_lo, hi) >= threshold) z.y += r.y * scale;
  if (an_hash(base + 2u, seed_lo, hi) >= threshold) z.z += r.z * scale;
  if (an_hash(base + 3u, seed_lo, hi) >= threshold) z.w += r.w * scale;
  return z;
}

template <typename TR>
__global__ void __launch_bounds__(64 * kAnWaves)
add_dropout_layernorm_fwd_kernel(const float* __restrict__ x, const TR* __restrict__ r, const float* __restrict__ r_bias,
                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                 float* __restrict__ y, float* __restrict__ z_out, float* __restrict__ stats,
                                 int64_t rows, uint32_t threshold, float scale, float eps, uint32_t seed_lo,
                                 uint32_t seed_hi, const unsigned long long* __restrict__ seed_device) {
  const int lane = threadIdx.x & 63;
  const int64_t row = int64_t(blockIdx.x) * kAnWaves + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (threshold != 0u) an_effective_seed(seed_lo, seed_hi, seed_device);
  const int64_t at = row * kAnC + lane * 4;
  const float4_t xv = *reinterpret_cast<const float4_t*>(x + at);
  float4_t rv = row4_load<TR>(r + at);
  if (r_bias != nullptr) rv += *reinterpret_cast<const float4_t*>(r_bias + lane * 4);      // r = Linear output WITHOUT its bias
  const float4_t g = *reinterpret_cast<const float4_t*>(gamma + lane * 4);
  const float4_t b = *reinterpret_cast<const float4_t*>(beta + lane * 4);
  const float4_t z = an_residual(xv, rv, row, lane, threshold, scale, seed_lo, seed_hi);
  const float mean = wave_sum((z.x + z.y) + (z.z + z.w)) * (1.f / kAnC);
  const float4_t c = z - mean;
  const float var = wave_sum((c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w)) * (1.f / kAnC);
  const float rstd = rsqrtf(var + eps);
  __builtin_nontemporal_store(c * rstd * g + b, reinterpret_cast<float4_t*>(y + at));
  *reinterpret_cast<float4_t*>(z_out + at) = z;
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// grad_x = dz, grad_r = keep * scale * dz, partial[block] = {sum_rows g * xhat, sum_rows g, sum_rows grad_r} over this
// workgroup's rows (the third row is the gradient of the bias folded into r, ABI 11)
template <typename TR>
__global__ void __launch_bounds__(64 * kAnWaves)
add_dropout_layernorm_bwd_kernel(const float* __restrict__ grad_y, const float* __restrict__ z,
                                 const float* __restrict__ stats, const float* __restrict__ gamma,
                                 float* __restrict__ grad_x, TR* __restrict__ grad_r, float* __restrict__ partial,
                                 int64_t rows, uint32_t threshold, float scale, uint32_t seed_lo, uint32_t seed_hi,
                                 const unsigned long long* __restrict__ seed_device) {
  __shared__ float4_t red[kAnParts][kAnWaves][64];
  if (threshold != 0u) an_effective_seed(seed_lo, seed_hi, seed_device);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4_t g = *reinterpret_cast<const float4_t*>(gamma + lane * 4);
  float4_t dg = {0.f, 0.f, 0.f, 0.f}, db = dg, dbias = dg;
  for (int64_t row = int64_t(blockIdx.x) * kAnWaves + wave; row < rows; row += int64_t(gridDim.x) * kAnWaves) {
    const int64_t at = row * kAnC + lane * 4;
    const float4_t gy = *reinterpret_cast<const float4_t*>(grad_y + at);
    const float4_t zv = *reinterpret_cast<const float4_t*>(z + at);
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float4_t xh = (zv - mean) * rstd;
    const float4_t gg = gy * g;
    const float m1 = wave_sum((gg.x + gg.y) + (gg.z + gg.w)) * (1.f / kAnC);
    const float m2 = wave_sum((gg.x * xh.x + gg.y * xh.y) + (gg.z * xh.z + gg.w * xh.w)) * (1.f / kAnC);
    const float4_t dz = (gg - m1 - xh * m2) * rstd;
    dg += gy * xh;
    db += gy;
    __builtin_nontemporal_store(dz, reinterpret_cast<float4_t*>(grad_x + at));
    float4_t dr = dz;
    if (threshold != 0u) {
      const uint32_t base = uint32_t(row) * uint32_t(kAnC) + uint32_t(lane) * 4u;
      const uint32_t hi = seed_hi ^ uint32_t(uint64_t(row) >> 24);
      dr.x = an_hash(base + 0u, seed_lo, hi) >= threshold ? dz.x * scale : 0.f;
      dr.y = an_hash(base + 1u, seed_lo, hi) >= threshold ? dz.y * scale : 0.f;
      dr.z = an_hash(base + 2u, seed_lo, hi) >= threshold ? dz.z * scale : 0.f;
      dr.w = an_hash(base + 3u, seed_lo, hi) >= threshold ? dz.w * scale : 0.f;
    }
    dbias += dr;
    row4_store_nt<TR>(grad_r + at, dr);
  }
  red[0][wave][lane] = dg;
  red[1][wave][lane] = db;
  red[2][wave][lane] = dbias;
  __syncthreads();
  if (wave < kAnParts) {      // wave 0 adds the gamma partials of the four waves, wave 1 the beta partials, wave 2 the bias: fixed order
    float4_t s = red[wave][0][lane];
#pragma unroll
    for (int w = 1; w < kAnWaves; ++w) s += red[wave][w][lane];
    *reinterpret_cast<float4_t*>(partial + (int64_t(blockIdx.x) * kAnParts + wave) * kAnC + lane * 4) = s;
  }
}

// grad_gamma[c] = sum_b partial[b][0][c], grad_beta[c] = sum_b partial[b][1][c], in a fixed order (deterministic).
// 16 workgroups x 1 024 threads: a workgroup owns 32 of the 512 columns (gamma | beta), thread (slice, column) adds the
// partial rows  slice, slice + 32, ...  -- 32 independent chains per column -- and the slices meet in a fixed-order tree.
// (Round 2's form walked all 1 024 partial rows with four chains per column on 4 workgroups: 122 us per call at the
// encoder's 25 500 rows, seven times the backward kernel it finishes -- 30 calls per training step; rocprofv3,
// profiles/r03_shapes.)
constexpr int kPgBlocks = 24, kPgCols = kAnParts * kAnC / kPgBlocks, kPgSlices = 1024 / kPgCols;     // 24, 32, 32
static_assert(kPgCols * kPgBlocks == kAnParts * kAnC && kPgCols * kPgSlices == 1024, "whole columns, whole slices");
__global__ void __launch_bounds__(1024)
layernorm_param_grad_kernel(const float* __restrict__ partial, float* __restrict__ grad_gamma,
                            float* __restrict__ grad_beta, float* __restrict__ grad_bias, int blocks) {
  __shared__ float red[kPgSlices][kPgCols];
  const int c = threadIdx.x % kPgCols, slice = threadIdx.x / kPgCols;
  const int col = int(blockIdx.x) * kPgCols + c;                    // 0..767: gamma | beta | bias
  float s = 0.f;
  if (col < 2 * kAnC || grad_bias != nullptr)
    for (int b = slice; b < blocks; b += kPgSlices) s += partial[int64_t(b) * kAnParts * kAnC + col];
  red[slice][c] = s;
  __syncthreads();
#pragma unroll
  for (int step = kPgSlices / 2; step > 0; step >>= 1) {
    if (slice < step) red[slice][c] += red[slice + step][c];
    __syncthreads();
  }
  if (slice == 0) {
    if (col < kAnC) grad_gamma[col] = red[0][c];
    else if (col < 2 * kAnC) grad_beta[col - kAnC] = red[0][c];
    else if (grad_bias != nullptr) grad_bias[col - 2 * kAnC] = red[0][c];
  }
}

static int an_check(const char* who, int dtype, int64_t rows, int channels, float p) {
  if (dtype != VNX_F32 && dtype != VNX_BF16 && dtype != VNX_F16) {
    set_error("%s: the branch is f32, bf16 or f16 (got dtype %d)", who, dtype);
    return VNX_ERR_UNSUPPORTED;
  }
  if (channels != kAnC) { set_error("%s: built for %d channels per row (got %d)", who, kAnC, channels); return VNX_ERR_UNSUPPORTED; }
  if (rows < 0 || rows >= (int64_t(1) << 40) || !(p >= 0.f && p < 1.f)) {
    set_error("%s: bad sizes rows=%lld p=%g", who, (long long)rows, double(p));
    return VNX_ERR_INVALID_ARGUMENT;
  }
  return VNX_OK;
}

static uint32_t an_threshold(float p) {      // keep iff hash >= threshold:  P(drop) = threshold / 2^32
  const double t = double(p) * 4294967296.0;
  return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 0xffffffffu : uint32_t(t + 0.5));
}

}  // namespace vnx

using namespace vnx;

extern "C" size_t vnx_add_dropout_layernorm_partial_bytes(void) { return size_t(kAnMaxBlocks) * kAnParts * kAnC * 4; }

extern "C" int vnx_add_dropout_layernorm_forward(int dtype, const void* x, const void* r, const void* r_bias, const void* gamma,
                                                 const void* beta, void* y, void* z, void* stats, long long rows,
                                                 int channels, float p, float eps, unsigned long long seed,
                                                 const unsigned long long* seed_device, void* hip_stream) {
  if (int st = an_check("vnx_add_dropout_layernorm_forward", dtype, rows, channels, p)) return st;
  if (rows == 0) return VNX_OK;
  if (!x || !r || !gamma || !beta || !y || !z || !stats) {
    set_error("vnx_add_dropout_layernorm_forward: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const int64_t blocks = (rows + kAnWaves - 1) / kAnWaves;
#define VNX_AN_FWD(TR)                                                                                                   \
  hipLaunchKernelGGL(add_dropout_layernorm_fwd_kernel<TR>, dim3(uint32_t(blocks)), dim3(64 * kAnWaves), 0,               \
                     (hipStream_t)hip_stream, (const float*)x, (const TR*)r, (const float*)r_bias, (const float*)gamma,  \
                     (const float*)beta, (float*)y, (float*)z, (float*)stats, int64_t(rows), an_threshold(p),           \
                     1.f / (1.f - p), eps, uint32_t(seed), uint32_t(seed >> 32), seed_device)
  if (dtype == VNX_BF16) VNX_AN_FWD(bf16_t); else if (dtype == VNX_F16) VNX_AN_FWD(f16_t); else VNX_AN_FWD(float);
#undef VNX_AN_FWD
  return check_launch("add_dropout_layernorm_fwd");
}

extern "C" int vnx_add_dropout_layernorm_backward(int dtype, const void* grad_y, const void* z, const void* stats,
                                                  const void* gamma, void* grad_x, void* grad_r, void* grad_gamma,
                                                  void* grad_beta, void* grad_r_bias, void* partial, long long rows, int channels, float p,
                                                  unsigned long long seed, const unsigned long long* seed_device,
                                                  void* hip_stream) {
  if (int st = an_check("vnx_add_dropout_layernorm_backward", dtype, rows, channels, p)) return st;
  if (!grad_gamma || !grad_beta || !partial) {
    set_error("vnx_add_dropout_layernorm_backward: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  int blocks = int(std::min<int64_t>(kAnMaxBlocks, (rows + kAnWaves - 1) / kAnWaves));
  if (rows > 0) {
    if (!grad_y || !z || !stats || !gamma || !grad_x || !grad_r) {
      set_error("vnx_add_dropout_layernorm_backward: null pointer argument");
      return VNX_ERR_INVALID_ARGUMENT;
    }
#define VNX_AN_BWD(TR)                                                                                                   \
    hipLaunchKernelGGL(add_dropout_layernorm_bwd_kernel<TR>, dim3(uint32_t(blocks)), dim3(64 * kAnWaves), 0, stream,       \
                       (const float*)grad_y, (const float*)z, (const float*)stats, (const float*)gamma, (float*)grad_x,   \
                       (TR*)grad_r, (float*)partial, int64_t(rows), an_threshold(p), 1.f / (1.f - p), uint32_t(seed),     \
                       uint32_t(seed >> 32), seed_device)
    if (dtype == VNX_BF16) VNX_AN_BWD(bf16_t); else if (dtype == VNX_F16) VNX_AN_BWD(f16_t); else VNX_AN_BWD(float);
#undef VNX_AN_BWD
  } else {
    blocks = 0;
  }
  hipLaunchKernelGGL(layernorm_param_grad_kernel, dim3(kPgBlocks), dim3(1024), 0, stream, (const float*)partial,
                     (float*)grad_gamma, (float*)grad_beta, (float*)grad_r_bias, blocks);
  return check_launch("add_dropout_layernorm_bwd");
}
