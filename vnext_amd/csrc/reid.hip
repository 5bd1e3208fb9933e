// reid.hip -- IDOL re-identification head: similarity matrix and bi-softmax association
// (SURVEY.md section 8 row a7).
//
// Reference (all tiny PyTorch launches, many of them issued per instance from Python):
//   projects/IDOL/idol/models/tracker.py:229-244   feats = mm(embeds, memo_embeds.t());
//                                                   (softmax(dim=1) + softmax(dim=0)) / 2, or cosine
//   projects/IDOL/idol/models/pos_neg_select.py:47,58-62  per-instance einsum('nc,kc->nk') for the raw
//                                                   dot ("contrast") and the L2-normalised ("aux_consin")
// Here: one MFMA kernel S = A . B^T for the whole [n, C] x [k, C] problem, with the L2
// normalisation (F.normalize, eps 1e-12) optionally fused into the epilogue, and one kernel for
// the bi-softmax.  This is the one GEMM-shaped piece of the hot path, so it runs on the matrix
// cores: v_mfma_f32_16x16x4_f32 -- fp32 in, fp32 accumulate, bit-identical to an fmaf chain --
// because the embeddings are fp32 and the association thresholds (tracker.py:259) are not
// re-tuned for reduced precision.  The problem is far too small to approach any compute peak
// (300 x 300 x 256 = 46 MFLOP); what is bought is launch count.
#include "vnx_common.h"

namespace vnx {

typedef float float4_t __attribute__((ext_vector_type(4)));

// One wave per 16x16 output tile.  Lane l = (r = l & 15, g = l >> 4).  Per 16-wide K chunk the
// lane loads A[row0+r][c0+4g .. +3] and B[col0+r][c0+4g .. +3] (16 B each); MFMA step s uses
// component s, i.e. K index c0 + 4g + s for lane group g -- the same mapping on both operands, so
// the four steps cover the chunk.  D: lane holds D[4g + i][r], i = 0..3.
__global__ void __launch_bounds__(256)
reid_similarity_kernel(const float* __restrict__ A, const float* __restrict__ B,
                       float* __restrict__ out, int n, int k, int C, int lda, int ldb, int ldo,
                       int normalize, int tiles_k) {
  const int lane = threadIdx.x & 63;
  const int wave = int(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int tile_n = wave / tiles_k, tile_k = wave - tile_n * tiles_k;
  if (tile_n * 16 >= n) return;
  const int r = lane & 15, g = lane >> 4;
  const int row = min(tile_n * 16 + r, n - 1);
  const int col = min(tile_k * 16 + r, k - 1);
  const float* a_ptr = A + int64_t(row) * lda + 4 * g;
  const float* b_ptr = B + int64_t(col) * ldb + 4 * g;

  float4_t acc = {0.f, 0.f, 0.f, 0.f};
  float na = 0.f, nb = 0.f;
  int c0 = 0;
  for (; c0 + 16 <= C; c0 += 16) {
    const float4_t a = *reinterpret_cast<const float4_t*>(a_ptr + c0);
    const float4_t b = *reinterpret_cast<const float4_t*>(b_ptr + c0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    na += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    nb += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
  }
  for (; c0 < C; c0 += 4) {  // tail: one K step of 4 per iteration, zero-filled past C
    const int c = c0 + g;
    const float a = c < C ? A[int64_t(row) * lda + c] : 0.f;
    const float b = c < C ? B[int64_t(col) * ldb + c] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    na += a * a;
    nb += b * b;
  }
  if (normalize) {
    // squared norms: sum the four K sub-sets (lanes r, r+16, r+32, r+48)
    na += __shfl_xor(na, 16, 64); na += __shfl_xor(na, 32, 64);
    nb += __shfl_xor(nb, 16, 64); nb += __shfl_xor(nb, 32, 64);
    // F.normalize: x / max(||x||, 1e-12)
    const float inv_a = 1.f / fmaxf(sqrtf(na), 1e-12f);  // for A row r
    const float inv_b = 1.f / fmaxf(sqrtf(nb), 1e-12f);  // for B row r = this lane's output column
    acc.x *= __shfl(inv_a, 4 * g + 0, 64) * inv_b;
    acc.y *= __shfl(inv_a, 4 * g + 1, 64) * inv_b;
    acc.z *= __shfl(inv_a, 4 * g + 2, 64) * inv_b;
    acc.w *= __shfl(inv_a, 4 * g + 3, 64) * inv_b;
  }
  const int oc = tile_k * 16 + r;
  if (oc < k) {
    const int orow = tile_n * 16 + 4 * g;
    if (orow + 0 < n) out[int64_t(orow + 0) * ldo + oc] = acc.x;
    if (orow + 1 < n) out[int64_t(orow + 1) * ldo + oc] = acc.y;
    if (orow + 2 < n) out[int64_t(orow + 2) * ldo + oc] = acc.z;
    if (orow + 3 < n) out[int64_t(orow + 3) * ldo + oc] = acc.w;
  }
}

// out = (softmax over columns of each row + softmax over rows of each column) / 2
// (tracker.py:232-235).  One workgroup: the matrices here are at most a few hundred square.
constexpr int kBsMax = 4096;

// STAGED: the matrix is first copied into LDS with one coalesced pass and every later read comes from there.  Round 2's
// form read S from global memory in each of its four passes -- per row two dependent round trips (max, then sum), 13 rows
// per wave at 50 x 50 on 4 waves: 27.6 us for a 50 x 50 matrix, five times the 300 x 300 similarity before it.
constexpr int kBsStageMax = 12288;      // elements (48 KiB) staged at most; larger matrices keep the global-memory form
// The staged form asks for (2 (n + k) + n k) x 4 bytes of dynamic LDS, and without an opt-in a launch may ask for 64 KiB at
// most: a few rows against the tracker's full memory bank (n = 6, k = 2 048: 65 584 B) passed the element test above and
// failed to launch.  The choice is made on the BYTES of the whole request.
constexpr size_t kBsStageLdsMax = 64 * 1024;

template <bool STAGED>
__global__ void __launch_bounds__(1024)
bisoftmax_kernel(const float* __restrict__ S_global, float* __restrict__ out, int n, int k, int lds_,
                 int ldo) {
  extern __shared__ float sm[];
  float* rmax = sm;            // [n]
  float* rsum = rmax + n;      // [n]
  float* cmax = rsum + n;      // [k]
  float* csum = cmax + k;      // [k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves = blockDim.x >> 6;
  const float* S = S_global;
  if constexpr (STAGED) {
    float* tile = csum + k;    // [n][k], dense
    for (int e = tid; e < n * k; e += blockDim.x) {
      const int i = e / k, j = e - i * k;
      tile[e] = S_global[int64_t(i) * lds_ + j];
    }
    __syncthreads();
    S = tile;
    lds_ = k;
  }
  // rows: one wave per row, lanes across columns
  for (int i = wave; i < n; i += waves) {
    float m = -INFINITY;
    for (int j = lane; j < k; j += 64) m = fmaxf(m, S[int64_t(i) * lds_ + j]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float s = 0.f;
    for (int j = lane; j < k; j += 64) s += expf(S[int64_t(i) * lds_ + j] - m);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) { rmax[i] = m; rsum[i] = s; }
  }
  // columns: one thread per column, walking the rows (coalesced across threads)
  for (int j = tid; j < k; j += blockDim.x) {
    float m = -INFINITY;
    for (int i = 0; i < n; ++i) m = fmaxf(m, S[int64_t(i) * lds_ + j]);
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += expf(S[int64_t(i) * lds_ + j] - m);
    cmax[j] = m; csum[j] = s;
  }
  __syncthreads();
  for (int64_t e = tid; e < int64_t(n) * k; e += blockDim.x) {
    const int i = int(e / k), j = int(e - int64_t(i) * k);
    const float s = S[int64_t(i) * lds_ + j];
    out[int64_t(i) * ldo + j] = 0.5f * (expf(s - rmax[i]) / rsum[i] + expf(s - cmax[j]) / csum[j]);
  }
}

}  // namespace vnx

using namespace vnx;

extern "C" int vnx_reid_similarity(int dtype, const void* a, const void* b, void* out, int n, int k,
                                   int channels, int lda, int ldb, int ldo, int normalize,
                                   void* hip_stream) {
  if (dtype != VNX_F32) {
    set_error("vnx_reid_similarity: only f32 is built (got dtype %d)", dtype);
    return VNX_ERR_UNSUPPORTED;
  }
  if (n < 0 || k < 0 || channels <= 0 || lda < channels || ldb < channels || ldo < k) {
    set_error("vnx_reid_similarity: bad sizes n=%d k=%d C=%d lda=%d ldb=%d ldo=%d", n, k, channels,
              lda, ldb, ldo);
    return VNX_ERR_INVALID_ARGUMENT;
  }
  if (n == 0 || k == 0) return VNX_OK;
  if (!a || !b || !out) {
    set_error("vnx_reid_similarity: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  if ((lda % 4) || (ldb % 4) || (uintptr_t(a) % 16) || (uintptr_t(b) % 16)) {
    set_error("vnx_reid_similarity: rows must be 16-byte aligned (lda, ldb multiples of 4 floats)");
    return VNX_ERR_UNSUPPORTED;
  }
  const int tiles_n = (n + 15) / 16, tiles_k = (k + 15) / 16;
  const int64_t waves = int64_t(tiles_n) * tiles_k;
  hipLaunchKernelGGL(reid_similarity_kernel, dim3(uint32_t((waves + 3) / 4)), dim3(256), 0,
                     (hipStream_t)hip_stream, (const float*)a, (const float*)b, (float*)out, n, k,
                     channels, lda, ldb, ldo, normalize, tiles_k);
  return check_launch("reid_similarity");
}

extern "C" int vnx_reid_bisoftmax(int dtype, const void* sim, void* out, int n, int k, int lds, int ldo,
                                  void* hip_stream) {
  if (dtype != VNX_F32) {
    set_error("vnx_reid_bisoftmax: only f32 is built (got dtype %d)", dtype);
    return VNX_ERR_UNSUPPORTED;
  }
  if (n < 0 || k < 0 || lds < k || ldo < k) {
    set_error("vnx_reid_bisoftmax: bad sizes n=%d k=%d lds=%d ldo=%d", n, k, lds, ldo);
    return VNX_ERR_INVALID_ARGUMENT;
  }
  if (n == 0 || k == 0) return VNX_OK;
  if (n > kBsMax || k > kBsMax) {
    set_error("vnx_reid_bisoftmax: built for association matrices up to %d x %d (got %d x %d)", kBsMax,
              kBsMax, n, k);
    return VNX_ERR_UNSUPPORTED;
  }
  if (!sim || !out) {
    set_error("vnx_reid_bisoftmax: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const int threads = (int64_t(n) * k <= 1024) ? 256 : 1024;
  const size_t staged_bytes = (size_t(2) * (size_t(n) + size_t(k)) + size_t(n) * size_t(k)) * 4;
  if (int64_t(n) * k <= kBsStageMax && staged_bytes <= kBsStageLdsMax)
    hipLaunchKernelGGL(bisoftmax_kernel<true>, dim3(1), dim3(threads), staged_bytes,
                       (hipStream_t)hip_stream, (const float*)sim, (float*)out, n, k, lds, ldo);
  else
    hipLaunchKernelGGL(bisoftmax_kernel<false>, dim3(1), dim3(threads), size_t(2 * (n + k)) * 4,
                       (hipStream_t)hip_stream, (const float*)sim, (float*)out, n, k, lds, ldo);
  return check_launch("reid_bisoftmax");
}
