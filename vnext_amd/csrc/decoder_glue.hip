// decoder_glue.hip -- two small element-wise chains of a decoder layer, each as one launch forward and one backward
// (the glue of the 6 + 6 layer stack).
//
// 1. Iterative box refinement (projects/SeqFormer/seqformer/models/deformable_transformer.py:366-380, IDOL :350-365):
//      new_reference = sigmoid(delta + inverse_sigmoid(reference))         (reference with 4 components)
//      new_reference = sigmoid(cat(delta[:2] + inverse_sigmoid(reference), delta[2:]))   (2 components: the first layer)
//    with inverse_sigmoid(x) = log(clamp(x, 0, 1).clamp(min=eps) / (1 - clamp(x, 0, 1)).clamp(min=eps)), eps = 1e-5
//    (util/misc.py:493-497): ATen runs 3 clamps, a subtraction, a division, a log, a slice + add + cat and a sigmoid per layer.
// 2. SeqFormer's temporal weighting of the frame-level context of an instance query (:305-312):
//      w = softmax(time_attention_weights(tgt_box), dim=frames);  tgt2 = (tgt2 * w).sum(frames)
//    ATen: softmax, a broadcast multiply and a reduction forward; two multiplies, two reductions and the softmax backward.
#include "vnx_common.h"

namespace vnx {

typedef float dg_f4 __attribute__((ext_vector_type(4)));

// ---- 1. boxes ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dg_logit(float x, float eps) {
  const float u = fminf(fmaxf(x, 0.f), 1.f);
  return logf(fmaxf(u, eps) / fmaxf(1.f - u, eps));
}
__device__ __forceinline__ float dg_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }
// d inverse_sigmoid / d x as autograd differentiates the clamps (a clamp passes the gradient where min <= x <= max)
__device__ __forceinline__ float dg_logit_grad(float x, float eps) {
  if (!(x >= 0.f && x <= 1.f)) return 0.f;
  const float a = x >= eps ? 1.f / x : 0.f;
  const float b = 1.f - x >= eps ? 1.f / (1.f - x) : 0.f;
  return a + b;
}

template <int R>      // components of a reference row: 2 or 4
__global__ void __launch_bounds__(256)
refine_boxes_fwd_kernel(const float* __restrict__ delta, const float* __restrict__ ref, float* __restrict__ out, int64_t rows,
                        float eps) {
  const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (r >= rows) return;
  dg_f4 d = *reinterpret_cast<const dg_f4*>(delta + 4 * r);
  if (R == 4) {
    const dg_f4 x = *reinterpret_cast<const dg_f4*>(ref + 4 * r);
    d.x += dg_logit(x.x, eps); d.y += dg_logit(x.y, eps); d.z += dg_logit(x.z, eps); d.w += dg_logit(x.w, eps);
  } else {
    d.x += dg_logit(ref[2 * r], eps); d.y += dg_logit(ref[2 * r + 1], eps);
  }
  *reinterpret_cast<dg_f4*>(out + 4 * r) = dg_f4{dg_sigmoid(d.x), dg_sigmoid(d.y), dg_sigmoid(d.z), dg_sigmoid(d.w)};
}

template <int R>
__global__ void __launch_bounds__(256)
refine_boxes_bwd_kernel(const float* __restrict__ grad_out, const float* __restrict__ out, const float* __restrict__ ref,
                        float* __restrict__ grad_delta, float* __restrict__ grad_ref, int64_t rows, float eps) {
  const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (r >= rows) return;
  const dg_f4 g = *reinterpret_cast<const dg_f4*>(grad_out + 4 * r), y = *reinterpret_cast<const dg_f4*>(out + 4 * r);
  const dg_f4 gd = g * y * (1.f - y);
  *reinterpret_cast<dg_f4*>(grad_delta + 4 * r) = gd;
  if (grad_ref == nullptr) return;
  if (R == 4) {
    const dg_f4 x = *reinterpret_cast<const dg_f4*>(ref + 4 * r);
    *reinterpret_cast<dg_f4*>(grad_ref + 4 * r) = dg_f4{gd.x * dg_logit_grad(x.x, eps), gd.y * dg_logit_grad(x.y, eps),
                                                        gd.z * dg_logit_grad(x.z, eps), gd.w * dg_logit_grad(x.w, eps)};
  } else {
    grad_ref[2 * r] = gd.x * dg_logit_grad(ref[2 * r], eps);
    grad_ref[2 * r + 1] = gd.y * dg_logit_grad(ref[2 * r + 1], eps);
  }
}

// ---- 2. temporal weighting: one wave per (clip, query), lane = channels 4 lane .. 4 lane + 3 of each 256-channel slab --------
constexpr int kTwMaxFrames = 16;

__device__ __forceinline__ float dg_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// (the frame loops run to the compile-time maximum under a `t < T` guard: fully unrolled, so the per-frame values stay in
//  registers instead of an indexed scratch array)
#define VNX_TW_FRAMES(t) _Pragma("unroll") for (int t = 0; t < kTwMaxFrames; ++t) if (t < T)

// x [N, T, Q, C], z [N, T, Q] -> out [N, Q, C] = sum_t softmax_t(z) x, w [N, T, Q] = the weights (saved for the backward)
__global__ void __launch_bounds__(256)
time_weighted_sum_fwd_kernel(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ out,
                             float* __restrict__ w, int N, int T, int Q, int C) {
  const int64_t nq = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (nq >= int64_t(N) * Q) return;
  const int lane = threadIdx.x & 63;
  const int n = int(nq / Q), q = int(nq - int64_t(n) * Q);
  float wt[kTwMaxFrames];
  float m = -INFINITY, s = 0.f;
  VNX_TW_FRAMES(t) { wt[t] = z[(int64_t(n) * T + t) * Q + q]; m = fmaxf(m, wt[t]); }
  VNX_TW_FRAMES(t) { wt[t] = expf(wt[t] - m); s += wt[t]; }
  const float inv = 1.f / s;
  VNX_TW_FRAMES(t) {
    wt[t] *= inv;
    if (lane == 0) w[(int64_t(n) * T + t) * Q + q] = wt[t];
  }
  for (int c = lane * 4; c < C; c += 256) {
    dg_f4 acc = {0.f, 0.f, 0.f, 0.f};
    VNX_TW_FRAMES(t) acc += wt[t] * *reinterpret_cast<const dg_f4*>(x + ((int64_t(n) * T + t) * Q + q) * C + c);
    *reinterpret_cast<dg_f4*>(out + nq * C + c) = acc;
  }
}

// grad_x[n, t, q, :] = w_t g;  grad_z = softmax backward of d w_t = <g, x_t>
__global__ void __launch_bounds__(256)
time_weighted_sum_bwd_kernel(const float* __restrict__ grad_out, const float* __restrict__ x, const float* __restrict__ w,
                             float* __restrict__ grad_x, float* __restrict__ grad_z, int N, int T, int Q, int C) {
  const int64_t nq = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (nq >= int64_t(N) * Q) return;
  const int lane = threadIdx.x & 63;
  const int n = int(nq / Q), q = int(nq - int64_t(n) * Q);
  float wt[kTwMaxFrames], dw[kTwMaxFrames];
  VNX_TW_FRAMES(t) { wt[t] = w[(int64_t(n) * T + t) * Q + q]; dw[t] = 0.f; }
  for (int c = lane * 4; c < C; c += 256) {
    const dg_f4 g = *reinterpret_cast<const dg_f4*>(grad_out + nq * C + c);
    VNX_TW_FRAMES(t) {
      const int64_t at = ((int64_t(n) * T + t) * Q + q) * C + c;
      const dg_f4 xv = *reinterpret_cast<const dg_f4*>(x + at);
      dw[t] += g.x * xv.x + g.y * xv.y + g.z * xv.z + g.w * xv.w;
      *reinterpret_cast<dg_f4*>(grad_x + at) = wt[t] * g;
    }
  }
  float dot = 0.f;
  VNX_TW_FRAMES(t) { dw[t] = dg_wave_sum(dw[t]); dot += wt[t] * dw[t]; }
  if (lane == 0) {
    VNX_TW_FRAMES(t) grad_z[(int64_t(n) * T + t) * Q + q] = wt[t] * (dw[t] - dot);
  }
}
#undef VNX_TW_FRAMES

}  // namespace vnx

using namespace vnx;

extern "C" int vnx_refine_boxes_forward(int dtype, const void* delta, const void* reference, void* out, long long rows,
                                        int ref_components, float eps, void* hip_stream) {
  if (dtype != VNX_F32 || (ref_components != 2 && ref_components != 4) || rows < 0) {
    set_error("vnx_refine_boxes_forward: fp32 rows of 4 against references of 2 or 4 components (dtype %d, %d components)", dtype,
              ref_components);
    return VNX_ERR_INVALID_ARGUMENT;
  }
  if (rows == 0) return VNX_OK;
  if (!delta || !reference || !out) {
    set_error("vnx_refine_boxes_forward: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const dim3 grid(uint32_t((rows + 255) / 256));
  if (ref_components == 4)
    hipLaunchKernelGGL(refine_boxes_fwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)hip_stream, (const float*)delta,
                       (const float*)reference, (float*)out, int64_t(rows), eps);
  else
    hipLaunchKernelGGL(refine_boxes_fwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)hip_stream, (const float*)delta,
                       (const float*)reference, (float*)out, int64_t(rows), eps);
  return check_launch("refine_boxes_fwd");
}

extern "C" int vnx_refine_boxes_backward(int dtype, const void* grad_out, const void* out, const void* reference, void* grad_delta,
                                         void* grad_reference, long long rows, int ref_components, float eps, void* hip_stream) {
  if (dtype != VNX_F32 || (ref_components != 2 && ref_components != 4) || rows < 0) {
    set_error("vnx_refine_boxes_backward: fp32 rows of 4 against references of 2 or 4 components (dtype %d, %d components)", dtype,
              ref_components);
    return VNX_ERR_INVALID_ARGUMENT;
  }
  if (rows == 0) return VNX_OK;
  if (!grad_out || !out || !reference || !grad_delta) {
    set_error("vnx_refine_boxes_backward: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const dim3 grid(uint32_t((rows + 255) / 256));
  if (ref_components == 4)
    hipLaunchKernelGGL(refine_boxes_bwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)hip_stream, (const float*)grad_out,
                       (const float*)out, (const float*)reference, (float*)grad_delta, (float*)grad_reference, int64_t(rows), eps);
  else
    hipLaunchKernelGGL(refine_boxes_bwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)hip_stream, (const float*)grad_out,
                       (const float*)out, (const float*)reference, (float*)grad_delta, (float*)grad_reference, int64_t(rows), eps);
  return check_launch("refine_boxes_bwd");
}

static int tw_check(const char* fn, int dtype, int clips, int frames, int queries, int channels) {
  if (dtype != VNX_F32 || clips < 0 || queries < 0 || frames < 1 || frames > kTwMaxFrames || channels < 4 || (channels & 3) != 0) {
    set_error("%s: fp32, 1..%d frames, channels a multiple of 4 (dtype %d, %d clips, %d frames, %d queries, %d channels)", fn,
              kTwMaxFrames, dtype, clips, frames, queries, channels);
    return VNX_ERR_INVALID_ARGUMENT;
  }
  return VNX_OK;
}

extern "C" int vnx_time_weighted_sum_forward(int dtype, const void* x, const void* logits, void* out, void* weights, int clips,
                                             int frames, int queries, int channels, void* hip_stream) {
  if (int st = tw_check("vnx_time_weighted_sum_forward", dtype, clips, frames, queries, channels)) return st;
  const int64_t rows = int64_t(clips) * queries;
  if (rows == 0) return VNX_OK;
  if (!x || !logits || !out || !weights) {
    set_error("vnx_time_weighted_sum_forward: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  hipLaunchKernelGGL(time_weighted_sum_fwd_kernel, dim3(uint32_t((rows + 3) / 4)), dim3(256), 0, (hipStream_t)hip_stream,
                     (const float*)x, (const float*)logits, (float*)out, (float*)weights, clips, frames, queries, channels);
  return check_launch("time_weighted_sum_fwd");
}

extern "C" int vnx_time_weighted_sum_backward(int dtype, const void* grad_out, const void* x, const void* weights, void* grad_x,
                                              void* grad_logits, int clips, int frames, int queries, int channels,
                                              void* hip_stream) {
  if (int st = tw_check("vnx_time_weighted_sum_backward", dtype, clips, frames, queries, channels)) return st;
  const int64_t rows = int64_t(clips) * queries;
  if (rows == 0) return VNX_OK;
  if (!grad_out || !x || !weights || !grad_x || !grad_logits) {
    set_error("vnx_time_weighted_sum_backward: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  hipLaunchKernelGGL(time_weighted_sum_bwd_kernel, dim3(uint32_t((rows + 3) / 4)), dim3(256), 0, (hipStream_t)hip_stream,
                     (const float*)grad_out, (const float*)x, (const float*)weights, (float*)grad_x, (float*)grad_logits, clips,
                     frames, queries, channels);
  return check_launch("time_weighted_sum_bwd");
}
