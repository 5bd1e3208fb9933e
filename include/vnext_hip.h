/*
 * include/vnext_hip.h -- C ABI of libvnext_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for VNext's data-parallel hot path.  Every entry
 * point takes plain device pointers, sizes and a HIP stream, returns an int
 * status, never allocates, never synchronises and never keeps a pointer after
 * it returns, so a call is re-entrant and hipGraph-capturable.  No torch type
 * appears here; the Python module `MultiScaleDeformableAttention` (repo root)
 * and vnext_amd/ bind these symbols through ctypes (INTEGRATION.md shows the
 * binding a VNext maintainer would add).
 *
 * Reference interfaces replaced (paths relative to the VNext tree):
 *   vnx_msda_forward   <- ms_deform_attn_forward
 *        projects/SeqFormer/seqformer/models/ops/src/ms_deform_attn.h:20-39,
 *        src/cuda/ms_deform_attn_cuda.cu:20-80, src/vision.cpp:14
 *   vnx_msda_backward  <- ms_deform_attn_backward
 *        .../src/ms_deform_attn.h:41-61, src/cuda/ms_deform_attn_cuda.cu:83-153,
 *        src/vision.cpp:15
 *
 * Tensor layouts are the reference's (all contiguous, ms_deform_attn_cuda.cu:28-38):
 *   value            [batch, spatial_size, num_heads, channels]
 *   spatial_shapes   [num_levels, 2]  int64, (H_l, W_l), DEVICE memory
 *   level_start_index[num_levels]     int64, DEVICE memory
 *   sampling_loc     [batch, num_query, num_heads, num_levels, num_point, 2]  (x, y) in [0,1]
 *   attn_weight      [batch, num_query, num_heads, num_levels, num_point]
 *   output / grad_output [batch, num_query, num_heads*channels]
 */
#ifndef VNEXT_HIP_H_
#define VNEXT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VNX_ABI_VERSION 15

/* element types */
enum {
  VNX_F32 = 0,
  VNX_F64 = 1,
  VNX_BF16 = 2,
  VNX_F16 = 3
};

/* status codes */
enum {
  VNX_OK = 0,
  VNX_ERR_INVALID_ARGUMENT = 1, /* null pointer, non-positive size, bad dtype combination */
  VNX_ERR_UNSUPPORTED = 2,      /* shape outside what the kernels address (see vnx_last_error) */
  VNX_ERR_WORKSPACE = 3,        /* workspace missing or too small */
  VNX_ERR_LAUNCH = 4            /* hipGetLastError() != hipSuccess after the launch */
};

/* Library identification.  vnx_abi_version() == VNX_ABI_VERSION. */
int vnx_abi_version(void);
/* Static string for a status code. */
const char* vnx_status_string(int status);
/* Thread-local detail of the last non-OK status returned on this thread. */
const char* vnx_last_error(void);

/*
 * Multi-scale deformable attention, forward.
 *   output[b,q,m,:] = sum_{l,k} attn[b,q,m,l,k] * bilinear(value_l[b,:,m,:], loc[b,q,m,l,k])
 * zero padding outside the maps, pixel convention x*W-0.5 / y*H-0.5
 * (ms_deform_im2col_cuda.cuh:285-288).  `output` is fully written (it does not
 * need the reference's zero pre-fill, ms_deform_attn_cuda.cu:54).
 *
 * value_dtype: type of value and output (F32, F64, BF16, F16).
 * loc_dtype  : type of sampling_loc and attn_weight; equal to value_dtype, or
 *              VNX_F32 with a 16-bit value (the autocast case).
 * The reference's im2col_step chunking (ms_deform_attn_cuda.cu:50-75) is a host
 * concern and is not part of this ABI: one call covers the whole batch.
 */
int vnx_msda_forward(int value_dtype, int loc_dtype,
                     const void* value, const int64_t* spatial_shapes,
                     const int64_t* level_start_index, const void* sampling_loc,
                     const void* attn_weight, void* output,
                     int batch, int spatial_size, int num_heads, int channels,
                     int num_levels, int num_query, int num_point,
                     void* hip_stream);

/* flags of vnx_msda_backward */
enum {
  /*
   * The caller guarantees the levels are packed back to back in order:
   * level_start_index[l] == sum_{j<l} H_j*W_j and the sum over all levels ==
   * spatial_size -- the layout the reference always builds
   * (projects/SeqFormer/seqformer/models/deformable_transformer.py:97-106).
   * Without the flag the library works out the same predicate ON THE DEVICE and
   * issues both the fast kernels and the general ones, each set exiting at once when
   * the predicate is not theirs (no host synchronisation, ~2 empty launches).
   */
  VNX_MSDA_LEVELS_PACKED = 1,
  /*
   * Fork the call: a backward call below 1 024 queries (32-channel heads) is two halves neither of which reads what the
   * other writes (grad_value; grad_sampling_loc + grad_attn_weight).  By default they are ONE launch whose workgroups take
   * either role (fp32 locations, 4 levels x 4 points: the decoders' calls) or two launches one after the other on `hip_stream`.  With this flag the grad_value kernel is launched on a side stream the library keeps per host thread
   * and device (created on first use, never synchronised with the host), between an event recorded on `hip_stream` and an
   * event `hip_stream` then waits for: the two kernels share the GPU, `hip_stream` sees the call as one operation, a stream
   * capture records a fork and a join.  Measured on MI355X / ROCm 7.2: the pair then spans 24.3
   * instead of 29.1 us, but the two cross-queue dependencies cost a captured graph 9 us per call and an eager caller
   * 19 us -- a loss today, hence opt-in.
   */
  VNX_MSDA_FORK = 2
};

/*
 * Bytes of scratch vnx_msda_backward needs for these sizes and flags.  32-channel heads: below 1 024
 * queries NONE (since ABI 12: the grad_value kernel decodes the op's own inputs; until ABI 11 the grad_loc
 * kernel left it 20 B per sample of records + unit tags); from 1 024 queries up (4 levels x 4 points) what
 * the grad_loc kernel hands the grad_value kernel -- 8 B per (batch, head, level, query tile), a tile
 * being the queries one wave of the grad_loc kernel handles (4 on calls of up to 262 144 query rows, 8
 * beyond), plus the fp32 partial rows in which the query pieces of the coarse levels meet (at most
 * 16 x min(S, 4 096) rows of 128 B per (batch, head) on calls of fewer than 32 (batch, head) pairs, 4 x
 * from 32 pairs up) -- size the scratch with this function, not by hand.  16-bit values additionally need an fp32 [B, S, M, 32] image of grad_value when the levels are
 * not promised packed (the general path accumulates with fp32 atomics), or on the record-fed path with
 * >= 1 024 queries (other level / point counts than 4 x 4).  Other head widths: that image for 16-bit
 * values, else 0.
 */
size_t vnx_msda_backward_workspace_bytes(int value_dtype, int loc_dtype, int batch,
                                         int spatial_size, int num_heads, int channels,
                                         int num_levels, int num_query, int num_point, int flags);

/*
 * Multi-scale deformable attention, backward (ms_deform_im2col_cuda.cuh:87-159).
 * All three gradient buffers are fully written; they need no zero pre-fill by
 * the caller (the reference's three at::zeros_like, ms_deform_attn_cuda.cu:121-123,
 * disappear, or happen on `hip_stream` inside the call where the general path
 * still needs one).
 *   grad_value        like value        (value_dtype)
 *   grad_sampling_loc like sampling_loc (loc_dtype)
 *   grad_attn_weight  like attn_weight  (loc_dtype)
 * With 32-channel heads and packed levels grad_value is produced without global
 * atomics (one owner per row, accumulation in LDS); otherwise it is accumulated
 * with hardware floating-point atomics as in the reference.  Either way its
 * low-order bits depend on scheduling.
 */
int vnx_msda_backward(int value_dtype, int loc_dtype,
                      const void* value, const int64_t* spatial_shapes,
                      const int64_t* level_start_index, const void* sampling_loc,
                      const void* attn_weight, const void* grad_output,
                      void* grad_value, void* grad_sampling_loc, void* grad_attn_weight,
                      int batch, int spatial_size, int num_heads, int channels,
                      int num_levels, int num_query, int num_point, int flags,
                      void* workspace, size_t workspace_bytes,
                      void* hip_stream);

/*
 * MSDeformAttn with the module's prologue fused in.  Instead of
 * sampling_locations / attention_weights the kernels take what the module computes them from:
 *   sampling_offsets  [batch, num_query, num_heads, num_levels, num_point, 2]  output of the
 *                     `sampling_offsets` Linear
 *   attention_logits  [batch, num_query, num_heads, num_levels * num_point]    output of the
 *                     `attention_weights` Linear, BEFORE the softmax
 *   reference_points  [batch / reference_batch_div, num_query, num_levels, ref_dim]
 * and evaluate   attention = softmax(logits)   and
 *   ref_dim 2:  location = reference + offsets / (W_l, H_l)
 *   ref_dim 4:  location = reference_xy + offsets / num_point * reference_wh * 0.5
 * (projects/IDOL/idol/models/ops/modules/ms_deform_attn.py:99-108; SeqFormer :99-112, :159-161)
 * inside the sampling kernels, so the two intermediate tensors never exist in memory.
 * reference_batch_div > 1: that many consecutive batch elements share one reference row (the
 * frames of a clip in SeqFormer's encoder).  Requirements: channels == 32,
 * num_levels * num_point == 16, value f32 or bf16 (offsets / logits / references all
 * `query_dtype`: f32, or bf16 with a bf16 value; ABI 14: `ref_dim | VNX_MSDA_REF_F32` says the
 * reference points are fp32 although offsets / logits are bf16 -- under torch.autocast the two
 * Linears emit bf16 while the reference points stay fp32, and rounding positions to 8 mantissa
 * bits would cost more than half a pixel at 720p), PACKED levels (level_start_index = running
 * sum of H*W; the backward's grad_value kernel does nothing on the device otherwise).  Anything
 * else: VNX_ERR_UNSUPPORTED -- use vnx_msda_forward / vnx_msda_backward.
 * Backward: grad_sampling_offsets / grad_attention_logits like their inputs; grad_value like
 * value; grad_reference_points (optional, fp32 [batch, num_query, num_levels, 2], ref_dim 2
 * and reference_batch_div 1 only) is zero-filled inside and accumulated over heads with fp32
 * atomics.  workspace: vnx_msda_fused_backward_workspace_bytes(...) bytes of device memory.
 */
#define VNX_MSDA_REF_F32 0x100      /* or-ed into ref_dim of vnx_msda_fused_forward / _backward (see above) */
int vnx_msda_fused_forward(int value_dtype, int query_dtype, const void* value, const int64_t* spatial_shapes,
                           const int64_t* level_start_index, const void* sampling_offsets,
                           const void* attention_logits, const void* reference_points, void* output,
                           int batch, int spatial_size, int num_heads, int channels, int num_levels,
                           int num_query, int num_point, int ref_dim, int reference_batch_div, void* hip_stream);
size_t vnx_msda_fused_backward_workspace_bytes(int value_dtype, int batch, int spatial_size, int num_heads,
                                               int num_levels, int num_query, int num_point);
int vnx_msda_fused_backward(int value_dtype, int query_dtype, const void* value, const int64_t* spatial_shapes,
                            const int64_t* level_start_index, const void* sampling_offsets,
                            const void* attention_logits, const void* reference_points, const void* grad_output,
                            void* grad_value, void* grad_sampling_offsets, void* grad_attention_logits,
                            float* grad_reference_points, int batch, int spatial_size, int num_heads,
                            int channels, int num_levels, int num_query, int num_point, int ref_dim,
                            int reference_batch_div, void* workspace, size_t workspace_bytes, void* hip_stream);

/*
 * CondInst-style dynamic mask head, forward (per-instance 1x1 conv stack 10->8->8->1 on
 * [relative coordinates | 8 mask features], ReLU between, then the x2 "aligned bilinear"
 * up-sampling), fused into one kernel.  Replaces the op chain
 *   CondInst_segm.dynamic_mask_with_coords + mask_heads_forward + parse_dynamic_params +
 *   compute_locations + aligned_bilinear
 *   (projects/SeqFormer/seqformer/models/segmentation_condInst.py:425-493, 404-422, 614-637,
 *    665-678, 640-662; same code in projects/IDOL/idol/models/segmentation_condInst.py:398-468).
 *   mask_feats       [num_images, channels=8, height, width]          (stride-8 mask features)
 *   reference_points [num_insts, 2]   (x, y) in image pixels
 *   params           [num_insts, num_params=169]  split [w0(80) w1(64) w2(8) b0(8) b1(8) b2(1)]
 *   inst_image       [num_insts] int32: the image each instance belongs to (the reference's
 *                    `num_insts` list, expanded; DEVICE memory)
 *   out              [num_insts, 2*height, 2*width], fully written
 * `stride` is mask_feat_stride (8): pixel centres are x*stride + stride/2.
 * Limits (VNX_ERR_UNSUPPORTED beyond them): width <= 3 711 columns of the stride-8 feature map -- images up to 29 688 pixels
 * wide; the kernel keeps one row of logits + 384 pixels per wave in LDS -- and height * width < 2^26.
 */
int vnx_dynamic_mask_head_forward(int dtype, const void* mask_feats, const void* reference_points,
                                  const void* params, const int32_t* inst_image, void* out,
                                  int num_images, int channels, int height, int width,
                                  int num_insts, int num_params, int stride, void* hip_stream);

/*
 * Dynamic mask head, backward (the training path, forward_mask_head_train,
 * segmentation_condInst.py:354-401): gradients of the chain above with respect to the mask
 * features, the reference points and the per-instance parameters, contracted with grad_out.
 * Replaces what autograd runs for the reference: the transposes of pad/interpolate/pad, three
 * grouped-conv backward-data + three backward-weight launches (groups = num_insts), the ReLU
 * masks and the reduction over the `repeat`ed feature map (:452-456).  Nothing is saved by
 * the forward; the hidden layers are recomputed.
 *   grad_out    [num_insts, 2*height, 2*width]
 *   grad_feats  [num_images, 8, height, width]   sum over the instances of each image
 *   grad_ref    [num_insts, 2]
 *   grad_params [num_insts, 169]
 * All three outputs are fully defined on return (zero-filled inside, then accumulated with
 * fp32 atomics: the order of the sums over pixels / instances is not fixed, as in the
 * reference's cuDNN/MIOpen weight-gradient kernels).
 */
int vnx_dynamic_mask_head_backward(int dtype, const void* mask_feats, const void* reference_points,
                                   const void* params, const int32_t* inst_image, const void* grad_out,
                                   void* grad_feats, void* grad_ref, void* grad_params,
                                   int num_images, int channels, int height, int width,
                                   int num_insts, int num_params, int stride, void* hip_stream);

/*
 * The training pair (ABI 15): the forward of a call whose backward will follow, and that backward.
 * `vnx_dynamic_mask_head_forward_train` is `vnx_dynamic_mask_head_forward` whose launch ALSO zero-fills the three gradient
 * buffers the backward accumulates into (each thread of the forward's grid a slice, before anything else: the buffers are
 * outputs of the later backward, nothing in the forward reads them); `vnx_dynamic_mask_head_backward_zeroed` is
 * `vnx_dynamic_mask_head_backward` WITHOUT its own zero-fill launch -- the caller guarantees that grad_feats / grad_ref /
 * grad_params hold zeros (from the forward above, or any other fill) and that nothing wrote to them since.  Same arguments,
 * same limits, same results; one launch fewer per training step (4.8 us of the 35-us training forward + backward at the
 * bench's shape).  What `forward_mask_head_train` (segmentation_condInst.py:354-401) needs from autograd is unchanged: the
 * Python side allocates the three buffers at forward time and hands them to the backward (vnext_amd/heads/dynamic_mask.py).
 */
int vnx_dynamic_mask_head_forward_train(int dtype, const void* mask_feats, const void* reference_points,
                                        const void* params, const int32_t* inst_image, void* out,
                                        void* grad_feats, void* grad_ref, void* grad_params,
                                        int num_images, int channels, int height, int width,
                                        int num_insts, int num_params, int stride, void* hip_stream);
int vnx_dynamic_mask_head_backward_zeroed(int dtype, const void* mask_feats, const void* reference_points,
                                          const void* params, const int32_t* inst_image, const void* grad_out,
                                          void* grad_feats, void* grad_ref, void* grad_params,
                                          int num_images, int channels, int height, int width,
                                          int num_insts, int num_params, int stride, void* hip_stream);

/*
 * IDOL re-identification head: similarity matrix  out[i, j] = <a_i, b_j>  for a [n, channels]
 * (row stride lda) and b [k, channels] (row stride ldb), on the matrix cores in exact fp32.
 * normalize != 0 fuses F.normalize(., p=2, dim=1, eps=1e-12) of both operands (cosine).
 * Replaces torch.mm(embeds, memo_embeds.t()) / the cosine variant
 * (projects/IDOL/idol/models/tracker.py:229-244) and the per-instance
 * einsum('nc,kc->nk') calls of projects/IDOL/idol/models/pos_neg_select.py:47,58-62.
 * Rows must be 16-byte aligned (lda, ldb multiples of 4).  out [n, k], row stride ldo.
 */
int vnx_reid_similarity(int dtype, const void* a, const void* b, void* out, int n, int k,
                        int channels, int lda, int ldb, int ldo, int normalize, void* hip_stream);

/*
 * Bi-directional softmax association score (tracker.py:232-235):
 *   out = (softmax(sim, dim=1) + softmax(sim, dim=0)) / 2      sim, out [n, k], n, k <= 4096
 */
int vnx_reid_bisoftmax(int dtype, const void* sim, void* out, int n, int k, int lds, int ldo,
                       void* hip_stream);

/*
 * Pairwise intersections of binarised masks:  inter[i, j] = |{p : logit_i[p] > 0 and logit_j[p] > 0}|
 * (sigmoid > 0.5 is logit > 0), exact int32, symmetric, diagonal = areas.  mask_logits [num_masks,
 * mask_pixels] fp32 contiguous.  Replaces the pairwise mask_iou launches of
 * projects/IDOL/idol/models/tracker.py:17-46 (the IoU is (inter + 1e-6) / (area_i + area_j - inter + 1e-6)).
 * Masks become bit words (one ballot per 64 pixels), intersections are popcounts.
 */
size_t vnx_mask_intersections_workspace_bytes(int num_masks, int mask_pixels);
int vnx_mask_intersections(const float* mask_logits, int num_masks, int mask_pixels, int32_t* inter,
                           void* workspace, size_t workspace_bytes, void* hip_stream);

/*
 * IDOL's online tracker with the memory bank on the device: IDOL_Tracker.match + update_memo + memo
 * (projects/IDOL/idol/models/tracker.py:103-298), one call per frame, no host round trip.
 * The fields are the constructor arguments of the reference class (tracker.py:52-98) plus the sizes
 * of the device-resident memory.
 */
typedef struct vnx_tracker_config {
  int capacity;              /* tracklet slots alive at a time, 1..2048 */
  int channels;              /* embedding width, multiple of 4 */
  int memory_len;            /* remembered embeddings per tracklet (long_embed / long_score), 1..16 */
  int memo_tracklet_frames;  /* a tracklet unseen for this many frames is dropped */
  int match_metric;          /* 0 bisoftmax, 1 softmax, 2 cosine */
  int long_match;            /* match against the score-weighted mean of the remembered embeddings */
  int frame_weight;          /* several candidates above 0.5: prefer the longer-lived tracklet */
  int temporal_weight;       /* long_match weights += 1/L, 2/L, ..., 1 (newest) */
  float nms_thr_pre;
  float nms_thr_post;
  float init_score_thr;
  float addnew_score_thr;
  float match_score_thr;
  float memo_momentum;
} vnx_tracker_config;

/*
 * state: a caller-owned device blob of vnx_tracker_state_bytes(cfg) bytes, 16-byte aligned;
 * vnx_tracker_reset empties it (start of a video).  The library keeps no pointer to it.
 * Its first three int32 are {tracklets created so far, tracklets that found no free slot (should stay 0:
 * raise `capacity`), frames processed}.
 */
size_t vnx_tracker_state_bytes(const vnx_tracker_config* cfg);
int vnx_tracker_reset(const vnx_tracker_config* cfg, void* state, void* hip_stream);
size_t vnx_tracker_frame_workspace_bytes(const vnx_tracker_config* cfg, int num_dets, int mask_pixels);
/*
 * One frame.  Detections in descending score order, as the reference's caller passes them:
 *   mask_logits [num_dets, mask_pixels] fp32, embeds [num_dets, channels] fp32 (16-byte aligned),
 *   det_scores [num_dets] fp32 (bboxes[:, 4]), labels [num_dets] int64, num_dets <= 512.
 * ids_out [num_dets] int64, for EVERY input detection: the tracklet id (>= 0), -1 = back-drop,
 * -2 = unassigned duplicate (the reference's values), -3 = removed by the mask NMS (the reference
 * drops those rows from its return value: `valids`, tracker.py:212-219).
 * Enqueues four kernels on hip_stream; no synchronisation, no allocation.  num_dets == 0 is a no-op.
 */
int vnx_tracker_frame(const vnx_tracker_config* cfg, void* state, const float* mask_logits,
                      const float* embeds, const float* det_scores, const int64_t* labels,
                      int num_dets, int mask_pixels, int frame_id, int64_t* ids_out,
                      void* workspace, size_t workspace_bytes, void* hip_stream);

/*
 * y = LayerNorm(x + dropout(r)) over rows of 256 channels, in one pass: the chain that closes every sub-layer of the
 * deformable transformer (projects/SeqFormer/seqformer/models/deformable_transformer.py:201-236,286-385:
 * `src = src + self.dropout1(src2); src = self.norm1(src)`), three ATen launches forward and four backward per site.
 *   x, r, y, z [rows, 256] fp32 contiguous; gamma, beta [256]; stats [rows, 2] = {mean, rstd} per row.
 *   dtype (ABI 14) names the element type of the BRANCH -- r and grad_r: VNX_F32, VNX_BF16 or VNX_F16 -- and nothing else: under
 *   torch.autocast(bfloat16) the Linear or attention output that arrives here is bf16 while the residual stream x, the
 *   LayerNorm and its output are fp32 (the eager chain's types: the sum promotes, autocast runs layer_norm in fp32).
 *   Arithmetic is fp32 either way; grad_r is rounded once on its way out.
 *   z = x + dropout(r) is an OUTPUT the backward needs (r is dead afterwards).  p = drop probability (0 in eval mode),
 *   kept elements are scaled by 1 / (1 - p).  The mask is not stored: element e is kept iff hash(seed, e) >= p * 2^32,
 *   and the backward recomputes it from the same `seed` -- pass the same value to both calls, a fresh one per
 *   forward call.  seed_device (may be null): a 64-bit word in DEVICE memory that the kernels read and mix into
 *   `seed`.  A host integer is baked into a captured hipGraph, so every replay of a captured training step would
 *   drop the same elements; a device word the caller bumps once per step (inside the graph) gives each replay fresh
 *   masks.  It must hold the same value when the matching backward runs.
 *   r_bias (may be null; ABI 11): a [256] vector added to every row of r before the dropout -- the bias of the Linear
 *   that produced r, when that GEMM ran without it -- so that its gradient falls out of this op's backward for free
 *   (grad_r_bias = column sums of grad_r) instead of costing the caller a reduction launch per Linear.
 * Backward: grad_x = d loss / d x, grad_r = d loss / d r (both [rows, 256]), grad_gamma, grad_beta [256] and, when
 * grad_r_bias is not null, grad_r_bias [256] (overwritten, not accumulated; summed in a fixed order).  partial: scratch
 * of vnx_add_dropout_layernorm_partial_bytes() bytes.
 */
size_t vnx_add_dropout_layernorm_partial_bytes(void);
int vnx_add_dropout_layernorm_forward(int dtype, const void* x, const void* r, const void* r_bias, const void* gamma, const void* beta,
                                      void* y, void* z, void* stats, long long rows, int channels, float p, float eps,
                                      unsigned long long seed, const unsigned long long* seed_device,
                                      void* hip_stream);
int vnx_add_dropout_layernorm_backward(int dtype, const void* grad_y, const void* z, const void* stats,
                                       const void* gamma, void* grad_x, void* grad_r, void* grad_gamma, void* grad_beta,
                                       void* grad_r_bias, void* partial, long long rows, int channels, float p, unsigned long long seed,
                                       const unsigned long long* seed_device, void* hip_stream);

/*
 * Bias / activation epilogues of a library GEMM that ran WITHOUT its bias, IN PLACE over h [rows, channels] -- dtype:
 * VNX_F32, or (ABI 14) VNX_BF16 / VNX_F16 for the output of a 16-bit GEMM under autocast: h, grad and grad_h in that type, the bias, its
 * gradient and the partial sums fp32, arithmetic fp32 -- (channels a multiple of 4, <= 4 096), with the bias gradient produced
 * by the backward pass itself:
 *   relu != 0: y = dropout(relu(h + bias)) -- the middle of a transformer FFN
 *     (projects/SeqFormer/seqformer/models/deformable_transformer.py:226-229,330-338: relu, dropout and, in the backward,
 *     masked_scale, threshold_backward and linear1's bias-gradient reduction: five ATen launches and 11.5 passes over the
 *     hidden tensor; here two launches + a small reduction and 2 + 3 passes);
 *   relu == 0 (p must be 0): y = h + bias.
 *   row_zero (may be null): one byte per row; rows with a non-zero byte are written as zeros -- the padding mask of
 *     `value = value_proj(x).masked_fill(mask[..., None], 0)` (projects/SeqFormer/seqformer/models/ops/modules/ms_deform_attn.py:94-96),
 *     which ATen runs as a copy + a fill forward and again backward, plus the bias reduction.
 * bias may be null.  Dropout as in vnx_add_dropout_layernorm_* (hash of (seed, element), seed_device for captured graphs).
 * Backward: grad_h = grad where the row is kept (and, with the ReLU, where y > 0, times 1 / (1 - p): y > 0 <=> the
 * element passed the ReLU and was kept, so y -- the forward's output -- is all it needs; y = null when relu == 0);
 * grad_h may be the same buffer as grad.  grad_bias [channels] = the column sums of grad_h (may be null; partial: scratch
 * of vnx_bias_relu_dropout_partial_bytes(channels) bytes, needed with grad_bias; fixed summation order).
 */
size_t vnx_bias_relu_dropout_partial_bytes(int channels);
int vnx_bias_relu_dropout_forward(int dtype, void* h, const void* bias, const unsigned char* row_zero, long long rows,
                                  int channels, int relu, float p, unsigned long long seed,
                                  const unsigned long long* seed_device, void* hip_stream);
int vnx_bias_relu_dropout_backward(int dtype, const void* grad, const void* y, const unsigned char* row_zero, void* grad_h,
                                   void* grad_bias, void* partial, long long rows, int channels, float p, void* hip_stream);

/*
 * Self-attention over the object queries of a decoder layer (ABI 13) -- what `nn.MultiheadAttention(256, 8, dropout)` does
 * between its input and output projections (projects/SeqFormer/seqformer/models/deformable_transformer.py:286-323 and the
 * `_box` twin; IDOL's decoder layer the same), for ALL heads in one launch forward and one backward:
 *   out[b, i, h] = sum_j dropout(softmax_j((q_i + bq) . (k_j + bk) / sqrt(head_dim)))_ij (v_j + bv)
 * qkv: fp32 [batch * queries][row_stride], a row = q | k | v (channels = heads * head_dim each) of one query as the
 * in-projection GEMMs left them, WITHOUT bias; in_proj_bias [3 * channels] (may be null) is added here.  head_dim must be 32.
 * out [batch, queries, channels] fp32; lse [batch * heads * queries] fp32 (the log-sum-exp of a row's scaled scores: all the
 * backward needs to recompute the probabilities, which are never stored).  Dropout on the probabilities as in
 * vnx_add_dropout_layernorm_* (hash of (seed, element), seed_device for captured graphs; p = 0: none).
 * Backward: grad_qkv [batch * queries][grad_row_stride], the same row layout (gradients of the rows BEFORE the bias; the
 * bias gradient is their column sum), every element written.  No workspace, no atomics, fixed summation order.
 */
int vnx_query_self_attention_forward(int dtype, const void* qkv, const void* in_proj_bias, void* out, void* lse, int batch,
                                     int queries, int heads, int head_dim, int row_stride, float p, unsigned long long seed,
                                     const unsigned long long* seed_device, void* hip_stream);
int vnx_query_self_attention_backward(int dtype, const void* qkv, const void* in_proj_bias, const void* out, const void* lse,
                                      const void* grad_out, void* grad_qkv, int batch, int queries, int heads, int head_dim,
                                      int row_stride, int grad_row_stride, float p, unsigned long long seed,
                                      const unsigned long long* seed_device, void* hip_stream);

/*
 * Two element-wise chains of a decoder layer, one launch forward and one backward each (ABI 13; fp32):
 *  - iterative box refinement (projects/SeqFormer/seqformer/models/deformable_transformer.py:366-380; IDOL :350-365):
 *      out[r] = sigmoid(delta[r] + inverse_sigmoid(reference[r]))        reference rows of 4 components, or
 *      out[r] = sigmoid((delta[r][:2] + inverse_sigmoid(reference[r]), delta[r][2:]))     of 2 (the first layer),
 *    inverse_sigmoid(x) = log(max(clamp(x, 0, 1), eps) / max(1 - clamp(x, 0, 1), eps))  (util/misc.py:493-497).
 *    delta, out [rows, 4]; reference [rows, ref_components].  Backward: grad_delta [rows, 4] and, unless null, grad_reference
 *    [rows, ref_components] (the clamps differentiated as autograd does: the gradient passes where min <= x <= max).
 *  - SeqFormer's temporal weighting of an instance query's frame-level context (:305-312):
 *      weights = softmax(logits, over the frames);  out[n, q, :] = sum_t weights[n, t, q] x[n, t, q, :]
 *    x [clips, frames, queries, channels] (channels a multiple of 4, frames <= 16), logits and weights [clips, frames,
 *    queries], out [clips, queries, channels].  Backward: grad_x (shape of x) and grad_logits from grad_out, x and the weights.
 */
int vnx_refine_boxes_forward(int dtype, const void* delta, const void* reference, void* out, long long rows, int ref_components,
                             float eps, void* hip_stream);
int vnx_refine_boxes_backward(int dtype, const void* grad_out, const void* out, const void* reference, void* grad_delta,
                              void* grad_reference, long long rows, int ref_components, float eps, void* hip_stream);
int vnx_time_weighted_sum_forward(int dtype, const void* x, const void* logits, void* out, void* weights, int clips, int frames,
                                  int queries, int channels, void* hip_stream);
int vnx_time_weighted_sum_backward(int dtype, const void* grad_out, const void* x, const void* weights, void* grad_x,
                                   void* grad_logits, int clips, int frames, int queries, int channels, void* hip_stream);

/* (The kernel-variant override of rounds 1-3 -- a process-wide A/B knob -- is no longer part of this library: it lives in
 *  the development build only, include/vnext_hip_dev.h.  Every call here selects its kernels from its own arguments.) */

#ifdef __cplusplus
}
#endif
#endif /* VNEXT_HIP_H_ */
