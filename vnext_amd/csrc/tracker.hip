// tracker.hip -- IDOL's online tracker, one frame per call, with the memory bank resident on the device
// ("tracker association on device").
//
// Reference: projects/IDOL/idol/models/tracker.py
//   :17-46    mask_iou / mask_nms      O(n^2) Python loop, one launch + one host sync per pair
//   :207-298  IDOL_Tracker.match       mask NMS, torch.mm + two softmaxes, a Python loop over detections with
//                                      `conf > thr` / `id > -1` host syncs, new ids, back-drop test, update_memo
//   :103-163  update_memo              per-tracklet momentum update, long_embed / long_score lists, expiry
//   :165-205  memo                     per-tracklet weighted mean of the remembered embeddings
// Here the tracklets live in a caller-owned device blob (slots), and a frame is four launches and no host
// round trip: masks -> bit words (ballot), intersections by popcount (exact integers; the reference sums
// int8 products), the [n, slots] similarity on the matrix cores (reid.hip), and ONE workgroup that runs the
// greedy passes: NMS on a bit matrix in LDS, bi-softmax restricted to the kept detections and the live slots,
// the sequential assignment on one wave (lanes across slots, DPP/shuffle arg-max, ties to the older
// tracklet as the reference's dict order gives), new ids, back-drops, memory update and expiry.
// The ids of a frame are written to device memory; the caller reads them whenever it wants (once per video).
#include "vnx_common.h"

namespace vnx {

constexpr int kTrkMaxDet = 512;     // detections per frame (the models emit <= 300 queries)
constexpr int kTrkMaxCap = 2048;    // tracklet slots
constexpr int kTrkMaxMem = 16;      // remembered embeddings per tracklet
constexpr int kTrkThreads = 1024;
constexpr int kTrkWords = kTrkMaxDet / 64;

struct TrkState {       // views into the caller's blob
  int* hdr;             // [0] tracklets created so far, [1] tracklets that found no free slot, [2] frames seen
  int* slot_id;         // [cap]  tracklet id, -1 = free
  int* last_frame;      // [cap]
  int* exist;           // [cap]  frames the tracklet was matched in ("exist_frame")
  int* label;           // [cap]
  int* long_len;        // [cap]  valid entries of long_embed / long_score, oldest first
  float* embed;         // [cap, C]       momentum embedding
  float* memo;          // [cap, C]       the embedding the next frame is matched against
  float* long_embed;    // [cap, mem, C]
  float* long_score;    // [cap, mem]
  size_t bytes;
};

static inline size_t up16(size_t x) { return (x + 15) & ~size_t(15); }

static TrkState trk_views(const vnx_tracker_config& c, void* base) {
  TrkState s;
  char* p = static_cast<char*>(base);
  size_t o = 0;
  auto take = [&](size_t bytes) { char* q = p + o; o += up16(bytes); return q; };
  const size_t cap = size_t(c.capacity), C = size_t(c.channels), mem = size_t(c.memory_len);
  s.hdr = reinterpret_cast<int*>(take(16 * 4));
  s.slot_id = reinterpret_cast<int*>(take(cap * 4));
  s.last_frame = reinterpret_cast<int*>(take(cap * 4));
  s.exist = reinterpret_cast<int*>(take(cap * 4));
  s.label = reinterpret_cast<int*>(take(cap * 4));
  s.long_len = reinterpret_cast<int*>(take(cap * 4));
  s.embed = reinterpret_cast<float*>(take(cap * C * 4));
  s.memo = reinterpret_cast<float*>(take(cap * C * 4));
  s.long_embed = reinterpret_cast<float*>(take(cap * mem * C * 4));
  s.long_score = reinterpret_cast<float*>(take(cap * mem * 4));
  s.bytes = o;
  return s;
}

struct TrkScratch { uint64_t* bits; int* inter; float* feats; float* scores; size_t bytes; };

static TrkScratch trk_scratch(int cap, int n, int pixels, void* base) {
  TrkScratch w;
  char* p = static_cast<char*>(base);
  size_t o = 0;
  auto take = [&](size_t bytes) { char* q = p + o; o += up16(bytes); return q; };
  const size_t words = size_t(pixels + 63) / 64;
  w.bits = reinterpret_cast<uint64_t*>(take(size_t(n) * words * 8));
  w.inter = reinterpret_cast<int*>(take(size_t(n) * n * 4));
  w.feats = reinterpret_cast<float*>(take(size_t(n) * cap * 4));
  w.scores = reinterpret_cast<float*>(take(size_t(n) * cap * 4));
  w.bytes = o;
  return w;
}

// ---- masks -> bits -> pairwise intersections ---------------------------------------------------------------
// sigmoid(x) > 0.5  <=>  x > 0 (tracker.py:33); one wave per 64 pixels, the ballot is the word.
__global__ void __launch_bounds__(256)
mask_pack_kernel(const float* __restrict__ logits, uint64_t* __restrict__ bits, int pixels, int words) {
  const int lane = threadIdx.x & 63;
  const int word = int(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (word >= words) return;
  const int i = blockIdx.y;
  const int px = word * 64 + lane;
  const bool on = px < pixels && logits[int64_t(i) * pixels + px] > 0.f;
  const uint64_t b = __ballot(on);
  if (lane == 0) bits[int64_t(i) * words + word] = b;
}

// inter[i, j] = |mask_i & mask_j| ; one wave per pair, lanes across the words.
__global__ void __launch_bounds__(256)
mask_inter_kernel(const uint64_t* __restrict__ bits, int* __restrict__ inter, int n, int words) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.y, j = int(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (j >= n || j < i) return;
  const uint64_t* a = bits + int64_t(i) * words;
  const uint64_t* b = bits + int64_t(j) * words;
  int acc = 0;
  for (int w = lane; w < words; w += 64) acc += __popcll(a[w] & b[w]);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) { inter[int64_t(i) * n + j] = acc; inter[int64_t(j) * n + i] = acc; }
}

// (intersection + 1e-6) / (union + 1e-6) in fp32, as mask_iou evaluates it (tracker.py:17-25)
__device__ __forceinline__ float iou_of(const int* inter, const int* area, int n, int i, int j) {
  const int in = inter[i * n + j];
  const int un = area[i] + area[j] - in;
  return __fdiv_rn(__fadd_rn(float(in), 1e-6f), __fadd_rn(float(un), 1e-6f));
}

struct Best { float v; int id; int j; };
__device__ __forceinline__ Best better(Best a, Best b) {   // larger value; ties to the older tracklet (smaller id)
  return (b.v > a.v || (b.v == a.v && b.id < a.id)) ? b : a;
}
__device__ __forceinline__ Best wave_best(Best x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Best o{__shfl_xor(x.v, off, 64), __shfl_xor(x.id, off, 64), __shfl_xor(x.j, off, 64)};
    x = better(x, o);
  }
  return x;
}

// ---- one frame -----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kTrkThreads)
tracker_frame_kernel(vnx_tracker_config cfg, TrkState st, const int* __restrict__ inter,
                     const float* __restrict__ feats, float* __restrict__ scores,
                     const float* __restrict__ embeds, const float* __restrict__ det_scores,
                     const int64_t* __restrict__ labels, int n, int frame_id, int64_t* __restrict__ ids_out) {
  __shared__ uint64_t s_over[kTrkMaxDet * kTrkWords];   // row i: detections j > i with IoU > nms_thr_pre
  __shared__ uint64_t s_keep[kTrkWords];
  __shared__ int s_area[kTrkMaxDet];
  __shared__ int s_ids[kTrkMaxDet];     // -3 removed by the NMS, -2 unassigned, -1 back-drop, >= 0 tracklet id
  __shared__ int s_slot[kTrkMaxDet];    // the slot a detection updates (matched) or fills (new); -1 none
  __shared__ int s_new[kTrkMaxDet];     // rank among this frame's new tracklets, -1 = not new
  __shared__ float s_rmax[kTrkMaxDet], s_rsum[kTrkMaxDet];
  __shared__ int s_free[kTrkMaxDet];
  __shared__ int s_misc[4];             // [0] new tracklets, [1] free slots found

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int kWaves = kTrkThreads / 64;
  const int cap = cfg.capacity, C = cfg.channels, mem = cfg.memory_len;
  const int nw = (n + 63) >> 6;

  // live slots at the start of the frame ("not self.empty", tracker.py:222)
  bool mine_active = false;
  for (int j = tid; j < cap; j += kTrkThreads) mine_active |= st.slot_id[j] >= 0;
  for (int i = tid; i < n; i += kTrkThreads) s_area[i] = inter[i * n + i];
  const bool any_active = __syncthreads_or(mine_active);

  // ---- mask NMS (tracker.py:28-46): the pair tests in parallel, the greedy pass on one wave
  for (int e = tid; e < n * nw; e += kTrkThreads) {
    const int i = e / nw, w = e - i * nw;
    uint64_t word = 0;
    for (int b = 0; b < 64; ++b) {
      const int j = w * 64 + b;
      if (j > i && j < n && iou_of(inter, s_area, n, i, j) > cfg.nms_thr_pre) word |= uint64_t(1) << b;
    }
    s_over[i * kTrkWords + w] = word;
  }
  __syncthreads();
  if (wave == 0) {
    uint64_t keep = 0;
    if (lane < nw) keep = (lane == nw - 1 && (n & 63)) ? ((uint64_t(1) << (n & 63)) - 1) : ~uint64_t(0);
    for (int i = 0; i + 1 < n; ++i) {
      const uint64_t owner = __shfl(keep, i >> 6, 64);
      if ((owner >> (i & 63)) & 1) {
        if (lane < nw) keep &= ~s_over[i * kTrkWords + lane];
      }
    }
    if (lane < nw) s_keep[lane] = keep;
  }
  __syncthreads();
  auto kept = [&](int i) { return (s_keep[i >> 6] >> (i & 63)) & 1; };
  for (int i = tid; i < n; i += kTrkThreads) { s_ids[i] = kept(i) ? -2 : -3; s_slot[i] = -1; s_new[i] = -1; }

  // ---- association scores over (kept detections) x (live slots)  (tracker.py:228-244)
  if (any_active) {
    if (cfg.match_metric != 2) {   // row soft-max statistics: one wave per kept detection
      for (int i = wave; i < n; i += kWaves) {
        if (!kept(i)) continue;
        float m = -INFINITY;
        for (int j = lane; j < cap; j += 64) if (st.slot_id[j] >= 0) m = fmaxf(m, feats[i * cap + j]);
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        float s = 0.f;
        for (int j = lane; j < cap; j += 64) if (st.slot_id[j] >= 0) s += expf(feats[i * cap + j] - m);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) { s_rmax[i] = m; s_rsum[i] = s; }
      }
    }
    __syncthreads();
    for (int j = tid; j < cap; j += kTrkThreads) {   // one thread per live slot, walking the kept detections
      if (st.slot_id[j] < 0) continue;
      float cm = -INFINITY, cs = 0.f;
      if (cfg.match_metric == 0) {
        for (int i = 0; i < n; ++i) if (kept(i)) cm = fmaxf(cm, feats[i * cap + j]);
        for (int i = 0; i < n; ++i) if (kept(i)) cs += expf(feats[i * cap + j] - cm);
      }
      for (int i = 0; i < n; ++i) {
        if (!kept(i)) continue;
        const float f = feats[i * cap + j];
        float v = f;
        if (cfg.match_metric == 0) v = 0.5f * (expf(f - s_rmax[i]) / s_rsum[i] + expf(f - cm) / cs);
        else if (cfg.match_metric == 1) v = expf(f - s_rmax[i]) / s_rsum[i];
        scores[i * cap + j] = v;
      }
    }
  }
  __syncthreads();

  // ---- sequential passes on one wave: assignment (tracker.py:245-263), new ids (:264-270 / :283-289)
  if (wave == 0) {
    const int cpl = (cap + 63) >> 6;    // slots per lane, slot = c * 64 + lane
    uint32_t live = 0, taken = 0;
    for (int c = 0; c < cpl; ++c) {
      const int j = c * 64 + lane;
      if (j < cap && st.slot_id[j] >= 0) live |= 1u << c;
    }
    if (any_active) {
      for (int i = 0; i < n; ++i) {
        if (!kept(i)) continue;
        Best plain{-INFINITY, 0x7fffffff, -1};
        int strong = 0;
        float fw_sum = 0.f;
        for (int c = 0; c < cpl; ++c) {
          if (!((live >> c) & 1)) continue;
          const int j = c * 64 + lane;
          const float v = ((taken >> c) & 1) ? 0.f : scores[i * cap + j];   // a taken slot reads 0 (:261-262)
          plain = better(plain, Best{v, st.slot_id[j], j});
          if (v > 0.5f) { ++strong; fw_sum += float(st.exist[j]); }
        }
        Best best = wave_best(plain);
        if (cfg.frame_weight) {
          for (int off = 32; off > 0; off >>= 1) { strong += __shfl_xor(strong, off, 64); fw_sum += __shfl_xor(fw_sum, off, 64); }
          if (strong > 1) {   // several candidates: weigh them by how long the tracklet has existed (:247-254)
            const float mean = __fdiv_rn(fw_sum, float(strong));
            Best w{-INFINITY, 0x7fffffff, -1};
            for (int c = 0; c < cpl; ++c) {
              if (!((live >> c) & 1)) continue;
              const int j = c * 64 + lane;
              const float v = ((taken >> c) & 1) ? 0.f : scores[i * cap + j];
              w = better(w, Best{__fmul_rn(v, v > 0.5f ? float(st.exist[j]) : mean), st.slot_id[j], j});
            }
            best = wave_best(w);
          }
        }
        if (best.j >= 0 && best.v > cfg.match_score_thr) {
          if ((best.j & 63) == lane) taken |= 1u << (best.j >> 6);
          if (lane == 0) { s_ids[i] = best.id; s_slot[i] = best.j; }
        }
      }
    }
    // new tracklets: unassigned detections above the score bar, numbered in detection order
    const float bar = any_active ? cfg.addnew_score_thr : cfg.init_score_thr;
    const int created = st.hdr[0];
    int fresh = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int i = c0 + lane;
      const bool is_new = i < n && s_ids[i] == -2 && det_scores[i] > bar;
      const uint64_t b = __ballot(is_new);
      if (is_new) {
        const int rank = fresh + __popcll(b & ((uint64_t(1) << lane) - 1));
        s_ids[i] = created + rank;
        s_new[i] = rank;
      }
      fresh += __popcll(b);
    }
    // ... and the free slots they move into, lowest first
    int found = 0;
    for (int c = 0; c < cpl && found < fresh; ++c) {
      const int j = c * 64 + lane;
      const bool is_free = j < cap && !((live >> c) & 1);
      const uint64_t b = __ballot(is_free);
      if (is_free) {
        const int k = found + __popcll(b & ((uint64_t(1) << lane) - 1));
        if (k < fresh) s_free[k] = j;
      }
      found += __popcll(b);
    }
    if (lane == 0) { s_misc[0] = fresh; s_misc[1] = min(found, fresh); }
  }
  __syncthreads();
  const int fresh = s_misc[0], found = s_misc[1];

  // ---- back-drops (tracker.py:272-279): an unassigned detection that overlaps no earlier kept one
  for (int i = tid; i < n; i += kTrkThreads) {
    if (s_ids[i] != -2) continue;
    bool alone = true;
    for (int j = 0; j < i; ++j)
      if (kept(j) && !(iou_of(inter, s_area, n, i, j) < cfg.nms_thr_post)) { alone = false; break; }
    if (alone) s_ids[i] = -1;
  }
  for (int i = tid; i < n; i += kTrkThreads)
    if (s_new[i] >= 0) s_slot[i] = s_new[i] < found ? s_free[s_new[i]] : -1;
  __syncthreads();

  // ---- update_memo (tracker.py:103-163): one wave per detection that carries a tracklet id
  const float keep_old = float(1.0 - double(cfg.memo_momentum)), take_new = cfg.memo_momentum;
  for (int i = wave; i < n; i += kWaves) {
    const int id = s_ids[i], s = s_slot[i];
    if (id < 0 || s < 0) continue;
    const bool is_new = s_new[i] >= 0;
    const int len = is_new ? 0 : st.long_len[s];
    const bool shift = len == mem;               // append, then drop the oldest entry (:148-151)
    const int pos = shift ? mem - 1 : len, new_len = pos + 1;
    const float score = det_scores[i];
    // the remembered scores after the append, oldest first, and the weights of the long-term embedding:
    // score (+ 1/L, 2/L, ..., 1 = torch.range(0, 1, 1/L)[1:], :181-184)
    float raw[kTrkMaxMem], wt[kTrkMaxMem];
#pragma unroll
    for (int p = 0; p < kTrkMaxMem; ++p) {
      raw[p] = (p < pos) ? st.long_score[s * mem + p + (shift ? 1 : 0)] : score;
      wt[p] = cfg.temporal_weight ? __fadd_rn(raw[p], float(double(p + 1) * (1.0 / double(new_len)))) : raw[p];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the old scores are in registers before lane 0 rewrites them
    float wsum = 0.f;
#pragma unroll
    for (int p = 0; p < kTrkMaxMem; ++p) if (p < new_len) wsum = __fadd_rn(wsum, wt[p]);
    float* L = st.long_embed + size_t(s) * mem * C;
    for (int c = lane; c < C; c += 64) {
      const float e = embeds[i * C + c];
      // (1 - momentum) * old + momentum * new, three roundings as torch evaluates it (:118-120)
      const float cur = is_new ? e : __fadd_rn(__fmul_rn(keep_old, st.embed[s * C + c]), __fmul_rn(take_new, e));
      st.embed[s * C + c] = cur;
      float m = cur;
      if (cfg.long_match) {       // sum_p long_embed[p] * w[p] / sum_p w[p]  (:185)
        float acc = 0.f;
#pragma unroll
        for (int p = 0; p < kTrkMaxMem; ++p) {
          if (p >= new_len) continue;
          const float row = (p == pos) ? e : L[(p + (shift ? 1 : 0)) * C + c];
          acc = __fadd_rn(acc, __fmul_rn(row, wt[p]));
          if (shift || p == pos) L[p * C + c] = row;      // append, dropping the oldest entry when full (:148-151)
        }
        m = __fdiv_rn(acc, wsum);
      } else {
        for (int p = 0; p < new_len; ++p) {
          const float row = (p == pos) ? e : L[(p + (shift ? 1 : 0)) * C + c];
          if (shift || p == pos) L[p * C + c] = row;
        }
      }
      st.memo[s * C + c] = m;
    }
    if (lane == 0) {
#pragma unroll
      for (int p = 0; p < kTrkMaxMem; ++p) if (p < new_len) st.long_score[s * mem + p] = raw[p];
      st.long_len[s] = new_len;
      st.last_frame[s] = frame_id;
      st.label[s] = int(labels[i]);
      st.exist[s] = is_new ? 1 : st.exist[s] + 1;
      if (is_new) st.slot_id[s] = id;
    }
  }
  __syncthreads();

  // ---- expiry (tracker.py:141-146,152-153) and the frame's record
  for (int j = tid; j < cap; j += kTrkThreads)
    if (st.slot_id[j] >= 0 && frame_id - st.last_frame[j] >= cfg.memo_tracklet_frames) st.slot_id[j] = -1;
  for (int i = tid; i < n; i += kTrkThreads) ids_out[i] = s_ids[i];
  if (tid == 0) {
    st.hdr[0] += fresh;
    st.hdr[1] += fresh - found;
    st.hdr[2] += 1;
  }
}

static int check_config(const vnx_tracker_config* c, const char* who) {
  if (!c) { set_error("%s: null config", who); return VNX_ERR_INVALID_ARGUMENT; }
  if (c->capacity < 1 || c->capacity > kTrkMaxCap || c->channels < 4 || (c->channels % 4) ||
      c->memory_len < 1 || c->memory_len > kTrkMaxMem || c->match_metric < 0 || c->match_metric > 2 ||
      c->memo_tracklet_frames < 0 || !(c->memo_momentum >= 0.f && c->memo_momentum <= 1.f)) {
    set_error("%s: config out of range (capacity 1..%d, channels multiple of 4, memory_len 1..%d, "
              "match_metric 0..2, memo_momentum 0..1)", who, kTrkMaxCap, kTrkMaxMem);
    return VNX_ERR_INVALID_ARGUMENT;
  }
  return VNX_OK;
}

}  // namespace vnx

using namespace vnx;

extern "C" size_t vnx_tracker_state_bytes(const vnx_tracker_config* cfg) {
  if (check_config(cfg, "vnx_tracker_state_bytes") != VNX_OK) return 0;
  return trk_views(*cfg, nullptr).bytes;
}

extern "C" int vnx_tracker_reset(const vnx_tracker_config* cfg, void* state, void* hip_stream) {
  if (int st = check_config(cfg, "vnx_tracker_reset")) return st;
  if (!state || (uintptr_t(state) % 16)) {
    set_error("vnx_tracker_reset: state must be a 16-byte aligned device pointer");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const TrkState s = trk_views(*cfg, state);
  hipStream_t stream = (hipStream_t)hip_stream;
  if (hipMemsetAsync(state, 0, s.bytes, stream) != hipSuccess ||
      hipMemsetAsync(s.slot_id, 0xff, size_t(cfg->capacity) * 4, stream) != hipSuccess) {
    set_error("vnx_tracker_reset: hipMemsetAsync failed");
    return VNX_ERR_LAUNCH;
  }
  return VNX_OK;
}

extern "C" size_t vnx_tracker_frame_workspace_bytes(const vnx_tracker_config* cfg, int num_dets, int mask_pixels) {
  if (check_config(cfg, "vnx_tracker_frame_workspace_bytes") != VNX_OK || num_dets < 0 || mask_pixels < 0) return 0;
  return trk_scratch(cfg->capacity, num_dets, mask_pixels, nullptr).bytes + 16;
}

static int launch_mask_intersections(const float* logits, int n, int pixels, uint64_t* bits, int* inter,
                                     hipStream_t stream) {
  const int words = (pixels + 63) / 64;
  hipLaunchKernelGGL(mask_pack_kernel, dim3(uint32_t((words + 3) / 4), uint32_t(n)), dim3(256), 0, stream,
                     logits, bits, pixels, words);
  hipLaunchKernelGGL(mask_inter_kernel, dim3(uint32_t((n + 3) / 4), uint32_t(n)), dim3(256), 0, stream,
                     (const uint64_t*)bits, inter, n, words);
  return check_launch("mask_intersections");
}

extern "C" size_t vnx_mask_intersections_workspace_bytes(int num_masks, int mask_pixels) {
  if (num_masks < 0 || mask_pixels < 0) return 0;
  return up16(size_t(num_masks) * (size_t(mask_pixels + 63) / 64) * 8) + 16;
}

extern "C" int vnx_mask_intersections(const float* mask_logits, int num_masks, int mask_pixels, int32_t* inter,
                                      void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (num_masks < 0 || mask_pixels < 1 || num_masks > 65535) {
    set_error("vnx_mask_intersections: bad sizes num_masks=%d mask_pixels=%d", num_masks, mask_pixels);
    return VNX_ERR_INVALID_ARGUMENT;
  }
  if (num_masks == 0) return VNX_OK;
  if (!mask_logits || !inter || !workspace || workspace_bytes < vnx_mask_intersections_workspace_bytes(num_masks, mask_pixels)) {
    set_error("vnx_mask_intersections: null pointer or workspace smaller than vnx_mask_intersections_workspace_bytes");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  uint64_t* bits = reinterpret_cast<uint64_t*>((uintptr_t(workspace) + 15) & ~uintptr_t(15));
  return launch_mask_intersections(mask_logits, num_masks, mask_pixels, bits, inter, (hipStream_t)hip_stream);
}

extern "C" int vnx_tracker_frame(const vnx_tracker_config* cfg, void* state, const float* mask_logits,
                                 const float* embeds, const float* det_scores, const int64_t* labels,
                                 int num_dets, int mask_pixels, int frame_id, int64_t* ids_out,
                                 void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (int st = check_config(cfg, "vnx_tracker_frame")) return st;
  if (num_dets < 0 || num_dets > kTrkMaxDet || mask_pixels < 1) {
    set_error("vnx_tracker_frame: built for up to %d detections per frame (got %d), mask_pixels=%d", kTrkMaxDet,
              num_dets, mask_pixels);
    return num_dets > kTrkMaxDet ? VNX_ERR_UNSUPPORTED : VNX_ERR_INVALID_ARGUMENT;
  }
  if (num_dets == 0) return VNX_OK;   // the reference leaves the memory untouched on an empty frame (:222,281)
  if (!state || !mask_logits || !embeds || !det_scores || !labels || !ids_out || !workspace) {
    set_error("vnx_tracker_frame: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  if (workspace_bytes < vnx_tracker_frame_workspace_bytes(cfg, num_dets, mask_pixels) || (uintptr_t(state) % 16) ||
      (uintptr_t(embeds) % 16)) {
    set_error("vnx_tracker_frame: workspace smaller than vnx_tracker_frame_workspace_bytes, or state / embeds not "
              "16-byte aligned");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  const TrkState s = trk_views(*cfg, state);
  const TrkScratch w = trk_scratch(cfg->capacity, num_dets, mask_pixels,
                                   reinterpret_cast<void*>((uintptr_t(workspace) + 15) & ~uintptr_t(15)));
  if (int st = launch_mask_intersections(mask_logits, num_dets, mask_pixels, w.bits, w.inter, stream)) return st;
  if (int st = vnx_reid_similarity(VNX_F32, embeds, s.memo, w.feats, num_dets, cfg->capacity, cfg->channels,
                                   cfg->channels, cfg->channels, cfg->capacity, cfg->match_metric == 2, hip_stream))
    return st;
  hipLaunchKernelGGL(tracker_frame_kernel, dim3(1), dim3(kTrkThreads), 0, stream, *cfg, s, (const int*)w.inter,
                     (const float*)w.feats, w.scores, embeds, det_scores, labels, num_dets, frame_id, ids_out);
  return check_launch("tracker_frame");
}
