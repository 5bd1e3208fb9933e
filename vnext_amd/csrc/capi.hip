// capi.hip -- the extern "C" surface of libvnext_hip.so (include/vnext_hip.h).
// Argument validation, kernel selection and error reporting live here; the
// kernels themselves are in msda_generic.hip / msda_d32.hip.
#include <stdarg.h>
#include <stdio.h>

#include "vnx_common.h"
#include "../../include/vnext_hip_debug.h"
#ifdef VNX_DEV_VARIANTS
#include "../../include/vnext_hip_dev.h"
#endif

namespace vnx {

static thread_local char t_error[512] = "";
#ifdef VNX_DEV_VARIANTS
std::atomic<int> g_kernel_variant{0};   // development build: A/B knob (vnx_set_kernel_variant); every entry point reads it once
#endif

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(t_error, sizeof t_error, fmt, ap);
  va_end(ap);
}

// The reference only printf()s a failed launch (ms_deform_im2col_cuda.cuh:948-952);
// here it becomes a status the caller must look at.
int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return VNX_ERR_LAUNCH;
  }
  return VNX_OK;
}

int msda_forward_generic(int, int, const void*, const int64_t*, const int64_t*, const void*,
                         const void*, void*, MsdaDims, hipStream_t);
int msda_backward_generic(int, int, const void*, const int64_t*, const int64_t*, const void*,
                          const void*, const void*, void*, void*, void*, MsdaDims,
                          int only_if_not_packed, hipStream_t);
int convert_f32_to(int, const void*, void*, int64_t, const int64_t*, const int64_t*, int, int,
                   hipStream_t);
int zero_if_not_packed(const int64_t*, const int64_t*, int, int, void*, size_t, hipStream_t);
bool msda_d32_fwd_supported(int vdt, int ldt, const MsdaDims& d);
#ifdef VNX_DEV_VARIANTS      // the LDS-staged forwards: tools/experiments/msda_tile/ (development build only)
bool msda_tile_fwd_supported(int vdt, int ldt, const MsdaDims& d);
int msda_forward_tile(const void*, const int64_t*, const int64_t*, const void*, const void*, void*, MsdaDims,
                      const FusedArgs*, int debug, hipStream_t);
bool msda_tile2_fwd_supported(int vdt, int ldt, const MsdaDims& d);
int msda_forward_tile2(const void*, const int64_t*, const int64_t*, const void*, const void*, void*, MsdaDims, hipStream_t);
#endif
bool msda_d32_bwd_supported(int vdt, int ldt, const MsdaDims& d);
int msda_forward_d32(int, int, const void*, const int64_t*, const int64_t*, const void*,
                     const void*, void*, MsdaDims, int variant, hipStream_t);
int msda_backward_d32(int, int, const void*, const int64_t*, const int64_t*, const void*,
                      const void*, const void*, void*, void*, void*, MsdaDims, int variant,
                      void* records, void* tile_summary, float* tile_copy, hipStream_t);
int msda_bwd_tile_queries(const MsdaDims& d, int variant);
bool msda_d32_gvtiles_supported(int vdt, int ldt, const MsdaDims& d);
size_t msda_gvtiles_summary_bytes(const MsdaDims& d, int tile_queries);
size_t msda_gvtiles_partial_bytes(const MsdaDims& d);
int msda_backward_gvtiles_d32(int vdt, int ldt, const int64_t*, const int64_t*, const void* loc, const void* attn,
                              const void* summaries, const void* grad_out, void* grad_value, MsdaDims,
                              int tile_queries, float* partials, bool compact, hipStream_t);
bool msda_d32_fused_supported(int vdt, int ldt, const MsdaDims& d);
int msda_fused_d32(bool backward, int vdt, int ldt, const void* value, const int64_t* shapes, const int64_t* lsi,
                   const void* raw_off, const void* raw_logit, const void* grad_out, void* out_or_grad_off,
                   void* grad_logit, MsdaDims d, void* records, const void* reference, float* grad_reference,
                   int ref_dim, int ref_div, void* grad_value_f32, void* tile_summary, float* tile_loc, float* tile_attn,
                   hipStream_t stream, int ref_f32);
bool msda_d32_gvrec_supported(int vdt, int ldt, const MsdaDims& d);
size_t msda_gvrec_record_bytes(const MsdaDims& d);
int msda_backward_gvrec_d32(int vdt, const int64_t*, const int64_t*, const void* records, const void*,
                            void*, MsdaDims, int variant, float* split_image, hipStream_t);
int msda_split_levels_convert(int vdt, const int64_t*, const int64_t*, const float* image, void* grad_value, MsdaDims, bool tiles,
                              hipStream_t);
bool msda_d32_gvdirect_supported(int vdt, int ldt, const MsdaDims& d);
bool msda_backward_pair_supported(int vdt, int ldt, const MsdaDims& d);
int msda_backward_pair_d32(int vdt, const void* value, const int64_t*, const int64_t*, const void* loc, const void* attn,
                           const void* grad_out, void* grad_value, void* grad_loc, void* grad_attn, MsdaDims, int order,
                           hipStream_t);
int msda_backward_gvdirect_d32(int vdt, int ldt, const int64_t*, const int64_t*, const void* loc, const void* attn,
                               const void* grad_out, void* grad_value, MsdaDims, bool compact, hipStream_t);

static int check_common(const char* fn, int vdt, int ldt, const void* value,
                        const int64_t* shapes, const int64_t* lsi, const void* loc,
                        const void* attn, const MsdaDims& d) {
  if (elem_size(vdt) == 0) {
    set_error("%s: unknown value_dtype %d", fn, vdt);
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const bool pair_ok = (ldt == vdt) || (ldt == VNX_F32 && (vdt == VNX_BF16 || vdt == VNX_F16));
  if (!pair_ok) {
    set_error("%s: loc_dtype %d must equal value_dtype %d, or be f32 with a 16-bit value", fn,
              ldt, vdt);
    return VNX_ERR_INVALID_ARGUMENT;
  }
  if (d.B < 0 || d.S < 0 || d.Lq < 0 || d.M <= 0 || d.D <= 0 || d.L <= 0 || d.P <= 0) {
    set_error("%s: bad sizes batch=%d spatial=%d heads=%d channels=%d levels=%d query=%d point=%d",
              fn, d.B, d.S, d.M, d.D, d.L, d.Lq, d.P);
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const bool empty = (d.B == 0 || d.Lq == 0);
  if (!shapes || !lsi || (!empty && (!loc || !attn)) || (!empty && d.S > 0 && !value)) {
    set_error("%s: null pointer argument", fn);
    return VNX_ERR_INVALID_ARGUMENT;
  }
  // level-local pixel offsets and per-batch element offsets are 32-bit in the kernels
  if (int64_t(d.S) * d.M * d.D >= (int64_t(1) << 31)) {
    set_error("%s: spatial_size*heads*channels = %lld does not fit 31 bits", fn,
              (long long)(int64_t(d.S) * d.M * d.D));
    return VNX_ERR_UNSUPPORTED;
  }
  return VNX_OK;
}

}  // namespace vnx

using namespace vnx;

namespace vnx {
// Units per level at least, tile-fed path.  1 since round 4: a level that fits one unit (<= 256 pixels: the coarsest level of
// both pyramids, and the 240-pixel level of 360p) is ONE rectangle cut into query pieces, not two half-rectangles that each
// scan and decode every tile of the level -- with the pieces meeting in partial rows instead of atomics the second row-unit
// buys nothing and costs the level's reads twice: encoder backward 166.5 -> 163.2 us at 360p, 622 -> 611 us at 720p B = 5
// (720p B = 2: 271.4 / 272.1, unchanged).  (Round 3 measured 2 against 1 as 175.2 vs 176.4 us: then the pieces flushed
// through atomics.)  Also re-measured on the partial-row scheme: a piece per 5 chunks instead of 10 190 us, per 20 chunks
// 194 us (360p) / 265.6 us (720p B = 2, - 6); 4 pieces for the middle levels 171 / 293 us.
#ifndef VNX_TILE_UNITS_MIN
#define VNX_TILE_UNITS_MIN 1
#endif
int gv_units_min(const MsdaDims& d, bool tiles, int v) {
  if (v >= 200 && v < 300) return v - 200 < 1 ? 1 : (v - 200 > 16 ? 16 : v - 200);
  // Tried with per-unit selection: enough units that each expects about one selection window of
  // samples (10 per level at the encoder shape).  Slower on MI355X -- 353 vs 332 us per encoder-shape
  // backward -- because a unit's chunk count is set by its DISTINCT queries (128 staged rows per
  // chunk), and finer units multiply the (query, unit) incidences.
  (void)d;
  return tiles ? VNX_TILE_UNITS_MIN : 2;
}
}  // namespace vnx

// ---- the side stream of the forked backward (include/vnext_hip.h: VNX_MSDA_FORK) -----------------------------------
// One lane per host thread and device, created on first use and kept: a non-blocking stream and two timing-less events.
// Creating them is not a stream operation, but a thread in a global-mode stream capture may not call "unsafe" runtime
// functions: the creation runs in relaxed mode.  No lane (creation failed) = the call stays on the caller's stream.
namespace vnx {
struct SideLane { int device; hipStream_t side; hipEvent_t fork, join; };
static SideLane* side_lane() {
  static thread_local SideLane lanes[16];
  static thread_local int n_lanes = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  for (int i = 0; i < n_lanes; ++i)
    if (lanes[i].device == dev) return lanes[i].side ? &lanes[i] : nullptr;
  if (n_lanes >= 16) return nullptr;
  SideLane& l = lanes[n_lanes++];
  l = SideLane{dev, nullptr, nullptr, nullptr};
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
  bool ok = hipStreamCreateWithFlags(&l.side, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&l.fork, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&l.join, hipEventDisableTiming) == hipSuccess;
  if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode);
  if (!ok) {
    (void)hipGetLastError();
    if (l.fork) (void)hipEventDestroy(l.fork);
    if (l.side) (void)hipStreamDestroy(l.side);
    l.side = nullptr;
    return nullptr;
  }
  return &l;
}
}  // namespace vnx

// ---- kernel-span stamps (development build only: include/vnext_hip_dev.h) ----------------------------------------
// The armed buffer is process-wide state, so it does not exist in the product library: there every launch passes a null
// stamp pointer and nothing outside a call's arguments can change what the call does or costs.
#ifdef VNX_DEV_VARIANTS
static unsigned long long* g_stamp_buf = nullptr;
static long long g_stamp_words = 0, g_stamp_used = 0;
static int g_stamp_n = 0;
static int g_stamp_kind[4096];
static long long g_stamp_off[4096], g_stamp_blocks[4096];
#endif
namespace vnx {
unsigned long long* take_stamp_region(int kernel, long long blocks) {
#ifdef VNX_DEV_VARIANTS
  if (!g_stamp_buf || g_stamp_n >= 4096 || g_stamp_used + 2 * blocks > g_stamp_words) return nullptr;
  g_stamp_kind[g_stamp_n] = kernel;
  g_stamp_off[g_stamp_n] = g_stamp_used;
  g_stamp_blocks[g_stamp_n] = blocks;
  ++g_stamp_n;
  unsigned long long* p = g_stamp_buf + g_stamp_used;
  g_stamp_used += 2 * blocks;
  return p;
#else
  (void)kernel; (void)blocks;
  return nullptr;
#endif
}
}  // namespace vnx

extern "C" {

int vnx_abi_version(void) { return VNX_ABI_VERSION; }

const char* vnx_status_string(int status) {
  switch (status) {
    case VNX_OK: return "ok";
    case VNX_ERR_INVALID_ARGUMENT: return "invalid argument";
    case VNX_ERR_UNSUPPORTED: return "unsupported shape";
    case VNX_ERR_WORKSPACE: return "workspace missing or too small";
    case VNX_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown status";
  }
}

const char* vnx_last_error(void) { return t_error; }

#ifdef VNX_DEV_VARIANTS
void vnx_set_kernel_variant(int variant) { g_kernel_variant.store(variant, std::memory_order_relaxed); }
int vnx_get_kernel_variant(void) { return g_kernel_variant.load(std::memory_order_relaxed); }
#endif

#ifdef VNX_DEV_VARIANTS
// buf: device memory of n_words 64-bit words, ZERO-filled by the caller before every measured run
// (slots of workgroups that never ran stay {0, 0} and are skipped); nullptr disarms
void vnx_debug_arm_stamps(void* buf, long long n_words) {
  g_stamp_buf = (unsigned long long*)buf;
  g_stamp_words = n_words;
  g_stamp_used = 0;
  g_stamp_n = 0;
}
// -> number of regions handed out since arming; per region: kernel kind (1 forward, 2
// grad_loc/attn, 3 grad_value), word offset into the buffer, number of workgroups
int vnx_debug_stamp_regions(int* kinds, long long* offsets, long long* blocks, int n) {
  for (int i = 0; i < n && i < g_stamp_n; ++i) {
    kinds[i] = g_stamp_kind[i];
    offsets[i] = g_stamp_off[i];
    blocks[i] = g_stamp_blocks[i];
  }
  return g_stamp_n;
}
#endif
int vnx_debug_wall_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
  return khz;
}

// The LDS-staged forwards (north-star row n1; tools/experiments/msda_tile/) were measured over
// two rounds against the per-query L2-gather kernel -- 64-65 vs 55 us at encoder-360p, within 2 % at 720p -- and are retired
// from the product build: the development build keeps them reachable (variants 700..702 / 720) with their parity tests.
#ifdef VNX_DEV_VARIANTS
static bool use_tile_forward(int vdt, int ldt, const MsdaDims& d, int variant) {
  return variant >= 700 && variant <= 702 && msda_tile_fwd_supported(vdt, ldt, d);
}
static bool use_tile2_forward(int vdt, int ldt, const MsdaDims& d, int variant) {
  return variant == 720 && msda_tile2_fwd_supported(vdt, ldt, d);
}
#endif

int vnx_msda_forward(int value_dtype, int loc_dtype, const void* value,
                     const int64_t* spatial_shapes, const int64_t* level_start_index,
                     const void* sampling_loc, const void* attn_weight, void* output, int batch,
                     int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                     int num_point, void* hip_stream) {
  const MsdaDims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  int st = check_common("vnx_msda_forward", value_dtype, loc_dtype, value, spatial_shapes,
                        level_start_index, sampling_loc, attn_weight, d);
  if (st != VNX_OK) return st;
  if (batch == 0 || num_query == 0) return VNX_OK;
  if (!output) {
    set_error("vnx_msda_forward: null output");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  const int variant = kernel_variant();
#ifdef VNX_DEV_VARIANTS
  if (use_tile2_forward(value_dtype, loc_dtype, d, variant))
    return msda_forward_tile2(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, output, d, stream);
  if (use_tile_forward(value_dtype, loc_dtype, d, variant))
    return msda_forward_tile(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, output, d, nullptr,
                             variant - 700, stream);
#endif
  if (variant != 1 && msda_d32_fwd_supported(value_dtype, loc_dtype, d))
    return msda_forward_d32(value_dtype, loc_dtype, value, spatial_shapes, level_start_index,
                            sampling_loc, attn_weight, output, d, variant, stream);
  return msda_forward_generic(value_dtype, loc_dtype, value, spatial_shapes, level_start_index,
                              sampling_loc, attn_weight, output, d, stream);
}

static bool bwd_fast_path(int vdt, int ldt, const MsdaDims& d, int variant) {
  return variant != 1 && !(variant >= 300 && variant < 400) &&
         msda_d32_bwd_supported(vdt, ldt, d) && msda_d32_gvrec_supported(vdt, ldt, d);
}

// Workspace layout of the backward: [sample records or tile words (fast path, 256-B aligned size) | fp32
// grad_value image (16-bit values whenever the general path may run)].
static size_t align256(size_t n) { return (n + 255) & ~size_t(255); }

// Which grad_value path a fast-path call takes: per-sample records + per-unit selection (msda_d32_gvrec.hip: calls
// with few, scattered queries -- the decoders') or per-tile words (msda_d32_gvtiles.hip: from 1 024 queries up -- the
// encoders').  Measured on MI355X, T = 5 encoder calls, model-like locations, cold: see DESIGN.md section 3.3b.
// Variants 430 / 431 force records / tiles for A/B runs; the variants that name a record-fed kernel keep it.
static bool use_tiles(int vdt, int ldt, const MsdaDims& d, int variant) {
  if (variant == 430 || variant == 408 || variant == 412 || variant == 420 || variant == 425) return false;
  if (d.P != 4 || d.L * d.P != 16 || !msda_d32_gvtiles_supported(vdt, ldt, d)) return false;
  return variant == 431 || d.Lq >= 1024;
}
// Calls below 1 024 queries (the decoders'): grad_value by the self-decoding kernel (msda_d32_gvdirect.hip) -- no records, no
// tags, no workspace, and no dependence on the grad_loc kernel, so the two run concurrently (side_lane below).  The
// development build keeps the record-fed kernels reachable for A/B runs (variant 430 and the variants that name one).
#ifndef VNX_PAIR_ORDER
#define VNX_PAIR_ORDER -1     // role order of the paired backward kernel's workgroup groups (msda_d32.hip): -1 = the launcher's choice
#endif
static bool use_direct(int vdt, int ldt, const MsdaDims& d, int variant) {
  if (variant == 430 || variant == 408 || variant == 412 || variant == 420 || variant == 425) return false;
  if (use_tiles(vdt, ldt, d, variant)) return false;
  // built for calls below 1 024 queries: beyond, every unit would re-stage all grad_out rows of its head once per pass of
  // VNX_GVD_QC queries, and 16-bit rows would be rounded once per pass -- such calls (L * P != 16 or P != 4 at encoder sizes) keep
  // the record-fed path, which accumulates in fp32
  if (d.Lq >= 1024) return false;
  return msda_d32_gvdirect_supported(vdt, ldt, d);
}
// RECORD-fed path with 16-bit values and enough queries for the query split of the coarse levels (gv_query_splits): the
// pieces of such a level meet through fp32 atomics, which need an fp32 target -- the same [B, S, M, 32] fp32 image the
// general path of unpacked levels uses (the two never run on the same call: one needs packed levels, the other unpacked
// ones).  The tile-fed path (every call the models make with >= 1 024 queries) needs none since round 4: its pieces store
// fp32 partial rows (msda_gvtiles_partial_bytes) and a finishing kernel writes grad_value in its own dtype.
static bool split_image_needed(int vdt, int ldt, const MsdaDims& d, int variant) {
  return (vdt == VNX_BF16 || vdt == VNX_F16) && d.P == 4 && d.Lq >= 1024 && !use_tiles(vdt, ldt, d, variant) &&
         !use_direct(vdt, ldt, d, variant);
}
// Tile path: does the grad_loc kernel leave a copy of the locations / weights laid out for the grad_value kernel
// ([batch][head][level][query][point], fp32, 12 B per sample)?  The fused backward always does (it has to materialise
// them anyway: 196.7 -> 194.9 us at encoder-360p against the op's own layout).  The plain backward does not: the 12 B per
// sample the grad_loc kernel then writes cost more than the grad_value kernel's tidier reads save (encoder-360p 186 vs
// 175 us, 720p B = 2 337 vs 322 us, B = 5 664 vs 654 us -- although that kernel fetches 2 GB per 720p launch, PMC).
// Variant 510 forces the copy for A/B runs.
static bool tile_copy_wanted(const MsdaDims& d, int variant) {
  (void)d;
  return variant == 510;
}
static size_t tile_copy_bytes(const MsdaDims& d) { return align256(size_t(12) * size_t(d.B) * d.Lq * d.M * d.L * d.P); }
// tile path: [tile words | location copy (variant 510 only) | partial rows of the query-split levels' pieces]
static size_t tiles_partials_offset(const MsdaDims& d, int variant) {
  return align256(msda_gvtiles_summary_bytes(d, msda_bwd_tile_queries(d, variant))) + (tile_copy_wanted(d, variant) ? tile_copy_bytes(d) : 0);
}
static size_t fast_path_scratch_bytes(int vdt, int ldt, const MsdaDims& d, int variant) {
  if (use_tiles(vdt, ldt, d, variant)) return tiles_partials_offset(d, variant) + align256(msda_gvtiles_partial_bytes(d));
  if (use_direct(vdt, ldt, d, variant)) return 0;
  return align256(msda_gvrec_record_bytes(d));
}

size_t vnx_msda_backward_workspace_bytes(int value_dtype, int loc_dtype, int batch,
                                         int spatial_size, int num_heads, int channels,
                                         int num_levels, int num_query, int num_point, int flags) {
  const MsdaDims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  const int variant = kernel_variant();
  const bool sixteen = (value_dtype == VNX_BF16 || value_dtype == VNX_F16);
  const size_t image = sixteen ? sizeof(float) * size_t(batch) * size_t(spatial_size) * size_t(num_heads) * size_t(channels) : 0;
  if (!bwd_fast_path(value_dtype, loc_dtype, d, variant)) return image;
  const size_t records = fast_path_scratch_bytes(value_dtype, loc_dtype, d, variant);
  // packed levels promised: the general path never runs, no fp32 image -- unless the query split needs it
  return records + (((flags & VNX_MSDA_LEVELS_PACKED) && !split_image_needed(value_dtype, loc_dtype, d, variant)) ? 0 : image);
}

int vnx_msda_backward(int value_dtype, int loc_dtype, const void* value,
                      const int64_t* spatial_shapes, const int64_t* level_start_index,
                      const void* sampling_loc, const void* attn_weight, const void* grad_output,
                      void* grad_value, void* grad_sampling_loc, void* grad_attn_weight, int batch,
                      int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                      int num_point, int flags, void* workspace, size_t workspace_bytes,
                      void* hip_stream) {
  const MsdaDims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  int st = check_common("vnx_msda_backward", value_dtype, loc_dtype, value, spatial_shapes,
                        level_start_index, sampling_loc, attn_weight, d);
  if (st != VNX_OK) return st;
  hipStream_t stream = (hipStream_t)hip_stream;
  const int variant = kernel_variant();
  const size_t n_value = size_t(batch) * size_t(spatial_size) * size_t(num_heads) * size_t(channels);
  const size_t need = vnx_msda_backward_workspace_bytes(value_dtype, loc_dtype, batch, spatial_size,
                                                        num_heads, channels, num_levels, num_query,
                                                        num_point, flags);
  if (n_value > 0 && !grad_value) {
    set_error("vnx_msda_backward: null grad_value");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const bool empty = (batch == 0 || num_query == 0);
  if (need > 0 && !empty && (!workspace || workspace_bytes < need)) {
    set_error("vnx_msda_backward: needs %zu workspace bytes (vnx_msda_backward_workspace_bytes), got %zu",
              need, workspace_bytes);
    return VNX_ERR_WORKSPACE;
  }
  if (!empty && (!grad_output || !grad_sampling_loc || !grad_attn_weight)) {
    set_error("vnx_msda_backward: null gradient pointer");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const bool sixteen = (value_dtype == VNX_BF16 || value_dtype == VNX_F16);
  const size_t image_bytes = sixteen ? sizeof(float) * n_value : 0;
  if (empty) {  // no queries: the gradient of value is all zeros, the other two are empty
    if (n_value > 0 &&
        hipMemsetAsync(grad_value, 0, n_value * size_t(elem_size(value_dtype)), stream) != hipSuccess) {
      set_error("vnx_msda_backward: hipMemsetAsync failed");
      return VNX_ERR_LAUNCH;
    }
    return VNX_OK;
  }

  if (bwd_fast_path(value_dtype, loc_dtype, d, variant)) {
    // (1) grad_loc / grad_attn: per-query gather kernel, no atomics; it also leaves one 16-B
    //     geometry record per sample in the workspace.
    // (2) grad_value: owner-computes units fed by those records; does nothing on the device
    //     unless the levels are packed.
    // (3) unless the caller promised packed levels: the general path, each kernel of which
    //     does nothing on the device when the levels ARE packed.  No host sync either way.
    // (The record-less predecessor of (2) -- every unit re-deriving its level's geometry -- and the side
    //  stream that overlapped it with (1) are archived under tools/experiments/msda_d32_gv.hip.)
    if (use_direct(value_dtype, loc_dtype, d, variant)) {
      // (1) grad_value from the op's own inputs and (2) grad_loc / grad_attn: neither reads what the other writes.  One after
      // the other on the caller's stream -- or, with VNX_MSDA_FORK, (1) on the side stream between two events and (2) on the
      // caller's stream, which then waits for (1).  Development build: 441 = fork, 442 = (1) alone, 100..199 = (2) alone (timing).
      // (Launching (2) without the packet's barrier bit -- hipExtAnyOrderLaunch, same stream, no events -- was tried: the flag is
      // not honoured on gfx9 parts (hip_ext.h says so; measured 23.70 vs 23.93 us eager, no overlap in the kernel trace).)
      const bool only_gl = variant >= 100 && variant < 200;
      const bool only_gv = variant == 442;
      // Both halves as ONE launch where the paired kernel is built for the call (msda_d32.hip: msda_bwd_pair_kernel; fp32,
      // L*P == 16, the one-wave grad_loc configuration -- the decoders' calls): the grad_value units first, the grad_loc work in the
      // remaining workgroups, sharing the GPU without a second queue.  Development build: 444 = the two launches instead,
      // 445 / 446 / 447 = the paired kernel with the grad_value groups first / the grad_loc groups first / alternating.
      if (!only_gl && !only_gv && !(flags & VNX_MSDA_FORK) && variant != 441 && variant != 444 &&
          msda_backward_pair_supported(value_dtype, loc_dtype, d) && (!sixteen || (flags & VNX_MSDA_LEVELS_PACKED))) {
        const int order = variant == 445 ? 0 : variant == 446 ? 1 : variant == 447 ? 2 : VNX_PAIR_ORDER;
        st = msda_backward_pair_d32(value_dtype, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, grad_value,
                                    grad_sampling_loc, grad_attn_weight, d, order, stream);
        if (st != VNX_OK) return st;
        if (!(flags & VNX_MSDA_LEVELS_PACKED)) {      // the general path, every kernel of which does nothing when the levels ARE packed
          st = zero_if_not_packed(spatial_shapes, level_start_index, num_levels, spatial_size, grad_value,
                                  n_value * size_t(elem_size(value_dtype)), stream);
          if (st != VNX_OK) return st;
          st = msda_backward_generic(value_dtype, loc_dtype, value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                     grad_output, grad_value, grad_sampling_loc, grad_attn_weight, d, /*only_if_not_packed=*/1, stream);
          if (st != VNX_OK) return st;
        }
        return VNX_OK;
      }
      SideLane* lane = (((flags & VNX_MSDA_FORK) || variant == 441) && !only_gl && !only_gv) ? side_lane() : nullptr;
      if (lane) {
        if (hipEventRecord(lane->fork, stream) != hipSuccess || hipStreamWaitEvent(lane->side, lane->fork, 0) != hipSuccess) {
          (void)hipGetLastError();
          lane = nullptr;
        }
      }
      int st_gv = VNX_OK;
      if (!only_gl)
        st_gv = msda_backward_gvdirect_d32(value_dtype, loc_dtype, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                           grad_output, grad_value, d, false, lane ? lane->side : stream);
      if (lane && hipEventRecord(lane->join, lane->side) != hipSuccess) {
        set_error("vnx_msda_backward: hipEventRecord on the side stream failed");
        st_gv = VNX_ERR_LAUNCH;
      }
      st = VNX_OK;
      if (!only_gv)
        st = msda_backward_d32(value_dtype, loc_dtype, value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                               grad_output, nullptr, grad_sampling_loc, grad_attn_weight, d,
                               only_gl ? variant : 100 + (variant < 100 ? variant : 0), nullptr, nullptr, nullptr, stream);
      if (lane && hipStreamWaitEvent(stream, lane->join, 0) != hipSuccess) {      // the join: always, once forked
        set_error("vnx_msda_backward: hipStreamWaitEvent on the caller's stream failed");
        return VNX_ERR_LAUNCH;
      }
      if (st_gv != VNX_OK) return st_gv;
      if (st != VNX_OK) return st;
      if (!(flags & VNX_MSDA_LEVELS_PACKED) && !only_gl && !only_gv) {
        // the general path, every kernel of which does nothing on the device when the levels ARE packed
        void* gv_acc = sixteen ? workspace : grad_value;
        const size_t acc_bytes = sixteen ? image_bytes : n_value * size_t(elem_size(value_dtype));
        st = zero_if_not_packed(spatial_shapes, level_start_index, num_levels, spatial_size, gv_acc, acc_bytes, stream);
        if (st != VNX_OK) return st;
        st = msda_backward_generic(value_dtype, loc_dtype, value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                   grad_output, gv_acc, grad_sampling_loc, grad_attn_weight, d, /*only_if_not_packed=*/1, stream);
        if (st != VNX_OK) return st;
        if (sixteen)
          return convert_f32_to(value_dtype, workspace, grad_value, int64_t(n_value), spatial_shapes, level_start_index,
                                num_levels, spatial_size, stream);
      }
      return VNX_OK;
    }
    const bool tiles = use_tiles(value_dtype, loc_dtype, d, variant);
    const size_t rec_bytes = fast_path_scratch_bytes(value_dtype, loc_dtype, d, variant);
    void* records = tiles ? nullptr : workspace;
    void* tile_words = tiles ? workspace : nullptr;
    float* tile_copy = (tiles && tile_copy_wanted(d, variant))
                           ? (float*)((char*)workspace + align256(msda_gvtiles_summary_bytes(d, msda_bwd_tile_queries(d, variant))))
                           : nullptr;
    const bool split16 = split_image_needed(value_dtype, loc_dtype, d, variant);
    void* image = (sixteen && (!(flags & VNX_MSDA_LEVELS_PACKED) || split16)) ? (void*)((char*)workspace + rec_bytes) : nullptr;
    float* split_image = split16 ? (float*)image : nullptr;
    float* partials = tiles ? (float*)((char*)workspace + tiles_partials_offset(d, variant)) : nullptr;
    const bool only_gl = variant >= 100 && variant < 200;  // timing ablations
    const bool only_gv = (variant >= 400 && variant < 430) || (variant > 431 && variant < 500);
    // records mode: the accumulation-image argument carries the fp32 target of the query pieces' atomics (grad_value itself
    // or the split image), whose rows of the query-split levels the kernel zeroes (gv_query_splits); tile mode: nothing
    st = msda_backward_d32(value_dtype, loc_dtype, value, spatial_shapes, level_start_index,
                           sampling_loc, attn_weight, grad_output,
                           tiles ? nullptr : (value_dtype == VNX_F32 ? grad_value : (void*)split_image), grad_sampling_loc,
                           grad_attn_weight, d, only_gl ? variant : 100 + (variant < 100 ? variant : 0),
                           records, tile_words, tile_copy, stream);
    if (st != VNX_OK) return st;
    if (!only_gl) {
      if (tiles)
        st = tile_copy
                 ? msda_backward_gvtiles_d32(value_dtype, VNX_F32, spatial_shapes, level_start_index, tile_copy,
                                             tile_copy + 2 * (int64_t(d.B) * d.Lq * d.M * d.L * d.P), tile_words, grad_output,
                                             grad_value, d, msda_bwd_tile_queries(d, variant), partials, true, stream)
                 : msda_backward_gvtiles_d32(value_dtype, loc_dtype, spatial_shapes, level_start_index, sampling_loc,
                                             attn_weight, tile_words, grad_output, grad_value, d,
                                             msda_bwd_tile_queries(d, variant), partials, false, stream);
      else
        st = msda_backward_gvrec_d32(value_dtype, spatial_shapes, level_start_index, records, grad_output,
                                     grad_value, d, variant, split_image, stream);
      if (st != VNX_OK) return st;
      if (split_image) {
        st = msda_split_levels_convert(value_dtype, spatial_shapes, level_start_index, split_image, grad_value, d,
                                       tiles, stream);
        if (st != VNX_OK) return st;
      }
    }
    if (!(flags & VNX_MSDA_LEVELS_PACKED) && !only_gl && !only_gv) {
      void* gv_acc = sixteen ? image : grad_value;
      const size_t acc_bytes = sixteen ? image_bytes : n_value * size_t(elem_size(value_dtype));
      st = zero_if_not_packed(spatial_shapes, level_start_index, num_levels, spatial_size, gv_acc,
                              acc_bytes, stream);
      if (st != VNX_OK) return st;
      st = msda_backward_generic(value_dtype, loc_dtype, value, spatial_shapes, level_start_index,
                                 sampling_loc, attn_weight, grad_output, gv_acc, grad_sampling_loc,
                                 grad_attn_weight, d, /*only_if_not_packed=*/1, stream);
      if (st != VNX_OK) return st;
      if (sixteen)
        return convert_f32_to(value_dtype, image, grad_value, int64_t(n_value), spatial_shapes,
                              level_start_index, num_levels, spatial_size, stream);
    }
    return VNX_OK;
  }

  void* gv_acc = sixteen ? workspace : grad_value;
  const size_t acc_bytes = sixteen ? image_bytes : n_value * size_t(elem_size(value_dtype));
  // general path: zero-filled image + hardware fp32/fp64 atomics
  if (acc_bytes > 0) {
    const hipError_t e = hipMemsetAsync(gv_acc, 0, acc_bytes, stream);
    if (e != hipSuccess) {
      set_error("vnx_msda_backward: hipMemsetAsync failed: %s", hipGetErrorString(e));
      return VNX_ERR_LAUNCH;
    }
  }
  if (variant >= 300 && variant < 400 && msda_d32_bwd_supported(value_dtype, loc_dtype, d))
    st = msda_backward_d32(value_dtype, loc_dtype, value, spatial_shapes, level_start_index,
                           sampling_loc, attn_weight, grad_output, gv_acc, grad_sampling_loc,
                           grad_attn_weight, d, variant - 300, nullptr, nullptr, nullptr, stream);
  else
    st = msda_backward_generic(value_dtype, loc_dtype, value, spatial_shapes, level_start_index,
                               sampling_loc, attn_weight, grad_output, gv_acc, grad_sampling_loc,
                               grad_attn_weight, d, /*only_if_not_packed=*/0, stream);
  if (st != VNX_OK) return st;
  if (sixteen)
    return convert_f32_to(value_dtype, workspace, grad_value, int64_t(n_value), nullptr, nullptr, 0, 0,
                          stream);
  return VNX_OK;
}


// ---- fused prologue (include/vnext_hip.h) -----------------------------------------------------
static int check_fused(const char* fn, int value_dtype, int q_dtype, const MsdaDims& d, int ref_dim, int ref_div) {
  if (d.B < 0 || d.S < 0 || d.M <= 0 || d.D <= 0 || d.L <= 0 || d.Lq < 0 || d.P <= 0) {
    set_error("%s: bad sizes", fn);
    return VNX_ERR_INVALID_ARGUMENT;
  }
  if ((ref_dim != 2 && ref_dim != 4) || ref_div <= 0 || (d.B % ref_div) != 0) {
    set_error("%s: reference points must have 2 or 4 components and batch must be a multiple of "
              "reference_batch_div (got ref_dim=%d, batch=%d, div=%d)", fn, ref_dim, d.B, ref_div);
    return VNX_ERR_INVALID_ARGUMENT;
  }
  if (!msda_d32_fused_supported(value_dtype, q_dtype, d) || !msda_d32_gvrec_supported(value_dtype, q_dtype, d)) {
    set_error("%s: the fused prologue is built for 32-channel heads, levels*points == 16, fp32 or bf16 "
              "(got D=%d, L*P=%d, dtypes %d/%d); use vnx_msda_forward/backward", fn, d.D, d.L * d.P,
              value_dtype, q_dtype);
    return VNX_ERR_UNSUPPORTED;
  }
  return VNX_OK;
}

int vnx_msda_fused_forward(int value_dtype, int query_dtype, const void* value, const int64_t* spatial_shapes,
                           const int64_t* level_start_index, const void* sampling_offsets,
                           const void* attention_logits, const void* reference_points, void* output, int batch,
                           int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                           int num_point, int ref_dim, int reference_batch_div, void* hip_stream) {
  const int ref_f32 = (ref_dim & VNX_MSDA_REF_F32) != 0;      // fp32 reference points beside 16-bit offsets / logits (ABI 14)
  ref_dim &= ~VNX_MSDA_REF_F32;
  const MsdaDims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  const int st = check_fused("vnx_msda_fused_forward", value_dtype, query_dtype, d, ref_dim, reference_batch_div);
  if (st != VNX_OK) return st;
  if (batch == 0 || num_query == 0) return VNX_OK;
  if (!value || !spatial_shapes || !level_start_index || !sampling_offsets || !attention_logits ||
      !reference_points || !output) {
    set_error("vnx_msda_fused_forward: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
#ifdef VNX_DEV_VARIANTS
  if (use_tile_forward(value_dtype, query_dtype, d, kernel_variant())) {
    const FusedArgs fa{reference_points, nullptr, ref_dim, reference_batch_div, nullptr};
    return msda_forward_tile(value, spatial_shapes, level_start_index, sampling_offsets, attention_logits, output, d, &fa, 0,
                             (hipStream_t)hip_stream);
  }
#endif
  return msda_fused_d32(false, value_dtype, query_dtype, value, spatial_shapes, level_start_index, sampling_offsets,
                        attention_logits, nullptr, output, nullptr, d, nullptr, reference_points, nullptr, ref_dim,
                        reference_batch_div, nullptr, nullptr, nullptr, nullptr, (hipStream_t)hip_stream, ref_f32);
}

// Scratch of the fused backward.  Record-fed grad_value: the sample records.  Tile-fed (use_tiles; the fused launcher
// always takes the automatic configuration, hence tile queries of variant 0): [tile words | decoded locations, fp32,
// 8 B per sample | softmax weights, fp32, 4 B per sample] -- the two tensors the fused prologue otherwise never
// materialises, 12 B per sample against the records' 16 + 4.
struct FusedScratch { bool tiles, direct; size_t words, loc, attn, partials, total; };
static FusedScratch fused_scratch(int vdt, const MsdaDims& d, int variant) {
  FusedScratch f{};
  f.tiles = use_tiles(vdt, VNX_F32, d, variant);
  f.direct = !f.tiles && use_direct(vdt, VNX_F32, d, variant);
  if (f.direct) {      // self-decoding grad_value kernel (below 1 024 queries): only the decoded locations / weights, fp32
    const size_t samples = size_t(d.B) * d.Lq * d.M * d.L * d.P;
    f.loc = align256(samples * 8);
    f.attn = align256(samples * 4);
    f.total = f.loc + f.attn;
  } else if (f.tiles) {
    const size_t samples = size_t(d.B) * d.Lq * d.M * d.L * d.P;
    f.words = align256(msda_gvtiles_summary_bytes(d, msda_bwd_tile_queries(d, 0)));
    f.loc = align256(samples * 8);
    f.attn = align256(samples * 4);
    f.partials = align256(msda_gvtiles_partial_bytes(d));      // the query pieces' partial rows, behind the three
    f.total = f.words + f.loc + f.attn + f.partials;
  } else {
    f.total = align256(msda_gvrec_record_bytes(d));
  }
  return f;
}

size_t vnx_msda_fused_backward_workspace_bytes(int value_dtype, int batch, int spatial_size, int num_heads, int num_levels,
                                               int num_query, int num_point) {
  const MsdaDims d{batch, spatial_size, num_heads, 32, num_levels, num_query, num_point};
  // the larger of the two layouts: the variant may change between this call and the backward (A/B runs); the record-fed
  // one (variant 430) needs the fp32 split image for 16-bit values
  const size_t image = split_image_needed(value_dtype, VNX_F32, d, 430) ? sizeof(float) * size_t(batch) * size_t(spatial_size) * num_heads * 32 : 0;
  const size_t a = fused_scratch(value_dtype, d, 430).total + image, b = fused_scratch(value_dtype, d, 0).total;
  return a > b ? a : b;
}

int vnx_msda_fused_backward(int value_dtype, int query_dtype, const void* value, const int64_t* spatial_shapes,
                            const int64_t* level_start_index, const void* sampling_offsets,
                            const void* attention_logits, const void* reference_points, const void* grad_output,
                            void* grad_value, void* grad_sampling_offsets, void* grad_attention_logits,
                            float* grad_reference_points, int batch, int spatial_size, int num_heads,
                            int channels, int num_levels, int num_query, int num_point, int ref_dim,
                            int reference_batch_div, void* workspace, size_t workspace_bytes, void* hip_stream) {
  const int ref_f32 = (ref_dim & VNX_MSDA_REF_F32) != 0;
  ref_dim &= ~VNX_MSDA_REF_F32;
  const MsdaDims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  int st = check_fused("vnx_msda_fused_backward", value_dtype, query_dtype, d, ref_dim, reference_batch_div);
  if (st != VNX_OK) return st;
  hipStream_t stream = (hipStream_t)hip_stream;
  if (grad_reference_points && (ref_dim != 2 || reference_batch_div != 1)) {
    set_error("vnx_msda_fused_backward: reference-point gradients are built for 2-d, per-batch references");
    return VNX_ERR_UNSUPPORTED;
  }
  if (grad_reference_points && batch > 0 && num_query > 0 &&
      hipMemsetAsync(grad_reference_points, 0, size_t(batch) * num_query * num_levels * 2 * sizeof(float), stream) !=
          hipSuccess)
    return check_launch("msda_fused_backward memset");
  if (batch == 0 || num_query == 0) {
    // no samples: grad_value is all zeros
    const size_t nbytes = size_t(batch) * spatial_size * num_heads * channels * size_t(elem_size(value_dtype));
    if (nbytes && hipMemsetAsync(grad_value, 0, nbytes, stream) != hipSuccess) return check_launch("memset");
    return VNX_OK;
  }
  if (!value || !spatial_shapes || !level_start_index || !sampling_offsets || !attention_logits ||
      !reference_points || !grad_output || !grad_value || !grad_sampling_offsets || !grad_attention_logits) {
    set_error("vnx_msda_fused_backward: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  const int variant = kernel_variant();
  const FusedScratch fs = fused_scratch(value_dtype, d, variant);
  const size_t rec_bytes = fs.total;
  const bool split16 = split_image_needed(value_dtype, VNX_F32, d, variant);
  const size_t need = rec_bytes + (split16 ? sizeof(float) * size_t(batch) * size_t(spatial_size) * num_heads * 32 : 0);
  if (!workspace || workspace_bytes < need) {
    set_error("vnx_msda_fused_backward: workspace of %zu bytes needed (got %zu)", need, workspace_bytes);
    return VNX_ERR_WORKSPACE;
  }
  float* split_image = split16 ? (float*)((char*)workspace + rec_bytes) : nullptr;
  void* fp32_target = (fs.tiles || fs.direct) ? nullptr : (value_dtype == VNX_F32 ? grad_value : (void*)split_image);
  if (fs.direct) {
    // (1) grad of the Linear outputs (+ reference points) and the decoded locations / softmax weights, laid out
    // [batch][head][level][query][point]; (2) grad_value by the self-decoding kernel from those (msda_d32_gvdirect.hip):
    // 12 B per sample where the record-fed kernel was left 16 + 4, no tags to select by, no chunk loop
    float* d_loc = (float*)workspace;
    float* d_attn = (float*)((char*)workspace + fs.loc);
    st = msda_fused_d32(true, value_dtype, query_dtype, value, spatial_shapes, level_start_index, sampling_offsets,
                        attention_logits, grad_output, grad_sampling_offsets, grad_attention_logits, d, nullptr,
                        reference_points, grad_reference_points, ref_dim, reference_batch_div, nullptr, nullptr, d_loc, d_attn,
                        stream, ref_f32);
    if (st != VNX_OK) return st;
    return msda_backward_gvdirect_d32(value_dtype, VNX_F32, spatial_shapes, level_start_index, d_loc, d_attn, grad_output,
                                      grad_value, d, true, stream);
  }
  if (fs.tiles) {
    // (1) grad of the Linear outputs (+ reference points), one word per (level, tile of queries) and the decoded
    // locations / weights; (2) grad_value from those.  Packed levels are required (no-op on the device otherwise).
    void* words = workspace;
    float* tile_loc = (float*)((char*)workspace + fs.words);
    float* tile_attn = (float*)((char*)workspace + fs.words + fs.loc);
    float* partials = (float*)((char*)workspace + fs.words + fs.loc + fs.attn);
    st = msda_fused_d32(true, value_dtype, query_dtype, value, spatial_shapes, level_start_index, sampling_offsets,
                        attention_logits, grad_output, grad_sampling_offsets, grad_attention_logits, d, nullptr,
                        reference_points, grad_reference_points, ref_dim, reference_batch_div, fp32_target, words, tile_loc,
                        tile_attn, stream, ref_f32);
    if (st != VNX_OK) return st;
    st = msda_backward_gvtiles_d32(value_dtype, VNX_F32, spatial_shapes, level_start_index, tile_loc, tile_attn, words,
                                   grad_output, grad_value, d, msda_bwd_tile_queries(d, 0), partials, true, stream);
  } else {
    // (1) grad of the Linear outputs (+ reference points) and the sample records; (2) grad_value from the records
    st = msda_fused_d32(true, value_dtype, query_dtype, value, spatial_shapes, level_start_index, sampling_offsets,
                        attention_logits, grad_output, grad_sampling_offsets, grad_attention_logits, d, workspace,
                        reference_points, grad_reference_points, ref_dim, reference_batch_div, fp32_target, nullptr, nullptr,
                        nullptr, stream, ref_f32);
    if (st != VNX_OK) return st;
    st = msda_backward_gvrec_d32(value_dtype, spatial_shapes, level_start_index, workspace, grad_output,
                                 grad_value, d, variant, split_image, stream);
  }
  if (st != VNX_OK) return st;
  if (split_image)
    return msda_split_levels_convert(value_dtype, spatial_shapes, level_start_index, split_image, grad_value, d,
                                     fs.tiles, stream);
  return VNX_OK;
}

}  // extern "C"

// ---- unit grid of the tile-fed grad_value kernel, seen from the host (include/vnext_hip_debug.h) -----------
namespace vnx { int msda_gvtiles_units_bound(const MsdaDims& d, int units_min); }
extern "C" int vnx_debug_gvtiles_units(const int64_t* host_shapes, int levels, int num_query, int batch, int heads,
                                       int units_min, int* units_used, int* units_bound, long long* partial_rows_used,
                                       long long* partial_rows_bound) {
  if (!host_shapes || levels <= 0 || !units_used || !units_bound) return VNX_ERR_INVALID_ARGUMENT;
  int64_t S = 0, rows = 0;
  int used = 0;
  for (int l = 0; l < levels; ++l) {
    const int H = int(host_shapes[2 * l]), W = int(host_shapes[2 * l + 1]);
    S += int64_t(H) * W;
    const int units = gv_level_units(H, W, units_min, true);
    const int qs = gv_query_splits(units, num_query, 4, true, batch * heads);      // the kernel's level table
    used += units * qs;
    if (qs > 1) rows += int64_t(qs) * H * W;      // the partial rows the level's query pieces store (gv_partial_rows_bound)
  }
  const MsdaDims d{batch, int(S), heads, 32, levels, num_query, 4};
  *units_used = used;
  *units_bound = vnx::msda_gvtiles_units_bound(d, units_min);
  if (partial_rows_used) *partial_rows_used = rows;
  if (partial_rows_bound) *partial_rows_bound = gv_partial_rows_bound(int(S), levels, batch * heads);
  return VNX_OK;
}

// ---- unit grid of the self-decoding grad_value kernel, seen from the host (include/vnext_hip_debug.h) -----------
namespace vnx { int msda_gvdirect_units_bound(const MsdaDims& d); }
extern "C" int vnx_debug_gvdirect_units(const int64_t* host_shapes, int levels, int num_query, int num_point, int batch_heads,
                                        int* units_used, int* units_bound, int* level_units, int* level_rows_per_unit,
                                        int* level_group_shift) {
  if (!host_shapes || levels <= 0 || !units_used || !units_bound) return VNX_ERR_INVALID_ARGUMENT;
  int64_t S = 0;
  for (int l = 0; l < levels; ++l) S += host_shapes[2 * l] * host_shapes[2 * l + 1];
  const int rows = gvd_rows_max(int(S));
  const int ut = gvd_units_min(int(S), levels, batch_heads, rows);
  int used = 0;
  for (int l = 0; l < levels; ++l) {
    const int H = int(host_shapes[2 * l]), W = int(host_shapes[2 * l + 1]);
    const GvdSplit sp = gvd_level_split(H * W, ut, num_query, num_point, rows);      // the kernel's level table
    used += sp.units;
    if (level_units) level_units[l] = sp.units;
    if (level_rows_per_unit) level_rows_per_unit[l] = sp.rpu;
    if (level_group_shift) level_group_shift[l] = sp.gshift;
  }
  const MsdaDims d{batch_heads, int(S), 1, 32, levels, num_query, num_point};
  *units_used = used;
  *units_bound = vnx::msda_gvdirect_units_bound(d);
  return VNX_OK;
}

// ---- row-gather probe (include/vnext_hip_debug.h) ---------------------------------------------------
namespace vnx {
typedef float probe_f4 __attribute__((ext_vector_type(4)));
template <int NF>
__global__ void __launch_bounds__(256) row_gather_probe_kernel(const probe_f4* __restrict__ rows, const uint32_t* __restrict__ idx,
                                                                size_t n_idx, float* __restrict__ sink) {
  const size_t set = (blockIdx.x * size_t(blockDim.x) + threadIdx.x) >> 3, sets = (size_t(gridDim.x) * blockDim.x) >> 3;
  const int ch = threadIdx.x & 7;
  probe_f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = set * NF; i + NF <= n_idx; i += sets * NF) {
    probe_f4 v[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) v[j] = rows[size_t(idx[i + j]) * 8 + ch];
#pragma unroll
    for (int j = 0; j < NF; ++j) acc += v[j];
  }
  if (acc.x == 1.2345e-31f) { sink[0] = acc.x; sink[1] = acc.y; sink[2] = acc.z; sink[3] = acc.w; }
}
}  // namespace vnx

extern "C" int vnx_debug_row_gather_probe(const void* rows, size_t n_rows, const uint32_t* idx, size_t n_idx, int in_flight,
                                          float* sink, void* hip_stream) {
  if (!rows || !idx || !sink || n_rows == 0) {
    vnx::set_error("vnx_debug_row_gather_probe: null pointer argument");
    return VNX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t st = (hipStream_t)hip_stream;
  const dim3 grid(8192), block(256);
#define VNX_PROBE(NF) hipLaunchKernelGGL((vnx::row_gather_probe_kernel<NF>), grid, block, 0, st, (const vnx::probe_f4*)rows, idx, n_idx, sink)
  if (in_flight >= 8) VNX_PROBE(8); else if (in_flight >= 4) VNX_PROBE(4); else if (in_flight >= 2) VNX_PROBE(2); else VNX_PROBE(1);
#undef VNX_PROBE
  return vnx::check_launch("row_gather_probe");
}
