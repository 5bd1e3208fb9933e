// msda_d32_gvdirect.hip -- grad_value, owner-computes, for calls with FEW queries (below 1 024: the decoders'),
// fed by nothing but the op's own inputs.
//
// Rounds 1-4 produced grad_value of such a call from per-sample records + unit tags the grad_loc kernel left in a
// workspace (msda_d32_gvrec.hip): the second kernel could not start before the first had finished, although it needs
// nothing that kernel computes -- only a decode of `sampling_loc` it can redo itself -- and it spent most of its life on
// the tags (load, ballots, scan, compaction: 8.1 of its 15 us at the T = 5 decoder call) and on a
// chunk loop (three chunks of 128 distinct queries for the units of a coarse level: the critical path).
//
// Here a unit (a pixel range of one level for one (batch, head); gvd_level_split in vnx_common.h) reads, in ONE pass,
//   * the locations and weights of its level's samples of every query -- 32 + 16 bytes per (query, head, level): 14 KB for
//     the 300 queries of a decoder call -- straight from `sampling_loc` / `attn_weight`, and decodes them (cuh:253-298);
//   * the grad_out rows of its head of every query (128 B each: 38 KB), streamed into LDS by `buffer_load ... lds`
//     (no register round trip) while the decode runs -- the first form; now through registers, requested behind the samples
//     (msda_d32_gvdirect_body.h says why);
// counting-sorts ALL taps that land in its rows by destination row (integer LDS atomics for ranks, DPP scan; the list
// holds the worst case, every sample on the unit's rows), and then 4-lane groups WALK the rows: a group sums one row's
// segment in eight registers and stores the row at once -- no rows kept in registers, no second barrier-separated chunk,
// nothing between the sort and the last store but LDS reads and 16-B stores.  No records, no tags, no workspace, no
// dependence on the grad_loc kernel.
// What bounds the form is what the units FETCH: each of them reads its level's samples (strided: 32 of every 1 024 bytes of
// `sampling_loc`) and stages all of its head's grad_out rows, so the traffic from L2 grows with the number of units.  The
// first form of the round had 24 units of <= 320 rows per (batch, head), rows in registers as in msda_d32_gvrec.hip:
// 18.9 us at the T = 5 decoder call, 6.5 us of it the loads (92 MB from L2 for 3.8 MB of inputs; timing ablations,
// DESIGN.md section 3.3e).  Units of up to VNX_GVD_ROWS rows need ten.
// Every level receives the same number of taps whatever its size: a row of a coarse level (80 taps at the 60-pixel level
// of a 300-query call) is spread over 1 << gshift adjacent groups whose partial sums meet through lane shuffles.
// A pass stages at most VNX_GVD_QC queries; calls with more run several passes, the later ones adding onto the rows the
// first stored (same owner, same lanes: no atomics).  Levels must be packed (checked on the device; capi.hip).
// Reference semantics: ms_deform_im2col_cuda.cuh:87-159 (the scatter this replaces), :253-298 (index decode).
#include "msda_d32_gvdirect_body.h"

namespace vnx {
namespace rec {

template <typename TV, typename TL, int P_T>
__global__ void __launch_bounds__(kThreads, VNX_GVD_UNITS_PER_CU * kWaves / 4)
msda_bwd_gv_direct_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                          const TL* __restrict__ loc, const TL* __restrict__ attn, const TV* __restrict__ grad_out,
                          TV* __restrict__ grad_value, MsdaDims d, int ut, int rows_max, int compact, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  msda_bwd_gv_direct_body<TV, TL, P_T>(shapes, lsi, loc, attn, grad_out, grad_value, d, ut, rows_max, compact, stamps, blockIdx.x, smem);
}

}  // namespace rec

#ifdef VNX_DEV_VARIANTS      // include/vnext_hip_dev.h: the phase stamps a -DVNX_GVD_STAMPS build leaves (zeros otherwise)
extern "C" int vnx_debug_read_gvd_stamps(unsigned long long* host, int n) {
#ifdef VNX_GVD_STAMPS
  return int(hipMemcpyFromSymbol(host, HIP_SYMBOL(rec::g_gvd_stamps), sizeof(unsigned long long) * size_t(n)));
#else
  for (int i = 0; i < n; ++i) host[i] = 0;
  return 0;
#endif
}
#endif

// Workgroups per (batch, head): the host knows S, not the level shapes.  A level of n pixels has
// max(ceil(n / ROWS), min(ut, n)) units (gvd_level_split): ceil(n / ROWS) + 1 bounds it for ut <= 2, and
// sum ceil(n_l / ROWS) <= S / ROWS + L.
int msda_gvdirect_units_bound(const MsdaDims& d, int ut, int rows) { return d.S / rows + d.L + (ut > 1 ? (ut - 1) * d.L : 0); }
int msda_gvdirect_units_bound(const MsdaDims& d) {      // the stand-alone launcher's
  const int rows = gvd_rows_max(d.S);
  return msda_gvdirect_units_bound(d, gvd_units_min(d.S, d.L, d.B * d.M, rows), rows);
}

bool msda_d32_gvdirect_supported(int vdt, int ldt, const MsdaDims& d) {
  if (d.D != 32 || vdt == VNX_F64) return false;
  if (vdt == VNX_F32 && ldt != VNX_F32) return false;
  if (d.L > rec::kLevelsMax || d.P > 64) return false;                 // a pass holds at least 20 queries
  if (d.S >= (1 << 27)) return false;                                  // units of a level in 18 bits
  // 24-bit stride multiplies; byte offsets into one batch element's locations and grad_out rows below 2^31 (buffer offsets)
  if (d.Lq >= (1 << 24) || d.M * 32 >= (1 << 24) || d.M * d.L * d.P >= (1 << 24)) return false;
  if (int64_t(d.Lq) * d.M * d.L * d.P * 8 >= (int64_t(1) << 31) || int64_t(d.Lq) * d.M * 32 * 4 >= (int64_t(1) << 31)) return false;
  // (the grid of either launcher: at most 2 L more units per (batch, head) than S / rows + L with the smallest rows in use)
  const int64_t blocks = (int64_t(d.B) * (d.S / VNX_GVD_ROWS_LARGE + 3 * d.L) + 1) * d.M;
  return blocks < (int64_t(1) << 30);
}

template <typename TV, typename TL>
static int launch_gvdirect(const int64_t* shapes, const int64_t* lsi, const void* loc, const void* attn, const void* grad_out,
                           void* grad_value, const MsdaDims& d, int compact, hipStream_t stream) {
#ifdef VNX_GVD_ROWS_ALONE      // A/B: rows per unit of the stand-alone launch
  const int rows = VNX_GVD_ROWS_ALONE;
#else
  const int rows = gvd_rows_max(d.S);
#endif
#ifdef VNX_GVD_UT_ALONE
  const int ut = VNX_GVD_UT_ALONE;
#else
  const int ut = gvd_units_min(d.S, d.L, d.B * d.M, rows);
#endif
  const int64_t blocks = ((int64_t(d.B) * msda_gvdirect_units_bound(d, ut, rows) + 1) & ~int64_t(1)) * d.M;   // (unit, batch) pairs: even (gv_decode_block)
  // more than 64 KiB of LDS per workgroup: the limit is raised once per kernel (and device: the attribute is per function)
#ifndef VNX_GVD_LDS_ALONE      // A/B: LDS bytes the stand-alone launch asks for (more than the kernel needs = fewer units per CU)
#define VNX_GVD_LDS_ALONE rec::kGvdLdsBytes
#endif
  const size_t lds = VNX_GVD_LDS_ALONE;
#define VNX_LAUNCH(PT)                                                                                                    \
  do {                                                                                                                    \
    static thread_local int raised_on = -1;                                                                               \
    int dev = 0;                                                                                                          \
    (void)hipGetDevice(&dev);                                                                                             \
    if (lds > 64 * 1024 && raised_on != dev) {                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rec::msda_bwd_gv_direct_kernel<TV, TL, PT>),                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess)                         \
        return check_launch("msda_bwd_gv_direct (LDS limit)");                                                             \
      raised_on = dev;                                                                                                    \
    }                                                                                                                     \
    hipLaunchKernelGGL((rec::msda_bwd_gv_direct_kernel<TV, TL, PT>), dim3(uint32_t(blocks)), dim3(rec::kThreads),        \
                       lds, stream, shapes, lsi, (const TL*)loc, (const TL*)attn, (const TV*)grad_out,     \
                       (TV*)grad_value, d, ut, rows, compact, take_stamp_region(kStampGradValue, blocks));                     \
  } while (0)
  if (d.P == 4) VNX_LAUNCH(4); else VNX_LAUNCH(0);
#undef VNX_LAUNCH
  return check_launch("msda_bwd_gv_direct");
}

// grad_value from the op's own inputs (compact = false) or from the fused grad_loc kernel's copy of the decoded locations /
// weights (compact = true: [batch][head][level][query][point], fp32); a no-op on the device when the levels are not packed.
int msda_backward_gvdirect_d32(int vdt, int ldt, const int64_t* shapes, const int64_t* lsi, const void* loc,
                               const void* attn, const void* grad_out, void* grad_value, MsdaDims d, bool compact,
                               hipStream_t stream) {
#define VNX_ARGS shapes, lsi, loc, attn, grad_out, grad_value, d, int(compact), stream
  if (vdt == VNX_F32) return launch_gvdirect<float, float>(VNX_ARGS);
  if (vdt == VNX_BF16 && ldt == VNX_F32) return launch_gvdirect<bf16_t, float>(VNX_ARGS);
  if (vdt == VNX_BF16 && ldt == VNX_BF16) return launch_gvdirect<bf16_t, bf16_t>(VNX_ARGS);
  if (vdt == VNX_F16 && ldt == VNX_F32) return launch_gvdirect<f16_t, float>(VNX_ARGS);
  if (vdt == VNX_F16 && ldt == VNX_F16) return launch_gvdirect<f16_t, f16_t>(VNX_ARGS);
#undef VNX_ARGS
  set_error("msda_backward_gvdirect_d32: unsupported dtype pair (%d, %d)", vdt, ldt);
  return VNX_ERR_INVALID_ARGUMENT;
}

}  // namespace vnx
