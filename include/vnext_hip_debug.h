/*
 * include/vnext_hip_debug.h -- measurement aids exported by libvnext_hip.so next to the drop-in ABI of
 * vnext_hip.h.  Nothing here is part of the boundary a VNext maintainer binds; bench.py and tools/ use
 * these to time kernels from inside (stamps) and to measure the memory system's own ceilings with the
 * access pattern of the kernels.  All device pointers, like the main ABI.
 */
#ifndef VNEXT_HIP_DEBUG_H_
#define VNEXT_HIP_DEBUG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Kernel-span stamps.  While a buffer is armed every launch of a tuned MSDA kernel takes a region of
 * 2 x gridDim 64-bit words and each workgroup leaves {its start, its last wave's end} there in
 * constant-rate wall-clock ticks (vnx_debug_wall_clock_khz).  buf: n_words zero-filled 64-bit words;
 * nullptr disarms.  vnx_debug_stamp_regions -> number of regions handed out since arming; per region the
 * kernel kind (1 forward, 2 grad_loc / grad_attn, 3 grad_value), word offset, workgroups.
 */
void vnx_debug_arm_stamps(void* buf, long long n_words);
int vnx_debug_stamp_regions(int* kinds, long long* offsets, long long* blocks, int n);
int vnx_debug_wall_clock_khz(void);

/*
 * Random row gather: the ceiling of the MSDA forward's access pattern on this memory system.  Reads
 * n_idx rows of 128 bytes, rows[idx[i]] (idx values < n_rows), with the kernels' lane map -- 8 lanes x
 * 16 B per row, 8 rows per wave instruction, `in_flight` (1..8) independent rows per lane before the first
 * use -- and folds them into sink[0..3] so that nothing is optimised away.  One launch on `hip_stream`;
 * the caller times it.  Returns VNX_OK or VNX_ERR_*.
 */
int vnx_debug_row_gather_probe(const void* rows, size_t n_rows, const uint32_t* idx, size_t n_idx, int in_flight,
                               float* sink, void* hip_stream);

/* Host-side view of the unit grid of the tile-fed grad_value kernel (vnext_amd/csrc/vnx_common.h: gv_level_grid;
 * msda_d32_gvtiles.hip): the launcher sizes the grid from the pixel count alone, the kernel derives the units from the
 * level shapes on the device -- the sum of the latter must never pass the former, or a unit's rows would be lost.
 * host_shapes: [levels][2] = (H, W) in HOST memory.  Writes the workgroups per (batch, head) the kernel will use and
 * the launcher's bound, and (either may be null) the fp32 partial rows per (batch, head) the query pieces of the split
 * levels store and the slab size the workspace reserves for them (gv_partial_rows_bound: a piece past its slab would
 * write into the next (batch, head)'s); 0 on success.  No GPU needed (tests/test_units_bound.py). */
int vnx_debug_gvtiles_units(const int64_t* host_shapes, int levels, int num_query, int batch, int heads, int units_min,
                            int* units_used, int* units_bound, long long* partial_rows_used, long long* partial_rows_bound);

#ifdef __cplusplus
}
#endif
#endif /* VNEXT_HIP_DEBUG_H_ */
