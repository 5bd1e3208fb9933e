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

/* Rate of the constant-rate wall clock the kernel-span stamps of the development build are taken in
 * (include/vnext_hip_dev.h: vnx_debug_arm_stamps), kHz; 0 when it cannot be read. */
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

/* The same check for the self-decoding grad_value kernel of calls below 1 024 queries (msda_d32_gvdirect.hip;
 * gvd_level_split in vnx_common.h; batch_heads = batch x heads decides whether small levels are cut in two).  Writes, per
 * (batch, head), the workgroups the kernel's level table yields and the
 * launcher's bound; per level (any of the three arrays may be null; `levels` entries each) the units, the rows per unit
 * and log2 of the 8-lane groups that share a row.  0 on success.  No GPU needed (tests/test_gvdirect_model.py). */
int vnx_debug_gvdirect_units(const int64_t* host_shapes, int levels, int num_query, int num_point, int batch_heads,
                             int* units_used, int* units_bound, int* level_units, int* level_rows_per_unit,
                             int* level_group_shift);

#ifdef __cplusplus
}
#endif
#endif /* VNEXT_HIP_DEBUG_H_ */
