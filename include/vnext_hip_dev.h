/*
 * include/vnext_hip_dev.h -- what the DEVELOPMENT build (vnext_amd/lib/libvnext_hip_dev.so: the product sources
 * compiled with -DVNX_DEV_VARIANTS plus the archived kernels of tools/experiments/msda_tile/) exports on top of
 * vnext_hip.h and vnext_hip_debug.h.  The product library (libvnext_hip.so) exports NONE of this: there a call
 * selects its kernels from its own arguments and nothing process-wide can change what it computes.
 *
 * Used by tests/ (parity of every kernel form, not only the automatically selected one), tools/kbench.hip and
 * the python tools (A/B timing).
 */
#ifndef VNEXT_HIP_DEV_H_
#define VNEXT_HIP_DEV_H_

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Kernel selection override (process-wide, development build only):
 *   0 = automatic (what the product library always does), 1 = force the generic kernels,
 *   2..69 forced wave / workgroup shapes of the tuned forward and grad_loc kernels, 200..216 grad_value units per
 *   level, 420 / 425 / 430 / 431 grad_value paths, 510 compact location copy, 700..702 / 720 the LDS-staged
 *   forwards of tools/experiments/msda_tile/ (DESIGN.md section 3.1c/d).
 *   100..199 and 400..499 are TIMING ABLATIONS that skip one of the two backward kernels: wrong results by
 *   construction.
 */
void vnx_set_kernel_variant(int variant);
int vnx_get_kernel_variant(void);

/*
 * Kernel-span stamps (process-wide state, hence development build only).  While a buffer is armed every launch of a tuned
 * MSDA kernel takes a region of 2 x gridDim 64-bit words and each workgroup leaves {its start, its last wave's end} there in
 * constant-rate wall-clock ticks (vnx_debug_wall_clock_khz, vnext_hip_debug.h).  buf: n_words zero-filled 64-bit words;
 * nullptr disarms.  vnx_debug_stamp_regions -> number of regions handed out since arming; per region the kernel kind
 * (1 forward, 2 grad_loc / grad_attn, 3 grad_value, 4 the paired backward kernel), word offset, workgroups.
 */
void vnx_debug_arm_stamps(void* buf, long long n_words);
int vnx_debug_stamp_regions(int* kinds, long long* offsets, long long* blocks, int n);

/* phase stamps of the record-fed grad_value kernel (variants 408 / 412) and of the tiled forward (701 / 702): copies
 * n 64-bit words of the kernel's fixed device array to `host`; returns a hipError_t as int */
int vnx_debug_read_rec_stamps(unsigned long long* host, int n);
int vnx_debug_read_tile_stamps(unsigned long long* host, int n);
/* the same for the self-decoding grad_value kernel (msda_d32_gvdirect.hip), 8 stamps per workgroup, of a library built with
 * -DVNX_GVD_STAMPS (zeros otherwise) */
int vnx_debug_read_gvd_stamps(unsigned long long* host, int n);

#ifdef __cplusplus
}
#endif
#endif /* VNEXT_HIP_DEV_H_ */
