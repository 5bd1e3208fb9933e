// msda_gv_common.h -- helpers shared by the owner-computes grad_value kernels
// (msda_d32_gvrec.hip: fed by sample records; msda_d32_gvtiles.hip: fed by query-tile summaries).
#pragma once

#include "vnx_common.h"

namespace vnx {
namespace rec {

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));

template <typename TV> __device__ __forceinline__ float4_t load4(const TV* p);
template <> __device__ __forceinline__ float4_t load4<float>(const float* p) {
  return *reinterpret_cast<const float4_t*>(p);
}
template <> __device__ __forceinline__ float4_t load4<bf16_t>(const bf16_t* p) {
  const uint2_t r = *reinterpret_cast<const uint2_t*>(p);
  float4_t v;
  v.x = __uint_as_float(r.x << 16); v.y = __uint_as_float(r.x & 0xffff0000u);
  v.z = __uint_as_float(r.y << 16); v.w = __uint_as_float(r.y & 0xffff0000u);
  return v;
}
template <> __device__ __forceinline__ float4_t load4<f16_t>(const f16_t* p) {
  const uint2_t r = *reinterpret_cast<const uint2_t*>(p);
  float4_t v;
  v.x = float(__builtin_bit_cast(_Float16, uint16_t(r.x & 0xffffu)));
  v.y = float(__builtin_bit_cast(_Float16, uint16_t(r.x >> 16)));
  v.z = float(__builtin_bit_cast(_Float16, uint16_t(r.y & 0xffffu)));
  v.w = float(__builtin_bit_cast(_Float16, uint16_t(r.y >> 16)));
  return v;
}
// four channels of a row through a buffer descriptor (byte offset; out of range reads zeros)
template <typename TV> __device__ __forceinline__ float4_t load4_buf(__amdgpu_buffer_rsrc_t src, uint32_t byte_off);
template <> __device__ __forceinline__ float4_t load4_buf<float>(__amdgpu_buffer_rsrc_t src, uint32_t byte_off) {
  return __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(src, int(byte_off), 0, 0));
}
template <> __device__ __forceinline__ float4_t load4_buf<bf16_t>(__amdgpu_buffer_rsrc_t src, uint32_t byte_off) {
  const uint2_t r = __builtin_bit_cast(uint2_t, __builtin_amdgcn_raw_buffer_load_b64(src, int(byte_off), 0, 0));
  float4_t v;
  v.x = __uint_as_float(r.x << 16); v.y = __uint_as_float(r.x & 0xffff0000u);
  v.z = __uint_as_float(r.y << 16); v.w = __uint_as_float(r.y & 0xffff0000u);
  return v;
}
template <> __device__ __forceinline__ float4_t load4_buf<f16_t>(__amdgpu_buffer_rsrc_t src, uint32_t byte_off) {
  const uint2_t r = __builtin_bit_cast(uint2_t, __builtin_amdgcn_raw_buffer_load_b64(src, int(byte_off), 0, 0));
  float4_t v;
  v.x = float(__builtin_bit_cast(_Float16, uint16_t(r.x & 0xffffu)));
  v.y = float(__builtin_bit_cast(_Float16, uint16_t(r.x >> 16)));
  v.z = float(__builtin_bit_cast(_Float16, uint16_t(r.y & 0xffffu)));
  v.w = float(__builtin_bit_cast(_Float16, uint16_t(r.y >> 16)));
  return v;
}
template <typename TV> __device__ __forceinline__ void store4(TV* p, float4_t v);
template <> __device__ __forceinline__ void store4<float>(float* p, float4_t v) {
  // grad_value is written once and read by another kernel much later: `nt` (decoder-360p backward 33.8 -> 31.4 us)
  // (same WRITE_SIZE / FETCH_SIZE per launch as plain stores; rocprofv3 18.9 vs 20.0 us for this kernel)
#if defined(VNX_GV_STORE_POLICY) && VNX_GV_STORE_POLICY == 1      // A/B (round 6): cache policy of the row stores, for the step (fwd behind bwd)
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
#elif defined(VNX_GV_STORE_POLICY) && VNX_GV_STORE_POLICY == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
#elif defined(VNX_GV_STORE_POLICY) && VNX_GV_STORE_POLICY == 3
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
#elif defined(VNX_GV_STORE_POLICY) && VNX_GV_STORE_POLICY == 4
  *reinterpret_cast<float4_t*>(p) = v;
#else
  __builtin_nontemporal_store(v, reinterpret_cast<float4_t*>(p));
#endif
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float4_t v) {
  uint2_t r;
  r.x = f32x2_to_bf16x2(v.x, v.y);
  r.y = f32x2_to_bf16x2(v.z, v.w);
#if defined(VNX_GV_STORE16_PLAIN) || (defined(VNX_GV_PAIR16) && !defined(VNX_GV_PAIR16_NT))      // A/B builds (tools/r3_call23.sh, r3_call33.sh)
  *reinterpret_cast<uint2_t*>(p) = r;
#else
  __builtin_nontemporal_store(r, reinterpret_cast<uint2_t*>(p));
#endif
}
template <> __device__ __forceinline__ void store4<f16_t>(f16_t* p, float4_t v) {
  uint2_t r;
  r.x = uint32_t(__builtin_bit_cast(uint16_t, _Float16(v.x))) | (uint32_t(__builtin_bit_cast(uint16_t, _Float16(v.y))) << 16);
  r.y = uint32_t(__builtin_bit_cast(uint16_t, _Float16(v.z))) | (uint32_t(__builtin_bit_cast(uint16_t, _Float16(v.w))) << 16);
  __builtin_nontemporal_store(r, reinterpret_cast<uint2_t*>(p));
}

// Workgroup -> (unit x batch index `rest`, head).  fp32: blockIdx % M picks the head (= the XCD when M == 8), rotated
// by the batch element (msda_d32.hip).  16-bit values (VNX_GV_PAIR16): a row is 64 B, HALF a cache line, the other half
// is the next head's row -- written by another XCD each half is a partial-line write (grad_value kernel at decoder-720p:
// 36.5 us in bf16 against 32.1 in fp32 for half the bytes, no extra traffic by PMC).  So a PAIR of heads shares an XCD
// (XCDs x and x + M/2 split the units of the pair between them), neighbouring workgroups of an XCD take the two heads of
// one unit, and the rows are written with plain stores: the two halves meet in that XCD's L2 and leave as one line.
template <bool PAIR>
__device__ __forceinline__ void gv_decode_block(uint32_t block, int M, int B, int& rest, int& b, int& m) {
  const int x = int(block % uint32_t(M)), j = int(block / uint32_t(M));
  if constexpr (PAIR) {
    const int half = M >> 1;
    const int xh = x % half, side = x / half;
    rest = (j >> 1) * 2 + side;
    b = rest % B;
    m = 2 * ((xh + b) % half) + (j & 1);
  } else {
    rest = j;
    b = rest % B;
    m = (x + b) % M;
  }
}
// The same map with the BATCH ELEMENT major (round 6; the tile-fed kernel): workgroup -> (batch element, unit), all
// `units_pb` units of a batch element before the next one's.  The workgroups resident on an XCD at one time then belong to
// one or two batch elements, whose grad_out rows, decoded samples and tile words of the XCD's head class (6 MB per batch
// element at 720p) are re-read from that XCD's 4-MB L2 by neighbouring units instead of from memory: with the batch element
// minor every resident group of units spans all B of them (the change that cut the slab forward's reads 929 -> 379 MB).
// b may come out as B for the padding workgroup of an odd grid: the caller leaves.
template <bool PAIR>
__device__ __forceinline__ void gv_decode_block_bm(uint32_t block, int M, int units_pb, int& unit, int& b, int& m) {
  const int x = int(block % uint32_t(M)), j = int(block / uint32_t(M));
  if constexpr (PAIR) {
    const int half = M >> 1;
    const int xh = x % half, side = x / half;
    const int rest = (j >> 1) * 2 + side;
    b = rest / units_pb;
    unit = rest - b * units_pb;
    m = 2 * ((xh + b) % half) + (j & 1);
  } else {
    b = j / units_pb;
    unit = j - b * units_pb;
    m = (x + b) % M;
  }
}
#ifdef VNX_GV_PAIR16
constexpr bool kGvPair16 = true;
#else
constexpr bool kGvPair16 = false;
#endif

// The value again, but opaque to the optimiser: what is derived from the result is recomputed where it is used instead of
// being hoisted out of the loops and then SPILLED (the selection kernel kept `tid | 0x200`, `tid >> 2`, `tid * 4`, ... live
// across its loops and spilled ten of them: 10 dwords x 64 lanes x 6 080 waves = 15.5 MB of scratch written back to
// memory per launch -- the whole excess of WRITE_SIZE over the 26.1 MB of grad_value rows, profiles/r02).
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  int x = int(v);
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return uint32_t(x);
}

// A barrier that orders LDS only.  __syncthreads() is a workgroup-scope fence over ALL memory: on gfx9 (one counter for
// vector loads and stores) it waits for every outstanding global load -- here the 38 KB of grad_out rows a workgroup has in
// flight while it decodes and sorts.  The fences below name the local address space, so only lgkmcnt is waited for.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

constexpr int kWaves = 8;                   // 512 threads, one sample per thread per chunk
constexpr int kThreads = 64 * kWaves;
constexpr int kGroups = kThreads / 8;       // 8-lane groups
constexpr int kRowsMax = kGvRowsMax;        // 320 rows: 40 KiB as an LDS slab
constexpr int kQcMax = 128;                 // queries per chunk (16 KiB of grad_out rows in LDS)
constexpr int kLevelsMax = 64;

#ifndef VNX_SEL_UNITS_PER_CU
#define VNX_SEL_UNITS_PER_CU 3
#endif

}  // namespace rec
}  // namespace vnx
