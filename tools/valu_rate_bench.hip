// Issue rate of common gfx950 vector instructions (development tool; table in DESIGN 3.6).  `waves` waves per SIMD each
// run a long unrolled sequence of one instruction over eight independent registers; reported: nanoseconds per
// instruction per SIMD (a 64-lane instruction through a 32-lane pipe at ~2 GHz is ~1 ns).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));

#define ONE(S, i) asm volatile(S : "+v"(a[i]), "+v"(b[i]) : "v"(x), "s"(w), "v"(x.x), "v"(x.y), "s"(ws));
#define ALL8(S) ONE(S, 0) ONE(S, 1) ONE(S, 2) ONE(S, 3) ONE(S, 4) ONE(S, 5) ONE(S, 6) ONE(S, 7)
#define BODY(S) ALL8(S) ALL8(S) ALL8(S) ALL8(S) ALL8(S) ALL8(S) ALL8(S) ALL8(S)

#define KERNEL(NAME, S)                                                                           \
  __global__ void __launch_bounds__(64) NAME(float* out, int iters, float seed) {                  \
    float2_t a[8]; float b[8];                                                                     \
    for (int i = 0; i < 8; ++i) { a[i] = float2_t{seed + i, seed - i}; b[i] = seed * i; }          \
    const float2_t x = {seed * 0.5f, seed * 0.25f};                                                \
    uint64_t w = (uint64_t(__float_as_uint(seed)) << 32) | __float_as_uint(seed * 0.75f);          \
    w = __builtin_amdgcn_readfirstlane(uint32_t(w)) |                                              \
        (uint64_t(__builtin_amdgcn_readfirstlane(uint32_t(w >> 32))) << 32);                       \
    const uint32_t ws = uint32_t(w);                                                               \
    for (int it = 0; it < iters; ++it) { BODY(S) }                                                 \
    float r = 0.f;                                                                                 \
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + b[i];                                       \
    if (r == 12345.678f) out[threadIdx.x] = r;                                                     \
  }

// %0 = a 64-bit VGPR pair (read-write), %2 = a 64-bit VGPR pair, %3 = a 64-bit SGPR pair; %1 = a VGPR (read-write), %4, %5 = VGPRs, %6 = an SGPR
KERNEL(k_fma, "v_fma_f32 %1, %4, %5, %1")
KERNEL(k_fma_s, "v_fma_f32 %1, %6, %5, %1")
KERNEL(k_pk_fma, "v_pk_fma_f32 %0, %2, %2, %0")
KERNEL(k_pk_fma_s, "v_pk_fma_f32 %0, %3, %2, %0 op_sel_hi:[0,1,1]")
KERNEL(k_pk_add, "v_pk_add_f32 %0, %2, %0")
KERNEL(k_pk_mul, "v_pk_mul_f32 %0, %2, %0")
KERNEL(k_add, "v_add_f32 %1, %4, %1")
KERNEL(k_mul, "v_mul_f32 %1, %4, %1")
KERNEL(k_max, "v_max_f32 %1, 0, %1")
KERNEL(k_max3, "v_max_f32_e64 %1, %4, %1")
KERNEL(k_maxi, "v_max_i32 %1, 0, %1")
KERNEL(k_med3, "v_med3_f32 %1, %1, %4, %5")
KERNEL(k_addu, "v_add_u32 %1, %4, %1")
KERNEL(k_and, "v_and_b32 %1, %4, %1")
KERNEL(k_lshladd, "v_lshl_add_u32 %1, %1, 2, %4")
KERNEL(k_mad24, "v_mad_i32_i24 %1, %1, %4, %5")
KERNEL(k_mullo, "v_mul_lo_u32 %1, %1, %4")
KERNEL(k_cvt, "v_cvt_f32_i32 %1, %1")
KERNEL(k_mov, "v_mov_b32 %1, %4")
KERNEL(k_mov64, "v_mov_b64 %0, %2")
KERNEL(k_cndmask, "v_cndmask_b32 %1, %1, %4, vcc")
KERNEL(k_cmp, "v_cmp_lt_i32 vcc, %1, %4")
KERNEL(k_dpp, "v_mov_b32_dpp %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_adddpp, "v_add_f32_dpp %1, %4, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_readlane, "v_readlane_b32 s20, %1, 3")
KERNEL(k_perm, "v_perm_b32 %1, %1, %4, %5")
KERNEL(k_bfe, "v_bfe_u32 %1, %1, 3, 5")
KERNEL(k_rcp, "v_rcp_f32 %1, %1")
KERNEL(k_exp, "v_exp_f32 %1, %1")

template <typename K>
double run(K kern, int waves_per_simd, int iters, float* out) {
  const int blocks = 1024 * waves_per_simd;       // 64-thread workgroups: one wave each
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 10, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return double(ms) * 1e6 / (double(iters) * 64) / waves_per_simd;     // ns per instruction per SIMD
}

int main() {
  float* out; (void)hipMalloc(&out, 4096);
#define ROW(NAME, LABEL) printf("%-44s %6.3f  %6.3f  %6.3f\n", LABEL, run(NAME, 1, 3000, out), run(NAME, 2, 3000, out), run(NAME, 4, 3000, out));
  printf("%-44s %6s  %6s  %6s   (ns per instruction per SIMD)\n", "instruction", "1 wave", "2", "4");
  ROW(k_fma, "v_fma_f32 v,v,v") ROW(k_fma_s, "v_fma_f32 s,v,v") ROW(k_pk_fma, "v_pk_fma_f32 v,v,v") ROW(k_pk_fma_s, "v_pk_fma_f32 s,v,v op_sel")
  ROW(k_pk_add, "v_pk_add_f32") ROW(k_pk_mul, "v_pk_mul_f32") ROW(k_add, "v_add_f32") ROW(k_mul, "v_mul_f32")
  ROW(k_max, "v_max_f32 0,v (e32)") ROW(k_max3, "v_max_f32 v,v (e64)") ROW(k_maxi, "v_max_i32 0,v") ROW(k_med3, "v_med3_f32")
  ROW(k_addu, "v_add_u32") ROW(k_and, "v_and_b32") ROW(k_lshladd, "v_lshl_add_u32") ROW(k_mad24, "v_mad_i32_i24") ROW(k_mullo, "v_mul_lo_u32")
  ROW(k_cvt, "v_cvt_f32_i32") ROW(k_mov, "v_mov_b32") ROW(k_mov64, "v_mov_b64") ROW(k_cndmask, "v_cndmask_b32") ROW(k_cmp, "v_cmp_lt_i32")
  ROW(k_dpp, "v_mov_b32_dpp wave_shr:1") ROW(k_adddpp, "v_add_f32_dpp row_shr:1") ROW(k_readlane, "v_readlane_b32") ROW(k_perm, "v_perm_b32")
  ROW(k_bfe, "v_bfe_u32") ROW(k_rcp, "v_rcp_f32") ROW(k_exp, "v_exp_f32")
  return 0;
}
