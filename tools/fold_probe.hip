// What the lane-exchange primitives of the mask head's transposed wave reduction do, lane by lane (development tool:
// v_permlane32_swap / v_permlane16_swap, DPP row_shr / row_shl / row_half_mirror / quad_perm with bank masks -- a bank is a
// quad of a row; a DPP asm statement right after the instruction that wrote its source reads unshuffled values, rows
// "shr8(a)" and "qp2301 bank3 (a)" below: hipcc adds no wait states for inline asm).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float fold32(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__global__ void k(float* o) {
  const int l = threadIdx.x;
  float a = float(l), b = float(1000 + l);
  auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  o[l] = __builtin_bit_cast(float, r32[0]); o[64 + l] = __builtin_bit_cast(float, r32[1]);
  auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  o[128 + l] = __builtin_bit_cast(float, r16[0]); o[192 + l] = __builtin_bit_cast(float, r16[1]);
  float t = a;
  asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(t));
  o[256 + l] = t;
  asm volatile("v_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(b));
  o[320 + l] = t;
  float p;
  asm volatile("v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=&v"(p) : "v"(a));
  o[384 + l] = p;
  float q = a;
  asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0x3" : "+v"(q));
  o[448 + l] = q;
  asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xc" : "+v"(q) : "v"(b));
  o[512 + l] = q;
  float u = a;
  asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0x5" : "+v"(u));
  asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xa" : "+v"(u) : "v"(b));
  o[576 + l] = u;
  o[640 + l] = fold32(a, b);
}
int main() {
  float* d; (void)hipMalloc(&d, 704 * 4); (void)hipMemset(d, 0, 704 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[704]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"pl32.dst", "pl32.src", "pl16.dst", "pl16.src", "shr8(a)", "+shl8(b)", "halfmirror(a)", "qp2301 bank3 (a)", "+bankC (b)", "qp1032 5/a", "fold32"};
  for (int r = 0; r < 11; ++r) {
    printf("%-18s", names[r]);
    for (int l = 0; l < 64; ++l) printf(" %g", h[64 * r + l]);
    printf("\n");
  }
  return 0;
}
