// tools/lds_atomic_bench.hip -- LDS fp32 atomic-add throughput on gfx950 (development tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MODE 0: ds_add_f32 (unsafeAtomicAdd), row per half-wave   1: plain += (read, add, write)
// MODE 2: ds_add_u32                                         3: ds_add_f32, whole wave one row (2 lanes/addr)
// MODE 4: ds_add_rtn_f32 (value used)
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters, int nrows, uint32_t seed) {
  extern __shared__ float slab[];
  const int tid = threadIdx.x;
  for (int i = tid; i < nrows * 32; i += 512) slab[i] = 0.f;
  __syncthreads();
  const int hw = tid >> 5, c = tid & 31;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const uint32_t row = hash(seed + blockIdx.x * 7919u + (MODE == 3 ? (tid >> 6) : hw) * 131u + it) % nrows;
    float* p = slab + row * 32 + c;
    if (MODE == 0 || MODE == 3) unsafeAtomicAdd(p, 1.0f);
    else if (MODE == 1) *p += 1.0f;
    else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned*>(p), 1u);
    else acc += atomicAdd(p, 1.0f);
  }
  __syncthreads();
  float s = acc;
  for (int i = tid; i < nrows * 32; i += 512) s += slab[i];
  if (s == 123.456f) out[0] = s;
}
template <int MODE>
int run(const char* name, float* out, int nrows, int blocks) {
  const int iters = 4096;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = size_t(nrows) * 128;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), lds, 0, out, iters, nrows, 1u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), lds, 0, out, iters, nrows, 2u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_instrs_per_cu = 8.0 * iters * (blocks / 256.0);
  printf("%-34s rows %4d blocks %4d: %8.1f us  -> %6.1f clk @2.1GHz per wave-instr per CU, %6.2f lanes/clk/CU\n", name, nrows, blocks,
         ms * 1e3, ms * 1e-3 * 2.1e9 / wave_instrs_per_cu, wave_instrs_per_cu * 64 / (ms * 1e-3 * 2.1e9));
  return 0;
}
int main() {
  float* out; CK(hipMalloc(&out, 4));
  for (int blocks : {256, 512}) for (int nrows : {384, 60, 8}) {
    run<0>("ds_add_f32 (row per half-wave)", out, nrows, blocks);
    run<3>("ds_add_f32 (wave on one row)", out, nrows, blocks);
    run<4>("ds_add_rtn_f32", out, nrows, blocks);
    run<2>("ds_add_u32", out, nrows, blocks);
    run<1>("plain read-add-write", out, nrows, blocks);
  }
  return 0;
}
