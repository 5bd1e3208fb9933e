// tools/atomic_bench.hip -- micro-benchmark of fp32 atomic-add address patterns on gfx950
// (development tool; informs the grad_value scatter layout in msda_d32.hip).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_bench.hip -o /tmp/atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

// mode 0: 2 lines / instr, 32 consecutive dwords each (row-per-half-wave)
// mode 1: 8 lines / instr, 8 lanes each at 16-B stride (float4-lane layout, component j)
// mode 2: 8 lines / instr, 8 lanes each, consecutive dwords (32 B per line per instr)
// mode 3: mode 0 with plain stores instead of atomics
// mode 4: 1 line / instr: 64 lanes -> 32 dwords, two lanes per dword
// mode 5: mode 0 with sc1 (system scope)
// mode 6: mode 0 but loads (gather) for reference
template <int MODE>
__global__ void k(float* buf, uint32_t n_lines, int iters, uint32_t seed, float* sink) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    uint32_t line, dword;
    if (MODE == 0 || MODE == 3 || MODE == 5 || MODE == 6) {
      line = hash(seed + (wave * iters + it) * 2 + (lane >> 5)) % n_lines; dword = lane & 31;
    } else if (MODE == 1) {
      line = hash(seed + (wave * iters + it / 4) * 8 + (lane >> 3)) % n_lines; dword = (lane & 7) * 4 + (it & 3);
    } else if (MODE == 2) {
      line = hash(seed + (wave * iters + it / 4) * 8 + (lane >> 3)) % n_lines; dword = (lane & 7) + 8 * (it & 3);
    } else {
      line = hash(seed + (wave * iters + it)) % n_lines; dword = lane & 31;
    }
    float* p = buf + size_t(line) * 32 + dword;
    if (MODE == 3) *p = 1.0f;
    else if (MODE == 5) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (MODE == 6) acc += *p;
    else unsafeAtomicAdd(p, 1.0f);
  }
  if (MODE == 6 && acc == 12345.f) *sink = acc;
}

template <int MODE>
int run(const char* name, float* buf, size_t bytes, float* sink, int iters) {
  const uint32_t n_lines = uint32_t(bytes / 128);
  const int blocks = 256 * 8, threads = 256;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, buf, n_lines, iters, 1u, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, buf, n_lines, iters, 77u + r, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double instrs = double(blocks) * threads / 64 * iters;
  const double lines_per = (MODE == 1 || MODE == 2) ? 8 : (MODE == 4 ? 1 : 2);
  const double dwords = instrs * 64;
  printf("%-44s region %7.1f MB: %8.1f us  %7.2f G instr/s  %7.2f G line-ops/s  %7.2f G dword/s  %6.2f TB/s(4B/dword)\n",
         name, bytes / 1e6, ms * 1e3, instrs / ms / 1e6, instrs * lines_per / ms / 1e6, dwords / ms / 1e6, dwords * 4 / ms / 1e9);
  return 0;
}

int main() {
  float* buf; float* sink;
  const size_t big = size_t(512) << 20;
  CK(hipMalloc(&buf, big)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, big));
  const size_t regions[] = {size_t(2) << 20, size_t(26) << 20, size_t(104) << 20, size_t(512) << 20};
  for (size_t r : regions) {
    run<0>("atomic 2 lines x 32 consecutive dwords", buf, r, sink, 64);
    run<4>("atomic 1 line, 2 lanes per dword", buf, r, sink, 64);
    run<1>("atomic 8 lines x 8 lanes @16B stride", buf, r, sink, 64);
    run<2>("atomic 8 lines x 8 consecutive dwords", buf, r, sink, 64);
    run<5>("atomic sc1 2 lines x 32 dwords", buf, r, sink, 64);
    run<3>("plain store 2 lines x 32 dwords", buf, r, sink, 64);
    run<6>("plain load 2 lines x 32 dwords", buf, r, sink, 64);
  }
  return 0;
}
