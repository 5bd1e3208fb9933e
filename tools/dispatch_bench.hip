// tools/dispatch_bench.hip -- how fast does the GPU start workgroups?  (development tool)
// An (almost) empty kernel over G workgroups of T threads with L bytes of dynamic LDS: time per launch by HIP events
// over a graph of back-to-back launches.  The grad_value kernel of the headline backward is 760 workgroups x 512 threads
// x 46 KB: its duration = the time to start them all + one light unit.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int REGS>
__global__ void __launch_bounds__(1024) k(float* out, int spin) {
  extern __shared__ float lds[];
  float acc[REGS];
#pragma unroll
  for (int i = 0; i < REGS; ++i) acc[i] = float(threadIdx.x + i);
  if (threadIdx.x == 0) lds[0] = 1.f;
  __syncthreads();
  for (int s = 0; s < spin; ++s) {
#pragma unroll
    for (int i = 0; i < REGS; ++i) acc[i] = acc[i] * 1.0001f + lds[0];
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < REGS; ++i) t += acc[i];
  if (t == 123.456f) out[blockIdx.x] = t;
}

template <int REGS>
static int run(const char* name, int G, int T, size_t L, int spin, float* out) {
  hipStream_t st; CK(hipStreamCreate(&st));
  CK(hipFuncSetAttribute((const void*)k<REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const int inner = 32;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < inner; ++i) hipLaunchKernelGGL(k<REGS>, dim3(G), dim3(T), L, st, out, spin);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int r = 0; r < 15; ++r) {
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms * 1e3f / inner);
  }
  std::sort(ts.begin(), ts.end());
  printf("%-10s G=%5d T=%4d LDS=%6zu spin=%4d : %7.2f us per launch\n", name, G, T, L, spin, ts[ts.size() / 2]);
  return 0;
}

int main() {
  float* out; CK(hipMalloc(&out, 1 << 20));
  for (int spin : {0, 200}) {
    run<8>("regs8", 760, 512, 0, spin, out);
    run<8>("regs8", 760, 512, 16384, spin, out);
    run<8>("regs8", 760, 512, 47000, spin, out);
    run<8>("regs8", 1520, 256, 23500, spin, out);
    run<8>("regs8", 3040, 128, 11750, spin, out);
    run<8>("regs8", 6080, 64, 5800, spin, out);
    run<64>("regs64", 760, 512, 47000, spin, out);
    run<64>("regs64", 760, 512, 0, spin, out);
    run<8>("regs8", 380, 512, 47000, spin, out);
    run<8>("regs8", 380, 1024, 47000, spin, out);
    run<8>("regs8", 3000, 64, 0, spin, out);
  }
  return 0;
}
