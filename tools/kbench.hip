// tools/kbench.hip -- stand-alone A/B timing + cross-check harness for the MSDA kernels of
// libvnext_hip.so (development tool; no torch, starts in a second on a fresh GPU box).
//
//   kbench [--shape dec360|enc360|dec720|enc720] [--B 5] [--lq N] [--dist U|M] [--op fwd|bwd|both|ffwd|fbwd]
//          [--variants 0,1,...] [--inner 24] [--reps 15] [--check] [--dma-test]
//
// Inputs are generated on the device (hash RNG): value ~ N(0,1); locations U[0,1)^2 (the reference
// test's convention, ops/test.py:34) or model-like "M" (encoder: pixel-centre reference points of the
// pyramid, deformable_transformer.py:183-190; decoder: random centres; offsets = head direction x
// (k+1) + N(0,1) pixels, ops/modules/ms_deform_attn.py:65-73).  Cold numbers rotate through > 320 MiB
// of distinct input sets; warm numbers reuse one set.  Every variant's outputs are compared with the
// generic kernels (variant 1) -- a cross-check between two implementations of this library; parity
// against the oracle lives in tests/.
#include <hip/hip_runtime.h>
#include <chrono>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../include/vnext_hip.h"
#include "../include/vnext_hip_dev.h"      // links against the development library (libvnext_hip_dev.so): forced variants

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)
#define VK(x)                                                                                  \
  do {                                                                                         \
    int s_ = (x);                                                                              \
    if (s_ != VNX_OK) {                                                                        \
      fprintf(stderr, "%s:%d %s -> %s: %s\n", __FILE__, __LINE__, #x, vnx_status_string(s_), vnx_last_error()); \
      exit(3);                                                                                 \
    }                                                                                          \
  } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float u01(uint32_t seed, uint64_t i) {
  return (hash32(seed ^ hash32(uint32_t(i) * 2654435761u + uint32_t(i >> 32))) >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float gauss(uint32_t seed, uint64_t i) {
  const float a = fmaxf(u01(seed, 2 * i), 1e-7f), b = u01(seed ^ 0x9e3779b9u, 2 * i + 1);
  return sqrtf(-2.f * logf(a)) * cosf(6.2831853f * b);
}
__global__ void fill_gauss(float* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] = gauss(seed, i);
}
// loc [B,Lq,M,L,P,2], attn [B,Lq,M,L,P]; shapes on host side passed by value (L <= 4 here)
struct Pyr { int H[4], W[4], start[4]; };
__global__ void fill_samples(float* loc, float* attn, int B, int Lq, int M, int L, int P, Pyr py, int S, int dist, uint32_t seed) {
  const size_t rows = size_t(B) * Lq * M;
  for (size_t r = blockIdx.x * size_t(blockDim.x) + threadIdx.x; r < rows; r += size_t(gridDim.x) * blockDim.x) {
    const int m = int(r % M);
    const size_t bq = r / M;
    const int q = int(bq % Lq);
    float rx, ry;
    if (dist == 1 && Lq == S) {   // the query is a pixel of the pyramid
      int l = 0;
      while (l + 1 < L && q >= py.start[l + 1]) ++l;
      const int i = q - py.start[l];
      rx = (i % py.W[l] + 0.5f) / py.W[l];
      ry = (i / py.W[l] + 0.5f) / py.H[l];
    } else {
      rx = u01(seed ^ 0x1234567u, 2 * bq); ry = u01(seed ^ 0x1234567u, 2 * bq + 1);
    }
    const float th = m * (6.2831853f / M);
    float dx = cosf(th), dy = sinf(th);
    const float mx = fmaxf(fabsf(dx), fabsf(dy));
    dx /= mx; dy /= mx;
    float lg[64], mxl = -1e30f, sum = 0.f;
    for (int s = 0; s < L * P; ++s) { lg[s] = gauss(seed ^ 0x777u, r * (L * P) + s); mxl = fmaxf(mxl, lg[s]); }
    for (int s = 0; s < L * P; ++s) { lg[s] = expf(lg[s] - mxl); sum += lg[s]; }
    for (int l = 0; l < L; ++l)
      for (int k = 0; k < P; ++k) {
        const size_t s = r * (L * P) + l * P + k;
        float x, y;
        if (dist == 0) { x = u01(seed, 2 * s); y = u01(seed, 2 * s + 1); }
        else {
          x = rx + (dx * (k + 1) + gauss(seed ^ 0x51u, 2 * s)) / py.W[l];
          y = ry + (dy * (k + 1) + gauss(seed ^ 0x51u, 2 * s + 1)) / py.H[l];
        }
        loc[2 * s] = x; loc[2 * s + 1] = y;
        attn[s] = lg[l * P + k] / sum;
      }
  }
}
// fused-prologue inputs that reproduce (loc, attn): reference points 0.5, offsets = (loc - 0.5) * (W, H), logits = log(attn)
__global__ void to_fused(const float* loc, const float* attn, float* off, float* logit, size_t n_s, int L, int P, Pyr py) {
  for (size_t s = blockIdx.x * size_t(blockDim.x) + threadIdx.x; s < n_s; s += size_t(gridDim.x) * blockDim.x) {
    const int l = int((s / P) % L);
    off[2 * s] = (loc[2 * s] - 0.5f) * py.W[l];
    off[2 * s + 1] = (loc[2 * s + 1] - 0.5f) * py.H[l];
    logit[s] = logf(fmaxf(attn[s], 1e-30f));
  }
}
// --dtype bf16: value / grad_out / out / grad_value in bf16 (locations and weights stay fp32: the autocast case, BASELINE config 3)
__global__ void f32_to_bf16(const float* src, uint16_t* dst, size_t n) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    uint32_t u = __float_as_uint(src[i]);
    u += 0x7fffu + ((u >> 16) & 1u);
    dst[i] = uint16_t(u >> 16);
  }
}
__global__ void bf16_to_f32(const uint16_t* src, float* dst, size_t n) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
    dst[i] = __uint_as_float(uint32_t(src[i]) << 16);
}
__global__ void fill_const(float* p, size_t n, float v) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] = v;
}
__global__ void max_abs_diff(const float* a, const float* b, size_t n, float* out /* [2]: max|a-b|, max|b| */) {
  float d = 0.f, m = 0.f;
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    const float x = a[i], y = b[i];
    const float e = fabsf(x - y);
    d = fmaxf(d, (e == e) ? e : 1e30f);
    m = fmaxf(m, fabsf(y));
  }
  atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(d));
  atomicMax(reinterpret_cast<unsigned*>(out) + 1, __float_as_uint(m));
}

// ---- LDS-DMA probe: what does `buffer_load_dwordx4 ... lds` leave in LDS for out-of-range lanes? -----
typedef float float4_t __attribute__((ext_vector_type(4)));
__global__ void dma_probe(const float* __restrict__ src, float* __restrict__ dst, int n_floats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = -7.f;   // poison
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n_floats * 4, 0x00020000);
  const int lane = threadIdx.x & 63;
  // even 8-lane sets read a valid row, odd sets an out-of-range offset
  const unsigned voff = ((lane >> 3) & 1) ? 0x80000000u : unsigned(((lane >> 3) * 5) * 128 + (lane & 7) * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float4_t v = *reinterpret_cast<float4_t*>(smem + threadIdx.x * 16);
  *reinterpret_cast<float4_t*>(dst + threadIdx.x * 4) = v;
}
// ---- address-pattern probe: 128-B lines at a fixed offset inside every 1-KiB record (one head's rows of a
// [B, S, 8, 32] fp32 tensor), read once each in a scattered order, per offset -----------------------------------
__global__ void stride_probe(const float4_t* __restrict__ base, size_t n_rec, int sub, float* sink, uint32_t seed) {
  const size_t set = (blockIdx.x * size_t(blockDim.x) + threadIdx.x) >> 3;      // one 8-lane set per record
  const int ch = threadIdx.x & 7;
  float4_t acc = {0.f, 0.f, 0.f, 0.f};
  const size_t sets = (size_t(gridDim.x) * blockDim.x) >> 3;
  for (size_t i = set; i < n_rec; i += sets) {
    const size_t rec = (i * 2654435761ull + seed) % n_rec;                        // scattered, every record once
    acc += base[rec * 64 + sub * 8 + ch];
  }
  if (acc.x == 123.456f) sink[0] = acc.y;
}
// ---- does `buffer_load ... lds` reach LDS addresses above 64 KiB (M0 wider than 16 bits)? ------------------------
__global__ void dma_high_probe(const float* __restrict__ src, float* __restrict__ dst, int n_floats, int hi_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lo_off = hi_off & 0xffff;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    reinterpret_cast<float*>(smem + lo_off)[i] = -7.f;
    reinterpret_cast<float*>(smem + hi_off)[i] = -9.f;
  }
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n_floats * 4, 0x00020000);
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + hi_off), 16, unsigned(lane * 16), 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  dst[threadIdx.x * 4 + 0] = reinterpret_cast<float*>(smem + hi_off)[threadIdx.x * 4];
  dst[threadIdx.x * 4 + 1] = reinterpret_cast<float*>(smem + lo_off)[threadIdx.x * 4];
}
static void run_dma_high_probe() {
  const int n = 64 * 4;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = float(i + 1);
  float *src, *dst;
  CK(hipMalloc(&src, n * 4)); CK(hipMalloc(&dst, 256 * 4));
  CK(hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_high_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  for (int hi_off : {0x11100, 0x20000}) {
    hipLaunchKernelGGL(dma_high_probe, dim3(1), dim3(64), 144 * 1024, 0, src, dst, n, hi_off);
    CK(hipDeviceSynchronize());
    std::vector<float> o(256);
    CK(hipMemcpy(o.data(), dst, 256 * 4, hipMemcpyDeviceToHost));
    int hi_ok = 1, lo_untouched = 1;
    for (int t = 0; t < 64; ++t) { hi_ok &= (o[t * 4] == h[t * 4]); lo_untouched &= (o[t * 4 + 1] == -7.f); }
    printf("[dma-high-probe] LDS offset 0x%x: data landed there: %d; the alias 64 KiB below untouched: %d (got %g / %g)\n", hi_off, hi_ok,
           lo_untouched, o[4], o[5]);
  }
  CK(hipFree(src)); CK(hipFree(dst));
}
static void run_stride_probe() {
  const size_t n_rec = size_t(384) << 10;       // 384 Ki records x 1 KiB = 384 MiB (> Infinity Cache)
  float4_t* buf; float* sink;
  CK(hipMalloc(&buf, n_rec * 1024)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, n_rec * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
    for (int sub = 0; sub < 8; ++sub) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(stride_probe, dim3(4096), dim3(256), 0, 0, buf, n_rec, sub, sink, 12345u + rep);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("[stride-probe] offset %4d B of every 1 KiB: %7.1f us  %5.2f TB/s (128-B lines)\n", sub * 128, ms * 1e3,
                      n_rec * 128.0 / (ms * 1e-3) / 1e12);
    }
  CK(hipFree(buf)); CK(hipFree(sink));
}
static void run_dma_probe() {
  const int n = 64 * 32;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = float(i + 1);
  float *src, *dst;
  CK(hipMalloc(&src, n * 4)); CK(hipMalloc(&dst, 256 * 4));
  CK(hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(dma_probe, dim3(1), dim3(64), 1024, 0, src, dst, n);
  CK(hipDeviceSynchronize());
  std::vector<float> o(256);
  CK(hipMemcpy(o.data(), dst, 256 * 4, hipMemcpyDeviceToHost));
  int ok_valid = 1, oob_zero = 1, oob_poison = 1;
  for (int lane = 0; lane < 64; ++lane)
    for (int j = 0; j < 4; ++j) {
      const float got = o[lane * 4 + j];
      if ((lane >> 3) & 1) { oob_zero &= (got == 0.f); oob_poison &= (got == -7.f); }
      else ok_valid &= (got == h[((lane >> 3) * 5) * 32 + (lane & 7) * 4 + j]);
    }
  printf("[dma-probe] valid lanes correct: %d; out-of-range lanes leave zeros: %d, leave LDS untouched: %d (sample %g)\n",
         ok_valid, oob_zero, oob_poison, o[8 * 4]);
  CK(hipFree(src)); CK(hipFree(dst));
}

extern "C" void vnx_debug_arm_stamps(void* buf, long long n_words);
extern "C" int vnx_debug_stamp_regions(int* kinds, long long* offsets, long long* blocks, int n);
struct Set {
  float *value, *loc, *attn, *go, *out, *gv, *gl, *ga, *off, *logit;
  void* ws;
  uint16_t *value16, *go16, *out16, *gv16;     // --dtype bf16
};

int main(int argc, char** argv) {
  std::string shape = "dec360", dist = "U", op = "fwd", variants = "0", dtype = "f32";
  int B = 5, lq = 0, inner = 24, reps = 15, voff = 0;   // voff: floats added to every `value` base (alignment experiments)
  double warm_s = 0.06;
  bool check = false, cold_only = false, dma = false, stamps = false, timeline = false, hbm = false, eager = false, gvd_stamps = false;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() { return std::string(i + 1 < argc ? argv[++i] : ""); };
    if (a == "--shape") shape = next();
    else if (a == "--dist") dist = next();
    else if (a == "--op") op = next();
    else if (a == "--variants") variants = next();
    else if (a == "--dtype") dtype = next();
    else if (a == "--B") B = atoi(next().c_str());
    else if (a == "--lq") lq = atoi(next().c_str());
    else if (a == "--inner") inner = atoi(next().c_str());
    else if (a == "--reps") reps = atoi(next().c_str());
    else if (a == "--voff") voff = atoi(next().c_str());
    else if (a == "--check") check = true;
    else if (a == "--cold-only") cold_only = true;
    else if (a == "--warmup-s") warm_s = atof(next().c_str());      // for rocprofv3 runs: every traced launch reads cold inputs
    else if (a == "--dma-test") dma = true;
    else if (a == "--stamps") stamps = true;
    else if (a == "--timeline") timeline = true;
    else if (a == "--hbm-probe") hbm = true;
    else if (a == "--eager") eager = true;        // also time plain (uncaptured) launches, cold inputs
    else if (a == "--gvd-stamps") gvd_stamps = true;   // phase stamps of the self-decoding grad_value kernel (library built with -DVNX_GVD_STAMPS)
    else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 1; }
  }
  if (dma) { run_dma_probe(); run_dma_high_probe(); }
  if (hbm) { run_stride_probe(); return 0; }
  const bool p720 = shape.size() >= 3 && shape.substr(shape.size() - 3) == "720";
  const int HW360[4][2] = {{48, 80}, {24, 40}, {12, 20}, {6, 10}}, HW720[4][2] = {{92, 160}, {46, 80}, {23, 40}, {12, 20}};
  Pyr py;
  int64_t hshapes[8], hlsi[4];
  int S = 0;
  for (int l = 0; l < 4; ++l) {
    py.H[l] = p720 ? HW720[l][0] : HW360[l][0];
    py.W[l] = p720 ? HW720[l][1] : HW360[l][1];
    py.start[l] = S; hlsi[l] = S;
    hshapes[2 * l] = py.H[l]; hshapes[2 * l + 1] = py.W[l];
    S += py.H[l] * py.W[l];
  }
  const int M = 8, D = 32, L = 4, P = 4;
  const int Lq = lq > 0 ? lq : (shape.substr(0, 3) == "dec" ? 300 : S);
  const int idist = dist == "U" ? 0 : 1;
  const bool b16 = dtype == "bf16";
  const int VDT = b16 ? VNX_BF16 : VNX_F32;
  const double ve = b16 ? 2.0 : 4.0;
  const size_t n_value = size_t(B) * S * M * D, n_s = size_t(B) * Lq * M * L * P, n_out = size_t(B) * Lq * M * D;
  const double bytes_fwd = ve * (n_value + n_out) + 4.0 * 3 * n_s, bytes_bwd = ve * (2 * n_value + n_out) + 4.0 * 6 * n_s;
  const size_t set_bytes = 4 * (n_value + 3 * n_s + 2 * n_out);
  int nsets = int(std::max<size_t>(2, (size_t(320) << 20) / set_bytes + 1));
  nsets = std::min(nsets, inner);
  int64_t *dshapes, *dlsi;
  CK(hipMalloc(&dshapes, sizeof hshapes)); CK(hipMalloc(&dlsi, sizeof hlsi));
  CK(hipMemcpy(dshapes, hshapes, sizeof hshapes, hipMemcpyHostToDevice));
  CK(hipMemcpy(dlsi, hlsi, sizeof hlsi, hipMemcpyHostToDevice));
  size_t ws_bytes = 0;     // the largest workspace any of the variants asks for (records vs tile words)
  {
    size_t pos0 = 0;
    while (pos0 < variants.size()) {
      size_t c = variants.find(',', pos0);
      if (c == std::string::npos) c = variants.size();
      vnx_set_kernel_variant(atoi(variants.substr(pos0, c - pos0).c_str()));
      ws_bytes = std::max(ws_bytes, vnx_msda_backward_workspace_bytes(VDT, VNX_F32, B, S, M, D, L, Lq, P, VNX_MSDA_LEVELS_PACKED));
      pos0 = c + 1;
    }
    for (int v : {0, 1, 412}) {
      vnx_set_kernel_variant(v);
      ws_bytes = std::max(ws_bytes, vnx_msda_backward_workspace_bytes(VDT, VNX_F32, B, S, M, D, L, Lq, P, VNX_MSDA_LEVELS_PACKED));
    }
    vnx_set_kernel_variant(0);
    ws_bytes = std::max(ws_bytes, vnx_msda_fused_backward_workspace_bytes(VDT, B, S, M, L, Lq, P));
  }
  std::vector<Set> sets(nsets);
  for (int i = 0; i < nsets; ++i) {
    Set& s = sets[i];
    CK(hipMalloc(&s.value, n_value * 4 + 4096)); s.value += voff; CK(hipMalloc(&s.loc, n_s * 8)); CK(hipMalloc(&s.attn, n_s * 4));
    CK(hipMalloc(&s.go, n_out * 4)); CK(hipMalloc(&s.out, n_out * 4)); CK(hipMalloc(&s.gv, n_value * 4));
    CK(hipMalloc(&s.gl, n_s * 8)); CK(hipMalloc(&s.ga, n_s * 4)); CK(hipMalloc(&s.ws, std::max<size_t>(ws_bytes, 256)));
    hipLaunchKernelGGL(fill_gauss, dim3(2048), dim3(256), 0, 0, s.value, n_value, 17u + i);
    hipLaunchKernelGGL(fill_gauss, dim3(2048), dim3(256), 0, 0, s.go, n_out, 917u + i);
    hipLaunchKernelGGL(fill_samples, dim3(2048), dim3(128), 0, 0, s.loc, s.attn, B, Lq, M, L, P, py, S, idist, 31u + 7u * i);
    s.off = s.logit = nullptr;
    s.value16 = s.go16 = s.out16 = s.gv16 = nullptr;
    if (b16) {
      CK(hipMalloc(&s.value16, n_value * 2 + 4096)); CK(hipMalloc(&s.go16, n_out * 2)); CK(hipMalloc(&s.out16, n_out * 2));
      CK(hipMalloc(&s.gv16, n_value * 2));
      hipLaunchKernelGGL(f32_to_bf16, dim3(2048), dim3(256), 0, 0, s.value, s.value16, n_value);
      hipLaunchKernelGGL(f32_to_bf16, dim3(2048), dim3(256), 0, 0, s.go, s.go16, n_out);
    }
    if (op == "ffwd" || op == "fbwd") {
      CK(hipMalloc(&s.off, n_s * 8)); CK(hipMalloc(&s.logit, n_s * 4));
      hipLaunchKernelGGL(to_fused, dim3(2048), dim3(256), 0, 0, s.loc, s.attn, s.off, s.logit, n_s, L, P, py);
    }
  }
  float* refpts = nullptr;
  if (op == "ffwd" || op == "fbwd") {
    CK(hipMalloc(&refpts, size_t(B) * Lq * L * 2 * 4));
    hipLaunchKernelGGL(fill_const, dim3(1024), dim3(256), 0, 0, refpts, size_t(B) * Lq * L * 2, 0.5f);
  }
  CK(hipDeviceSynchronize());
  printf("shape=%s dtype=%s dist=%s B=%d Lq=%d S=%d points=%zu  alg MB fwd=%.1f bwd=%.1f  sets=%d inner=%d\n", shape.c_str(), dtype.c_str(), dist.c_str(), B,
         Lq, S, n_s, bytes_fwd / 1e6, bytes_bwd / 1e6, nsets, inner);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  auto fwd = [&](Set& s) {
    if (op == "ffwd" && vnx_get_kernel_variant() != 1) {
      if (b16) VK(vnx_msda_fused_forward(VNX_BF16, VNX_F32, s.value16, dshapes, dlsi, s.off, s.logit, refpts, s.out16, B, S, M, D, L, Lq, P, 2, 1, st));
      else VK(vnx_msda_fused_forward(VNX_F32, VNX_F32, s.value, dshapes, dlsi, s.off, s.logit, refpts, s.out, B, S, M, D, L, Lq, P, 2, 1, st));
      return;
    }
    if (b16) { VK(vnx_msda_forward(VNX_BF16, VNX_F32, s.value16, dshapes, dlsi, s.loc, s.attn, s.out16, B, S, M, D, L, Lq, P, st)); return; }
    VK(vnx_msda_forward(VNX_F32, VNX_F32, s.value, dshapes, dlsi, s.loc, s.attn, s.out, B, S, M, D, L, Lq, P, st));
  };
  auto bwd = [&](Set& s) {
    if (op == "fbwd" && vnx_get_kernel_variant() != 1) {     // fused prologue: gradients of the Linear outputs land in gl / ga
      VK(vnx_msda_fused_backward(b16 ? VNX_BF16 : VNX_F32, VNX_F32, b16 ? (void*)s.value16 : (void*)s.value, dshapes, dlsi, s.off, s.logit,
                                 refpts, b16 ? (void*)s.go16 : (void*)s.go, b16 ? (void*)s.gv16 : (void*)s.gv, s.gl, s.ga, nullptr, B, S, M, D,
                                 L, Lq, P, 2, 1, s.ws, ws_bytes, st));
      return;
    }
    if (b16) {
      VK(vnx_msda_backward(VNX_BF16, VNX_F32, s.value16, dshapes, dlsi, s.loc, s.attn, s.go16, s.gv16, s.gl, s.ga, B, S, M, D, L, Lq, P,
                           VNX_MSDA_LEVELS_PACKED, s.ws, ws_bytes, st));
      return;
    }
    VK(vnx_msda_backward(VNX_F32, VNX_F32, s.value, dshapes, dlsi, s.loc, s.attn, s.go, s.gv, s.gl, s.ga, B, S, M, D, L, Lq, P,
                         VNX_MSDA_LEVELS_PACKED, s.ws, ws_bytes, st));
  };
  // bf16 results of set 0 -> its fp32 buffers (for the cross-check)
  auto widen = [&](Set& s, bool is_bwd) {
    if (!b16) return;
    if (!is_bwd) hipLaunchKernelGGL(bf16_to_f32, dim3(2048), dim3(256), 0, st, s.out16, s.out, n_out);
    else hipLaunchKernelGGL(bf16_to_f32, dim3(2048), dim3(256), 0, st, s.gv16, s.gv, n_value);
  };
  // reference results from the generic kernels (variant 1) on set 0
  float *r_out = nullptr, *r_gv = nullptr, *r_gl = nullptr, *r_ga = nullptr, *dd = nullptr;
  CK(hipMalloc(&dd, 8));
  if (check) {
    CK(hipMalloc(&r_out, n_out * 4)); CK(hipMalloc(&r_gv, n_value * 4)); CK(hipMalloc(&r_gl, n_s * 8)); CK(hipMalloc(&r_ga, n_s * 4));
    vnx_set_kernel_variant(1);
    if (op != "fbwd") { fwd(sets[0]); widen(sets[0], false); }
    if (op != "ffwd") { bwd(sets[0]); widen(sets[0], true); }     // (fbwd: only grad_value is comparable)
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(r_out, sets[0].out, n_out * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(r_gv, sets[0].gv, n_value * 4, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(r_gl, sets[0].gl, n_s * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(r_ga, sets[0].ga, n_s * 4, hipMemcpyDeviceToDevice));
  }
  auto diff = [&](const float* a, const float* b, size_t n) {
    CK(hipMemset(dd, 0, 8));
    hipLaunchKernelGGL(max_abs_diff, dim3(1024), dim3(256), 0, 0, a, b, n, dd);
    float h[2];
    CK(hipMemcpy(h, dd, 8, hipMemcpyDeviceToHost));
    return h[0] / std::max(h[1], 1e-30f);
  };
  bool step_mode = false;      // --op step: a graph of forward + backward pairs (the headline's step)
  auto time_graph = [&](bool is_bwd, bool cold) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < inner; ++i) { Set& s = sets[cold ? i % nsets : 0]; if (step_mode) { fwd(s); bwd(s); } else if (is_bwd) bwd(s); else fwd(s); }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    {   // warm-up: the first ~50 ms of a run are up to 10 % slow on this part (clocks / TLBs); replay until they are over
      const auto t0 = std::chrono::steady_clock::now();
      do { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < warm_s);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ts.push_back(ms * 1e3f / inner);
    }
    std::sort(ts.begin(), ts.end());
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ts[ts.size() / 2];
  };
  // plain launches on the stream, no graph: what an eager caller (a training step) pays per call incl. the host's launch path
  auto time_eager = [&](bool is_bwd) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int r = 0; r < reps + 3; ++r) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < inner; ++i) { Set& s = sets[i % nsets]; if (is_bwd) bwd(s); else fwd(s); }
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 3) ts.push_back(ms * 1e3f / inner);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
  };
  size_t pos = 0;
  while (pos < variants.size()) {
    size_t c = variants.find(',', pos);
    if (c == std::string::npos) c = variants.size();
    const int v = atoi(variants.substr(pos, c - pos).c_str());
    pos = c + 1;
    vnx_set_kernel_variant(v);
    for (int is_bwd = 0; is_bwd < 2; ++is_bwd) {
      if ((is_bwd && (op == "fwd" || op == "ffwd")) || (!is_bwd && (op == "bwd" || op == "fbwd"))) continue;
      char chk[256] = "";
      if (check) {
        if (!is_bwd) { fwd(sets[0]); widen(sets[0], false); CK(hipStreamSynchronize(st)); snprintf(chk, sizeof chk, " | relerr out %.2e", diff(sets[0].out, r_out, n_out)); }
        else {
          bwd(sets[0]); widen(sets[0], true); CK(hipStreamSynchronize(st));
          snprintf(chk, sizeof chk, " | relerr gv %.2e gloc %.2e gattn %.2e", diff(sets[0].gv, r_gv, n_value), diff(sets[0].gl, r_gl, 2 * n_s),
                   diff(sets[0].ga, r_ga, n_s));
        }
      }
      const float cold = time_graph(is_bwd, true), warm = cold_only ? cold : time_graph(is_bwd, false);
      const double by = is_bwd ? bytes_bwd : bytes_fwd;
      printf("  variant %4d %s: cold %8.2f us %6.2f TB/s %7.2f Gpt/s | warm %8.2f us %6.2f TB/s%s\n", v, is_bwd ? "bwd" : "fwd", cold,
             by / cold / 1e6, n_s / cold / 1e3, warm, by / warm / 1e6, chk);
      if (is_bwd && op == "step") {
        step_mode = true;
        const float sc = time_graph(true, true);
        step_mode = false;
        printf("  variant %4d fwd+bwd step: cold %8.2f us %7.2f Gpt/s\n", v, sc, n_s / sc / 1e3);
      }
      if (eager) printf("  variant %4d %s: eager (no graph) %8.2f us per call\n", v, is_bwd ? "bwd" : "fwd", time_eager(is_bwd));
      fflush(stdout);
    }
  }
  if (gvd_stamps) {   // mean phase durations of msda_bwd_gv_direct_kernel over its workgroups, cold inputs, grad_value kernel alone
    vnx_set_kernel_variant(442);
    const char* nm[8] = {"", "table", "issue", "load+decode+rank", "barrier", "scan+barrier", "scatter+rows+barrier", "walk+store"};
    double sum[8] = {0}, dur_sum = 0; int n = 0;
    std::vector<double> st0, en;
    for (int rep = 0; rep < 6; ++rep) {
      for (int i = 0; i < nsets; ++i) bwd(sets[i]);       // the last launch's stamps are read: preceded by cold launches
      CK(hipStreamSynchronize(st));
      std::vector<unsigned long long> hs(4096 * 8);
      vnx_debug_read_gvd_stamps(hs.data(), 4096 * 8);
      unsigned long long t0 = ~0ull;
      for (int w = 0; w < 4096; ++w) if (hs[w * 8 + 7] > hs[w * 8] && hs[w * 8 + 1]) t0 = std::min(t0, hs[w * 8]);
      for (int w = 0; w < 4096; ++w) {
        const unsigned long long* t = &hs[w * 8];
        if (!(t[7] > t[0]) || t[1] == 0) continue;
        ++n;
        for (int k = 1; k < 8; ++k) sum[k] += (t[k] - t[k - 1]) * 0.01;
        dur_sum += (t[7] - t[0]) * 0.01;
        st0.push_back((t[0] - t0) * 0.01); en.push_back((t[7] - t0) * 0.01);
      }
    }
    std::sort(st0.begin(), st0.end()); std::sort(en.begin(), en.end());
    auto pc = [](const std::vector<double>& x, double p) { return x.empty() ? 0.0 : x[std::min(x.size() - 1, size_t(p * x.size()))]; };
    printf("  gv_direct phases over %d workgroups (us):", n);
    for (int k = 1; k < 8; ++k) printf(" %s %.2f", nm[k], sum[k] / std::max(n, 1));
    printf(" | total %.2f | start p50 %.2f p90 %.2f max %.2f | end p50 %.2f p90 %.2f max %.2f\n", dur_sum / std::max(n, 1), pc(st0, .5), pc(st0, .9),
           st0.empty() ? 0 : st0.back(), pc(en, .5), pc(en, .9), en.empty() ? 0 : en.back());
    vnx_set_kernel_variant(0);
  }
  if (timeline) {   // per-workgroup {start, end} of the forward kernel on cold inputs (100 MHz wall clock, 10 ns ticks)
    size_t pos2 = 0;
    while (pos2 < variants.size()) {
      size_t c = variants.find(',', pos2);
      if (c == std::string::npos) c = variants.size();
      const int v = atoi(variants.substr(pos2, c - pos2).c_str());
      pos2 = c + 1;
      vnx_set_kernel_variant(v);
      const long long words = 2ll * 65536 * (nsets + 1);
      unsigned long long* buf;
      CK(hipMalloc(&buf, words * 8)); CK(hipMemset(buf, 0, words * 8));
      vnx_debug_arm_stamps(buf, words);
      for (int i = 0; i < nsets; ++i) { if (op == "bwd") bwd(sets[i]); else fwd(sets[i]); }
      CK(hipStreamSynchronize(st));
      int kinds[64]; long long offs[64], nblk[64];
      const int nreg = vnx_debug_stamp_regions(kinds, offs, nblk, 64);
      vnx_debug_arm_stamps(nullptr, 0);
      if (op == "bwd") {   // grad_value kernel: fixed stamp array, slots 0 (start) and 12 (end), variant 412
        vnx_set_kernel_variant(412);
        for (int i = 0; i < nsets; ++i) bwd(sets[i]);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> hs(4096 * 16);
        vnx_debug_read_rec_stamps(hs.data(), 4096 * 16);
        std::vector<double> st0, dur, en;
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < 4096; ++w) if (hs[w * 16 + 12] > hs[w * 16]) t0 = std::min(t0, hs[w * 16]);
        double hm[8] = {0}, hx[8] = {0}; int hn[8] = {0};
        for (int w = 0; w < 4096; ++w) if (hs[w * 16 + 12] > hs[w * 16]) {
          st0.push_back((hs[w * 16] - t0) * 0.01); en.push_back((hs[w * 16 + 12] - t0) * 0.01);
          dur.push_back((hs[w * 16 + 12] - hs[w * 16]) * 0.01);
          hm[w % 8] += en.back(); hx[w % 8] = std::max(hx[w % 8], en.back()); ++hn[w % 8];
        }
        std::sort(st0.begin(), st0.end()); std::sort(dur.begin(), dur.end()); std::sort(en.begin(), en.end());
        auto pc = [](const std::vector<double>& x, double p) { return x.empty() ? 0.0 : x[std::min(x.size() - 1, size_t(p * x.size()))]; };
        printf("  grad_value kernel (%zu workgroups, us): start p50 %.2f p90 %.2f max %.2f | duration p10 %.2f p50 %.2f p90 %.2f max %.2f | end p50 %.2f p90 %.2f max %.2f\n    end by blockIdx %% 8 (mean/max):",
               st0.size(), pc(st0, .5), pc(st0, .9), st0.empty() ? 0 : st0.back(), pc(dur, .1), pc(dur, .5), pc(dur, .9), dur.empty() ? 0 : dur.back(),
               pc(en, .5), pc(en, .9), en.empty() ? 0 : en.back());
        for (int k = 0; k < 8; ++k) printf(" %.1f/%.1f", hm[k] / std::max(hn[k], 1), hx[k]);
        printf("\n");
        {   // duration by unit index (blockIdx / (M * B): units are numbered from the coarsest level back)
          double um[32] = {0}, ux[32] = {0}; int un[32] = {0};
          for (int w = 0; w < 4096; ++w) if (hs[w * 16 + 12] > hs[w * 16]) {
            const int u = std::min(31, w / (M * B));
            const double dd = (hs[w * 16 + 12] - hs[w * 16]) * 0.01;
            um[u] += dd; ux[u] = std::max(ux[u], dd); ++un[u];
          }
          printf("    duration by unit (mean/max us):");
          for (int u = 0; u < 32; ++u) if (un[u]) printf(" %d:%.1f/%.1f", u, um[u] / un[u], ux[u]);
          printf("\n");
        }
        {   // phase stamps (diagnostic library builds with -DVNX_SEL_STAMPS fill slots 1..11), light vs heavy workgroups
          const char* nm[13] = {"", "table", "tags", "bar", "compact", "stage", "sort", "apply", "bar", "", "", "rest", "store"};
          for (int heavy = 0; heavy < 2; ++heavy) {
            double sum[13] = {0}; int n = 0;
            for (int w = 0; w < 4096; ++w) {
              const unsigned long long* t = &hs[w * 16];
              if (!(t[12] > t[0]) || t[1] == 0) continue;
              const double d = (t[12] - t[0]) * 0.01;
              if ((d > 11.0) != bool(heavy)) continue;
              ++n;
              unsigned long long prev = t[0];
              for (int k = 1; k <= 12; ++k) if (t[k] >= prev && t[k] != 0) { sum[k] += (t[k] - prev) * 0.01; prev = t[k]; }
            }
            if (n) { printf("    %s workgroups (%d): ", heavy ? "heavy (> 11 us)" : "light", n); for (int k = 1; k <= 12; ++k) if (nm[k][0]) printf(" %s %.2f", nm[k], sum[k] / n); printf("\n"); }
          }
        }
        vnx_set_kernel_variant(v);
      }
      if (nreg > 0) {
        const int r = std::min(nreg, 64) - 1;
        std::vector<unsigned long long> h(2 * nblk[r]);
        CK(hipMemcpy(h.data(), buf + offs[r], 16 * nblk[r], hipMemcpyDeviceToHost));
        std::vector<double> st0, dur, en;
        unsigned long long t0 = ~0ull;
        for (long long w = 0; w < nblk[r]; ++w) if (h[2 * w]) t0 = std::min(t0, h[2 * w]);
        for (long long w = 0; w < nblk[r]; ++w) if (h[2 * w]) {
          st0.push_back((h[2 * w] - t0) * 0.01); en.push_back((h[2 * w + 1] - t0) * 0.01); dur.push_back((h[2 * w + 1] - h[2 * w]) * 0.01);
        }
        std::sort(st0.begin(), st0.end()); std::sort(dur.begin(), dur.end()); std::sort(en.begin(), en.end());
        auto pc = [](const std::vector<double>& x, double p) { return x.empty() ? 0.0 : x[std::min(x.size() - 1, size_t(p * x.size()))]; };
        {   // who are the stragglers?  end time by head (blockIdx % 8 = XCD) and by position in the grid
          double hm[8] = {0}, hx[8] = {0}; int hn[8] = {0};
          const int nseg = 10; double sm[nseg] = {0}, sx[nseg] = {0}; int sn[nseg] = {0};
          for (long long w = 0; w < nblk[r]; ++w) if (h[2 * w]) {
            const double e = (h[2 * w + 1] - t0) * 0.01;
            const int hd = int(w % 8), sg = int(w * nseg / nblk[r]);   // hd = blockIdx % 8 (the XCD)
            hm[hd] += e; hx[hd] = std::max(hx[hd], e); ++hn[hd];
            sm[sg] += e; sx[sg] = std::max(sx[sg], e); ++sn[sg];
          }
          printf("    end by head  (mean/max):");
          for (int k = 0; k < 8; ++k) printf(" %.1f/%.1f", hm[k] / std::max(hn[k], 1), hx[k]);
          printf("\n    end by grid tenth (mean/max):");
          for (int k = 0; k < nseg; ++k) printf(" %.1f/%.1f", sm[k] / std::max(sn[k], 1), sx[k]);
          printf("\n");
        }
        printf("  timeline variant %d (%lld workgroups, us): start p10 %.2f p50 %.2f p90 %.2f max %.2f | duration p10 %.2f p50 %.2f p90 %.2f max %.2f | end p10 %.2f p50 %.2f p90 %.2f max %.2f\n",
               v, nblk[r], pc(st0, .1), pc(st0, .5), pc(st0, .9), st0.back(), pc(dur, .1), pc(dur, .5), pc(dur, .9), dur.back(), pc(en, .1), pc(en, .5),
               pc(en, .9), en.back());
      }
      CK(hipFree(buf));
    }
  }
  for (int which = 701; stamps && which <= 702; ++which) {   // phase stamps of the tiled forward (701: first item of
    vnx_set_kernel_variant(which);                             // every workgroup, 702: second item)
    fwd(sets[0]); CK(hipStreamSynchronize(st));
    fwd(sets[1 % nsets]); CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h(2048 * 16);
    vnx_debug_read_tile_stamps(h.data(), 2048 * 16);
    double sum[16] = {0}; int n = 0;
    unsigned long long t0min = ~0ull, t14max = 0;
    for (int w = 0; w < 2048; ++w) {
      const unsigned long long* t = &h[w * 16];
      if (t[14] <= t[0] || t[0] == 0) continue;
      ++n;
      for (int k = 1; k < 15; ++k) sum[k] += double(t[k] - t[k - 1]);
      t0min = std::min(t0min, t[0]); t14max = std::max(t14max, t[14]);
    }
    printf("  tile stamps over %d workgroups (first item each), mean ticks per phase:\n   ", n);
    const char* names[15] = {"", "setup+qtab", "decode", "barrier", "windows", "stage01", "wait+bar", "gather0", "bar+stage2", "gather1",
                             "wait+bar+stage3", "gather2", "wait+bar", "gather3", "store"};
    double tot = 0;
    for (int k = 1; k < 15; ++k) { printf(" %s=%.0f", names[k], sum[k] / std::max(n, 1)); tot += sum[k] / std::max(n, 1); }
    printf("\n    total %.0f ticks per item; first start -> last end of first items %.0f ticks\n", tot, double(t14max - t0min));
  }
  vnx_set_kernel_variant(0);
  return 0;
}
