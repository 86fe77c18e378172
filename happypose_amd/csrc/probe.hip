// Diagnostics: what the fp16 matrix pipe of this GPU sustains, measured -- the practical ceiling the conv roofline is
// read against.  gfx950 clocks to its power budget: with zero operands v_mfma_f32_32x32x16_f16 runs at ~2.4 GHz and
// reaches the dense peak of MI355X_MICROARCH.md (2.5 PFLOP/s); with random fp16 operands (what activations and weights
// are) the same instruction stream pulls the shader clock down to ~1.6 GHz and sustains about two thirds of that.
// hp_probe_mfma_rate launches one 8-wave workgroup per CU, two waves per SIMD issuing back-to-back MFMAs on four
// independent accumulators from registers, and reports chip-wide TFLOP/s plus the shader clock (s_memtime ticks per
// 100-MHz s_memrealtime tick).
#include <vector>

#include "common.h"

namespace hp {
namespace {

typedef _Float16 pr_halfx8 __attribute__((ext_vector_type(8)));
typedef float pr_floatx16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_rate_kernel(const pr_halfx8* src, float* sink, unsigned long long* stamps, int iters) {
  const int tid = threadIdx.x;
  pr_halfx8 a[4], b[4];
  for (int q = 0; q < 4; ++q) { a[q] = src[(tid * 4 + q) & 4095]; b[q] = src[(tid * 4 + q + 17) & 4095]; }
  pr_floatx16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 6; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(g + i) & 3], b[(g * 3 + i) & 3], acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  sink[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

}  // namespace
}  // namespace hp

extern "C" int hp_probe_mfma_rate(int random_data, double* tflops, double* shader_mhz, void* stream) {
  using namespace hp;
  HP_REQUIRE(tflops && shader_mhz, "hp_probe_mfma_rate: null output");
  int dev = 0, cus = 256;
  HP_CHECK_HIP(hipGetDevice(&dev));
  HP_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  std::vector<_Float16> h(4096 * 8);
  unsigned s = 12345u;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = random_data ? (_Float16)(((s >> 8) / 16777216.0f - 0.5f) * 4.f) : (_Float16)0.f;
  }
  pr_halfx8* d = nullptr; float* sink = nullptr; unsigned long long* st = nullptr;
  HP_CHECK_HIP(hipMalloc((void**)&d, h.size() * 2));
  HP_CHECK_HIP(hipMalloc((void**)&sink, (size_t)cus * 512 * 4));
  HP_CHECK_HIP(hipMalloc((void**)&st, (size_t)cus * 16));
  HP_CHECK_HIP(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  hipStream_t q = (hipStream_t)stream;
  hipEvent_t e0, e1;
  HP_CHECK_HIP(hipEventCreate(&e0)); HP_CHECK_HIP(hipEventCreate(&e1));
  const int iters = 4000;
  float ms = 0.f;
  for (int rep = 0; rep < 4; ++rep) {  // the clock needs a few milliseconds to settle: the last repetition counts
    HP_CHECK_HIP(hipEventRecord(e0, q));
    hipLaunchKernelGGL(mfma_rate_kernel, dim3(cus), dim3(512), 0, q, d, sink, st, iters);
    HP_CHECK_HIP(hipEventRecord(e1, q));
    HP_CHECK_HIP(hipEventSynchronize(e1));
    HP_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  }
  std::vector<unsigned long long> hs((size_t)cus * 2);
  HP_CHECK_HIP(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, real = 0;
  for (int b = 0; b < cus; ++b) { cyc += (double)hs[2 * b]; real += (double)hs[2 * b + 1]; }
  *tflops = (double)cus * 8 * 24.0 * iters * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
  *shader_mhz = real > 0 ? cyc / real * 100.0 : 0.0;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(d); (void)hipFree(sink); (void)hipFree(st);
  return HP_OK;
}
