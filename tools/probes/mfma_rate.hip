// Probe: what the fp16 matrix pipe of gfx950 sustains under full load, and at which shader clock.
//   mfma_rate [mode] [data]   mode 0 = 8 waves/CU all issuing MFMAs (two per SIMD), 1 = 4 waves/CU (one per SIMD),
//                             2 = 8 waves/CU ping-pong (half the waves in an MFMA burst, half reading LDS)
//                             data 0 = zeros, 1 = random fp16 values (power-hungry, like real activations)
// Prints achieved TFLOP/s over the whole chip, cycles per MFMA per SIMD, and the shader clock measured as
// s_memtime ticks per s_memrealtime tick (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(const halfx8* src, float* sink, unsigned long long* stamps, int iters) {
  __shared__ halfx8 lds[4096];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = src[(blockIdx.x * 4096 + i) & 65535];
  __syncthreads();
  halfx8 a[4], b[4];
  for (int q = 0; q < 4; ++q) { a[q] = src[(tid * 4 + q) & 65535]; b[q] = src[(tid * 4 + q + 17) & 65535]; }
  floatx16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const bool odd = wave >= 4;
  if (MODE == 2 && odd) __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {  // L segment: 16 fragment reads
      halfx8 f[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) f[q] = lds[(tid * 16 + q * 67 + it) & 4095];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q) { a[q] = f[q] ; b[q] = f[4 + q]; }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int g = 0; g < 6; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(g + i) & 3], b[(g * 3 + i) & 3], acc[i], 0, 0, 0);
    if (MODE == 2) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 2 && !odd) __builtin_amdgcn_s_barrier();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  sink[blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) { stamps[(blockIdx.x * 8 + wave) * 2] = t1 - t0; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0, data = argc > 2 ? atoi(argv[2]) : 1, iters = 20000;
  std::vector<_Float16> h(65536 * 8);
  srand(1);
  for (auto& v : h) v = data ? (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f) : (_Float16)0.f;
  halfx8* d; float* sink; unsigned long long* st;
  hipMalloc(&d, h.size() * 2); hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int threads = mode == 1 ? 256 : 512, blocks = 256;
  hipMalloc(&sink, blocks * 512 * 4); hipMalloc(&st, blocks * 8 * 2 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(threads), 0, 0, d, sink, st, iters);
    else hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(threads), 0, 0, d, sink, st, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs(blocks * 16);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0; const int nw = threads / 64;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < nw; ++w) { cyc += hs[(b * 8 + w) * 2]; real += hs[(b * 8 + w) * 2 + 1]; }
    const double waves = blocks * nw, mfma = 24.0 * iters;
    const double flops = waves * mfma * 2.0 * 32 * 32 * 16;
    printf("mode %d data %d: %.1f ms  %.0f TFLOP/s  cycles/MFMA/wave %.1f  clock %.0f MHz\n", mode, data, ms, flops / ms / 1e9,
           cyc / waves / mfma, cyc / real * 100.0);
  }
  return 0;
}
