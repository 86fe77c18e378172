// Probe: what bounds the tile epilogue of the 3x3 conv kernels -- the HBM write rate of the whole chip, or the store
// path of one CU?  Every workgroup (512 threads, one per CU: 134 KB of dynamic LDS) writes T tiles of 256 rows x 512 B
// (a 256 x 128 fp32 output tile of a Cout = 128 layer) in one of three patterns, with G workgroups in the grid:
//   0  16 x global_store_dwordx4 per thread, thread -> (row = tid / 32 + 16 k, 16-B piece tid % 32)   (conv_epilogue.h)
//   1  64 x global_store_dword per thread, wave w lane l -> 2 rows x 128 B per instruction               (accumulator layout)
//   2  pattern 0, but the stores are issued INSIDE an MFMA loop (one store per `gap` MFMAs): does a wave that keeps
//      multiplying hide them?
// If the time per tile does not depend on G the bound is per CU (issue / latency); if it grows with G it is bandwidth.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/store_burst tools/probes/store_burst.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

__global__ __launch_bounds__(512) void burst(float* y, int tiles, int pattern, int mfma_per_tile, int gap, float* sink) {
  extern __shared__ unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f4 v = {(float)tid, 1.f, 2.f, 3.f};
  f16v acc = {0};
  h8 a = {(_Float16)(tid & 7), 1, 2, 3, 4, 5, 6, 7}, b = {(_Float16)1, 2, 3, (_Float16)(lane & 3), 5, 6, 7, 8};
  for (int t = 0; t < tiles; ++t) {
    float* const base = y + ((size_t)blockIdx.x * tiles + t) * 256 * 128;
    if (pattern == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int row = tid / 32 + 16 * k;
        *reinterpret_cast<f4*>(base + row * 128 + 4 * (tid % 32)) = v;
      }
      for (int i = 0; i < mfma_per_tile; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    } else if (pattern == 1) {
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        const int row = wave * 32 + (k & 15) * 2 + (lane >> 5);  // two rows x 32 lanes x 4 B = 2 x 128 B per instruction
        base[row * 128 + (k >> 4) * 32 + (lane & 31)] = v[0];
      }
      for (int i = 0; i < mfma_per_tile; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    } else {
      int k = 0;
      for (int i = 0; i < mfma_per_tile; ++i) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (i % gap == 0 && k < 16) {
          const int row = tid / 32 + 16 * k;
          *reinterpret_cast<f4*>(base + row * 128 + 4 * (tid % 32)) = v;
          ++k;
        }
      }
      for (; k < 16; ++k) {
        const int row = tid / 32 + 16 * k;
        *reinterpret_cast<f4*>(base + row * 128 + 4 * (tid % 32)) = v;
      }
    }
  }
  if (acc[0] == 12345.f) sink[0] = acc[3];
}

int main(int argc, char** argv) {
  const int tiles = 8;
  float* y; float* sink;
  CK(hipMalloc(&y, (size_t)256 * tiles * 256 * 128 * 4));
  CK(hipMalloc(&sink, 64));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&burst), hipFuncAttributeMaxDynamicSharedMemorySize, 134 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int G, int pattern, int mfma, int gap) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(burst, dim3(G), dim3(512), 134 * 1024, 0, y, tiles, pattern, mfma, gap, sink);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(burst, dim3(G), dim3(512), 134 * 1024, 0, y, tiles, pattern, mfma, gap, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_tile = ms * 1e3 / reps / tiles;
    const double tbs = (double)G * 131072.0 / (us_tile * 1e-6) / 1e12;
    std::printf("G %3d pattern %d mfma/tile %5d gap %3d : %7.2f us per tile, %6.2f TB/s chip-wide, %5.2f B/clk/CU at 2.4 GHz\n", G, pattern, mfma, gap,
                us_tile, tbs, 131072.0 / (us_tile * 1e-6 * 2.4e9));
  };
  for (int pattern = 0; pattern < 2; ++pattern)
    for (int G : {16, 32, 64, 128, 256}) run(G, pattern, 0, 1);
  // MFMA only (no stores: pattern 2 with gap beyond the loop never stores inside; the tail loop still stores -> use tiles of pure MFMA via pattern 0 + mfma)
  for (int G : {64, 256}) {
    run(G, 0, 864, 1);   // stores, THEN 864 MFMAs per wave (36 taps x 24): serial
    run(G, 2, 864, 96);  // the same MFMAs with a store every 96
    run(G, 2, 864, 24);
  }
  return 0;
}
