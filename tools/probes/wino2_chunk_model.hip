// Probe: cycles per 16-channel chunk of a Winograd F(2x2,3x3) kernel on split-fp16 operands with the tiling
//   item = 64 tiles x 64 couts, 8 waves = 2 tile halves (32 tiles) x 4 position rows (4 positions), v_mfma_f32_32x32x16_f16
// -- the instruction MIX only (no real data flow): per wave and chunk 16 ds_read_b128 of pixels, the row / column
// transform and the hi/lo split as VALU work, 16 ds_read_b128 of weight fragments, 24 MFMAs; per thread 8 + 4 16-B global
// loads and ds_write_b128 of the next chunk's staging, two workgroup barriers.  Tells whether the shape is worth building.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(const floatx4* src, float* sink, unsigned long long* stamps, int iters, int mode) {
  extern __shared__ floatx4 lds[];  // [pix 1280][U 2 x 4096]
  floatx4* const pix = lds;
  floatx4* const ul = lds + 1280;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 1280 + 8192; i += 512) lds[i] = src[(blockIdx.x * 977 + i) & 65535];
  __syncthreads();
  floatx16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  floatx4 stage_p[4], stage_u[8];
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  int ubuf = 0;
  for (int it = 0; it < iters; ++it) {
    // global loads of the next chunk's staging (L2 resident source)
    if (mode & 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) stage_p[k] = src[(it * 4099 + tid * 4 + k * 2048 + blockIdx.x * 64) & 65535];
#pragma unroll
      for (int k = 0; k < 8; ++k) stage_u[k] = src[(it * 8209 + tid * 8 + k + 32768) & 65535];
    }
    // pixels of this lane: 2 rows x 4 columns x 2 channel quads
    floatx4 d[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) d[p] = pix[((lane & 31) * 9 + (p & 7) * 33 + (p >> 3) * 320 + (lane >> 5) * 640 + it) % 1280];
    // row + column transform of one position row, 8 channels: 4 + 8 float4 ops per channel quad
    floatx4 V[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      floatx4 t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = d[8 * h + j] - d[8 * h + 4 + j];
      V[4 * h + 0] = t[0] - t[2]; V[4 * h + 1] = t[1] + t[2]; V[4 * h + 2] = t[2] - t[1]; V[4 * h + 3] = t[1] - t[3];
    }
    halfx8 vh[4], vl[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const halfx4 h0 = __builtin_convertvector(V[p], halfx4), h1 = __builtin_convertvector(V[4 + p], halfx4);
      const halfx4 l0 = __builtin_convertvector(V[p] - __builtin_convertvector(h0, floatx4), halfx4);
      const halfx4 l1 = __builtin_convertvector(V[4 + p] - __builtin_convertvector(h1, floatx4), halfx4);
      vh[p] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      vl[p] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    const floatx4* ub = ul + ubuf * 4096;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const halfx8 wh = __builtin_bit_cast(halfx8, ub[((p * 2 + nb) * 64 + lane) * 2 + (tid >> 6) * 8 % 3072]);
        const halfx8 wl = __builtin_bit_cast(halfx8, ub[((p * 2 + nb) * 64 + lane) * 2 + 1 + (tid >> 6) * 8 % 3072]);
        floatx16 c = acc[p * 2 + nb];
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, vh[p], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, vl[p], c, 0, 0, 0);
        acc[p * 2 + nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, vh[p], c, 0, 0, 0);
      }
    if (mode & 1) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) pix[(tid + k * 512 + it) % 1280] = stage_p[k];
#pragma unroll
      for (int k = 0; k < 8; ++k) ul[(ubuf ^ 1) * 4096 + (tid * 8 + k) % 4096] = stage_u[k];
      __syncthreads();
    }
    ubuf ^= 1;
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  sink[blockIdx.x * 512 + tid] = s;
  if (lane == 0) { stamps[(blockIdx.x * 8 + (tid >> 6)) * 2] = t1 - t0; stamps[(blockIdx.x * 8 + (tid >> 6)) * 2 + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 1, iters = 4000;
  std::vector<float> h(65536 * 4);
  srand(1);
  for (auto& v : h) v = (rand() % 2001 - 1000) / 1000.0f;
  floatx4* d; float* sink; unsigned long long* st;
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipMalloc(&d, h.size() * 4); hipMalloc(&sink, (size_t)cus * 512 * 4); hipMalloc(&st, (size_t)cus * 16 * 8);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const size_t lds = (1280 + 8192) * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(cus), dim3(512), lds, 0, d, sink, st, iters, mode);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> hs((size_t)cus * 16);
  hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < cus * 8; ++i) { cyc += hs[2 * i]; rt += hs[2 * i + 1]; }
  cyc /= cus * 8; rt /= cus * 8;
  printf("mode %d: %.0f cycles per chunk (64 tiles x 64 couts x 16 channels), shader clock %.0f MHz, %.2f us per chunk; pure MFMA time 24 x 32 x 2 waves = 1536 cycles; direct split conv: 3456 MFMA cycles for the same outputs\n",
         mode, cyc / iters, cyc / rt * 100.0, rt / 100.0 / iters);
  return 0;
}
