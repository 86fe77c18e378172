// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the rasteriser (the guide only
// calibrates 16 B/lane streaming reads: MI355X_MICROARCH.md "HBM").  Each kernel moves a known number of bytes through a
// 1 GiB buffer (4x the Infinity Cache); tools/pmc_calibrate.py divides the counters by those bytes.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calibrate tools/probes/pmc_calibrate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

extern "C" __global__ void cal_read16(const uint4* __restrict__ p, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = p[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345u) *sink = acc;
}
extern "C" __global__ void cal_read4(const uint32_t* __restrict__ p, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 0x12345u) *sink = acc;
}
extern "C" __global__ void cal_write16(uint4* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
extern "C" __global__ void cal_write4(uint32_t* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
// the band kernel's output pass: a workgroup owns 2 rows x 320 px of one view and writes them to `ch` NCHW planes
extern "C" __global__ void cal_write4_bands(float* p, int views, int ch, int h, int w) {
  int bands = h / 2;
  for (int b = blockIdx.x; b < views * bands; b += gridDim.x) {
    int v = b / bands, r0 = (b % bands) * 2;
    for (int c = 0; c < ch; ++c)
      for (int i = threadIdx.x; i < 2 * w; i += blockDim.x)
        p[((size_t)(v * ch + c) * h + r0 + i / w) * w + i % w] = (float)i;
  }
}
// one lane = one 16-B gather at a hashed index (texture row-pair taps)
extern "C" __global__ void cal_gather16(const uint4* __restrict__ p, size_t n, size_t count, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = p[mix((uint32_t)i) % n];
    acc += v.x ^ v.w;
  }
  if (acc == 0x12345u) *sink = acc;
}
// one lane = one 64-B record (4 x 16 B) at a hashed index (set-up records read by the band kernel)
extern "C" __global__ void cal_gather64(const uint4* __restrict__ p, size_t nrec, size_t count, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const uint4* r = p + (size_t)(mix((uint32_t)i) % nrec) * 4;
    uint4 a = r[0], b = r[1], c = r[2], d = r[3];
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345u) *sink = acc;
}
// one lane = one 64-B record write at consecutive slots (the set-up kernel's record store)
extern "C" __global__ void cal_write64_rec(uint4* p, size_t nrec) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
    uint4* r = p + i * 4;
    uint4 v = make_uint4((uint32_t)i, 1u, 2u, 3u);
    r[0] = v; r[1] = v; r[2] = v; r[3] = v;
  }
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  void *a, *sink;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 1, bytes));
  CK(hipDeviceSynchronize());
  const int grid = 256 * 8, blk = 256;
  const size_t gathers = (size_t)1 << 24;
  const int views = 256, ch = 7, h = 240, w = 320;
  for (int rep = 0; rep < 3; ++rep) {
    cal_read16<<<grid, blk>>>((const uint4*)a, bytes / 16, (uint32_t*)sink);
    cal_read4<<<grid, blk>>>((const uint32_t*)a, bytes / 4, (uint32_t*)sink);
    cal_write16<<<grid, blk>>>((uint4*)a, bytes / 16);
    cal_write4<<<grid, blk>>>((uint32_t*)a, bytes / 4);
    cal_write4_bands<<<grid, blk>>>((float*)a, views, ch, h, w);
    cal_gather16<<<grid, blk>>>((const uint4*)a, bytes / 16, gathers, (uint32_t*)sink);
    cal_gather64<<<grid, blk>>>((const uint4*)a, bytes / 64, gathers, (uint32_t*)sink);
    cal_write64_rec<<<grid, blk>>>((uint4*)a, bytes / 64);
    CK(hipDeviceSynchronize());
  }
  printf("bytes cal_read16 %zu\nbytes cal_read4 %zu\nbytes cal_write16 %zu\nbytes cal_write4 %zu\nbytes cal_write4_bands %zu\n"
         "bytes cal_gather16 %zu\nbytes cal_gather64 %zu\nbytes cal_write64_rec %zu\n",
         bytes, bytes, bytes, bytes, (size_t)views * ch * h * w * 4, gathers * 16, gathers * 64, bytes);
  return 0;
}
