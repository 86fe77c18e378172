// Probe for the hazard behind the two-lane non-determinism ("the co-scheduling non-determinism"): do packed-fp32
// instructions with operand swizzles (what clang's SLP vectoriser forms from scalar fp32 code: v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 with op_sel / neg modifiers) return the values of the scalar instructions when ANOTHER queue's MFMA stream
// shares the SIMDs?
//   victim  (stream A): every lane runs the plane set-up arithmetic of raster.hip's setup_subtri on data it loads from global
//                       memory -- once as the compiler packs it (this translation unit is built WITH the SLP vectoriser) and
//                       once in a noinline copy built from asm-fenced scalar operations -- and counts lanes whose bits differ;
//   aggressor (stream B): back-to-back v_mfma_f32_32x32x16_f16 on every SIMD.
//   pk_beside_mfma [launches] [aggressor 0/1]
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/pk_beside_mfma tools/probes/pk_beside_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

struct In { float px[3], py[3], w[3], u[3], v[3]; float area; };
struct Out { float nu[3], nv[3], wp[3]; };

#pragma clang fp contract(off)
// the arithmetic as raster.hip writes it: adjacent u / v chains, which the SLP vectoriser packs
__device__ __forceinline__ Out planes_packed(const In& s) {
  const float inv = 1.0f / (s.area * (1.0f / 65536.0f));
  const float dx1 = s.px[1] - s.px[0], dy1 = s.py[1] - s.py[0], dx2 = s.px[2] - s.px[0], dy2 = s.py[2] - s.py[0];
  Out o;
  float qu[3], qv[3];
  for (int k = 0; k < 3; ++k) { qu[k] = s.u[k] * s.w[k]; qv[k] = s.v[k] * s.w[k]; }
  const float du1 = qu[1] - qu[0], du2 = qu[2] - qu[0], dv1 = qv[1] - qv[0], dv2 = qv[2] - qv[0];
  o.nu[1] = fmaf(du1, dy2, -(du2 * dy1)) * inv; o.nv[1] = fmaf(dv1, dy2, -(dv2 * dy1)) * inv;
  o.nu[2] = fmaf(du2, dx1, -(du1 * dx2)) * inv; o.nv[2] = fmaf(dv2, dx1, -(dv1 * dx2)) * inv;
  o.nu[0] = fmaf(-o.nu[2], s.py[0], fmaf(-o.nu[1], s.px[0], qu[0]));
  o.nv[0] = fmaf(-o.nv[2], s.py[0], fmaf(-o.nv[1], s.px[0], qv[0]));
  const float dw1 = s.w[1] - s.w[0], dw2 = s.w[2] - s.w[0];
  o.wp[1] = fmaf(dw1, dy2, -(dw2 * dy1)) * inv; o.wp[2] = fmaf(dw2, dx1, -(dw1 * dx2)) * inv;
  o.wp[0] = fmaf(-o.wp[2], s.py[0], fmaf(-o.wp[1], s.px[0], s.w[0]));
  return o;
}
// the same operations, every result fenced through an asm statement: nothing for the vectoriser to pair
__device__ __forceinline__ float F(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __noinline__ Out planes_scalar(const In& s) {
  const float inv = F(1.0f / F(s.area * (1.0f / 65536.0f)));
  const float dx1 = F(s.px[1] - s.px[0]), dy1 = F(s.py[1] - s.py[0]), dx2 = F(s.px[2] - s.px[0]), dy2 = F(s.py[2] - s.py[0]);
  Out o;
  float qu[3], qv[3];
  for (int k = 0; k < 3; ++k) { qu[k] = F(s.u[k] * s.w[k]); qv[k] = F(s.v[k] * s.w[k]); }
  const float du1 = F(qu[1] - qu[0]), du2 = F(qu[2] - qu[0]), dv1 = F(qv[1] - qv[0]), dv2 = F(qv[2] - qv[0]);
  o.nu[1] = F(F(fmaf(du1, dy2, -F(du2 * dy1))) * inv); o.nv[1] = F(F(fmaf(dv1, dy2, -F(dv2 * dy1))) * inv);
  o.nu[2] = F(F(fmaf(du2, dx1, -F(du1 * dx2))) * inv); o.nv[2] = F(F(fmaf(dv2, dx1, -F(dv1 * dx2))) * inv);
  o.nu[0] = F(fmaf(-o.nu[2], s.py[0], F(fmaf(-o.nu[1], s.px[0], qu[0]))));
  o.nv[0] = F(fmaf(-o.nv[2], s.py[0], F(fmaf(-o.nv[1], s.px[0], qv[0]))));
  const float dw1 = F(s.w[1] - s.w[0]), dw2 = F(s.w[2] - s.w[0]);
  o.wp[1] = F(F(fmaf(dw1, dy2, -F(dw2 * dy1))) * inv); o.wp[2] = F(F(fmaf(dw2, dx1, -F(dw1 * dx2))) * inv);
  o.wp[0] = F(fmaf(-o.wp[2], s.py[0], F(fmaf(-o.wp[1], s.px[0], s.w[0]))));
  return o;
}

__global__ __launch_bounds__(256) void victim(const In* in, int n, unsigned long long* bad, Out* first_bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const In s = in[i];
  const Out a = planes_packed(s), b = planes_scalar(s);
  bool same = true;
  for (int k = 0; k < 3; ++k)
    same &= __float_as_uint(a.nu[k]) == __float_as_uint(b.nu[k]) && __float_as_uint(a.nv[k]) == __float_as_uint(b.nv[k]) &&
            __float_as_uint(a.wp[k]) == __float_as_uint(b.wp[k]);
  if (!same && atomicAdd(bad, 1ull) == 0) { first_bad[0] = a; first_bad[1] = b; }
}

__global__ __launch_bounds__(512) void aggressor(const halfx8* src, float* sink, int iters) {
  const int tid = threadIdx.x;
  halfx8 a[4], b[4];
  for (int q = 0; q < 4; ++q) { a[q] = src[(tid * 4 + q) & 65535]; b[q] = src[(tid * 4 + q + 17) & 65535]; }
  floatx16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int g = 0; g < 6; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(g + i) & 3], b[(g * 3 + i) & 3], acc[i], 0, 0, 0);
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  sink[blockIdx.x * blockDim.x + tid] = s;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 2000, with_mfma = argc > 2 ? atoi(argv[2]) : 1;
  const int n = 1 << 20;
  std::vector<In> h(n);
  srand(7);
  auto rnd = [](float lo, float hi) { return lo + (hi - lo) * (float)rand() / (float)RAND_MAX; };
  for (auto& s : h) {
    for (int k = 0; k < 3; ++k) { s.px[k] = rnd(0.f, 4.f); s.py[k] = rnd(0.f, 4.f); s.w[k] = rnd(1.2f, 2.5f); s.u[k] = rnd(0.f, 1.f); s.v[k] = rnd(0.f, 1.f); }
    s.area = rnd(3000.f, 200000.f);
  }
  In* d_in; unsigned long long* d_bad; Out* d_first; halfx8* d_src; float* d_sink;
  hipMalloc(&d_in, n * sizeof(In)); hipMemcpy(d_in, h.data(), n * sizeof(In), hipMemcpyHostToDevice);
  hipMalloc(&d_bad, 8); hipMemset(d_bad, 0, 8); hipMalloc(&d_first, 2 * sizeof(Out));
  std::vector<_Float16> hs(65536 * 8);
  for (auto& x : hs) x = (_Float16)rnd(-1.f, 1.f);
  hipMalloc(&d_src, hs.size() * 2); hipMemcpy(d_src, hs.data(), hs.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&d_sink, 256 * 512 * 4);
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  for (int l = 0; l < launches; ++l) {
    if (with_mfma && (l & 3) == 0) hipLaunchKernelGGL(aggressor, dim3(256), dim3(512), 0, sb, d_src, d_sink, 400);
    hipLaunchKernelGGL(victim, dim3(n / 256), dim3(256), 0, sa, d_in, n, d_bad, d_first);
  }
  hipDeviceSynchronize();
  unsigned long long bad = 0; Out fb[2];
  hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(fb, d_first, sizeof(fb), hipMemcpyDeviceToHost);
  printf("{\"launches\": %d, \"lanes_per_launch\": %d, \"aggressor\": %d, \"lanes_whose_packed_result_differs\": %llu", launches, n, with_mfma, bad);
  if (bad) printf(", \"first\": {\"packed_nu\": [%g, %g, %g], \"scalar_nu\": [%g, %g, %g]}", fb[0].nu[0], fb[0].nu[1], fb[0].nu[2], fb[1].nu[0], fb[1].nu[1], fb[1].nu[2]);
  printf("}\n");
  return 0;
}
