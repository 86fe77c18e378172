// Probe: does v_mfma_f32_32x32x16_f16 on gfx950 honour fp16 subnormal inputs?  (The split-fp16 conv
// path relies on it for the low halves of small activations.)  Prints the products of a subnormal A
// entry with a normal B entry; expected 2^-20 * 2^10 = 2^-10 = 9.765625e-04 when not flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void probe(float* out, float asub) {
  const int lane = threadIdx.x;
  halfx8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
  // A[row = lane & 31][k = 8 (lane >> 5) + q], B[k][col = lane & 31]
  if (lane == 0) a[0] = (_Float16)asub;     // A[0][0] subnormal
  if (lane < 32) b[0] = (_Float16)1024.0f;  // B[0][*] = 2^10
  floatx16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  // C[row][col]: lane holds col = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  if (lane == 0) out[0] = c[0];
  if (lane == 1) out[1] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 8);
  for (float s : {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24 */, 6.103515625e-05f /* 2^-14 normal */}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, s);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a = %.10e  -> a * 1024 via MFMA = %.10e %.10e (exact %.10e)\n", s, h[0], h[1], s * 1024.0f);
  }
  return 0;
}
