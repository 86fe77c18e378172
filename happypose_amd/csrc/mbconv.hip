// Memory-bound pieces of the MBConv block of EfficientNet (CP/models/efficientnet.py:20-150,
// the backbone of the released CosyPose checkpoints -- SURVEY.md 8f-1): depthwise k x k
// convolution + folded BatchNorm + swish, and squeeze-excitation (global mean -> 1x1 reduce ->
// swish -> 1x1 expand -> sigmoid).  The 1x1 expansion / projection convolutions are GEMMs and
// run on the split-fp16 implicit-GEMM kernel (conv_igemm_split.hip; conv.hip when the exact-fp32 kernels are forced),
// which also applies the excitation gate while it stages the projection's input, so the gated tensor never exists in memory.
//
// All tensors NHWC fp32 with C % 4 == 0 (every EfficientNet width is a multiple of 8).
// "Same" padding follows Conv2dStaticSamePadding (efficientnet_utils.py:183-212): the pad
// (top/left, bottom/right) is fixed per layer by the planner; bottom/right padding is implicit
// in the bounds test.
#include "conv.h"

#include <cstdlib>

namespace hp {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ floatx4 swish4(floatx4 v) {
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = v[q] * __builtin_amdgcn_rcpf(1.f + __expf(-v[q]));  // v_rcp_f32: 1 ulp (mbconv_front.hip)
  return v;
}

// one lane = 4 channels of one output pixel; consecutive lanes = consecutive channel quads, so a
// wave reads / writes contiguous 1-KB runs; the k*k taps of neighbouring pixels are L1/L2 hits
template <int K>
__global__ __launch_bounds__(256) void dwconv_swish_nhwc(DwArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int C4 = a.C / 4;
  if (idx >= (int64_t)a.n * a.Ho * a.Wo * C4) return;
  const int c = (int)(idx % C4) * 4;
  int64_t p = idx / C4;
  const int ow = (int)(p % a.Wo); p /= a.Wo;
  const int oh = (int)(p % a.Ho);
  const int img = (int)(p / a.Ho);
  floatx4 acc = *reinterpret_cast<const floatx4*>(a.bias + c);
  const int ih0 = oh * a.stride - a.pad_t, iw0 = ow * a.stride - a.pad_l;
#pragma unroll
  for (int dy = 0; dy < K; ++dy) {
    const int ih = ih0 + dy;
    if ((unsigned)ih >= (unsigned)a.H) continue;
#pragma unroll
    for (int dx = 0; dx < K; ++dx) {
      const int iw = iw0 + dx;
      if ((unsigned)iw >= (unsigned)a.W) continue;
      const floatx4 xv = *reinterpret_cast<const floatx4*>(a.x + (((int64_t)img * a.H + ih) * a.W + iw) * a.C + c);
      const floatx4 wv = *reinterpret_cast<const floatx4*>(a.w + (dy * K + dx) * a.C + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = fmaf(xv[q], wv[q], acc[q]);
    }
  }
  *reinterpret_cast<floatx4*>(a.y + (((int64_t)img * a.Ho + oh) * a.Wo + ow) * a.C + c) = swish4(acc);
}

// Strip kernel: a workgroup owns R output rows of one image for an even share of at most 64 channel quads (dw_quads_per_block) and
// as many column phases as fit 256 lanes.  A lane keeps ONE channel quad (3 x 3: its taps stay in registers; 5 x 5: in LDS, see WLDS)
// and walks the columns of the strip; per column it streams the (R-1)*S+K input
// rows once and feeds every output row they reach, so an output costs ((R-1)*S+K)*K/R quad loads instead of K*K
// (k5: 10 vs 25 -- the one-output-per-lane kernel above is bound by the L1 request rate, not by HBM).  The FMA order
// per output is the same (dy, dx ascending), so the conv result is bit-identical.  The lanes also sum what they store:
// the per-(image, strip, channel) sums are the squeeze-excitation pooling partials (se_pool_kernel re-read the whole
// tensor for them); lanes of one channel quad meet in LDS in a fixed order (reproducible).
// output rows per strip: 3 x 3: 8 on the maps with >= 30 rows (input rows re-read by neighbouring strips: (R-1)S+K for R
// outputs), 4 on the small maps, where strips are what fills the GPU; 5 x 5 (taps in LDS, see WLDS): 4 everywhere (round 5,
// one forward of 64: all 5 x 5 launches 860 us at 4 rows, 897 at 8 rows on the 30-row maps, 972 at 8 rows everywhere)
__host__ __device__ constexpr int dw_rows(int Ho, int k, int stride) { return Ho >= 30 && k == 3 ? 8 : 4; }

__host__ __device__ inline int dw_quads_per_block(int C4) {
  const int nb = (C4 + 63) / 64;
  return (C4 + nb - 1) / nb;
}

// WLDS (the 5 x 5 kernels): the taps live in LDS ([tap][quad], read at use) instead of 100 registers per lane -- with them in
// registers the 5 x 5 instantiations need all 256 VGPRs: ONE wave per SIMD, every tap load of a column exposed (66 us for a
// 126-MB layer).  Same FMA order, bit-identical outputs.
template <int K, int S, int R, bool WLDS>
__global__ __launch_bounds__(256) void dwconv_strip_kernel(DwArgs a) {
  constexpr int NR = (R - 1) * S + K;
  __shared__ floatx4 red[256];
  __shared__ floatx4 wl[WLDS ? K * K * 64 : 1];
  const int C4 = a.C >> 2;
  // channel quads per workgroup: even shares of at most 64 (dw_quads_per_block) -- 256 per workgroup left 112 of 256 lanes idle
  // at C = 576 and made ONE 4-wave workgroup per (image, strip): 256 workgroups on the 15 x 20 maps, nothing to hide the tap loads
  const int qb = dw_quads_per_block(C4);
  const int cq0 = blockIdx.x * qb;
  const int nq = C4 - cq0 < qb ? C4 - cq0 : qb;
  const int nph = 256 / nq;
  const int tid = threadIdx.x;
  const int ph = tid / nq, q = tid - ph * nq;
  const int img = blockIdx.z, strip = blockIdx.y, oh0 = strip * R;
  const int c = 4 * (cq0 + q);
  floatx4 psum = {0.f, 0.f, 0.f, 0.f};
  if (WLDS) {
    for (int i = tid; i < K * K * nq; i += 256) {
      const int t = i / nq, qq = i - t * nq;
      wl[t * 64 + qq] = *reinterpret_cast<const floatx4*>(a.w + t * a.C + 4 * (cq0 + qq));
    }
    __syncthreads();
  }
  if (ph < nph) {
    floatx4 w[WLDS ? 1 : K * K];
    if (!WLDS) {
#pragma unroll
      for (int t = 0; t < K * K; ++t) w[t] = *reinterpret_cast<const floatx4*>(a.w + t * a.C + c);
    }
    const floatx4 bias = *reinterpret_cast<const floatx4*>(a.bias + c);
    const int ih0 = oh0 * S - a.pad_t;
    // the image through a buffer resource: a tap is a 32-bit offset (row base + dx C), a tap outside the image an out-of-bounds
    // offset that reads zeros -- the 64-bit address of every tap was 9 VALU instructions beside its 16 FMAs (round 5)
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (int64_t)img * a.H * a.W * a.C), 0,
                                                                         a.H * a.W * a.C * 4, 0x00020000);
    constexpr unsigned kOob = 0xFFFFFFF0u;
    const unsigned c_b = 4u * (unsigned)c, C_b = 4u * (unsigned)a.C;
    float* const yimg = a.y + (int64_t)img * a.Ho * a.Wo * a.C + c;
    for (int ow = ph; ow < a.Wo; ow += nph) {
      floatx4 acc[R];
#pragma unroll
      for (int o = 0; o < R; ++o) acc[o] = bias;
      const int iw0 = ow * S - a.pad_l;
      if constexpr (WLDS) {
        // rolled over the input rows (tap row dy = r - o S is a wave-uniform run-time value: a scalar branch, the taps indexed in
        // LDS): five loads in flight per wave instead of (R - 1) S + K rows of them, ~70 registers, eight waves per SIMD
#pragma unroll 4
        for (int r = 0; r < NR; ++r) {
          const int ih = ih0 + r;
          const bool rok = (unsigned)ih < (unsigned)a.H;
          const unsigned row_b = (unsigned)(ih * a.W + iw0) * C_b + c_b;  // (wraps for taps left of / above the image: those are masked)
          floatx4 xv[K];
#pragma unroll
          for (int dx = 0; dx < K; ++dx) {
            const int iw = iw0 + dx;
            const bool ok = rok && (unsigned)iw < (unsigned)a.W;
            xv[dx] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(ok ? row_b + (unsigned)dx * C_b : kOob), 0, 0));
          }
#pragma unroll
          for (int o = 0; o < R; ++o) {
            const int dy = r - o * S;
            if (dy >= 0 && dy < K) {
#pragma unroll
              for (int dx = 0; dx < K; ++dx) {
                const floatx4 wv = wl[(dy * K + dx) * 64 + q];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(xv[dx][e], wv[e], acc[o][e]);
              }
            }
          }
        }
      } else
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int ih = ih0 + r;
        const bool rok = (unsigned)ih < (unsigned)a.H;
        const unsigned row_b = (unsigned)(ih * a.W + iw0) * C_b + c_b;
        floatx4 xv[K];
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
          const int iw = iw0 + dx;
          const bool ok = rok && (unsigned)iw < (unsigned)a.W;
          xv[dx] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(ok ? row_b + (unsigned)dx * C_b : kOob), 0, 0));
        }
#pragma unroll
        for (int o = 0; o < R; ++o) {
          const int dy = r - o * S;  // compile-time after unrolling
          if (dy >= 0 && dy < K) {
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
              const floatx4 wv = w[WLDS ? 0 : dy * K + dx];
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(xv[dx][e], wv[e], acc[o][e]);
            }
          }
        }
      }
#pragma unroll
      for (int o = 0; o < R; ++o) {
        if (oh0 + o < a.Ho) {
          const floatx4 v = swish4(acc[o]);
          *reinterpret_cast<floatx4*>(yimg + ((int64_t)(oh0 + o) * a.Wo + ow) * a.C) = v;
          psum += v;
        }
      }
    }
  }
  red[tid] = psum;
  __syncthreads();
  if (tid < nq && a.pool_partial) {
    floatx4 sum = red[tid];
    for (int p = 1; p < nph; ++p) sum += red[p * nq + tid];
    *reinterpret_cast<floatx4*>(a.pool_partial + ((int64_t)img * gridDim.y + strip) * a.C + 4 * (cq0 + tid)) = sum;
  }
}

// sums over the pixels of each (image, channel) in kSeStrips strips: block = 64 channels x 4 pixel
// lanes of one strip, fixed summation order (pixel lanes stride the strip, then a 4-way LDS sum; the
// strips are added in order by se_gate_kernel) -> bitwise reproducible, and enough blocks to fill
// the chip on the 120x160 maps
constexpr int kSeStrips = 32;

__global__ __launch_bounds__(256) void se_pool_kernel(const float* y, float* partial, int HW, int C) {
  __shared__ float part[4][64];
  const int img = blockIdx.y, strip = blockIdx.z, c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  const int per = (HW + kSeStrips - 1) / kSeStrips;
  const int p0 = strip * per, p1 = p0 + per < HW ? p0 + per : HW;
  float s = 0.f;
  if (c < C) {
    const float* base = y + (int64_t)img * HW * C + c;
    for (int p = p0 + pl; p < p1; p += 4) s += base[(int64_t)p * C];
  }
  part[pl][threadIdx.x & 63] = s;
  __syncthreads();
  if (pl == 0 && c < C)
    partial[((int64_t)img * kSeStrips + strip) * C + c] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// squeeze-excitation vector in three small, wide launches (one block per image left most of the
// chip idle and re-read both weight matrices per image: 2 ms of a forward):
//   se_mean_kernel:   pooled[n][c]  = sum of the strip partials / HW            (fixed order)
//   se_reduce_kernel: sq[n][j]      = swish(W1[j] . pooled[n] + b1[j])          one wave per (n, j)
//   se_expand_kernel: gate[n][c]    = sigmoid(W2t[.][c] . sq[n] + b2[c])        W2 transposed: coalesced
// 64 (image, channel) pairs x 4 interleaved groups of partials per workgroup, combined in LDS in a fixed order (the 150 tile
// partials of the first fused block were a 150-deep dependent chain per lane: 63 us)
__global__ __launch_bounds__(256) void se_mean_kernel(const float* partial, float* pooled, int HW, int C, int total, int n_partials) {
  __shared__ float red[4][64];
  const int idx = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  float s = 0.f;
  if (idx < total) {
    const int img = idx / C, c = idx - img * C;
    for (int st = g; st < n_partials; st += 4) s += partial[((int64_t)img * n_partials + st) * C + c];
  }
  red[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0 && idx < total) pooled[idx] = (((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]) / (float)HW;
}

__global__ __launch_bounds__(256) void se_reduce_kernel(const float* pooled, const float* w1, const float* b1, float* sq, int C,
                                                        int Cse) {
  const int img = blockIdx.y, j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= Cse) return;
  const float* pv = pooled + (int64_t)img * C;
  const float* w = w1 + (int64_t)j * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s = fmaf(w[c], pv[c], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) { s += b1[j]; sq[(int64_t)img * Cse + j] = s / (1.f + __expf(-s)); }
}

__global__ __launch_bounds__(256) void se_expand_kernel(const float* sq, const float* w2t, const float* b2, float* gate, int C,
                                                        int Cse) {
  __shared__ float sv[128];
  const int img = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < Cse) sv[threadIdx.x] = sq[(int64_t)img * Cse + threadIdx.x];
  __syncthreads();
  if (c >= C) return;
  float s = b2[c];
  for (int j = 0; j < Cse; ++j) s = fmaf(w2t[(int64_t)j * C + c], sv[j], s);
  gate[(int64_t)img * C + c] = 1.f / (1.f + __expf(-s));
}

}  // namespace

int dwconv_pool_strips(int Ho, int k, int stride) { return (Ho + dw_rows(Ho, k, stride) - 1) / dw_rows(Ho, k, stride); }

// a.pool_partial (or null) receives [n][dwconv_pool_strips(Ho)][C] sums of the outputs (strip kernel only)
int launch_dwconv(const DwArgs& a, hipStream_t stream) {
  if ((a.stride == 1 || a.stride == 2) && (a.k == 3 || a.k == 5) && a.n < 65536) {
    const int qb = dw_quads_per_block(a.C / 4);
    const dim3 grid((unsigned)((a.C / 4 + qb - 1) / qb), (unsigned)dwconv_pool_strips(a.Ho, a.k, a.stride), (unsigned)a.n);
#define HP_DW(K_, S_)                                                                                                        \
  if (a.k == K_ && a.stride == S_) {                                                                                        \
    if (dw_rows(a.Ho, K_, S_) == 8) hipLaunchKernelGGL((dwconv_strip_kernel<K_, S_, 8, K_ == 5>), grid, dim3(256), 0, stream, a);   \
    else hipLaunchKernelGGL((dwconv_strip_kernel<K_, S_, 4, K_ == 5>), grid, dim3(256), 0, stream, a);                      \
  }
    HP_DW(3, 1) HP_DW(3, 2) HP_DW(5, 1) HP_DW(5, 2)
#undef HP_DW
    return check_launch("dwconv_strip_kernel");
  }
  if (a.pool_partial) return fail(HP_ERR_ARG, "dwconv: pooled sums need the strip kernel");
  const int64_t total = (int64_t)a.n * a.Ho * a.Wo * (a.C / 4);
  const dim3 grid((unsigned)((total + 255) / 256));
  if (a.k == 3) hipLaunchKernelGGL(dwconv_swish_nhwc<3>, grid, dim3(256), 0, stream, a);
  else if (a.k == 5) hipLaunchKernelGGL(dwconv_swish_nhwc<5>, grid, dim3(256), 0, stream, a);
  else return fail(HP_ERR_ARG, "dwconv_swish_nhwc: kernel size must be 3 or 5");
  return check_launch("dwconv_swish_nhwc");
}

bool dwconv_pools(const DwArgs& a) {
  return (a.stride == 1 || a.stride == 2) && (a.k == 3 || a.k == 5) && a.n < 65536;
}

int se_partial_floats(int n, int C) { return n * kSeStrips * C; }

// w2t = the expand weights transposed to [Cse][C]; sq = workspace [n][Cse].  n_partials > 0: `partial` already holds
// [n][n_partials][C] sums written by the depthwise kernel (no pooling pass); 0: pool y here.
int launch_se(const float* y, float* partial, float* pooled, float* sq, float* gate, const float* w1, const float* b1,
              const float* w2t, const float* b2, int n, int HW, int C, int Cse, int n_partials, hipStream_t stream) {
  if (Cse > 128) return fail(HP_ERR_ARG, "squeeze-excitation: more than 128 squeezed channels");
  int rc;
  if (n_partials <= 0) {
    hipLaunchKernelGGL(se_pool_kernel, dim3((C + 63) / 64, n, kSeStrips), dim3(256), 0, stream, y, partial, HW, C);
    if ((rc = check_launch("se_pool_kernel"))) return rc;
    n_partials = kSeStrips;
  }
  hipLaunchKernelGGL(se_mean_kernel, dim3((n * C + 63) / 64), dim3(256), 0, stream, partial, pooled, HW, C, n * C, n_partials);
  if ((rc = check_launch("se_mean_kernel"))) return rc;
  hipLaunchKernelGGL(se_reduce_kernel, dim3((Cse + 3) / 4, n), dim3(256), 0, stream, pooled, w1, b1, sq, C, Cse);
  if ((rc = check_launch("se_reduce_kernel"))) return rc;
  hipLaunchKernelGGL(se_expand_kernel, dim3((C + 255) / 256, n), dim3(256), 0, stream, sq, w2t, b2, gate, C, Cse);
  return check_launch("se_expand_kernel");
}

}  // namespace hp
