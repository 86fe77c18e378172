// Memory-bound companions of the conv stack: 3x3/2 max pooling (NHWC) and the network head
// (spatial mean -> optional 512x512 fc -> pose / logits linear heads).
// Reference: MP/models/torchvision_resnet.py:325-341 (maxpool, avgpool, fc),
// MP/models/wide_resnet.py:120-129, MP/models/pose_rigid.py:352-374 (heads).
#include "conv.h"

namespace hp {

// ---- 3x3 stride-2 pad-1 max pooling, NHWC, 4 channels per lane ------------------------
// grid.y = output row (img * Ho + oh), threads over (ow, channel quad): the index decode is one
// multiply-high per thread -- with a flat 64-bit index it was three 64-bit divisions (~600 instructions)
// for nine loads, which made this bandwidth kernel instruction-bound.
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc(const float* x, float* y, int H, int W, int C, int Ho, int Wo,
                                                         FastDiv fd_c4, FastDiv fd_ho) {
  const int C4 = C / 4;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Wo * C4) return;
  const int ow = fdiv(t, fd_c4), c4 = t - ow * C4;
  const int row = blockIdx.y, img = fdiv(row, fd_ho), oh = row - img * Ho;
  const float* xin = x + (int64_t)img * H * W * C + 4 * c4;
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int ih = oh * 2 - 1 + dy;
    if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int iw = ow * 2 - 1 + dx;
      if ((unsigned)iw >= (unsigned)W) continue;
      const float4 v = *reinterpret_cast<const float4*>(xin + (ih * W + iw) * C);
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  }
  *reinterpret_cast<float4*>(y + ((int64_t)row * Wo + ow) * C + 4 * c4) = m;
}

// ---- nearest-neighbour resize (F.interpolate(size=..., mode="nearest"): src = min(floor(dst * in / out), in - 1)) and
// the stride-2 subsampling max_pool2d(x, 1, 2, 0) of the FPN (torchvision ops/feature_pyramid_network.py:
// FeaturePyramidNetwork.forward top-down pathway, LastLevelMaxPool), NHWC, 4 channels per lane
__global__ __launch_bounds__(256) void resize_nearest_nhwc(const float* x, float* y, int H, int W, int C, int Ho, int Wo,
                                                           int stride_mode, FastDiv fd_c4, FastDiv fd_ho) {
  const int C4 = C / 4;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Wo * C4) return;
  const int ow = fdiv(t, fd_c4), c4 = t - ow * C4;
  const int row = blockIdx.y, img = fdiv(row, fd_ho), oh = row - img * Ho;
  int ih, iw;
  if (stride_mode) { ih = 2 * oh; iw = 2 * ow; }
  else {
    ih = (int)floorf((float)oh * ((float)H / (float)Ho)); ih = ih < H - 1 ? ih : H - 1;
    iw = (int)floorf((float)ow * ((float)W / (float)Wo)); iw = iw < W - 1 ? iw : W - 1;
  }
  *reinterpret_cast<float4*>(y + ((int64_t)row * Wo + ow) * C + 4 * c4) =
      *reinterpret_cast<const float4*>(x + (((int64_t)img * H + ih) * W + iw) * C + 4 * c4);
}

// ---- detector pre-processing with GeneralizedRCNNTransform.resize: bilinear, align_corners = False, the scale
// recomputed from the sizes (torch upsample_bilinear2d: src = (dst + 0.5) * in / out - 0.5, clamped at 0; the neighbour
// index clamped at in - 1), then (v - mean) / std; pixels of the padded canvas outside [0, ho) x [0, wo) are zero
// (batch_images pads the NORMALISED image with zeros).  torchvision normalises first and interpolates after; both are
// affine per channel, so the order only moves the rounding.
__global__ __launch_bounds__(256) void normalize_resize_nchw_to_nhwc4(const float* x, float* y, int n, int hi, int wi, int ho, int wo,
                                                                      int hp, int wp, float m0, float m1, float m2, float s0,
                                                                      float s1, float s2) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)n * hp * wp;
  if (t >= total) return;
  const int px = (int)(t % wp);
  const int64_t r = t / wp;
  const int py = (int)(r % hp), img = (int)(r / hp);
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (py < ho && px < wo) {
    const float sy = (float)hi / (float)ho, sx = (float)wi / (float)wo;
    float fy = sy * ((float)py + 0.5f) - 0.5f, fx = sx * ((float)px + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const int64_t plane = (int64_t)hi * wi;
    const float* src = x + (int64_t)img * 3 * plane;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* pc = src + c * plane;
      v[c] = hy * (hx * pc[(int64_t)y0 * wi + x0] + lx * pc[(int64_t)y0 * wi + x1]) +
             ly * (hx * pc[(int64_t)y1 * wi + x0] + lx * pc[(int64_t)y1 * wi + x1]);
    }
    o = make_float4((v[0] - m0) / s0, (v[1] - m1) / s1, (v[2] - m2) / s2, 0.f);
  }
  *reinterpret_cast<float4*>(y + 4 * t) = o;
}

// ---- detector pre-processing: GeneralizedRCNNTransform.normalize ((image - mean) / std per channel,
// torchvision models/detection/transform.py) + NCHW [b,3,h,w] -> NHWC [b,h,w,4] (pad channel 0)
__global__ __launch_bounds__(256) void normalize_nchw_to_nhwc4(const float* x, float* y, int64_t hw, int64_t total,
                                                               float m0, float m1, float m2, float s0, float s1, float s2) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int64_t img = t / hw, p = t - img * hw;
  const float* src = x + img * 3 * hw + p;
  *reinterpret_cast<float4*>(y + 4 * t) = make_float4((src[0] - m0) / s0, (src[hw] - m1) / s1, (src[2 * hw] - m2) / s2, 0.f);
}

__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const _Float16* p) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 v = *reinterpret_cast<const h4*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

// spatial mean of x [HW][C] into feat[C] (LDS); part = LDS scratch [2048].
// channel quads x G groups of positions, 4 independent loads in flight per lane: the mean is a chain of HW dependent
// loads otherwise (47 us per launch for 10 MB).  Partial sums meet in LDS in a fixed order (deterministic).
template <typename T>
__device__ __forceinline__ void head_pool_t(const T* x, int HW, int C, float* feat, float* part) {
  const int tid = threadIdx.x;
  if ((C & 3) == 0) {
    const int nq = C >> 2;
    const int G = nq < 256 ? 256 / nq : 1;
    for (int q = tid; q < nq * G; q += 256) {
      const int g = q / nq, c4 = q - g * nq;
      float4 s[4] = {};
      int p = g;
      for (; p + 3 * G < HW; p += 4 * G) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 v = load4(x + (int64_t)(p + u * G) * C + 4 * c4);
          s[u].x += v.x; s[u].y += v.y; s[u].z += v.z; s[u].w += v.w;
        }
      }
      for (; p < HW; p += G) {
        const float4 v = load4(x + (int64_t)p * C + 4 * c4);
        s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
      }
      float* dst = part + g * C + 4 * c4;
      dst[0] = (s[0].x + s[1].x) + (s[2].x + s[3].x); dst[1] = (s[0].y + s[1].y) + (s[2].y + s[3].y);
      dst[2] = (s[0].z + s[1].z) + (s[2].z + s[3].z); dst[3] = (s[0].w + s[1].w) + (s[2].w + s[3].w);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      float s = part[c];
      for (int g = 1; g < G; ++g) s += part[g * C + c];
      feat[c] = s / (float)HW;
    }
  } else {
    for (int c = tid; c < C; c += 256) {
      float s = 0.f;
      for (int p = 0; p < HW; ++p) s += (float)x[(int64_t)p * C + c];
      feat[c] = s / (float)HW;
    }
  }
}

// ---- head: spatial mean -> [fc 512x512 + bias] -> pose / logits linear heads -----------
// features [HW][C] NHWC.  Without the fc (CosyPose WideResNet, EfficientNet) one workgroup per sample does everything.
// With it (torchvision ResNet, MP/models/torchvision_resnet.py:337-341) the work is three launches: a workgroup per
// sample re-read the 1 MB fc matrix once per sample through four waves -- 450 us per forward, 10 % of a MegaPose
// refiner step -- so the fc runs as one wave per OUTPUT row over all samples (the row stays in registers).
__device__ __forceinline__ void head_pool(const HeadArgs& a, int b, float* feat, float* part) {
  const int tid = threadIdx.x;
  if (a.x_is_half) {
    const _Float16* x = reinterpret_cast<const _Float16*>(a.x) + (int64_t)b * a.HW * a.C;
    head_pool_t(x, a.HW, a.C, feat, part);
  } else {
    const float* x = reinterpret_cast<const float*>(a.x) + (int64_t)b * a.HW * a.C;
    head_pool_t(x, a.HW, a.C, feat, part);
  }
  (void)tid;
}

// linear heads from LDS features f[C]: one wave per output row, lanes stride the features
__device__ __forceinline__ void head_linear(const HeadArgs& a, int b, const float* f) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.features) for (int c = tid; c < a.C; c += 256) a.features[(int64_t)b * a.C + c] = f[c];
  for (int o = wave; o < a.pose_dim + a.n_logits; o += 4) {
    const bool is_pose = o < a.pose_dim;
    const int oo = is_pose ? o : o - a.pose_dim;
    const float* w = (is_pose ? a.pose_w : a.logit_w) + (int64_t)oo * a.C;
    float s = 0.f;
    for (int c = lane; c < a.C; c += 64) s = fmaf(w[c], f[c], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
      if (is_pose) { if (a.pose_out) a.pose_out[(int64_t)b * a.pose_dim + oo] = s + a.pose_b[oo]; }
      else { if (a.logit_out) a.logit_out[(int64_t)b * a.n_logits + oo] = s + a.logit_b[oo]; }
    }
  }
}

// no fc: everything per sample
__global__ __launch_bounds__(256) void head_kernel(HeadArgs a) {
  __shared__ float feat[2048];   // C <= 2048 (512 ResNets, 1536 EfficientNet-b3)
  __shared__ float part[2048];   // partial spatial sums, [G][C] with G * C <= 2048
  head_pool(a, blockIdx.x, feat, part);
  __syncthreads();
  head_linear(a, blockIdx.x, feat);
}

// fc path, launch 1: pooled features -> ws_pool [b][C]
__global__ __launch_bounds__(256) void head_pool_kernel(HeadArgs a) {
  __shared__ float feat[2048];
  __shared__ float part[2048];
  head_pool(a, blockIdx.x, feat, part);
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += 256) a.ws_pool[(int64_t)blockIdx.x * a.C + c] = feat[c];
}

constexpr int kFcSamples = 64;  // samples per workgroup of head_fc_kernel (batch 576: 9 x 128 workgroups instead of 128)

// fc path, launch 2: ws_fc[b][o] = fc_w[o] . ws_pool[b] + fc_b[o]; one wave per output row o (C == 512: 8 weights per
// lane in registers), looping over the samples; the summation order per (b, o) is the lanes' strided partial sums and a
// butterfly, as in the one-launch kernel
__global__ __launch_bounds__(256) void head_fc_kernel(HeadArgs a, int batch) {
  const int lane = threadIdx.x & 63, o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= a.C) return;
  const int b0 = blockIdx.y * kFcSamples, b1 = b0 + kFcSamples < batch ? b0 + kFcSamples : batch;  // samples of this workgroup
  const float* w = a.fc_w + (int64_t)o * a.C;
  float wr[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) wr[j] = w[lane + 64 * j];
  const float bo = a.fc_b[o];
  for (int b = b0; b < b1; ++b) {
    const float* f = a.ws_pool + (int64_t)b * a.C;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s = fmaf(wr[j], f[lane + 64 * j], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) a.ws_fc[(int64_t)b * a.C + o] = s + bo;
  }
}

// fc path, launch 3: linear heads on ws_fc
__global__ __launch_bounds__(256) void head_linear_kernel(HeadArgs a) {
  __shared__ float f[2048];
  for (int c = threadIdx.x; c < a.C; c += 256) f[c] = a.ws_fc[(int64_t)blockIdx.x * a.C + c];
  __syncthreads();
  head_linear(a, blockIdx.x, f);
}

int launch_resize_nearest(const float* x, float* y, int n, int H, int W, int C, int Ho, int Wo, int stride_mode, hipStream_t stream) {
  hipLaunchKernelGGL(resize_nearest_nhwc, dim3((Wo * (C / 4) + 255) / 256, n * Ho), dim3(256), 0, stream, x, y, H, W, C, Ho, Wo,
                     stride_mode, make_fastdiv((unsigned)(C / 4)), make_fastdiv((unsigned)Ho));
  return check_launch("resize_nearest_nhwc");
}

int launch_normalize_nhwc4(const float* x, float* y, int n, int h, int w, const float* mean3, const float* std3, hipStream_t stream) {
  const int64_t hw = (int64_t)h * w, total = hw * n;
  hipLaunchKernelGGL(normalize_nchw_to_nhwc4, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, y, hw, total,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return check_launch("normalize_nchw_to_nhwc4");
}

int launch_normalize_resize_nhwc4(const float* x, float* y, int n, int hi, int wi, int ho, int wo, int hp, int wp, const float* mean3,
                                  const float* std3, hipStream_t stream) {
  const int64_t total = (int64_t)n * hp * wp;
  hipLaunchKernelGGL(normalize_resize_nchw_to_nhwc4, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, y, n, hi, wi,
                     ho, wo, hp, wp, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return check_launch("normalize_resize_nchw_to_nhwc4");
}

int launch_maxpool(const float* x, float* y, int n, int H, int W, int C, int Ho, int Wo, hipStream_t stream) {
  if ((int64_t)n * Ho >= 65536 || (int64_t)H * W * C >= (1ll << 31)) return fail(HP_ERR_ARG, "maxpool3x3s2_nhwc: tensor too large");
  const int per_row = Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_nhwc, dim3((unsigned)((per_row + 255) / 256), (unsigned)(n * Ho)), dim3(256), 0, stream, x, y, H, W,
                     C, Ho, Wo, make_fastdiv((unsigned)(C / 4)), make_fastdiv((unsigned)Ho));
  return check_launch("maxpool3x3s2_nhwc");
}

int launch_head(const HeadArgs& a, int batch, hipStream_t stream) {
  if (a.C > 2048) return fail(HP_ERR_ARG, "head: more than 2048 features");
  if (!a.fc_w) {
    hipLaunchKernelGGL(head_kernel, dim3(batch), dim3(256), 0, stream, a);
    return check_launch("head_kernel");
  }
  if (a.C != 512 || !a.ws_pool || !a.ws_fc) return fail(HP_ERR_ARG, "head: the fc path needs 512 features and its workspaces");
  hipLaunchKernelGGL(head_pool_kernel, dim3(batch), dim3(256), 0, stream, a);
  int rc = check_launch("head_pool_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(head_fc_kernel, dim3(a.C / 4, (batch + kFcSamples - 1) / kFcSamples), dim3(256), 0, stream, a, batch);
  if ((rc = check_launch("head_fc_kernel"))) return rc;
  hipLaunchKernelGGL(head_linear_kernel, dim3(batch), dim3(256), 0, stream, a);
  return check_launch("head_linear_kernel");
}


}  // namespace hp
