// torchvision roi_align (0.14.1, aligned=False) sample placement, boundary rules and the separable fold of the sampling_ratio^2
// bilinear samples of an output pixel -- shared by the stand-alone crop kernels (crop.hip) and the fused render + crop
// kernel (raster.hip).  Reference: TB/lib3d/cropping.py:155-197 -> torchvision.ops.roi_align.
#pragma once

#include "common.h"

namespace hp {

#pragma clang fp contract(off)

constexpr int kMaxSR = 4;
constexpr int kSpan = 5;  // distinct rows / columns handled by the separable path

struct Axis {
  int lo[kMaxSR], hi[kMaxSR];
  float w0[kMaxSR], w1[kMaxSR];
  bool valid[kMaxSR];
};

// torchvision's sample placement + boundary rules along one axis (size = H or W)
__device__ __forceinline__ void make_axis(float start, int p, float bin, int g, int size, Axis& ax) {
#pragma unroll
  for (int s = 0; s < kMaxSR; ++s) {
    if (s < g) {
      float y = start + (float)p * bin + ((float)s + 0.5f) * bin / (float)g;
      ax.valid[s] = !(y < -1.0f || y > (float)size);
      if (y <= 0.0f) y = 0.0f;
      int y_low = (int)y, y_high;
      if (y_low >= size - 1) { y_high = y_low = size - 1; y = (float)y_low; } else { y_high = y_low + 1; }
      const float l = y - (float)y_low;
      ax.lo[s] = y_low; ax.hi[s] = y_high; ax.w1[s] = l; ax.w0[s] = 1.0f - l;
    } else {
      ax.valid[s] = false; ax.lo[s] = ax.hi[s] = 0; ax.w0[s] = ax.w1[s] = 0.0f;
    }
  }
}

// first index touched by the valid samples, number of indices, and the per-index summed weights
__device__ __forceinline__ void fold_axis(const Axis& ax, int g, int& first, int& span, float (&wsum)[kSpan]) {
  first = 1 << 30;
  int last = -1;
#pragma unroll
  for (int s = 0; s < kMaxSR; ++s)
    if (s < g && ax.valid[s]) { first = min(first, ax.lo[s]); last = max(last, ax.hi[s]); }
  span = last >= first ? last - first + 1 : 0;
  if (span == 0) first = 0;
#pragma unroll
  for (int k = 0; k < kSpan; ++k) {
    float w = 0.0f;
#pragma unroll
    for (int s = 0; s < kMaxSR; ++s) {
      if (s < g && ax.valid[s]) {
        w += (ax.lo[s] - first == k) ? ax.w0[s] : 0.0f;
        w += (ax.hi[s] - first == k) ? ax.w1[s] : 0.0f;
      }
    }
    wsum[k] = w;
  }
}

// Literal 16-sample evaluation (torchvision's loop) for strongly down-sampling crops; rolled loops so that it does
// not cost the common path registers.  Inlined: a real call needs a stack, and a kernel with scratch memory cannot be
// replayed from a captured hipGraph on this ROCm (second replay faults) -- see happypose_amd/graphs.py.
__device__ __forceinline__ void slow_pixel(const float* plane, int H, int W, float y1, float x1, int ph, int pw,
                                        float bin_h, float bin_w, int g, bool want_valid, float& acc, float& vacc) {
#pragma unroll 1
  for (int iy = 0; iy < g; ++iy) {
    float y = y1 + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)g;
    if (y < -1.0f || y > (float)H) continue;
    if (y <= 0.0f) y = 0.0f;
    int yl = (int)y, yh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
    const float ly = y - (float)yl, hy = 1.0f - ly;
#pragma unroll 1
    for (int ix = 0; ix < g; ++ix) {
      float x = x1 + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)g;
      if (x < -1.0f || x > (float)W) continue;
      if (x <= 0.0f) x = 0.0f;
      int xl = (int)x, xh;
      if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
      const float lx = x - (float)xl, hx = 1.0f - lx;
      const float v00 = plane[yl * W + xl], v01 = plane[yl * W + xh];
      const float v10 = plane[yh * W + xl], v11 = plane[yh * W + xh];
      const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
      acc += w1 * v00 + w2 * v01 + w3 * v10 + w4 * v11;
      if (want_valid)  // validity mask (depth > 0) through the same interpolation
        vacc += w1 * (v00 > 0.0f ? 1.0f : 0.0f) + w2 * (v01 > 0.0f ? 1.0f : 0.0f) +
                w3 * (v10 > 0.0f ? 1.0f : 0.0f) + w4 * (v11 > 0.0f ? 1.0f : 0.0f);
    }
  }
}

// folded weights of one output row (or column) of a crop
struct Fold { int first, span; float w[kSpan]; };

}  // namespace hp
