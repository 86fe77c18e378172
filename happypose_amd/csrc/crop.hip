// Crop-and-resize of the observed image around every hypothesis: roi_align semantics of
// torchvision 0.14.1 (aligned=False, spatial_scale=1) as the reference calls it
// (TB/lib3d/cropping.py:155-197, CP/lib3d/cropping.py:129-134), including the RGB-D rule
// and (optionally) the depth normalisation of normalize_images
// (MP/models/pose_rigid.py:455-544) fused into the store.
//
// Gather kernel, HBM/L2 bound: every hypothesis reads the SAME 640x480 frame (3.7-4.9 MB,
// L2/MALL resident), so the traffic that reaches HBM is the output.  One lane = one output
// pixel for all channels: the 16 bilinear sample positions/weights are computed once and
// reused per channel; consecutive lanes take consecutive output columns, so source reads
// of a wave fall into a few adjacent cache lines and NCHW stores are coalesced.
#include "common.h"

namespace hp {

#pragma clang fp contract(off)

struct CropArgs {
  const float* images; int Bi, C, NC, H, W;
  const float* boxes; const int32_t* im_ids; int n, oh, ow, sr;
  float* out; hp_strides os;
  const float* depth_norm_z; int depth_norm_mode;
};

constexpr int kMaxSR = 4;

__global__ __launch_bounds__(256) void crop_kernel(CropArgs a) {
  const int r = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.oh * a.ow) return;
  const int ph = p / a.ow, pw = p % a.ow;
  const float* box = a.boxes + 4 * (int64_t)r;
  const float x1 = box[0], y1 = box[1];
  float roi_w = box[2] - x1, roi_h = box[3] - y1;
  roi_w = roi_w < 1.0f ? 1.0f : roi_w;  // aligned=False
  roi_h = roi_h < 1.0f ? 1.0f : roi_h;
  const float bin_h = roi_h / (float)a.oh, bin_w = roi_w / (float)a.ow;
  const int g = a.sr;
  const float count = (float)(g * g);
  const int H = a.H, W = a.W;

  // sample positions: separable in y and x
  int yl[kMaxSR], yh[kMaxSR], xl[kMaxSR], xh[kMaxSR];
  float wy0[kMaxSR], wy1[kMaxSR], wx0[kMaxSR], wx1[kMaxSR];
  bool vy[kMaxSR], vx[kMaxSR];
#pragma unroll
  for (int s = 0; s < kMaxSR; ++s) {
    if (s < g) {
      float y = y1 + (float)ph * bin_h + ((float)s + 0.5f) * bin_h / (float)g;
      float x = x1 + (float)pw * bin_w + ((float)s + 0.5f) * bin_w / (float)g;
      vy[s] = !(y < -1.0f || y > (float)H);
      vx[s] = !(x < -1.0f || x > (float)W);
      if (y <= 0.0f) y = 0.0f;
      if (x <= 0.0f) x = 0.0f;
      int y_low = (int)y, x_low = (int)x, y_high, x_high;
      if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
      if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
      float ly = y - (float)y_low, lx = x - (float)x_low;
      yl[s] = y_low; yh[s] = y_high; xl[s] = x_low; xh[s] = x_high;
      wy1[s] = ly; wy0[s] = 1.0f - ly; wx1[s] = lx; wx0[s] = 1.0f - lx;
    } else {
      vy[s] = vx[s] = false; yl[s] = yh[s] = xl[s] = xh[s] = 0;
      wy0[s] = wy1[s] = wx0[s] = wx1[s] = 0.0f;
    }
  }
  const float* img = a.images + (int64_t)a.im_ids[r] * a.C * H * W;
  const int64_t obase = (int64_t)r * a.os.s_item + (int64_t)ph * a.os.s_row + (int64_t)pw * a.os.s_col;
  float valid_acc = 0.0f;
  for (int c = 0; c < a.NC; ++c) {
    const float* plane = img + (int64_t)c * H * W;
    float acc = 0.0f, vacc = 0.0f;
#pragma unroll
    for (int iy = 0; iy < kMaxSR; ++iy) {
#pragma unroll
      for (int ix = 0; ix < kMaxSR; ++ix) {
        if (iy < g && ix < g && vy[iy] && vx[ix]) {
          const float v00 = plane[yl[iy] * W + xl[ix]], v01 = plane[yl[iy] * W + xh[ix]];
          const float v10 = plane[yh[iy] * W + xl[ix]], v11 = plane[yh[iy] * W + xh[ix]];
          const float w1 = wy0[iy] * wx0[ix], w2 = wy0[iy] * wx1[ix];
          const float w3 = wy1[iy] * wx0[ix], w4 = wy1[iy] * wx1[ix];
          acc += w1 * v00 + w2 * v01 + w3 * v10 + w4 * v11;
          if (c == 3)  // validity mask (depth > 0) through the same interpolation
            vacc += w1 * (v00 > 0.0f ? 1.0f : 0.0f) + w2 * (v01 > 0.0f ? 1.0f : 0.0f) +
                    w3 * (v10 > 0.0f ? 1.0f : 0.0f) + w4 * (v11 > 0.0f ? 1.0f : 0.0f);
        }
      }
    }
    float val = acc / count;
    if (c == 3) {
      valid_acc = vacc / count;
      if (valid_acc < 0.99f) val = 0.0f;  // TB/lib3d/cropping.py:184-195
      if (a.depth_norm_mode != 0) {
        const float zn = a.depth_norm_z[r];
        if (a.depth_norm_mode == 1) val = val / zn;
        else if (a.depth_norm_mode == 2) val = fminf(fmaxf(val / zn, 0.0f), 2.0f) - 1.0f;
        else val = fminf(fmaxf(val - zn, -2.0f), 2.0f);
      }
    }
    a.out[obase + (int64_t)c * a.os.s_chan] = val;
  }
}

}  // namespace hp

extern "C" int hp_crop_roi_align(const float* d_images, int Bi, int C, int n_channels, int H, int W,
                                 const float* d_boxes, const int32_t* d_im_ids, int n, int out_h,
                                 int out_w, int sampling_ratio, float* d_out,
                                 const hp_strides* out_strides, const float* d_depth_norm_z,
                                 int depth_norm_mode, void* stream) {
  using namespace hp;
  HP_REQUIRE(d_images && out_strides, "hp_crop_roi_align: null pointer");
  HP_REQUIRE(C == 3 || C == 4, "hp_crop_roi_align: images must have 3 (rgb) or 4 (rgbd) channels");
  HP_REQUIRE((n_channels == 3 || n_channels == 4) && n_channels <= C, "hp_crop_roi_align: n_channels must be 3 or 4 and <= C");
  HP_REQUIRE(sampling_ratio >= 1 && sampling_ratio <= kMaxSR, "hp_crop_roi_align: sampling_ratio must be 1..4");
  HP_REQUIRE(n >= 0 && out_h > 0 && out_w > 0 && H > 0 && W > 0 && Bi > 0, "hp_crop_roi_align: bad sizes");
  HP_REQUIRE(depth_norm_mode >= 0 && depth_norm_mode <= 3, "hp_crop_roi_align: bad depth_norm_mode");
  HP_REQUIRE(depth_norm_mode == 0 || d_depth_norm_z, "hp_crop_roi_align: depth_norm_z missing");
  if (n == 0) return HP_OK;
  HP_REQUIRE(d_boxes && d_im_ids && d_out, "hp_crop_roi_align: null pointer");
  CropArgs a{d_images, Bi, C, n_channels, H, W, d_boxes, d_im_ids, n, out_h, out_w, sampling_ratio,
             d_out, *out_strides, d_depth_norm_z, depth_norm_mode};
  dim3 grid((out_h * out_w + 255) / 256, n);
  hipLaunchKernelGGL(crop_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("crop_kernel");
}
