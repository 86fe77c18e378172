// Crop-and-resize of the observed image around every hypothesis: roi_align semantics of
// torchvision 0.14.1 (aligned=False, spatial_scale=1) as the reference calls it
// (TB/lib3d/cropping.py:155-197, CP/lib3d/cropping.py:129-134), including the RGB-D rule
// and (optionally) the depth normalisation of normalize_images
// (MP/models/pose_rigid.py:455-544) fused into the store.
//
// Gather kernel, HBM/L2 bound: every hypothesis reads the SAME 640x480 frame (3.7-4.9 MB,
// L2/MALL resident), so the traffic that reaches HBM is the output.  One lane = one output
// pixel for all channels; a workgroup is a 16 x 16 pixel tile, a wave 4 rows of 16 columns, so
// source reads of a wave fall into a few adjacent cache lines.
//
// The 4x4 bilinear samples of an output pixel are SEPARABLE: sample (iy, ix) contributes
// (wy_lo[iy] I[y_lo] + wy_hi[iy] I[y_hi]) x (wx_lo[ix] I[x_lo] + wx_hi[ix] I[x_hi]), so the
// pixel is sum_r sum_c Wy[r] Wx[c] I[r][c] over the DISTINCT rows/columns the samples touch.
// Crops up-sample or mildly down-sample the frame (bin <= ~2.5 px), so that is 2-5 rows x 2-5
// columns = 4-25 loads per channel instead of 64.  Pixels whose samples span more than 5
// rows or columns (bin > 2.67 px: crop boxes wider than ~850 px) take the literal 16-sample
// path.  The factorisation only re-associates the fp32 sum (<= 1e-6 relative).
#include <cstdlib>

#include "common.h"
#include "crop_math.h"

namespace hp {

#pragma clang fp contract(off)

struct CropArgs {
  const float* images; int Bi, C, NC, H, W;
  const float* boxes; const int32_t* im_ids; int n, oh, ow, sr;
  void* out; int out_half; hp_strides os;  // fp32, or fp16 when the crop feeds an fp16 network input directly
  const float* depth_norm_z; int depth_norm_mode;
  int full_record8;  // fp32 NHWC destination, records of a multiple of 8 floats, 32-B aligned: the crop owns the first 8 floats
};


// ---- second-generation kernel: one workgroup = a 32 x 64 tile of output pixels.  The 16 x 16 kernel above spent most of
// its ~600 vector instructions per pixel on (a) building the folded axis weights -- 32 lanes work, 224 wait, once per 256
// pixels -- and (b) a fully predicated 5 x 5 x 4 tap loop although a typical crop (bin <= 1 source pixel) folds into 2-3
// rows x 2-3 columns.  Here the folds of 32 rows + 64 columns are built once per 2048 pixels, a thread keeps its column
// fold in registers and walks 8 rows, and the tap loops run to the TILE's largest span (workgroup-uniform bounds), the
// channel count is a template parameter.  Same taps, same order of operations inside a pixel as above (and as
// oracle/csrc/oracle.c up to the separable re-association stated in DESIGN.md).
constexpr int kTR = 32, kTC = 64;

template <int NC>
__global__ __launch_bounds__(256) void crop_tile_kernel(CropArgs a) {
  __shared__ Fold folds_y[kTR], folds_x[kTC];
  __shared__ int span_max[2];
  const int r = blockIdx.z;
  const int tid = threadIdx.x;
  const int row0 = blockIdx.y * kTR, col0 = blockIdx.x * kTC;
  const float* box = a.boxes + 4 * (int64_t)r;
  const float x1 = box[0], y1 = box[1];
  float roi_w = box[2] - x1, roi_h = box[3] - y1;
  roi_w = roi_w < 1.0f ? 1.0f : roi_w;  // aligned=False
  roi_h = roi_h < 1.0f ? 1.0f : roi_h;
  const float bin_h = roi_h / (float)a.oh, bin_w = roi_w / (float)a.ow;
  const int g = a.sr;
  const float count = (float)(g * g);
  const int H = a.H, W = a.W;
  if (tid < 2) span_max[tid] = 0;
  __syncthreads();
  if (tid < kTR + kTC) {  // lanes 0-31: the tile's rows, 32-95: its columns
    const bool is_x = tid >= kTR;
    const int k = is_x ? tid - kTR : tid;
    Axis t;
    Fold f;
    if (is_x) make_axis(x1, col0 + k, bin_w, g, W, t);
    else make_axis(y1, row0 + k, bin_h, g, H, t);
    fold_axis(t, g, f.first, f.span, f.w);
    if (is_x) folds_x[k] = f; else folds_y[k] = f;
    const bool inside = is_x ? col0 + k < a.ow : row0 + k < a.oh;
    if (inside) atomicMax(&span_max[is_x ? 1 : 0], f.span);
  }
  __syncthreads();
  const int nr_max = span_max[0], nc_max = span_max[1];
  const bool separable = nr_max <= kSpan && nc_max <= kSpan;
  const int tx = tid & (kTC - 1), tyg = tid / kTC;  // column of this thread, first of its rows
  const int pw = col0 + tx;
  if (pw >= a.ow) return;
  const Fold fx = folds_x[tx];
  const int c0 = fx.first, nc = fx.span;
  const int im_id = a.im_ids[r];
  const bool bad_id = (unsigned)im_id >= (unsigned)a.Bi;  // reads frame 0, writes zeros (the reference's indexing would raise)
  const float* img = a.images + (int64_t)(bad_id ? 0 : im_id) * a.C * H * W;
  const int HW = H * W;
  const float zn = (NC == 4 && a.depth_norm_mode != 0) ? a.depth_norm_z[r] : 1.0f;
#pragma unroll 1
  for (int k = 0; k < kTR / (256 / kTC); ++k) {
    const int py = tyg + (256 / kTC) * k, ph = row0 + py;
    if (ph >= a.oh) break;
    const Fold fy = folds_y[py];
    const int r0 = fy.first, nr = fy.span;
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
    float vacc = 0.0f;
    if (separable) {
      const int off0 = r0 * W + c0;
#pragma unroll
      for (int i = 0; i < kSpan; ++i) {  // unrolled (weights stay in registers); workgroup-uniform exit at the tile's span
        if (i >= nr_max) break;
        if (i < nr) {
          const int off = off0 + i * W;
          float racc[NC];
#pragma unroll
          for (int c = 0; c < NC; ++c) racc[c] = 0.f;
          float rv = 0.0f;
#pragma unroll
          for (int j = 0; j < kSpan; ++j) {
            if (j >= nc_max) break;
            if (j < nc) {
              const float wj = fx.w[j];
#pragma unroll
              for (int c = 0; c < NC; ++c) {
                const float v = img[c * HW + off + j];
                racc[c] += wj * v;
                if (c == 3) rv += wj * (v > 0.0f ? 1.0f : 0.0f);
              }
            }
          }
          const float wi = fy.w[i];
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[c] += wi * racc[c];
          vacc += wi * rv;
        }
      }
    } else {
#pragma unroll  // unrolled: a runtime channel index would put acc[] in scratch memory
      for (int c = 0; c < NC; ++c) {
        float sacc = 0.0f, sv = 0.0f;
        slow_pixel(img + (int64_t)c * HW, H, W, y1, x1, ph, pw, bin_h, bin_w, g, c == 3, sacc, sv);
        acc[c] = sacc;
        if (c == 3) vacc = sv;
      }
    }
    float outv[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float val = acc[c] / count;
      if (c == 3) {
        if (vacc / count < 0.99f) val = 0.0f;  // TB/lib3d/cropping.py:184-195
        if (a.depth_norm_mode == 1) val = val / zn;
        else if (a.depth_norm_mode == 2) val = fminf(fmaxf(val / zn, 0.0f), 2.0f) - 1.0f;
        else if (a.depth_norm_mode == 3) val = fminf(fmaxf(val - zn, -2.0f), 2.0f);
      }
      outv[c] = bad_id ? 0.0f : val;
    }
    const int64_t obase = (int64_t)r * a.os.s_item + (int64_t)ph * a.os.s_row + (int64_t)pw * a.os.s_col;
    typedef float float3v __attribute__((ext_vector_type(3)));
    if (a.out_half && a.os.s_chan == 1 && a.full_record8) {
      // fp16 network input (16-half records = one 32-B sector): same ownership rule, sixteen halves in two 16-B stores
      typedef _Float16 halfx8v __attribute__((ext_vector_type(8)));
      _Float16* const o = reinterpret_cast<_Float16*>(a.out) + obase;
      const _Float16 z = (_Float16)0.f;
      *reinterpret_cast<halfx8v*>(o) = halfx8v{(_Float16)outv[0], (_Float16)outv[1], (_Float16)outv[2], NC == 4 ? (_Float16)outv[NC - 1] : z, z, z, z, z};
      *reinterpret_cast<halfx8v*>(o + 8) = halfx8v{z, z, z, z, z, z, z, z};
    } else if (a.out_half) {  // strides count fp16 elements
      _Float16* const o = reinterpret_cast<_Float16*>(a.out);
#pragma unroll
      for (int c = 0; c < NC; ++c) o[obase + (int64_t)c * a.os.s_chan] = (_Float16)outv[c];
    } else if (a.os.s_chan == 1 && a.full_record8) {
      // The crop runs first and owns the first 8 floats = the first 32-B sector of every pixel record (CosyPose: the whole
      // record; MegaPose RGB-D: its four channels + the head of view 0's render): it writes the WHOLE sector (its channels +
      // zeros; the rasteriser overwrites the rest afterwards, pads stay zero) -- a full-sector store instead of 12 / 16 B
      // of it (a partial-sector write costs the memory system a read-modify-write)
      typedef float float4v __attribute__((ext_vector_type(4)));
      float* const o = reinterpret_cast<float*>(a.out) + obase;
      *reinterpret_cast<float4v*>(o) = float4v{outv[0], outv[1], outv[2], NC == 4 ? outv[NC - 1] : 0.f};
      *reinterpret_cast<float4v*>(o + 4) = float4v{0.f, 0.f, 0.f, 0.f};
    } else if (a.os.s_chan == 1 && NC == 3) {
      *reinterpret_cast<float3v*>(reinterpret_cast<float*>(a.out) + obase) = float3v{outv[0], outv[1], outv[2]};
    } else {
      float* const o = reinterpret_cast<float*>(a.out);
#pragma unroll
      for (int c = 0; c < NC; ++c) o[obase + (int64_t)c * a.os.s_chan] = outv[c];
    }
  }
}

}  // namespace hp

namespace hp {
static int crop_launch(const float* d_images, int Bi, int C, int n_channels, int H, int W, const float* d_boxes,
                       const int32_t* d_im_ids, int n, int out_h, int out_w, int sampling_ratio, void* d_out, int out_half,
                       const hp_strides* out_strides, const float* d_depth_norm_z, int depth_norm_mode, void* stream) {
  HP_REQUIRE(d_images && out_strides, "hp_crop_roi_align: null pointer");
  HP_REQUIRE(C == 3 || C == 4, "hp_crop_roi_align: images must have 3 (rgb) or 4 (rgbd) channels");
  HP_REQUIRE((n_channels == 3 || n_channels == 4) && n_channels <= C, "hp_crop_roi_align: n_channels must be 3 or 4 and <= C");
  HP_REQUIRE(sampling_ratio >= 1 && sampling_ratio <= kMaxSR, "hp_crop_roi_align: sampling_ratio must be 1..4");
  HP_REQUIRE(n >= 0 && out_h > 0 && out_w > 0 && H > 0 && W > 0 && Bi > 0, "hp_crop_roi_align: bad sizes");
  const int full_record = depth_norm_mode & HP_CROP_FULL_RECORD8;
  depth_norm_mode &= ~HP_CROP_FULL_RECORD8;
  HP_REQUIRE(depth_norm_mode >= 0 && depth_norm_mode <= 3, "hp_crop_roi_align: bad depth_norm_mode");
  HP_REQUIRE(depth_norm_mode == 0 || d_depth_norm_z, "hp_crop_roi_align: depth_norm_z missing");
  if (n == 0) return HP_OK;
  HP_REQUIRE(d_boxes && d_im_ids && d_out, "hp_crop_roi_align: null pointer");
  {
    const int unit = out_half ? 16 : 8;  // elements per 32-B sector
    HP_REQUIRE(!full_record || (out_strides->s_chan == 1 && out_strides->s_col >= unit && out_strides->s_col % unit == 0 &&
                                out_strides->s_row % unit == 0 && out_strides->s_item % unit == 0 &&
                                (reinterpret_cast<uintptr_t>(d_out) & 31) == 0),
               "hp_crop_roi_align: HP_CROP_FULL_RECORD8 needs an NHWC destination whose pixel records are multiples of 32 B, 32-B aligned");
  }
  CropArgs a{d_images, Bi, C, n_channels, H, W, d_boxes, d_im_ids, n, out_h, out_w, sampling_ratio,
             d_out, out_half, *out_strides, d_depth_norm_z, depth_norm_mode, full_record ? 1 : 0};
  dim3 grid((out_w + kTC - 1) / kTC, (out_h + kTR - 1) / kTR, n);
  if (n_channels == 3) hipLaunchKernelGGL(crop_tile_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(crop_tile_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("crop_tile_kernel");
}
}  // namespace hp

extern "C" int hp_crop_roi_align(const float* d_images, int Bi, int C, int n_channels, int H, int W,
                                 const float* d_boxes, const int32_t* d_im_ids, int n, int out_h,
                                 int out_w, int sampling_ratio, float* d_out,
                                 const hp_strides* out_strides, const float* d_depth_norm_z,
                                 int depth_norm_mode, void* stream) {
  return hp::crop_launch(d_images, Bi, C, n_channels, H, W, d_boxes, d_im_ids, n, out_h, out_w, sampling_ratio, d_out, 0,
                         out_strides, d_depth_norm_z, depth_norm_mode, stream);
}

extern "C" int hp_crop_roi_align_f16(const float* d_images, int Bi, int C, int n_channels, int H, int W,
                                     const float* d_boxes, const int32_t* d_im_ids, int n, int out_h,
                                     int out_w, int sampling_ratio, void* d_out_f16,
                                     const hp_strides* out_strides, const float* d_depth_norm_z,
                                     int depth_norm_mode, void* stream) {
  return hp::crop_launch(d_images, Bi, C, n_channels, H, W, d_boxes, d_im_ids, n, out_h, out_w, sampling_ratio, d_out_f16, 1,
                         out_strides, d_depth_norm_z, depth_norm_mode, stream);
}
