// Detector stages between the dense networks: the parts of torchvision's Mask-RCNN inference path
// (the reference's DetectorMaskRCNN, MP/models/mask_rcnn.py:22-42) that are not convolutions.
//
//   hp_rpn_decode        AnchorGenerator.grid_anchors + BoxCoder.decode (weights 1,1,1,1; dw / dh clamped at
//                        log(1000 / 16)) + clip_boxes_to_image + remove_small_boxes + sigmoid for the anchors a level's
//                        top-k selected (models/detection/anchor_utils.py, _utils.py, rpn.py: filter_proposals)
//   hp_nms               batched NMS (ops/boxes.py: batched_nms / nms): boxes sorted by decreasing score, suppression
//                        inside a group (FPN level or class) when IoU > threshold.  The pairwise masks are built on the
//                        device, the greedy pass over them runs on the host (as torchvision's own CUDA nms does): the
//                        call SYNCHRONISES the stream
//   hp_roi_align_levels  MultiScaleRoIAlign (ops/poolers.py): LevelMapper (k = floor(4 + log2(sqrt(area) / 224) + 1e-6)
//                        clamped to the pyramid) + roi_align(aligned=False, sampling_ratio) on that level's NHWC map
//   hp_box_postprocess   softmax over classes + BoxCoder.decode (weights 10,10,5,5) per class + clip
//                        (models/detection/roi_heads.py: postprocess_detections, first half)
//   hp_paste_masks       maskrcnn_inference (sigmoid of the predicted class's channel) + paste_masks_in_image
//                        (roi_heads.py: expand_masks / expand_boxes with padding 1, bilinear F.interpolate to the box,
//                        paste) from the mask head's [n][14][14 x 2 x 2][C] layout
// Everything is fp32; layouts are the NHWC maps hp_net_copy_feature_map hands out.
#include <cmath>
#include <vector>

#include "common.h"

namespace hp {
namespace {

#pragma clang fp contract(off)

constexpr float kXformClip = 4.135166556742356f;  // log(1000 / 16)

struct RpnArgs {
  const float* obj;      // [n] objectness logits of the selected anchors
  const int32_t* idx;    // [n] anchor index inside the level: (y * w + x) * A + a
  const float* deltas;   // the level's map [h][w][4 A] of ONE image
  float base[12];        // A = 3 base anchors (x1, y1, x2, y2), rounded as torchvision does
  int n, w, A, stride_h, stride_w;
  float im_h, im_w, min_size;
  float* boxes;          // [n][4]
  float* scores;         // [n]
  uint8_t* valid;        // [n]
};

__global__ __launch_bounds__(256) void rpn_decode_kernel(RpnArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int id = a.idx[i];
  const int an = id % a.A, pos = id / a.A, y = pos / a.w, x = pos - y * a.w;
  const float sx = (float)(x * a.stride_w), sy = (float)(y * a.stride_h);
  const float ax1 = sx + a.base[4 * an], ay1 = sy + a.base[4 * an + 1], ax2 = sx + a.base[4 * an + 2], ay2 = sy + a.base[4 * an + 3];
  const float* d = a.deltas + (int64_t)pos * 4 * a.A + 4 * an;
  const float wd = ax2 - ax1, ht = ay2 - ay1;
  const float cx = ax1 + 0.5f * wd, cy = ay1 + 0.5f * ht;
  const float dx = d[0], dy = d[1];
  const float dw = fminf(d[2], kXformClip), dh = fminf(d[3], kXformClip);
  const float pcx = dx * wd + cx, pcy = dy * ht + cy;
  const float pw = expf(dw) * wd, ph = expf(dh) * ht;
  float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
  x1 = fminf(fmaxf(x1, 0.f), a.im_w); x2 = fminf(fmaxf(x2, 0.f), a.im_w);
  y1 = fminf(fmaxf(y1, 0.f), a.im_h); y2 = fminf(fmaxf(y2, 0.f), a.im_h);
  a.boxes[4 * i] = x1; a.boxes[4 * i + 1] = y1; a.boxes[4 * i + 2] = x2; a.boxes[4 * i + 3] = y2;
  a.scores[i] = 1.f / (1.f + expf(-a.obj[i]));
  a.valid[i] = (x2 - x1 >= a.min_size && y2 - y1 >= a.min_size) ? 1 : 0;
}

// pairwise suppression masks: bit j of mask[i][jb] = box 64 jb + j is suppressed by box i (j > i, same group, IoU > thr)
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* boxes, const int32_t* group, int n, float thr, unsigned long long* mask,
                                                      int words) {
  const int ib = blockIdx.y, jb = blockIdx.x;
  if (jb < ib) return;
  __shared__ float jbx[64][4];
  __shared__ int jg[64];
  const int t = threadIdx.x;
  const int j0 = jb * 64;
  if (j0 + t < n) {
    jbx[t][0] = boxes[4 * (j0 + t)]; jbx[t][1] = boxes[4 * (j0 + t) + 1]; jbx[t][2] = boxes[4 * (j0 + t) + 2]; jbx[t][3] = boxes[4 * (j0 + t) + 3];
    jg[t] = group[j0 + t];
  }
  __syncthreads();
  const int i = ib * 64 + t;
  if (i >= n) return;
  const float x1 = boxes[4 * i], y1 = boxes[4 * i + 1], x2 = boxes[4 * i + 2], y2 = boxes[4 * i + 3];
  const float area_i = (x2 - x1) * (y2 - y1);
  const int gi = group[i];
  unsigned long long m = 0;
  const int jn = min(64, n - j0);
  for (int j = (jb == ib ? t + 1 : 0); j < jn; ++j) {
    if (jg[j] != gi) continue;
    const float xx1 = fmaxf(x1, jbx[j][0]), yy1 = fmaxf(y1, jbx[j][1]), xx2 = fminf(x2, jbx[j][2]), yy2 = fminf(y2, jbx[j][3]);
    const float iw = fmaxf(xx2 - xx1, 0.f), ih = fmaxf(yy2 - yy1, 0.f);
    const float inter = iw * ih;
    const float area_j = (jbx[j][2] - jbx[j][0]) * (jbx[j][3] - jbx[j][1]);
    const float iou = inter / (area_i + area_j - inter);
    if (iou > thr) m |= 1ull << j;
  }
  mask[(int64_t)i * words + jb] = m;
}

struct RoiArgs {
  const float* feat[4];
  int fh[4], fw[4];
  float scale[4];
  int n_levels, C, k_min, K, out, sr;
  const float* rois;  // [K][5]: image index, x1, y1, x2, y2
  float* y;           // [K][out][out][C]
  int32_t* level_out; // [K] or null
};

__device__ __forceinline__ float4 bilerp4(const float* f, int H, int W, int C, float y, float x, int c4) {
  // torchvision roi_align bilinear_interpolate (ops/csrc/cpu/roi_align_common.h): samples outside [-1, size] are 0
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return make_float4(0.f, 0.f, 0.f, 0.f);
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
  const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
  const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  const float4 v1 = *reinterpret_cast<const float4*>(f + ((int64_t)yl * W + xl) * C + 4 * c4);
  const float4 v2 = *reinterpret_cast<const float4*>(f + ((int64_t)yl * W + xh) * C + 4 * c4);
  const float4 v3 = *reinterpret_cast<const float4*>(f + ((int64_t)yh * W + xl) * C + 4 * c4);
  const float4 v4 = *reinterpret_cast<const float4*>(f + ((int64_t)yh * W + xh) * C + 4 * c4);
  return make_float4(w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x, w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y,
                     w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z, w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w);
}

// grid (out * out, K), block C / 4 threads
__global__ void roi_align_levels_kernel(RoiArgs a) {
  const int k = blockIdx.y, bin = blockIdx.x, ph = bin / a.out, pw = bin - ph * a.out, c4 = threadIdx.x;
  const float* r = a.rois + 5 * (int64_t)k;
  const int img = (int)r[0];
  // LevelMapper (ops/poolers.py): floor(lvl0 + log2(sqrt(area) / s0) + eps), clamped
  const float s = sqrtf((r[3] - r[1]) * (r[4] - r[2]));
  float lv = floorf((4.0f + log2f(s / 224.0f)) + 1e-6f);
  lv = fminf(fmaxf(lv, (float)a.k_min), (float)(a.k_min + a.n_levels - 1));
  const int l = (int)lv - a.k_min;
  if (a.level_out && bin == 0 && c4 == 0) a.level_out[k] = l;
  const int H = a.fh[l], W = a.fw[l], C = a.C;
  const float sc = a.scale[l];
  const float* f = a.feat[l] + (int64_t)img * H * W * C;
  const float x1 = r[1] * sc, y1 = r[2] * sc, x2 = r[3] * sc, y2 = r[4] * sc;
  float rw = x2 - x1, rh = y2 - y1;
  rw = rw < 1.0f ? 1.0f : rw;  // aligned = False
  rh = rh < 1.0f ? 1.0f : rh;
  const float bh = rh / (float)a.out, bw = rw / (float)a.out;
  const int g = a.sr;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int iy = 0; iy < g; ++iy) {
    const float yy = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)g;
    for (int ix = 0; ix < g; ++ix) {
      const float xx = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)g;
      const float4 v = bilerp4(f, H, W, C, yy, xx, c4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  const float cnt = (float)(g * g);
  *reinterpret_cast<float4*>(a.y + (((int64_t)k * a.out + ph) * a.out + pw) * C + 4 * c4) =
      make_float4(acc.x / cnt, acc.y / cnt, acc.z / cnt, acc.w / cnt);
}

// one thread per (proposal, class): softmax score + decoded, clipped box
__global__ __launch_bounds__(256) void box_postprocess_kernel(const float* logits, int ld_logits, const float* reg, int ld_reg,
                                                              const float* props, int n, int n_classes, float im_h, float im_w,
                                                              float* scores, float* boxes) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n * n_classes) return;
  const int i = t / n_classes, c = t - i * n_classes;
  const float* lg = logits + (int64_t)i * ld_logits;
  float mx = lg[0];
  for (int k = 1; k < n_classes; ++k) mx = fmaxf(mx, lg[k]);
  float den = 0.f;
  for (int k = 0; k < n_classes; ++k) den += expf(lg[k] - mx);
  scores[t] = expf(lg[c] - mx) / den;
  const float* p = props + 4 * (int64_t)i;
  const float* d = reg + (int64_t)i * ld_reg + 4 * c;
  const float wd = p[2] - p[0], ht = p[3] - p[1], cx = p[0] + 0.5f * wd, cy = p[1] + 0.5f * ht;
  const float dx = d[0] / 10.f, dy = d[1] / 10.f, dw = fminf(d[2] / 5.f, kXformClip), dh = fminf(d[3] / 5.f, kXformClip);
  const float pcx = dx * wd + cx, pcy = dy * ht + cy, pw = expf(dw) * wd, ph = expf(dh) * ht;
  float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
  x1 = fminf(fmaxf(x1, 0.f), im_w); x2 = fminf(fmaxf(x2, 0.f), im_w);
  y1 = fminf(fmaxf(y1, 0.f), im_h); y2 = fminf(fmaxf(y2, 0.f), im_h);
  float* o = boxes + 4 * (int64_t)t;
  o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
}

// grid (H, n), threads over W.  mask logits [n][14][14][2][2][ld] (the mask head's deconvolution kept as four 1x1 convs)
__global__ __launch_bounds__(256) void paste_masks_kernel(const float* ml, int ld, const int32_t* labels, const float* boxes, int n,
                                                          int H, int W, float* out) {
  const int k = blockIdx.y, y = blockIdx.x;
  const int M = 28, P = 1, MP = M + 2 * P;
  const float scale = (float)MP / (float)M;
  const float* b = boxes + 4 * (int64_t)k;
  // expand_boxes, then .to(int64): truncation towards zero
  const float wh = (b[2] - b[0]) * 0.5f * scale, hh = (b[3] - b[1]) * 0.5f * scale, xc = (b[2] + b[0]) * 0.5f, yc = (b[3] + b[1]) * 0.5f;
  const long long bx0 = (long long)(xc - wh), by0 = (long long)(yc - hh), bx1 = (long long)(xc + wh), by1 = (long long)(yc + hh);
  long long w = bx1 - bx0 + 1, h = by1 - by0 + 1;
  w = w > 1 ? w : 1; h = h > 1 ? h : 1;
  const long long x_0 = bx0 > 0 ? bx0 : 0, x_1 = bx1 + 1 < W ? bx1 + 1 : W, y_0 = by0 > 0 ? by0 : 0, y_1 = by1 + 1 < H ? by1 + 1 : H;
  const int cls = labels[k];
  const float* m = ml + (int64_t)k * 14 * 14 * 4 * ld + cls;
  auto prob = [&](int py, int px) -> float {  // padded 30 x 30 probability map
    py -= P; px -= P;
    if ((unsigned)py >= (unsigned)M || (unsigned)px >= (unsigned)M) return 0.f;
    const float v = m[((((int64_t)(py >> 1) * 14 + (px >> 1)) * 2 + (py & 1)) * 2 + (px & 1)) * ld];
    return 1.f / (1.f + expf(-v));
  };
  const float sy = (float)MP / (float)h, sx = (float)MP / (float)w;
  for (int x = threadIdx.x; x < W; x += 256) {
    float v = 0.f;
    if (y >= y_0 && y < y_1 && x >= x_0 && x < x_1) {
      // F.interpolate(mode="bilinear", align_corners=False) of the 30 x 30 map to (h, w), sampled at (y - by0, x - bx0)
      float fy = sy * ((float)(y - by0) + 0.5f) - 0.5f, fx = sx * ((float)(x - bx0) + 0.5f) - 0.5f;
      fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
      const int iy0 = (int)fy, ix0 = (int)fx;
      const int iy1 = iy0 + (iy0 < MP - 1 ? 1 : 0), ix1 = ix0 + (ix0 < MP - 1 ? 1 : 0);
      const float ly = fy - (float)iy0, lx = fx - (float)ix0;
      v = (1.f - ly) * ((1.f - lx) * prob(iy0, ix0) + lx * prob(iy0, ix1)) + ly * ((1.f - lx) * prob(iy1, ix0) + lx * prob(iy1, ix1));
    }
    out[((int64_t)k * H + y) * W + x] = v;
  }
}

}  // namespace
}  // namespace hp

using namespace hp;

extern "C" int hp_rpn_decode(const float* d_objectness, const int32_t* d_anchor_idx, int n, const float* d_deltas_map, int map_w,
                             int n_anchors, const float* h_base_anchors, int stride_h, int stride_w, float im_h, float im_w,
                             float min_size, float* d_boxes, float* d_scores, uint8_t* d_valid, void* stream) {
  HP_REQUIRE(n >= 0 && n_anchors == 3 && map_w > 0, "hp_rpn_decode: bad sizes (3 anchors per location)");
  if (n == 0) return HP_OK;
  HP_REQUIRE(d_objectness && d_anchor_idx && d_deltas_map && h_base_anchors && d_boxes && d_scores && d_valid, "hp_rpn_decode: null pointer");
  RpnArgs a{};
  a.obj = d_objectness; a.idx = d_anchor_idx; a.deltas = d_deltas_map; a.n = n; a.w = map_w; a.A = n_anchors;
  for (int i = 0; i < 12; ++i) a.base[i] = h_base_anchors[i];
  a.stride_h = stride_h; a.stride_w = stride_w; a.im_h = im_h; a.im_w = im_w; a.min_size = min_size;
  a.boxes = d_boxes; a.scores = d_scores; a.valid = d_valid;
  hipLaunchKernelGGL(rpn_decode_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("rpn_decode_kernel");
}

extern "C" int hp_nms(const float* d_boxes, const int32_t* d_group, int n, float iou_threshold, uint8_t* h_keep, void* stream) {
  HP_REQUIRE(n >= 0 && n <= 65536, "hp_nms: at most 65536 boxes");
  if (n == 0) return HP_OK;
  HP_REQUIRE(d_boxes && d_group && h_keep, "hp_nms: null pointer");
  const int words = (n + 63) / 64;
  unsigned long long* d_mask = nullptr;
  HP_CHECK_HIP(hipMalloc((void**)&d_mask, (size_t)n * words * 8));
  HP_CHECK_HIP(hipMemsetAsync(d_mask, 0, (size_t)n * words * 8, (hipStream_t)stream));
  hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words), dim3(64), 0, (hipStream_t)stream, d_boxes, d_group, n, iou_threshold, d_mask, words);
  std::vector<unsigned long long> m((size_t)n * words);
  hipError_t e = hipMemcpyAsync(m.data(), d_mask, m.size() * 8, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(d_mask);
  if (e != hipSuccess) return fail(HP_ERR_HIP, std::string("hp_nms: ") + hipGetErrorString(e));
  std::vector<unsigned long long> removed(words, 0ull);
  for (int i = 0; i < n; ++i) {
    if (removed[i >> 6] >> (i & 63) & 1ull) { h_keep[i] = 0; continue; }
    h_keep[i] = 1;
    const unsigned long long* row = m.data() + (size_t)i * words;
    for (int wd = i >> 6; wd < words; ++wd) removed[wd] |= row[wd];
  }
  return HP_OK;
}

extern "C" int hp_roi_align_levels(const float* const* h_feat_ptrs, const int* h_heights, const int* h_widths, const float* h_scales,
                                   int n_levels, int k_min, int C, const float* d_rois, int K, int out_size, int sampling_ratio,
                                   float* d_out, int32_t* d_levels, void* stream) {
  HP_REQUIRE(n_levels >= 1 && n_levels <= 4 && C % 4 == 0 && C >= 4 && C <= 4096 && K >= 0 && out_size >= 1 && sampling_ratio >= 1,
             "hp_roi_align_levels: bad sizes");
  if (K == 0) return HP_OK;
  HP_REQUIRE(h_feat_ptrs && h_heights && h_widths && h_scales && d_rois && d_out, "hp_roi_align_levels: null pointer");
  RoiArgs a{};
  for (int l = 0; l < n_levels; ++l) { a.feat[l] = h_feat_ptrs[l]; a.fh[l] = h_heights[l]; a.fw[l] = h_widths[l]; a.scale[l] = h_scales[l]; }
  a.n_levels = n_levels; a.C = C; a.k_min = k_min; a.K = K; a.out = out_size; a.sr = sampling_ratio; a.rois = d_rois; a.y = d_out;
  a.level_out = d_levels;
  hipLaunchKernelGGL(roi_align_levels_kernel, dim3(out_size * out_size, K), dim3(C / 4), 0, (hipStream_t)stream, a);
  return check_launch("roi_align_levels_kernel");
}

extern "C" int hp_box_postprocess(const float* d_class_logits, int ld_logits, const float* d_box_regression, int ld_regression,
                                  const float* d_proposals, int n, int n_classes, float im_h, float im_w, float* d_scores,
                                  float* d_boxes, void* stream) {
  HP_REQUIRE(n >= 0 && n_classes >= 2 && ld_logits >= n_classes && ld_regression >= 4 * n_classes, "hp_box_postprocess: bad sizes");
  if (n == 0) return HP_OK;
  HP_REQUIRE(d_class_logits && d_box_regression && d_proposals && d_scores && d_boxes, "hp_box_postprocess: null pointer");
  hipLaunchKernelGGL(box_postprocess_kernel, dim3((n * n_classes + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_class_logits,
                     ld_logits, d_box_regression, ld_regression, d_proposals, n, n_classes, im_h, im_w, d_scores, d_boxes);
  return check_launch("box_postprocess_kernel");
}

extern "C" int hp_paste_masks(const float* d_mask_logits, int ld, const int32_t* d_labels, const float* d_boxes, int n, int H, int W,
                              float* d_out, void* stream) {
  HP_REQUIRE(n >= 0 && H > 0 && W > 0 && ld >= 1, "hp_paste_masks: bad sizes");
  if (n == 0) return HP_OK;
  HP_REQUIRE(d_mask_logits && d_labels && d_boxes && d_out, "hp_paste_masks: null pointer");
  hipLaunchKernelGGL(paste_masks_kernel, dim3(H, n), dim3(256), 0, (hipStream_t)stream, d_mask_logits, ld, d_labels, d_boxes, n, H, W,
                     d_out);
  return check_launch("paste_masks_kernel");
}
