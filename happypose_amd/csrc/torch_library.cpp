// The hot-path operators as a COMPILED PyTorch operator library: TORCH_LIBRARY(happypose_amd) with kernels for the CUDA
// dispatch key (= HIP on ROCm) and shape functions for Meta.  north_star names the product "PyTorch-ROCm custom ops"; the
// drop-in boundary stays the C ABI (include/happypose_amd.h) -- every kernel below is argument checking, output allocation
// on the caller's device and ONE hp_* call on torch's current HIP stream.  No CPU kernels: a CPU tensor raises
// NotImplementedError from the dispatcher.
//
// Built into happypose_amd/lib/libhappypose_amd_torch.so by happypose_amd/build.py (g++, host code only; links the C-ABI
// library) and loaded with torch.ops.load_library by happypose_amd/torch_ops.py.
//
// Objects behind C handles (hp_mesh_store*, hp_net*) travel through the schemas as `int` = the handle's address; an address
// must have been announced with happypose_amd::register_handle (torch_ops.ticket does it and withdraws it when the owning
// Python object dies), so a stale integer raises instead of being dereferenced.
//
// Reference counterparts: crop_images (TB/lib3d/cropping.py:155-197), Panda3dBatchRenderer.render
// (TB/renderer/panda3d_batch_renderer.py:271-349), PosePredictor.update_pose (MP/models/pose_rigid.py:456-481),
// PosePredictor.crop_inputs / compute_crops_multiview / make_TCO_multiview (:235-335, TB/lib3d/multiview.py:166-251),
// PosePredictor.net_forward (:352-374).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/happypose_amd.h"

namespace {

using at::Tensor;
using c10::optional;

enum HandleKind : int64_t { kStore = 1, kNet = 2 };
std::mutex g_mu;
std::unordered_map<int64_t, int64_t> g_handles;  // address -> kind

void register_handle(int64_t address, int64_t kind) {
  TORCH_CHECK(address != 0 && (kind == kStore || kind == kNet), "happypose_amd::register_handle: bad argument");
  std::lock_guard<std::mutex> lock(g_mu);
  g_handles[address] = kind;
}

void release_handle(int64_t address) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_handles.erase(address);
}

template <class T>
T* resolve(int64_t address, HandleKind kind, const char* what) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_handles.find(address);
  TORCH_CHECK_VALUE(it != g_handles.end() && it->second == kind, "happypose_amd op: ", address, " does not name a live ", what);
  return reinterpret_cast<T*>(static_cast<uintptr_t>(address));
}

void check(int rc, const char* fn) {
  TORCH_CHECK(rc == HP_OK, fn, " failed (", rc, "): ", hp_last_error());
}

Tensor f32(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), "happypose_amd op: ", name, " must be a device tensor");
  return t.to(at::kFloat).contiguous();
}
Tensor i32(const Tensor& t, const Tensor& like, const char* name) {
  return t.to(like.device(), at::kInt).contiguous();
}
const float* fp(const Tensor& t) { return t.data_ptr<float>(); }
const float* fp(const optional<Tensor>& t) { return t.has_value() ? t->data_ptr<float>() : nullptr; }
// a ROCm build of torch keeps the device type "cuda": its guards / streams are the ...MasqueradingAsCUDA classes
using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;
void* stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }
hp_strides nchw(int64_t c, int64_t h, int64_t w) { return hp_strides{c * h * w, 0, h * w, w, 1}; }

// ---- crop -----------------------------------------------------------------------------------------------------------
Tensor crop_roi_align(const Tensor& images, const Tensor& boxes, const Tensor& im_ids, int64_t out_h, int64_t out_w, int64_t sampling_ratio) {
  TORCH_CHECK(images.dim() == 4 && images.scalar_type() == at::kFloat && images.is_contiguous(), "crop_roi_align: images must be contiguous fp32 [B,C,H,W]");
  TORCH_CHECK(boxes.dim() == 2 && boxes.size(1) == 4 && im_ids.dim() == 1 && im_ids.size(0) == boxes.size(0), "crop_roi_align: boxes [n,4], im_ids [n]");
  const DeviceGuard guard(images.device());
  const Tensor b = f32(boxes.to(images.device()), "boxes"), ids = i32(im_ids, images, "im_ids");
  const int64_t n = b.size(0), C = images.size(1);
  Tensor out = at::empty({n, C, out_h, out_w}, images.options());
  const hp_strides st = nchw(C, out_h, out_w);
  check(hp_crop_roi_align(fp(images), (int)images.size(0), (int)C, (int)C, (int)images.size(2), (int)images.size(3), fp(b), ids.data_ptr<int32_t>(),
                          (int)n, (int)out_h, (int)out_w, (int)sampling_ratio, out.data_ptr<float>(), &st, nullptr, 0, stream_of(images)),
        "hp_crop_roi_align");
  return out;
}
Tensor crop_roi_align_meta(const Tensor& images, const Tensor& boxes, const Tensor& im_ids, int64_t out_h, int64_t out_w, int64_t) {
  return at::empty({boxes.size(0), images.size(1), out_h, out_w}, images.options());
}

// ---- pose update ----------------------------------------------------------------------------------------------------
Tensor pose_update(const Tensor& TCO, const Tensor& K_crop, const Tensor& pose9, const optional<Tensor>& tCR) {
  const int64_t b = TCO.size(0);
  TORCH_CHECK(TCO.dim() == 3 && TCO.size(1) == 4 && TCO.size(2) == 4 && pose9.dim() == 2 && pose9.size(0) == b && pose9.size(1) == 9,
              "pose_update: TCO [b,4,4], pose9 [b,9]");
  TORCH_CHECK(K_crop.dim() >= 3 && K_crop.size(0) == b && K_crop.size(-1) == 3 && K_crop.size(-2) == 3, "pose_update: K_crop [b,(V,)3,3]");
  const DeviceGuard guard(TCO.device());
  const Tensor T = f32(TCO, "TCO"), K = f32(K_crop, "K_crop"), p = f32(pose9, "pose9");
  optional<Tensor> c;
  if (tCR.has_value()) {
    TORCH_CHECK(tCR->dim() == 2 && tCR->size(0) == b && tCR->size(1) == 3, "pose_update: tCR [b,3]");
    c = f32(*tCR, "tCR");
  }
  Tensor out = at::empty_like(T);
  check(hp_pose_update((int)b, fp(T), fp(K), (int)(K.numel() / std::max<int64_t>(b, 1)), fp(p), fp(c), out.data_ptr<float>(), stream_of(T)), "hp_pose_update");
  return out;
}
Tensor pose_update_meta(const Tensor& TCO, const Tensor&, const Tensor&, const optional<Tensor>&) { return at::empty_like(TCO); }

// ---- rasteriser -----------------------------------------------------------------------------------------------------
std::vector<Tensor> rasterize(int64_t store, const Tensor& obj_ids, const Tensor& TCO, const Tensor& K, int64_t height, int64_t width, bool normals,
                              bool depth, bool msaa, bool aniso) {
  const hp_mesh_store* s = resolve<hp_mesh_store>(store, kStore, "MeshStore");
  const int64_t n = TCO.size(0);
  TORCH_CHECK(TCO.dim() == 3 && TCO.size(1) == 4 && TCO.size(2) == 4 && K.dim() == 3 && K.size(0) == n && K.size(1) == 3 && K.size(2) == 3 &&
                  obj_ids.dim() == 1 && obj_ids.size(0) == n, "rasterize: obj_ids [n], TCO [n,4,4], K [n,3,3]");
  const DeviceGuard guard(TCO.device());
  const Tensor T = f32(TCO, "TCO"), Kc = f32(K, "K"), ids = i32(obj_ids, T, "obj_ids");
  const auto o = T.options();
  Tensor rgb = at::empty({n, 3, height, width}, o), nrm, dep;
  if (normals) nrm = at::empty({n, 3, height, width}, o);
  if (depth) dep = at::empty({n, 1, height, width}, o);
  const hp_strides cs = nchw(3, height, width), ds = nchw(1, height, width);
  const int flags = HP_RASTER_QUANT8 | (msaa ? HP_RASTER_MSAA4 : 0) | (aniso ? HP_RASTER_TEX_ANISO : 0);
  check(hp_rasterize(s, (int)n, 1, ids.data_ptr<int32_t>(), fp(T), fp(Kc), nullptr, 0, nullptr, nullptr, (int)height, (int)width, flags, rgb.data_ptr<float>(),
                     normals ? nrm.data_ptr<float>() : nullptr, &cs, depth ? dep.data_ptr<float>() : nullptr, &ds, nullptr, nullptr, 0, stream_of(T)),
        "hp_rasterize");
  std::vector<Tensor> out{rgb};
  if (normals) out.push_back(nrm);
  if (depth) out.push_back(dep);
  return out;
}
std::vector<Tensor> rasterize_meta(int64_t, const Tensor&, const Tensor& TCO, const Tensor&, int64_t height, int64_t width, bool normals, bool depth, bool, bool) {
  const int64_t n = TCO.size(0);
  const auto o = TCO.options().dtype(at::kFloat);
  std::vector<Tensor> out{at::empty({n, 3, height, width}, o)};
  if (normals) out.push_back(at::empty({n, 3, height, width}, o));
  if (depth) out.push_back(at::empty({n, 1, height, width}, o));
  return out;
}

// ---- per-iteration geometry -----------------------------------------------------------------------------------------
int64_t views_of(const std::string& multiview_type, int* code) {
  // MP/models/pose_rigid.py:289-305: "TCO" | "TCO+front_1view" | "TCO+front_3views" | "TCO+front_5views"
  if (multiview_type == "TCO") { *code = 0; return 1; }
  if (multiview_type == "TCO+front_1view") { *code = 1; return 2; }
  if (multiview_type == "TCO+front_3views") { *code = 3; return 4; }
  if (multiview_type == "TCO+front_5views") { *code = 5; return 6; }
  TORCH_CHECK_VALUE(false, "pose_prep: unknown multiview_type '", multiview_type, "'");
}

std::vector<Tensor> pose_prep(int64_t store, const Tensor& TCO, const Tensor& K, const Tensor& im_ids, const Tensor& obj_ids, const Tensor& point_ids,
                              const optional<Tensor>& point_ids_extra, int64_t im_h, int64_t im_w, int64_t crop_h, int64_t crop_w,
                              std::string multiview_type, bool normalize, double lamb) {
  const hp_mesh_store* s = resolve<hp_mesh_store>(store, kStore, "MeshStore");
  int mv = 0;
  const int64_t V = views_of(multiview_type, &mv), b = TCO.size(0);
  TORCH_CHECK(TCO.dim() == 3 && TCO.size(1) == 4 && TCO.size(2) == 4 && K.dim() == 3 && K.size(1) == 3 && K.size(2) == 3, "pose_prep: TCO [b,4,4], K [n_images,3,3]");
  TORCH_CHECK(im_ids.dim() == 1 && im_ids.size(0) == b && obj_ids.dim() == 1 && obj_ids.size(0) == b, "pose_prep: im_ids [b], obj_ids [b]");
  TORCH_CHECK(V == 1 || point_ids_extra.has_value(), "pose_prep: the look-at views need point_ids_extra (the 200-point sub-sample)");
  const DeviceGuard guard(TCO.device());
  const Tensor T = f32(TCO, "TCO"), Kc = f32(K, "K"), im = i32(im_ids, T, "im_ids"), ob = i32(obj_ids, T, "obj_ids"), pm = i32(point_ids, T, "point_ids");
  Tensor pe;
  if (V > 1) pe = i32(*point_ids_extra, T, "point_ids_extra");
  const auto o = T.options();
  Tensor TCO_out = at::empty({b, 4, 4}, o), tCR = at::empty({b, 3}, o), TCV_O = at::empty({b, V, 4, 4}, o), boxes_rend = at::empty({b, 4}, o),
         boxes_crop = at::empty({b, 4}, o), K_crop = at::empty({b, V, 3, 3}, o);
  check(hp_pose_prep(s, (int)b, (int)V, mv, normalize ? 1 : 0, fp(T), fp(Kc), (int)Kc.size(0), im.data_ptr<int32_t>(), ob.data_ptr<int32_t>(),
                     pm.data_ptr<int32_t>(), (int)pm.numel(), V > 1 ? pe.data_ptr<int32_t>() : nullptr, V > 1 ? (int)pe.numel() : 0, (int)im_h, (int)im_w,
                     (int)crop_h, (int)crop_w, (float)lamb, TCO_out.data_ptr<float>(), tCR.data_ptr<float>(), TCV_O.data_ptr<float>(),
                     boxes_rend.data_ptr<float>(), boxes_crop.data_ptr<float>(), K_crop.data_ptr<float>(), stream_of(T)),
        "hp_pose_prep");
  return {TCO_out, tCR, TCV_O, boxes_rend, boxes_crop, K_crop};
}
std::vector<Tensor> pose_prep_meta(int64_t, const Tensor& TCO, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const optional<Tensor>&, int64_t,
                                   int64_t, int64_t, int64_t, std::string multiview_type, bool, double) {
  int mv = 0;
  const int64_t V = views_of(multiview_type, &mv), b = TCO.size(0);
  const auto o = TCO.options().dtype(at::kFloat);
  return {at::empty({b, 4, 4}, o), at::empty({b, 3}, o), at::empty({b, V, 4, 4}, o), at::empty({b, 4}, o), at::empty({b, 4}, o), at::empty({b, V, 3, 3}, o)};
}

// ---- network --------------------------------------------------------------------------------------------------------
// x against the planned input map: hp_net_forward strides by h * w * c_pad per sample, so a smaller H or W (a traced Meta shape, a
// caller's mistake) would be an out-of-bounds read, not an error
void check_net_input(const hp_net* n, const Tensor& x) {
  int h = 0, w = 0, c_pad = 0, dev = -1;
  check(hp_net_input_dims(n, &h, &w, &c_pad, &dev), "hp_net_input_dims");
  TORCH_CHECK(!x.is_cuda() || x.device().index() == dev, "net_forward: x is on device ", (int)x.device().index(), ", the network was built on device ", dev);
  TORCH_CHECK(x.dim() == 4 && x.size(1) == h && x.size(2) == w, "net_forward: x must be NHWC [b, ", h, ", ", w, ", c] for this network, got ", x.sizes());
  if (x.scalar_type() == at::kHalf) {
    TORCH_CHECK(hp_net_precision(n) == HP_PRECISION_F16 && x.size(3) == hp_net_input_channels_f16(n),
                "net_forward: an fp16 input needs the fp16 plan and its record width");
  } else {
    TORCH_CHECK(x.scalar_type() == at::kFloat && x.size(3) == c_pad, "net_forward: x must be fp32 [b, h, w, ", c_pad, "]");
  }
}

std::vector<Tensor> net_forward(int64_t net, const Tensor& x) {
  hp_net* n = resolve<hp_net>(net, kNet, "Net");
  int pose_dim = 0, n_logits = 0, n_features = 0;
  check(hp_net_output_dims(n, &pose_dim, &n_logits, &n_features), "hp_net_output_dims");
  TORCH_CHECK(x.is_cuda() && x.dim() == 4 && x.is_contiguous(), "net_forward: x must be a contiguous NHWC device tensor");
  check_net_input(n, x);
  const DeviceGuard guard(x.device());
  const int64_t b = x.size(0);
  const auto o = x.options().dtype(at::kFloat);
  Tensor pose, logits;
  if (pose_dim > 0) pose = at::empty({b, pose_dim}, o);
  if (n_logits > 0) logits = at::empty({b, n_logits}, o);
  float* const dp = pose_dim > 0 ? pose.data_ptr<float>() : nullptr;
  float* const dl = n_logits > 0 ? logits.data_ptr<float>() : nullptr;
  if (x.scalar_type() == at::kHalf) {
    check(hp_net_forward_f16in(n, x.data_ptr(), (int)b, dp, dl, nullptr, stream_of(x)), "hp_net_forward_f16in");
  } else {
    check(hp_net_forward(n, fp(x), (int)b, dp, dl, nullptr, stream_of(x)), "hp_net_forward");
  }
  std::vector<Tensor> out;
  if (pose_dim > 0) out.push_back(pose);
  if (n_logits > 0) out.push_back(logits);
  return out;
}
std::vector<Tensor> net_forward_meta(int64_t net, const Tensor& x) {
  hp_net* n = resolve<hp_net>(net, kNet, "Net");
  int pose_dim = 0, n_logits = 0, n_features = 0;
  check(hp_net_output_dims(n, &pose_dim, &n_logits, &n_features), "hp_net_output_dims");
  check_net_input(n, x);
  const auto o = x.options().dtype(at::kFloat);
  std::vector<Tensor> out;
  if (pose_dim > 0) out.push_back(at::empty({x.size(0), pose_dim}, o));
  if (n_logits > 0) out.push_back(at::empty({x.size(0), n_logits}, o));
  return out;
}

// ---- single conv layer (parity tests / layer-level users) -----------------------------------------------------------
Tensor conv2d_nhwc(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad, const optional<Tensor>& bias, const optional<Tensor>& residual,
                   const optional<Tensor>& pre_scale, const optional<Tensor>& pre_shift, int64_t act) {
  TORCH_CHECK(x.is_cuda() && x.dim() == 4 && w.dim() == 4 && x.size(3) == w.size(3) && x.is_contiguous() && w.is_contiguous(),
              "conv2d_nhwc: x [n,h,w,cin], w [cout,kh,kw,cin], contiguous");
  TORCH_CHECK(x.scalar_type() == w.scalar_type() && (x.scalar_type() == at::kFloat || x.scalar_type() == at::kHalf), "conv2d_nhwc: fp32 or fp16 operands");
  const DeviceGuard guard(x.device());
  const int64_t n = x.size(0), h = x.size(1), wd = x.size(2), cin = x.size(3), cout = w.size(0), kh = w.size(1), kw = w.size(2);
  const int64_t ho = (h + 2 * pad - kh) / stride + 1, wo = (wd + 2 * pad - kw) / stride + 1;
  Tensor y = at::empty({n, ho, wo, cout}, x.options());
  auto raw = [](const optional<Tensor>& t) -> const void* { return t.has_value() ? t->data_ptr() : nullptr; };
  if (x.scalar_type() == at::kFloat) {
    for (const auto* t : {&bias, &residual, &pre_scale, &pre_shift}) TORCH_CHECK(!t->has_value() || (*t)->scalar_type() == at::kFloat, "conv2d_nhwc: fp32 side inputs");
    check(hp_conv2d_nhwc(fp(x), (int)n, (int)h, (int)wd, (int)cin, fp(w), (int)cout, (int)kh, (int)kw, (int)stride, (int)pad, fp(bias), fp(residual),
                         fp(pre_scale), fp(pre_shift), (int)act, y.data_ptr<float>(), stream_of(x)),
          "hp_conv2d_nhwc");
  } else {
    for (const auto* t : {&residual, &pre_scale, &pre_shift}) TORCH_CHECK(!t->has_value() || (*t)->scalar_type() == at::kHalf, "conv2d_nhwc: fp16 side inputs");
    TORCH_CHECK(!bias.has_value() || bias->scalar_type() == at::kFloat, "conv2d_nhwc: the bias of the fp16 kernel is fp32");
    check(hp_conv2d_nhwc_f16(x.data_ptr(), (int)n, (int)h, (int)wd, (int)cin, w.data_ptr(), (int)cout, (int)kh, (int)kw, (int)stride, (int)pad, fp(bias),
                             raw(residual), raw(pre_scale), raw(pre_shift), (int)act, y.data_ptr(), stream_of(x)),
          "hp_conv2d_nhwc_f16");
  }
  return y;
}
Tensor conv2d_nhwc_meta(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad, const optional<Tensor>&, const optional<Tensor>&, const optional<Tensor>&,
                        const optional<Tensor>&, int64_t) {
  return at::empty({x.size(0), (x.size(1) + 2 * pad - w.size(1)) / stride + 1, (x.size(2) + 2 * pad - w.size(2)) / stride + 1, w.size(0)}, x.options());
}

}  // namespace

TORCH_LIBRARY(happypose_amd, m) {
  m.def("register_handle(int address, int kind) -> ()", &register_handle);
  m.def("release_handle(int address) -> ()", &release_handle);
  m.def("crop_roi_align(Tensor images, Tensor boxes, Tensor im_ids, int out_h, int out_w, int sampling_ratio=4) -> Tensor");
  m.def("pose_update(Tensor TCO, Tensor K_crop, Tensor pose9, Tensor? tCR=None) -> Tensor");
  m.def("rasterize(int store, Tensor obj_ids, Tensor TCO, Tensor K, int height, int width, bool normals=False, bool depth=False, bool msaa=False, "
        "bool aniso=False) -> Tensor[]");
  m.def("pose_prep(int store, Tensor TCO, Tensor K, Tensor im_ids, Tensor obj_ids, Tensor point_ids, Tensor? point_ids_extra, int im_h, int im_w, "
        "int crop_h, int crop_w, str multiview_type='TCO', bool normalize=False, float lamb=1.4) -> Tensor[]");
  m.def("net_forward(int net, Tensor x) -> Tensor[]");
  m.def("conv2d_nhwc(Tensor x, Tensor w, int stride, int pad, Tensor? bias=None, Tensor? residual=None, Tensor? pre_scale=None, Tensor? pre_shift=None, "
        "int act=0) -> Tensor");
}

TORCH_LIBRARY_IMPL(happypose_amd, CUDA, m) {
  m.impl("crop_roi_align", &crop_roi_align);
  m.impl("pose_update", &pose_update);
  m.impl("rasterize", &rasterize);
  m.impl("pose_prep", &pose_prep);
  m.impl("net_forward", &net_forward);
  m.impl("conv2d_nhwc", &conv2d_nhwc);
}

TORCH_LIBRARY_IMPL(happypose_amd, Meta, m) {
  m.impl("crop_roi_align", &crop_roi_align_meta);
  m.impl("pose_update", &pose_update_meta);
  m.impl("rasterize", &rasterize_meta);
  m.impl("pose_prep", &pose_prep_meta);
  m.impl("net_forward", &net_forward_meta);
  m.impl("conv2d_nhwc", &conv2d_nhwc_meta);
}
