// Small per-hypothesis geometry of the refinement loop as fused device kernels: the
// reference runs these as ~60 tiny torch ops plus two device<->host round trips per
// iteration (make_TCO_multiview goes through numpy + Panda3D NodePath objects,
// TB/lib3d/multiview.py:189-190,212-222).  Here: one launch before the render
// (hp_pose_prep) and one after the network (hp_pose_update).
#include "common.h"

namespace hp {

#pragma clang fp contract(off)

constexpr int kT = 256;

// ---- tiny matrix helpers (row-major) -------------------------------------------------
__device__ __forceinline__ void normalize_T_dev(const float* Tin, float* T) {
  // TB/lib3d/transform_ops.py:107-120 + TB/lib3d/rotations.py:22-36: Gram-Schmidt on the
  // first two COLUMNS of R; columns of the result are (x, y, z).
  float xr[3] = {Tin[0], Tin[4], Tin[8]};
  float yr[3] = {Tin[1], Tin[5], Tin[9]};
  float nx = sqrtf(xr[0] * xr[0] + xr[1] * xr[1] + xr[2] * xr[2]);
  float x[3] = {xr[0] / nx, xr[1] / nx, xr[2] / nx};
  float z[3] = {x[1] * yr[2] - x[2] * yr[1], x[2] * yr[0] - x[0] * yr[2], x[0] * yr[1] - x[1] * yr[0]};
  float nz = sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
  z[0] /= nz; z[1] /= nz; z[2] /= nz;
  float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
  T[0] = x[0]; T[1] = y[0]; T[2] = z[0]; T[3] = Tin[3];
  T[4] = x[1]; T[5] = y[1]; T[6] = z[1]; T[7] = Tin[7];
  T[8] = x[2]; T[9] = y[2]; T[10] = z[2]; T[11] = Tin[11];
  T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

// look-at pose (camera -> cam0, OpenCV axes) in double, as the reference's numpy/Panda path
__device__ __forceinline__ void look_at_cv(const double* pos, const double* tgt, double* M /*[12]*/) {
  double f[3] = {tgt[0] - pos[0], tgt[1] - pos[1], tgt[2] - pos[2]};
  double nf = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
  f[0] /= nf; f[1] /= nf; f[2] /= nf;
  const double up[3] = {0.0, -1.0, 0.0};
  double r[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
  double nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  r[0] /= nr; r[1] /= nr; r[2] /= nr;
  double u[3] = {r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0]};
  // columns: right, -up', forward ; translation pos
  M[0] = r[0]; M[1] = -u[0]; M[2] = f[0]; M[3] = pos[0];
  M[4] = r[1]; M[5] = -u[1]; M[6] = f[1]; M[7] = pos[1];
  M[8] = r[2]; M[9] = -u[2]; M[10] = f[2]; M[11] = pos[2];
}

// TCV_O = inv(TC0_CV) @ TCO for view v of "TCO+front_{1,3,5}views" (TB/lib3d/multiview.py:28-92,
// 166-251; closed form derived in SURVEY.md A.9).  View 0 is the input pose itself.
__device__ __forceinline__ void view_pose(const float* T, int v, float* TV) {
  if (v == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) TV[k] = T[k];
    return;
  }
  bool fin = true;
#pragma unroll
  for (int k = 0; k < 12; ++k) fin &= isfinite(T[k]);
  double tcr[3] = {fin ? (double)T[3] : 0.0, fin ? (double)T[7] : 0.0, fin ? (double)T[11] : 0.0};
  double radius = sqrt(tcr[0] * tcr[0] + tcr[1] * tcr[1] + tcr[2] * tcr[2]);
  double M[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  if (radius > 0.0) {
    double zero[3] = {0, 0, 0};
    double base[12];
    look_at_cv(zero, tcr, base);
    // offsets in the Panda frame of `base`: x = right (cv x), y = forward (cv z), z = up (-cv y)
    const int off[6][3] = {{0, 0, 0}, {0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 0, 1}, {0, 0, -1}};
    double pos[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      pos[c] = radius * (off[v][0] * base[4 * c + 0] + off[v][1] * base[4 * c + 2] - off[v][2] * base[4 * c + 1]);
    look_at_cv(pos, tcr, M);
  }
  float Mf[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) Mf[k] = (float)M[k];
  // invert_transform_matrices (TB/lib3d/transform_ops.py:59-67) in fp32, then @ TCO
  float Ri[9] = {Mf[0], Mf[4], Mf[8], Mf[1], Mf[5], Mf[9], Mf[2], Mf[6], Mf[10]};
  float ti[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) ti[r] = -(Ri[3 * r] * Mf[3] + Ri[3 * r + 1] * Mf[7] + Ri[3 * r + 2] * Mf[11]);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float acc = Ri[3 * r] * T[c] + Ri[3 * r + 1] * T[4 + c] + Ri[3 * r + 2] * T[8 + c];
      TV[4 * r + c] = acc + ti[r] * T[12 + c];
    }
  }
  TV[12] = T[12]; TV[13] = T[13]; TV[14] = T[14]; TV[15] = T[15];
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

struct PrepArgs {
  const float* points; int n_pad;
  const float* TCO_in; const float* K; const int32_t* im_ids; const int32_t* obj_ids;
  const int32_t* ids_main; int n_main; const int32_t* ids_extra; int n_extra;
  int b, n_views, normalize, im_h, im_w, crop_h, crop_w; float lamb;
  int n_images, n_obj;  // rows of K / objects of the store: an id outside them poisons the hypothesis (NaN outputs)
  float* TCO_out; float* tCR; float* TCV_O; float* boxes_rend; float* boxes_crop; float* K_crop;
  // remove_TCO_rendering (TB/lib3d/multiview.py:189-236, MP/models/pose_rigid.py:609-611): the identity view is still
  // evaluated (it defines TCO / tCR / the crop box / the crop's K) but is not one of the n_views rendered views: its K goes
  // to K_crop_main, the look-at views fill slots 0..n_views-1 of TCV_O / K_crop with THEIR OWN 200-point crop intrinsics
  int skip_tco; float* K_crop_main;
};

__global__ __launch_bounds__(kT) void pose_prep_kernel(PrepArgs a) {
  __shared__ float red[4][4];
  const int nvt = a.n_views + a.skip_tco;  // evaluated views: the rendered ones (+ the identity view when it is not rendered)
  const int i = blockIdx.x / nvt, v = blockIdx.x % nvt;
  const int slot = v - a.skip_tco;         // position among the rendered views, -1: the unrendered identity view
  const int tid = threadIdx.x;
  // The pose chain (normalize_T, the double-precision look-at of the extra views, P = K @ TV) is evaluated by lane 0
  // ONLY and the projection matrix handed to the other lanes through LDS.  Every lane used to evaluate it for itself:
  // redundant, and -- measured -- not reproducible: with conv launches of another stream sharing the CUs, single
  // lanes sporadically (~1 launch in 500) came out of the fp64 look-at with a different P than lane 0, their points
  // projected a few pixels off and K_crop of that view changed by up to 100 px (two-lane MegaPose refiner; found by
  // the graph-replay test).  One evaluation per view also makes "all lanes agree" true by construction.
  __shared__ float Psh[12];
  float T[16], TV[16], K[9], P[12];
  const int im_id = a.im_ids[i], ob_id = a.obj_ids[i];
  // an image / object id outside the tables never leaves them: the hypothesis reads row 0 and its outputs are NaN
  // (the reference's indexing would raise; a device-side check cannot, and NaN poses render as zero images)
  const bool bad_id = (unsigned)im_id >= (unsigned)a.n_images || (unsigned)ob_id >= (unsigned)a.n_obj;
  if (tid == 0) {
    float Tin[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) Tin[k] = a.TCO_in[16 * (int64_t)i + k];
    if (a.normalize) normalize_T_dev(Tin, T);
    else {
#pragma unroll
      for (int k = 0; k < 16; ++k) T[k] = Tin[k];
    }
    view_pose(T, v, TV);
    const float* Kp = a.K + 9 * (int64_t)(bad_id ? 0 : im_id);
#pragma unroll
    for (int k = 0; k < 9; ++k) K[k] = Kp[k];
    // P = K @ TV[:3]  (TB/lib3d/camera_geometry.py:52)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        Psh[4 * r + c] = fmaf(K[3 * r + 2], TV[8 + c], fmaf(K[3 * r + 1], TV[4 + c], K[3 * r] * TV[c]));
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 12; ++k) P[k] = Psh[k];

  const int32_t* ids = v == 0 ? a.ids_main : a.ids_extra;
  const int npts = v == 0 ? a.n_main : a.n_extra;
  const float* pts = a.points + (int64_t)(bad_id ? 0 : ob_id) * a.n_pad * 3;
  float x1 = INFINITY, y1 = INFINITY, x2 = -INFINITY, y2 = -INFINITY;
  for (int j = tid; j < npts; j += kT) {
    const float* p = pts + 3 * (int64_t)ids[j];
    float su = fmaf(P[2], p[2], fmaf(P[1], p[1], P[0] * p[0])) + P[3];
    float sv = fmaf(P[6], p[2], fmaf(P[5], p[1], P[4] * p[0])) + P[7];
    float sz = fmaf(P[10], p[2], fmaf(P[9], p[1], P[8] * p[0])) + P[11];
    sz = fmaxf(0.1f, sz);  // project_points_robust z clamp (:53-54)
    float u = su / sz, w = sv / sz;
    x1 = fminf(x1, u); x2 = fmaxf(x2, u); y1 = fminf(y1, w); y2 = fmaxf(y2, w);
  }
  x1 = wave_min(x1); y1 = wave_min(y1); x2 = wave_max(x2); y2 = wave_max(y2);
  if ((tid & 63) == 0) { red[tid >> 6][0] = x1; red[tid >> 6][1] = y1; red[tid >> 6][2] = x2; red[tid >> 6][3] = y2; }
  __syncthreads();
  if (tid != 0) return;
#pragma unroll
  for (int wv = 1; wv < 4; ++wv) {
    x1 = fminf(x1, red[wv][0]); y1 = fminf(y1, red[wv][1]);
    x2 = fmaxf(x2, red[wv][2]); y2 = fmaxf(y2, red[wv][3]);
  }
  // centre = projection of the reference point (tOR = 0 -> origin of the object frame)
  float cz = fmaxf(0.1f, P[11]);
  float xc = P[3] / cz, yc = P[7] / cz;
  // deepim_boxes (TB/lib3d/cropping.py:27-75), obs box == rendered box
  float xdist = fmaxf(fabsf(x1 - xc), fabsf(x2 - xc));
  float ydist = fmaxf(fabsf(y1 - yc), fabsf(y2 - yc));
  const int wmax = a.im_h > a.im_w ? a.im_h : a.im_w, hmin = a.im_h > a.im_w ? a.im_w : a.im_h;
  const float r = (float)((double)wmax / (double)hmin);
  float width = fmaxf(xdist, ydist * r) * 2.0f * a.lamb;
  float height = fmaxf(xdist / r, ydist) * 2.0f * a.lamb;
  float bx1 = xc - width / 2.0f, by1 = yc - height / 2.0f, bx2 = xc + width / 2.0f, by2 = yc + height / 2.0f;
  // get_K_crop_resize (TB/lib3d/camera_geometry.py:70-122)
  const float fw = (float)(a.crop_h > a.crop_w ? a.crop_h : a.crop_w);
  const float fh = (float)(a.crop_h > a.crop_w ? a.crop_w : a.crop_h);
  float cw = bx2 - bx1, ch = by2 - by1;
  float cj = (bx1 + bx2) / 2.0f, ci = (by1 + by2) / 2.0f;
  float cx = K[2] + (cw - 1.0f) / 2.0f - cj;
  float cy = K[5] + (ch - 1.0f) / 2.0f - ci;
  float ocx = cx - (cw - 1.0f) / 2.0f, ocy = cy - (ch - 1.0f) / 2.0f;
  float sx = fw / cw, sy = fh / ch;
  float Kn[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) Kn[k] = K[k];
  Kn[0] = sx * K[0];
  Kn[4] = sy * K[4];
  Kn[2] = (fw - 1.0f) / 2.0f + sx * ocx;
  Kn[5] = (fh - 1.0f) / 2.0f + sy * ocy;
  if (bad_id) {
    const float nan = __builtin_nanf("");
#pragma unroll
    for (int k = 0; k < 9; ++k) Kn[k] = nan;
#pragma unroll
    for (int k = 0; k < 16; ++k) { T[k] = nan; TV[k] = nan; }
    x1 = y1 = x2 = y2 = bx1 = by1 = bx2 = by2 = nan;
  }
  if (slot >= 0) {
    float* Ko = a.K_crop + 9 * ((int64_t)i * a.n_views + slot);
#pragma unroll
    for (int k = 0; k < 9; ++k) Ko[k] = Kn[k];
    if (a.TCV_O) {
#pragma unroll
      for (int k = 0; k < 16; ++k) a.TCV_O[16 * ((int64_t)i * a.n_views + slot) + k] = TV[k];
    }
  }
  if (v == 0 && a.K_crop_main) {
#pragma unroll
    for (int k = 0; k < 9; ++k) a.K_crop_main[9 * (int64_t)i + k] = Kn[k];
  }

  if (v == 0) {
    if (a.TCO_out) {
#pragma unroll
      for (int k = 0; k < 16; ++k) a.TCO_out[16 * (int64_t)i + k] = T[k];
    }
    if (a.tCR) { a.tCR[3 * i] = T[3]; a.tCR[3 * i + 1] = T[7]; a.tCR[3 * i + 2] = T[11]; }
    if (a.boxes_rend) { float* o = a.boxes_rend + 4 * (int64_t)i; o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; }
    if (a.boxes_crop) { float* o = a.boxes_crop + 4 * (int64_t)i; o[0] = bx1; o[1] = by1; o[2] = bx2; o[3] = by2; }
  }
}

// ---- pose update ---------------------------------------------------------------------
__global__ void pose_update_kernel(int b, const float* TCO, const float* K_crop, int k_stride,
                                   const float* pose9, const float* tCRp, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  const float* T = TCO + 16 * (int64_t)i;
  const float* p = pose9 + 9 * (int64_t)i;
  const float* Kc = K_crop + (int64_t)k_stride * i;
  // compute_rotation_matrix_from_ortho6d (TB/lib3d/rotations.py:22-36)
  float nx = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  float x[3] = {p[0] / nx, p[1] / nx, p[2] / nx};
  float z[3] = {x[1] * p[5] - x[2] * p[4], x[2] * p[3] - x[0] * p[5], x[0] * p[4] - x[1] * p[3]};
  float nz = sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
  z[0] /= nz; z[1] /= nz; z[2] /= nz;
  float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
  float dR[9] = {x[0], y[0], z[0], x[1], y[1], z[1], x[2], y[2], z[2]};
  // pose_update_with_reference_point (TB/lib3d/cosypose_ops.py:34-62)
  float tc[3] = {T[3], T[7], T[11]};
  float tr[3] = {tc[0], tc[1], tc[2]};
  if (tCRp) { tr[0] = tCRp[3 * i]; tr[1] = tCRp[3 * i + 1]; tr[2] = tCRp[3 * i + 2]; }
  float zsrc = tr[2];
  float ztgt = p[8] * zsrc;
  float ox = (p[6] / Kc[0] + tr[0] / zsrc) * ztgt;
  float oy = (p[7] / Kc[4] + tr[1] / zsrc) * ztgt;
  float d[3] = {tc[0] - tr[0], tc[1] - tr[1], tc[2] - tr[2]};
  float* O = out + 16 * (int64_t)i;
  float tro[3] = {ox, oy, ztgt};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      O[4 * r + c] = fmaf(dR[3 * r + 2], T[8 + c], fmaf(dR[3 * r + 1], T[4 + c], dR[3 * r] * T[c]));
    O[4 * r + 3] = fmaf(dR[3 * r + 2], d[2], fmaf(dR[3 * r + 1], d[1], dR[3 * r] * d[0])) + tro[r];
  }
  O[12] = T[12]; O[13] = T[13]; O[14] = T[14]; O[15] = T[15];
}

// ---- coarse initialisation -----------------------------------------------------------
struct InitArgs {
  const float* points; int n_pad; int n;
  const float* boxes; const int32_t* box_ids; const float* K; const int32_t* im_ids;
  const int32_t* obj_ids; const float* R; const int32_t* rot_ids; float* out;
  const int32_t* point_ids; int n_points;  // optional deterministic sub-sample (null = all n_pad)
  int n_boxes, n_images, n_rots, n_obj;    // table sizes: an id outside its table gives a NaN pose
};

__global__ __launch_bounds__(kT) void tco_init_kernel(InitArgs a) {
  __shared__ float red[4][4];
  const int i = blockIdx.x, tid = threadIdx.x;
  const int box_id = a.box_ids ? a.box_ids[i] : i, im_id = a.im_ids[i], rot_id = a.rot_ids ? a.rot_ids[i] : i, ob_id = a.obj_ids[i];
  const bool bad_id = (unsigned)box_id >= (unsigned)a.n_boxes || (unsigned)im_id >= (unsigned)a.n_images ||
                      (a.R && (unsigned)rot_id >= (unsigned)a.n_rots) || (unsigned)ob_id >= (unsigned)a.n_obj;
  const float* box = a.boxes + 4 * (int64_t)(bad_id ? 0 : box_id);
  const float* K = a.K + 9 * (int64_t)(bad_id ? 0 : im_id);
  float R[9] = {0, 1, 0, 0, 0, -1, -1, 0, 0};  // z-up canonical orientation (:196-201)
  if (a.R) {
    const float* Rp = a.R + 9 * (int64_t)(bad_id ? 0 : rot_id);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = Rp[k];
  }
  const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  const float bcx = (box[0] + box[2]) / 2.0f, bcy = (box[1] + box[3]) / 2.0f;
  const float z_guess = 1.0f;
  float tx = ((bcx - cx) * z_guess) / fx, ty = ((bcy - cy) * z_guess) / fy;
  const float* pts = a.points + (int64_t)(bad_id ? 0 : ob_id) * a.n_pad * 3;
  float x1 = INFINITY, y1 = INFINITY, x2 = -INFINITY, y2 = -INFINITY;
  const int npts = a.point_ids ? a.n_points : a.n_pad;
  for (int j = tid; j < npts; j += kT) {
    const float* p = pts + 3 * (int64_t)(a.point_ids ? a.point_ids[j] : j);
    float X = fmaf(R[2], p[2], fmaf(R[1], p[1], R[0] * p[0])) + tx;
    float Y = fmaf(R[5], p[2], fmaf(R[4], p[1], R[3] * p[0])) + ty;
    x1 = fminf(x1, X); x2 = fmaxf(x2, X); y1 = fminf(y1, Y); y2 = fmaxf(y2, Y);
  }
  x1 = wave_min(x1); y1 = wave_min(y1); x2 = wave_max(x2); y2 = wave_max(y2);
  if ((tid & 63) == 0) { red[tid >> 6][0] = x1; red[tid >> 6][1] = y1; red[tid >> 6][2] = x2; red[tid >> 6][3] = y2; }
  __syncthreads();
  if (tid != 0) return;
#pragma unroll
  for (int wv = 1; wv < 4; ++wv) {
    x1 = fminf(x1, red[wv][0]); y1 = fminf(y1, red[wv][1]);
    x2 = fmaxf(x2, red[wv][2]); y2 = fmaxf(y2, red[wv][3]);
  }
  float bdx = (box[2] - box[0]) + 1.0f, bdy = (box[3] - box[1]) + 1.0f;
  float zdx = fx * (x2 - x1) / bdx, zdy = fy * (y2 - y1) / bdy;
  float z = (zdy + zdx) / 2.0f;
  float* O = a.out + 16 * (int64_t)i;
  O[0] = R[0]; O[1] = R[1]; O[2] = R[2]; O[3] = ((bcx - cx) * z) / fx;
  O[4] = R[3]; O[5] = R[4]; O[6] = R[5]; O[7] = ((bcy - cy) * z) / fy;
  O[8] = R[6]; O[9] = R[7]; O[10] = R[8]; O[11] = z;
  O[12] = 0.f; O[13] = 0.f; O[14] = 0.f; O[15] = 1.f;
  if (bad_id) {
#pragma unroll
    for (int k = 0; k < 16; ++k) O[k] = __builtin_nanf("");
  }
}

}  // namespace hp

static int pose_prep_impl(const hp_mesh_store* store, int b, int n_views, int multiview_type, int remove_tco,
                          int normalize, const float* d_TCO_in, const float* d_K, int n_images,
                          const int32_t* d_im_ids, const int32_t* d_obj_ids,
                          const int32_t* d_point_ids_main, int n_points_main,
                          const int32_t* d_point_ids_extra, int n_points_extra, int im_h,
                          int im_w, int crop_h, int crop_w, float lamb, float* d_TCO_out,
                          float* d_tCR, float* d_TCV_O, float* d_boxes_rend,
                          float* d_boxes_crop, float* d_K_crop, float* d_K_crop_main, void* stream) {
  using namespace hp;
  HP_REQUIRE(store && store->points, "hp_pose_prep: mesh store has no point table");
  HP_REQUIRE(b >= 0 && n_images >= 1, "hp_pose_prep: negative batch / no intrinsics");
  const int nvt = n_views + (remove_tco ? 1 : 0);  // the identity view is evaluated either way
  HP_REQUIRE((multiview_type == 0 && nvt == 1) || (multiview_type == 1 && nvt == 2) ||
                 (multiview_type == 3 && nvt == 4) || (multiview_type == 5 && nvt == 6),
             "hp_pose_prep: n_views does not match multiview_type");
  HP_REQUIRE(!remove_tco || (n_views >= 2 && d_K_crop_main), "hp_pose_prep: remove_TCO_rendering needs >= 2 rendered views and d_K_crop_main");
  if (b == 0) return HP_OK;
  HP_REQUIRE(d_TCO_in && d_K && d_im_ids && d_obj_ids && d_K_crop, "hp_pose_prep: null input");
  HP_REQUIRE(d_point_ids_main && n_points_main > 0 && n_points_main <= store->n_pad,
             "hp_pose_prep: n_points must be in (0, n_pad]");
  HP_REQUIRE(nvt == 1 || (d_point_ids_extra && n_points_extra > 0 && n_points_extra <= store->n_pad),
             "hp_pose_prep: extra-view point ids missing");
  PrepArgs a{store->points, store->n_pad, d_TCO_in, d_K, d_im_ids, d_obj_ids,
             d_point_ids_main, n_points_main, d_point_ids_extra, n_points_extra,
             b, n_views, normalize, im_h, im_w, crop_h, crop_w, lamb, n_images, store->n_obj,
             d_TCO_out, d_tCR, d_TCV_O, d_boxes_rend, d_boxes_crop, d_K_crop, remove_tco ? 1 : 0, d_K_crop_main};
  hipLaunchKernelGGL(pose_prep_kernel, dim3(b * nvt), dim3(kT), 0, (hipStream_t)stream, a);
  return check_launch("pose_prep_kernel");
}

extern "C" int hp_pose_prep(const hp_mesh_store* store, int b, int n_views, int multiview_type,
                            int normalize, const float* d_TCO_in, const float* d_K, int n_images,
                            const int32_t* d_im_ids, const int32_t* d_obj_ids,
                            const int32_t* d_point_ids_main, int n_points_main,
                            const int32_t* d_point_ids_extra, int n_points_extra, int im_h,
                            int im_w, int crop_h, int crop_w, float lamb, float* d_TCO_out,
                            float* d_tCR, float* d_TCV_O, float* d_boxes_rend,
                            float* d_boxes_crop, float* d_K_crop, void* stream) {
  return pose_prep_impl(store, b, n_views, multiview_type, 0, normalize, d_TCO_in, d_K, n_images, d_im_ids, d_obj_ids,
                        d_point_ids_main, n_points_main, d_point_ids_extra, n_points_extra, im_h, im_w, crop_h, crop_w, lamb,
                        d_TCO_out, d_tCR, d_TCV_O, d_boxes_rend, d_boxes_crop, d_K_crop, nullptr, stream);
}

extern "C" int hp_pose_prep_views(const hp_mesh_store* store, int b, int n_views, int multiview_type, int remove_tco_rendering,
                                  int normalize, const float* d_TCO_in, const float* d_K, int n_images,
                                  const int32_t* d_im_ids, const int32_t* d_obj_ids,
                                  const int32_t* d_point_ids_main, int n_points_main,
                                  const int32_t* d_point_ids_extra, int n_points_extra, int im_h,
                                  int im_w, int crop_h, int crop_w, float lamb, float* d_TCO_out,
                                  float* d_tCR, float* d_TCV_O, float* d_boxes_rend,
                                  float* d_boxes_crop, float* d_K_crop, float* d_K_crop_main, void* stream) {
  return pose_prep_impl(store, b, n_views, multiview_type, remove_tco_rendering, normalize, d_TCO_in, d_K, n_images, d_im_ids,
                        d_obj_ids, d_point_ids_main, n_points_main, d_point_ids_extra, n_points_extra, im_h, im_w, crop_h, crop_w,
                        lamb, d_TCO_out, d_tCR, d_TCV_O, d_boxes_rend, d_boxes_crop, d_K_crop, d_K_crop_main, stream);
}

extern "C" int hp_pose_update(int b, const float* d_TCO, const float* d_K_crop, int k_stride,
                              const float* d_pose9, const float* d_tCR, float* d_TCO_out,
                              void* stream) {
  using namespace hp;
  HP_REQUIRE(b >= 0 && k_stride >= 9, "hp_pose_update: bad sizes");
  if (b == 0) return HP_OK;
  HP_REQUIRE(d_TCO && d_K_crop && d_pose9 && d_TCO_out, "hp_pose_update: null pointer");
  if (b == 0) return HP_OK;
  hipLaunchKernelGGL(pose_update_kernel, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, b,
                     d_TCO, d_K_crop, k_stride, d_pose9, d_tCR, d_TCO_out);
  return check_launch("pose_update_kernel");
}

extern "C" int hp_tco_init_autodepth(const hp_mesh_store* store, int n, const float* d_boxes, int n_boxes,
                                     const int32_t* d_box_ids, const float* d_K, int n_images,
                                     const int32_t* d_im_ids, const int32_t* d_obj_ids,
                                     const float* d_R, int n_rots, const int32_t* d_rot_ids,
                                     const int32_t* d_point_ids, int n_points, float* d_TCO_out,
                                     void* stream) {
  using namespace hp;
  HP_REQUIRE(store && store->points, "hp_tco_init_autodepth: mesh store has no point table");
  HP_REQUIRE(n >= 0, "hp_tco_init_autodepth: negative count");
  HP_REQUIRE(n_boxes >= 1 && n_images >= 1 && (!d_R || n_rots >= 1), "hp_tco_init_autodepth: empty box / intrinsics / rotation table");
  if (n == 0) return HP_OK;
  HP_REQUIRE(d_boxes && d_K && d_im_ids && d_obj_ids && d_TCO_out, "hp_tco_init_autodepth: null input");
  HP_REQUIRE(!d_point_ids || (n_points > 0 && n_points <= store->n_pad),
             "hp_tco_init_autodepth: n_points must be in (0, n_pad]");
  InitArgs a{store->points, store->n_pad, n, d_boxes, d_box_ids, d_K, d_im_ids, d_obj_ids,
             d_R, d_rot_ids, d_TCO_out, d_point_ids, n_points, n_boxes, n_images, n_rots, store->n_obj};
  hipLaunchKernelGGL(tco_init_kernel, dim3(n), dim3(kT), 0, (hipStream_t)stream, a);
  return check_launch("tco_init_kernel");
}
