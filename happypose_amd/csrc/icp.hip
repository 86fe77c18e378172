// Depth refinement: point-to-plane ICP between the depth rendered at the predicted pose and the
// measured depth map -- the step right after the refiner loop (run_depth_refiner=True,
// MP/inference/pose_estimator.py:404-410).
//
// Reference: MP/inference/icp_refiner.py:135-303 + refiner_utils.py:27-53.  Its pipeline is kept:
//   mask      = rendered > 0 & measured > 0 & |measured - rendered| <= 0.1 ("threshold" mask), or a
//               caller-supplied mask; & 0.2 < measured < 5                      (:163-166, :282-290)
//   points    = back-projection with getXYZ's integer pixel table int16(u - cx), int16(v - cy)
//               (:111-132); fewer than n_min_points on either side -> the pose is kept (:187-188)
//   start     = translate by the difference of the two centroids                 (:192-197)
//   ICP       = rigid registration of the rendered points to the measured points; result applied
//               on the left of the pose; rejected when the residual exceeds the tolerance (:199-211)
// The registration itself is OpenCV's cv2.ppf_match_3d_ICP there (multi-level "picky" ICP on a
// FLANN kd-tree, with normals from an inpainted + Gaussian-filtered depth map) -- a third-party
// algorithm that is not in this image and cannot be restated bit for bit: PARITY UNPINNED.  Here it
// is a projective-association point-to-plane ICP (the standard GPU formulation):
//   * target table per image: depth smoothed with a hole-aware 5x5 binomial filter, points
//     X(u,v), normals from central differences of X (np.gradient spacing 2 as in get_normal);
//   * per iteration every source point p' = R p + t is projected into the image; its
//     correspondence is the target point q at that pixel, kept when valid and |q - p'| <= tol;
//     residual r = n.(q - p'), Jacobian [p' x n, n]; the 6x6 normal equations are accumulated in
//     two deterministic stages (per-block partial sums, then a fixed-order sum in fp64), solved by
//     Cholesky and the increment composed as a rotation vector;
//   * residual reported = RMS of r over the inliers of the last iteration.
// oracle/icp.py restates exactly this definition on the CPU; tests compare against it and check
// that known perturbations of a synthetic scene are recovered.
#include <cmath>
#include <vector>

#include "common.h"

namespace hp {
namespace {

constexpr int kBlocksPerView = 64;
constexpr int kAccum = 32;  // 21 (upper triangle of J^T J) + 6 (J^T r) + count + sum r^2 + 3 spare

struct IcpArgs {
  const float* depth_r;   // [n][H][W]
  const float* depth_m;   // [B][H][W]
  const uint8_t* masks;   // [B][H][W] or null
  const int32_t* im_ids;  // [n]
  const float* K;         // [n][9]
  const float* tgt;       // [B][H][W][6]  point + normal of the measured depth (point.z = 0: invalid)
  float* T;               // [n][12]       current increment, rows of [R | t]
  float* partial;         // [n][kBlocksPerView][kAccum]
  int n, H, W;
  float tol, delta_thresh;
  int mode;               // 0: centroids, 1: ICP normal equations
};

__device__ __forceinline__ float ipix(int u, float c) { return (float)(short)((float)u - c); }  // int16 table of getXYZ

// ---- target table: smoothed depth -> points and normals ---------------------------------
__global__ __launch_bounds__(256) void icp_target_kernel(const float* depth, const float* K, const int32_t* first_pred_of_image,
                                                         float* tgt, int B, int H, int W) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * H * W) return;
  const int b = (int)(idx / ((int64_t)H * W));
  const int rem = (int)(idx - (int64_t)b * H * W);
  const int v = rem / W, u = rem - v * W;
  const float* d = depth + (int64_t)b * H * W;
  float* out = tgt + idx * 6;
  const int kp = first_pred_of_image[b];
  const float d0 = d[rem];
  if (kp < 0 || !(d0 > 0.f)) {
#pragma unroll
    for (int q = 0; q < 6; ++q) out[q] = 0.f;
    return;
  }
  const float* k = K + (int64_t)kp * 9;
  const float fx = k[0], fy = k[4], cx = k[2], cy = k[5];
  const float wgt[5] = {1.f, 4.f, 6.f, 4.f, 1.f};
  auto smooth = [&](int uu, int vv) -> float {  // hole-aware 5x5 binomial filter
    float s = 0.f, ws = 0.f;
    for (int dy = -2; dy <= 2; ++dy) {
      const int y = vv + dy;
      if ((unsigned)y >= (unsigned)H) continue;
      for (int dx = -2; dx <= 2; ++dx) {
        const int x = uu + dx;
        if ((unsigned)x >= (unsigned)W) continue;
        const float z = d[y * W + x];
        if (z > 0.f) { const float w8 = wgt[dy + 2] * wgt[dx + 2]; s += w8 * z; ws += w8; }
      }
    }
    return ws > 0.f ? s / ws : 0.f;
  };
  auto point = [&](int uu, int vv, float z, float (&X)[3]) {
    X[0] = ipix(uu, cx) * z / fx; X[1] = ipix(vv, cy) * z / fy; X[2] = z;
  };
  // central differences (clamped at the border), spacing as np.gradient(depth, 2)
  const int ul = u > 0 ? u - 1 : u, ur = u < W - 1 ? u + 1 : u, vu = v > 0 ? v - 1 : v, vd = v < H - 1 ? v + 1 : v;
  float Xl[3], Xr[3], Xu[3], Xd[3];
  const float zl = smooth(ul, v), zr = smooth(ur, v), zu = smooth(u, vu), zd = smooth(u, vd);
  point(ul, v, zl, Xl); point(ur, v, zr, Xr); point(u, vu, zu, Xu); point(u, vd, zd, Xd);
  float n[3] = {0.f, 0.f, 0.f};
  if (zl > 0.f && zr > 0.f && zu > 0.f && zd > 0.f) {
    const float ax = Xr[0] - Xl[0], ay = Xr[1] - Xl[1], az = Xr[2] - Xl[2];
    const float bx = Xd[0] - Xu[0], by = Xd[1] - Xu[1], bz = Xd[2] - Xu[2];
    n[0] = ay * bz - az * by; n[1] = az * bx - ax * bz; n[2] = ax * by - ay * bx;
    const float nn = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (nn > 0.f) { n[0] /= nn; n[1] /= nn; n[2] /= nn; } else { n[0] = n[1] = n[2] = 0.f; }
  }
  float X[3];
  point(u, v, d0, X);
  out[0] = X[0]; out[1] = X[1]; out[2] = X[2]; out[3] = n[0]; out[4] = n[1]; out[5] = n[2];
}

// ---- one pass over the pixels of every prediction: partial sums per block ----------------
__global__ __launch_bounds__(256) void icp_accumulate_kernel(IcpArgs a) {
  const int n = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int HW = a.H * a.W;
  const int b = a.im_ids[n];
  const float* dr = a.depth_r + (int64_t)n * HW;
  const float* dm = a.depth_m + (int64_t)b * HW;
  const uint8_t* mk = a.masks ? a.masks + (int64_t)b * HW : nullptr;
  const float* tg = a.tgt + (int64_t)b * HW * 6;
  const float* k = a.K + (int64_t)n * 9;
  const float fx = k[0], fy = k[4], cx = k[2], cy = k[5];
  const float* T = a.T + (int64_t)n * 12;
  float acc[kAccum];
#pragma unroll
  for (int i = 0; i < kAccum; ++i) acc[i] = 0.f;
  const int per = (HW + kBlocksPerView - 1) / kBlocksPerView;
  const int p0 = blk * per, p1 = p0 + per < HW ? p0 + per : HW;
  for (int p = p0 + tid; p < p1; p += 256) {
    const float zr = dr[p], zm = dm[p];
    bool ok = zr > 0.f && zm > 0.2f && zm < 5.f;
    if (mk) ok = ok && mk[p] != 0;
    else ok = ok && fabsf(zm - zr) <= a.delta_thresh;
    if (!ok) continue;
    const int v = p / a.W, u = p - v * a.W;
    const float sx = ipix(u, cx) * zr / fx, sy = ipix(v, cy) * zr / fy, sz = zr;
    if (a.mode == 0) {  // centroids of the two point sets (same pixels)
      acc[0] += sx; acc[1] += sy; acc[2] += sz;
      acc[3] += ipix(u, cx) * zm / fx; acc[4] += ipix(v, cy) * zm / fy; acc[5] += zm;
      acc[27] += 1.f;
      continue;
    }
    const float px = T[0] * sx + T[1] * sy + T[2] * sz + T[3];
    const float py = T[4] * sx + T[5] * sy + T[6] * sz + T[7];
    const float pz = T[8] * sx + T[9] * sy + T[10] * sz + T[11];
    if (!(pz > 0.f)) continue;
    const int uu = (int)rintf(fx * px / pz + cx), vv = (int)rintf(fy * py / pz + cy);
    if ((unsigned)uu >= (unsigned)a.W || (unsigned)vv >= (unsigned)a.H) continue;
    const int q = vv * a.W + uu;
    const float* t6 = tg + (int64_t)q * 6;
    const float qz = t6[2];
    const float nx = t6[3], ny = t6[4], nz = t6[5];
    if (!(qz > 0.2f && qz < 5.f) || (nx == 0.f && ny == 0.f && nz == 0.f)) continue;
    if (mk ? mk[q] == 0 : !(dr[q] > 0.f && fabsf(dm[q] - dr[q]) <= a.delta_thresh)) continue;  // q is in the target set
    const float ex = t6[0] - px, ey = t6[1] - py, ez = qz - pz;
    if (ex * ex + ey * ey + ez * ez > a.tol * a.tol) continue;
    const float r = nx * ex + ny * ey + nz * ez;
    const float J[6] = {py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx, nx, ny, nz};
    int o = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) acc[o++] += J[i] * J[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] += J[i] * r;
    acc[27] += 1.f;
    acc[28] += r * r;
  }
  // block reduction in a fixed order: lanes by shuffle, then the 4 waves through LDS
  __shared__ float red[4][kAccum];
#pragma unroll
  for (int i = 0; i < kAccum; ++i) {
    float s = acc[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((tid & 63) == 0) red[tid >> 6][i] = s;
  }
  __syncthreads();
  if (tid < kAccum) a.partial[((int64_t)n * kBlocksPerView + blk) * kAccum + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// ---- per prediction: fixed-order sum of the partials (fp64), solve, update the increment ----
__global__ void icp_update_kernel(const float* partial, float* T, float* stats, int n_pred, int mode, int n_min_points) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_pred) return;
  double s[kAccum];
  for (int i = 0; i < kAccum; ++i) s[i] = 0.0;
  for (int b = 0; b < kBlocksPerView; ++b)
    for (int i = 0; i < kAccum; ++i) s[i] += (double)partial[((int64_t)n * kBlocksPerView + b) * kAccum + i];
  float* Tn = T + (int64_t)n * 12;
  float* st = stats + (int64_t)n * 4;  // {count of the start set, inliers, rms residual, failed}
  const double cnt = s[27];
  if (mode == 0) {
    st[0] = (float)cnt;
    st[3] = cnt < (double)n_min_points ? 1.f : 0.f;
    const double inv = cnt > 0 ? 1.0 / cnt : 0.0;
    for (int i = 0; i < 12; ++i) Tn[i] = (i % 5 == 0) ? 1.f : 0.f;  // identity rotation
    Tn[3] = (float)((s[3] - s[0]) * inv); Tn[7] = (float)((s[4] - s[1]) * inv); Tn[11] = (float)((s[5] - s[2]) * inv);
    return;
  }
  st[1] = (float)cnt;
  st[2] = cnt > 0 ? (float)sqrt(s[28] / cnt) : -1.f;
  if (mode == 2 || st[3] != 0.f) return;  // mode 2: evaluate the residual of the final increment only
  if (cnt < 6.0) { st[3] = 1.f; return; }
  double A[6][6], bvec[6];
  int o = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) { A[i][j] = A[j][i] = s[o++]; }
  double tr = 0.0;
  for (int i = 0; i < 6; ++i) { bvec[i] = s[21 + i]; tr += A[i][i]; }
  for (int i = 0; i < 6; ++i) A[i][i] += 1e-9 * tr + 1e-12;
  // Cholesky A = L L^T, then forward / backward substitution
  double L[6][6] = {};
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double v = A[i][j];
      for (int kk = 0; kk < j; ++kk) v -= L[i][kk] * L[j][kk];
      if (i == j) { if (!(v > 0.0)) { st[3] = 1.f; return; } L[i][i] = sqrt(v); }
      else L[i][j] = v / L[j][j];
    }
  double y[6], x[6];
  for (int i = 0; i < 6; ++i) { double v = bvec[i]; for (int kk = 0; kk < i; ++kk) v -= L[i][kk] * y[kk]; y[i] = v / L[i][i]; }
  for (int i = 5; i >= 0; --i) { double v = y[i]; for (int kk = i + 1; kk < 6; ++kk) v -= L[kk][i] * x[kk]; x[i] = v / L[i][i]; }
  // increment: rotation vector x[0:3] (Rodrigues), translation x[3:6]; T <- dT * T
  const double th = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  if (th > 1e-12) {
    const double kx = x[0] / th, ky = x[1] / th, kz = x[2] / th, c = cos(th), sn = sin(th), v1 = 1.0 - c;
    R[0][0] = c + kx * kx * v1;      R[0][1] = kx * ky * v1 - kz * sn; R[0][2] = kx * kz * v1 + ky * sn;
    R[1][0] = ky * kx * v1 + kz * sn; R[1][1] = c + ky * ky * v1;      R[1][2] = ky * kz * v1 - kx * sn;
    R[2][0] = kz * kx * v1 - ky * sn; R[2][1] = kz * ky * v1 + kx * sn; R[2][2] = c + kz * kz * v1;
  }
  double Tn_new[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j) {
      double v = 0.0;
      for (int kk = 0; kk < 3; ++kk) v += R[i][kk] * (double)Tn[kk * 4 + j];
      Tn_new[i * 4 + j] = v;
    }
    Tn_new[i * 4 + 3] += x[3 + i];
  }
  for (int i = 0; i < 12; ++i) Tn[i] = (float)Tn_new[i];
}

// ---- result: T_icp * TCO where the registration succeeded, TCO otherwise -------------------
__global__ void icp_finalize_kernel(const float* T, const float* stats, const float* TCO, float* TCO_out, int32_t* retval,
                                    float* residual, int n_pred, float tol, int n_min_points) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_pred) return;
  const float* st = stats + (int64_t)n * 4;
  const bool ok = st[3] == 0.f && st[2] >= 0.f && st[2] <= tol && st[1] >= (float)n_min_points;
  const float* Tn = T + (int64_t)n * 12;
  const float* P = TCO + (int64_t)n * 16;
  float* O = TCO_out + (int64_t)n * 16;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float v;
      if (!ok) v = P[i * 4 + j];
      else if (i == 3) v = (j == 3) ? 1.f : 0.f;
      else v = Tn[i * 4 + 0] * P[0 * 4 + j] + Tn[i * 4 + 1] * P[1 * 4 + j] + Tn[i * 4 + 2] * P[2 * 4 + j] + (j == 3 ? Tn[i * 4 + 3] : 0.f);
      O[i * 4 + j] = v;
    }
  if (retval) retval[n] = ok ? 0 : -1;
  if (residual) residual[n] = st[2];
}

struct IcpWorkspace { float* tgt = nullptr; size_t tgt_bytes = 0; float* small = nullptr; size_t small_bytes = 0; int32_t* first = nullptr; size_t first_bytes = 0; };

template <typename T>
int grow(T** p, size_t* have, size_t need) {
  if (*have >= need) return HP_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *have = 0;
  HP_CHECK_HIP(hipMalloc((void**)p, need));
  *have = need;
  return HP_OK;
}

}  // namespace
}  // namespace hp

using namespace hp;

extern "C" int hp_icp_refine(int n, int B, int H, int W, const float* d_depth_rendered, const float* d_depth_measured,
                             const uint8_t* d_masks, const int32_t* d_im_ids, const int32_t* h_im_ids, const float* d_K,
                             const float* d_TCO, int n_iterations, int n_min_points, float tolerance,
                             float depth_delta_thresh, float* d_TCO_out, int32_t* d_retval, float* d_residual,
                             void* stream) {
  HP_REQUIRE(n >= 0 && B >= 1 && H > 0 && W > 0 && n_iterations >= 1, "hp_icp_refine: bad sizes");
  if (n == 0) return HP_OK;
  HP_REQUIRE(d_depth_rendered && d_depth_measured && d_im_ids && h_im_ids && d_K && d_TCO && d_TCO_out,
             "hp_icp_refine: null pointer");
  static IcpWorkspace ws;  // grown on demand; calls are expected on one stream per process
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const size_t HW = (size_t)H * W;
  if ((rc = grow(&ws.tgt, &ws.tgt_bytes, (size_t)B * HW * 6 * sizeof(float)))) return rc;
  const size_t small_floats = (size_t)n * (12 + 4 + (size_t)kBlocksPerView * kAccum);
  if ((rc = grow(&ws.small, &ws.small_bytes, small_floats * sizeof(float)))) return rc;
  if ((rc = grow(&ws.first, &ws.first_bytes, (size_t)B * sizeof(int32_t)))) return rc;
  // the intrinsics of an image are those of its first prediction (K is indexed per prediction)
  std::vector<int32_t> first(B, -1);
  for (int i = n - 1; i >= 0; --i) {
    HP_REQUIRE(h_im_ids[i] >= 0 && h_im_ids[i] < B, "hp_icp_refine: batch_im_id out of range");
    first[h_im_ids[i]] = i;
  }
  HP_CHECK_HIP(hipMemcpyAsync(ws.first, first.data(), (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, st));
  HP_CHECK_HIP(hipStreamSynchronize(st));  // `first` is a stack-lifetime host buffer
  float* T = ws.small;
  float* stats = T + (size_t)n * 12;
  float* partial = stats + (size_t)n * 4;
  const int64_t total = (int64_t)B * HW;
  hipLaunchKernelGGL(icp_target_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_depth_measured, d_K, ws.first,
                     ws.tgt, B, H, W);
  if ((rc = check_launch("icp_target_kernel"))) return rc;
  IcpArgs a{};
  a.depth_r = d_depth_rendered; a.depth_m = d_depth_measured; a.masks = d_masks; a.im_ids = d_im_ids; a.K = d_K;
  a.tgt = ws.tgt; a.T = T; a.partial = partial; a.n = n; a.H = H; a.W = W; a.tol = tolerance; a.delta_thresh = depth_delta_thresh;
  for (int it = 0; it <= n_iterations; ++it) {
    a.mode = it == 0 ? 0 : 1;  // pass 0: centroid start; the final pass only evaluates the residual
    hipLaunchKernelGGL(icp_accumulate_kernel, dim3(kBlocksPerView, n), dim3(256), 0, st, a);
    if ((rc = check_launch("icp_accumulate_kernel"))) return rc;
    hipLaunchKernelGGL(icp_update_kernel, dim3((n + 63) / 64), dim3(64), 0, st, partial, T, stats, n, a.mode, n_min_points);
    if ((rc = check_launch("icp_update_kernel"))) return rc;
  }
  // residual of the final increment
  a.mode = 1;
  hipLaunchKernelGGL(icp_accumulate_kernel, dim3(kBlocksPerView, n), dim3(256), 0, st, a);
  if ((rc = check_launch("icp_accumulate_kernel"))) return rc;
  hipLaunchKernelGGL(icp_update_kernel, dim3((n + 63) / 64), dim3(64), 0, st, partial, T, stats, n, 2, n_min_points);
  if ((rc = check_launch("icp_update_kernel"))) return rc;
  hipLaunchKernelGGL(icp_finalize_kernel, dim3((n + 63) / 64), dim3(64), 0, st, T, stats, d_TCO, d_TCO_out, d_retval, d_residual, n,
                     tolerance, n_min_points);
  return check_launch("icp_finalize_kernel");
}
