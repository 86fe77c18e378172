/*
 * CPU oracle (plain C) for the two pixel-producing stages of the render-and-compare
 * path.  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Built by oracle/Makefile
 * into oracle/_build/liboracle.so; loaded by oracle/native.py.
 *
 *  - hp_oracle_roi_align : torchvision 0.14.1 `roi_align` (aligned=False) as the
 *    reference calls it (TB/lib3d/cropping.py:155-197, CP/lib3d/cropping.py:129-134:
 *    output (240,320), sampling_ratio=4, spatial_scale=1).  PARITY UNPINNED (torchvision
 *    is not importable in the build container; restated from its published CPU kernel).
 *  - hp_oracle_rasterize : the DEFINITION of what the HIP rasteriser must output -- an
 *    analytic pinhole rasterisation of one textured mesh per view with the camera model,
 *    clip range, two-sidedness, depth decode, mask and eye-normal colour code of the
 *    reference's Panda3D renderer (TB/renderer/types.py:92-137,
 *    TB/renderer/utils.py:46-79, TB/renderer/panda3d_scene_renderer.py:59-141,221-230,
 *    320-390, TB/renderer/panda3d_batch_renderer.py:62-125,194-286).  PIXEL PARITY
 *    UNPINNED (Panda3D/OpenGL cannot run here and no reference test pins pixels).
 *    Known, documented deviations from Panda3D: one sample at the pixel centre instead of
 *    MSAAx4, bilinear level-0 texture filtering instead of trilinear-mipmap+aniso16.
 *
 * All arithmetic is float32 with explicit fmaf() so that the HIP kernels can follow the
 * same operation order; built with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* thread count of the OpenMP loops below (bench.py times the port at 1 thread and at all cores) */
void hp_oracle_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------------------------ */
/* roi_align                                                                            */
/* ------------------------------------------------------------------------------------ */
static inline float bilinear_tv(const float* plane, int H, int W, float y, float x) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.0f;
  if (y <= 0.0f) y = 0.0f;
  if (x <= 0.0f) x = 0.0f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  float ly = y - (float)y_low, lx = x - (float)x_low;
  float hy = 1.0f - ly, hx = 1.0f - lx;
  float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  return w1 * plane[y_low * W + x_low] + w2 * plane[y_low * W + x_high] +
         w3 * plane[y_high * W + x_low] + w4 * plane[y_high * W + x_high];
}

/* images [Bi][C][H][W]; boxes [n][4] (x1,y1,x2,y2); im_ids [n]; out [n][C][oh][ow]. */
void hp_oracle_roi_align(const float* images, int Bi, int C, int H, int W,
                         const float* boxes, const int32_t* im_ids, int n,
                         int oh, int ow, int sampling_ratio, float* out) {
  (void)Bi;
#pragma omp parallel for schedule(dynamic, 1)
  for (int r = 0; r < n; ++r) {
    const float x1 = boxes[4 * r + 0], y1 = boxes[4 * r + 1];
    const float x2 = boxes[4 * r + 2], y2 = boxes[4 * r + 3];
    float roi_w = x2 - x1, roi_h = y2 - y1;
    if (roi_w < 1.0f) roi_w = 1.0f; /* aligned=False */
    if (roi_h < 1.0f) roi_h = 1.0f;
    const float bin_h = roi_h / (float)oh, bin_w = roi_w / (float)ow;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / (float)oh);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / (float)ow);
    const float count = (float)(gh * gw > 1 ? gh * gw : 1);
    for (int c = 0; c < C; ++c) {
      const float* plane = images + ((size_t)im_ids[r] * C + c) * H * W;
      float* o = out + ((size_t)r * C + c) * oh * ow;
      for (int ph = 0; ph < oh; ++ph)
        for (int pw = 0; pw < ow; ++pw) {
          float acc = 0.0f;
          for (int iy = 0; iy < gh; ++iy) {
            const float y = y1 + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
              const float x = x1 + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
              acc += bilinear_tv(plane, H, W, y, x);
            }
          }
          o[ph * ow + pw] = acc / count;
        }
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* rasteriser                                                                           */
/* ------------------------------------------------------------------------------------ */
#define HP_R_NORMALS 1
#define HP_R_DEPTH 2
#define HP_R_MASK 4
#define HP_R_QUANT8 8 /* emulate the 8-bit framebuffer: round(c*255)/255 */
#define HP_R_TEX_ANISO 64 /* texture filtering of the reference's renderer (TB/renderer/panda3d_scene_renderer.py:68-69:
                           * "texture-minfilter mipmap", "texture-anisotropic-degree 16"): trilinear over the mip chain the
                           * store keeps behind level 0 + anisotropic sampling as EXT_texture_filter_anisotropic sketches
                           * it -- Px, Py = lengths of the texel-space derivatives along screen x / y, N = min(ceil(Pmax / Pmin),
                           * 16) probes along the major axis at LOD log2(Pmax / N), averaged.  NOT pinned: the extension
                           * leaves the footprint to the implementation and Panda3D is absent. */
#define HP_R_MSAA4 32 /* 4x multisampling of the colour / normal buffers (the reference's framebuffer state:
                       * TB/renderer/panda3d_scene_renderer.py:70-71 "framebuffer-multisample 1 / multisamples 4", buffers made
                       * from FrameBufferProperties.getDefault(), TB/renderer/types.py:207).  OpenGL semantics: coverage and
                       * depth are evaluated per SAMPLE, the fragment is shaded ONCE per pixel and primitive at the pixel
                       * centre (attributes extrapolated when the centre lies outside the primitive; no centroid qualifier in
                       * Panda3D's default shaders) and the 8-bit colour written to the samples it covers; the resolve
                       * averages the four samples (background = the clear colour 0) back into an 8-bit value.  Sample
                       * positions: the standard 4x pattern every current implementation uses (D3D's, also Mesa's and the
                       * vendors' GL): (0.375, 0.125), (0.875, 0.375), (0.125, 0.625), (0.625, 0.875).  NOT pinned: the sample
                       * positions are implementation-defined in OpenGL and Panda3D is absent here.  Depth and mask are
                       * left at the pixel centre (a fifth, non-averaged sample): every consumer back-projects pixel centres. */

/* The renderer conventions that cannot be pinned without Panda3D (include/happypose_amd.h: hp_raster_conventions -- the same
 * record, mirrored here so that kernel == oracle holds for EVERY candidate a calibration tries): multisample positions, the
 * anisotropic filter's probe-count / level-of-detail rule, and the axis / sign map of the eye-normal code.  Process-wide. */
typedef struct {
  float msaa_x[4], msaa_y[4];
  int aniso_max, aniso_round, lod_from;
  float lod_bias, aniso_ratio_bias;
  int normal_axis[3];
  float normal_sign[3];
} hp_oracle_raster_conventions;
static const hp_oracle_raster_conventions HP_ORACLE_CONV_DEFAULT = {{0.375f, 0.875f, 0.125f, 0.625f}, {0.125f, 0.375f, 0.625f, 0.875f},
                                                                     16, 0, 0, 0.0f, 0.0f, {0, 1, 2}, {1.0f, -1.0f, -1.0f}};
static hp_oracle_raster_conventions g_conv = {{0.375f, 0.875f, 0.125f, 0.625f}, {0.125f, 0.375f, 0.625f, 0.875f},
                                              16, 0, 0, 0.0f, 0.0f, {0, 1, 2}, {1.0f, -1.0f, -1.0f}};
void hp_oracle_set_raster_conventions(const hp_oracle_raster_conventions* c) { g_conv = c ? *c : HP_ORACLE_CONV_DEFAULT; }

typedef struct {
  const float* verts;    /* [Vtot][3] metres, object frame */
  const float* normals;  /* [Vtot][3] unit, object frame */
  const float* uvs;      /* [Vtot][2] (v up, OpenGL convention) */
  const uint8_t* colors; /* [Vtot][4] RGBA vertex colours (used when tex_off < 0) */
  const int32_t* faces;  /* [Ftot][3] vertex ids local to the object */
  const uint8_t* tex;    /* texture pool, RGBA8 row-major, row 0 = top */
  const int64_t* obj;    /* [n_obj][8]: vert_off n_verts face_off n_faces tex_off tex_w tex_h pad */
  int n_obj;
} hp_oracle_meshes;

#define Z_NEAR 0.1f
#define Z_FAR 10.0f
#define KEY_EMPTY 0xFFFFFFFFFFFFFFFFull

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float quant8(float c, int on) {
  c = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
  if (!on) return c;
  return floorf(fmaf(c, 255.0f, 0.5f)) / 255.0f;
}

/* value of the 32^3 "normal code" 3-D texture along one axis (it is separable): texel i
 * holds floor(i*255/32) (TB/renderer/utils.py:63-79), repeat wrap, linear filter. */
static inline float normal_code(float n) {
  float s = n - floorf(n);             /* repeat wrap */
  float x = fmaf(s, 32.0f, -0.5f);
  float xf = floorf(x);
  float f = x - xf;
  int i0 = ((int)xf + 32) & 31, i1 = (i0 + 1) & 31;
  float t0 = floorf((float)i0 * 255.0f / 32.0f), t1 = floorf((float)i1 * 255.0f / 32.0f);
  return fmaf(f, t1 - t0, t0) / 255.0f;
}

static inline void tex_fetch(const uint8_t* tex, int tw, int th, float u, float v, float* rgb) {
  float x = fmaf(u, (float)tw, -0.5f);
  float y = fmaf(1.0f - v, (float)th, -0.5f);
  float xf = floorf(x), yf = floorf(y);
  float fx = x - xf, fy = y - yf;
  int x0 = (int)xf % tw; if (x0 < 0) x0 += tw;
  int y0 = (int)yf % th; if (y0 < 0) y0 += th;
  int x1 = x0 + 1 == tw ? 0 : x0 + 1;
  int y1 = y0 + 1 == th ? 0 : y0 + 1;
  const uint8_t* p00 = tex + 4 * ((size_t)y0 * tw + x0);
  const uint8_t* p01 = tex + 4 * ((size_t)y0 * tw + x1);
  const uint8_t* p10 = tex + 4 * ((size_t)y1 * tw + x0);
  const uint8_t* p11 = tex + 4 * ((size_t)y1 * tw + x1);
  for (int c = 0; c < 3; ++c) {
    float a = fmaf(fx, (float)p01[c] - (float)p00[c], (float)p00[c]);
    float b = fmaf(fx, (float)p11[c] - (float)p10[c], (float)p10[c]);
    rgb[c] = fmaf(fy, b - a, a) / 255.0f;
  }
}

/* bilinear fetch of mip level `lvl` (level k is max(1, tw >> k) x max(1, th >> k), stored behind the levels before it) */
static inline void tex_fetch_level(const uint8_t* tex, int tw, int th, int lvl, float u, float v, float* rgb) {
  size_t off = 0;
  int w = tw, h = th;
  for (int k = 0; k < lvl; ++k) { off += (size_t)4 * w * h; w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1; }
  tex_fetch(tex + off, w, h, u, v, rgb);
}

/* trilinear + anisotropic fetch (HP_R_TEX_ANISO): (ux, vx) / (uy, vy) = d(u, v) / d(screen x) / d(screen y) */
static inline void tex_fetch_aniso(const uint8_t* tex, int tw, int th, int nlev, float u, float v, float ux, float vx,
                                   float uy, float vy, float* rgb) {
  const float px = sqrtf(fmaf(ux * (float)tw, ux * (float)tw, vx * (float)th * (vx * (float)th)));
  const float py = sqrtf(fmaf(uy * (float)tw, uy * (float)tw, vy * (float)th * (vy * (float)th)));
  const int along_x = px >= py;
  const float pmax = along_x ? px : py, pmin = along_x ? py : px;
  const float ratio = pmax / pmin + g_conv.aniso_ratio_bias, amax = (float)g_conv.aniso_max;
  float nf = pmin > 0.0f ? (g_conv.aniso_round == 0 ? ceilf(ratio) : g_conv.aniso_round == 1 ? rintf(ratio) : floorf(ratio)) : amax;
  if (!(nf >= 1.0f)) nf = 1.0f;   /* NaN / zero footprints */
  if (nf > amax) nf = amax;
  const int N = (int)nf;
  const float la = g_conv.lod_from == 0 ? pmax / nf : g_conv.lod_from == 1 ? pmin : pmax;
  float lod = la > 0.0f ? log2f(la) + g_conv.lod_bias : 0.0f;
  if (!(lod > 0.0f)) lod = 0.0f;  /* magnification: level 0 */
  if (lod > (float)(nlev - 1)) lod = (float)(nlev - 1);
  const int l0 = (int)lod;
  const float fl = lod - (float)l0;
  const float du = along_x ? ux : uy, dv = along_x ? vx : vy;
  float acc[3] = {0.0f, 0.0f, 0.0f};
  for (int i = 1; i <= N; ++i) {
    const float t = (float)i / (float)(N + 1) - 0.5f;
    const float su = fmaf(t, du, u), sv = fmaf(t, dv, v);
    float c0[3], c1[3];
    tex_fetch_level(tex, tw, th, l0, su, sv, c0);
    if (fl > 0.0f && l0 + 1 < nlev) {
      tex_fetch_level(tex, tw, th, l0 + 1, su, sv, c1);
      for (int c = 0; c < 3; ++c) c0[c] = fmaf(fl, c1[c] - c0[c], c0[c]);
    }
    for (int c = 0; c < 3; ++c) acc[c] += c0[c];
  }
  for (int c = 0; c < 3; ++c) rgb[c] = acc[c] / (float)N;
}

/*
 * ---- the rasteriser's geometry, round 5: the way OpenGL hardware defines it ----------------------------------------------
 * (replaces the homogeneous fp32 edge functions of rounds 1-4: their sample tests were order-dependent at round-off level,
 * could not be evaluated incrementally and needed an IEEE division per covered sample.)
 *
 *  1. Vertices: camera coordinates by the fmaf chains below, w = 1 / z, window coordinates (Xh w, Yh w) SNAPPED to a
 *     1/256-pixel grid (rintf, clamped to a guard band of +-16384 px): int32 (x, y).  Two vertices at the same position snap
 *     to the same point whatever their index: texture seams are watertight.
 *  2. Near plane: a triangle with a corner in front of z = 0.1 m is CLIPPED against that plane (1 or 2 sub-triangles; the new
 *     corners are interpolated from the inside corner to the outside one, so both triangles of a shared edge compute the same
 *     point); corners carry their barycentric coordinates with respect to the original triangle.  Far plane and the image
 *     borders need no clipping: the depth test rejects beyond-range samples and only image pixels are visited.
 *  3. Coverage: EXACT integer edge functions E_k(X, Y) = (x_b - x_a)(Y - y_a) - (y_b - y_a)(X - x_a) on the snapped corners
 *     (triangle orientation normalised so that the interior is the positive side), sample positions on the same 1/256 grid,
 *     top-left fill rule: a sample ON an edge belongs to the triangle iff the edge vector (ex, ey) has ey > 0 or (ey == 0
 *     and ex < 0) -- every point of a shared edge belongs to exactly one of the two triangles, whatever their winding.
 *  4. Depth: w = 1 / z is AFFINE in window coordinates (that is what a hardware depth buffer interpolates); the plane
 *     W(x, y) through the three corners is evaluated per sample, the depth test keeps the largest W in [1 / far, 1 / near]
 *     (64-bit key {~bits(W) : sub-triangle id}: ties go to the lower id, independent of submission order), and the metric
 *     depth of a pixel is 1 / W at its centre (TB/renderer/utils.py:46-60 decodes the same quantity from the depth buffer).
 *  5. Attributes: perspective-correct interpolation as planes as well -- for an attribute q the plane N_q through
 *     (corner, q_k w_k); q = N_q / W.  u, v (texture coordinates) and the barycentric coordinates b1, b2 (for normals and
 *     vertex colours) are interpolated this way; du/dx = (N_u,x - u W_x) / W etc. for the anisotropic footprint.
 *
 * obj_ids [n]; TCO [n][16]; K [n][9]; ambient [n][3]; n_lights point lights per view:
 * light_pos [n][n_lights][3] (object frame, metres), light_col [n][n_lights][3].
 * Outputs are addressed as base + view*sv + chan*sc + row*sr + col*sp (element strides), so
 * NCHW (sc=h*w, sr=w, sp=1) and NHWC slices are both expressible.  Any output pointer may
 * be NULL.
 */
#define SUBPIX 256
#define GUARD_SUB 4194304.0f /* 2^22 sub-pixel units = 16384 px */

typedef struct { float c[3]; float b[3]; } clipv;             /* camera-space corner + barycentrics w.r.t. the original triangle */
typedef struct { float q0, qx, qy; } plane;                   /* q(fx, fy) = q0 + qx fx + qy fy, (fx, fy) = pixels from the origin pixel's corner */
typedef struct {
  int64_t id;                                                 /* f, or n_faces + f for the second half of a clipped quad */
  int32_t v[3];                                               /* the original triangle's vertex ids (object-local) */
  int64_t rx[3], ry[3];                                       /* snapped corners relative to (256 ox, 256 oy), orientation normalised */
  int ox, oy;                                                 /* origin pixel of the planes / relative coordinates */
  int j0, i0, j1, i1;                                         /* candidate pixels: columns j0..j1, rows i0..i1 (inside the image) */
  plane W, NU, NV, NB1, NB2;
} subtri;

static inline int32_t snap_sub(float s) {
  float r = s * (float)SUBPIX;
  if (!(r >= -GUARD_SUB)) r = -GUARD_SUB; /* NaN lands here */
  if (!(r <= GUARD_SUB)) r = GUARD_SUB;
  return (int32_t)rintf(r);
}
static inline int64_t floor_div(int64_t a, int64_t b) { int64_t q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return -floor_div(-a, b); }
static inline float plane_at(const plane* p, float fx, float fy) { return fmaf(p->qy, fy, fmaf(p->qx, fx, p->q0)); }

/* plane through the corners' values q[0..2]; (px, py) = corners in pixels from the origin, inv = 1 / (2 area) */
static inline plane make_plane(const float* q, const float* px, const float* py, float inv) {
  const float dx1 = px[1] - px[0], dy1 = py[1] - py[0], dx2 = px[2] - px[0], dy2 = py[2] - py[0];
  const float dq1 = q[1] - q[0], dq2 = q[2] - q[0];
  plane p;
  p.qx = fmaf(dq1, dy2, -(dq2 * dy1)) * inv;
  p.qy = fmaf(dq2, dx1, -(dq1 * dx2)) * inv;
  p.q0 = fmaf(-p.qy, py[0], fmaf(-p.qx, px[0], q[0]));
  return p;
}

/* Set-up of one sub-triangle from three camera-space corners (all with z >= near).  Returns 0 when nothing can be covered. */
static int setup_subtri(const clipv* cv3, const float* Kv, const float* uv3 /* [3][2] of the ORIGINAL vertices */, int h, int w,
                        int lo_x, int hi_x, int lo_y, int hi_y, subtri* s) {
  int64_t x[3], y[3];
  float wk[3], bb[3][3];
  for (int k = 0; k < 3; ++k) {
    const float cx = cv3[k].c[0], cy = cv3[k].c[1], cz = cv3[k].c[2];
    const float Xh = fmaf(Kv[0], cx, fmaf(Kv[1], cy, Kv[2] * cz));
    const float Yh = fmaf(Kv[4], cy, Kv[5] * cz);
    wk[k] = 1.0f / cz;
    x[k] = snap_sub(Xh * wk[k]);
    y[k] = snap_sub(Yh * wk[k]);
    for (int c = 0; c < 3; ++c) bb[k][c] = cv3[k].b[c];
  }
  int64_t area2 = (x[1] - x[0]) * (y[2] - y[0]) - (x[2] - x[0]) * (y[1] - y[0]);
  if (area2 == 0) return 0;
  if (area2 < 0) { /* normalise the orientation: swap corners 1 and 2 */
    int64_t t = x[1]; x[1] = x[2]; x[2] = t; t = y[1]; y[1] = y[2]; y[2] = t;
    float tf = wk[1]; wk[1] = wk[2]; wk[2] = tf;
    for (int c = 0; c < 3; ++c) { tf = bb[1][c]; bb[1][c] = bb[2][c]; bb[2][c] = tf; }
    area2 = -area2;
  }
  int64_t xmin = x[0], xmax = x[0], ymin = y[0], ymax = y[0];
  for (int k = 1; k < 3; ++k) {
    if (x[k] < xmin) xmin = x[k];
    if (x[k] > xmax) xmax = x[k];
    if (y[k] < ymin) ymin = y[k];
    if (y[k] > ymax) ymax = y[k];
  }
  /* origin pixel of the sub-triangle's planes and relative coordinates: the pixel that holds the bounding box's minimum,
   * clamped into the image -- independent of the sample pattern, so single-sample and multisampled renders see the same
   * planes (and the same centre depths) */
  int64_t j0 = floor_div(xmin, SUBPIX), i0 = floor_div(ymin, SUBPIX);
  if (j0 < 0) j0 = 0;
  if (j0 > w - 1) j0 = w - 1;
  if (i0 < 0) i0 = 0;
  if (i0 > h - 1) i0 = h - 1;
  /* candidate pixels: pixel j has a sample with xmin <= 256 j + s <= xmax for some sample offset s in [lo, hi] */
  int64_t ja = ceil_div(xmin - hi_x, SUBPIX), j1 = floor_div(xmax - lo_x, SUBPIX);
  int64_t ia = ceil_div(ymin - hi_y, SUBPIX), i1 = floor_div(ymax - lo_y, SUBPIX);
  if (ja < j0) ja = j0;
  if (j1 > w - 1) j1 = w - 1;
  if (ia < i0) ia = i0;
  if (i1 > h - 1) i1 = h - 1;
  if (ja > j1 || ia > i1) return 0;
  s->ox = (int)j0; s->oy = (int)i0; s->j0 = (int)ja; s->i0 = (int)ia; s->j1 = (int)j1; s->i1 = (int)i1;
  float px[3], py[3];
  for (int k = 0; k < 3; ++k) {
    s->rx[k] = x[k] - (int64_t)SUBPIX * j0;
    s->ry[k] = y[k] - (int64_t)SUBPIX * i0;
    px[k] = (float)s->rx[k] * (1.0f / (float)SUBPIX);   /* exact: |rx| < 2^24 */
    py[k] = (float)s->ry[k] * (1.0f / (float)SUBPIX);
  }
  const float det = (float)area2 * (1.0f / (float)(SUBPIX * SUBPIX)); /* (float)area2: correctly rounded */
  const float inv = 1.0f / det;
  float q[3];
  s->W = make_plane(wk, px, py, inv);
  for (int k = 0; k < 3; ++k) q[k] = fmaf(bb[k][0], uv3[0], fmaf(bb[k][1], uv3[2], bb[k][2] * uv3[4])) * wk[k];
  s->NU = make_plane(q, px, py, inv);
  for (int k = 0; k < 3; ++k) q[k] = fmaf(bb[k][0], uv3[1], fmaf(bb[k][1], uv3[3], bb[k][2] * uv3[5])) * wk[k];
  s->NV = make_plane(q, px, py, inv);
  for (int k = 0; k < 3; ++k) q[k] = bb[k][1] * wk[k];
  s->NB1 = make_plane(q, px, py, inv);
  for (int k = 0; k < 3; ++k) q[k] = bb[k][2] * wk[k];
  s->NB2 = make_plane(q, px, py, inv);
  return 1;
}

/* sample (X, Y) (sub-pixel units from the origin) inside the sub-triangle: exact integers + top-left rule */
static inline int sample_inside(const subtri* s, int64_t X, int64_t Y) {
  for (int k = 0; k < 3; ++k) {
    const int a = (k + 1) % 3, b = (k + 2) % 3;
    const int64_t ex = s->rx[b] - s->rx[a], ey = s->ry[b] - s->ry[a];
    const int64_t E = ex * (Y - s->ry[a]) - ey * (X - s->rx[a]);
    const int owns = ey > 0 || (ey == 0 && ex < 0);
    if (E < 0 || (E == 0 && !owns)) return 0;
  }
  return 1;
}

/* One fragment-shader invocation: colour and normal code of sub-triangle s at the CENTRE of pixel (i, j) -- albedo (texture or
 * vertex colours) x (ambient + Lambert point lights), eye-space normal code; attributes from the planes (extrapolated when
 * the centre lies outside the triangle: multisampled edge pixels). */
static void shade_centre(const hp_oracle_meshes* M, const subtri* s, const float* T, const float* Kv, const float* amb,
                         int n_lights, const float* light_pos, const float* light_col, int view, int64_t voff,
                         int64_t toff, int tw, int th, int nlev, int aniso, int i, int j, int q8, float* o_rgb,
                         float* o_n) {
  const float fx = (float)(j - s->ox) + 0.5f, fy = (float)(i - s->oy) + 0.5f;
  const float Wc = plane_at(&s->W, fx, fy);
  const float iw = 1.0f / Wc;
  const float b1 = plane_at(&s->NB1, fx, fy) * iw, b2 = plane_at(&s->NB2, fx, fy) * iw;
  const float b0 = (1.0f - b1) - b2;
  const int64_t g0 = voff + s->v[0], g1 = voff + s->v[1], g2 = voff + s->v[2];
  float alb[3];
  if (toff >= 0) {
    const float tu = plane_at(&s->NU, fx, fy) * iw, tv = plane_at(&s->NV, fx, fy) * iw;
    if (aniso && nlev > 1) {
      const float ux = fmaf(-tu, s->W.qx, s->NU.qx) * iw, vx = fmaf(-tv, s->W.qx, s->NV.qx) * iw;
      const float uy = fmaf(-tu, s->W.qy, s->NU.qy) * iw, vy = fmaf(-tv, s->W.qy, s->NV.qy) * iw;
      tex_fetch_aniso(M->tex + toff, tw, th, nlev, tu, tv, ux, vx, uy, vy, alb);
    } else {
      tex_fetch(M->tex + toff, tw, th, tu, tv, alb);
    }
  } else {
    for (int c = 0; c < 3; ++c)
      alb[c] = fmaf(b0, (float)M->colors[4 * g0 + c],
                    fmaf(b1, (float)M->colors[4 * g1 + c], b2 * (float)M->colors[4 * g2 + c])) / 255.0f;
  }
  /* interpolated object-space normal -> camera (OpenCV) frame, unit length */
  float no[3], nc[3];
  for (int c = 0; c < 3; ++c)
    no[c] = fmaf(b0, M->normals[3 * g0 + c], fmaf(b1, M->normals[3 * g1 + c], b2 * M->normals[3 * g2 + c]));
  nc[0] = fmaf(T[0], no[0], fmaf(T[1], no[1], T[2] * no[2]));
  nc[1] = fmaf(T[4], no[0], fmaf(T[5], no[1], T[6] * no[2]));
  nc[2] = fmaf(T[8], no[0], fmaf(T[9], no[1], T[10] * no[2]));
  float nn = sqrtf(fmaf(nc[0], nc[0], fmaf(nc[1], nc[1], nc[2] * nc[2])));
  if (nn > 0.0f) { nc[0] /= nn; nc[1] /= nn; nc[2] /= nn; }
  /* lighting: ambient + Lambert point lights (no attenuation) */
  float lit[3] = {amb[0], amb[1], amb[2]};
  if (n_lights > 0) {
    /* camera-space position of the surface point */
    const float pu = (float)j + 0.5f, pv = (float)i + 0.5f, Z = iw;
    float py = (pv - Kv[5]) * Z / Kv[4];
    float px = ((pu - Kv[2]) * Z - Kv[1] * py) / Kv[0];
    for (int l = 0; l < n_lights; ++l) {
      const float* lp = light_pos + 3 * ((size_t)view * n_lights + l);
      const float* lc = light_col + 3 * ((size_t)view * n_lights + l);
      float lx = fmaf(T[0], lp[0], fmaf(T[1], lp[1], fmaf(T[2], lp[2], T[3]))) - px;
      float ly = fmaf(T[4], lp[0], fmaf(T[5], lp[1], fmaf(T[6], lp[2], T[7]))) - py;
      float lz = fmaf(T[8], lp[0], fmaf(T[9], lp[1], fmaf(T[10], lp[2], T[11]))) - Z;
      float ln = sqrtf(fmaf(lx, lx, fmaf(ly, ly, lz * lz)));
      float ndl = ln > 0.0f ? fmaf(nc[0], lx, fmaf(nc[1], ly, nc[2] * lz)) / ln : 0.0f;
      if (ndl > 0.0f) for (int c = 0; c < 3; ++c) lit[c] = fmaf(lc[c], ndl, lit[c]);
    }
  }
  for (int c = 0; c < 3; ++c) o_rgb[c] = quant8(alb[c] * lit[c], q8);
  /* eye-normal colour code; Panda/GL eye space is (x right, y up, z backward) */
  for (int c = 0; c < 3; ++c) o_n[c] = quant8(normal_code(g_conv.normal_sign[c] * nc[g_conv.normal_axis[c]]), q8);
}

/* sample offsets of the conventions record on the 1/256 grid (hardware keeps them on such a grid; D3D: 1/16) */
static inline int sample_sub(float s) { return (int)rintf(s * (float)SUBPIX); }

void hp_oracle_rasterize(const hp_oracle_meshes* M, int n, const int32_t* obj_ids,
                         const float* TCO, const float* K, const float* ambient,
                         int n_lights, const float* light_pos, const float* light_col,
                         int h, int w, int flags,
                         float* rgb, float* nrm, float* depth, uint8_t* mask,
                         int64_t sv, int64_t sc, int64_t sr, int64_t sp,
                         int64_t dsv, int64_t dsr, int64_t dsp) {
  const float depth_max = Z_NEAR / (1.0f - (1.0f - 1e-3f) * (Z_FAR - Z_NEAR) / Z_FAR);
  const float W_NEAR = 1.0f / Z_NEAR, W_FAR = 1.0f / Z_FAR;
#pragma omp parallel
  {
    const int msaa = (flags & HP_R_MSAA4) && (rgb || nrm);
    const int ns = msaa ? 5 : 1;                 /* keys per pixel: 4 colour samples + the centre, or the centre alone */
    int SX[5], SY[5], lo_x = SUBPIX / 2, hi_x = SUBPIX / 2, lo_y = SUBPIX / 2, hi_y = SUBPIX / 2;
    for (int k = 0; k < 4; ++k) {
      SX[k] = sample_sub(g_conv.msaa_x[k]); SY[k] = sample_sub(g_conv.msaa_y[k]);
      if (msaa) {
        if (SX[k] < lo_x) lo_x = SX[k];
        if (SX[k] > hi_x) hi_x = SX[k];
        if (SY[k] < lo_y) lo_y = SY[k];
        if (SY[k] > hi_y) hi_y = SY[k];
      }
    }
    SX[4] = SY[4] = SUBPIX / 2;
    uint64_t* zbuf = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)h * w * ns);
    float* cam = NULL; size_t cam_cap = 0;
    subtri* st = NULL; size_t st_cap = 0;
#pragma omp for schedule(dynamic, 1)
    for (int view = 0; view < n; ++view) {
      const float* T = TCO + 16 * view;
      const float* Kv = K + 9 * view;
      int finite = 1;
      for (int i = 0; i < 16; ++i) finite &= isfinite(T[i]) ? 1 : 0;
      for (int i = 0; i < 9; ++i) finite &= isfinite(Kv[i]) ? 1 : 0;
      const int64_t* ob = M->obj + 8 * obj_ids[view];
      const int64_t voff = ob[0], nv = ob[1], foff = ob[2], nf = ob[3], toff = ob[4];
      const int tw = (int)ob[5], th = (int)ob[6], nlev = (int)ob[7] > 0 ? (int)ob[7] : 1;
      const int aniso = (flags & HP_R_TEX_ANISO) != 0;
      for (size_t p = 0; p < (size_t)h * w * ns; ++p) zbuf[p] = KEY_EMPTY;
      if ((size_t)nv * 3 > cam_cap) { cam_cap = (size_t)nv * 3; cam = (float*)realloc(cam, cam_cap * 4); }
      if ((size_t)nf * 2 > st_cap) { st_cap = (size_t)nf * 2; st = (subtri*)realloc(st, st_cap * sizeof(subtri)); }
      /* slot of a sub-triangle = its id: f, or nf + f (second half of a clipped quad); 0 in `used` = no such sub-triangle */
      unsigned char* used = (unsigned char*)calloc((size_t)nf * 2, 1);

      if (finite) {
        /* vertex stage: camera coordinates */
        for (int64_t i = 0; i < nv; ++i) {
          const float* p = M->verts + 3 * (voff + i);
          cam[3 * i + 0] = fmaf(T[0], p[0], fmaf(T[1], p[1], fmaf(T[2], p[2], T[3])));
          cam[3 * i + 1] = fmaf(T[4], p[0], fmaf(T[5], p[1], fmaf(T[6], p[2], T[7])));
          cam[3 * i + 2] = fmaf(T[8], p[0], fmaf(T[9], p[1], fmaf(T[10], p[2], T[11])));
        }
        /* set-up (+ near-plane clipping), then coverage + depth */
        for (int64_t f = 0; f < nf; ++f) {
          const int32_t* tri = M->faces + 3 * (foff + f);
          const float* C0 = cam + 3 * tri[0];
          const float* C1 = cam + 3 * tri[1];
          const float* C2 = cam + 3 * tri[2];
          const float zmin = fminf(C0[2], fminf(C1[2], C2[2])), zmax = fmaxf(C0[2], fmaxf(C1[2], C2[2]));
          if (!(zmax >= Z_NEAR) || !(zmin <= Z_FAR)) continue;
          const float uv3[6] = {M->uvs[2 * (voff + tri[0])], M->uvs[2 * (voff + tri[0]) + 1], M->uvs[2 * (voff + tri[1])],
                                M->uvs[2 * (voff + tri[1]) + 1], M->uvs[2 * (voff + tri[2])], M->uvs[2 * (voff + tri[2]) + 1]};
          clipv poly[4];
          int np = 0;
          const float* Cc[3] = {C0, C1, C2};
          if (zmin >= Z_NEAR) {
            for (int k = 0; k < 3; ++k)
              for (int c = 0; c < 3; ++c) { poly[k].c[c] = Cc[k][c]; poly[k].b[c] = c == k ? 1.0f : 0.0f; }
            np = 3;
          } else {
            /* clip against z = near.  The corners are first rotated (cyclically: the winding stays) so that corner 0 is
             * inside and corner 2 outside -- r = the inside corner when there is one, the corner after the outside one when
             * there are two -- then the polygon is [V0, I(0->1), I(0->2)] or [V0, V1, I(1->2), I(0->2)], I(i->o) = the point
             * of the edge on the plane, interpolated FROM the inside corner TO the outside one (both triangles of a shared
             * edge compute the same point).  Barycentrics refer to the ORIGINAL corner order. */
            const int in0 = C0[2] >= Z_NEAR, in1 = C1[2] >= Z_NEAR, in2 = C2[2] >= Z_NEAR;
            const int n_in = in0 + in1 + in2;
            int r;
            if (n_in == 1) r = in0 ? 0 : in1 ? 1 : 2;
            else r = !in0 ? 1 : !in1 ? 2 : 0;
            clipv V[3];
            for (int k = 0; k < 3; ++k) {
              const int o = (k + r) % 3;
              for (int c = 0; c < 3; ++c) { V[k].c[c] = Cc[o][c]; V[k].b[c] = c == o ? 1.0f : 0.0f; }
            }
#define ISECT(dst, I, O)                                                           \
            do {                                                                   \
              const float t_ = ((I).c[2] - Z_NEAR) / ((I).c[2] - (O).c[2]);        \
              (dst).c[0] = fmaf(t_, (O).c[0] - (I).c[0], (I).c[0]);                \
              (dst).c[1] = fmaf(t_, (O).c[1] - (I).c[1], (I).c[1]);                \
              (dst).c[2] = Z_NEAR;                                                 \
              for (int c_ = 0; c_ < 3; ++c_) (dst).b[c_] = fmaf(t_, (O).b[c_] - (I).b[c_], (I).b[c_]); \
            } while (0)
            poly[0] = V[0];
            if (n_in == 1) {
              ISECT(poly[1], V[0], V[1]);
              ISECT(poly[2], V[0], V[2]);
              np = 3;
            } else {
              poly[1] = V[1];
              ISECT(poly[2], V[1], V[2]);
              ISECT(poly[3], V[0], V[2]);
              np = 4;
            }
#undef ISECT
          }
          for (int part = 0; part + 2 < np; ++part) {
            const clipv cv3[3] = {poly[0], poly[part + 1], poly[part + 2]};
            subtri* s = st + (part == 0 ? f : nf + f);
            if (!setup_subtri(cv3, Kv, uv3, h, w, lo_x, hi_x, lo_y, hi_y, s)) continue;
            s->id = part == 0 ? f : nf + f;
            s->v[0] = tri[0]; s->v[1] = tri[1]; s->v[2] = tri[2];
            used[s->id] = 1;
            for (int i = s->i0; i <= s->i1; ++i)
              for (int j = s->j0; j <= s->j1; ++j)
                for (int sm = 0; sm < ns; ++sm) {
                  const int slot = msaa ? sm : 4;  /* single-sample mode: the centre only */
                  const int64_t X = (int64_t)SUBPIX * (j - s->ox) + SX[slot], Y = (int64_t)SUBPIX * (i - s->oy) + SY[slot];
                  if (!sample_inside(s, X, Y)) continue;
                  const float fx = (float)(j - s->ox) + (float)SX[slot] * (1.0f / (float)SUBPIX);
                  const float fy = (float)(i - s->oy) + (float)SY[slot] * (1.0f / (float)SUBPIX);
                  const float Ws = plane_at(&s->W, fx, fy);
                  if (!(Ws <= W_NEAR) || !(Ws >= W_FAR)) continue;
                  const uint64_t key = ((uint64_t)(~f2u(Ws)) << 32) | (uint32_t)s->id;
                  uint64_t* zp = zbuf + ((size_t)i * w + j) * ns + sm;
                  if (key < *zp) *zp = key;
                }
          }
        }
      }

      /* resolve: attributes of the nearest triangle at every covered pixel (single sample), or the mean over the four
       * samples of the colours their triangles have at the pixel centre (HP_R_MSAA4) */
      const float* amb = ambient + 3 * view;
      const int q8 = flags & HP_R_QUANT8;
      for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
          const uint64_t* keys = zbuf + ((size_t)i * w + j) * ns;
          const uint64_t ckey = keys[ns - 1]; /* the pixel centre */
          float o_rgb[3] = {0, 0, 0}, o_n[3] = {0, 0, 0}, o_d = 0.0f;
          if (ckey != KEY_EMPTY) {
            const float Z = 1.0f / u2f(~(uint32_t)(ckey >> 32));
            o_d = Z > depth_max ? 0.0f : Z;
          }
          if (!msaa) {
            if (ckey != KEY_EMPTY)
              shade_centre(M, st + (ckey & 0xFFFFFFFFull), T, Kv, amb, n_lights, light_pos, light_col, view, voff, toff, tw, th, nlev,
                           aniso, i, j, q8, o_rgb, o_n);
          } else {
            int64_t cf[4]; float crgb[4][3], cn[4][3]; int ncached = 0;
            float a_rgb[3] = {0, 0, 0}, a_n[3] = {0, 0, 0};
            unsigned k_rgb[3] = {0, 0, 0}, k_n[3] = {0, 0, 0}; /* sums of the samples' 8-bit colour codes */
            for (int sm = 0; sm < 4; ++sm) {
              if (keys[sm] == KEY_EMPTY) continue; /* the clear colour: 0 */
              const int64_t f = (int64_t)(keys[sm] & 0xFFFFFFFFull);
              int k = 0;
              while (k < ncached && cf[k] != f) ++k;
              if (k == ncached) { /* one fragment-shader invocation per pixel and primitive */
                cf[k] = f;
                shade_centre(M, st + f, T, Kv, amb, n_lights, light_pos, light_col, view, voff, toff, tw, th, nlev, aniso, i, j, q8,
                             crgb[k], cn[k]);
                ++ncached;
              }
              for (int c = 0; c < 3; ++c) {
                a_rgb[c] += crgb[k][c]; a_n[c] += cn[k][c];
                k_rgb[c] += (unsigned)floorf(fmaf(crgb[k][c], 255.0f, 0.5f)); k_n[c] += (unsigned)floorf(fmaf(cn[k][c], 255.0f, 0.5f));
              }
            }
            /* the resolve of an 8-bit multisampled colour buffer: the mean of the four samples' 8-bit values, rounded half
             * up -- integer arithmetic (a float mean puts every sum that is 2 mod 4 exactly on a rounding tie).  Without
             * the 8-bit quantisation (diagnostics) the plain float mean. */
            for (int c = 0; c < 3; ++c) {
              if (q8) { o_rgb[c] = (float)((k_rgb[c] + 2u) >> 2) / 255.0f; o_n[c] = (float)((k_n[c] + 2u) >> 2) / 255.0f; }
              else { o_rgb[c] = a_rgb[c] * 0.25f; o_n[c] = a_n[c] * 0.25f; }
            }
          }
          const int64_t base = (int64_t)view * sv + (int64_t)i * sr + (int64_t)j * sp;
          if (rgb) for (int c = 0; c < 3; ++c) rgb[base + c * sc] = o_rgb[c];
          if (nrm && (flags & HP_R_NORMALS)) for (int c = 0; c < 3; ++c) nrm[base + c * sc] = o_n[c];
          const int64_t dbase = (int64_t)view * dsv + (int64_t)i * dsr + (int64_t)j * dsp;
          if (depth && (flags & HP_R_DEPTH)) depth[dbase] = o_d;
          if (mask && (flags & HP_R_MASK)) mask[(size_t)view * h * w + (size_t)i * w + j] = o_d > 0.0f;
        }
      free(used);
    }
    free(zbuf);
    free(cam);
    free(st);
  }
}
