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

/* cross product of homogeneous screen vertices with a canonical operand order (lower
 * vertex id first) so the edge function of a shared edge is bit-identical (up to sign) in
 * both triangles: no cracks between neighbours. */
static inline void edge_fn(const float* A, int ia, const float* B, int ib, float* e) {
  const float* P = A; const float* Q = B; float sgn = 1.0f;
  if (ib < ia) { P = B; Q = A; sgn = -1.0f; }
  e[0] = sgn * fmaf(P[1], Q[2], -(P[2] * Q[1]));
  e[1] = sgn * fmaf(P[2], Q[0], -(P[0] * Q[2]));
  e[2] = sgn * fmaf(P[0], Q[1], -(P[1] * Q[0]));
}

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
 * obj_ids [n]; TCO [n][16]; K [n][9]; ambient [n][3]; n_lights point lights per view:
 * light_pos [n][n_lights][3] (object frame, metres), light_col [n][n_lights][3].
 * Outputs are addressed as base + view*sv + chan*sc + row*sr + col*sp (element strides), so
 * NCHW (sc=h*w, sr=w, sp=1) and NHWC slices are both expressible.  Any output pointer may
 * be NULL.  mask is uint8 with the same (sv, sr, sp) strides divided by... its own strides.
 */
/* One fragment-shader invocation: colour and normal code of triangle f at the CENTRE of pixel (i, j) -- albedo (texture or
 * vertex colours) x (ambient + Lambert point lights), eye-space normal code; perspective-correct barycentrics from the
 * edge functions (extrapolated when the centre lies outside the triangle: multisampled edge pixels). */
static void shade_centre(const hp_oracle_meshes* M, const float* sv3, const float* T, const float* Kv, const float* amb,
                         int n_lights, const float* light_pos, const float* light_col, int view, int64_t voff, int64_t foff,
                         int64_t toff, int tw, int th, int nlev, int aniso, int64_t f, int i, int j, int q8, float* o_rgb,
                         float* o_n) {
  const int32_t* tri = M->faces + 3 * (foff + f);
  const float* V0 = sv3 + 3 * tri[0];
  const float* V1 = sv3 + 3 * tri[1];
  const float* V2 = sv3 + 3 * tri[2];
  float e0[3], e1[3], e2[3];
  edge_fn(V1, tri[1], V2, tri[2], e0);
  edge_fn(V2, tri[2], V0, tri[0], e1);
  edge_fn(V0, tri[0], V1, tri[1], e2);
  const float pu = (float)j + 0.5f, pv = (float)i + 0.5f;
  float l0 = fmaf(e0[0], pu, fmaf(e0[1], pv, e0[2]));
  float l1 = fmaf(e1[0], pu, fmaf(e1[1], pv, e1[2]));
  float l2 = fmaf(e2[0], pu, fmaf(e2[1], pv, e2[2]));
  float s = l0 + l1 + l2;
  float b0 = l0 / s, b1 = l1 / s, b2 = l2 / s; /* perspective-correct barycentrics */
  const float Z = fmaf(l0, V0[2], fmaf(l1, V1[2], l2 * V2[2])) / s; /* the depth the coverage pass stored for a covered centre */
  const int64_t g0 = voff + tri[0], g1 = voff + tri[1], g2 = voff + tri[2];
  /* albedo */
  float alb[3];
  if (toff >= 0) {
    float tu = fmaf(b0, M->uvs[2 * g0], fmaf(b1, M->uvs[2 * g1], b2 * M->uvs[2 * g2]));
    float tv = fmaf(b0, M->uvs[2 * g0 + 1], fmaf(b1, M->uvs[2 * g1 + 1], b2 * M->uvs[2 * g2 + 1]));
    if (aniso && nlev > 1) {
      /* screen-space derivatives of the perspective-correct barycentrics: b_i = l_i / s, l_i affine in (x, y) */
      const float sx = e0[0] + e1[0] + e2[0], sy = e0[1] + e1[1] + e2[1];
      const float bx[3] = {(e0[0] - b0 * sx) / s, (e1[0] - b1 * sx) / s, (e2[0] - b2 * sx) / s};
      const float by[3] = {(e0[1] - b0 * sy) / s, (e1[1] - b1 * sy) / s, (e2[1] - b2 * sy) / s};
      const float ux = fmaf(bx[0], M->uvs[2 * g0], fmaf(bx[1], M->uvs[2 * g1], bx[2] * M->uvs[2 * g2]));
      const float vx = fmaf(bx[0], M->uvs[2 * g0 + 1], fmaf(bx[1], M->uvs[2 * g1 + 1], bx[2] * M->uvs[2 * g2 + 1]));
      const float uy = fmaf(by[0], M->uvs[2 * g0], fmaf(by[1], M->uvs[2 * g1], by[2] * M->uvs[2 * g2]));
      const float vy = fmaf(by[0], M->uvs[2 * g0 + 1], fmaf(by[1], M->uvs[2 * g1 + 1], by[2] * M->uvs[2 * g2 + 1]));
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

void hp_oracle_rasterize(const hp_oracle_meshes* M, int n, const int32_t* obj_ids,
                         const float* TCO, const float* K, const float* ambient,
                         int n_lights, const float* light_pos, const float* light_col,
                         int h, int w, int flags,
                         float* rgb, float* nrm, float* depth, uint8_t* mask,
                         int64_t sv, int64_t sc, int64_t sr, int64_t sp,
                         int64_t dsv, int64_t dsr, int64_t dsp) {
  const float depth_max = Z_NEAR / (1.0f - (1.0f - 1e-3f) * (Z_FAR - Z_NEAR) / Z_FAR);
#pragma omp parallel
  {
    const int msaa = (flags & HP_R_MSAA4) && (rgb || nrm);
    const int ns = msaa ? 5 : 1;                 /* keys per pixel: 4 colour samples + the centre, or the centre alone */
    float SX[5], SY[5], lo_x = 1.0f, hi_x = 0.0f, lo_y = 1.0f, hi_y = 0.0f;  /* the four colour samples + the centre */
    for (int k = 0; k < 4; ++k) {
      SX[k] = g_conv.msaa_x[k]; SY[k] = g_conv.msaa_y[k];
      lo_x = fminf(lo_x, SX[k]); hi_x = fmaxf(hi_x, SX[k]); lo_y = fminf(lo_y, SY[k]); hi_y = fmaxf(hi_y, SY[k]);
    }
    SX[4] = SY[4] = 0.5f;
    uint64_t* zbuf = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)h * w * ns);
    float* sv3 = NULL; size_t sv_cap = 0;
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
      if ((size_t)nv * 3 > sv_cap) { sv_cap = (size_t)nv * 3; sv3 = (float*)realloc(sv3, sv_cap * 4); }

      if (finite) {
        /* vertex stage: homogeneous pixel coordinates (Xh, Yh, W) = K * (R p + t) */
        for (int64_t i = 0; i < nv; ++i) {
          const float* p = M->verts + 3 * (voff + i);
          float cx = fmaf(T[0], p[0], fmaf(T[1], p[1], fmaf(T[2], p[2], T[3])));
          float cy = fmaf(T[4], p[0], fmaf(T[5], p[1], fmaf(T[6], p[2], T[7])));
          float cz = fmaf(T[8], p[0], fmaf(T[9], p[1], fmaf(T[10], p[2], T[11])));
          sv3[3 * i + 0] = fmaf(Kv[0], cx, fmaf(Kv[1], cy, Kv[2] * cz));
          sv3[3 * i + 1] = fmaf(Kv[4], cy, Kv[5] * cz);
          sv3[3 * i + 2] = cz;
        }
        /* coverage + depth: 2-D homogeneous rasterisation (no explicit clipping) */
        for (int64_t f = 0; f < nf; ++f) {
          const int32_t* tri = M->faces + 3 * (foff + f);
          const float* V0 = sv3 + 3 * tri[0];
          const float* V1 = sv3 + 3 * tri[1];
          const float* V2 = sv3 + 3 * tri[2];
          float zmin = fminf(V0[2], fminf(V1[2], V2[2])), zmax = fmaxf(V0[2], fmaxf(V1[2], V2[2]));
          if (!(zmax >= Z_NEAR) || !(zmin <= Z_FAR)) continue;
          float e0[3], e1[3], e2[3];
          edge_fn(V1, tri[1], V2, tri[2], e0);
          edge_fn(V2, tri[2], V0, tri[0], e1);
          edge_fn(V0, tri[0], V1, tri[1], e2);
          float det = fmaf(V0[0], e0[0], fmaf(V0[1], e0[1], V0[2] * e0[2]));
          if (!(det != 0.0f) || !isfinite(det)) continue;
          int x0 = 0, x1 = w - 1, y0 = 0, y1 = h - 1;
          if (zmin > 1e-6f) { /* all in front: tight screen bbox */
            float u0 = V0[0] / V0[2], u1 = V1[0] / V1[2], u2 = V2[0] / V2[2];
            float v0 = V0[1] / V0[2], v1 = V1[1] / V1[2], v2 = V2[1] / V2[2];
            float umin = fminf(u0, fminf(u1, u2)), umax = fmaxf(u0, fmaxf(u1, u2));
            float vmin = fminf(v0, fminf(v1, v2)), vmax = fmaxf(v0, fmaxf(v1, v2));
            if (!(umax >= 0.0f) || !(umin <= (float)w) || !(vmax >= 0.0f) || !(vmin <= (float)h)) continue;
            /* pixel centre j+0.5 in [umin, umax]  <=>  j in [ceil(umin-0.5), floor(umax-0.5)] */
            float a = ceilf(umin - 0.5f), b = floorf(umax - 0.5f);
            float c = ceilf(vmin - 0.5f), d = floorf(vmax - 0.5f);
            if (msaa) { /* some sample of pixel j inside [umin, umax]: the offsets run from lo to hi (default 0.125 to 0.875) */
              a = ceilf(umin - hi_x); b = floorf(umax - lo_x); c = ceilf(vmin - hi_y); d = floorf(vmax - lo_y);
            }
            x0 = a < 0.0f ? 0 : (int)a; x1 = b > (float)(w - 1) ? w - 1 : (int)b;
            y0 = c < 0.0f ? 0 : (int)c; y1 = d > (float)(h - 1) ? h - 1 : (int)d;
          }
          for (int i = y0; i <= y1; ++i) {
            for (int j = x0; j <= x1; ++j) {
              for (int sm = 0; sm < ns; ++sm) {
                const int slot = msaa ? sm : 4;  /* single-sample mode: the centre only */
                const float pv = (float)i + SY[slot], pu = (float)j + SX[slot];
                float l0 = fmaf(e0[0], pu, fmaf(e0[1], pv, e0[2]));
                float l1 = fmaf(e1[0], pu, fmaf(e1[1], pv, e1[2]));
                float l2 = fmaf(e2[0], pu, fmaf(e2[1], pv, e2[2]));
                float s = l0 + l1 + l2;
                int in_pos = (l0 >= 0.0f) & (l1 >= 0.0f) & (l2 >= 0.0f) & (s > 0.0f);
                int in_neg = (l0 <= 0.0f) & (l1 <= 0.0f) & (l2 <= 0.0f) & (s < 0.0f);
                if (!(in_pos | in_neg)) continue;
                /* camera-space depth: affine over the 3-D triangle, i.e. linear in the perspective-correct barycentrics
                 * l_i / s.  (Z = det / s is the same number algebraically, but det -- a 3x3 determinant of homogeneous
                 * pixel coordinates ~1e2 whose value is ~area * z^3 ~ 0.1 -- cancels to ~5e-4 relative in fp32 on
                 * pixel-sized triangles: 0.1 mm of depth noise at 0.4 m.  Interpolating the vertex depths is accurate to
                 * ~1e-7 m, the precision class of the 24-bit depth buffer the reference reads back,
                 * TB/renderer/utils.py:46-60.) */
                float Z = fmaf(l0, V0[2], fmaf(l1, V1[2], l2 * V2[2])) / s;
                if (!(Z >= Z_NEAR) || !(Z <= Z_FAR)) continue;
                uint64_t key = ((uint64_t)f2u(Z) << 32) | (uint32_t)f;
                uint64_t* zp = zbuf + ((size_t)i * w + j) * ns + sm;
                if (key < *zp) *zp = key;
              }
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
            const float Z = u2f((uint32_t)(ckey >> 32));
            o_d = Z > depth_max ? 0.0f : Z;
          }
          if (!msaa) {
            if (ckey != KEY_EMPTY)
              shade_centre(M, sv3, T, Kv, amb, n_lights, light_pos, light_col, view, voff, foff, toff, tw, th, nlev, aniso,
                           (int64_t)(ckey & 0xFFFFFFFFull), i, j, q8, o_rgb, o_n);
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
                shade_centre(M, sv3, T, Kv, amb, n_lights, light_pos, light_col, view, voff, foff, toff, tw, th, nlev, aniso, f, i,
                             j, q8, crgb[k], cn[k]);
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
    }
    free(zbuf);
    free(sv3);
  }
}
