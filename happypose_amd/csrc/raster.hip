// HIP rasteriser for gfx950: renders one textured mesh per view for thousands of views per
// launch, straight into the (strided) tensors the conv stem reads.
//
// Replaces Panda3dBatchRenderer.render -> worker processes -> Panda3D/OpenGL
// (TB/renderer/panda3d_batch_renderer.py:194-286, TB/renderer/panda3d_scene_renderer.py:320-390).
// The arithmetic (operation order included) follows the CPU definition in oracle/csrc/oracle.c
// ("the rasteriser's geometry, round 5") so that coverage and depth agree bit for bit.
//
// Round 5: the geometry is defined the way OpenGL hardware defines it -- vertices snapped to a 1/256-px grid, exact integer
// edge functions with a top-left fill rule, 1/z and the attributes as PLANES in window coordinates -- and the work is split
// so that nothing is derived twice:
//   * raster_setup_kernel, one lane per (view, triangle): camera transform of the three corners (the mesh is L2 / MALL
//     resident and shared by every view of the object), near-plane clipping (rare), snapping, back-face culling of closed
//     components on the EXACT sign of the snapped area, plane set-up, one 128-B record per sub-triangle (FaceRec below) and
//     the append of its id to the list of every band of rows it can touch;
//   * raster_kernel, one workgroup per (view, band): coverage is a walk over the candidate pixels of each listed record with
//     nine integer multiply-adds per pixel and three adds per sample (the per-sample IEEE division, the nine-product
//     homogeneous edge set-up per (triangle, band) and per shading invocation, and the list -> corner indices -> vertex
//     records chain of rounds 1-4 are gone); the band's 64-bit z-buffer {~bits(1/z) : id} lives in LDS (ds_min_u64); shading
//     evaluates the record's planes at the pixel centre (six FMAs and one division for u, v and their screen derivatives)
//     and goes straight to the texture; the output pass is pixel-parallel (coalesced NCHW planes or NHWC record runs).
//   * HBM traffic = the output tensor + the records (64 - 96 B per visible triangle and view) + first touch of mesh / texture.
//   * workgroups are renumbered so that each XCD gets a contiguous range of views: consecutive hypotheses share an object,
//     hence mesh, texture and records stay in that XCD's L2.
#include <atomic>
#include <cmath>
#include <mutex>

#include "common.h"
#include "crop_math.h"

#ifndef HP_RASTER_BAND_PIXELS
#define HP_RASTER_BAND_PIXELS 3200
#endif
#ifndef HP_RASTER_THREADS
#define HP_RASTER_THREADS 512
#endif

namespace hp {

#pragma clang fp contract(off)

constexpr float kZNear = 0.1f;
constexpr float kZFar = 10.0f;
constexpr int kSub = 256;                  // sub-pixel grid of the snapped vertices and of the sample positions
constexpr float kGuardSub = 4194304.0f;    // guard band: +-2^22 sub-pixel units = +-16384 px
constexpr int kSmallLimit = 8192;          // |corner - origin| <= 32 px on both axes: 32-bit edge functions (v_mad_i32_i24)
constexpr unsigned long long kKeyEmpty = 0xFFFFFFFFFFFFFFFFull;
constexpr int kBandPixels = HP_RASTER_BAND_PIXELS;  // LDS z-buffer of a band, 8 B per pixel
// HP_RASTER_MSAA4 (the reference's framebuffer state, see oracle.c HP_R_MSAA4): five keys per pixel -- the four colour
// samples of the standard 4x pattern and the pixel centre (depth / mask stay centre-sampled) -- in a 25-KB z-buffer:
// 2 rows of 320 pixels per band, four 256-thread workgroups per CU (band_threads below); renders wider than 640 px take
// 512 threads on 6400 keys.
constexpr int kSamplesMsaa = 5;
#ifndef HP_RASTER_BAND_KEYS_MSAA
#define HP_RASTER_BAND_KEYS_MSAA 3200
#endif
constexpr int kMaxViews = 8;  // views per item a record layout can describe
constexpr int kBandKeysMsaa = HP_RASTER_BAND_KEYS_MSAA;
constexpr int kBandKeysMsaaWide = 6400;
// The renderer conventions nobody can pin without Panda3D (hp_raster_conventions in the header), as the kernels see them:
// the store's record plus what the host derives from it once per launch.  Kernel arguments (SGPRs).
struct RasterConv {
  int sxi[5], syi[5];            // sample offsets inside the pixel in 1/256 px: the four colour samples + the centre (slot 4)
  float sxf[5], syf[5];          // the same in pixels (exact)
  int lo_x, hi_x, lo_y, hi_y;    // smallest / largest offset among the samples THIS launch tests (bounding-box growth)
  float aniso_max, lod_bias, ratio_bias;
  int aniso_round, lod_from;
  int n_axis[3]; float n_sign[3];
};
// probe count and level of detail of the anisotropic filter from the footprint (pmax >= pmin, texel units)
template <class CV>
__device__ __forceinline__ void aniso_footprint(const CV& cv, float pmax, float pmin, int nlev, float& nf, float& lod) {
  const float r = pmax / pmin + cv.ratio_bias;
  nf = pmin > 0.0f ? (cv.aniso_round == 0 ? ceilf(r) : cv.aniso_round == 1 ? rintf(r) : floorf(r)) : cv.aniso_max;
  if (!(nf >= 1.0f)) nf = 1.0f;
  if (nf > cv.aniso_max) nf = cv.aniso_max;
  const float la = cv.lod_from == 0 ? pmax / nf : cv.lod_from == 1 ? pmin : pmax;
  lod = la > 0.0f ? log2f(la) + cv.lod_bias : 0.0f;
  if (!(lod > 0.0f)) lod = 0.0f;
  if (lod > (float)(nlev - 1)) lod = (float)(nlev - 1);
}
constexpr int kBigQueue = 512;
constexpr int kBigArea = 128;  // candidate pixels above which a triangle is walked cooperatively
constexpr int kSmallArea = 6;  // ... up to which a record goes to the front of its band's list (raster_setup_kernel)
#ifndef HP_RASTER_THREADS_MSAA
#define HP_RASTER_THREADS_MSAA 256
#endif
// NS: keys per pixel (1 / 5); WIDE: the multisampled instantiation for renders wider than 640 px (512 threads, 6400 keys)
constexpr __host__ __device__ int band_threads(int ns, bool wide = false) { return ns == 1 ? HP_RASTER_THREADS : wide ? 512 : HP_RASTER_THREADS_MSAA; }
constexpr int kBinThreads = 256;   // set-up kernel: small workgroups (six per CU) -- a 1024-thread workgroup left one per CU waiting at its barriers (66 us per 128 views instead of ~25)
constexpr int kMaxBands = 768;  // 720 one-row multisampled bands of a 1280 x 720 render

struct RasterArgs {
  const float4* verts4;   // xyz + pad
  const float4* normals4;
  const float* uvs;
  const uint8_t* colors;
  const int4* faces4;     // {i0, i1, i2, cull flag} (MeshStore::faces4)
  const uint8_t* tex;
  const uint8_t* tex_quads; const int64_t* tex_quads_off;  // MeshStore::tex_quads
  const int64_t* obj;
  const float* cull;      // MeshStore::cull ([n_obj][8]) or null: back faces of closed components are not set up
  const int32_t* obj_ids;
  const float* TCO;
  const float* K;
  const float* ambient;
  const float* light_pos;
  const float* light_col;
  const float* depth_norm_z;
  float* rgb;
  float* nrm;
  float* depth;
  uint8_t* mask;
  hp_strides cs, ds;
  int n, views_per_item, n_lights, h, w, flags, depth_norm_mode;
  int band_rows, n_bands;
  int msaa;             // 1: five keys per pixel (HP_RASTER_MSAA4 and a colour / normal output), 0: the centre only
  float depth_max;
  // per-(view, band) lists of sub-triangle ids built by raster_setup_kernel
  int32_t* bin_count;   // [chunk views][n_bands][2] (small boxes from the front of the list, the others from the back); zero between launches
  int32_t* bin_list;    // [chunk views][n_bands][bin_cap]
  int bin_cap, view0, max_faces, max_verts;
  int4* xverts;         // [chunk views][max_verts] {x, y, bits(1 / z), bits(z)} written by raster_xform_kernel
  // set-up records of the chunk's views: [view][rec_slots = 2 * max_faces] x 128 B (FaceRec)
  uint4* recs;
  uint4* recs_wide;     // [view][rec_slots][2]: the corners of a BIG sub-triangle as int32 (rare: the lines are only touched then)
  int rec_slots, rec_q;  // rec_q: uint4s per record -- 4 (64 B: coverage + texture planes) or 8 (128 B: + barycentric planes, need_attr)
  int need_attr;        // the shading needs barycentrics (normals, point lights or vertex colours): sector 2 of the records
  // ---- record mode (rec != nullptr): the network input [item][row][col][rec_col elements], fp32 or fp16.  View v of an
  // item writes its render channels (rgb, normals, depth as requested) at element v_c0[v] of the pixel record and, when
  // v_crop_n[v] > 0, the observed crop's source channels [v_crop_src0[v], + v_crop_n[v]) at element v_crop_c0[v]
  void* rec;
  int rec_half, rec_own_all;           // fp16 records; V == 1, reference order: the view owns the whole 16-half record (pads written too)
  int64_t rec_item, rec_row, rec_col;  // element strides
  int v_c0[kMaxViews], v_crop_c0[kMaxViews], v_crop_src0[kMaxViews], v_crop_n[kMaxViews];
  int want_nrm, want_depth;            // record mode: which render channels exist
  // fused crop: the frame(s), one box per item, roi_align sampling ratio, depth rule / normalisation of source channel 3
  const float* images; int Bi, Ct, IH, IW, sr, crop_nc, crop_depth_mode;
  const float* boxes; const int32_t* im_ids;
  unsigned w_magic;                    // p / w = (p * w_magic) >> 32 for p < 2^16
  RasterConv cv;                       // the store's conventions record, derived at launch
};

__device__ __forceinline__ float quant8(float c, int on) {
  c = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
  if (!on) return c;
  return floorf(fmaf(c, 255.0f, 0.5f)) / 255.0f;
}

__device__ __forceinline__ float normal_code(float n) {
  float s = n - floorf(n);
  float x = fmaf(s, 32.0f, -0.5f);
  float xf = floorf(x);
  float f = x - xf;
  int i0 = ((int)xf + 32) & 31, i1 = (i0 + 1) & 31;
  float t0 = floorf((float)i0 * 255.0f / 32.0f), t1 = floorf((float)i1 * 255.0f / 32.0f);
  return fmaf(f, t1 - t0, t0) / 255.0f;
}

// x / 255.0f, correctly rounded, in three instructions instead of the ~10 of the IEEE expansion: q = RN(x / 255) follows from
// one multiply by RN(1 / 255) and one Newton correction with the exact remainder.  Checked exhaustively against the division
// for every float in [2^-80, 512] and for 0 (746,586,113 values, 0 mismatches; the colours it is applied to lie in [0, 255]).
__device__ __forceinline__ float div255(float x) {
  constexpr float rc = 1.0f / 255.0f;
  const float q = x * rc;
  const float r = fmaf(-q, 255.0f, x);
  return fmaf(r, rc, q);
}

__device__ __forceinline__ void tex_fetch(const uint8_t* tex, int tw, int th, float u, float v, float* rgb) {
  float x = fmaf(u, (float)tw, -0.5f);
  float y = fmaf(1.0f - v, (float)th, -0.5f);
  float xf = floorf(x), yf = floorf(y);
  float fx = x - xf, fy = y - yf;
  int x0 = (int)xf % tw; if (x0 < 0) x0 += tw;
  int y0 = (int)yf % th; if (y0 < 0) y0 += th;
  int x1 = x0 + 1 == tw ? 0 : x0 + 1;
  int y1 = y0 + 1 == th ? 0 : y0 + 1;
  const uchar4 p00 = *reinterpret_cast<const uchar4*>(tex + 4 * ((size_t)y0 * tw + x0));
  const uchar4 p01 = *reinterpret_cast<const uchar4*>(tex + 4 * ((size_t)y0 * tw + x1));
  const uchar4 p10 = *reinterpret_cast<const uchar4*>(tex + 4 * ((size_t)y1 * tw + x0));
  const uchar4 p11 = *reinterpret_cast<const uchar4*>(tex + 4 * ((size_t)y1 * tw + x1));
  const float c00[3] = {(float)p00.x, (float)p00.y, (float)p00.z};
  const float c01[3] = {(float)p01.x, (float)p01.y, (float)p01.z};
  const float c10[3] = {(float)p10.x, (float)p10.y, (float)p10.z};
  const float c11[3] = {(float)p11.x, (float)p11.y, (float)p11.z};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float a = fmaf(fx, c01[c] - c00[c], c00[c]);
    float b = fmaf(fx, c11[c] - c10[c], c10[c]);
    rgb[c] = div255(fmaf(fy, b - a, a));
  }
}

// bilinear fetch of mip level `lvl` (level k is max(1, tw >> k) x max(1, th >> k), stored behind the levels before it)
__device__ __forceinline__ void tex_fetch_level(const uint8_t* tex, int tw, int th, int lvl, float u, float v, float* rgb) {
  size_t off = 0;
  int w = tw, h = th;
  for (int k = 0; k < lvl; ++k) { off += (size_t)4 * w * h; w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1; }
  tex_fetch(tex + off, w, h, u, v, rgb);
}

// trilinear + anisotropic fetch (HP_RASTER_TEX_ANISO; oracle.c tex_fetch_aniso, same operations in the same order)
template <class CV>
__device__ __forceinline__ void tex_fetch_aniso(const CV& cv, const uint8_t* tex, int tw, int th, int nlev, float u, float v, float ux,
                                                float vx, float uy, float vy, float* rgb) {
  const float px = sqrtf(fmaf(ux * (float)tw, ux * (float)tw, vx * (float)th * (vx * (float)th)));
  const float py = sqrtf(fmaf(uy * (float)tw, uy * (float)tw, vy * (float)th * (vy * (float)th)));
  const bool along_x = px >= py;
  const float pmax = along_x ? px : py, pmin = along_x ? py : px;
  float nf, lod;
  aniso_footprint(cv, pmax, pmin, nlev, nf, lod);
  const int N = (int)nf;
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
#pragma unroll
      for (int c = 0; c < 3; ++c) c0[c] = fmaf(fl, c1[c] - c0[c], c0[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] += c0[c];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) rgb[c] = acc[c] / (float)N;
}

// ---- the same filter, organised for the machine (power-of-two textures; others take tex_fetch_aniso above): the mip
// levels' offsets / sizes come from a per-workgroup table (LDS) and the probes are processed in PAIRS so that the 16 texel
// loads of two trilinear probes are in flight together (the rolled loop pays one L2 round trip per probe).
struct MipTable {
  int off[16], w[16], h[16];
  int qoff[16];      // power-of-two textures: byte offset of the level in the ROW-PAIR copy (MeshStore::tex_quads), entries per row = w + 1
  int p2;            // the object has a row-pair copy (both sizes of level 0 are powers of two; then every level's are)
  float tt[16][16];  // probe position t = i / (N + 1) - 0.5 at [N - 1][i - 1]: one IEEE division per entry and workgroup, not per probe
};

struct BiTexels { uchar4 a, b, c, d; };

// The same filter for power-of-two textures (every level a power of two): wraps are masks, the 2 x 2 footprint of a bilinear
// tap is ONE 16-B load from the row-pair copy of the texture (MeshStore::tex_quads; round 5: the four 4-B gathers per tap
// made the shading phase bound by the texture-address path), addresses are 32-bit offsets from the object's copy (a scalar
// base), probe positions come from the workgroup's table.  Identical values: the integer identities hold for every operand,
// the table entries are the quotients the loop used to recompute.
struct BiTapP2 { uint32_t o; float fx, fy; };
__device__ __forceinline__ BiTapP2 bi_setup_p2(int qoff, int w, int h, float u, float v) {
  const float x = fmaf(u, (float)w, -0.5f);
  const float y = fmaf(1.0f - v, (float)h, -0.5f);
  const float xf = floorf(x), yf = floorf(y);
  BiTapP2 t;
  t.fx = x - xf; t.fy = y - yf;
  const int x0 = (int)xf & (w - 1), y0 = (int)yf & (h - 1);
  // entry x0 of row-pair y0: {T(x0, y0), T(x0, y1)}, the next entry {T(x1, y0), T(x1, y1)} (entry w repeats entry 0)
  t.o = (uint32_t)qoff + 8u * (uint32_t)(y0 * (w + 1) + x0);
  return t;
}
typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
__device__ __forceinline__ BiTexels bi_load_p2(const uint8_t* texq, const BiTapP2& t) {
#if defined(HP_TEXQ_4X4)
  const uint32_t* const qp = reinterpret_cast<const uint32_t*>(texq + t.o);
  struct { uint32_t x, y, z, w; } q{qp[0], qp[1], qp[2], qp[3]};
  asm volatile("" ::: "memory");
#elif defined(HP_TEXQ_2X8)
  typedef uint32_t u32x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
  const u32x2_a8 qa = *reinterpret_cast<const u32x2_a8*>(texq + t.o), qb = *reinterpret_cast<const u32x2_a8*>(texq + t.o + 8);
  struct { uint32_t x, y, z, w; } q{qa.x, qa.y, qb.x, qb.y};
#else
  const u32x4_a8 q = *reinterpret_cast<const u32x4_a8*>(texq + t.o);  // the whole 2 x 2 footprint: one 16-B load, 8-B aligned
#endif
  BiTexels r;
  auto px = [](uint32_t v) { return make_uchar4((unsigned char)(v & 255u), (unsigned char)((v >> 8) & 255u), (unsigned char)((v >> 16) & 255u), (unsigned char)(v >> 24)); };
  r.a = px(q.x); r.c = px(q.y); r.b = px(q.z); r.d = px(q.w);
  return r;
}
__device__ __forceinline__ void bi_finish_p2(float fx, float fy, const BiTexels& q, float* rgb) {
  const float c00[3] = {(float)q.a.x, (float)q.a.y, (float)q.a.z};
  const float c01[3] = {(float)q.b.x, (float)q.b.y, (float)q.b.z};
  const float c10[3] = {(float)q.c.x, (float)q.c.y, (float)q.c.z};
  const float c11[3] = {(float)q.d.x, (float)q.d.y, (float)q.d.z};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float a = fmaf(fx, c01[c] - c00[c], c00[c]);
    float b = fmaf(fx, c11[c] - c10[c], c10[c]);
    rgb[c] = div255(fmaf(fy, b - a, a));
  }
}

template <class CV>
__device__ __forceinline__ void tex_fetch_aniso_p2(const CV& cv, const uint8_t* tex, const MipTable& mt, int tw, int th, int nlev, float u, float v,
                                                   float ux, float vx, float uy, float vy, float* rgb) {
  const float px = sqrtf(fmaf(ux * (float)tw, ux * (float)tw, vx * (float)th * (vx * (float)th)));
  const float py = sqrtf(fmaf(uy * (float)tw, uy * (float)tw, vy * (float)th * (vy * (float)th)));
  const bool along_x = px >= py;
  const float pmax = along_x ? px : py, pmin = along_x ? py : px;
  float nf, lod;
  aniso_footprint(cv, pmax, pmin, nlev, nf, lod);
  const int N = (int)nf;
  const int l0 = (int)lod;
  const float fl = lod - (float)l0;
  const float du = along_x ? ux : uy, dv = along_x ? vx : vy;
  const bool two = fl > 0.0f && l0 + 1 < nlev;
  const int l1 = two ? l0 + 1 : l0;
#ifdef HP_TEX_NO_TABLE  // diagnostics: the level geometry / probe positions recomputed per fetch instead of read from the LDS table
  int off0 = 0, w0 = tw, h0 = th;
  for (int k = 0; k < l0; ++k) { off0 += 8 * (w0 + 1) * h0; w0 = w0 > 1 ? w0 >> 1 : 1; h0 = h0 > 1 ? h0 >> 1 : 1; }
  int off1 = off0, w1 = w0, h1 = h0;
  if (l1 != l0) { off1 += 8 * (w0 + 1) * h0; w1 = w0 > 1 ? w0 >> 1 : 1; h1 = h0 > 1 ? h0 >> 1 : 1; }
  float tt[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) tt[k] = (float)(k + 1) / (float)(N + 1) - 0.5f;
#else
  const int off0 = mt.qoff[l0], w0 = mt.w[l0], h0 = mt.h[l0];
  const int off1 = mt.qoff[l1], w1 = mt.w[l1], h1 = mt.h[l1];
  const float* const tt = mt.tt[N - 1];
#endif
#ifdef HP_TEX_DEBUG_MATH  // diagnostics: colour = a hash of everything the fetch addresses are computed from; no texel is read
  {
#if HP_TEX_DEBUG_MATH == 1   // the interpolated coordinates and their derivatives (the records' planes)
    uint32_t hsh = __float_as_uint(u) * 31u ^ __float_as_uint(v) * 131u ^ __float_as_uint(ux) * 7u ^ __float_as_uint(vx) * 13u ^ __float_as_uint(uy) * 3u ^ __float_as_uint(vy) * 5u;
#elif HP_TEX_DEBUG_MATH == 2  // the footprint arithmetic on them (sqrt, division, ceil, log2)
    uint32_t hsh = (uint32_t)N * 2654435761u ^ (uint32_t)l0 * 40503u ^ __float_as_uint(fl);
#elif HP_TEX_DEBUG_MATH == 3  // the level table
    uint32_t hsh = (uint32_t)off0 ^ (uint32_t)w0 * 17u ^ (uint32_t)h0 * 257u ^ (uint32_t)off1 * 3u ^ __float_as_uint(tt[0]);
#else
    uint32_t hsh = (uint32_t)N * 2654435761u ^ (uint32_t)l0 * 40503u ^ __float_as_uint(fl) ^ __float_as_uint(u) * 31u ^ __float_as_uint(v) * 131u ^
                   __float_as_uint(du) * 7u ^ __float_as_uint(dv) * 13u ^ (uint32_t)off0 ^ (uint32_t)w0 * 17u;
#endif
    hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
    rgb[0] = (float)(hsh & 255u) / 255.0f; rgb[1] = (float)((hsh >> 8) & 255u) / 255.0f; rgb[2] = (float)((hsh >> 16) & 255u) / 255.0f;
    return;
  }
#endif
  float acc[3] = {0.0f, 0.0f, 0.0f};
  for (int i = 1; i <= N; i += 2) {
    const bool second = i + 1 <= N;
    const float ta = tt[i - 1];
    const float tb = tt[second ? i : i - 1];
    const float sua = fmaf(ta, du, u), sva = fmaf(ta, dv, v), sub = fmaf(tb, du, u), svb = fmaf(tb, dv, v);
    const BiTapP2 a0 = bi_setup_p2(off0, w0, h0, sua, sva), a1 = bi_setup_p2(off1, w1, h1, sua, sva);
    const BiTapP2 b0 = bi_setup_p2(off0, w0, h0, sub, svb), b1 = bi_setup_p2(off1, w1, h1, sub, svb);
    const BiTexels qa0 = bi_load_p2(tex, a0), qa1 = bi_load_p2(tex, a1), qb0 = bi_load_p2(tex, b0), qb1 = bi_load_p2(tex, b1);
    float ca[3], cb[3], c1[3];
    bi_finish_p2(a0.fx, a0.fy, qa0, ca);
    if (two) {
      bi_finish_p2(a1.fx, a1.fy, qa1, c1);
#pragma unroll
      for (int c = 0; c < 3; ++c) ca[c] = fmaf(fl, c1[c] - ca[c], ca[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] += ca[c];
    if (second) {
      bi_finish_p2(b0.fx, b0.fy, qb0, cb);
      if (two) {
        bi_finish_p2(b1.fx, b1.fy, qb1, c1);
#pragma unroll
        for (int c = 0; c < 3; ++c) cb[c] = fmaf(fl, c1[c] - cb[c], cb[c]);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += cb[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) rgb[c] = acc[c] / (float)N;
}

// ---- set-up records ---------------------------------------------------------------------------------------------------
// One 128-B record (8 x uint4, one cache line) per sub-triangle and view, written by raster_setup_kernel at
// recs[(view * rec_slots + id) * 8], id = f, or n_faces + f for the second half of a near-clipped quad:
//   q0 = {ox | oy << 16, flags (bit 0: big, bit 1: probe-count class), rx0 | ry0 << 16, rx1 | ry1 << 16}     coverage: origin pixel, corners relative to
//   q1 = {rx2 | ry2 << 16, W.q0, W.qx, W.qy}                                         (256 ox, 256 oy) as int16 pairs, 1 / z plane
//   q2 = {NU.q0, NU.qx, NU.qy, NV.q0}                                               shading: numerator planes of u and v,
//   q3 = {NV.qx, NV.qy, v0, v1}                                                      original vertex ids (object-local)
//   q4 = {NB1.q0, NB1.qx, NB1.qy, NB2.q0}                                           (need_attr) numerator planes of the
//   q5 = {NB2.qx, NB2.qy, v2, 0}                                                     barycentrics b1, b2
//   q6 = {rx0, ry0, rx1, ry1}, q7 = {rx2, ry2, 0, 0}                                (big) the corners as int32
// A plane q(fx, fy) = q0 + qx fx + qy fy, (fx, fy) = pixels from the origin pixel's top-left corner.  Corners are
// orientation-normalised (interior = positive side of the three edge functions).
struct PlaneQ { float q0, qx, qy; };
struct Corner { float c[3]; float b[3]; };  // camera-space corner + barycentrics w.r.t. the original triangle

__device__ __forceinline__ int snap_sub(float s) {
  float r = s * (float)kSub;
  if (!(r >= -kGuardSub)) r = -kGuardSub;  // NaN lands here
  if (!(r <= kGuardSub)) r = kGuardSub;
  return (int)rintf(r);
}
__device__ __forceinline__ float plane_at(const PlaneQ& p, float fx, float fy) { return fmaf(p.qy, fy, fmaf(p.qx, fx, p.q0)); }
__device__ __forceinline__ PlaneQ make_plane(float qa, float qb, float qc, const float (&px)[3], const float (&py)[3], float inv) {
  const float dx1 = px[1] - px[0], dy1 = py[1] - py[0], dx2 = px[2] - px[0], dy2 = py[2] - py[0];
  const float dq1 = qb - qa, dq2 = qc - qa;
  PlaneQ p;
  p.qx = fmaf(dq1, dy2, -(dq2 * dy1)) * inv;
  p.qy = fmaf(dq2, dx1, -(dq1 * dx2)) * inv;
  p.q0 = fmaf(-p.qy, py[0], fmaf(-p.qx, px[0], qa));
  return p;
}

// Set-up of one sub-triangle from its projected, snapped corners (oracle.c setup_subtri: same operations in the same order;
// project_corner = its first loop, run per VERTEX by raster_xform_kernel for unclipped triangles).  Writes its record and returns the
// rows [row_lo, row_hi] its candidate pixels span; false when nothing can be covered (or the face is culled).
// cull_flag: 0, or the face's orientation flag (+1 / -1) when this view may cull: a closed component's face whose inward
// side is turned to the camera.  The camera looks along +z with x right and y down, so a face whose winding normal
// (b - a) x (c - a) points at the camera has NEGATIVE snapped area.
struct SnapCorner { int x, y; float w; float b[3]; };  // snapped window position (1/256 px), 1 / z, barycentrics w.r.t. the original triangle
__device__ __forceinline__ SnapCorner project_corner(const float (&Kv)[9], const Corner& c) {
  const float Xh = fmaf(Kv[0], c.c[0], fmaf(Kv[1], c.c[1], Kv[2] * c.c[2]));
  const float Yh = fmaf(Kv[4], c.c[1], Kv[5] * c.c[2]);
  SnapCorner r;
  r.w = 1.0f / c.c[2];
  r.x = snap_sub(Xh * r.w);
  r.y = snap_sub(Yh * r.w);
  r.b[0] = c.b[0]; r.b[1] = c.b[1]; r.b[2] = c.b[2];
  return r;
}
__device__ __forceinline__ bool setup_subtri(const RasterArgs& a, const SnapCorner& c0, const SnapCorner& c1, const SnapCorner& c2,
                                             const float2 (&uv)[3], const int4 tri, int cull_flag, int tw, int th, uint4* rec, uint4* wrec, int& row_lo, int& row_hi, int& cols) {
  int x[3] = {c0.x, c1.x, c2.x}, y[3] = {c0.y, c1.y, c2.y};
  float wk[3] = {c0.w, c1.w, c2.w};
  long long area2 = (long long)(x[1] - x[0]) * (long long)(y[2] - y[0]) - (long long)(x[2] - x[0]) * (long long)(y[1] - y[0]);
  if (area2 == 0) return false;
  if (cull_flag != 0 && ((area2 > 0) == (cull_flag > 0))) return false;
  float b1[3] = {c1.b[0], c1.b[1], c1.b[2]}, b2[3] = {c2.b[0], c2.b[1], c2.b[2]};
  if (area2 < 0) {  // normalise the orientation: swap corners 1 and 2
    int t = x[1]; x[1] = x[2]; x[2] = t; t = y[1]; y[1] = y[2]; y[2] = t;
    float tf = wk[1]; wk[1] = wk[2]; wk[2] = tf;
#pragma unroll
    for (int c = 0; c < 3; ++c) { tf = b1[c]; b1[c] = b2[c]; b2[c] = tf; }
    area2 = -area2;
  }
  const float* const bb[3] = {c0.b, b1, b2};
  const int xmin = min(x[0], min(x[1], x[2])), xmax = max(x[0], max(x[1], x[2]));
  const int ymin = min(y[0], min(y[1], y[2])), ymax = max(y[0], max(y[1], y[2]));
  // origin pixel: the one holding the bounding box's minimum, clamped into the image (x >> 8 = floor division)
  const int j0 = min(max(xmin >> 8, 0), a.w - 1), i0 = min(max(ymin >> 8, 0), a.h - 1);
  // candidate pixels: some sample offset s in [lo, hi] with xmin <= 256 j + s <= xmax (ceil(a / 256) = -((-a) >> 8))
  const int ja = max(-((a.cv.hi_x - xmin) >> 8), j0), j1 = min((xmax - a.cv.lo_x) >> 8, a.w - 1);
  const int ia = max(-((a.cv.hi_y - ymin) >> 8), i0), i1 = min((ymax - a.cv.lo_y) >> 8, a.h - 1);
  if (ja > j1 || ia > i1) return false;
  int rx[3], ry[3];
  float px[3], py[3];
  int amax = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rx[k] = x[k] - kSub * j0; ry[k] = y[k] - kSub * i0;
    amax = max(amax, max(abs(rx[k]), abs(ry[k])));
    px[k] = (float)rx[k] * (1.0f / (float)kSub);
    py[k] = (float)ry[k] * (1.0f / (float)kSub);
  }
  const bool big = amax > kSmallLimit;
  const float det = (float)area2 * (1.0f / (float)(kSub * kSub));
  const float inv = 1.0f / det;
  const PlaneQ W = make_plane(wk[0], wk[1], wk[2], px, py, inv);
  float q[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) q[k] = fmaf(bb[k][0], uv[0].x, fmaf(bb[k][1], uv[1].x, bb[k][2] * uv[2].x)) * wk[k];
  const PlaneQ NU = make_plane(q[0], q[1], q[2], px, py, inv);
#pragma unroll
  for (int k = 0; k < 3; ++k) q[k] = fmaf(bb[k][0], uv[0].y, fmaf(bb[k][1], uv[1].y, bb[k][2] * uv[2].y)) * wk[k];
  const PlaneQ NV = make_plane(q[0], q[1], q[2], px, py, inv);
  auto pk = [](int lo, int hi) { return (uint32_t)(lo & 0xFFFF) | ((uint32_t)hi << 16); };
  auto fb = [](float f) { return __float_as_uint(f); };
  // probe-count class of the anisotropic filter at the triangle's centroid (tw > 0: a mip-mapped texture is filtered): 0: at
  // most 4 probes, 1: more.  A HINT for the band kernel (it groups shading invocations by it); it changes no pixel.
  uint32_t cls = 0;
  if (tw > 0) {
    const float fx = (px[0] + px[1] + px[2]) * (1.0f / 3.0f), fy = (py[0] + py[1] + py[2]) * (1.0f / 3.0f);
    const float iw = __frcp_rn(plane_at(W, fx, fy));
    const float tu = plane_at(NU, fx, fy) * iw, tv = plane_at(NV, fx, fy) * iw;
    const float ux = (NU.qx - tu * W.qx) * iw * (float)tw, vx = (NV.qx - tv * W.qx) * iw * (float)th;
    const float uy = (NU.qy - tu * W.qy) * iw * (float)tw, vy = (NV.qy - tv * W.qy) * iw * (float)th;
    const float p2x = ux * ux + vx * vx, p2y = uy * uy + vy * vy;
    const float hi2 = fmaxf(p2x, p2y), lo2 = fminf(p2x, p2y);
#ifndef HP_CLS_RATIO2
#define HP_CLS_RATIO2 16.0f
#endif
    cls = (hi2 > HP_CLS_RATIO2 * lo2) ? 1u : 0u;  // Pmax / Pmin > 4: more than four probes = more than two rounds of the probe-pair loop
  }
#ifdef HP_SABL_NOREC
  if (a.w > 0 && fb(W.q0 + NU.q0 + NV.q0) != 0x12345u) { row_lo = ia; row_hi = i1; cols = j1 - ja + 1; return true; }
#endif
  rec[0] = make_uint4(pk(j0, i0), (big ? 1u : 0u) | (cls << 1), pk(rx[0], ry[0]), pk(rx[1], ry[1]));
  rec[1] = make_uint4(pk(rx[2], ry[2]), fb(W.q0), fb(W.qx), fb(W.qy));
  rec[2] = make_uint4(fb(NU.q0), fb(NU.qx), fb(NU.qy), fb(NV.q0));
  rec[3] = make_uint4(fb(NV.qx), fb(NV.qy), (uint32_t)tri.x, (uint32_t)tri.y);
  if (a.need_attr) {
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] = bb[k][1] * wk[k];
    const PlaneQ NB1 = make_plane(q[0], q[1], q[2], px, py, inv);
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] = bb[k][2] * wk[k];
    const PlaneQ NB2 = make_plane(q[0], q[1], q[2], px, py, inv);
    rec[4] = make_uint4(fb(NB1.q0), fb(NB1.qx), fb(NB1.qy), fb(NB2.q0));
    rec[5] = make_uint4(fb(NB2.qx), fb(NB2.qy), (uint32_t)tri.z, 0u);
  }
  if (big) {
    wrec[0] = make_uint4((uint32_t)rx[0], (uint32_t)ry[0], (uint32_t)rx[1], (uint32_t)ry[1]);
    wrec[1] = make_uint4((uint32_t)rx[2], (uint32_t)ry[2], 0u, 0u);
  }
  row_lo = ia; row_hi = i1; cols = j1 - ja + 1;
  return true;
}

struct ViewXform { float T[12], Kv[9]; bool finite; };

__device__ __forceinline__ ViewXform load_view(const RasterArgs& a, int view) {
  ViewXform x;
#pragma unroll
  for (int k = 0; k < 12; ++k) x.T[k] = a.TCO[16 * (int64_t)view + k];
#pragma unroll
  for (int k = 0; k < 9; ++k) x.Kv[k] = a.K[9 * (int64_t)view + k];
  bool finite = true;
#pragma unroll
  for (int k = 0; k < 12; ++k) finite &= isfinite(x.T[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) finite &= isfinite(a.TCO[16 * (int64_t)view + 12 + k]);
#pragma unroll
  for (int k = 0; k < 9; ++k) finite &= isfinite(x.Kv[k]);
  x.finite = finite;
  return x;
}

// the point of the edge I -> O on the plane z = near, interpolated FROM the inside corner TO the outside one
__device__ __forceinline__ Corner isect_near(const Corner& I, const Corner& O) {
  const float t = (I.c[2] - kZNear) / (I.c[2] - O.c[2]);
  Corner r;
  r.c[0] = fmaf(t, O.c[0] - I.c[0], I.c[0]);
  r.c[1] = fmaf(t, O.c[1] - I.c[1], I.c[1]);
  r.c[2] = kZNear;
#pragma unroll
  for (int c = 0; c < 3; ++c) r.b[c] = fmaf(t, O.b[c] - I.b[c], I.b[c]);
  return r;
}

// Pass 0: one lane per (view, vertex): camera + intrinsics transform, perspective division and snapping, once instead of once
// per triangle corner (a vertex is shared by six triangles).  Record: {x, y (1/256 px), bits(1 / z), bits(z)}.
__global__ __launch_bounds__(256) void raster_xform_kernel(RasterArgs a) {
  const int lv = blockIdx.y, view = a.view0 + lv;
  const int v = blockIdx.x * 256 + threadIdx.x;
  const int item = view / a.views_per_item;
  const int64_t* ob = a.obj + 8 * (int64_t)a.obj_ids[item];
  if (v >= (int)ob[1]) return;
  float T[12], Kv[9];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = a.TCO[16 * (int64_t)view + k];
#pragma unroll
  for (int k = 0; k < 9; ++k) Kv[k] = a.K[9 * (int64_t)view + k];
  const float4 p = a.verts4[ob[0] + v];
  Corner c;
  c.c[0] = fmaf(T[0], p.x, fmaf(T[1], p.y, fmaf(T[2], p.z, T[3])));
  c.c[1] = fmaf(T[4], p.x, fmaf(T[5], p.y, fmaf(T[6], p.z, T[7])));
  c.c[2] = fmaf(T[8], p.x, fmaf(T[9], p.y, fmaf(T[10], p.z, T[11])));
  c.b[0] = c.b[1] = c.b[2] = 0.f;
  const SnapCorner sc = project_corner(Kv, c);
  a.xverts[(int64_t)lv * a.max_verts + v] = make_int4(sc.x, sc.y, (int)__float_as_uint(sc.w), (int)__float_as_uint(c.c[2]));
}

// Pass 1: one lane per (view, triangle): near-plane clipping (rare), culling, plane set-up -> the
// sub-triangle's record, and its id appended to the list of every band its candidate rows touch (appends are aggregated
// per workgroup: one returning LDS atomic per touched band and lane, one global atomic per band and workgroup).  A view
// with a non-finite pose or intrinsics sets nothing up: zero images (panda3d_batch_renderer.py:81-111).
__global__ __launch_bounds__(kBinThreads) void raster_setup_kernel(RasterArgs a) {
  const int lv = blockIdx.y;               // view within the chunk
  const int view = a.view0 + lv;
  const int f = blockIdx.x * kBinThreads + threadIdx.x;
  const ViewXform x = load_view(a, view);
  const int item = view / a.views_per_item;
  const int64_t* ob = a.obj + 8 * (int64_t)a.obj_ids[item];
  const int nf = x.finite ? (int)ob[3] : 0;
  // the view may cull: some component of the object is closed (MeshStore::cull), the camera is outside the object's bounding
  // sphere, and the whole sphere lies beyond the near plane (a clipped object shows its inside)
  bool view_cull = false;
  if (a.cull && x.finite) {
    const float* const cu = a.cull + 8 * (int64_t)a.obj_ids[item];
    const float cx = fmaf(x.T[0], cu[0], fmaf(x.T[1], cu[1], fmaf(x.T[2], cu[2], x.T[3])));
    const float cy = fmaf(x.T[4], cu[0], fmaf(x.T[5], cu[1], fmaf(x.T[6], cu[2], x.T[7])));
    const float cz = fmaf(x.T[8], cu[0], fmaf(x.T[9], cu[1], fmaf(x.T[10], cu[2], x.T[11])));
    const float r = cu[3] * 1.001f + 1e-6f;  // (a scaled rotation is not expected in T; the margin covers its rounding)
    view_cull = cu[4] != 0.f && cx * cx + cy * cy + cz * cz > r * r && cz - r > kZNear;
  }
  int b0 = 1, b1 = 0;  // band range of the (first) sub-triangle: empty
  int c0 = 1, c1 = 0;  // ... of the second half of a near-clipped quad (rare)
  int lo = 0, hi = -1, cols = 0, lo2 = 0, hi2 = -1, cols2 = 0;  // candidate rows / columns of the two
  // a mip-mapped texture will be filtered anisotropically: the set-up classifies the triangles' probe counts
  const int ftw = (a.flags & HP_RASTER_TEX_ANISO) && ob[4] >= 0 && ob[7] > 1 ? (int)ob[5] : 0, fth = (int)ob[6];
  uint4* const recs = a.recs + (int64_t)lv * a.rec_slots * a.rec_q;
  uint4* const wide = a.recs_wide + (int64_t)lv * a.rec_slots * 2;
  if (f < nf) {
    const int4 tri = a.faces4[ob[2] + f];
    const int64_t voff = ob[0];
    const int4* const xv = a.xverts + (int64_t)lv * a.max_verts;
    const int4 v0 = xv[tri.x], v1 = xv[tri.y], v2 = xv[tri.z];
    const float z0 = __uint_as_float((uint32_t)v0.w), z1 = __uint_as_float((uint32_t)v1.w), z2 = __uint_as_float((uint32_t)v2.w);
    const float zmin = fminf(z0, fminf(z1, z2)), zmax = fmaxf(z0, fmaxf(z1, z2));
    if ((zmax >= kZNear) && (zmin <= kZFar)) {
      const float2 uv[3] = {*reinterpret_cast<const float2*>(a.uvs + 2 * (voff + tri.x)), *reinterpret_cast<const float2*>(a.uvs + 2 * (voff + tri.y)),
                            *reinterpret_cast<const float2*>(a.uvs + 2 * (voff + tri.z))};
      if (zmin >= kZNear) {
        const SnapCorner s0{v0.x, v0.y, __uint_as_float((uint32_t)v0.z), {1.f, 0.f, 0.f}};
        const SnapCorner s1{v1.x, v1.y, __uint_as_float((uint32_t)v1.z), {0.f, 1.f, 0.f}};
        const SnapCorner s2{v2.x, v2.y, __uint_as_float((uint32_t)v2.z), {0.f, 0.f, 1.f}};
        if (setup_subtri(a, s0, s1, s2, uv, tri, view_cull ? tri.w : 0, ftw, fth, recs + (int64_t)f * a.rec_q, wide + (int64_t)f * 2, lo, hi, cols)) {
          b0 = lo / a.band_rows; b1 = hi / a.band_rows;
        }
      } else {
        // near-plane clipping (oracle.c): camera-space corners again (the pre-pass keeps window coordinates only), rotated
        // cyclically so that corner 0 is inside and corner 2 outside; then the polygon is [V0, I(0->1), I(0->2)] or
        // [V0, V1, I(1->2), I(0->2)]
        const float4 pp[3] = {a.verts4[voff + tri.x], a.verts4[voff + tri.y], a.verts4[voff + tri.z]};
        Corner V[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          V[k].c[0] = fmaf(x.T[0], pp[k].x, fmaf(x.T[1], pp[k].y, fmaf(x.T[2], pp[k].z, x.T[3])));
          V[k].c[1] = fmaf(x.T[4], pp[k].x, fmaf(x.T[5], pp[k].y, fmaf(x.T[6], pp[k].z, x.T[7])));
          V[k].c[2] = fmaf(x.T[8], pp[k].x, fmaf(x.T[9], pp[k].y, fmaf(x.T[10], pp[k].z, x.T[11])));
#pragma unroll
          for (int c = 0; c < 3; ++c) V[k].b[c] = c == k ? 1.0f : 0.0f;
        }
        const bool in0 = V[0].c[2] >= kZNear, in1 = V[1].c[2] >= kZNear, in2 = V[2].c[2] >= kZNear;
        const int n_in = (int)in0 + (int)in1 + (int)in2;
        const int r = n_in == 1 ? (in0 ? 0 : in1 ? 1 : 2) : (!in0 ? 1 : !in1 ? 2 : 0);
        const Corner R0 = r == 0 ? V[0] : r == 1 ? V[1] : V[2];
        const Corner R1 = r == 0 ? V[1] : r == 1 ? V[2] : V[0];
        const Corner R2 = r == 0 ? V[2] : r == 1 ? V[0] : V[1];
        const SnapCorner S0 = project_corner(x.Kv, R0);
        if (n_in == 1) {
          const SnapCorner P1 = project_corner(x.Kv, isect_near(R0, R1)), P2 = project_corner(x.Kv, isect_near(R0, R2));
          if (setup_subtri(a, S0, P1, P2, uv, tri, 0, ftw, fth, recs + (int64_t)f * a.rec_q, wide + (int64_t)f * 2, lo, hi, cols)) { b0 = lo / a.band_rows; b1 = hi / a.band_rows; }
        } else {
          const SnapCorner S1 = project_corner(x.Kv, R1);
          const SnapCorner P2 = project_corner(x.Kv, isect_near(R1, R2)), P3 = project_corner(x.Kv, isect_near(R0, R2));
          if (setup_subtri(a, S0, S1, P2, uv, tri, 0, ftw, fth, recs + (int64_t)f * a.rec_q, wide + (int64_t)f * 2, lo, hi, cols)) { b0 = lo / a.band_rows; b1 = hi / a.band_rows; }
          if (setup_subtri(a, S0, P2, P3, uv, tri, 0, ftw, fth, recs + (int64_t)(nf + f) * a.rec_q, wide + (int64_t)(nf + f) * 2, lo2, hi2, cols2)) { c0 = lo2 / a.band_rows; c1 = hi2 / a.band_rows; }
        }
      }
    }
  }
  // Appends.  The order of a list is arbitrary (the z-buffer minimum {depth : id} does not depend on it) -- but the band kernel
  // walks the candidate pixels of 64 listed records per wave in lockstep, to the LARGEST box of the 64.  So a band's list is
  // filled from both ends: records with at most kSmallArea candidate pixels in that band from the front, the others from the
  // back (counters [band][0] and [band][1]): waves see boxes of similar size (round 5: 1.7x fewer walk steps).
  __shared__ int wg_cnt[2 * kMaxBands], wg_base[2 * kMaxBands];
  const int tid = threadIdx.x;
  for (int b = tid; b < 2 * a.n_bands; b += kBinThreads) wg_cnt[b] = 0;
  __syncthreads();
  auto band_class = [&](int b, int rlo, int rhi, int ncols) {  // 1: more than kSmallArea candidate pixels inside band b
    const int r0 = max(rlo, b * a.band_rows), r1 = min(rhi, (b + 1) * a.band_rows - 1);
    return (r1 - r0 + 1) * ncols > kSmallArea ? 1 : 0;
  };
  int32_t* const cnt_v = a.bin_count + (int64_t)lv * a.n_bands * 2;
  int32_t* const list_v = a.bin_list + (int64_t)lv * a.n_bands * a.bin_cap;
  auto put_direct = [&](int b, int cls, int id) {  // one global atomic per append (rare paths)
    const int slot = atomicAdd(&cnt_v[2 * b + cls], 1);
    if (slot + cnt_v[2 * b + 1 - cls] < a.bin_cap) list_v[(int64_t)b * a.bin_cap + (cls ? a.bin_cap - 1 - slot : slot)] = id;
  };
  int slot[8], kls[8];  // local slots / classes in the bands b0 .. b0 + 7
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    slot[k] = 0; kls[k] = 0;
    if (b0 <= b1 && b0 + k <= b1) {
      kls[k] = band_class(b0 + k, lo, hi, cols);
      slot[k] = atomicAdd(&wg_cnt[2 * (b0 + k) + kls[k]], 1);
    }
  }
  for (int b = b0 + 8; b <= b1; ++b) put_direct(b, band_class(b, lo, hi, cols), f);  // rare: a triangle taller than eight bands
  for (int b = c0; b <= c1; ++b) put_direct(b, band_class(b, lo2, hi2, cols2), nf + f);  // rarer: the second half of a near-clipped quad
  __syncthreads();
  for (int b = tid; b < 2 * a.n_bands; b += kBinThreads)
    if (wg_cnt[b] > 0) wg_base[b] = atomicAdd(&cnt_v[b], wg_cnt[b]);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (b0 <= b1 && b0 + k <= b1) {
      const int b = b0 + k, sl = wg_base[2 * b + kls[k]] + slot[k];
      // (a record id appends at most once per band and bin_cap = rec_slots = all the ids there are -- f and nf + f, the two halves
      // of a near-clipped quad, can both land in a band --: the two ends cannot meet, the bound checks are defensive only)
      if (sl < a.bin_cap) list_v[(int64_t)b * a.bin_cap + (kls[k] ? a.bin_cap - 1 - sl : sl)] = f;
    }
}

// One fragment-shader invocation: colour and normal code of sub-triangle `id` at the CENTRE of pixel (i, j) (attributes
// extrapolated when the centre lies outside the triangle: multisampled edge pixels) -- oracle.c shade_centre.
struct ShadeCtx {
  const float* T; const float* Kv; const float* amb; const uint4* recs;
  int64_t voff, toff; int tw, th, view, q8, nlev, aniso;
  const MipTable* mips;  // per-workgroup table of the object's mip levels (LDS)
  const uint8_t* texq;   // the object's row-pair texture copy (mips->p2)
  bool need_normal;      // the view renders normals or has point lights: otherwise the normal is never looked at
};
template <bool ANISO, class A>
__device__ __forceinline__ void shade_centre(const A& a, const ShadeCtx& cx, int id, int i, int j, float* o_rgb, float* o_n) {
  const float* T = cx.T; const float* Kv = cx.Kv; const float* amb = cx.amb;
  const int64_t voff = cx.voff, toff = cx.toff;
  const int tw = cx.tw, th = cx.th, view = cx.view, q8 = cx.q8;
  const uint4* const r = cx.recs + (int64_t)(id & 0x3FFFFFFF) * a.rec_q;  // (bit 30 of a key's low word: the probe-count class)
#ifdef HP_REC_SCOPE  // diagnostics: the record read word by word with scoped atomic loads (1: agent = L2-served, 2: system = memory-served)
  auto ldw = [&](int k) {
    const uint32_t* wp = reinterpret_cast<const uint32_t*>(r) + k;
    return HP_REC_SCOPE == 2 ? __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  const uint4 r0 = make_uint4(ldw(0), ldw(1), ldw(2), ldw(3)), r1 = make_uint4(ldw(4), ldw(5), ldw(6), ldw(7));
  const uint4 r2 = make_uint4(ldw(8), ldw(9), ldw(10), ldw(11)), r3 = make_uint4(ldw(12), ldw(13), ldw(14), ldw(15));
#else
  const uint4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
#endif
  const int ox = (int)(short)(r0.x & 0xFFFFu), oy = (int)r0.x >> 16;
  const PlaneQ W{__uint_as_float(r1.y), __uint_as_float(r1.z), __uint_as_float(r1.w)};
  const float fx = (float)(j - ox) + 0.5f, fy = (float)(i - oy) + 0.5f;
  const float Wc = plane_at(W, fx, fy);
  const float iw = 1.0f / Wc;
  const bool need_b = cx.need_normal || toff < 0;
  float b0 = 0.f, b1 = 0.f, b2 = 0.f;
  int64_t g0 = 0, g1 = 0, g2 = 0;
  if (need_b) {
    const uint4 r4 = r[4], r5 = r[5];
    const PlaneQ NB1{__uint_as_float(r4.x), __uint_as_float(r4.y), __uint_as_float(r4.z)};
    const PlaneQ NB2{__uint_as_float(r4.w), __uint_as_float(r5.x), __uint_as_float(r5.y)};
    b1 = plane_at(NB1, fx, fy) * iw; b2 = plane_at(NB2, fx, fy) * iw;
    b0 = (1.0f - b1) - b2;
    g0 = voff + (int)r3.z; g1 = voff + (int)r3.w; g2 = voff + (int)r5.z;
  }
  float alb[3];
  if (toff >= 0) {
    const PlaneQ NU{__uint_as_float(r2.x), __uint_as_float(r2.y), __uint_as_float(r2.z)};
    const PlaneQ NV{__uint_as_float(r2.w), __uint_as_float(r3.x), __uint_as_float(r3.y)};
    const float tu = plane_at(NU, fx, fy) * iw, tv = plane_at(NV, fx, fy) * iw;
    if (ANISO && cx.nlev > 1) {
      // screen-space derivatives of the perspective-correct coordinates: u = N_u / W, both affine in (x, y)
      const float ux = fmaf(-tu, W.qx, NU.qx) * iw, vx = fmaf(-tv, W.qx, NV.qx) * iw;
      const float uy = fmaf(-tu, W.qy, NU.qy) * iw, vy = fmaf(-tv, W.qy, NV.qy) * iw;
      if (cx.mips->p2) tex_fetch_aniso_p2(a.cv, cx.texq, *cx.mips, tw, th, cx.nlev, tu, tv, ux, vx, uy, vy, alb);
      else tex_fetch_aniso(a.cv, a.tex + toff, tw, th, cx.nlev, tu, tv, ux, vx, uy, vy, alb);
    } else {
      tex_fetch(a.tex + toff, tw, th, tu, tv, alb);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      alb[c] = fmaf(b0, (float)a.colors[4 * g0 + c],
                    fmaf(b1, (float)a.colors[4 * g1 + c], b2 * (float)a.colors[4 * g2 + c])) / 255.0f;  // barycentrics may extrapolate: plain division
  }
  float no[3], nc[3] = {0.f, 0.f, 0.f};
  if (cx.need_normal) {
    const float4 n0 = a.normals4[g0], n1 = a.normals4[g1], n2 = a.normals4[g2];
    no[0] = fmaf(b0, n0.x, fmaf(b1, n1.x, b2 * n2.x));
    no[1] = fmaf(b0, n0.y, fmaf(b1, n1.y, b2 * n2.y));
    no[2] = fmaf(b0, n0.z, fmaf(b1, n1.z, b2 * n2.z));
    nc[0] = fmaf(T[0], no[0], fmaf(T[1], no[1], T[2] * no[2]));
    nc[1] = fmaf(T[4], no[0], fmaf(T[5], no[1], T[6] * no[2]));
    nc[2] = fmaf(T[8], no[0], fmaf(T[9], no[1], T[10] * no[2]));
    float nn = sqrtf(fmaf(nc[0], nc[0], fmaf(nc[1], nc[1], nc[2] * nc[2])));
    if (nn > 0.0f) { nc[0] /= nn; nc[1] /= nn; nc[2] /= nn; }
  }
  float lit[3] = {amb[0], amb[1], amb[2]};
  if (a.n_lights > 0) {
    const float pu = (float)j + 0.5f, pv = (float)i + 0.5f, Z = iw;
    float py = (pv - Kv[5]) * Z / Kv[4];
    float px = ((pu - Kv[2]) * Z - Kv[1] * py) / Kv[0];
    for (int l = 0; l < a.n_lights; ++l) {
      const float* lp = a.light_pos + 3 * ((int64_t)view * a.n_lights + l);
      const float* lc = a.light_col + 3 * ((int64_t)view * a.n_lights + l);
      float lx = fmaf(T[0], lp[0], fmaf(T[1], lp[1], fmaf(T[2], lp[2], T[3]))) - px;
      float ly = fmaf(T[4], lp[0], fmaf(T[5], lp[1], fmaf(T[6], lp[2], T[7]))) - py;
      float lz = fmaf(T[8], lp[0], fmaf(T[9], lp[1], fmaf(T[10], lp[2], T[11]))) - Z;
      float ln = sqrtf(fmaf(lx, lx, fmaf(ly, ly, lz * lz)));
      float ndl = ln > 0.0f ? fmaf(nc[0], lx, fmaf(nc[1], ly, nc[2] * lz)) / ln : 0.0f;
      if (ndl > 0.0f) {
#pragma unroll
        for (int c = 0; c < 3; ++c) lit[c] = fmaf(lc[c], ndl, lit[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) o_rgb[c] = quant8(alb[c] * lit[c], q8);
  if (cx.need_normal) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {  // default: (nx, -ny, -nz), GL eye space seen from the OpenCV camera frame
      const int ax = a.cv.n_axis[c];
      o_n[c] = quant8(normal_code(a.cv.n_sign[c] * (ax == 0 ? nc[0] : ax == 1 ? nc[1] : nc[2])), q8);
    }
  } else {
    o_n[0] = o_n[1] = o_n[2] = 0.f;
  }
}

// ---- the band kernel ------------------------------------------------------------------------------------------------
// LDS of a workgroup (dynamic): zb[npix_max * NS] u64 | ex[npix_max] u16 | plist[npix_max] u16 | folds_y[rows] | folds_x[w]
// zb: during coverage the 64-bit keys {depth bits : triangle id}; after shading the centre slot of a pixel holds
// {depth bits (0xFFFFFFFF = no depth) : r | g << 8 | b << 16 | nx << 24} and ex holds ny | nz << 8 (8-bit colour codes).
struct BandLds {
  unsigned long long* zb; unsigned short* ex; unsigned short* plist; Fold* fy; Fold* fx;
  uint32_t* nsum;  // multisampling: per pixel, the three 10-bit sums of the normal codes of its samples
};
// list entries per pixel: the covered pixels (single sample) or the shading invocations {pixel | sample << 14} (multisampling: <= 4)
__host__ __device__ constexpr int band_list_per_pixel(int ns) { return ns == 1 ? 1 : 4; }
__device__ __forceinline__ BandLds carve_lds(unsigned char* base, int npix_max, int ns, int rows) {
  BandLds l;
  l.zb = reinterpret_cast<unsigned long long*>(base);
  base += (size_t)npix_max * ns * 8;
  l.ex = reinterpret_cast<unsigned short*>(base);
  base += (size_t)((npix_max * 2 + 15) & ~15);  // odd render widths: keep plist / nsum / the folds 16-B aligned
  l.plist = reinterpret_cast<unsigned short*>(base);
  base += (size_t)((npix_max * band_list_per_pixel(ns) * 2 + 15) & ~15);
  l.nsum = reinterpret_cast<uint32_t*>(base);
  if (ns > 1) base += (size_t)npix_max * 4;
  l.fy = reinterpret_cast<Fold*>(base);
  l.fx = l.fy + rows;
  return l;
}
static size_t band_lds_bytes(int npix_max, int ns, int rows, int w, bool crop) {
  return (size_t)npix_max * ns * 8 + (size_t)((npix_max * 2 + 15) & ~15) + (size_t)((npix_max * band_list_per_pixel(ns) * 2 + 15) & ~15) +
         (ns > 1 ? (size_t)npix_max * 4 : 0) + (crop ? (size_t)(rows + w) * sizeof(Fold) : 0) + 16;
}

__device__ __forceinline__ unsigned code8(float c) {  // quant8(c) = code8(c) / 255
  c = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
  return (unsigned)floorf(fmaf(c, 255.0f, 0.5f));
}

// stores n consecutive floats (n <= 8) at dst, 4-B aligned: the widest pieces first
__device__ __forceinline__ void store_run(float* dst, const float* v, int n) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f3 __attribute__((ext_vector_type(3)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  switch (n) {
    case 1: dst[0] = v[0]; break;
    case 2: *reinterpret_cast<f2*>(dst) = f2{v[0], v[1]}; break;
    case 3: *reinterpret_cast<f3*>(dst) = f3{v[0], v[1], v[2]}; break;
    case 4: *reinterpret_cast<f4*>(dst) = f4{v[0], v[1], v[2], v[3]}; break;
    case 5: *reinterpret_cast<f4*>(dst) = f4{v[0], v[1], v[2], v[3]}; dst[4] = v[4]; break;
    case 6: *reinterpret_cast<f4*>(dst) = f4{v[0], v[1], v[2], v[3]}; *reinterpret_cast<f2*>(dst + 4) = f2{v[4], v[5]}; break;
    case 7: *reinterpret_cast<f4*>(dst) = f4{v[0], v[1], v[2], v[3]}; *reinterpret_cast<f3*>(dst + 4) = f3{v[4], v[5], v[6]}; break;
    case 8: *reinterpret_cast<f4*>(dst) = f4{v[0], v[1], v[2], v[3]}; *reinterpret_cast<f4*>(dst + 4) = f4{v[4], v[5], v[6], v[7]}; break;
    default: break;
  }
}

// The separable roi_align taps of one output pixel for NC consecutive source channels starting at `sc0` (crop_math.h: the
// same taps in the same order as crop_tile_kernel).  Branch-free inside: the loops run to the band's largest spans (uniform
// bounds), taps past a pixel's own span carry weight 0 and read a clamped -- in-image -- address, so the loads of a row are
// issued together instead of one basic block (and one L2 round trip) per tap.
template <int NC>
__device__ __forceinline__ void crop_taps(const float* __restrict__ img, int HW, int IW, int IH, int sc0, const Fold& fy, const Fold& fx,
                                          int nr_max, int nc_max, float (&acc)[4], float& vacc) {
#pragma unroll
  for (int r = 0; r < kSpan; ++r) {
    if (r >= nr_max) break;
    const int rr = min(fy.first + r, IH - 1);
    float racc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) racc[c] = 0.f;
    float rv = 0.0f;
#pragma unroll
    for (int c2 = 0; c2 < kSpan; ++c2) {
      if (c2 >= nc_max) break;
      const int off = rr * IW + min(fx.first + c2, IW - 1);
      const float wj = fx.w[c2];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
#ifdef HP_RABL_CROP_NOLOAD
        const float v = wj + (float)off;
#else
        const float v = img[(sc0 + c) * HW + off];
#endif
        racc[c] += wj * v;
        if (sc0 + c == 3) rv += wj * (v > 0.0f ? 1.0f : 0.0f);
      }
    }
    const float wi = fy.w[r];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] += wi * racc[c];
    vacc += wi * rv;
  }
}

#ifdef HP_RASTER_STAMPS
// -DHP_RASTER_STAMPS (tools/raster_stamps.py): shader cycles between the phases of the band kernel, summed over the workgroups
// (thread 0's clock), [phase] and workgroup count in [15]
__device__ unsigned long long hp_rstamp[16];
#define HP_STAMP(k)                                                                         \
  do {                                                                                      \
    if (threadIdx.x == 0) {                                                                 \
      const unsigned long long now_ = __builtin_readcyclecounter();                          \
      atomicAdd(&hp_rstamp[k], now_ - stamp_t_);                                            \
      stamp_t_ = now_;                                                                      \
    }                                                                                       \
  } while (0)
#else
#define HP_STAMP(k) do { } while (0)
#endif

// ---- coverage ---------------------------------------------------------------------------------------------------------
// The three edge functions of a record in the form E_k(X, Y) = A_k X + B_k Y + C_k on sub-pixel coordinates relative to the
// origin, the top-left rule folded into C_k (- 1 on the edges that do not own their points): inside <=> all E_k >= 0.
// Values: |A|, |B| <= 2^15, |C| < 2^29, X, Y < 2^14 for a small record -- everything fits 32 bits, operands fit 24.
struct EdgeSet { int A[3], B[3], C[3]; };
__device__ __forceinline__ void edge_setup(const int (&rx)[3], const int (&ry)[3], EdgeSet& e) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int ia = (k + 1) % 3, ib = (k + 2) % 3;
    const int ex = rx[ib] - rx[ia], ey = ry[ib] - ry[ia];
    const bool owns = ey > 0 || (ey == 0 && ex < 0);
    e.A[k] = -ey; e.B[k] = ex;
    e.C[k] = __mul24(ey, rx[ia]) - __mul24(ex, ry[ia]) - (owns ? 0 : 1);
  }
}

// candidate pixels of a record inside the band: columns [jlo, jhi], rows [ilo, ihi] relative to the origin pixel
template <class A>
__device__ __forceinline__ bool candidate_range(const A& a, const int (&rx)[3], const int (&ry)[3], int ox, int oy, int row0, int row1,
                                                int& jlo, int& jhi, int& ilo, int& ihi) {
  const int xmin = min(rx[0], min(rx[1], rx[2])), xmax = max(rx[0], max(rx[1], rx[2]));
  const int ymin = min(ry[0], min(ry[1], ry[2])), ymax = max(ry[0], max(ry[1], ry[2]));
  jlo = max(-((a.cv.hi_x - xmin) >> 8), 0); jhi = min((xmax - a.cv.lo_x) >> 8, a.w - 1 - ox);
  ilo = max(max(-((a.cv.hi_y - ymin) >> 8), 0), row0 - oy); ihi = min(min((ymax - a.cv.lo_y) >> 8, a.h - 1 - oy), row1 - oy);
  return jlo <= jhi && ilo <= ihi;
}

// HALF: fp16 destinations (the input of an fp16 network plan) -- its own instantiation so that the fp32 path's register
// budget does not carry the 16-half record assembly
template <int NS, bool HALF, bool ANISO, bool WIDE>
__global__ __launch_bounds__(band_threads(NS, WIDE), (NS == 1 && !HALF && !ANISO) ? 6 : 4) void raster_kernel(RasterArgs a_in, int npix_max) {
  constexpr int kThreads = band_threads(NS, WIDE);
#ifdef HP_RASTER_STAMPS
  unsigned long long stamp_t_ = __builtin_readcyclecounter();
  if (threadIdx.x == 0) atomicAdd(&hp_rstamp[15], 1ull);
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int big_q[kBigQueue];
  __shared__ int big_n, n_cov, n_cov_hi, span_max[2];  // n_cov / n_cov_hi: list entries of the two probe-count classes (front / back)
  __shared__ MipTable mips;

  // XCD-aware renumbering: dispatch order b -> XCD b % 8; give each XCD a contiguous range.  (One workgroup per (view, band):
  // a persistent grid pulling items from per-XCD queues was built in round 5 and LOST -- 1110 us per 128 views with the queue
  // atomics on the workgroup's critical path, 531 us with static striding, against 446 us: the hardware dispatcher balances
  // the uneven bands better, and four resident workgroups per CU already hide each other's start-up.)
  const RasterArgs& a = a_in;
  const int total = a.n * a.n_bands;  // a.n = views of this chunk
  const int per_xcd = (total + 7) / 8;
  const int lin = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lin >= total) return;
  const int tid = threadIdx.x;
  const int lv = lin / a.n_bands;
  const int view = a.view0 + lv;
  const int band = lin % a.n_bands;
  const int row0 = band * a.band_rows;
  const int row1 = min(a.h, row0 + a.band_rows) - 1;
  const int npix = (row1 - row0 + 1) * a.w;
  const BandLds L = carve_lds(smem, npix_max, NS, a.band_rows);
  unsigned long long* const zb = L.zb;

  // the band's list: the first entry's loads are issued before anything else (a non-finite view set nothing up: cnt = 0)
  const int cnt_s = min(a.bin_count[2 * lin], a.bin_cap), cnt = cnt_s + min(a.bin_count[2 * lin + 1], a.bin_cap - cnt_s);
  const int32_t* const list = a.bin_list + (int64_t)lin * a.bin_cap;
  const uint4* const recs = a.recs + (int64_t)lv * a.rec_slots * a.rec_q;
  auto entry = [&](int k) { return k < cnt_s ? list[k] : list[a.bin_cap - 1 - (k - cnt_s)]; };  // small boxes first, the others from the back
  int id_n = tid < cnt ? entry(tid) : 0;
  uint4 q0_n = recs[(int64_t)id_n * a.rec_q], q1_n = recs[(int64_t)id_n * a.rec_q + 1];

  const int item = view / a.views_per_item, vi = view % a.views_per_item;
  const int64_t* ob = a.obj + 8 * (int64_t)a.obj_ids[item];
  const int64_t voff = ob[0], toff = ob[4];
  const int tw = (int)ob[5], th = (int)ob[6];

  // ---- fused crop: folded roi_align weights of the band's rows and of every column (crop_math.h) ----
  const int ncrop = (a.rec && a.images && vi < kMaxViews) ? a.v_crop_n[vi] : 0;
  float cx1 = 0.f, cy1 = 0.f, bin_h = 1.f, bin_w = 1.f;
  if (ncrop > 0) {
    const float* box = a.boxes + 4 * (int64_t)item;
    cx1 = box[0]; cy1 = box[1];
    float roi_w = box[2] - cx1, roi_h = box[3] - cy1;
    roi_w = roi_w < 1.0f ? 1.0f : roi_w;  // aligned=False
    roi_h = roi_h < 1.0f ? 1.0f : roi_h;
    bin_h = roi_h / (float)a.h; bin_w = roi_w / (float)a.w;
    if (tid < 2) span_max[tid] = 0;
    __syncthreads();
    const int nrows = row1 - row0 + 1;
    for (int t = tid; t < nrows + a.w; t += kThreads) {
      const bool is_x = t >= nrows;
      const int k = is_x ? t - nrows : t;
      Axis ax;
      Fold f;
      if (is_x) make_axis(cx1, k, bin_w, a.sr, a.IW, ax);
      else make_axis(cy1, row0 + k, bin_h, a.sr, a.IH, ax);
      fold_axis(ax, a.sr, f.first, f.span, f.w);
      if (is_x) L.fx[k] = f; else L.fy[k] = f;
      atomicMax(&span_max[is_x ? 1 : 0], f.span);
    }
  }

  // a band no triangle touches skips the z-buffer passes altogether: the output pass streams the background (and the crop)
  const bool band_empty = cnt == 0;
  if (!band_empty) {
    for (int p = tid; p < npix * NS; p += kThreads) zb[p] = kKeyEmpty;
    if (tid == 0) { big_n = 0; n_cov = 0; n_cov_hi = 0; }
  }
  __syncthreads();
  HP_STAMP(0);  // head: list / record loads in flight, z-buffer initialised
  if (!band_empty && tid == 0) { a.bin_count[2 * lin] = 0; a.bin_count[2 * lin + 1] = 0; }  // consumed (every thread has read them by now): zero for the next launch
#ifdef HP_RABL_NO_COVER
  const int cnt_loop = 0;
#else
  const int cnt_loop = cnt;
#endif
  const float w_near = 1.0f / kZNear, w_far = 1.0f / kZFar;
  // ---- coverage + depth: a lane owns a listed record; the next one's loads are in flight while it walks ----
  for (int k = tid; k < cnt_loop; k += kThreads) {
    const int id = id_n;
    const uint4 q0 = q0_n, q1 = q1_n;
    if (k + kThreads < cnt_loop) {
      id_n = entry(k + kThreads);
      q0_n = recs[(int64_t)id_n * a.rec_q]; q1_n = recs[(int64_t)id_n * a.rec_q + 1];
    }
    const int ox = (int)(short)(q0.x & 0xFFFFu), oy = (int)q0.x >> 16;
    const uint32_t key_lo = (uint32_t)id | ((q0.y & 2u) << 29);  // bit 30: the probe-count class (shading groups invocations by it)
    if (q0.y & 1u) {  // big: corners beyond the 32-bit range of the edge functions -> cooperative walk in 64 bits
      const int q = atomicAdd(&big_n, 1);
      if (q < kBigQueue) big_q[q] = id;
      continue;
    }
    const int rx[3] = {(int)(short)(q0.z & 0xFFFFu), (int)(short)(q0.w & 0xFFFFu), (int)(short)(q1.x & 0xFFFFu)};
    const int ry[3] = {(int)q0.z >> 16, (int)q0.w >> 16, (int)q1.x >> 16};
    int jlo, jhi, ilo, ihi;
    if (!candidate_range(a, rx, ry, ox, oy, row0, row1, jlo, jhi, ilo, ihi)) continue;
    if ((jhi - jlo + 1) * (ihi - ilo + 1) > kBigArea) {
      const int q = atomicAdd(&big_n, 1);
      if (q < kBigQueue) { big_q[q] = id; continue; }
    }
    EdgeSet e;
    edge_setup(rx, ry, e);
    const float W0 = __uint_as_float(q1.y), Wx = __uint_as_float(q1.z), Wy = __uint_as_float(q1.w);
    // the samples' offsets from the pixel's corner, per edge
    int off[3][NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int sl = NS == 1 ? 4 : s;
#pragma unroll
      for (int c = 0; c < 3; ++c) off[c][s] = __mul24(e.A[c], a.cv.sxi[sl]) + __mul24(e.B[c], a.cv.syi[sl]);
    }
    // one candidate pixel per step, row-major over the box (a nested row / column loop runs, per wave, the largest column
    // count of every row step: 2.5x the steps)
    const int bw = jhi - jlo + 1;
    int di = ilo, dj = jlo;
    for (int left = bw * (ihi - ilo + 1); left > 0; --left) {
      const int X0 = dj << 8, Y0 = di << 8;
      const int e0 = __mul24(e.A[0], X0) + __mul24(e.B[0], Y0) + e.C[0];
      const int e1 = __mul24(e.A[1], X0) + __mul24(e.B[1], Y0) + e.C[1];
      const int e2 = __mul24(e.A[2], X0) + __mul24(e.B[2], Y0) + e.C[2];
      unsigned in = 0;
#pragma unroll
      for (int s = 0; s < NS; ++s) in |= (((e0 + off[0][s]) | (e1 + off[1][s]) | (e2 + off[2][s])) >= 0 ? 1u : 0u) << s;
      if (in) {
        const float dif = (float)di, djf = (float)dj;
        unsigned long long* const zpix = zb + (size_t)((oy + di - row0) * a.w + ox + dj) * NS;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int sl = NS == 1 ? 4 : s;
          const float Ws = fmaf(Wy, dif + a.cv.syf[sl], fmaf(Wx, djf + a.cv.sxf[sl], W0));
          if (((in >> s) & 1u) && (Ws <= w_near) && (Ws >= w_far))
            atomicMin(&zpix[s], ((unsigned long long)(~__float_as_uint(Ws)) << 32) | key_lo);
        }
      }
      if (++dj > jhi) { dj = jlo; ++di; }
    }
  }
  if (!band_empty) __syncthreads();
  HP_STAMP(1);  // coverage walk
  // ---- large footprints and far-reaching corners: the whole workgroup walks the candidate pixels, edge functions in
  // double precision (exact: |A X| < 2^43) ----
  const int nbig = band_empty ? 0 : min(big_n, kBigQueue);
  for (int q = 0; q < nbig; ++q) {
    const int id = big_q[q];
    const uint4* const r = recs + (int64_t)id * a.rec_q;
    const uint4 q0 = r[0], q1 = r[1];
    const int ox = (int)(short)(q0.x & 0xFFFFu), oy = (int)q0.x >> 16;
    const uint32_t key_lo = (uint32_t)id | ((q0.y & 2u) << 29);
    int rx[3], ry[3];
    if (q0.y & 1u) {
      const uint4* const wr = a.recs_wide + ((int64_t)lv * a.rec_slots + id) * 2;
      const uint4 q6 = wr[0], q7 = wr[1];
      rx[0] = (int)q6.x; ry[0] = (int)q6.y; rx[1] = (int)q6.z; ry[1] = (int)q6.w; rx[2] = (int)q7.x; ry[2] = (int)q7.y;
    } else {
      rx[0] = (int)(short)(q0.z & 0xFFFFu); rx[1] = (int)(short)(q0.w & 0xFFFFu); rx[2] = (int)(short)(q1.x & 0xFFFFu);
      ry[0] = (int)q0.z >> 16; ry[1] = (int)q0.w >> 16; ry[2] = (int)q1.x >> 16;
    }
    int jlo, jhi, ilo, ihi;
    if (!candidate_range(a, rx, ry, ox, oy, row0, row1, jlo, jhi, ilo, ihi)) continue;
    double A[3], B[3], C[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int ia = (c + 1) % 3, ib = (c + 2) % 3;
      const int ex = rx[ib] - rx[ia], ey = ry[ib] - ry[ia];
      const bool owns = ey > 0 || (ey == 0 && ex < 0);
      A[c] = -(double)ey; B[c] = (double)ex;
      C[c] = (double)ey * (double)rx[ia] - (double)ex * (double)ry[ia] - (owns ? 0.0 : 1.0);
    }
    const float W0 = __uint_as_float(q1.y), Wx = __uint_as_float(q1.z), Wy = __uint_as_float(q1.w);
    const int bw = jhi - jlo + 1;
    const int area = bw * (ihi - ilo + 1);
    for (int p = tid; p < area; p += kThreads) {
      const int pr = p / bw;
      const int di = ilo + pr, dj = jlo + (p - pr * bw);
      unsigned long long* const zpix = zb + (size_t)((oy + di - row0) * a.w + ox + dj) * NS;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int sl = NS == 1 ? 4 : s;
        const double X = (double)((dj << 8) + a.cv.sxi[sl]), Y = (double)((di << 8) + a.cv.syi[sl]);
        const double e0 = fma(A[0], X, fma(B[0], Y, C[0])), e1 = fma(A[1], X, fma(B[1], Y, C[1])), e2 = fma(A[2], X, fma(B[2], Y, C[2]));
        if (e0 < 0.0 || e1 < 0.0 || e2 < 0.0) continue;
        const float Ws = fmaf(Wy, (float)di + a.cv.syf[sl], fmaf(Wx, (float)dj + a.cv.sxf[sl], W0));
        if (!(Ws <= w_near) || !(Ws >= w_far)) continue;
        atomicMin(&zpix[s], ((unsigned long long)(~__float_as_uint(Ws)) << 32) | key_lo);
      }
    }
  }
  if (!band_empty) __syncthreads();
  HP_STAMP(2);  // cooperative walk of large footprints

  // ---- the object's mip table (LDS), filled while the coverage results settle: its inputs hang on a chain of dependent scalar
  // loads (obj_ids -> obj row) that used to sit in front of the kernel's first barrier ----
  const int64_t qbase = a.tex_quads_off ? a.tex_quads_off[a.obj_ids[item]] : -1;
  if (tid < 16) {  // level k of the texture: max(1, tw >> k) x max(1, th >> k), stored behind the levels before it
    int off = 0, qo = 0, lw = tw, lh = th;
    for (int k = 0; k < tid; ++k) { off += 4 * lw * lh; qo += 8 * (lw + 1) * lh; lw = lw > 1 ? lw >> 1 : 1; lh = lh > 1 ? lh >> 1 : 1; }
    mips.off[tid] = off; mips.w[tid] = lw; mips.h[tid] = lh; mips.qoff[tid] = qo;
#ifdef HP_TEX_NO_QUADS
    if (tid == 0) mips.p2 = 0;
#else
    if (tid == 0) mips.p2 = qbase >= 0;
#endif
  }
  if (ANISO && tid < 256) {
    const int N = (tid >> 4) + 1, i = (tid & 15) + 1;
    mips.tt[N - 1][i - 1] = (float)i / (float)(N + 1) - 0.5f;
  }
  // ---- shading of the compacted covered pixels (8-bit colour codes back into the z-buffer slots) ----
  const int q8 = 1;  // colours are 8-bit quantised like the reference's uint8 read-back (HP_RASTER_QUANT8 is implied)
  float amb[3] = {1.0f, 1.0f, 1.0f};
  if (a.ambient) { amb[0] = a.ambient[3 * view]; amb[1] = a.ambient[3 * view + 1]; amb[2] = a.ambient[3 * view + 2]; }
  const bool need_normal = (a.rec ? a.want_nrm != 0 : a.nrm != nullptr) || a.n_lights > 0;
  const ShadeCtx cx{a.TCO + 16 * (int64_t)view, a.K + 9 * (int64_t)view, amb, recs, voff, toff, tw, th, view, q8, (int)ob[7] > 0 ? (int)ob[7] : 1,
                    (a.flags & HP_RASTER_TEX_ANISO) != 0, &mips, a.tex_quads + (qbase >= 0 ? qbase : 0), need_normal};
  const bool want_colour = a.rec ? true : (a.rgb != nullptr || a.nrm != nullptr);
  const bool coded = !band_empty;  // colours travel as 8-bit codes through LDS
  const int list_cap = npix_max * band_list_per_pixel(NS);  // entries of L.plist
  if (coded && NS == 1) {
    const int lane = tid & 63;
    for (int p0 = 0; p0 < npix; p0 += kThreads) {
      const int p = p0 + tid;
      bool cov = false;
      if (p < npix) {
        cov = zb[p] != kKeyEmpty && want_colour;
        if (!cov) {  // no fragment: colour codes 0, the depth bits stay
          zb[p] &= 0xFFFFFFFF00000000ull;
          L.ex[p] = 0;
        }
      }
      // two lists in one array: invocations of triangles with at most two anisotropic probes from the front, the others from
      // the back -- the probe loop of a wave runs to its largest probe count (the class is bit 30 of the key's low word)
      const bool hi = ANISO && cov && ((uint32_t)zb[p < npix ? p : 0] >> 30) != 0;
      const unsigned long long m = __ballot(cov && !hi), mh = ANISO ? __ballot(hi) : 0ull;
      int base = 0, base_h = 0;
      if (lane == 0 && m != 0) base = atomicAdd(&n_cov, __popcll(m));
      if (ANISO && lane == 0 && mh != 0) base_h = atomicAdd(&n_cov_hi, __popcll(mh));
      base = __shfl(base, 0);
      if (cov && !hi) L.plist[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)p;
      if (ANISO) {
        base_h = __shfl(base_h, 0);
        if (hi) L.plist[list_cap - 1 - (base_h + __popcll(mh & ((1ull << lane) - 1ull)))] = (unsigned short)p;
      }
    }
    __syncthreads();
    HP_STAMP(3);  // compaction (single sample)
    const int ncov_lo = n_cov, ncov = ncov_lo + (ANISO ? n_cov_hi : 0);
#ifdef HP_RABL_NO_SHADE
    const int ncov_loop = a.w < 0 ? ncov : 0;
#else
    const int ncov_loop = ncov;
#endif
    for (int q = tid; q < ncov_loop; q += kThreads) {
      const int p = L.plist[q < ncov_lo ? q : list_cap - 1 - (q - ncov_lo)];
      const int pr = (int)(((unsigned long long)p * a.w_magic) >> 32);
      const int i = row0 + pr, j = p - pr * a.w;
      unsigned cr[3], cn[3];
      float o_rgb[3], o_n[3];
      shade_centre<ANISO>(a, cx, (int)(zb[p] & 0xFFFFFFFFull), i, j, o_rgb, o_n);
#pragma unroll
      for (int c = 0; c < 3; ++c) { cr[c] = code8(o_rgb[c]); cn[c] = code8(o_n[c]); }
      const unsigned long long hi = zb[p] & 0xFFFFFFFF00000000ull;
      zb[p] = hi | (unsigned long long)(cr[0] | (cr[1] << 8) | (cr[2] << 16) | (cn[0] << 24));
      L.ex[p] = (unsigned short)(cn[1] | (cn[2] << 8));
    }
  } else if (coded) {
    // Multisampling: one fragment-shader invocation per pixel and triangle (the first of the triangle's samples stands
    // for it, weighted by the number of samples the triangle owns); the pixel's colour is the mean of the four samples'
    // 8-bit colours (uncovered samples: the clear colour 0), rounded half up -- integer arithmetic, no rounding ties.
    // The INVOCATIONS are compacted, not the pixels: a lane looping over its pixel's four samples ran the shader four
    // times per wave with half of the lanes idle (1.97 invocations per covered pixel); the shading is bound by its
    // latency chains, so the number of passes is what counts.  Sums: 3 x 10 bits in the low word of the pixel's centre key
    // (its triangle id is not needed any more) and in nsum, by LDS atomics -- integer, order-independent.
    const int lane = tid & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t* const zlo = reinterpret_cast<uint32_t*>(zb);  // little endian: word 2 k = low half of key k
    // three pixels per thread and round: ONE prefix sum / list reservation per wave for the lot (a round per pixel spent most of
    // its time in the returning LDS atomics and ballots: 9.5 k of 93 k cycles per workgroup)
    constexpr int kG = 3;
    for (int p0 = 0; p0 < npix; p0 += kG * kThreads) {
      unsigned inv_lo[kG], inv_hi[kG];  // bit sm: sample sm starts an invocation (of a triangle with at most / more than two anisotropic probes)
      int cnt = 0, cnt_h = 0;
#pragma unroll
      for (int g = 0; g < kG; ++g) {
        const int p = p0 + g * kThreads + tid;
        unsigned inv = 0, ihi = 0;
        if (p < npix) {
          uint32_t kf[4];
          bool kc[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) { const unsigned long long kt = zb[p * NS + t]; kc[t] = kt != kKeyEmpty; kf[t] = (uint32_t)kt; }
          const bool cov = (kc[0] | kc[1] | kc[2] | kc[3]) && want_colour;
          zb[p * NS + (NS - 1)] &= 0xFFFFFFFF00000000ull;  // colour sums 0, the depth bits stay
          L.nsum[p] = 0;
          if (cov) {
#pragma unroll
            for (int sm = 0; sm < 4; ++sm) {
              bool first = kc[sm];
#pragma unroll
              for (int t = 0; t < sm; ++t) first &= !(kc[t] && kf[t] == kf[sm]);
              inv |= first ? (1u << sm) : 0u;
              if (ANISO) ihi |= (first && (kf[sm] >> 30)) ? (1u << sm) : 0u;
            }
          }
        }
        inv_hi[g] = ihi; inv_lo[g] = inv & ~ihi;
        cnt += __popc(inv_lo[g]); cnt_h += __popc(ihi);
      }
      int pre = 0, tot = 0, pre_h = 0, tot_h = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {  // counts <= 12
        const unsigned long long m = __ballot((cnt >> b) & 1);
        pre += __popcll(m & lt_mask) << b;
        tot += __popcll(m) << b;
        if (ANISO) {
          const unsigned long long mh = __ballot((cnt_h >> b) & 1);
          pre_h += __popcll(mh & lt_mask) << b;
          tot_h += __popcll(mh) << b;
        }
      }
      int base = 0, base_h = 0;
      if (lane == 0 && tot != 0) base = atomicAdd(&n_cov, tot);
      if (ANISO && lane == 0 && tot_h != 0) base_h = atomicAdd(&n_cov_hi, tot_h);
      base = __shfl(base, 0) + pre;
      base_h = list_cap - 1 - (__shfl(base_h, 0) + pre_h);
#pragma unroll
      for (int g = 0; g < kG; ++g) {
        const int p = p0 + g * kThreads + tid;
#pragma unroll
        for (int sm = 0; sm < 4; ++sm) {
          if (inv_lo[g] & (1u << sm)) L.plist[base++] = (unsigned short)(p | (sm << 14));
          if (ANISO && (inv_hi[g] & (1u << sm))) L.plist[base_h--] = (unsigned short)(p | (sm << 14));
        }
      }
    }
    __syncthreads();
    HP_STAMP(3);  // compaction of the invocations
    const int ninv_lo = n_cov, ninv = ninv_lo + (ANISO ? n_cov_hi : 0);
#ifdef HP_RABL_NO_SHADE
    const int ninv_loop = a.w < 0 ? ninv : 0;
#else
    const int ninv_loop = ninv;
#endif
    for (int q = tid; q < ninv_loop; q += kThreads) {
      const unsigned e = L.plist[q < ninv_lo ? q : list_cap - 1 - (q - ninv_lo)];
      const int p = (int)(e & 0x3FFFu), sm = (int)(e >> 14);
      const int pr = (int)(((unsigned long long)p * a.w_magic) >> 32);
      const int i = row0 + pr, j = p - pr * a.w;
      const uint32_t f = zlo[2 * (p * NS + sm)];
      unsigned mult = 1;  // samples of this pixel the triangle owns
      for (int t = sm + 1; t < 4; ++t) {
        const unsigned long long kt = zb[p * NS + t];
        mult += (kt != kKeyEmpty && (uint32_t)kt == f) ? 1u : 0u;
      }
      float r3[3], n3[3];
      shade_centre<ANISO>(a, cx, (int)f, i, j, r3, n3);
      atomicAdd(&zlo[2 * (p * NS + (NS - 1))], mult * (code8(r3[0]) | (code8(r3[1]) << 10) | (code8(r3[2]) << 20)));
      if (cx.need_normal) atomicAdd(&L.nsum[p], mult * (code8(n3[0]) | (code8(n3[1]) << 10) | (code8(n3[2]) << 20)));
    }
  }
  __syncthreads();
  HP_STAMP(4);  // shading

  // ---- output pass: pixel-parallel; crop taps + the view's run(s) of the pixel record, or the strided planes ----
  const float zn = a.depth_norm_z ? a.depth_norm_z[item] : 1.0f;
  const int64_t cbase = (int64_t)item * a.cs.s_item + (int64_t)vi * a.cs.s_view;
  const int64_t dbase = (int64_t)item * a.ds.s_item + (int64_t)vi * a.ds.s_view;
  const bool half_out = HALF;  // fp16 network input written directly
  // crop source
  const float* img = nullptr;
  bool bad_id = false;
  int nr_max = 0, nc_max = 0, csrc0 = 0;
  bool separable = true;
  if (ncrop > 0) {
    const int im_id = a.im_ids[item];
    bad_id = (unsigned)im_id >= (unsigned)a.Bi;  // reads frame 0, writes zeros (the reference's indexing would raise)
    img = a.images + (int64_t)(bad_id ? 0 : im_id) * a.Ct * a.IH * a.IW;
    nr_max = span_max[0]; nc_max = span_max[1];
    separable = nr_max <= kSpan && nc_max <= kSpan;
    csrc0 = a.v_crop_src0[vi];
  }
  const int HW = a.IH * a.IW;
  const float count = (float)(a.sr * a.sr);
  for (int p = tid; p < npix; p += kThreads) {
    const int pr = (int)(((unsigned long long)p * a.w_magic) >> 32);
    const int i = row0 + pr, j = p - pr * a.w;
    float cropv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.rec) {
      // ---- crop channels of this pixel (roi_align of the frame; same taps in the same order as crop_tile_kernel) ----
#ifdef HP_RABL_NO_CROP
      if (ncrop > 0 && a.w < 0) {
#else
      if (ncrop > 0) {
#endif
        const Fold fy = L.fy[pr], fx = L.fx[j];
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float vacc = 0.0f;
        if (separable) {
          if (ncrop == 3) crop_taps<3>(img, HW, a.IW, a.IH, csrc0, fy, fx, nr_max, nc_max, acc, vacc);
          else if (ncrop == 4) crop_taps<4>(img, HW, a.IW, a.IH, csrc0, fy, fx, nr_max, nc_max, acc, vacc);
          else if (ncrop == 1) crop_taps<1>(img, HW, a.IW, a.IH, csrc0, fy, fx, nr_max, nc_max, acc, vacc);
          else crop_taps<2>(img, HW, a.IW, a.IH, csrc0, fy, fx, nr_max, nc_max, acc, vacc);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c < ncrop) {
            float val = acc[c] / count;
            if (csrc0 + c == 3) {
              if (vacc / count < 0.99f) val = 0.0f;  // TB/lib3d/cropping.py:184-195
              if (a.crop_depth_mode == 1) val = val / zn;
              else if (a.crop_depth_mode == 2) val = fminf(fmaxf(val / zn, 0.0f), 2.0f) - 1.0f;
              else if (a.crop_depth_mode == 3) val = fminf(fmaxf(val - zn, -2.0f), 2.0f);
            }
            cropv[c] = bad_id ? 0.0f : val;
          }
        }
      }
    }
    float o_rgb[3] = {0.f, 0.f, 0.f}, o_n[3] = {0.f, 0.f, 0.f}, o_d = 0.0f;
    if (!band_empty) {
      const unsigned long long slot = zb[p * NS + (NS - 1)];
      const uint32_t zbits = (uint32_t)(slot >> 32);
      if (zbits != 0xFFFFFFFFu) {
        const float Z = 1.0f / __uint_as_float(~zbits);  // the key holds ~bits(1 / z)
        o_d = Z > a.depth_max ? 0.0f : Z;
      }
      if (coded && NS == 1) {
        const uint32_t lo = (uint32_t)slot;
        const uint32_t ex = L.ex[p];
        o_rgb[0] = div255((float)(lo & 255u)); o_rgb[1] = div255((float)((lo >> 8) & 255u)); o_rgb[2] = div255((float)((lo >> 16) & 255u));  // = the IEEE quotient (div255)
        o_n[0] = div255((float)(lo >> 24)); o_n[1] = div255((float)(ex & 255u)); o_n[2] = div255((float)(ex >> 8));
      } else if (coded) {
        // the resolve of the 8-bit multisampled buffers: the mean of the four samples' codes, rounded half up (the sums: 3 x 10 bits)
        const uint32_t sr = (uint32_t)slot, sn = L.nsum[p];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          o_rgb[c] = div255((float)((((sr >> (10 * c)) & 1023u) + 2u) >> 2));
          o_n[c] = div255((float)((((sn >> (10 * c)) & 1023u) + 2u) >> 2));
        }
      }
    }
    float d_out = o_d;
    if (a.depth_norm_mode == 1) d_out = o_d / zn;
    else if (a.depth_norm_mode == 2) d_out = fminf(fmaxf(o_d / zn, 0.0f), 2.0f) - 1.0f;
    else if (a.depth_norm_mode == 3) d_out = fminf(fmaxf(o_d - zn, -2.0f), 2.0f);

    if (a.rec) {
      // ---- the record ----
      float rend[7];
      int nrend = 3;
      rend[0] = o_rgb[0]; rend[1] = o_rgb[1]; rend[2] = o_rgb[2];
      if (a.want_nrm) { rend[3] = o_n[0]; rend[4] = o_n[1]; rend[5] = o_n[2]; nrend = 6; }
      if (a.want_depth) {
        if (nrend == 6) rend[6] = d_out; else rend[3] = d_out;
        ++nrend;
      }
      const int64_t pix = (int64_t)item * a.rec_item + (int64_t)i * a.rec_row + (int64_t)j * a.rec_col;
      const int c0 = a.v_c0[vi], cc0 = a.v_crop_c0[vi];
#ifdef HP_RABL_NO_STORE
      if (a.w > 0 && cropv[0] + rend[0] != -123.0f) continue;
#endif
      if (HALF) {
        _Float16* const o = reinterpret_cast<_Float16*>(a.rec) + pix;
        if (a.rec_own_all) {  // one view owns the whole 16-half record: two 16-B stores, pads included
          typedef _Float16 halfx8v __attribute__((ext_vector_type(8)));
          // the reference's order: crop channels [0, ncrop), render channels behind them, zeros to the end of the record
          _Float16 hv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {  // compile-time k: no indexed register array (a stack would break hipGraph replay)
            const float r3 = (k >= 3 && k - 3 < 7 && k - 3 < nrend) ? rend[k - 3 < 7 ? (k >= 3 ? k - 3 : 0) : 0] : 0.f;
            const float r4 = (k >= 4 && k - 4 < 7 && k - 4 < nrend) ? rend[k - 4 < 7 ? (k >= 4 ? k - 4 : 0) : 0] : 0.f;
            const float cv = k < 4 ? cropv[k < 4 ? k : 0] : 0.f;
            hv[k] = (_Float16)(k < ncrop ? cv : (ncrop == 3 ? r3 : r4));
          }
          *reinterpret_cast<halfx8v*>(o) = halfx8v{hv[0], hv[1], hv[2], hv[3], hv[4], hv[5], hv[6], hv[7]};
          *reinterpret_cast<halfx8v*>(o + 8) = halfx8v{hv[8], hv[9], hv[10], hv[11], hv[12], hv[13], hv[14], hv[15]};
        } else {
          if (separable) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < ncrop) o[cc0 + c] = (_Float16)cropv[c];
          }
#pragma unroll
          for (int c = 0; c < 7; ++c)
            if (c < nrend) o[c0 + c] = (_Float16)rend[c];
        }
      } else {
        float* const o = reinterpret_cast<float*>(a.rec) + pix;
        if (ncrop > 0 && separable) store_run(o + cc0, cropv, ncrop);
        store_run(o + c0, rend, nrend);
      }
      if (ncrop > 0 && !separable) {
        // strongly down-sampling crop (a bin spans more than kSpan source pixels: boxes wider than ~850 px): the literal
        // sampling_ratio^2-sample evaluation, one channel at a time in a rolled loop (rare; keeps it out of the
        // register budget of the common path), stored after the record so that it lands on top of the zeros above
#pragma unroll 1
        for (int c = 0; c < ncrop; ++c) {
          const int sc = csrc0 + c;
          float sacc = 0.0f, svv = 0.0f;
          slow_pixel(img + (int64_t)sc * HW, a.IH, a.IW, cy1, cx1, i, j, bin_h, bin_w, a.sr, sc == 3, sacc, svv);
          float val = sacc / count;
          if (sc == 3) {
            if (svv / count < 0.99f) val = 0.0f;  // TB/lib3d/cropping.py:184-195
            if (a.crop_depth_mode == 1) val = val / zn;
            else if (a.crop_depth_mode == 2) val = fminf(fmaxf(val / zn, 0.0f), 2.0f) - 1.0f;
            else if (a.crop_depth_mode == 3) val = fminf(fmaxf(val - zn, -2.0f), 2.0f);
          }
          if (bad_id) val = 0.0f;
          if (HALF) reinterpret_cast<_Float16*>(a.rec)[pix + cc0 + c] = (_Float16)val;
          else reinterpret_cast<float*>(a.rec)[pix + cc0 + c] = val;
        }
      }
      continue;
    }

    // ---- strided planes (hp_rasterize: NCHW tensors or any strides) ----
    const int64_t co = cbase + (int64_t)i * a.cs.s_row + (int64_t)j * a.cs.s_col;
    typedef float float3v __attribute__((ext_vector_type(3)));
    if (a.rgb) {
      if (half_out) {
        _Float16* const o = reinterpret_cast<_Float16*>(a.rgb);
#pragma unroll
        for (int c = 0; c < 3; ++c) o[co + c * a.cs.s_chan] = (_Float16)o_rgb[c];
      } else if (a.cs.s_chan == 1) *reinterpret_cast<float3v*>(a.rgb + co) = float3v{o_rgb[0], o_rgb[1], o_rgb[2]};
      else {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.rgb[co + c * a.cs.s_chan] = o_rgb[c];
      }
    }
    if (a.nrm) {
      if (half_out) {
        _Float16* const o = reinterpret_cast<_Float16*>(a.nrm);
#pragma unroll
        for (int c = 0; c < 3; ++c) o[co + c * a.cs.s_chan] = (_Float16)o_n[c];
      } else if (a.cs.s_chan == 1) *reinterpret_cast<float3v*>(a.nrm + co) = float3v{o_n[0], o_n[1], o_n[2]};
      else {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.nrm[co + c * a.cs.s_chan] = o_n[c];
      }
    }
    if (a.mask) a.mask[((int64_t)view * a.h + i) * a.w + j] = o_d > 0.0f ? 1 : 0;
    if (a.depth) {
      const int64_t dof = dbase + (int64_t)i * a.ds.s_row + (int64_t)j * a.ds.s_col;
      if (half_out) reinterpret_cast<_Float16*>(a.depth)[dof] = (_Float16)d_out;
      else a.depth[dof] = d_out;
    }
  }
  HP_STAMP(6);  // output pass issued
}

}  // namespace hp

// Rasteriser scratch of a mesh store for `n` views of `n_bands` bands: per-(view, band) lists of sub-triangle ids, their
// counters and the per-(view, sub-triangle) set-up records.  Grows, never shrinks; refuses to grow under stream capture (a
// captured launch would keep the pointer that is freed here) -- run the call once eagerly, or reserve.
static int raster_scratch(hp::MeshStore* ms, int n, int n_bands, hipStream_t st, int* chunk_out) {
  const size_t rec_view = (size_t)2 * (size_t)ms->max_faces * (128 + 32);  // records (up to 128 B) + the wide corners (32 B) per slot
  const size_t xv_view = (size_t)ms->max_verts * sizeof(int4);
  const size_t list_view = (size_t)n_bands * 2 * (size_t)ms->max_faces * sizeof(int32_t);  // bin_cap = rec_slots ids per band (launch_raster)
  const size_t per_view = list_view + rec_view + xv_view;
  // Scratch budget: min(8 GB, 1/16 of the device's free memory at the first call); what is allocated is what the largest call
  // needs (a multisampled 240 x 320 view of a 16 k-face object: 15 MB of list address space + 5 MB of records).  A call that
  // exceeds it renders in chunks of views.  HP_RASTER_LIST_BUDGET_MB overrides; HP_RASTER_CHUNK_VIEWS=<n> forces chunks (the
  // tests' way to run the chunked path), HP_RASTER_CHUNK_SYNC=1 synchronises the stream after every chunk (diagnostics).
  static const size_t budget = [] {
    if (hp::dbg(hp::DBG_RASTER_LIST_BUDGET_MB) > 0) return (size_t)hp::dbg(hp::DBG_RASTER_LIST_BUDGET_MB) << 20;
    size_t free_b = 0, total_b = 0;
    size_t b = (size_t)8192 << 20;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 16 < b) b = free_b / 16;
    return b < ((size_t)64 << 20) ? ((size_t)64 << 20) : b;
  }();
  int chunk = (int)std::min<size_t>(budget / (per_view ? per_view : 1), 1u << 30);
  const int chunk_env = hp::dbg(hp::DBG_RASTER_CHUNK_VIEWS);
  if (chunk_env > 0 && chunk_env < chunk) chunk = chunk_env;
  if (chunk < 1) chunk = 1;
  if (chunk > n) chunk = n;
  *chunk_out = chunk;
  const size_t need_list = (size_t)chunk * list_view, need_cnt = (size_t)chunk * n_bands * 2 * sizeof(int32_t);
  const size_t need_rec = (size_t)chunk * rec_view, need_xv = (size_t)chunk * xv_view;
  if (ms->bin_list_bytes >= need_list && ms->bin_count_bytes >= need_cnt && ms->recs_bytes >= need_rec && ms->xverts_bytes >= need_xv) return HP_OK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (st) (void)hipStreamIsCapturing(st, &cap);
  HP_REQUIRE(cap == hipStreamCaptureStatusNone,
             "hp_rasterize: the rasteriser scratch would have to grow during stream capture (hp_mesh_store_reserve_raster first)");
  auto grow = [](void** p, size_t* have, size_t need, bool zero) -> int {
    if (*have >= need) return HP_OK;
    if (*p) (void)hipFree(*p);  // hipFree waits for the device: nothing in flight reads it any more
    *p = nullptr; *have = 0;
    HP_CHECK_HIP(hipMalloc(p, need));
    if (zero) HP_CHECK_HIP(hipMemset(*p, 0, need));  // (synchronous: done before any launch that follows)
    *have = need;
    return HP_OK;
  };
  int rc = grow((void**)&ms->bin_list, &ms->bin_list_bytes, need_list, false);
  if (!rc) rc = grow((void**)&ms->bin_count, &ms->bin_count_bytes, need_cnt, true);  // the kernels keep the counters at zero between launches
  if (!rc) rc = grow((void**)&ms->recs, &ms->recs_bytes, need_rec, false);
  if (!rc) rc = grow((void**)&ms->xverts, &ms->xverts_bytes, need_xv, false);
  ms->scratch_generation += 1;
  if (hp::dbg(hp::DBG_RASTER_CANARY))
    std::fprintf(stderr, "[hp raster scratch] store %p: bin_list %p (%zu B) bin_count %p recs %p (%zu B) xverts %p (%zu B)\n", (void*)ms, (void*)ms->bin_list,
                 ms->bin_list_bytes, (void*)ms->bin_count, (void*)ms->recs, ms->recs_bytes, (void*)ms->xverts, ms->xverts_bytes);
  return rc;
}

namespace hp {
// band layout of a render: rows per band / threads per workgroup / LDS keys
struct BandPlan { int band_rows, n_bands, npix_max; bool wide; };
static bool band_plan(int h, int w, int msaa, int rows_env, BandPlan* bp) {
  bp->wide = false;
  int budget = msaa ? kBandKeysMsaa / kSamplesMsaa : kBandPixels;  // pixels of a band
  if (msaa && w > budget) { budget = kBandKeysMsaaWide / kSamplesMsaa; bp->wide = true; }  // 641 .. 1280 px: the 512-thread instantiation
  if (w > budget) return false;
  bp->band_rows = rows_env > 0 && !msaa && rows_env * w <= 2 * kBandPixels ? rows_env : budget / w;
  if (bp->band_rows > h) bp->band_rows = h;
  bp->n_bands = (h + bp->band_rows - 1) / bp->band_rows;
  bp->npix_max = bp->band_rows * w;
  return true;
}
}  // namespace hp

extern "C" int hp_mesh_store_reserve_raster(hp_mesh_store* store, int n_views, int h, int w, int flags) {
  using namespace hp;
  HP_REQUIRE(store != nullptr, "hp_mesh_store_reserve_raster: null mesh store");
  HP_REQUIRE(n_views >= 0 && h > 0 && w > 0 && w <= kBandPixels, "hp_mesh_store_reserve_raster: bad size");
  if (n_views == 0) return HP_OK;
  int chunk = 0;
  BandPlan bp;
  HP_REQUIRE(band_plan(h, w, 0, 0, &bp), "hp_mesh_store_reserve_raster: image too wide for one band");
  int rc = raster_scratch(store, n_views, bp.n_bands, nullptr, &chunk);
  if (!rc && (flags & HP_RASTER_MSAA4) && band_plan(h, w, 1, 0, &bp)) rc = raster_scratch(store, n_views, bp.n_bands, nullptr, &chunk);
  return rc;
}

extern "C" int64_t hp_mesh_store_scratch_generation(const hp_mesh_store* store) { return store ? store->scratch_generation : -1; }

namespace hp {
static const hp_raster_conventions kDefaultConventions = {{0.375f, 0.875f, 0.125f, 0.625f}, {0.125f, 0.375f, 0.625f, 0.875f}, 16, 0, 0, 0.0f, 0.0f,
                                                          {0, 1, 2}, {1.0f, -1.0f, -1.0f}};
// state of a new store: the default record, culling on (HP_RASTER_NO_CULL=1: off)
void raster_store_defaults(MeshStore* s) {
  s->conventions = kDefaultConventions;
  s->backface_culling = hp::dbg(hp::DBG_RASTER_NO_CULL) ? 0 : 1;
}

static RasterConv derive_conventions(const hp_raster_conventions& c, int msaa) {
  RasterConv r{};
  auto sub = [](float s) { const int v = (int)std::rint(s * (float)kSub); return v < 0 ? 0 : v > kSub - 1 ? kSub - 1 : v; };  // hardware keeps sample positions on such a grid
  for (int k = 0; k < 4; ++k) { r.sxi[k] = sub(c.msaa_x[k]); r.syi[k] = sub(c.msaa_y[k]); }
  r.sxi[4] = r.syi[4] = kSub / 2;
  r.lo_x = r.hi_x = r.lo_y = r.hi_y = kSub / 2;  // the centre is always tested
  for (int k = 0; k < 5; ++k) {
    r.sxf[k] = (float)r.sxi[k] * (1.0f / (float)kSub); r.syf[k] = (float)r.syi[k] * (1.0f / (float)kSub);
    if (msaa && k < 4) {
      r.lo_x = std::min(r.lo_x, r.sxi[k]); r.hi_x = std::max(r.hi_x, r.sxi[k]);
      r.lo_y = std::min(r.lo_y, r.syi[k]); r.hi_y = std::max(r.hi_y, r.syi[k]);
    }
  }
  r.aniso_max = (float)c.aniso_max; r.lod_bias = c.lod_bias; r.ratio_bias = c.aniso_ratio_bias; r.aniso_round = c.aniso_round; r.lod_from = c.lod_from;
  for (int k = 0; k < 3; ++k) { r.n_axis[k] = c.normal_axis[k]; r.n_sign[k] = c.normal_sign[k]; }
  return r;
}

// Common launch path of hp_rasterize (strided planes) and hp_render_inputs (network-input records + fused crop).
static int launch_raster(const hp_mesh_store* store, RasterArgs a, int n, bool crop, hipStream_t st) {
  const int h = a.h, w = a.w;
  a.msaa = (a.flags & HP_RASTER_MSAA4) && (a.rec || a.rgb || a.nrm) ? 1 : 0;  // depth-only renders have nothing to multisample
  a.cv = derive_conventions(store->conventions, a.msaa);
  a.cull = store->backface_culling ? store->cull : nullptr;
  const int ns = a.msaa ? kSamplesMsaa : 1;
  const int rows_env = 0;  // band height: band_plan's choice
  BandPlan bp;
  HP_REQUIRE(band_plan(h, w, a.msaa, rows_env, &bp), "hp_rasterize: image too wide for one band");
  a.band_rows = bp.band_rows; a.n_bands = bp.n_bands;
  HP_REQUIRE(a.n_bands <= kMaxBands, "hp_rasterize: unsupported resolution (too many bands)");
  const int npix_max = bp.npix_max;
  HP_REQUIRE(npix_max < 65536, "hp_rasterize: band too large");
  HP_REQUIRE(!a.msaa || npix_max < 16384, "hp_rasterize: multisampled band too large");  // invocation entries: pixel | sample << 14
  a.w_magic = (unsigned)(0x100000000ull / (unsigned)w + 1);
  a.depth_max = kZNear / (1.0f - (1.0f - 1e-3f) * (kZFar - kZNear) / kZFar);
  // the shading needs barycentrics (sector 2 of the records): normals, point lights, or an object without a texture
  a.need_attr = (a.rec ? a.want_nrm != 0 : a.nrm != nullptr) || a.n_lights > 0 || store->any_untextured;
  // Views are processed in chunks so that the scratch stays within a fixed budget.  The scratch is owned by the store and only
  // ever grows (hp_mesh_store_reserve_raster pre-sizes it; predictors do that at construction for their largest batch); a
  // growth bumps the store's scratch generation, which tells holders of captured hipGraphs that the pointers their launches
  // carry are gone.  Callers sharing one store must be stream-ordered.
  hp::MeshStore* ms = const_cast<hp_mesh_store*>(store);
  a.max_faces = (int)store->max_faces;
  a.rec_slots = 2 * a.max_faces;
  a.bin_cap = a.rec_slots;  // every record id (f, or nf + f: the second half of a near-clipped quad) at most once per band
  int chunk = 0;
  {
    const int rc = raster_scratch(ms, n, a.n_bands, st, &chunk);
    if (rc) return rc;
  }
  if (hp::dbg(hp::DBG_RASTER_CANARY)) {  // diagnostics: whatever this call reads without having written it is NaN
    HP_CHECK_HIP(hipMemsetAsync(ms->recs, 0xFF, ms->recs_bytes, st));
    HP_CHECK_HIP(hipMemsetAsync(ms->xverts, 0xFF, ms->xverts_bytes, st));
  }
  a.bin_list = ms->bin_list;
  a.bin_count = ms->bin_count;
  a.recs = ms->recs;
  a.rec_q = a.need_attr ? 8 : 4;
  a.recs_wide = ms->recs + (size_t)chunk * a.rec_slots * 8;  // behind the chunk's (up to 128-B) records
  a.xverts = ms->xverts;
  a.max_verts = (int)store->max_verts;
  const size_t lds = band_lds_bytes(npix_max, ns, a.band_rows, w, crop);
  HP_REQUIRE(lds <= 150 * 1024, "hp_rasterize: band does not fit the LDS");
  const bool half = (a.flags & HP_RASTER_OUT_F16) != 0, aniso = (a.flags & HP_RASTER_TEX_ANISO) != 0;
  typedef void (*BandKernel)(RasterArgs, int);
  static const BandKernel kernels[12] = {
      raster_kernel<1, false, false, false>, raster_kernel<1, false, true, false>, raster_kernel<1, true, false, false>, raster_kernel<1, true, true, false>,
      raster_kernel<kSamplesMsaa, false, false, false>, raster_kernel<kSamplesMsaa, false, true, false>, raster_kernel<kSamplesMsaa, true, false, false>,
      raster_kernel<kSamplesMsaa, true, true, false>,
      raster_kernel<kSamplesMsaa, false, false, true>, raster_kernel<kSamplesMsaa, false, true, true>, raster_kernel<kSamplesMsaa, true, false, true>,
      raster_kernel<kSamplesMsaa, true, true, true>};
  const int ki = (bp.wide ? 8 : 4 * a.msaa) + 2 * (half ? 1 : 0) + (aniso ? 1 : 0);
  static size_t opted[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (opted[ki] < lds) {
    HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernels[ki]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    opted[ki] = lds;
  }
  for (int v0 = 0; v0 < n; v0 += chunk) {
    const int nv = n - v0 < chunk ? n - v0 : chunk;
    a.view0 = v0;
    a.n = nv;
    hipLaunchKernelGGL(raster_xform_kernel, dim3((a.max_verts + 255) / 256, nv), dim3(256), 0, st, a);
    hipLaunchKernelGGL(raster_setup_kernel, dim3((a.max_faces + kBinThreads - 1) / kBinThreads, nv), dim3(kBinThreads), 0, st, a);
    const int total = nv * a.n_bands;
    hipLaunchKernelGGL(kernels[ki], dim3(8 * ((total + 7) / 8)), dim3(band_threads(ns, bp.wide)), lds, st, a, npix_max);
    const bool chunk_sync = hp::dbg(hp::DBG_RASTER_CHUNK_SYNC) != 0;  // diagnostics (see raster_scratch)
    if (chunk_sync && v0 + chunk < n) (void)hipStreamSynchronize(st);
  }
  return check_launch("raster_kernel");
}
}  // namespace hp

extern "C" int hp_mesh_store_set_raster_conventions(hp_mesh_store* store, const hp_raster_conventions* c) {
  using namespace hp;
  HP_REQUIRE(store != nullptr, "hp_mesh_store_set_raster_conventions: null mesh store");
  hp_raster_conventions v = c ? *c : kDefaultConventions;
  for (int k = 0; k < 4; ++k)
    HP_REQUIRE(v.msaa_x[k] > 0.0f && v.msaa_x[k] < 1.0f && v.msaa_y[k] > 0.0f && v.msaa_y[k] < 1.0f,
               "hp_mesh_store_set_raster_conventions: sample positions must lie inside the pixel, in (0, 1)");
  HP_REQUIRE(v.aniso_max >= 1 && v.aniso_max <= 16, "hp_mesh_store_set_raster_conventions: aniso_max must be in 1..16");
  HP_REQUIRE(v.aniso_round >= 0 && v.aniso_round <= 2 && v.lod_from >= 0 && v.lod_from <= 2, "hp_mesh_store_set_raster_conventions: unknown rule");
  HP_REQUIRE(std::isfinite(v.lod_bias) && std::isfinite(v.aniso_ratio_bias), "hp_mesh_store_set_raster_conventions: biases must be finite");
  for (int k = 0; k < 3; ++k)
    HP_REQUIRE(v.normal_axis[k] >= 0 && v.normal_axis[k] <= 2 && (v.normal_sign[k] == 1.0f || v.normal_sign[k] == -1.0f),
               "hp_mesh_store_set_raster_conventions: normal_axis in 0..2, normal_sign +1 / -1");
  store->conventions = v;
  return HP_OK;
}

extern "C" int hp_mesh_store_get_raster_conventions(const hp_mesh_store* store, hp_raster_conventions* out) {
  using namespace hp;
  HP_REQUIRE(store != nullptr && out != nullptr, "hp_mesh_store_get_raster_conventions: null argument");
  *out = store->conventions;
  return HP_OK;
}

extern "C" int hp_mesh_store_set_backface_culling(hp_mesh_store* store, int on) {
  if (!store) return -1;
  const int prev = store->backface_culling ? 1 : 0;
  store->backface_culling = on ? 1 : 0;
  return prev;
}

extern "C" int hp_mesh_store_get_backface_culling(const hp_mesh_store* store) { return store ? (store->backface_culling ? 1 : 0) : -1; }

extern "C" int hp_rasterize(const hp_mesh_store* store, int n, int views_per_item,
                            const int32_t* d_obj_ids, const float* d_TCO, const float* d_K,
                            const float* d_ambient, int n_lights, const float* d_light_pos,
                            const float* d_light_col, int h, int w, int flags, float* d_rgb,
                            float* d_nrm, const hp_strides* color_strides, float* d_depth,
                            const hp_strides* depth_strides, uint8_t* d_mask,
                            const float* d_depth_norm_z, int depth_norm_mode, void* stream) {
  using namespace hp;
  HP_REQUIRE(store != nullptr, "hp_rasterize: null mesh store");
  HP_REQUIRE(n >= 0 && views_per_item >= 1 && n % views_per_item == 0,
             "hp_rasterize: n must be a multiple of views_per_item");
  HP_REQUIRE(h > 0 && w > 0, "hp_rasterize: unsupported resolution");
  HP_REQUIRE(!d_mask || d_depth, "Binary mask can only be rendered if depth is rendered");
  if (n == 0) return HP_OK;
  HP_REQUIRE(d_TCO && d_K && d_obj_ids, "hp_rasterize: null pose/intrinsics/object ids");
  HP_REQUIRE(!(d_rgb || d_nrm) || color_strides, "hp_rasterize: colour strides missing");
  HP_REQUIRE(!d_depth || depth_strides, "hp_rasterize: depth strides missing");
  HP_REQUIRE(n_lights == 0 || (d_light_pos && d_light_col), "hp_rasterize: lights missing");
  HP_REQUIRE(depth_norm_mode >= 0 && depth_norm_mode <= 3, "hp_rasterize: bad depth_norm_mode");
  HP_REQUIRE(depth_norm_mode == 0 || d_depth_norm_z, "hp_rasterize: depth_norm_z missing");
  RasterArgs a{};
  a.verts4 = store->verts4; a.normals4 = store->normals4; a.uvs = store->uvs; a.colors = store->colors;
  a.faces4 = store->faces4; a.tex = store->tex; a.tex_quads = store->tex_quads; a.tex_quads_off = store->tex_quads_off; a.obj = store->obj;
  a.obj_ids = d_obj_ids; a.TCO = d_TCO; a.K = d_K; a.ambient = d_ambient;
  a.light_pos = d_light_pos; a.light_col = d_light_col; a.depth_norm_z = d_depth_norm_z;
  a.rgb = d_rgb; a.nrm = d_nrm; a.depth = d_depth; a.mask = d_mask;
  if (color_strides) a.cs = *color_strides;
  if (depth_strides) a.ds = *depth_strides;
  a.n = n; a.views_per_item = views_per_item; a.n_lights = n_lights; a.h = h; a.w = w;
  a.flags = flags; a.depth_norm_mode = depth_norm_mode;
  return launch_raster(store, a, n, false, (hipStream_t)stream);
}

extern "C" int hp_render_inputs(const hp_mesh_store* store, int n_items, int views_per_item, const int32_t* d_obj_ids,
                                const float* d_TCV_O, const float* d_KV, const float* d_ambient, int n_lights,
                                const float* d_light_pos, const float* d_light_col, int h, int w, int flags,
                                const float* d_images, int Bi, int Ct, int H, int W, const float* d_boxes,
                                const int32_t* d_im_ids, int sampling_ratio, const float* d_depth_norm_z, int depth_norm_mode,
                                void* d_x, int record_elems, const hp_input_layout* layout, void* stream) {
  using namespace hp;
  HP_REQUIRE(store != nullptr, "hp_render_inputs: null mesh store");
  HP_REQUIRE(n_items >= 0 && views_per_item >= 1 && views_per_item <= kMaxViews, "hp_render_inputs: 1..8 views per item");
  HP_REQUIRE(h > 0 && w > 0 && record_elems > 0 && layout && d_x, "hp_render_inputs: bad sizes / null pointer");
  if (n_items == 0) return HP_OK;
  HP_REQUIRE(d_TCV_O && d_KV && d_obj_ids, "hp_render_inputs: null pose/intrinsics/object ids");
  HP_REQUIRE(n_lights == 0 || (d_light_pos && d_light_col), "hp_render_inputs: lights missing");
  HP_REQUIRE(depth_norm_mode >= 0 && depth_norm_mode <= 3, "hp_render_inputs: bad depth_norm_mode");
  HP_REQUIRE(depth_norm_mode == 0 || d_depth_norm_z, "hp_render_inputs: depth_norm_z missing");
  const int want_nrm = (flags & HP_RENDER_NORMALS) != 0, want_depth = (flags & HP_RENDER_DEPTH) != 0;
  const int n_rc = 3 + 3 * want_nrm + want_depth;
  bool crop = false;
  for (int v = 0; v < views_per_item; ++v) {
    HP_REQUIRE(layout->view_c0[v] >= 0 && layout->view_c0[v] + n_rc <= record_elems, "hp_render_inputs: a view's render channels leave the record");
    const int nc = layout->crop_n[v];
    HP_REQUIRE(nc >= 0 && nc <= 4, "hp_render_inputs: 0..4 crop channels per view");
    if (nc > 0) {
      crop = true;
      HP_REQUIRE(layout->crop_src0[v] >= 0 && layout->crop_src0[v] + nc <= Ct && layout->crop_c0[v] >= 0 &&
                 layout->crop_c0[v] + nc <= record_elems, "hp_render_inputs: crop channels leave the frame / the record");
    }
  }
  if (crop) {
    HP_REQUIRE(d_images && d_boxes && d_im_ids, "hp_render_inputs: crop source missing");
    HP_REQUIRE((Ct == 3 || Ct == 4) && H > 0 && W > 0 && Bi > 0, "hp_render_inputs: frames must be [Bi][3|4][H][W]");
    HP_REQUIRE(sampling_ratio >= 1 && sampling_ratio <= kMaxSR, "hp_render_inputs: sampling_ratio must be 1..4");
  }
  RasterArgs a{};
  a.verts4 = store->verts4; a.normals4 = store->normals4; a.uvs = store->uvs; a.colors = store->colors;
  a.faces4 = store->faces4; a.tex = store->tex; a.tex_quads = store->tex_quads; a.tex_quads_off = store->tex_quads_off; a.obj = store->obj;
  a.obj_ids = d_obj_ids; a.TCO = d_TCV_O; a.K = d_KV; a.ambient = d_ambient;
  a.light_pos = d_light_pos; a.light_col = d_light_col; a.depth_norm_z = d_depth_norm_z;
  a.n = n_items * views_per_item; a.views_per_item = views_per_item; a.n_lights = n_lights; a.h = h; a.w = w;
  a.flags = flags & ~(HP_RENDER_NORMALS | HP_RENDER_DEPTH);
  a.depth_norm_mode = want_depth ? depth_norm_mode : 0;
  a.rec = d_x;
  a.rec_half = (flags & HP_RASTER_OUT_F16) != 0;
  a.rec_col = record_elems; a.rec_row = (int64_t)record_elems * w; a.rec_item = (int64_t)record_elems * w * h;
  a.rec_own_all = a.rec_half && views_per_item == 1 && record_elems == 16 && (reinterpret_cast<uintptr_t>(d_x) & 15) == 0 &&
                  layout->crop_c0[0] == 0 && layout->crop_src0[0] == 0 && (layout->crop_n[0] == 3 || layout->crop_n[0] == 4) &&
                  layout->view_c0[0] == layout->crop_n[0];
  a.want_nrm = want_nrm; a.want_depth = want_depth;
  for (int v = 0; v < kMaxViews; ++v) {
    const bool on = v < views_per_item;
    a.v_c0[v] = on ? layout->view_c0[v] : 0;
    a.v_crop_c0[v] = on ? layout->crop_c0[v] : 0;
    a.v_crop_src0[v] = on ? layout->crop_src0[v] : 0;
    a.v_crop_n[v] = on ? layout->crop_n[v] : 0;
  }
  a.images = crop ? d_images : nullptr; a.Bi = Bi; a.Ct = Ct; a.IH = H; a.IW = W; a.sr = sampling_ratio;
  a.crop_depth_mode = depth_norm_mode;
  a.boxes = d_boxes; a.im_ids = d_im_ids;
  return launch_raster(store, a, a.n, crop, (hipStream_t)stream);
}

#ifdef HP_RASTER_STAMPS
extern "C" int hp_debug_raster_stamps(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hp::hp_rstamp), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(hp::hp_rstamp), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
