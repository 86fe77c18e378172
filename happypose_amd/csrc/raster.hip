// HIP rasteriser for gfx950: renders one textured mesh per view for thousands of views per
// launch, straight into the (strided) tensors the conv stem reads.
//
// Replaces Panda3dBatchRenderer.render -> worker processes -> Panda3D/OpenGL
// (TB/renderer/panda3d_batch_renderer.py:194-286, TB/renderer/panda3d_scene_renderer.py:320-390).
// The arithmetic (operation order included) follows the CPU definition in
// oracle/csrc/oracle.c so that coverage decisions agree pixel for pixel.
//
// Mapping to the hardware
//   * one workgroup (256 threads = 4 waves) per (view, horizontal band of the image); the
//     band's 64-bit z-buffer {depth bits : triangle id} lives in LDS (<= 76.8 KB, two
//     workgroups per CU) and is updated with ds_min_u64 -- no z-buffer traffic to HBM;
//   * a thread owns a triangle: it transforms the three vertices itself (9+6 FMAs each, the
//     mesh is L2/MALL resident and shared by every view of the object), rejects it against
//     the band, sets up the three homogeneous edge functions and walks its (tiny: ~3 px)
//     bounding box; triangles with a large footprint are queued in LDS and walked by the
//     whole workgroup so one lane never serialises thousands of pixels;
//   * the resolve pass is pixel-parallel: consecutive lanes take consecutive columns, so
//     NCHW planes are written with fully coalesced 256-B wave stores; texture, uv and
//     normal gathers hit L2;
//   * HBM traffic is therefore the output tensor (+ first touch of mesh/texture) -- the
//     kernel's roofline is HBM bandwidth ("rasteriser").
//   * workgroups are renumbered so that each XCD gets a contiguous range of views:
//     consecutive hypotheses share an object, hence the mesh stays in that XCD's L2.
//
// Round 3: ONE WRITER PER PIXEL RECORD.  The kernel can also produce the observed crop (roi_align of the frame, crop_math.h)
// for its pixels and stores the crop channels and the render channels of a view as one run of the network-input record
// (hp_render_inputs): the separate crop launch and the 12-B-of-24/32-B partial-sector stores of two kernels (a
// read-modify-write in the memory system: rocprofv3 counted 3x the algorithmic bytes, profiles/r03a_raster_hbm_traffic.json)
// are gone.  Inside a workgroup the work is re-organised in four passes: coverage (triangle-parallel, LDS z-buffer) ->
// compaction of the covered pixels -> shading of the compacted list (every lane has a fragment; the 8-bit colour codes go
// back into the z-buffer slots) -> pixel-parallel output pass (crop taps + record assembly, coalesced stores).
#include <atomic>
#include <cmath>
#include <mutex>

#include "common.h"
#include "crop_math.h"

// Band size: a band is one workgroup.  Only the bands the object covers carry work, so the launch is as long as the busiest
// CU's sequence of busy bands.  LDS per workgroup: 8 B (z key / colour codes) + 2 B (codes) + 2 B (covered-pixel list) per
// pixel + the crop folds: 10 rows of 320 pixels = 49 KB -> three 512-thread workgroups per CU, which is also what the
// fp32 kernel's 80 VGPRs allow (24 waves per CU).
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
constexpr unsigned long long kKeyEmpty = 0xFFFFFFFFFFFFFFFFull;
constexpr int kBandPixels = HP_RASTER_BAND_PIXELS;  // LDS z-buffer of a band, 8 B per pixel
// HP_RASTER_MSAA4 (the reference's framebuffer state, see oracle.c HP_R_MSAA4): five keys per pixel -- the four colour
// samples of the standard 4x pattern and the pixel centre (depth / mask stay centre-sampled) -- in a 25-KB z-buffer:
// 2 rows of 320 pixels per band, four 256-thread workgroups per CU (band_threads below; rounds 2-3: 4 rows, 512 threads).
// The coverage pass tests 5 samples on a bounding box that grows by the sample spread: ~10x its single-sample work,
// 3.6x the whole rasteriser (128 views: 240 -> 880 us)
constexpr int kSamplesMsaa = 5;
#ifndef HP_RASTER_BAND_KEYS_MSAA
#define HP_RASTER_BAND_KEYS_MSAA 3200
#endif
constexpr int kMaxViews = 8;  // views per item a record layout can describe
constexpr int kBandKeysMsaa = HP_RASTER_BAND_KEYS_MSAA;
// The renderer conventions nobody can pin without Panda3D (hp_raster_conventions in the header), as the kernels see them:
// the record itself plus what the host derives from it once per launch.  Kernel arguments (SGPRs): with the default record
// every expression below evaluates the operations of the former compile-time constants on the same values -- bit-identical.
struct RasterConv {
  float sx[5], sy[5];          // the four colour samples + the pixel centre (slot 4)
  float lo_x, hi_x, lo_y, hi_y;  // smallest / largest sample offset per axis (bounding-box growth)
  float dxa[4], dya[4];        // |offset of sample s from the pixel centre| (conservative reject of the coverage walk)
  float aniso_max, lod_bias, ratio_bias;
  int aniso_round, lod_from;
  int n_axis[3]; float n_sign[3];
};
__device__ __forceinline__ float sample_x(const RasterConv& cv, int ns, int sm) { return ns == 1 ? 0.5f : cv.sx[sm]; }
__device__ __forceinline__ float sample_y(const RasterConv& cv, int ns, int sm) { return ns == 1 ? 0.5f : cv.sy[sm]; }
// probe count and level of detail of the anisotropic filter from the footprint (pmax >= pmin, texel units)
__device__ __forceinline__ void aniso_footprint(const RasterConv& cv, float pmax, float pmin, int nlev, float& nf, float& lod) {
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
constexpr int kBigArea = 128;  // bbox pixels above which a triangle is walked cooperatively
// The multisampled band kernel (late round 4): 256 threads on 2-row bands (3200 keys, 25 KB of z-buffer) instead of 512 threads on
// 4 rows -- FOUR workgroups per CU instead of two at the same 16 waves.  A band is a chain of dependent gathers (list ->
// corners -> vertices -> z-buffer -> attributes -> texels) on a few hundred triangles -- one pass of its threads either way --
// so the chains in flight per CU are what counts: 608 -> 563 us per 128 C2 views (C3 unchanged; 128 threads x 1 row 678,
// 256 x 1 row 740, 192 x 2 rows 622, 512 x 3 rows 675).  -DHP_RASTER_THREADS_MSAA=512 -DHP_RASTER_BAND_KEYS_MSAA=6400: rounds 2-3.
#ifndef HP_RASTER_THREADS_MSAA
#define HP_RASTER_THREADS_MSAA 256
#endif
constexpr __host__ __device__ int band_threads(int ns) { return ns == 1 ? HP_RASTER_THREADS : HP_RASTER_THREADS_MSAA; }
constexpr int kBinThreads = 1024;  // binning kernel
constexpr int kMaxBands = 512;  // 480 one-row multisampled bands of a 640-wide render

struct RasterArgs {
  const float4* verts4;   // xyz + pad
  const float4* normals4;
  const float* uvs;
  const uint8_t* colors;
  const int32_t* faces;
  const uint8_t* tex;
  const int64_t* obj;
  const float* cull;      // MeshStore::cull ([n_obj][8]) or null: back faces of closed objects are not binned (raster_bin_kernel)
  const float4* face_planes;  // MeshStore::face_planes
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
  // per-(view, band) triangle lists built by raster_bin_kernel
  int32_t* bin_count;   // [chunk views][n_bands]
  int32_t* bin_list;    // [chunk views][n_bands][bin_cap]
  int bin_cap, view0, max_faces, max_verts;
  // screen-space vertices of the chunk's views, written by raster_xform_kernel:
  // [view][vertex] {X, Y, Z, X/Z}, {Y/Z, -, -, -}
  float4* xverts;
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
  RasterConv cv;                       // hp_raster_set_conventions, copied at launch
};

__device__ __forceinline__ void edge_fn(const float* P0, int i0, const float* P1, int i1, float* e) {
  const float* P = P0;
  const float* Q = P1;
  float sgn = 1.0f;
  if (i1 < i0) { P = P1; Q = P0; sgn = -1.0f; }
  e[0] = sgn * fmaf(P[1], Q[2], -(P[2] * Q[1]));
  e[1] = sgn * fmaf(P[2], Q[0], -(P[0] * Q[2]));
  e[2] = sgn * fmaf(P[0], Q[1], -(P[1] * Q[0]));
}

// vertices are stored as float4 (xyz + pad): one 16-B gather per vertex
__device__ __forceinline__ void xform_vertex(const float* T, const float* Kv, const float4 p, float* o) {
  float cx = fmaf(T[0], p.x, fmaf(T[1], p.y, fmaf(T[2], p.z, T[3])));
  float cy = fmaf(T[4], p.x, fmaf(T[5], p.y, fmaf(T[6], p.z, T[7])));
  float cz = fmaf(T[8], p.x, fmaf(T[9], p.y, fmaf(T[10], p.z, T[11])));
  o[0] = fmaf(Kv[0], cx, fmaf(Kv[1], cy, Kv[2] * cz));
  o[1] = fmaf(Kv[4], cy, Kv[5] * cz);
  o[2] = cz;
}

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
__device__ __forceinline__ void tex_fetch_aniso(const RasterConv& cv, const uint8_t* tex, int tw, int th, int nlev, float u, float v, float ux,
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
#ifdef HP_RABL_COUNT
// tools/raster_walk_count.py: [0] low half: walk iterations (wave level), high half: probe-pair iterations (wave level), [1] sum of bbox
// areas, [2] triangles, [3] survivors, [4] wave batches, [5] shading invocations with the anisotropic filter, [6] their probes,
// [7] those blending two levels
__device__ unsigned long long hp_dbg_cnt[8];
#endif
struct MipTable {
  int off[16], w[16], h[16];
  int sh[16];        // log2(w) + 2 when the texture's sizes are powers of two (row pitch in bytes as a shift)
  int p2;            // both sizes of level 0 are powers of two (then every level's are)
  float tt[16][16];  // probe position t = i / (N + 1) - 0.5 at [N - 1][i - 1]: one IEEE division per entry and workgroup, not per probe
};

struct BiTexels { uchar4 a, b, c, d; };

// The same filter for power-of-two textures (every level a power of two): wraps are masks, row pitches shifts, texel
// addresses 32-bit offsets from the view's texture (a scalar base), probe positions from the workgroup's table.  Identical
// values: the integer identities hold for every operand, the table entries are the quotients the loop used to recompute.
struct BiTapP2 { uint32_t o00, o01, o10, o11; float fx, fy; };
__device__ __forceinline__ BiTapP2 bi_setup_p2(int off, int w, int h, int sh, float u, float v) {
  const float x = fmaf(u, (float)w, -0.5f);
  const float y = fmaf(1.0f - v, (float)h, -0.5f);
  const float xf = floorf(x), yf = floorf(y);
  BiTapP2 t;
  t.fx = x - xf; t.fy = y - yf;
  const int x0 = (int)xf & (w - 1), y0 = (int)yf & (h - 1);
  const int x1 = (x0 + 1) & (w - 1), y1 = (y0 + 1) & (h - 1);
  const uint32_t r0 = (uint32_t)off + ((uint32_t)y0 << sh), r1 = (uint32_t)off + ((uint32_t)y1 << sh);
  t.o00 = r0 + 4u * (uint32_t)x0; t.o01 = r0 + 4u * (uint32_t)x1;
  t.o10 = r1 + 4u * (uint32_t)x0; t.o11 = r1 + 4u * (uint32_t)x1;
  return t;
}
__device__ __forceinline__ BiTexels bi_load_p2(const uint8_t* tex, const BiTapP2& t) {
  BiTexels r;
  r.a = *reinterpret_cast<const uchar4*>(tex + t.o00); r.b = *reinterpret_cast<const uchar4*>(tex + t.o01);
  r.c = *reinterpret_cast<const uchar4*>(tex + t.o10); r.d = *reinterpret_cast<const uchar4*>(tex + t.o11);
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

__device__ __forceinline__ void tex_fetch_aniso_p2(const RasterConv& cv, const uint8_t* tex, const MipTable& mt, int tw, int th, int nlev, float u, float v,
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
  const int off0 = mt.off[l0], w0 = mt.w[l0], h0 = mt.h[l0], s0 = mt.sh[l0];
  const int off1 = mt.off[l1], w1 = mt.w[l1], h1 = mt.h[l1], s1 = mt.sh[l1];
  const float* const tt = mt.tt[N - 1];
#ifdef HP_RABL_COUNT
  atomicAdd(&hp_dbg_cnt[5], 1ull); atomicAdd(&hp_dbg_cnt[6], (unsigned long long)N); if (two) atomicAdd(&hp_dbg_cnt[7], 1ull);
#endif
  float acc[3] = {0.0f, 0.0f, 0.0f};
  for (int i = 1; i <= N; i += 2) {
    const bool second = i + 1 <= N;
#ifdef HP_RABL_COUNT
    if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&hp_dbg_cnt[0], 1ull << 32);
#endif
    const float ta = tt[i - 1];
    const float tb = tt[second ? i : i - 1];
    const float sua = fmaf(ta, du, u), sva = fmaf(ta, dv, v), sub = fmaf(tb, du, u), svb = fmaf(tb, dv, v);
    const BiTapP2 a0 = bi_setup_p2(off0, w0, h0, s0, sua, sva), a1 = bi_setup_p2(off1, w1, h1, s1, sua, sva);
    const BiTapP2 b0 = bi_setup_p2(off0, w0, h0, s0, sub, svb), b1 = bi_setup_p2(off1, w1, h1, s1, sub, svb);
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

struct TriSetup {
  float e0[3], e1[3], e2[3];
  float z0, z1, z2;  // camera-space depth of the three corners
  float det;
  int x0, x1, y0, y1;  // inclusive pixel bbox clipped to the band; empty if x0 > x1
};

// the six 16-B records of a triangle's corners ({X, Y, Z, X/Z}, {Y/Z, ...} per vertex), loaded ahead of their use
struct TriVerts { float4 p0, p1, p2; float v0, v1, v2; };
__device__ __forceinline__ TriVerts load_tri_verts(const float4* xv, const int32_t* tri) {
  TriVerts t;
  t.p0 = xv[2 * tri[0]]; t.p1 = xv[2 * tri[1]]; t.p2 = xv[2 * tri[2]];
  t.v0 = xv[2 * tri[0] + 1].x; t.v1 = xv[2 * tri[1] + 1].x; t.v2 = xv[2 * tri[2] + 1].x;
  return t;
}

// Screen-space vertices + inclusive pixel bbox (clipped to the image).  Returns false when the
// triangle is outside the clip range or the image.
__device__ __forceinline__ bool tri_bbox(const RasterArgs& a, const TriVerts& t, float (&V0)[3],
                                         float (&V1)[3], float (&V2)[3], int& x0, int& x1, int& y0, int& y1) {
  const float4 p0 = t.p0, p1 = t.p1, p2 = t.p2;
  V0[0] = p0.x; V0[1] = p0.y; V0[2] = p0.z;
  V1[0] = p1.x; V1[1] = p1.y; V1[2] = p1.z;
  V2[0] = p2.x; V2[1] = p2.y; V2[2] = p2.z;
  float zmin = fminf(V0[2], fminf(V1[2], V2[2])), zmax = fmaxf(V0[2], fmaxf(V1[2], V2[2]));
  if (!(zmax >= kZNear) || !(zmin <= kZFar)) return false;
  x0 = 0; x1 = a.w - 1; y0 = 0; y1 = a.h - 1;
  if (zmin > 1e-6f) {
    const float u0 = p0.w, u1 = p1.w, u2 = p2.w;  // X / Z, Y / Z: divided once per vertex and view
    const float v0 = t.v0, v1 = t.v1, v2 = t.v2;
    float umin = fminf(u0, fminf(u1, u2)), umax = fmaxf(u0, fmaxf(u1, u2));
    float vmin = fminf(v0, fminf(v1, v2)), vmax = fmaxf(v0, fmaxf(v1, v2));
    if (!(umax >= 0.0f) || !(umin <= (float)a.w) || !(vmax >= 0.0f) || !(vmin <= (float)a.h)) return false;
    float xa = ceilf(umin - 0.5f), xb = floorf(umax - 0.5f);
    float ya = ceilf(vmin - 0.5f), yb = floorf(vmax - 0.5f);
    if (a.msaa) {  // some sample of pixel j inside [umin, umax]: offsets run from lo to hi (0.125 to 0.875 in the default pattern)
      xa = ceilf(umin - a.cv.hi_x); xb = floorf(umax - a.cv.lo_x); ya = ceilf(vmin - a.cv.hi_y); yb = floorf(vmax - a.cv.lo_y);
    }
    x0 = xa < 0.0f ? 0 : (int)xa; x1 = xb > (float)(a.w - 1) ? a.w - 1 : (int)xb;
    y0 = ya < 0.0f ? 0 : (int)ya; y1 = yb > (float)(a.h - 1) ? a.h - 1 : (int)yb;
  }
  return y0 <= y1 && x0 <= x1;
}

// Returns false when the triangle cannot touch rows [row0, row1] of this view.
__device__ __forceinline__ bool setup_triangle(const RasterArgs& a, const TriVerts& t, const int32_t* tri, int row0,
                                               int row1, TriSetup& s) {
  float V0[3], V1[3], V2[3];
  if (!tri_bbox(a, t, V0, V1, V2, s.x0, s.x1, s.y0, s.y1)) return false;
  if (s.y0 < row0) s.y0 = row0;
  if (s.y1 > row1) s.y1 = row1;
  if (s.y0 > s.y1) return false;
  edge_fn(V1, tri[1], V2, tri[2], s.e0);
  edge_fn(V2, tri[2], V0, tri[0], s.e1);
  edge_fn(V0, tri[0], V1, tri[1], s.e2);
  s.det = fmaf(V0[0], s.e0[0], fmaf(V0[1], s.e0[1], V0[2] * s.e0[2]));
  if (!(s.det != 0.0f) || !isfinite(s.det)) return false;
  s.z0 = V0[2]; s.z1 = V1[2]; s.z2 = V2[2];
  return true;
}

template <int NS>
__device__ __forceinline__ void shade_pixel(const RasterConv& cv, const TriSetup& s, int i, int j, uint32_t f,
                                            unsigned long long* zb, int row0, int w) {
#ifdef HP_RABL_COUNT
  int n_in = 0;
#endif
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) {
    const float pv = (float)i + sample_y(cv, NS, sm), pu = (float)j + sample_x(cv, NS, sm);
    float l0 = fmaf(s.e0[0], pu, fmaf(s.e0[1], pv, s.e0[2]));
    float l1 = fmaf(s.e1[0], pu, fmaf(s.e1[1], pv, s.e1[2]));
    float l2 = fmaf(s.e2[0], pu, fmaf(s.e2[1], pv, s.e2[2]));
    float sum = l0 + l1 + l2;
    bool in_pos = (l0 >= 0.0f) & (l1 >= 0.0f) & (l2 >= 0.0f) & (sum > 0.0f);
    bool in_neg = (l0 <= 0.0f) & (l1 <= 0.0f) & (l2 <= 0.0f) & (sum < 0.0f);
    if (!(in_pos | in_neg)) continue;
#ifdef HP_RABL_COUNT
    ++n_in;
#endif
    // depth = the vertex depths interpolated with the perspective-correct barycentrics (oracle.c explains why not det / sum)
#ifdef HP_RABL_NO_SAMPLE_DIV
    float Z = fmaf(l0, s.z0, fmaf(l1, s.z1, l2 * s.z2)) * sum;
#else
    float Z = fmaf(l0, s.z0, fmaf(l1, s.z1, l2 * s.z2)) / sum;
#endif
    if (!(Z >= kZNear) || !(Z <= kZFar)) continue;
    unsigned long long key = ((unsigned long long)__float_as_uint(Z) << 32) | f;
#ifndef HP_RABL_NO_SAMPLE_ATOMIC
    atomicMin(&zb[((i - row0) * w + j) * NS + sm], key);
#else
    if (key == 12345ull) zb[0] = key;
#endif
  }
#ifdef HP_RABL_COUNT
  if (NS > 1) { atomicAdd(&hp_dbg_cnt[7], (unsigned long long)(n_in > 0) << 32); atomicAdd(&hp_dbg_cnt[6], (unsigned long long)n_in << 32); }
#endif
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

// Pass 0: one lane per (view, vertex): camera + intrinsics transform and the perspective division,
// once instead of once per (triangle corner, band, covered pixel).  Same operations in the same
// order as xform_vertex / the divisions of the per-triangle code it replaces.
__global__ __launch_bounds__(256) void raster_xform_kernel(RasterArgs a) {
  const int lv = blockIdx.y, view = a.view0 + lv;
  const int v = blockIdx.x * 256 + threadIdx.x;
  // the band counters of this view start at zero for the binning pass that follows in stream order (a
  // hipMemsetAsync did this: captured in a hipGraph, the memset node is skipped from the second replay on --
  // counters kept growing past the lists)
  if (blockIdx.x == 0)
    for (int b = threadIdx.x; b < a.n_bands; b += 256) a.bin_count[(int64_t)lv * a.n_bands + b] = 0;
  const int item = view / a.views_per_item;
  const int64_t* ob = a.obj + 8 * (int64_t)a.obj_ids[item];
  if (v >= (int)ob[1]) return;
  float T[12], Kv[9], o[3];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = a.TCO[16 * (int64_t)view + k];
#pragma unroll
  for (int k = 0; k < 9; ++k) Kv[k] = a.K[9 * (int64_t)view + k];
  xform_vertex(T, Kv, a.verts4[ob[0] + v], o);
  float4* dst = a.xverts + 2 * ((int64_t)lv * a.max_verts + v);
  dst[0] = make_float4(o[0], o[1], o[2], o[0] / o[2]);
  dst[1] = make_float4(o[1] / o[2], 0.f, 0.f, 0.f);
}

// Pass 1: one lane per (view, triangle) -> append the triangle to the list of every band its
// bounding box touches.  Appends are aggregated per wave (one atomic per band per wave).
__global__ __launch_bounds__(kBinThreads) void raster_bin_kernel(RasterArgs a) {
#ifdef HP_RASTER_ACQUIRE
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  const int lv = blockIdx.y;               // view within the chunk
  const int view = a.view0 + lv;
  const int f = blockIdx.x * kBinThreads + threadIdx.x;
  const ViewXform x = load_view(a, view);
  const int item = view / a.views_per_item;
  const int64_t* ob = a.obj + 8 * (int64_t)a.obj_ids[item];
  const int nf = x.finite ? (int)ob[3] : 0;
  // the view may cull: the object is closed (MeshStore::cull), the camera is outside its bounding sphere, and the whole
  // sphere lies beyond the near plane (a clipped object shows its inside)
  float cull_sign = 0.f, cam[3] = {0.f, 0.f, 0.f};
  if (a.cull && x.finite) {
    const float* const cu = a.cull + 8 * (int64_t)a.obj_ids[item];
    const float cx = fmaf(x.T[0], cu[0], fmaf(x.T[1], cu[1], fmaf(x.T[2], cu[2], x.T[3])));
    const float cy = fmaf(x.T[4], cu[0], fmaf(x.T[5], cu[1], fmaf(x.T[6], cu[2], x.T[7])));
    const float cz = fmaf(x.T[8], cu[0], fmaf(x.T[9], cu[1], fmaf(x.T[10], cu[2], x.T[11])));
    const float r = cu[3] * 1.001f + 1e-6f;  // (a scaled rotation is not expected in T; the margin covers its rounding)
    if (cx * cx + cy * cy + cz * cz > r * r && cz - r > kZNear) cull_sign = cu[4];
    // the camera centre in the object's frame: - R^T t
    cam[0] = -(x.T[0] * x.T[3] + x.T[4] * x.T[7] + x.T[8] * x.T[11]);
    cam[1] = -(x.T[1] * x.T[3] + x.T[5] * x.T[7] + x.T[9] * x.T[11]);
    cam[2] = -(x.T[2] * x.T[3] + x.T[6] * x.T[7] + x.T[10] * x.T[11]);
  }
  int b0 = 1, b1 = 0;  // empty band range
  if (f < nf) {
    const int32_t* fbase = a.faces + 3 * ob[2];
    int32_t tri[3] = {fbase[3 * f], fbase[3 * f + 1], fbase[3 * f + 2]};
    float V0[3], V1[3], V2[3];
    int x0, x1, y0, y1;
    if (tri_bbox(a, load_tri_verts(a.xverts + 2 * (int64_t)lv * a.max_verts, tri), V0, V1, V2, x0, x1, y0, y1)) {
      b0 = y0 / a.band_rows;
      b1 = y1 / a.band_rows;
      // Back faces of a CLOSED object seen from outside are never visible (the renders are two-sided like the reference's,
      // panda3d_scene_renderer.py:102: every ray meets a front face of the closed surface first, and the canonical edge
      // functions make that surface watertight).  The first version took the facing from the screen-space vertices of the
      // vertex pass (the sign of V0 . ((V1 - V0) x (V2 - V0))): correct, but in two-lane steps the decision was then NOT
      // reproducible for a few triangles per launch (single-pixel colour differences from run to run; with a decision that does
      // not read those records, or without culling, bit-identical) -- so it reads object-space data only.
      if (cull_sign != 0.f) {
        // the camera is on the inner side of the face's plane, by 100x the rounding of the test (object-space data only: the
        // face's plane from hp_mesh_store_create and the camera centre - R^T t; nothing a previous launch wrote)
        const float4 pl = a.face_planes[ob[2] + f];
        const float t0 = pl.x * cam[0], t1 = pl.y * cam[1], t2 = pl.z * cam[2];
        const float sd = (t0 + t1) + (t2 - pl.w);
        const float mag = fabsf(t0) + fabsf(t1) + fabsf(t2) + fabsf(pl.w);
        if (sd * cull_sign < -1e-5f * mag) { b0 = 1; b1 = 0; }
      }
    }
  }
  // Appends are aggregated per workgroup: a lane takes its slot(s) from LDS counters (one returning LDS atomic per band it
  // touches -- a triangle of these meshes touches one or two), one global atomic per band and workgroup reserves the
  // workgroup's range in the band's list (same-address L2 atomics serialise).  (The first version walked the band RANGE of
  // every wave with ballots: mesh order is not screen order, so a wave spanned 20-40 of the 60 four-row bands.)  The order
  // of a list is arbitrary either way; the z-buffer minimum does not depend on it.
  __shared__ int wg_cnt[kMaxBands], wg_base[kMaxBands];
  const int tid = threadIdx.x;
  for (int b = tid; b < a.n_bands; b += kBinThreads) wg_cnt[b] = 0;
  __syncthreads();
  int slot0 = 0, slot1 = 0;  // local slots in the first two bands
  if (b0 <= b1) {
    slot0 = atomicAdd(&wg_cnt[b0], 1);
    if (b1 > b0) slot1 = atomicAdd(&wg_cnt[b0 + 1], 1);
  }
  int extra[6];  // bands b0 + 2 .. b0 + 7 of a triangle taller than two bands
#pragma unroll
  for (int k = 0; k < 6; ++k) extra[k] = (b0 <= b1 && b0 + 2 + k <= b1) ? atomicAdd(&wg_cnt[b0 + 2 + k], 1) : 0;
  for (int b = b0 + 8; b <= b1; ++b) {  // rare: beyond eight bands a triangle appends itself with a global atomic per band
    const int slot = atomicAdd(&a.bin_count[lv * a.n_bands + b], 1);
    if (slot < a.bin_cap) a.bin_list[((int64_t)lv * a.n_bands + b) * a.bin_cap + slot] = f;
  }
  __syncthreads();
  for (int b = tid; b < a.n_bands; b += kBinThreads)
    if (wg_cnt[b] > 0) wg_base[b] = atomicAdd(&a.bin_count[lv * a.n_bands + b], wg_cnt[b]);
  __syncthreads();
  if (b0 <= b1) {
    auto put = [&](int b, int local) {
      const int slot = wg_base[b] + local;
      if (slot < a.bin_cap) a.bin_list[((int64_t)lv * a.n_bands + b) * a.bin_cap + slot] = f;
    };
    put(b0, slot0);
    if (b1 > b0) put(b0 + 1, slot1);
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (b0 + 2 + k <= b1) put(b0 + 2 + k, extra[k]);
  }
}

// One fragment-shader invocation: colour and normal code of triangle f at the CENTRE of pixel (i, j) (attributes
// extrapolated when the centre lies outside the triangle: multisampled edge pixels) -- oracle.c shade_centre.
struct ShadeCtx {
  const float* T; const float* Kv; const float* amb; const float4* xv; const int32_t* fbase;
  int64_t voff, toff; int tw, th, view, q8, nlev, aniso;
  const MipTable* mips;  // per-workgroup table of the object's mip levels (LDS)
  bool need_normal;      // the view renders normals or has point lights: otherwise the normal is never looked at
};
template <bool ANISO>
__device__ __forceinline__ void shade_centre(const RasterArgs& a, const ShadeCtx& cx, int f, int i, int j, float* o_rgb, float* o_n) {
  const float* T = cx.T; const float* Kv = cx.Kv; const float* amb = cx.amb; const float4* xv = cx.xv;
  const int32_t* fbase = cx.fbase;
  const int64_t voff = cx.voff, toff = cx.toff;
  const int tw = cx.tw, th = cx.th, view = cx.view, q8 = cx.q8;
  int32_t tri[3] = {fbase[3 * f], fbase[3 * f + 1], fbase[3 * f + 2]};
  float e0[3], e1[3], e2[3];
  const float4 q0 = xv[2 * tri[0]], q1 = xv[2 * tri[1]], q2 = xv[2 * tri[2]];
  const float V0[3] = {q0.x, q0.y, q0.z}, V1[3] = {q1.x, q1.y, q1.z}, V2[3] = {q2.x, q2.y, q2.z};
  edge_fn(V1, tri[1], V2, tri[2], e0);
  edge_fn(V2, tri[2], V0, tri[0], e1);
  edge_fn(V0, tri[0], V1, tri[1], e2);
  const float pu = (float)j + 0.5f, pv = (float)i + 0.5f;
  float l0 = fmaf(e0[0], pu, fmaf(e0[1], pv, e0[2]));
  float l1 = fmaf(e1[0], pu, fmaf(e1[1], pv, e1[2]));
  float l2 = fmaf(e2[0], pu, fmaf(e2[1], pv, e2[2]));
  float sum = l0 + l1 + l2;
  float b0 = l0 / sum, b1 = l1 / sum, b2 = l2 / sum;
  const float Z = fmaf(l0, V0[2], fmaf(l1, V1[2], l2 * V2[2])) / sum;  // = the key's depth when the centre is covered
  const int64_t g0 = voff + tri[0], g1 = voff + tri[1], g2 = voff + tri[2];
  float alb[3];
  if (toff >= 0) {
    const float2 t0 = *reinterpret_cast<const float2*>(a.uvs + 2 * g0);
    const float2 t1 = *reinterpret_cast<const float2*>(a.uvs + 2 * g1);
    const float2 t2 = *reinterpret_cast<const float2*>(a.uvs + 2 * g2);
    float tu = fmaf(b0, t0.x, fmaf(b1, t1.x, b2 * t2.x));
    float tv = fmaf(b0, t0.y, fmaf(b1, t1.y, b2 * t2.y));
    if (ANISO && cx.nlev > 1) {
      // screen-space derivatives of the perspective-correct barycentrics: b_i = l_i / sum, l_i affine in (x, y)
      const float sx = e0[0] + e1[0] + e2[0], sy = e0[1] + e1[1] + e2[1];
      const float bx[3] = {(e0[0] - b0 * sx) / sum, (e1[0] - b1 * sx) / sum, (e2[0] - b2 * sx) / sum};
      const float by[3] = {(e0[1] - b0 * sy) / sum, (e1[1] - b1 * sy) / sum, (e2[1] - b2 * sy) / sum};
      const float ux = fmaf(bx[0], t0.x, fmaf(bx[1], t1.x, bx[2] * t2.x));
      const float vx = fmaf(bx[0], t0.y, fmaf(bx[1], t1.y, bx[2] * t2.y));
      const float uy = fmaf(by[0], t0.x, fmaf(by[1], t1.x, by[2] * t2.x));
      const float vy = fmaf(by[0], t0.y, fmaf(by[1], t1.y, by[2] * t2.y));
      if (cx.mips->p2) tex_fetch_aniso_p2(a.cv, a.tex + toff, *cx.mips, tw, th, cx.nlev, tu, tv, ux, vx, uy, vy, alb);
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

// HALF: fp16 destinations (the input of an fp16 network plan) -- its own instantiation so that the fp32 path's register
// budget (80 VGPRs: three 512-thread workgroups per CU) does not carry the 16-half record assembly
template <int NS, bool HALF, bool ANISO>
__global__ __launch_bounds__(band_threads(NS), (NS == 1 && !HALF && !ANISO) ? 6 : 4) void raster_kernel(RasterArgs a, int npix_max) {
  constexpr int kThreads = band_threads(NS);  // (shadows the single-sample constant)
#ifdef HP_RASTER_ACQUIRE
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int big_q[kBigQueue];
  __shared__ int big_n, n_cov, span_max[2];
  __shared__ MipTable mips;
  __shared__ uint32_t cov_q[NS == 1 ? 1 : kThreads / 64][NS == 1 ? 1 : 128];  // multisampling: (lane, pixel) pairs awaiting their sample tests

  // XCD-aware renumbering: dispatch order b -> XCD b % 8; give each XCD a contiguous range.
  const int total = a.n * a.n_bands;  // a.n = views of this chunk
  const int per_xcd = (total + 7) / 8;
  const int lin = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lin >= total) return;
  const int view = a.view0 + lin / a.n_bands;
  const int band = lin % a.n_bands;
  const int row0 = band * a.band_rows;
  const int row1 = min(a.h, row0 + a.band_rows) - 1;
  const int npix = (row1 - row0 + 1) * a.w;
  const int tid = threadIdx.x;
  const BandLds L = carve_lds(smem, npix_max, NS, a.band_rows);
  unsigned long long* const zb = L.zb;

  float T[12], Kv[9];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = a.TCO[16 * (int64_t)view + k];
#pragma unroll
  for (int k = 0; k < 9; ++k) Kv[k] = a.K[9 * (int64_t)view + k];
  bool finite = true;
#pragma unroll
  for (int k = 0; k < 12; ++k) finite &= isfinite(T[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) finite &= isfinite(a.TCO[16 * (int64_t)view + 12 + k]);
#pragma unroll
  for (int k = 0; k < 9; ++k) finite &= isfinite(Kv[k]);

  const int item = view / a.views_per_item, vi = view % a.views_per_item;
  const int64_t* ob = a.obj + 8 * (int64_t)a.obj_ids[item];
  const int64_t voff = ob[0], foff = ob[2], toff = ob[4];
  const int nf = finite ? (int)ob[3] : 0;
  const int tw = (int)ob[5], th = (int)ob[6];
  const float4* xv = a.xverts + 2 * (int64_t)(lin / a.n_bands) * a.max_verts;
  const int32_t* fbase = a.faces + 3 * foff;

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

  if (tid == 0) {  // level k of the texture: max(1, tw >> k) x max(1, th >> k), stored behind the levels before it
    int off = 0, lw = tw, lh = th;
    for (int k = 0; k < 16; ++k) {
      mips.off[k] = off; mips.w[k] = lw; mips.h[k] = lh;
      mips.sh[k] = 33 - __clz(lw);  // log2(lw) + 2 for a power of two
      off += 4 * lw * lh; lw = lw > 1 ? lw >> 1 : 1; lh = lh > 1 ? lh >> 1 : 1;
    }
    mips.p2 = tw > 0 && th > 0 && (tw & (tw - 1)) == 0 && (th & (th - 1)) == 0;
  }
  if (ANISO && tid < 256) {
    const int N = (tid >> 4) + 1, i = (tid & 15) + 1;
    mips.tt[N - 1][i - 1] = (float)i / (float)(N + 1) - 0.5f;
  }
  // ---- coverage + depth: only the triangles binned to this band ----
  const int cnt = nf > 0 ? min(a.bin_count[lin], a.bin_cap) : 0;
  const int32_t* list = a.bin_list + (int64_t)lin * a.bin_cap;
  // a band no triangle touches skips the z-buffer passes altogether: the output pass streams the background (and the crop)
  const bool band_empty = cnt == 0;
  if (!band_empty) {
    for (int p = tid; p < npix * NS; p += kThreads) zb[p] = kKeyEmpty;
    if (tid == 0) { big_n = 0; n_cov = 0; }
  }
  __syncthreads();
#ifdef HP_RABL_NO_COVER
  const int cnt_loop = 0;
#else
  const int cnt_loop = cnt;
#endif
  // three-stage software pipeline over the dependent gathers list -> corner indices -> vertex records: while triangle k
  // is rasterised, the vertex records of k + T, the corner indices of k + 2T and the list entry of k + 3T are in flight
  // (every stage needs the previous stage's data of the same triangle); indices past the end read entry 0: harmless
  int f0 = tid < cnt_loop ? list[tid] : 0;
  int f1 = tid + kThreads < cnt_loop ? list[tid + kThreads] : 0;
  int f2 = tid + 2 * kThreads < cnt_loop ? list[tid + 2 * kThreads] : 0;
  int32_t tri0[3] = {fbase[3 * f0], fbase[3 * f0 + 1], fbase[3 * f0 + 2]};
  int32_t tri1[3] = {fbase[3 * f1], fbase[3 * f1 + 1], fbase[3 * f1 + 2]};
  TriVerts tv0 = load_tri_verts(xv, tri0);
  if (NS == 1) {
    for (int k = tid; k < cnt_loop; k += kThreads) {
      const int f = f0;
      const int32_t tri[3] = {tri0[0], tri0[1], tri0[2]};
      const TriVerts tv = tv0;
      f0 = f1; tri0[0] = tri1[0]; tri0[1] = tri1[1]; tri0[2] = tri1[2];
      tv0 = load_tri_verts(xv, tri0);
      f1 = f2;
      tri1[0] = fbase[3 * f1]; tri1[1] = fbase[3 * f1 + 1]; tri1[2] = fbase[3 * f1 + 2];
      f2 = k + 3 * kThreads < cnt_loop ? list[k + 3 * kThreads] : 0;
      TriSetup s;
      if (!setup_triangle(a, tv, tri, row0, row1, s)) continue;
      const int area = (s.x1 - s.x0 + 1) * (s.y1 - s.y0 + 1);
      if (area > kBigArea) {
        int q = atomicAdd(&big_n, 1);
        if (q < kBigQueue) { big_q[q] = f; continue; }
      }
      for (int i = s.y0; i <= s.y1; ++i)
        for (int j = s.x0; j <= s.x1; ++j) shade_pixel<NS>(a.cv, s, i, j, (uint32_t)f, zb, row0, a.w);
    }
  } else {
    // Multisampling.  The bounding box is grown by the sample spread, so most of its pixels have no sample inside the
    // triangle, and the five exact sample tests of the pixels that do are ~100 instructions.  Two measures keep the lanes
    // of a wave busy: (1) a pixel is dropped when the edge functions at its CENTRE, widened by the largest sample offset
    // (and by a bound on the rounding of the fp32 evaluation), leave no sample on the inner side of some edge for either
    // orientation -- the five exact tests would all fail anyway; (2) the surviving (triangle, pixel) pairs of the whole
    // wave go through a per-wave queue in LDS and are popped 64 at a time, each lane fetching the owner lane's edge
    // functions with ds_bpermute: the expensive part runs on dense lanes whatever the sizes of the 64 bounding boxes are.
    // plain LDS accesses (a `volatile` pointer made them FLAT stores with system scope + s_waitcnt vmcnt(0) per walk step,
    // which also drained the triangle prefetch).  One wave writes and reads its own queue: the LDS executes a wave's
    // accesses in order, and the wave barriers below keep the compiler from moving them across one another.
    uint32_t* const q = cov_q[tid >> 6];
    const int lane = tid & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    int qh = 0, qt = 0;  // wave-uniform
    for (int kb = 0; kb < cnt_loop; kb += kThreads) {
      const int k = kb + tid;
      const int f = f0;
      const int32_t tri[3] = {tri0[0], tri0[1], tri0[2]};
      const TriVerts tv = tv0;
      f0 = f1; tri0[0] = tri1[0]; tri0[1] = tri1[1]; tri0[2] = tri1[2];
      tv0 = load_tri_verts(xv, tri0);
      f1 = f2;
      tri1[0] = fbase[3 * f1]; tri1[1] = fbase[3 * f1 + 1]; tri1[2] = fbase[3 * f1 + 2];
      f2 = k + 3 * kThreads < cnt_loop ? list[k + 3 * kThreads] : 0;
      TriSetup s{};
      bool valid = k < cnt_loop && setup_triangle(a, tv, tri, row0, row1, s);
      if (valid && (s.x1 - s.x0 + 1) * (s.y1 - s.y0 + 1) > kBigArea) {
        int bq = atomicAdd(&big_n, 1);
        if (bq < kBigQueue) { big_q[bq] = f; valid = false; }
      }
      float m[3] = {0.f, 0.f, 0.f};
      if (valid) {
        const float* const ee[3] = {s.e0, s.e1, s.e2};
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const float ax = fabsf(ee[e][0]), ay = fabsf(ee[e][1]);
          const float sl = 4e-6f * fmaf(ax, (float)(s.x1 + 1), fmaf(ay, (float)(s.y1 + 1), fabsf(ee[e][2])));
          // the largest |edge function difference| between a sample and the pixel centre (default pattern: offsets (0.125, 0.375), (0.375, 0.125))
          const float m01 = fmaxf(fmaf(a.cv.dxa[0], ax, a.cv.dya[0] * ay), fmaf(a.cv.dxa[1], ax, a.cv.dya[1] * ay));
          const float m23 = fmaxf(fmaf(a.cv.dxa[2], ax, a.cv.dya[2] * ay), fmaf(a.cv.dxa[3], ax, a.cv.dya[3] * ay));
          m[e] = fmaxf(m01, m23) * 1.001f + sl;
        }
      }
#ifdef HP_RABL_COUNT
      {
        const int ar = valid ? (s.x1 - s.x0 + 1) * (s.y1 - s.y0 + 1) : 0;
        atomicAdd(&hp_dbg_cnt[1], (unsigned long long)ar);
        if (valid) atomicAdd(&hp_dbg_cnt[2], 1ull);
        if (lane == 0) atomicAdd(&hp_dbg_cnt[4], 1ull);
      }
#endif
      int ci = s.y0, cj = s.x0;
#ifdef HP_RABL_NO_WALK
      bool more = valid && s.z0 == 12345.f;
#else
      bool more = valid;
#endif
      while (__any(more)) {
        bool surv = false;
        const int pi = ci, pj = cj;
        if (more) {
          const float pu = (float)cj + 0.5f, pv = (float)ci + 0.5f;
          const float l0 = fmaf(s.e0[0], pu, fmaf(s.e0[1], pv, s.e0[2]));
          const float l1 = fmaf(s.e1[0], pu, fmaf(s.e1[1], pv, s.e1[2]));
          const float l2 = fmaf(s.e2[0], pu, fmaf(s.e2[1], pv, s.e2[2]));
          const bool no_pos = (l0 + m[0] < 0.0f) | (l1 + m[1] < 0.0f) | (l2 + m[2] < 0.0f);
          const bool no_neg = (l0 - m[0] > 0.0f) | (l1 - m[1] > 0.0f) | (l2 - m[2] > 0.0f);
          surv = !(no_pos & no_neg);
          if (++cj > s.x1) { cj = s.x0; ++ci; }
          more = ci <= s.y1;
        }
        const unsigned long long sm = __ballot(surv);
#ifdef HP_RABL_COUNT
        if (lane == 0) { atomicAdd(&hp_dbg_cnt[0], 1ull); atomicAdd(&hp_dbg_cnt[3], (unsigned long long)__popcll(sm)); }
#endif
        if (surv) q[(qt + __popcll(sm & lt_mask)) & 127] = ((uint32_t)lane << 16) | (uint32_t)((pi - row0) * a.w + pj);  // npix_max < 65536
        qt += __popcll(sm);
        __builtin_amdgcn_wave_barrier();
        while (qt - qh >= 64 || (qt > qh && !__any(more))) {  // a full wave of pairs, or the rest once every box is walked
          const int n = min(64, qt - qh);
          const uint32_t e = q[(qh + lane) & 127];
          __builtin_amdgcn_wave_barrier();
          const int owner = (int)(e >> 16) & 63;
          const int ep = (int)(e & 0xFFFFu);
          const int er = (int)(((unsigned long long)ep * a.w_magic) >> 32);
          TriSetup t;
#pragma unroll
          for (int c = 0; c < 3; ++c) { t.e0[c] = __shfl(s.e0[c], owner); t.e1[c] = __shfl(s.e1[c], owner); t.e2[c] = __shfl(s.e2[c], owner); }
          t.z0 = __shfl(s.z0, owner); t.z1 = __shfl(s.z1, owner); t.z2 = __shfl(s.z2, owner);
          const int tf = __shfl(f, owner);
#ifndef HP_RABL_NO_POP
          if (lane < n) shade_pixel<NS>(a.cv, t, row0 + er, ep - er * a.w, (uint32_t)tf, zb, row0, a.w);
#else
          if (lane < n && t.e0[0] == 12345.f && tf == 77) zb[0] = 0;
#endif
          qh += n;
        }
      }
    }
  }
  if (!band_empty) __syncthreads();
  const int nbig = band_empty ? 0 : min(big_n, kBigQueue);
  for (int q = 0; q < nbig; ++q) {
    const int f = big_q[q];
    int32_t tri[3] = {fbase[3 * f], fbase[3 * f + 1], fbase[3 * f + 2]};
    TriSetup s;
    if (!setup_triangle(a, load_tri_verts(xv, tri), tri, row0, row1, s)) continue;
    const int bw = s.x1 - s.x0 + 1;
    const int area = bw * (s.y1 - s.y0 + 1);
    for (int p = tid; p < area; p += kThreads)
      shade_pixel<NS>(a.cv, s, s.y0 + p / bw, s.x0 + p % bw, (uint32_t)f, zb, row0, a.w);
  }
  if (!band_empty) __syncthreads();

  // ---- shading of the compacted covered pixels (8-bit colour codes back into the z-buffer slots) ----
  const int q8 = 1;  // colours are 8-bit quantised like the reference's uint8 read-back (HP_RASTER_QUANT8 is implied)
  float amb[3] = {1.0f, 1.0f, 1.0f};
  if (a.ambient) { amb[0] = a.ambient[3 * view]; amb[1] = a.ambient[3 * view + 1]; amb[2] = a.ambient[3 * view + 2]; }
  const ShadeCtx cx{T, Kv, amb, xv, fbase, voff, toff, tw, th, view, q8, (int)ob[7] > 0 ? (int)ob[7] : 1,
                    (a.flags & HP_RASTER_TEX_ANISO) != 0, &mips, (a.rec ? a.want_nrm != 0 : a.nrm != nullptr) || a.n_lights > 0};
  const bool want_colour = a.rec ? true : (a.rgb != nullptr || a.nrm != nullptr);
  const bool coded = !band_empty;  // colours travel as 8-bit codes through LDS
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
      const unsigned long long m = __ballot(cov);
      int base = 0;
      if (lane == 0 && m != 0) base = atomicAdd(&n_cov, __popcll(m));
      base = __shfl(base, 0);
      if (cov) L.plist[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)p;
    }
    __syncthreads();
    const int ncov = n_cov;
#ifdef HP_RABL_NO_SHADE
    const int ncov_loop = a.w < 0 ? ncov : 0;
#else
    const int ncov_loop = ncov;
#endif
    for (int q = tid; q < ncov_loop; q += kThreads) {
      const int p = L.plist[q];
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
    for (int p0 = 0; p0 < npix; p0 += kThreads) {
      const int p = p0 + tid;
      unsigned inv = 0;  // bit sm: sample sm starts an invocation
      if (p < npix) {
        uint32_t kf[4];
        bool kc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { const unsigned long long kt = zb[p * NS + t]; kc[t] = kt != kKeyEmpty; kf[t] = (uint32_t)kt; }
        const bool cov = (kc[0] | kc[1] | kc[2] | kc[3]) && want_colour;
        zb[p * NS + (NS - 1)] &= 0xFFFFFFFF00000000ull;  // colour sums / codes 0, the depth bits stay
        L.nsum[p] = 0;
        L.ex[p] = 0;
        if (cov) {
#pragma unroll
          for (int sm = 0; sm < 4; ++sm) {
            bool first = kc[sm];
#pragma unroll
            for (int t = 0; t < sm; ++t) first &= !(kc[t] && kf[t] == kf[sm]);
            inv |= first ? (1u << sm) : 0u;
          }
        }
      }
      const int cnt = __popc(inv);
      int pre = 0, tot = 0;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const unsigned long long m = __ballot((cnt >> b) & 1);
        pre += __popcll(m & lt_mask) << b;
        tot += __popcll(m) << b;
      }
      int base = 0;
      if (lane == 0 && tot != 0) base = atomicAdd(&n_cov, tot);
      base = __shfl(base, 0) + pre;
#pragma unroll
      for (int sm = 0; sm < 4; ++sm)
        if (inv & (1u << sm)) L.plist[base++] = (unsigned short)(p | (sm << 14));
    }
    __syncthreads();
    const int ninv = n_cov;
#ifdef HP_RABL_NO_SHADE
    const int ninv_loop = a.w < 0 ? ninv : 0;
#else
    const int ninv_loop = ninv;
#endif
    for (int q = tid; q < ninv_loop; q += kThreads) {
      const unsigned e = L.plist[q];
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
    __syncthreads();
    for (int p = tid; p < npix; p += kThreads) {
      const uint32_t sr = zlo[2 * (p * NS + (NS - 1))], sn = L.nsum[p];
      if ((sr | sn) == 0) continue;  // nothing shaded: the codes are 0 already
      unsigned cr[3], cn[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { cr[c] = (((sr >> (10 * c)) & 1023u) + 2u) >> 2; cn[c] = (((sn >> (10 * c)) & 1023u) + 2u) >> 2; }
      zlo[2 * (p * NS + (NS - 1))] = cr[0] | (cr[1] << 8) | (cr[2] << 16) | (cn[0] << 24);
      L.ex[p] = (unsigned short)(cn[1] | (cn[2] << 8));
    }
  }
  __syncthreads();

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
        const float Z = __uint_as_float(zbits);
        o_d = Z > a.depth_max ? 0.0f : Z;
      }
      if (coded) {
        const uint32_t lo = (uint32_t)slot;
        const uint32_t ex = L.ex[p];
        o_rgb[0] = (float)(lo & 255u) / 255.0f; o_rgb[1] = (float)((lo >> 8) & 255u) / 255.0f; o_rgb[2] = (float)((lo >> 16) & 255u) / 255.0f;
        o_n[0] = (float)(lo >> 24) / 255.0f; o_n[1] = (float)(ex & 255u) / 255.0f; o_n[2] = (float)(ex >> 8) / 255.0f;
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
}

}  // namespace hp

// Rasteriser scratch of a mesh store for `n` views of `n_bands` bands: per-(view, band) triangle lists, their counters
// and the per-(view, vertex) screen-space records.  Grows, never shrinks; refuses to grow under stream capture (a
// captured launch would keep the pointer that is freed here) -- run the call once eagerly, or reserve.
static int raster_scratch(hp::MeshStore* ms, int n, int n_bands, hipStream_t st, int* chunk_out) {
  const size_t per_view = (size_t)n_bands * (size_t)ms->max_faces * sizeof(int32_t);
  // List budget: 8 GB of the 288 (what is allocated is what the largest call needs: 7.7 MB per multisampled 240 x 320 view); it was
  // 512 MB through round 3 (five chunks for a lane's 576 coarse views).  One chunk per call saves the repeated launches;
  // HP_RASTER_CHUNK_VIEWS=<n> forces chunks (the tests' way to run the chunked path), HP_RASTER_CHUNK_SYNC=1 synchronises the
  // stream after every chunk (diagnostics).
  static const size_t budget_mb = std::getenv("HP_RASTER_LIST_BUDGET_MB") ? (size_t)std::atoll(std::getenv("HP_RASTER_LIST_BUDGET_MB")) : 8192;
  const size_t budget = budget_mb << 20;
  int chunk = (int)std::min<size_t>(budget / (per_view ? per_view : 1), 1u << 30);
  static const int chunk_env = std::getenv("HP_RASTER_CHUNK_VIEWS") ? std::atoi(std::getenv("HP_RASTER_CHUNK_VIEWS")) : 0;
  if (chunk_env > 0 && chunk_env < chunk) chunk = chunk_env;
  if (chunk < 1) chunk = 1;
  if (chunk > n) chunk = n;
  *chunk_out = chunk;
  const size_t need_list = (size_t)chunk * per_view, need_cnt = (size_t)chunk * n_bands * sizeof(int32_t);
  const size_t need_xv = (size_t)chunk * (size_t)ms->max_verts * 2 * sizeof(float4);
  if (ms->bin_list_bytes >= need_list && ms->bin_count_bytes >= need_cnt && ms->xverts_bytes >= need_xv) return HP_OK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (st) (void)hipStreamIsCapturing(st, &cap);
  HP_REQUIRE(cap == hipStreamCaptureStatusNone,
             "hp_rasterize: the rasteriser scratch would have to grow during stream capture (hp_mesh_store_reserve_raster first)");
  auto grow = [](void** p, size_t* have, size_t need) -> int {
    if (*have >= need) return HP_OK;
    if (*p) (void)hipFree(*p);  // hipFree waits for the device: nothing in flight reads it any more
    *p = nullptr; *have = 0;
    HP_CHECK_HIP(hipMalloc(p, need));
    *have = need;
    return HP_OK;
  };
  int rc = grow((void**)&ms->bin_list, &ms->bin_list_bytes, need_list);
  if (!rc) rc = grow((void**)&ms->bin_count, &ms->bin_count_bytes, need_cnt);
  if (!rc) rc = grow((void**)&ms->xverts, &ms->xverts_bytes, need_xv);
  ms->scratch_generation += 1;
  return rc;
}

// hp_raster_set_backface_culling / HP_RASTER_NO_CULL=1 (the initial state): bin every triangle, the round-3 behaviour
static std::atomic<int> g_cull{-1};
static bool raster_cull_enabled() {
  int v = g_cull.load(std::memory_order_relaxed);
  if (v < 0) {
    v = std::getenv("HP_RASTER_NO_CULL") ? 0 : 1;
    g_cull.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}

static int raster_bands(int h, int w, int msaa) {
  const int band_rows = msaa ? hp::kBandKeysMsaa / (hp::kSamplesMsaa * w) : hp::kBandPixels / w;
  return (h + band_rows - 1) / band_rows;
}

extern "C" int hp_mesh_store_reserve_raster(hp_mesh_store* store, int n_views, int h, int w, int flags) {
  using namespace hp;
  HP_REQUIRE(store != nullptr, "hp_mesh_store_reserve_raster: null mesh store");
  HP_REQUIRE(n_views >= 0 && h > 0 && w > 0 && w <= kBandPixels, "hp_mesh_store_reserve_raster: bad size");
  if (n_views == 0) return HP_OK;
  int chunk = 0;
  int rc = raster_scratch(store, n_views, raster_bands(h, w, 0), nullptr, &chunk);
  if (!rc && (flags & HP_RASTER_MSAA4) && kSamplesMsaa * w <= kBandKeysMsaa)
    rc = raster_scratch(store, n_views, raster_bands(h, w, 1), nullptr, &chunk);
  return rc;
}

extern "C" int64_t hp_mesh_store_scratch_generation(const hp_mesh_store* store) { return store ? store->scratch_generation : -1; }

namespace hp {
static const hp_raster_conventions kDefaultConventions = {{0.375f, 0.875f, 0.125f, 0.625f}, {0.125f, 0.375f, 0.625f, 0.875f}, 16, 0, 0, 0.0f, 0.0f,
                                                          {0, 1, 2}, {1.0f, -1.0f, -1.0f}};
static hp_raster_conventions g_conventions = kDefaultConventions;  // process-wide, read at launch time
static std::mutex g_conventions_mutex;

static RasterConv derive_conventions() {
  hp_raster_conventions c;
  { std::lock_guard<std::mutex> lock(g_conventions_mutex); c = g_conventions; }
  RasterConv r{};
  r.lo_x = r.lo_y = 1.0f; r.hi_x = r.hi_y = 0.0f;
  for (int k = 0; k < 4; ++k) {
    r.sx[k] = c.msaa_x[k]; r.sy[k] = c.msaa_y[k];
    r.lo_x = std::fmin(r.lo_x, c.msaa_x[k]); r.hi_x = std::fmax(r.hi_x, c.msaa_x[k]);
    r.lo_y = std::fmin(r.lo_y, c.msaa_y[k]); r.hi_y = std::fmax(r.hi_y, c.msaa_y[k]);
    r.dxa[k] = std::fabs(c.msaa_x[k] - 0.5f); r.dya[k] = std::fabs(c.msaa_y[k] - 0.5f);
  }
  r.sx[4] = r.sy[4] = 0.5f;
  r.aniso_max = (float)c.aniso_max; r.lod_bias = c.lod_bias; r.ratio_bias = c.aniso_ratio_bias; r.aniso_round = c.aniso_round; r.lod_from = c.lod_from;
  for (int k = 0; k < 3; ++k) { r.n_axis[k] = c.normal_axis[k]; r.n_sign[k] = c.normal_sign[k]; }
  return r;
}

// Common launch path of hp_rasterize (strided planes) and hp_render_inputs (network-input records + fused crop).
static int launch_raster(const hp_mesh_store* store, RasterArgs a, int n, bool crop, hipStream_t st) {
  const int h = a.h, w = a.w;
  a.cv = derive_conventions();
  a.msaa = (a.flags & HP_RASTER_MSAA4) && (a.rec || a.rgb || a.nrm) ? 1 : 0;  // depth-only renders have nothing to multisample
  const int ns = a.msaa ? kSamplesMsaa : 1;
  static const int rows_env = std::getenv("HP_RASTER_ROWS") ? std::atoi(std::getenv("HP_RASTER_ROWS")) : 0;
  const int budget = a.msaa ? kBandKeysMsaa / kSamplesMsaa : kBandPixels;  // pixels of a band
  HP_REQUIRE(w <= budget, "hp_rasterize: image too wide for one band");
  a.band_rows = rows_env > 0 && !a.msaa && rows_env * w <= 2 * kBandPixels ? rows_env : budget / w;
  if (a.band_rows > h) a.band_rows = h;
  a.n_bands = (h + a.band_rows - 1) / a.band_rows;
  HP_REQUIRE(a.n_bands <= kMaxBands, "hp_rasterize: unsupported resolution (too many bands)");
  const int npix_max = a.band_rows * w;
  HP_REQUIRE(npix_max < 65536, "hp_rasterize: band too large");
  HP_REQUIRE(!a.msaa || npix_max < 16384, "hp_rasterize: multisampled band too large");  // invocation entries: pixel | sample << 14
  a.w_magic = (unsigned)(0x100000000ull / (unsigned)w + 1);
  a.depth_max = kZNear / (1.0f - (1.0f - 1e-3f) * (kZFar - kZNear) / kZFar);
  // Views are processed in chunks so that the per-(view, band) triangle lists stay within a fixed scratch budget.  The
  // scratch is owned by the store and only ever grows (hp_mesh_store_reserve_raster pre-sizes it; predictors do that at
  // construction for their largest batch); a growth bumps the store's scratch generation, which tells holders of
  // captured hipGraphs that the pointers their launches carry are gone.  Callers sharing one store must be stream-ordered.
  hp::MeshStore* ms = const_cast<hp_mesh_store*>(store);
  a.max_faces = (int)store->max_faces;
  a.bin_cap = a.max_faces;
  a.max_verts = (int)store->max_verts;
  int chunk = 0;
  {
    const int rc = raster_scratch(ms, n, a.n_bands, st, &chunk);
    if (rc) return rc;
  }
  a.bin_list = ms->bin_list;
  a.bin_count = ms->bin_count;
  a.xverts = ms->xverts;
  const size_t lds = band_lds_bytes(npix_max, ns, a.band_rows, w, crop);
  HP_REQUIRE(lds <= 150 * 1024, "hp_rasterize: band does not fit the LDS");
  const bool half = (a.flags & HP_RASTER_OUT_F16) != 0, aniso = (a.flags & HP_RASTER_TEX_ANISO) != 0;
  typedef void (*BandKernel)(RasterArgs, int);
  static const BandKernel kernels[8] = {
      raster_kernel<1, false, false>, raster_kernel<1, false, true>, raster_kernel<1, true, false>, raster_kernel<1, true, true>,
      raster_kernel<kSamplesMsaa, false, false>, raster_kernel<kSamplesMsaa, false, true>, raster_kernel<kSamplesMsaa, true, false>,
      raster_kernel<kSamplesMsaa, true, true>};
  const int ki = 4 * a.msaa + 2 * (half ? 1 : 0) + (aniso ? 1 : 0);
  static size_t opted[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (opted[ki] < lds) {
    HP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernels[ki]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    opted[ki] = lds;
  }
  for (int v0 = 0; v0 < n; v0 += chunk) {
    const int nv = n - v0 < chunk ? n - v0 : chunk;
    a.view0 = v0;
    a.n = nv;
    hipLaunchKernelGGL(raster_xform_kernel, dim3((a.max_verts + 255) / 256, nv), dim3(256), 0, st, a);
    hipLaunchKernelGGL(raster_bin_kernel, dim3((a.max_faces + kBinThreads - 1) / kBinThreads, nv), dim3(kBinThreads), 0, st, a);
    const int total = nv * a.n_bands;
    hipLaunchKernelGGL(kernels[ki], dim3(8 * ((total + 7) / 8)), dim3(band_threads(ns)), lds, st, a, npix_max);
    static const bool chunk_sync = std::getenv("HP_RASTER_CHUNK_SYNC") != nullptr;  // diagnostics (see raster_scratch)
    if (chunk_sync && v0 + chunk < n) (void)hipStreamSynchronize(st);
  }
  return check_launch("raster_kernel");
}
}  // namespace hp

extern "C" int hp_raster_set_conventions(const hp_raster_conventions* c) {
  using namespace hp;
  hp_raster_conventions v = c ? *c : kDefaultConventions;
  for (int k = 0; k < 4; ++k)
    HP_REQUIRE(v.msaa_x[k] > 0.0f && v.msaa_x[k] < 1.0f && v.msaa_y[k] > 0.0f && v.msaa_y[k] < 1.0f,
               "hp_raster_set_conventions: sample positions must lie inside the pixel, in (0, 1)");
  HP_REQUIRE(v.aniso_max >= 1 && v.aniso_max <= 16, "hp_raster_set_conventions: aniso_max must be in 1..16");
  HP_REQUIRE(v.aniso_round >= 0 && v.aniso_round <= 2 && v.lod_from >= 0 && v.lod_from <= 2, "hp_raster_set_conventions: unknown rule");
  HP_REQUIRE(std::isfinite(v.lod_bias) && std::isfinite(v.aniso_ratio_bias), "hp_raster_set_conventions: biases must be finite");
  for (int k = 0; k < 3; ++k)
    HP_REQUIRE(v.normal_axis[k] >= 0 && v.normal_axis[k] <= 2 && (v.normal_sign[k] == 1.0f || v.normal_sign[k] == -1.0f),
               "hp_raster_set_conventions: normal_axis in 0..2, normal_sign +1 / -1");
  std::lock_guard<std::mutex> lock(g_conventions_mutex);
  g_conventions = v;
  return HP_OK;
}

extern "C" int hp_raster_set_backface_culling(int on) {
  const int prev = raster_cull_enabled() ? 1 : 0;
  g_cull.store(on ? 1 : 0, std::memory_order_relaxed);
  return prev;
}

extern "C" int hp_raster_get_conventions(hp_raster_conventions* out) {
  using namespace hp;
  HP_REQUIRE(out != nullptr, "hp_raster_get_conventions: null argument");
  std::lock_guard<std::mutex> lock(g_conventions_mutex);
  *out = g_conventions;
  return HP_OK;
}

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
  a.faces = store->faces; a.tex = store->tex; a.obj = store->obj; a.cull = raster_cull_enabled() ? store->cull : nullptr; a.face_planes = store->face_planes;
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
  a.faces = store->faces; a.tex = store->tex; a.obj = store->obj; a.cull = raster_cull_enabled() ? store->cull : nullptr; a.face_planes = store->face_planes;
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

#ifdef HP_RABL_COUNT
extern "C" int hp_debug_raster_counters(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hp::hp_dbg_cnt), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(hp::hp_dbg_cnt), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
