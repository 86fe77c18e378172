// C-ABI glue: error reporting, device queries and the device-resident mesh store.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"

namespace hp {

static thread_local std::string g_error;

void set_error(const std::string& msg) { g_error = msg; }

}  // namespace hp

using namespace hp;

extern "C" int hp_version(void) { return 100; }

extern "C" const char* hp_last_error(void) { return g_error.c_str(); }

extern "C" int hp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int hp_device_name(char* buf, int len) {
  HP_REQUIRE(buf && len > 0, "hp_device_name: bad buffer");
  int dev = 0;
  HP_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HP_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  std::snprintf(buf, (size_t)len, "%s (%s)", prop.name, prop.gcnArchName);
  return HP_OK;
}

namespace {

template <typename T>
int upload(T** dst, const T* src, size_t count) {
  *dst = nullptr;
  if (count == 0 || src == nullptr) return HP_OK;
  HP_CHECK_HIP(hipMalloc((void**)dst, count * sizeof(T)));
  HP_CHECK_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return HP_OK;
}

// Back-face culling data of one object (MeshStore::cull, the flag in MeshStore::faces4).  The reference renders two-sided
// (TB/renderer/panda3d_scene_renderer.py:102), so a triangle facing away from the camera may only be dropped when it provably
// cannot be seen.  Per CONNECTED COMPONENT of the position-welded mesh (texture seams duplicate vertices; faces are connected
// through shared edges): if the component is a closed, consistently oriented surface (every directed edge occurs once and so
// does its reverse) with a signed volume safely away from zero, then from any camera position outside the component every ray
// meets one of its outward-facing faces before any inward-facing one -- the inward-facing faces of THAT component are never
// visible, whatever the other components do (nested shells, a part with flipped winding, open sheets: each gets its own
// flag).  flag = +1 / -1: outward = the winding's front / back side; 0: never culled.  The rasteriser culls only when the
// camera is outside the object's bounding sphere (hence outside every component) and the sphere lies beyond the near plane.
void mesh_cull_flags(const float* v, int64_t nv, const int32_t* f, int64_t nf, float* out8, int32_t* flags) {
  for (int k = 0; k < 8; ++k) out8[k] = 0.f;
  for (int64_t t = 0; t < nf; ++t) flags[t] = 0;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int64_t i = 0; i < nv; ++i)
    for (int c = 0; c < 3; ++c) { lo[c] = std::min(lo[c], (double)v[3 * i + c]); hi[c] = std::max(hi[c], (double)v[3 * i + c]); }
  const double cx = 0.5 * (lo[0] + hi[0]), cy = 0.5 * (lo[1] + hi[1]), cz = 0.5 * (lo[2] + hi[2]);
  double r2 = 0.0;
  for (int64_t i = 0; i < nv; ++i) {
    const double dx = v[3 * i] - cx, dy = v[3 * i + 1] - cy, dz = v[3 * i + 2] - cz;
    r2 = std::max(r2, dx * dx + dy * dy + dz * dz);
  }
  out8[0] = (float)cx; out8[1] = (float)cy; out8[2] = (float)cz; out8[3] = (float)(std::sqrt(r2) * 1.0001 + 1e-7);
  // weld by exact position
  std::vector<int32_t> order((size_t)nv), canon((size_t)nv);
  for (int64_t i = 0; i < nv; ++i) order[i] = (int32_t)i;
  auto less = [&](int32_t a, int32_t b) { return std::memcmp(v + 3 * (int64_t)a, v + 3 * (int64_t)b, 12) < 0; };
  std::sort(order.begin(), order.end(), less);
  for (int64_t i = 0; i < nv; ++i)
    canon[order[i]] = (i > 0 && std::memcmp(v + 3 * (int64_t)order[i], v + 3 * (int64_t)order[i - 1], 12) == 0) ? canon[order[i - 1]] : order[i];
  // union-find over the faces: two faces sharing an (undirected) welded edge are connected
  std::vector<int32_t> parent((size_t)nf);
  for (int64_t t = 0; t < nf; ++t) parent[t] = (int32_t)t;
  auto find = [&](int32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
  struct Edge { uint64_t key; int32_t face; };
  std::vector<Edge> directed, undirected;
  directed.reserve((size_t)nf * 3); undirected.reserve((size_t)nf * 3);
  std::vector<char> degenerate((size_t)nf, 0);
  std::vector<double> fvol((size_t)nf, 0.0);
  for (int64_t t = 0; t < nf; ++t) {
    const int32_t c3[3] = {canon[f[3 * t]], canon[f[3 * t + 1]], canon[f[3 * t + 2]]};
    if (c3[0] == c3[1] || c3[1] == c3[2] || c3[0] == c3[2]) { degenerate[t] = 1; continue; }  // no area, no edges
    const float* pa = v + 3 * (int64_t)c3[0]; const float* pb = v + 3 * (int64_t)c3[1]; const float* pc = v + 3 * (int64_t)c3[2];
    fvol[t] = (double)pa[0] * ((double)pb[1] * pc[2] - (double)pb[2] * pc[1]) - (double)pa[1] * ((double)pb[0] * pc[2] - (double)pb[2] * pc[0]) +
              (double)pa[2] * ((double)pb[0] * pc[1] - (double)pb[1] * pc[0]);
    for (int e = 0; e < 3; ++e) {
      const uint32_t a = (uint32_t)c3[e], b = (uint32_t)c3[(e + 1) % 3];
      directed.push_back({((uint64_t)a << 32) | b, (int32_t)t});
      undirected.push_back({((uint64_t)std::min(a, b) << 32) | std::max(a, b), (int32_t)t});
    }
  }
  if (directed.empty()) return;
  auto by_key = [](const Edge& x, const Edge& y) { return x.key < y.key; };
  std::sort(undirected.begin(), undirected.end(), by_key);
  for (size_t i = 1; i < undirected.size(); ++i)
    if (undirected[i].key == undirected[i - 1].key) {
      const int32_t ra = find(undirected[i].face), rb = find(undirected[i - 1].face);
      if (ra != rb) parent[ra] = rb;
    }
  std::sort(directed.begin(), directed.end(), by_key);
  std::vector<char> bad((size_t)nf, 0);  // per component root
  for (size_t i = 0; i < directed.size(); ++i) {
    if (i > 0 && directed[i].key == directed[i - 1].key) bad[find(directed[i].face)] = 1;  // a directed edge used twice
    const uint64_t rev = (directed[i].key << 32) | (directed[i].key >> 32);
    const Edge probe{rev, 0};
    if (!std::binary_search(directed.begin(), directed.end(), probe, by_key)) bad[find(directed[i].face)] = 1;  // boundary edge
  }
  std::vector<double> vol((size_t)nf, 0.0), mag((size_t)nf, 0.0);
  for (int64_t t = 0; t < nf; ++t)
    if (!degenerate[t]) { const int32_t r = find((int32_t)t); vol[r] += fvol[t]; mag[r] += std::fabs(fvol[t]); }
  bool any = false;
  for (int64_t t = 0; t < nf; ++t) {
    if (degenerate[t]) continue;
    const int32_t r = find((int32_t)t);
    if (bad[r] || !(std::fabs(vol[r]) > 1e-9 * mag[r]) || mag[r] == 0.0) continue;
    flags[t] = vol[r] > 0.0 ? 1 : -1;
    any = true;
  }
  out8[4] = any ? 1.f : 0.f;
}

// Row-pair copy of one power-of-two texture (level 0 + its mip chain) for the anisotropic filter -- see MeshStore::tex_quads.
// Returns the bytes appended.
size_t build_tex_quads(const uint8_t* tex, int64_t w, int64_t h, int64_t nlev, std::vector<uint8_t>& out) {
  const size_t start = out.size();
  for (int64_t k = 0; k < nlev; ++k) {
    const size_t base = out.size();
    out.resize(base + (size_t)8 * (size_t)(w + 1) * (size_t)h);
    uint8_t* dst = out.data() + base;
    for (int64_t y = 0; y < h; ++y) {
      const uint8_t* r0 = tex + 4 * y * w;
      const uint8_t* r1 = tex + 4 * ((y + 1) % h) * w;
      for (int64_t x = 0; x <= w; ++x) {
        const int64_t xs = x == w ? 0 : x;
        std::memcpy(dst + 8 * (y * (w + 1) + x), r0 + 4 * xs, 4);
        std::memcpy(dst + 8 * (y * (w + 1) + x) + 4, r1 + 4 * xs, 4);
      }
    }
    tex += 4 * w * h;
    w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1;
  }
  return out.size() - start;
}

}  // namespace

extern "C" hp_mesh_store* hp_mesh_store_create(const float* h_verts, const float* h_normals,
                                               const float* h_uvs, const uint8_t* h_colors,
                                               int64_t n_verts_total, const int32_t* h_faces,
                                               int64_t n_faces_total, const uint8_t* h_tex,
                                               int64_t tex_bytes, const int64_t* h_obj, int n_obj,
                                               const float* h_points, int n_pad) {
  if (!h_verts || !h_normals || !h_uvs || !h_colors || !h_faces || !h_obj || n_obj <= 0 ||
      n_verts_total <= 0 || n_faces_total <= 0) {
    set_error("hp_mesh_store_create: null or empty geometry");
    return nullptr;
  }
  // validate the descriptor table: kernels index with it unchecked
  for (int o = 0; o < n_obj; ++o) {
    const int64_t* r = h_obj + 8 * o;
    int64_t chain = 0;  // bytes of level 0 and the mip levels stored behind it (entry 7 = number of levels, 0 / 1 = level 0 only)
    for (int64_t k = 0, w = r[5], h = r[6]; r[4] >= 0 && k < (r[7] > 0 ? r[7] : 1) && w > 0 && h > 0; ++k) {
      chain += 4 * w * h;
      w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1;
    }
    const bool tex_ok = r[4] < 0 || (r[5] > 0 && r[6] > 0 && r[7] >= 0 && r[7] <= 32 && r[4] + chain <= tex_bytes && h_tex);
    if (r[0] < 0 || r[1] <= 0 || r[0] + r[1] > n_verts_total || r[2] < 0 || r[3] <= 0 ||
        r[2] + r[3] > n_faces_total || !tex_ok) {
      set_error("hp_mesh_store_create: object descriptor " + std::to_string(o) + " out of range");
      return nullptr;
    }
    for (int64_t f = 3 * r[2]; f < 3 * (r[2] + r[3]); ++f)
      if (h_faces[f] < 0 || h_faces[f] >= r[1]) {
        set_error("hp_mesh_store_create: face index out of range in object " + std::to_string(o));
        return nullptr;
      }
  }
  hp_mesh_store* s = new hp_mesh_store();
  s->n_obj = n_obj;
  hp::raster_store_defaults(s);
  s->n_pad = h_points ? n_pad : 0;
  for (int o = 0; o < n_obj; ++o) {
    if (h_obj[8 * o + 1] > s->max_verts) s->max_verts = h_obj[8 * o + 1];
    if (h_obj[8 * o + 3] > s->max_faces) s->max_faces = h_obj[8 * o + 3];
    if (h_obj[8 * o + 4] < 0) s->any_untextured = true;
  }
  int rc = 0;
  rc |= upload(&s->verts, h_verts, (size_t)n_verts_total * 3);
  rc |= upload(&s->normals, h_normals, (size_t)n_verts_total * 3);
  {
    std::vector<float4> v4((size_t)n_verts_total), n4((size_t)n_verts_total);
    for (int64_t i = 0; i < n_verts_total; ++i) {
      v4[i] = make_float4(h_verts[3 * i], h_verts[3 * i + 1], h_verts[3 * i + 2], 0.f);
      n4[i] = make_float4(h_normals[3 * i], h_normals[3 * i + 1], h_normals[3 * i + 2], 0.f);
    }
    rc |= upload(&s->verts4, v4.data(), v4.size());
    rc |= upload(&s->normals4, n4.data(), n4.size());
  }
  rc |= upload(&s->uvs, h_uvs, (size_t)n_verts_total * 2);
  rc |= upload(&s->colors, h_colors, (size_t)n_verts_total * 4);
  rc |= upload(&s->tex, h_tex, (size_t)(tex_bytes > 0 ? tex_bytes : 0));
  rc |= upload(&s->obj, h_obj, (size_t)n_obj * 8);
  {
    std::vector<uint8_t> quads;
    std::vector<int64_t> qoff((size_t)n_obj, -1);
    for (int o = 0; o < n_obj; ++o) {
      const int64_t* r = h_obj + 8 * o;
      if (r[4] < 0 || (r[5] & (r[5] - 1)) != 0 || (r[6] & (r[6] - 1)) != 0) continue;
      qoff[o] = (int64_t)quads.size();
      build_tex_quads(h_tex + r[4], r[5], r[6], r[7] > 0 ? r[7] : 1, quads);
    }
    quads.resize(quads.size() + 16);  // the last entry's 16-B load stays inside the allocation
    rc |= upload(&s->tex_quads, quads.data(), quads.size());
    rc |= upload(&s->tex_quads_off, qoff.data(), qoff.size());
  }
  {
    std::vector<float> cull((size_t)n_obj * 8);
    std::vector<int32_t> flags((size_t)n_faces_total);
    std::vector<int4> f4((size_t)n_faces_total);
    for (int o = 0; o < n_obj; ++o)
      mesh_cull_flags(h_verts + 3 * h_obj[8 * o], h_obj[8 * o + 1], h_faces + 3 * h_obj[8 * o + 2], h_obj[8 * o + 3], cull.data() + 8 * o,
                      flags.data() + h_obj[8 * o + 2]);
    for (int64_t t = 0; t < n_faces_total; ++t) f4[t] = make_int4(h_faces[3 * t], h_faces[3 * t + 1], h_faces[3 * t + 2], flags[t]);
    rc |= upload(&s->cull, cull.data(), cull.size());
    rc |= upload(&s->faces4, f4.data(), f4.size());
  }
  if (h_points) rc |= upload(&s->points, h_points, (size_t)n_obj * n_pad * 3);
  if (rc) {
    hp_mesh_store_destroy(s);
    return nullptr;
  }
  return s;
}

extern "C" void hp_mesh_store_destroy(hp_mesh_store* s) {
  if (!s) return;
  (void)hipFree(s->verts); (void)hipFree(s->normals); (void)hipFree(s->uvs); (void)hipFree(s->colors);
  (void)hipFree(s->faces4); (void)hipFree(s->tex); (void)hipFree(s->obj); (void)hipFree(s->points); (void)hipFree(s->cull);
  (void)hipFree(s->bin_list); (void)hipFree(s->bin_count); (void)hipFree(s->recs); (void)hipFree(s->xverts);
  (void)hipFree(s->tex_quads); (void)hipFree(s->tex_quads_off);
  (void)hipFree(s->verts4); (void)hipFree(s->normals4);
  delete s;
}

extern "C" const float* hp_mesh_store_points(const hp_mesh_store* s) { return s ? s->points : nullptr; }
