// Shared helpers of the happypose_amd HIP library (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <mutex>
#include <string>

#include "../../include/happypose_amd.h"
#include "debug.h"

namespace hp {

void set_error(const std::string& msg);

inline int fail(int code, const std::string& msg) {
  set_error(msg);
  return code;
}

#define HP_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      return ::hp::fail(HP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    }                                                                                   \
  } while (0)

#define HP_REQUIRE(cond, msg)                               \
  do {                                                      \
    if (!(cond)) return ::hp::fail(HP_ERR_ARG, (msg));      \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(HP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  return HP_OK;
}

// First-launch set-up of ONE kernel instantiation (dynamic-LDS opt-in, scratch query): `static FirstLaunch fl;` in its launcher,
// `fl.once([&](FirstLaunch& s) { ...; return HP_OK; })` runs the set-up exactly once per process whichever host thread launches
// first -- concurrent first launches on distinct streams wait for it instead of racing on a plain flag -- and
// hands every caller its result code.
struct FirstLaunch {
  std::once_flag flag;
  int rc = HP_OK;
  bool spills = false;  // the kernel uses scratch (count_scratch_launch)
  template <class F>
  int once(F&& f) {
    std::call_once(flag, [&] { rc = f(*this); });
    return rc;
  }
};

// device-resident object set (see include/happypose_amd.h)
struct MeshStore {
  float* verts = nullptr;
  float* normals = nullptr;
  float4* verts4 = nullptr;    // the same, padded to 16 B (one gather per vertex in the rasteriser)
  float4* normals4 = nullptr;
  float* uvs = nullptr;
  uint8_t* colors = nullptr;
  uint8_t* tex = nullptr;
  // the power-of-two textures once more in ROW-PAIR layout for the anisotropic filter (api.cpp: build_tex_quads): per level,
  // for every row y an array of w + 1 entries {T(x, y), T(x, (y + 1) % h)} (8 B; entry w repeats entry 0) -- the 2 x 2 footprint of a
  // bilinear tap at (x0, y0) is the 16 contiguous bytes at entry x0 of row-pair y0: ONE load instead of four gathers
  uint8_t* tex_quads = nullptr;
  int64_t* tex_quads_off = nullptr;  // [n_obj] byte offset of the object's level 0 in tex_quads, -1: none (no texture / not power-of-two)
  int64_t* obj = nullptr;  // [n_obj][8]
  float* points = nullptr; // [n_obj][n_pad][3]
  // [n_obj][8] view-level back-face culling record of the rasteriser's set-up pass: bounding sphere (cx, cy, cz, r) in the vertices'
  // units, entry 4: 1 when at least one face of the object may be culled (api.cpp: mesh_cull_flags), 3 unused
  float* cull = nullptr;
  // [faces of all objects] {i0, i1, i2, cull flag}: the corner indices (object-local) and the face's orientation flag -- +1 / -1:
  // the face belongs to a CLOSED, consistently oriented connected component whose signed volume is positive / negative (its
  // back side can never be seen from outside the object); 0: never culled
  int4* faces4 = nullptr;
  // rasteriser scratch (grown on demand, see raster.hip): per-(view, band) triangle lists, their counters (all zero between
  // launches: the band kernel clears what it has consumed) and the per-(view, sub-triangle) set-up records
  int32_t* bin_list = nullptr;
  size_t bin_list_bytes = 0;
  int32_t* bin_count = nullptr;
  size_t bin_count_bytes = 0;
  int4* xverts = nullptr;     // [view][max_verts] {x, y (snapped, 1/256 px), bits(1 / z), bits(z)} of the vertex pre-pass
  size_t xverts_bytes = 0;
  uint4* recs = nullptr;      // [view][2 * max_faces][8]: 128-B set-up record of every sub-triangle (raster.hip: FaceRec)
  size_t recs_bytes = 0;
  int64_t scratch_generation = 0;  // bumped whenever the scratch above is reallocated (captured graphs hold the old pointers)
  // renderer state of THIS store (no process-wide state: two stores may differ): the conventions record of
  // hp_mesh_store_set_raster_conventions and the back-face culling switch of hp_mesh_store_set_backface_culling
  hp_raster_conventions conventions;
  int backface_culling = 1;
  bool any_untextured = false;  // some object has no texture: its vertex colours are interpolated (needs the barycentric planes)
  int n_obj = 0;
  int n_pad = 0;
  int64_t max_verts = 0;  // max vertices of a single object
  int64_t max_faces = 0;
};

void raster_store_defaults(MeshStore* s);  // raster.hip: default conventions record, culling on

}  // namespace hp

struct hp_mesh_store : hp::MeshStore {};
