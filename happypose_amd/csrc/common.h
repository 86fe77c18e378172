// Shared helpers of the happypose_amd HIP library (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <string>

#include "../../include/happypose_amd.h"

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

// device-resident object set (see include/happypose_amd.h)
struct MeshStore {
  float* verts = nullptr;
  float* normals = nullptr;
  float4* verts4 = nullptr;    // the same, padded to 16 B (one gather per vertex in the rasteriser)
  float4* normals4 = nullptr;
  float* uvs = nullptr;
  uint8_t* colors = nullptr;
  int32_t* faces = nullptr;
  uint8_t* tex = nullptr;
  int64_t* obj = nullptr;  // [n_obj][8]
  float* points = nullptr; // [n_obj][n_pad][3]
  // [n_obj][8] back-face culling record of the rasteriser's binning pass: bounding sphere (cx, cy, cz, r) in the vertices' units,
  // orientation sign (+1 / -1: the object is a closed, consistently oriented surface with positive / negative signed volume;
  // 0: not provably closed -> never culled), 3 unused (api.cpp: mesh_cull_record)
  float* cull = nullptr;
  float4* face_planes = nullptr;  // [faces of all objects] (n, n . a) of every face, n = (b - a) x (c - a) in the vertices' units: the facing test
  // rasteriser scratch: per-(view, band) triangle lists (grown on demand, see raster.hip)
  int32_t* bin_list = nullptr;
  size_t bin_list_bytes = 0;
  int32_t* bin_count = nullptr;
  size_t bin_count_bytes = 0;
  float4* xverts = nullptr;  // per-(view, vertex) screen-space vertices of the current chunk (2 float4 each)
  size_t xverts_bytes = 0;
  int64_t scratch_generation = 0;  // bumped whenever the scratch above is reallocated (captured graphs hold the old pointers)
  int n_obj = 0;
  int n_pad = 0;
  int64_t max_verts = 0;  // max vertices of a single object
  int64_t max_faces = 0;
};

}  // namespace hp

struct hp_mesh_store : hp::MeshStore {};
