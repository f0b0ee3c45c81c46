// lf_scene_types.h -- device-side layout of the static scene behind the scene-radiance term
// (lf_scene.hip).  Internal: nothing here is part of the ABI.
#pragma once

// One node of the device BVH = the boxes of its TWO children, 64 bytes (four 16-byte loads per
// visit, and a visit decides about two subtrees).  The boxes are FLOATS rounded outward from the
// double boxes of the primitives: the box test only culls -- every primitive of a leaf that is
// reached still runs the reference's double-precision test (scene/sphere.cpp:11-111,
// scene/triangle.cpp:25-112) -- so it may be conservative but never lossy.
// child >= 0: index of another node.  child < 0: a leaf of 1 .. 4 primitives,
// ~child = first * 4 + (count - 1).  kLfNoChild: nothing (a scene of fewer than two leaves).
constexpr int kLfNoChild = (int)0x80000000u;
struct alignas(64) LfBvhNode {
  float lo0[3], hi0[3], lo1[3], hi1[3];
  int child[2];
  int pad[2];
};
// what the intersection tests read: a sphere's centre, r, r^2, or a triangle's p0, p1 - p0, p2 - p0
struct alignas(16) LfPrim { double d[9]; int type, material; };  // type 0 sphere, 1 triangle
// a triangle's three vertex normals: read once per camera ray, for the closest hit only
struct LfPrimNormals { double n[9]; };
struct LfMaterial { int kind, pad; double rgb[3]; };   // 0 diffuse (reflectance), 1 emission (radiance)
// 0 directional (v = dirToLight), 1 point (v = position), 2 infinite hemisphere, 3 area (v = position,
// dir, dim_x, dim_y, area = |dim_x| |dim_y|: scene/light.h:80-97)
struct LfLight { int type, pad; double v[3], rgb[3], dir[3], dim_x[3], dim_y[3], area; };
struct LfSceneDev {
  LfBvhNode* nodes; LfPrim* prims; LfPrimNormals* normals; LfMaterial* materials; LfLight* lights;
  int n_nodes, n_prims, n_materials, n_lights;
  int n_soft_lights;   // lights that are sampled (hemisphere, area, environment): they need the counter RNG
  int n_env_lights;    // lights of type 4 (they need lf_set_environment_map)
};
// EnvironmentLight (scene/environment_light.cpp): the map (HDRImageBuffer::data, w*h RGB doubles)
// and the tables its init() derives (:19-59), built on the host in the reference's order of
// operations; w = 0: no environment
struct LfEnvDev {
  const double* data; const double* pdf; const double* conds; const double* marginal;
  int w, h;
};
