// lf_collada.h -- COLLADA (.dae) -> flat scene for the device path.
//
// Replaces, for the scene term and find_sun_pos, what the reference does between the file and its
// BVH primitives / light list:
//   Collada::ColladaParser::load            src/scene/collada/collada.cpp:131-218 (+ :232-954)
//   Application::load / init_*              src/application/application.cpp:232-365
//   GLScene::Mesh, HalfedgeMesh::build      src/scene/gl_scene/mesh.cpp:21-45, src/util/halfEdgeMesh.cpp:28-404
//   SceneObjects::Mesh (triangle order)     src/scene/object.cpp:14-45
//   GLScene::*Light::get_static_light       src/scene/gl_scene/{directional,point,area,spot,ambient}_light.h
//   SceneObjects::DirectionalLight          src/scene/light.cpp:11-16 (the "position" of the sun)
// The output is bit-for-bit what those classes hold (tests/test_collada_loader.py compares it with
// dumps of the real reference, oracle/ref_driver.cpp `collada`), quirks included:
//   * 4x4 products are A^T B (the reference's AVX build dots two *columns*, CGL/src/matrix4x4.cpp:131-133);
//   * a mesh triangle is (p[d-1], p[0], p[1]) of its polygon (the face keeps its last half-edge);
//   * vertex normals are Vertex::computeNormal's area-weighted sums, whose boundary branch walks
//     h->next()->twin() (src/util/halfEdgeMesh.h:492-515);
//   * the i-th smallest vertex index gets the i-th position of the source array.
// Refused with an error instead of being guessed: <rotate>/<translate>/<scale> node transforms (the
// reference multiplies an uninitialised Matrix4x4 there, collada.cpp:262-318), non-manifold or
// inconsistently oriented meshes and everything else the reference answers with exit().
#pragma once

#include <string>
#include <vector>

namespace lfamd {

struct ColladaVec3 { double x = 0, y = 0, z = 0; };

enum ColladaBsdf { BSDF_DIFFUSE = 0, BSDF_EMISSION, BSDF_MIRROR, BSDF_GLASS, BSDF_REFRACTION,
                   BSDF_MICROFACET };
struct ColladaMaterial { int kind = BSDF_DIFFUSE; ColladaVec3 rgb; };

enum ColladaLightType { LIGHT_DIRECTIONAL = 0, LIGHT_POINT, LIGHT_AREA, LIGHT_HEMISPHERE, LIGHT_SPOT };
struct ColladaLight {
  int type = LIGHT_POINT;
  ColladaVec3 radiance;
  ColladaVec3 position;       // point / area / spot; directional: posLight (what find_sun_pos projects)
  ColladaVec3 direction;      // directional: dirToLight; area / spot: direction
  ColladaVec3 dim_x, dim_y;   // area
};

struct ColladaCamera {
  bool present = false;
  double hFov = 0, vFov = 0, nClip = 0, fClip = 0;   // CameraInfo's floats, widened
  ColladaVec3 pos, dir, up;                          // c_pos, c_dir of Application::load; up_dir
};

struct ColladaSphere { ColladaVec3 o; double r = 0; int material = 0; };
struct ColladaTriangle { ColladaVec3 p[3], n[3]; int material = 0; };

// one entry per scene node, in the order the reference's SceneInfo::nodes holds them
struct ColladaItem { enum Kind { CAMERA, LIGHT, SPHERE, MESH } kind; int index; int count; };

struct ColladaScene {
  ColladaCamera camera;                       // what Application::load ends up with (the last camera node;
                                              // its position is transformed from the previous node's)
  std::vector<ColladaCamera> camera_nodes;    // the state after each camera node, in node order
  std::vector<ColladaLight> lights;
  std::vector<ColladaSphere> spheres;
  std::vector<ColladaTriangle> triangles;
  std::vector<ColladaMaterial> materials;
  std::vector<ColladaItem> items;   // MESH: index = first triangle, count = triangles
};

// Returns false and fills `error` when the file cannot be loaded the way the reference would.
bool load_collada(const std::string& path, ColladaScene& out, std::string& error);

// The dump format of `ref_dump collada` (hex floats), for the parity test.
std::string dump_collada(const ColladaScene& scene);

}  // namespace lfamd
