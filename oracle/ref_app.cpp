// TEST INFRASTRUCTURE -- not shipped, not linked into the product.
//
// The reference's own render controller, end to end and headless: this driver does what
// src/application/main.cpp:160-196 + Application::load / set_up_pathtracer (application.cpp:232-308,
// :649-655) do for `pathtracer -f out.png ...` -- parse a COLLADA file with the reference's parser,
// build the GLScene objects, convert them to the static scene, configure the Camera, construct a
// **RaytracedRenderer** with the AppConfig defaults (application.h:47-69) -- and then calls the
// reference's RaytracedRenderer::render_to_file: start_raytracing, the tile queue, N worker threads
// calling PathTracer::raytrace_pixel / write_to_framebuffer, save_image with lodepng
// (src/pathtracer/raytraced_renderer.cpp:287-374, :622-755), all compiled from the reference's
// sources where they lie (oracle/Makefile target `app`).  Application itself is not used: it needs
// GLU / Freetype headers the image lacks, and nothing on this path lives there.
//
// Two binaries come out of this one file:
//   oracle/_ref/ref_app      every object the reference's          -> the golden PNG
//   oracle/_ref/ref_app_amd  the same, pathtracer.o replaced by lens-flare_amd/host/pathtracer_amd.cpp
// tests/test_gpu_dropin.py runs the second on the GPU box and compares the PNG it writes, byte for
// byte, with what the first wrote here (tests/golden/app_*.png; generator oracle/make_golden_app.py).
//
//   ref_app <scene.dae> <camera file> <W> <H> <ns_aa> <threads> <aperture.png> <ghost aperture.png>
//           <flare radius> <flare intensity> <out.png> [<autofocus x> <autofocus y>]
// After the render, with the two optional arguments, RaytracedRenderer::autofocus(x, y)
// (raytraced_renderer.cpp:677-679) is called like the UI would and the camera's focal distance is
// printed as a hex float ("FOCAL <value>").
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "pathtracer/raytraced_renderer.h"
#include "pathtracer/camera.h"
#include "scene/collada/collada.h"
#include "scene/gl_scene/ambient_light.h"
#include "scene/gl_scene/area_light.h"
#include "scene/gl_scene/directional_light.h"
#include "scene/gl_scene/mesh.h"
#include "scene/gl_scene/point_light.h"
#include "scene/gl_scene/scene.h"
#include "scene/gl_scene/sphere.h"
#include "scene/gl_scene/spot_light.h"

using namespace CGL;

int main(int argc, char** argv) {
  if (argc < 12) {
    fprintf(stderr, "usage: see the header of oracle/ref_app.cpp\n");
    return 2;
  }
  const std::string dae = argv[1], camfile = argv[2];
  const size_t W = strtoul(argv[3], 0, 10), H = strtoul(argv[4], 0, 10);
  const size_t ns_aa = strtoul(argv[5], 0, 10), threads = strtoul(argv[6], 0, 10);
  const std::string aperture = argv[7], ghost_aperture = argv[8];
  const double flare_radius = atof(argv[9]), flare_intensity = atof(argv[10]);
  const std::string out = argv[11];

  Collada::SceneInfo* info = new Collada::SceneInfo();
  if (Collada::ColladaParser::load(dae.c_str(), info) < 0) { fprintf(stderr, "cannot load %s\n", dae.c_str()); return 1; }

  // the node loop of Application::load
  Camera camera;
  std::vector<GLScene::SceneLight*> lights;
  std::vector<GLScene::SceneObject*> objects;
  for (Collada::Node& node : info->nodes) {
    Collada::Instance* instance = node.instance;
    if (!instance) continue;
    const Matrix4x4& transform = node.transform;
    switch (instance->type) {
      case Collada::Instance::CAMERA:
        camera.configure(*static_cast<Collada::CameraInfo*>(instance), W, H);   // Application::init_camera
        break;
      case Collada::Instance::LIGHT: {                                          // Application::init_light
        Collada::LightInfo& li = static_cast<Collada::LightInfo&>(*instance);
        switch (li.light_type) {
          case Collada::LightType::AMBIENT: lights.push_back(new GLScene::AmbientLight(li)); break;
          case Collada::LightType::DIRECTIONAL: lights.push_back(new GLScene::DirectionalLight(li, transform)); break;
          case Collada::LightType::AREA: lights.push_back(new GLScene::AreaLight(li, transform)); break;
          case Collada::LightType::POINT: lights.push_back(new GLScene::PointLight(li, transform)); break;
          case Collada::LightType::SPOT: lights.push_back(new GLScene::SpotLight(li, transform)); break;
          default: break;
        }
        break;
      }
      case Collada::Instance::SPHERE: {                                         // Application::init_sphere
        Collada::SphereInfo& si = static_cast<Collada::SphereInfo&>(*instance);
        const Vector3D position = (transform * Vector4D(0, 0, 0, 1)).projectTo3D();
        const double scale = (transform * Vector4D(1, 0, 0, 0)).to3D().norm();
        objects.push_back(new GLScene::Sphere(si, position, scale));
        break;
      }
      case Collada::Instance::POLYMESH:                                         // Application::init_polymesh
        objects.push_back(new GLScene::Mesh(static_cast<Collada::PolymeshInfo&>(*instance), transform));
        break;
      default: break;
    }
  }
  GLScene::Scene* gl_scene = new GLScene::Scene(objects, lights);
  camera.load_settings(camfile);   // the -c flag (main.cpp:192-193 -> Application::load_camera)

  // `new RaytracedRenderer(...)` of Application::Application with AppConfig's defaults
  RaytracedRenderer renderer(ns_aa, /*max_ray_depth*/ 1, /*ns_area_light*/ 1, /*ns_diff*/ 1, /*ns_glsy*/ 1,
                             /*ns_refr*/ 1, threads, /*samples_per_batch*/ 32, /*max_tolerance*/ 0.05f,
                             /*envmap*/ NULL, /*direct_hemisphere_sample*/ false, out, /*lensRadius*/ 0.0,
                             /*focalDistance*/ 4.7, aperture, flare_radius, flare_intensity, ghost_aperture);
  // Application::set_up_pathtracer + render_to_file
  renderer.set_camera(&camera);
  renderer.set_scene(gl_scene->get_static_scene());
  renderer.set_frame_size(W, H);
  renderer.render_to_file(out, (size_t)-1, 0, 0, 0);

  if (argc >= 14) {
    renderer.autofocus(Vector2D(atof(argv[12]), atof(argv[13])));
    printf("FOCAL %a\n", camera.focalDistance);
  }
  fflush(stdout);
  // (the renderer's destructor joins nothing: the workers have finished; leave through _exit-free return)
  renderer.stop();
  return 0;
}
