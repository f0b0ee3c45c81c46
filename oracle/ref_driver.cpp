// TEST INFRASTRUCTURE -- not shipped, not linked into the product.
//
// Object-level driver for the *real* reference hot path.  It is compiled by
// oracle/Makefile (target `ref`) against the reference's own translation units
// where they lie under /root/reference (nothing from the reference is copied
// into this repository); the resulting binary goes to oracle/_ref/ref_dump.
// It fills the public fields of the reference's PathTracer the way
// RaytracedRenderer::start_raytracing does (src/pathtracer/raytraced_renderer.cpp:300-311)
// and visits pixels the way raytrace_tile does (:324-328, :637-641), then dumps raw
// buffers that oracle/make_golden.py turns into the fixtures under tests/golden/.
//
// Sub-commands (all output is raw little-endian binary or hex-float text):
//   ref_dump aperture <png> <out.f32>
//   ref_dump trace <out.txt>
//   ref_dump convert <out.txt>
//   ref_dump frame <camfile> <W> <H> <ns_aa> <flare_radius> <flare_intensity>
//            <aperture.png> <ghost.png> <lights: lx,ly,lz,Lr,Lg,Lb[;...]>
//            <visit: tiles | list:<file>> <outprefix>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "pathtracer/pathtracer.h"
#include "pathtracer/camera.h"
#include "pathtracer/bsdf.h"
#include "scene/light.h"
#include "scene/object.h"
#include "scene/sphere.h"
#include "scene/triangle.h"
#include "util/halfEdgeMesh.h"
#include "util/image.h"

using namespace CGL;
using namespace CGL::SceneObjects;

// Non-static helpers of the reference (src/pathtracer/pathtracer.cpp:511-689, :901-945).
namespace CGL {
Vector2D trace_ray_auto_before(float r, float theta, int i, int j, std::vector<Matrix3x3> color_R);
Vector2D trace_ray_auto_after(float r, float theta, int i, int j, std::vector<Matrix3x3> color_R);
extern std::vector<Matrix3x3> R_red, R_green, R_blue, Ls;
double convertCoordinate(size_t pixel_coord, int length, bool y);
}

static void write_raw(const std::string& path, const void* p, size_t bytes) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { perror(path.c_str()); exit(2); }
  fwrite(p, 1, bytes, f);
  fclose(f);
}

static int cmd_aperture(int argc, char** argv) {
  if (argc < 4) return 1;
  CameraApertureTexture tex;
  tex.init(argv[2]);
  write_raw(argv[3], tex.aperture.data(), tex.aperture.size() * sizeof(float));
  fprintf(stderr, "APERTURE %zu %zu %d %d %d %d %a\n", tex.width, tex.height, tex.min_x, tex.min_y,
          tex.max_x, tex.max_y, tex.total_value);
  return 0;
}

static int cmd_trace(int argc, char** argv) {
  if (argc < 3) return 1;
  FILE* f = fopen(argv[2], "w");
  const float thetas[] = {0.05f, 0.1f, 0.4f, 0.78f, -0.3f, 0.7823f, -1.2f};
  std::vector<Matrix3x3>* cols[3] = {&R_red, &R_green, &R_blue};
  for (float th : thetas)
    for (int c = 0; c < 3; c++) {
      for (int i = 0; i < 5; i++)
        for (int j = i + 1; j < 5; j++)
          for (int s = 0; s < 2; s++) {
            float r = s ? -14.5f : 14.5f;
            Vector2D v = trace_ray_auto_before(r, th, i, j, *cols[c]);
            fprintf(f, "before %a %d %d %d %a %a %a\n", th, c, i, j, r, v.x, v.y);
          }
      for (int i = 6; i < 9; i++)
        for (int j = i + 1; j < 9; j++)
          for (int s = 0; s < 2; s++) {
            float r = s ? -14.5f : 14.5f;
            Vector2D v = trace_ray_auto_after(r, th, i, j, *cols[c]);
            fprintf(f, "after %a %d %d %d %a %a %a\n", th, c, i, j, r, v.x, v.y);
          }
    }
  fclose(f);
  return 0;
}

static int cmd_convert(int argc, char** argv) {
  if (argc < 3) return 1;
  FILE* f = fopen(argv[2], "w");
  const int lens[] = {1920, 1080, 256, 97, 65, 3840, 2160};
  for (int len : lens)
    for (int yflag = 0; yflag < 2; yflag++)
      for (size_t p = 0; p < (size_t)len; p += (len > 300 ? 37 : 1))
        fprintf(f, "%d %d %zu %a\n", len, yflag, p, convertCoordinate(p, len, yflag != 0));
  fclose(f);
  return 0;
}

static int cmd_frame(int argc, char** argv) {
  if (argc < 13) return 1;
  int a = 2;
  std::string camfile = argv[a++];
  size_t W = strtoul(argv[a++], 0, 10), H = strtoul(argv[a++], 0, 10);
  size_t ns_aa = strtoul(argv[a++], 0, 10);
  double flare_radius = atof(argv[a++]), flare_intensity = atof(argv[a++]);
  std::string ap_png = argv[a++], gh_png = argv[a++];
  std::string lightspec = argv[a++];
  std::string visit = argv[a++];
  std::string out = argv[a++];

  Camera cam;
  cam.load_settings(camfile);
  cam.aperture_texture = new CameraApertureTexture();
  cam.aperture_texture->init(ap_png);
  cam.ghost_aperture_texture = new CameraApertureTexture();
  cam.ghost_aperture_texture->init(gh_png);

  // one diffuse sphere far behind the camera target so no camera ray hits it (scene term = 0)
  DiffuseBSDF* bsdf = new DiffuseBSDF(Vector3D(0.5, 0.5, 0.5));
  SphereObject* sph = new SphereObject(Vector3D(1e4, 1e4, 1e4), 1.0, bsdf);
  std::vector<SceneObject*> objs{sph};
  // the DirectionalLight ctor negates posLight (src/scene/light.cpp:11-16): pass -pos
  std::vector<SceneLight*> lights;
  {
    size_t p0 = 0;
    while (p0 < lightspec.size()) {
      size_t p1 = lightspec.find(';', p0);
      if (p1 == std::string::npos) p1 = lightspec.size();
      double v[6];
      if (sscanf(lightspec.substr(p0, p1 - p0).c_str(), "%lf,%lf,%lf,%lf,%lf,%lf", &v[0], &v[1],
                 &v[2], &v[3], &v[4], &v[5]) != 6) return 3;
      lights.push_back(new DirectionalLight(Vector3D(v[3], v[4], v[5]),
                                            Vector3D(-v[0], -v[1], -v[2]),
                                            Vector3D(-v[0], -v[1], -v[2])));
      p0 = p1 + 1;
    }
  }
  std::vector<Primitive*> prims = sph->get_primitives();
  // optional scene geometry: spheres, triangles and point lights from a text file
  //   sphere cx cy cz r  d|e  a b c          (diffuse reflectance or emitted radiance)
  //   tri  9 x vertex coords  9 x vertex normals  d|e  a b c
  //   point px py pz  Lr Lg Lb
  std::string scenefile = argc > a ? argv[a++] : "";
  if (!scenefile.empty()) {
    std::ifstream sf(scenefile);
    std::string kind;
    while (sf >> kind) {
      if (kind == "sphere") {
        double cx, cy, cz, rr, c0, c1, c2; std::string mk;
        sf >> cx >> cy >> cz >> rr >> mk >> c0 >> c1 >> c2;
        BSDF* b = mk == "e" ? (BSDF*)new EmissionBSDF(Vector3D(c0, c1, c2))
                            : (BSDF*)new DiffuseBSDF(Vector3D(c0, c1, c2));
        SphereObject* so = new SphereObject(Vector3D(cx, cy, cz), rr, b);
        objs.push_back(so);
        for (Primitive* p : so->get_primitives()) prims.push_back(p);
      } else if (kind == "tri") {
        double v[18], c0, c1, c2; std::string mk;
        for (int k = 0; k < 18; k++) sf >> v[k];
        sf >> mk >> c0 >> c1 >> c2;
        BSDF* b = mk == "e" ? (BSDF*)new EmissionBSDF(Vector3D(c0, c1, c2))
                            : (BSDF*)new DiffuseBSDF(Vector3D(c0, c1, c2));
        // a Mesh built from an empty HalfedgeMesh, then pointed at our own vertex arrays:
        // Triangle's constructor (src/scene/triangle.cpp:9-21) only reads positions/normals/bsdf
        HalfedgeMesh* hem = new HalfedgeMesh();
        Mesh* m = new Mesh(*hem, b);
        m->positions = new Vector3D[3];
        m->normals = new Vector3D[3];
        for (int k = 0; k < 3; k++) {
          m->positions[k] = Vector3D(v[3 * k], v[3 * k + 1], v[3 * k + 2]);
          m->normals[k] = Vector3D(v[9 + 3 * k], v[10 + 3 * k], v[11 + 3 * k]);
        }
        objs.push_back(m);
        prims.push_back(new Triangle(m, 0, 1, 2));
      } else if (kind == "point") {
        double px, py, pz, l0, l1, l2;
        sf >> px >> py >> pz >> l0 >> l1 >> l2;
        lights.push_back(new PointLight(Vector3D(l0, l1, l2), Vector3D(px, py, pz)));
      }
    }
  }
  Scene scene(objs, lights);
  BVHAccel bvh(prims, 4);

  PathTracer pt;
  pt.ns_aa = ns_aa;
  pt.max_ray_depth = 1;
  pt.ns_area_light = 1;
  pt.ns_diff = pt.ns_glsy = pt.ns_refr = 1;
  pt.samplesPerBatch = 32;
  pt.maxTolerance = 0.05;
  pt.direct_hemisphere_sample = false;
  pt.envLight = NULL;
  pt.flare_radius = flare_radius;
  pt.flare_intensity = flare_intensity;
  pt.axis_ray = Vector2D(0, 0);
  pt.angle_to_sun = 0;

  // start_raytracing order (raytraced_renderer.cpp:300-311)
  pt.clear();
  pt.set_frame_size(W, H);
  pt.bvh = &bvh;
  pt.camera = &cam;
  pt.scene = &scene;
  pt.flare_origins.clear();
  pt.flare_radiance.clear();
  pt.find_sun_pos();
  pt.generate_ghost_buffer();

  FILE* meta = fopen((out + ".meta.txt").c_str(), "w");
  fprintf(meta, "W %zu\nH %zu\nns_aa %zu\nn_flares %zu\n", W, H, ns_aa, pt.flare_origins.size());
  for (size_t l = 0; l < pt.flare_origins.size(); l++)
    fprintf(meta, "flare %a %a %a %a %a\n", pt.flare_origins[l].x, pt.flare_origins[l].y,
            pt.flare_radiance[l].x, pt.flare_radiance[l].y, pt.flare_radiance[l].z);
  fprintf(meta, "axis_ray %a %a\nangle_to_sun %a\n", pt.axis_ray.x, pt.axis_ray.y,
          (double)pt.angle_to_sun);
  fprintf(meta, "sizeof_Vector3D %zu\n", sizeof(Vector3D));
  fclose(meta);

  {
    std::vector<double> g(W * H * 3);
    for (size_t i = 0; i < W * H; i++) {
      g[3 * i] = pt.ghost_buffer.data[i].x;
      g[3 * i + 1] = pt.ghost_buffer.data[i].y;
      g[3 * i + 2] = pt.ghost_buffer.data[i].z;
    }
    write_raw(out + ".ghost.f64", g.data(), g.size() * 8);
  }
  if (pt.flare_origins.empty()) {
    fprintf(stderr, "no flare in frame: stopping before raytrace_pixel (reference UB)\n");
    return 0;
  }

  ImageBuffer fb(W, H);
  std::vector<uint32_t> order;
  if (visit == "tiles") {
    // tile queue (raytraced_renderer.cpp:314-328) + raytrace_tile (:622-647), single worker
    const size_t T = 32;
    for (size_t ty = 0; ty < H; ty += T)
      for (size_t tx = 0; tx < W; tx += T) {
        size_t x1 = std::min(tx + T, W), y1 = std::min(ty + T, H);
        for (size_t y = ty; y < y1; y++)
          for (size_t x = tx; x < x1; x++) {
            pt.raytrace_pixel(x, y);
            order.push_back((uint32_t)(x + y * W));
          }
        pt.write_to_framebuffer(fb, tx, ty, x1, y1);
      }
  } else {
    std::ifstream lf(visit.substr(5));
    size_t x, y;
    while (lf >> x >> y) {
      pt.raytrace_pixel(x, y);
      order.push_back((uint32_t)(x + y * W));
    }
    pt.write_to_framebuffer(fb, 0, 0, W, H);
  }
  {
    std::vector<double> s(W * H * 3);
    for (size_t i = 0; i < W * H; i++) {
      s[3 * i] = pt.sampleBuffer.data[i].x;
      s[3 * i + 1] = pt.sampleBuffer.data[i].y;
      s[3 * i + 2] = pt.sampleBuffer.data[i].z;
    }
    write_raw(out + ".sample.f64", s.data(), s.size() * 8);
  }
  write_raw(out + ".rgba.u32", fb.data.data(), fb.data.size() * 4);
  write_raw(out + ".order.u32", order.data(), order.size() * 4);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  std::string c = argv[1];
  if (c == "aperture") return cmd_aperture(argc, argv);
  if (c == "trace") return cmd_trace(argc, argv);
  if (c == "convert") return cmd_convert(argc, argv);
  if (c == "frame") return cmd_frame(argc, argv);
  return 1;
}
