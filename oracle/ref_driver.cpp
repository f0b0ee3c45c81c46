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
//   ref_dump collada <scene.dae> <out.txt>      (row f3: the reference's ColladaParser + the
//            GLScene -> SceneObjects conversion Application::load performs, dumped as hex floats)
//   ref_dump members <camfile> <W> <H> <aperture.png> <ghost.png> <lights> <scenefile> <out.txt>
//            every other public member of PathTracer (pathtracer.h:42-101), called one by one the way a
//            host could: the ghost / starburst helpers, the single-ray integrator queries, autofocus
// Only in the drop-in build (-DLF_DROPIN, oracle/Makefile `dropin`), for what the reference lacks:
//   ref_dump lensrays <lens file> <camfile> <aperture.png | -> <samples.txt: x y pu pv per line> <out.txt>
//            CGL::LensCamera::generate_rays (lens-flare_amd/host/lens_camera_amd.h)
//   ... and `frame` with REF_LENS_CAMERA=<lens file> [REF_LENS_SPP, REF_LENS_SUN_RADIUS] hands the
//   renderer a LensCamera instead of a Camera (the geometric march then fills ghost_buffer)

#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

// the dump reads private members of the reference's BSDF classes (reflectance, radiance); the
// reference's sources are left untouched, only this test tool sees them as public
// (likewise Mesh::indices, so that a hand-made one-triangle Mesh answers get_primitives() the way
// RaytracedRenderer::build_accel asks for it, raytraced_renderer.cpp:377-401)
#define private public
#include "pathtracer/bsdf.h"
#include "scene/object.h"
#undef private
#include "pathtracer/pathtracer.h"
#include "pathtracer/camera.h"
#include "scene/collada/collada.h"
#include "scene/gl_scene/scene.h"
#include "scene/gl_scene/mesh.h"
#include "scene/gl_scene/sphere.h"
#include "scene/gl_scene/ambient_light.h"
#include "scene/gl_scene/area_light.h"
#include "scene/gl_scene/directional_light.h"
#include "scene/gl_scene/point_light.h"
#include "scene/gl_scene/spot_light.h"
#include "scene/environment_light.h"
#include "scene/light.h"
#include "scene/object.h"
#include "scene/sphere.h"
#include "scene/triangle.h"
#include "util/halfEdgeMesh.h"
#include "util/image.h"
#ifdef LF_DROPIN
#include "lens_camera_amd.h"
extern "C" void lf_dropin_fetch_ghost_buffer(CGL::PathTracer* pt);
extern "C" size_t lf_dropin_host_glue_calls(void);
#endif

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

  Camera base_cam;
  base_cam.load_settings(camfile);
  base_cam.aperture_texture = new CameraApertureTexture();
  base_cam.aperture_texture->init(ap_png);
  base_cam.ghost_aperture_texture = new CameraApertureTexture();
  base_cam.ghost_aperture_texture->init(gh_png);
#ifdef LF_DROPIN
  // the camera route into the geometric march: the renderer is handed a LensCamera
  LensCamera lens_cam(base_cam);
  if (getenv("REF_LENS_CAMERA"))
    lens_cam.set_lens(getenv("REF_LENS_CAMERA"), getenv("REF_LENS_SPP") ? atoi(getenv("REF_LENS_SPP")) : 64,
                      getenv("REF_LENS_SUN_RADIUS") ? (float)atof(getenv("REF_LENS_SUN_RADIUS")) : 0.05f);
  // (the scene through the lens as well: on by default; REF_LENS_IMAGE_SCENE=0 keeps the reference's pinhole
  // for the scene term; REF_LENS_WORLD_PER_MM = scene units per lens millimetre; REF_LENS_CHROMATIC: a ray
  // per wavelength)
  if (getenv("REF_LENS_IMAGE_SCENE")) lens_cam.image_scene = atoi(getenv("REF_LENS_IMAGE_SCENE")) != 0;
  if (getenv("REF_LENS_WORLD_PER_MM")) lens_cam.world_per_mm = atof(getenv("REF_LENS_WORLD_PER_MM"));
  if (getenv("REF_LENS_CHROMATIC")) lens_cam.chromatic = true;
  Camera& cam = getenv("REF_LENS_CAMERA") ? static_cast<Camera&>(lens_cam) : base_cam;
#else
  Camera& cam = base_cam;
#endif

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
  //   area  pos(3) dir(3) dim_x(3) dim_y(3)  Lr Lg Lb ;  hemi Lr Lg Lb   (sampled lights, REF_NS_AREA_LIGHT each)
  std::string scenefile = argc > a ? argv[a++] : "";
  EnvironmentLight* env_light = NULL;
  if (!scenefile.empty()) {
    std::ifstream sf(scenefile);
    std::string kind;
    while (sf >> kind) {
      if (kind == "sphere") {
        double cx, cy, cz, rr, c0, c1, c2; std::string mk;
        sf >> cx >> cy >> cz >> rr >> mk >> c0 >> c1 >> c2;
        BSDF* b = mk == "e" ? (BSDF*)new EmissionBSDF(Vector3D(c0, c1, c2))
                : mk == "m" ? (BSDF*)new MirrorBSDF(Vector3D(c0, c1, c2))   // one of the stub BSDFs
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
        m->indices = {0, 1, 2};   // get_primitives() -> this one triangle (what a host that walks
                                  // scene->objects sees; the BVH below is built from `prims` as before)
        objs.push_back(m);
        prims.push_back(new Triangle(m, 0, 1, 2));
      } else if (kind == "point") {
        double px, py, pz, l0, l1, l2;
        sf >> px >> py >> pz >> l0 >> l1 >> l2;
        lights.push_back(new PointLight(Vector3D(l0, l1, l2), Vector3D(px, py, pz)));
      } else if (kind == "area") {   // area  pos(3) dir(3) dim_x(3) dim_y(3)  radiance(3)
        double v[15];
        for (int k = 0; k < 15; k++) sf >> v[k];
        lights.push_back(new AreaLight(Vector3D(v[12], v[13], v[14]), Vector3D(v[0], v[1], v[2]),
                                       Vector3D(v[3], v[4], v[5]), Vector3D(v[6], v[7], v[8]),
                                       Vector3D(v[9], v[10], v[11])));
      } else if (kind == "env") {    // env  w h file.f64 [light]   (EnvironmentLight over w*h*3 raw doubles;
                                     //  "light": also appended to scene->lights, as
                                     //  RaytracedRenderer::set_scene does, raytraced_renderer.cpp:127-128)
        size_t ew, eh; std::string ef, as_light;
        sf >> ew >> eh >> ef >> as_light;
        HDRImageBuffer* eb = new HDRImageBuffer();
        eb->resize(ew, eh);
        std::vector<double> raw(ew * eh * 3);
        FILE* fe = fopen(ef.c_str(), "rb");
        if (!fe || fread(raw.data(), 8, raw.size(), fe) != raw.size()) return 4;
        fclose(fe);
        for (size_t i = 0; i < ew * eh; i++) eb->data[i] = Vector3D(raw[3 * i], raw[3 * i + 1], raw[3 * i + 2]);
        env_light = new EnvironmentLight(eb);   // (init() also writes probability_debug.png into the cwd)
        if (as_light == "light") lights.push_back(env_light);
      } else if (kind == "hemi") {   // hemi  radiance(3)
        double l0, l1, l2;
        sf >> l0 >> l1 >> l2;
        lights.push_back(new InfiniteHemisphereLight(Vector3D(l0, l1, l2)));
      }
    }
  }
  Scene scene(objs, lights);
  BVHAccel bvh(prims, 4);

  PathTracer pt;
  pt.ns_aa = ns_aa;
  pt.max_ray_depth = 1;
  pt.ns_area_light = getenv("REF_NS_AREA_LIGHT") ? strtoul(getenv("REF_NS_AREA_LIGHT"), 0, 10) : 1;   // the -l flag
  pt.ns_diff = pt.ns_glsy = pt.ns_refr = 1;
  pt.samplesPerBatch = 32;
  pt.maxTolerance = 0.05;
  pt.direct_hemisphere_sample = getenv("REF_HEMISPHERE") != NULL;   // the -H flag
  pt.envLight = env_light;
  pt.flare_radius = flare_radius;
  pt.flare_intensity = flare_intensity;
  pt.axis_ray = Vector2D(0, 0);
  pt.angle_to_sun = 0;

  // REF_MT_BURN=n: n draws of the pixel / light samplers' generator are thrown away first, so that two
  // runs of one scene share no random number (adaptive sampling otherwise stops both at the same
  // sample count pixel after pixel, and their streams never part)
  if (getenv("REF_MT_BURN")) {
    UniformGridSampler2D burner;
    for (size_t k = strtoul(getenv("REF_MT_BURN"), 0, 10); k > 0; k--) (void)burner.get_sample();
  }

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
#ifdef LF_DROPIN
  lf_dropin_fetch_ghost_buffer(&pt);   // the public field, which this tool dumps and no host reads
#endif

  FILE* meta = fopen((out + ".meta.txt").c_str(), "w");
  fprintf(meta, "W %zu\nH %zu\nns_aa %zu\nn_flares %zu\n", W, H, ns_aa, pt.flare_origins.size());
  for (size_t l = 0; l < pt.flare_origins.size(); l++)
    fprintf(meta, "flare %a %a %a %a %a\n", pt.flare_origins[l].x, pt.flare_origins[l].y,
            pt.flare_radiance[l].x, pt.flare_radiance[l].y, pt.flare_radiance[l].z);
  fprintf(meta, "axis_ray %a %a\nangle_to_sun %a\n", pt.axis_ray.x, pt.axis_ray.y,
          (double)pt.angle_to_sun);
  fprintf(meta, "sizeof_Vector3D %zu\n", sizeof(Vector3D));
#ifdef LF_DROPIN
  fprintf(meta, "host_glue_calls %zu\n", lf_dropin_host_glue_calls());
#endif
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

// ---- row f3: COLLADA -> flat scene, through the reference's own classes ------------------------
static void put3(FILE* o, const char* tag, const Vector3D& v) {
  fprintf(o, " %s %a %a %a", tag, v.x, v.y, v.z);
}
static void put_bsdf(FILE* o, BSDF* b) {
  if (DiffuseBSDF* d = dynamic_cast<DiffuseBSDF*>(b)) { fprintf(o, " bsdf diffuse"); put3(o, "rgb", d->reflectance); }
  else if (EmissionBSDF* e = dynamic_cast<EmissionBSDF*>(b)) { fprintf(o, " bsdf emission"); put3(o, "rgb", e->radiance); }
  else if (dynamic_cast<MirrorBSDF*>(b)) fprintf(o, " bsdf mirror");
  else if (dynamic_cast<GlassBSDF*>(b)) fprintf(o, " bsdf glass");
  else if (dynamic_cast<RefractionBSDF*>(b)) fprintf(o, " bsdf refraction");
  else if (dynamic_cast<MicrofacetBSDF*>(b)) fprintf(o, " bsdf microfacet");
  else fprintf(o, " bsdf other");
}

static int cmd_collada(int argc, char** argv) {
  if (argc < 4) return 2;
  Collada::SceneInfo* info = new Collada::SceneInfo();
  if (Collada::ColladaParser::load(argv[2], info) < 0) { fprintf(stderr, "cannot load %s\n", argv[2]); return 1; }
  FILE* o = fopen(argv[3], "w");
  if (!o) return 1;
  // the node loop of Application::load (src/application/application.cpp:232-275)
  Vector3D c_pos = Vector3D(), c_dir = Vector3D();
  for (size_t i = 0; i < info->nodes.size(); i++) {
    Collada::Node& node = info->nodes[i];
    Collada::Instance* instance = node.instance;
    if (!instance) { fprintf(o, "node_without_instance\n"); continue; }
    const Matrix4x4& transform = node.transform;
    switch (instance->type) {
      case Collada::Instance::CAMERA: {
        Collada::CameraInfo* c = static_cast<Collada::CameraInfo*>(instance);
        c_pos = (transform * Vector4D(c_pos, 1)).to3D();
        c_dir = (transform * Vector4D(c->view_dir, 1)).to3D().unit();
        fprintf(o, "camera hfov %a vfov %a nclip %a fclip %a", (double)c->hFov, (double)c->vFov,
                (double)c->nClip, (double)c->fClip);
        put3(o, "pos", c_pos); put3(o, "dir", c_dir); put3(o, "up", c->up_dir);
        fprintf(o, "\n");
        break;
      }
      case Collada::Instance::LIGHT: {
        Collada::LightInfo& li = static_cast<Collada::LightInfo&>(*instance);
        GLScene::SceneLight* gl = nullptr;   // Application::init_light (:322-343)
        switch (li.light_type) {
          case Collada::LightType::AMBIENT: gl = new GLScene::AmbientLight(li); break;
          case Collada::LightType::DIRECTIONAL: gl = new GLScene::DirectionalLight(li, transform); break;
          case Collada::LightType::AREA: gl = new GLScene::AreaLight(li, transform); break;
          case Collada::LightType::POINT: gl = new GLScene::PointLight(li, transform); break;
          case Collada::LightType::SPOT: gl = new GLScene::SpotLight(li, transform); break;
          default: break;
        }
        if (!gl) { fprintf(o, "light none\n"); break; }
        SceneObjects::SceneLight* sl = gl->get_static_light();
        if (DirectionalLight* d = dynamic_cast<DirectionalLight*>(sl)) {
          fprintf(o, "light directional"); put3(o, "rad", d->radiance); put3(o, "dir_to_light", d->dirToLight);
          put3(o, "pos_light", d->posLight);
        } else if (PointLight* pl = dynamic_cast<PointLight*>(sl)) {
          fprintf(o, "light point"); put3(o, "rad", pl->radiance); put3(o, "pos", pl->position);
        } else if (AreaLight* a = dynamic_cast<AreaLight*>(sl)) {
          fprintf(o, "light area"); put3(o, "rad", a->radiance); put3(o, "pos", a->position);
          put3(o, "dir", a->direction); put3(o, "dim_x", a->dim_x); put3(o, "dim_y", a->dim_y);
        } else if (InfiniteHemisphereLight* h = dynamic_cast<InfiniteHemisphereLight*>(sl)) {
          fprintf(o, "light hemisphere"); put3(o, "rad", h->radiance);
        } else if (SpotLight* sp = dynamic_cast<SpotLight*>(sl)) {
          fprintf(o, "light spot"); put3(o, "rad", sp->radiance); put3(o, "pos", sp->position);
        } else fprintf(o, "light other");
        fprintf(o, "\n");
        break;
      }
      case Collada::Instance::SPHERE: {   // Application::init_sphere (:350-355)
        Collada::SphereInfo& si = static_cast<Collada::SphereInfo&>(*instance);
        const Vector3D position = (transform * Vector4D(0, 0, 0, 1)).projectTo3D();
        double scale = (transform * Vector4D(1, 0, 0, 0)).to3D().norm();
        GLScene::Sphere* gs = new GLScene::Sphere(si, position, scale);
        SphereObject* so = dynamic_cast<SphereObject*>(gs->get_static_object());
        fprintf(o, "sphere"); put3(o, "o", so->o); fprintf(o, " r %a", so->r); put_bsdf(o, so->get_bsdf());
        fprintf(o, "\n");
        break;
      }
      case Collada::Instance::POLYMESH: { // Application::init_polymesh (:357-360)
        Collada::PolymeshInfo& pm = static_cast<Collada::PolymeshInfo&>(*instance);
        GLScene::Mesh* gm = new GLScene::Mesh(pm, transform);
        SceneObjects::SceneObject* obj = gm->get_static_object();
        std::vector<Primitive*> prims = obj->get_primitives();
        fprintf(o, "mesh %zu", prims.size()); put_bsdf(o, obj->get_bsdf()); fprintf(o, "\n");
        for (Primitive* p : prims) {
          Triangle* t = dynamic_cast<Triangle*>(p);
          fprintf(o, "tri"); put3(o, "p1", t->p1); put3(o, "p2", t->p2); put3(o, "p3", t->p3);
          put3(o, "n1", t->n1); put3(o, "n2", t->n2); put3(o, "n3", t->n3); fprintf(o, "\n");
        }
        break;
      }
      default: fprintf(o, "other_instance\n"); break;
    }
  }
  fclose(o);
  return 0;
}

// ---- every other public member of PathTracer, one by one ----------------------------------------
static void put3v(FILE* o, const char* tag, const Vector3D& v) { fprintf(o, "%s %a %a %a\n", tag, v.x, v.y, v.z); }
static void put_ghost_sum(FILE* o, const char* tag, PathTracer& pt) {
  // position-weighted checksums of the ghost buffer + its non-zero pixels, enough to pin every texel
  double s0 = 0, s1 = 0, s2 = 0; size_t nz = 0;
  for (size_t i = 0; i < pt.ghost_buffer.data.size(); i++) {
    const Vector3D& v = pt.ghost_buffer.data[i];
    const double w = 1.0 + (double)(i % 97);
    s0 += w * v.x; s1 += w * v.y; s2 += w * v.z;
    if (v.x != 0 || v.y != 0 || v.z != 0) nz++;
  }
  fprintf(o, "%s %a %a %a %zu\n", tag, s0, s1, s2, nz);
}

static int cmd_members(int argc, char** argv) {
  if (argc < 10) return 1;
  int a = 2;
  std::string camfile = argv[a++];
  size_t W = strtoul(argv[a++], 0, 10), H = strtoul(argv[a++], 0, 10);
  std::string ap_png = argv[a++], gh_png = argv[a++], lightspec = argv[a++], scenefile = argv[a++], outp = argv[a++];
  Camera cam;
  cam.load_settings(camfile);
  cam.aperture_texture = new CameraApertureTexture();
  cam.aperture_texture->init(ap_png);
  cam.ghost_aperture_texture = new CameraApertureTexture();
  cam.ghost_aperture_texture->init(gh_png);
  std::vector<SceneObject*> objs;
  std::vector<SceneLight*> lights;
  std::vector<Primitive*> prims;
  {
    double v[6];
    if (sscanf(lightspec.c_str(), "%lf,%lf,%lf,%lf,%lf,%lf", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) != 6) return 3;
    lights.push_back(new DirectionalLight(Vector3D(v[3], v[4], v[5]), Vector3D(-v[0], -v[1], -v[2]), Vector3D(-v[0], -v[1], -v[2])));
  }
  {
    std::ifstream sf(scenefile);
    std::string kind;
    while (sf >> kind) {
      if (kind == "sphere") {
        double cx, cy, cz, rr, c0, c1, c2; std::string mk;
        sf >> cx >> cy >> cz >> rr >> mk >> c0 >> c1 >> c2;
        BSDF* b = mk == "e" ? (BSDF*)new EmissionBSDF(Vector3D(c0, c1, c2)) : (BSDF*)new DiffuseBSDF(Vector3D(c0, c1, c2));
        SphereObject* so = new SphereObject(Vector3D(cx, cy, cz), rr, b);
        objs.push_back(so);
        for (Primitive* p : so->get_primitives()) prims.push_back(p);
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
  pt.ns_aa = 0;   // no pixel-jitter draws: the shared generator then hands pixel k (tile order) draws 32k .. 32k+31
  pt.max_ray_depth = 1;
  pt.ns_area_light = 1;
  pt.ns_diff = pt.ns_glsy = pt.ns_refr = 1;
  pt.samplesPerBatch = 32;
  pt.maxTolerance = 0.05;
  pt.direct_hemisphere_sample = false;
  pt.envLight = NULL;
  pt.flare_radius = 25;
  pt.flare_intensity = 1;
  pt.axis_ray = Vector2D(0, 0);
  pt.angle_to_sun = 0;
  pt.clear();
  pt.set_frame_size(W, H);
  pt.bvh = &bvh;
  pt.camera = &cam;
  pt.scene = &scene;
  pt.find_sun_pos();
  pt.generate_ghost_buffer();
  FILE* o = fopen(outp.c_str(), "w");
  if (!o) return 2;
  fprintf(o, "n_flares %zu\n", pt.flare_origins.size());
  // calculate_irradiance_falloff, pixels in the order the tile queue would visit them
  for (size_t x = 0; x < 3; x++) put3v(o, "falloff", pt.calculate_irradiance_falloff(x, 0, x == 1 ? 3.0 : 5.0));
  put3v(o, "starburst", pt.raytrace_starburst(3, 0));   // the fourth pixel: starburst + its falloff draws
  // shift_vertex / compute_phase
  const float sv[3][4] = {{-1, 1, 7.5f, -3.25f}, {1, -1, 0.4f, 12.f}, {0.3f, 0.2f, 55.f, 0.f}};
  for (int k = 0; k < 3; k++) {
    Vector2D v = pt.shift_vertex(sv[k][0], sv[k][1], sv[k][2], sv[k][3]);
    fprintf(o, "shift_vertex %a %a\n", v.x, v.y);
  }
  const double uv[3][2] = {{0.0, 0.0}, {0.123, -0.37}, {-0.5, 0.498}};
  for (int k = 0; k < 3; k++) {
    Vector2D sp;
    std::complex<double> e = pt.compute_phase(0, uv[k][0], uv[k][1], sp);
    fprintf(o, "compute_phase %a %a %a %a\n", e.real(), e.imag(), sp.x, sp.y);
  }
  // the ghost helpers, on top of the frame's ghost buffer
#ifdef LF_DROPIN
  lf_dropin_fetch_ghost_buffer(&pt);
#endif
  put_ghost_sum(o, "ghost_frame", pt);
  pt.draw_ghost("red", 14.25f, -37.5f);
  put_ghost_sum(o, "ghost_after_draw_red", pt);
  pt.draw_ghost("blue", -3.0f, 9.0f);
  put_ghost_sum(o, "ghost_after_draw_blue", pt);
  pt.rasterize_textured_triangle(10.25f, 40.5f, 0.f, 0.f, 30.75f, 5.5f, 0.f, 300.f, 52.f, 33.25f, 420.f, 17.f, Vector3D(0.25, 2.0, 0.5));
  put_ghost_sum(o, "ghost_after_triangle", pt);
  pt.fill_textured_pixel(2.5f, 1.5f, 10.f, 20.f, 30.5f, 4.5f, 400.f, 40.f, 12.5f, 28.5f, 100.f, 450.f, 14, 10, Vector3D(3.0, 0.0, 1.5));
  pt.fill_textured_pixel(2.5f, 1.5f, 10.f, 20.f, 30.5f, 4.5f, 400.f, 40.f, 12.5f, 28.5f, 100.f, 450.f, 60, 3, Vector3D(3.0, 0.0, 1.5));   // outside: no change
  put_ghost_sum(o, "ghost_after_pixels", pt);
  put3v(o, "ghost_px_14_10", pt.ghost_buffer.get_pixel_value(14, 10));
  // the integrator members: camera rays through a few pixels, shaded through each entry point
  const double pxs[9][2] = {{0.5, 0.5}, {0.31, 0.62}, {0.7, 0.35}, {0.05, 0.95}, {0.52, 0.41},
                            {0.25, 0.75}, {0.4, 0.8}, {0.6, 0.2}, {0.45, 0.55}};
  for (int k = 0; k < 9; k++) {
    Ray r = cam.generate_ray(pxs[k][0], pxs[k][1]);
    put3v(o, "est_radiance", pt.est_radiance_global_illumination(r));
    Ray r2 = cam.generate_ray(pxs[k][0], pxs[k][1]);
    Intersection isect;
    if (bvh.intersect(r2, &isect)) {
      fprintf(o, "hit %a\n", isect.t);
      put3v(o, "zero_bounce", pt.zero_bounce_radiance(r2, isect));
      put3v(o, "one_bounce", pt.one_bounce_radiance(r2, isect));
      put3v(o, "importance", pt.estimate_direct_lighting_importance(r2, isect));
      Ray r3 = r2; r3.depth = 1;
      put3v(o, "at_least_one", pt.at_least_one_bounce_radiance(r3, isect));
    } else {
      fprintf(o, "miss\n");
    }
  }
  // autofocus, like RaytracedRenderer::autofocus from the UI thread
  const double locs[3][2] = {{W * 0.5, H * 0.5}, {W * 0.31, H * 0.62}, {W * 0.05, H * 0.95}};
  for (int k = 0; k < 3; k++) {
    cam.focalDistance = -1;
    pt.autofocus(Vector2D(locs[k][0], locs[k][1]));
    fprintf(o, "autofocus %a\n", cam.focalDistance);
  }
#ifdef LF_DROPIN
  fprintf(o, "host_glue_calls %zu\n", lf_dropin_host_glue_calls());
#endif
  fclose(o);
  return 0;
}

#ifdef LF_DROPIN
static int cmd_lensrays(int argc, char** argv) {
  if (argc < 7) return 1;
  Camera base;
  base.load_settings(argv[3]);
  base.aperture_texture = NULL;
  base.ghost_aperture_texture = NULL;
  if (std::string(argv[4]) != "-") {
    base.aperture_texture = new CameraApertureTexture();
    base.aperture_texture->init(argv[4]);
  }
  LensCamera cam(base);
  cam.set_lens(argv[2]);
  std::vector<double> in;
  {
    std::ifstream f(argv[5]);
    double v;
    while (f >> v) in.push_back(v);
  }
  const size_t n = in.size() / 4;
  std::vector<Ray> rays;
  std::vector<double> w;
  cam.generate_rays(n, in.data(), &rays, &w, getenv("REF_LENS_LAMBDA") ? atoi(getenv("REF_LENS_LAMBDA")) : -1);
  FILE* o = fopen(argv[6], "w");
  if (!o) return 2;
  for (size_t i = 0; i < n; i++)
    fprintf(o, "%a %a %a %a %a %a %a %zu %a %a\n", rays[i].o.x, rays[i].o.y, rays[i].o.z, rays[i].d.x, rays[i].d.y,
            rays[i].d.z, w[i], rays[i].depth, rays[i].min_t, rays[i].max_t);
  // the single-ray forms agree with the batch
  bool alive = false;
  double wt = 0;
  Ray r1 = cam.generate_ray(in[0], in[1], in[2], in[3], &alive, &wt);
  fprintf(o, "single %a %a %a %a %a %a %a %d\n", r1.o.x, r1.o.y, r1.o.z, r1.d.x, r1.d.y, r1.d.z, wt, alive ? 1 : 0);
  Ray r2 = cam.generate_ray(0.5, 0.5);   // pupil point drawn from random_uniform()
  fprintf(o, "drawn %a %a %a %a %a %a %zu\n", r2.o.x, r2.o.y, r2.o.z, r2.d.x, r2.d.y, r2.d.z, r2.depth);
  fclose(o);
  return 0;
}
#endif

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  std::string c = argv[1];
  if (c == "aperture") return cmd_aperture(argc, argv);
  if (c == "trace") return cmd_trace(argc, argv);
  if (c == "convert") return cmd_convert(argc, argv);
  if (c == "frame") return cmd_frame(argc, argv);
  if (c == "collada") return cmd_collada(argc, argv);
  if (c == "members") return cmd_members(argc, argv);
#ifdef LF_DROPIN
  if (c == "lensrays") return cmd_lensrays(argc, argv);
#endif
  return 1;
}
