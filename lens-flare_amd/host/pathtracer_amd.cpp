// pathtracer_amd.cpp -- the drop-in: the reference's OWN `class CGL::PathTracer`
// (src/pathtracer/pathtracer.h:25-143, compiled against that header UNCHANGED) with the lens-flare
// members implemented on the MI355X through the C ABI of include/lensflare.h.  This translation
// unit takes the place of src/pathtracer/pathtracer.cpp in libpt31 (CMakeLists.txt:20-33); nothing
// else in the reference tree changes: RaytracedRenderer::start_raytracing / raytrace_tile
// (raytraced_renderer.cpp:287-355, :622-647) keep calling
//     clear(); set_frame_size(); find_sun_pos(); generate_ghost_buffer();        (main thread)
//     raytrace_pixel(x, y) ...; write_to_framebuffer(tile);                      (worker threads)
//
// What runs where: find_sun_pos, the paraxial ghost trace + quad rasteriser, the starburst, the
// irradiance falloff, the sample loop of raytrace_pixel (camera rays, closest hit, direct lighting)
// and the tonemap all run on the device; raytrace_pixel itself is the per-pixel hand-over of the
// composed value.  The device context hangs off the PathTracer in a side table keyed by `this`
// (the class has no spare member and its header must not change).
//
// Built and exercised as test infrastructure by oracle/Makefile (target `dropin`: the reference's
// other objects linked unmodified, pathtracer.o replaced by this file) and tests/test_gpu_dropin.py.
// The integrator methods that are not on the flare path (estimate_direct_lighting_*, *_bounce_radiance,
// autofocus) are not defined here: in the reference tree they stay where they are, in this build
// nothing calls them.  No CPU fallback: without a device the constructor ends the program.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "pathtracer/pathtracer.h"   // the reference's header, as it is

#include "pathtracer/bsdf.h"
#include "pathtracer/camera.h"
#include "scene/environment_light.h"
#include "scene/light.h"
#include "scene/object.h"
#include "scene/sphere.h"
#include "scene/triangle.h"

#include "lensflare.h"

using namespace CGL::SceneObjects;

namespace CGL {

namespace {

// Camera::hFov has no accessor (camera.h:188).  Reading a private member through an explicit
// template instantiation is standard C++ (access checks do not apply to explicit-instantiation
// arguments); the header stays untouched and no macro redefines `private`.
template <typename Tag, typename Tag::type Member>
struct PrivateMember {
  friend typename Tag::type get(Tag) { return Member; }
};
struct CameraHFov {
  typedef double Camera::*type;
  friend type get(CameraHFov);
};
template struct PrivateMember<CameraHFov, &Camera::hFov>;
// ... and EnvironmentLight keeps its map private (environment_light.h:33)
struct EnvLightMap {
  typedef const HDRImageBuffer* EnvironmentLight::*type;
  friend type get(EnvLightMap);
};
template struct PrivateMember<EnvLightMap, &EnvironmentLight::envMap>;

struct DeviceState {
  lf_ctx* ctx = nullptr;
  std::vector<double> sample;   // host mirror of the composed sensor buffer
  std::vector<double> star;     // ... of raytrace_starburst(x, y)
  const void* textures_of = nullptr;   // camera whose aperture textures are on the device
  const void* scene_of = nullptr;      // scene that is on the device
  const void* env_of = nullptr;        // EnvironmentLight whose map is on the device
  bool frame_ready = false;
};

std::mutex g_mu;
std::map<const PathTracer*, DeviceState> g_state;

DeviceState& state(const PathTracer* pt) {
  std::lock_guard<std::mutex> lock(g_mu);
  return g_state[pt];
}

void check(const DeviceState& s, lf_status st, const char* what) {
  if (st == LF_OK) return;
  fprintf(stderr, "[PathTracer/MI355X] %s: %s\n", what, lf_last_error(s.ctx));
  exit(1);
}

// the static scene, flattened the way lf_set_scene takes it (include/lensflare.h)
void upload_scene(DeviceState& s, PathTracer* pt) {
  std::vector<double> sph, tp, tn, mats, lights;
  std::vector<int> sph_m, tri_m;
  std::map<BSDF*, int> mat_of;
  auto material = [&](BSDF* b) -> int {
    auto it = mat_of.find(b);
    if (it != mat_of.end()) return it->second;
    double kind, rgb[3];
    if (dynamic_cast<EmissionBSDF*>(b)) {
      Vector3D e = b->get_emission();
      kind = 1; rgb[0] = e.x; rgb[1] = e.y; rgb[2] = e.z;
    } else {
      // every other BSDF through its public f(): the diffuse one returns reflectance / pi whatever the
      // directions (bsdf.cpp:52-60); the Mirror / Refraction / Glass / Microfacet stubs return 0
      // (advanced_bsdf.cpp:17-133), i.e. under this integrator they are black occluders
      Vector3D f = b->f(Vector3D(0, 0, 1), Vector3D(0, 0, 1));
      kind = 2; rgb[0] = f.x; rgb[1] = f.y; rgb[2] = f.z;
    }
    const int id = (int)(mats.size() / 4);
    mats.insert(mats.end(), {kind, rgb[0], rgb[1], rgb[2]});
    mat_of[b] = id;
    return id;
  };
  for (SceneObject* obj : pt->scene->objects)
    for (Primitive* prim : obj->get_primitives()) {
      if (Sphere* sp = dynamic_cast<Sphere*>(prim)) {
        sph.insert(sph.end(), {sp->o.x, sp->o.y, sp->o.z, sp->r});
        sph_m.push_back(material(sp->get_bsdf()));
      } else if (Triangle* t = dynamic_cast<Triangle*>(prim)) {
        const Vector3D* v[6] = {&t->p1, &t->p2, &t->p3, &t->n1, &t->n2, &t->n3};
        for (int k = 0; k < 3; k++) tp.insert(tp.end(), {v[k]->x, v[k]->y, v[k]->z});
        for (int k = 3; k < 6; k++) tn.insert(tn.end(), {v[k]->x, v[k]->y, v[k]->z});
        tri_m.push_back(material(t->get_bsdf()));
      }
    }
  // scene->lights in order, in the general row form of lf_set_scene_lights
  for (SceneLight* l : pt->scene->lights) {
    double row[16] = {0};
    auto put = [&](int at, const Vector3D& v) { row[at] = v.x; row[at + 1] = v.y; row[at + 2] = v.z; };
    if (DirectionalLight* d = dynamic_cast<DirectionalLight*>(l)) { row[0] = 0; put(1, d->radiance); put(4, d->dirToLight); }
    else if (PointLight* p = dynamic_cast<PointLight*>(l)) { row[0] = 1; put(1, p->radiance); put(4, p->position); }
    else if (InfiniteHemisphereLight* h = dynamic_cast<InfiniteHemisphereLight*>(l)) { row[0] = 2; put(1, h->radiance); }
    else if (AreaLight* a = dynamic_cast<AreaLight*>(l)) {
      row[0] = 3; put(1, a->radiance); put(4, a->position); put(7, a->direction); put(10, a->dim_x); put(13, a->dim_y);
    } else if (dynamic_cast<EnvironmentLight*>(l)) {
      row[0] = 4;   // the map itself: upload_environment
    } else {
      fprintf(stderr, "[PathTracer/MI355X] spot / sphere / mesh lights are stubs in the reference (light.cpp): not rendered\n");
      exit(1);
    }
    lights.insert(lights.end(), row, row + 16);
  }
  check(s, lf_set_scene(s.ctx, (int)sph_m.size(), sph.data(), sph_m.data(), (int)tri_m.size(), tp.data(),
                        tn.data(), tri_m.data(), (int)(mats.size() / 4), mats.data(), 0, nullptr), "lf_set_scene");
  check(s, lf_set_scene_lights(s.ctx, (int)(lights.size() / 16), lights.data()), "lf_set_scene_lights");
  check(s, lf_set_light_samples(s.ctx, (int)std::max<size_t>(1, pt->ns_area_light)), "lf_set_light_samples");
  s.scene_of = pt->scene;
}

// PathTracer::envLight (pathtracer.h:119): the map behind it, texel by texel (Vector3D is 24 or 32
// bytes wide); the sampling tables are rebuilt on the other side of the ABI
void upload_environment(DeviceState& s, PathTracer* pt) {
  if (s.env_of == pt->envLight) return;
  if (!pt->envLight) {
    check(s, lf_set_environment_map(s.ctx, 0, 0, nullptr), "lf_set_environment_map");
  } else {
    const HDRImageBuffer* m = (*pt->envLight).*get(EnvLightMap());
    std::vector<double> rgb;
    rgb.reserve(3 * m->w * m->h);
    for (size_t i = 0; i < m->w * m->h; i++) rgb.insert(rgb.end(), {m->data[i].x, m->data[i].y, m->data[i].z});
    check(s, lf_set_environment_map(s.ctx, (int)m->w, (int)m->h, rgb.data()), "lf_set_environment_map");
  }
  s.env_of = pt->envLight;
}

}  // namespace

PathTracer::PathTracer() {
  gridSampler = new UniformGridSampler2D();          // pathtracer.cpp:14-24: the host still owns these
  hemisphereSampler = new UniformHemisphereSampler3D();
  tm_gamma = 2.2f;
  tm_level = 1.0f;
  tm_key = 0.18;
  tm_wht = 5.0f;
  ghost_buffer = HDRImageBuffer();
  bvh = NULL; scene = NULL; camera = NULL; envLight = NULL;
  DeviceState& s = state(this);
  const char* dev = getenv("LF_DEVICE");
  if (lf_create(&s.ctx, dev ? atoi(dev) : 0) != LF_OK) {
    fprintf(stderr, "[PathTracer/MI355X] no gfx950 device: this build has no CPU path\n");
    exit(1);
  }
}

PathTracer::~PathTracer() {
  delete gridSampler;
  delete hemisphereSampler;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_state.find(this);
  if (it != g_state.end()) {
    if (it->second.ctx) lf_destroy(it->second.ctx);
    g_state.erase(it);
  }
}

void PathTracer::set_frame_size(size_t width, size_t height) {     // pathtracer.cpp:66-69
  sampleBuffer.resize(width, height);
  sampleCountBuffer.resize(width * height);
  DeviceState& s = state(this);
  check(s, lf_set_frame(s.ctx, (int)width, (int)height), "lf_set_frame");
  s.frame_ready = false;
}

void PathTracer::clear() {                                         // pathtracer.cpp:71-79
  bvh = NULL;
  scene = NULL;
  camera = NULL;
  sampleBuffer.clear();
  sampleCountBuffer.clear();
  sampleBuffer.resize(0, 0);
  sampleCountBuffer.resize(0, 0);
  state(this).frame_ready = false;
}

void PathTracer::write_to_framebuffer(ImageBuffer& framebuffer, size_t x0, size_t y0, size_t x1,
                                      size_t y1) {                 // pathtracer.cpp:81-84 -> image.h:208-223
  DeviceState& s = state(this);
  static std::mutex mu;   // lf_write_to_framebuffer launches the tonemap once: one caller at a time
  std::lock_guard<std::mutex> lock(mu);
  check(s, lf_write_to_framebuffer(s.ctx, (int)x0, (int)y0, (int)x1, (int)y1,
                                   &framebuffer.data[x0 + y0 * framebuffer.w], framebuffer.w),
        "lf_write_to_framebuffer");
}

void PathTracer::find_sun_pos() {                                  // pathtracer.cpp:32-64
  DeviceState& s = state(this);
  std::vector<double> L;
  for (SceneLight* light : scene->lights)
    if (DirectionalLight* d = dynamic_cast<DirectionalLight*>(light))
      L.insert(L.end(), {d->posLight.x, d->posLight.y, d->posLight.z, d->radiance.x, d->radiance.y,
                         d->radiance.z});
  double c2w[9], pos[3] = {camera->position().x, camera->position().y, camera->position().z};
  for (int i = 0; i < 9; i++) c2w[i] = camera->c2w(i / 3, i % 3);
  check(s, lf_set_camera(s.ctx, c2w, pos, (*camera).*get(CameraHFov()), camera->v_fov()), "lf_set_camera");
  check(s, lf_set_sampling(s.ctx, (int)samplesPerBatch, maxTolerance, camera->near_clip(),
                           camera->far_clip()), "lf_set_sampling");
  // the members keep what they held (axis_ray / angle_to_sun survive a frame without a sun, the
  // vectors are appended to: emplace_back at :42-43)
  std::vector<double> o, r;
  for (auto& f : flare_origins) { o.push_back(f.x); o.push_back(f.y); }
  for (auto& f : flare_radiance) { r.push_back(f.x); r.push_back(f.y); r.push_back(f.z); }
  double ax[2] = {axis_ray.x, axis_ray.y};
  check(s, lf_set_flares(s.ctx, (int)flare_origins.size(), o.data(), r.data(), ax, angle_to_sun), "lf_set_flares");
  check(s, lf_find_sun_pos(s.ctx, L.data(), (int)(L.size() / 6)), "lf_find_sun_pos");
  int n = 0;
  double oo[2 * LF_MAX_FLARES], rr[3 * LF_MAX_FLARES], a2[2];
  float ang = 0;
  check(s, lf_get_flares(s.ctx, &n, oo, rr, a2, &ang), "lf_get_flares");
  for (int k = 0; k < n; k++) {
    flare_origins.emplace_back(oo[2 * k], oo[2 * k + 1]);
    flare_radiance.emplace_back(rr[3 * k], rr[3 * k + 1], rr[3 * k + 2]);
  }
  axis_ray = Vector2D(a2[0], a2[1]);
  angle_to_sun = ang;
  s.frame_ready = false;
}

void PathTracer::generate_ghost_buffer() {                         // pathtracer.cpp:714-817
  // Main thread, once per frame, before the workers start: the reference fills ghost_buffer here and
  // evaluates scene radiance + starburst later, pixel by pixel, inside raytrace_pixel.  The device
  // renders all of it now; raytrace_pixel hands the composed pixels over.
  DeviceState& s = state(this);
  if (s.textures_of != camera) {   // CameraApertureTexture::init already decoded the PNGs (camera.h:26-83)
    CameraApertureTexture* t[2] = {camera->aperture_texture, camera->ghost_aperture_texture};
    for (int k = 0; k < 2; k++)
      check(s, lf_set_aperture(s.ctx, (lf_aperture_slot)k, t[k]->aperture.data(), (int)t[k]->width,
                               (int)t[k]->height), "lf_set_aperture");
    s.textures_of = camera;
  }
  // the public fields as they are now (the host may have edited them after find_sun_pos)
  std::vector<double> o, r;
  for (auto& f : flare_origins) { o.push_back(f.x); o.push_back(f.y); }
  for (auto& f : flare_radiance) { r.push_back(f.x); r.push_back(f.y); r.push_back(f.z); }
  double ax[2] = {axis_ray.x, axis_ray.y};
  check(s, lf_set_flares(s.ctx, (int)flare_origins.size(), o.data(), r.data(), ax, angle_to_sun), "lf_set_flares");
  check(s, lf_set_params(s.ctx, (int)ns_aa, flare_radius, flare_intensity), "lf_set_params");
  // the reference's shared std::mt19937 in its visit order (32x32 tiles, one worker); a host that
  // runs several workers has no reproducible order anyway and may switch to lf_set_jitter_counter
  if (getenv("LF_COUNTER_JITTER")) check(s, lf_set_jitter_counter(s.ctx, 0x1e45f1a4eULL), "lf_set_jitter_counter");
  else check(s, lf_set_jitter_mt19937(s.ctx, 5489, nullptr, 0), "lf_set_jitter_mt19937");
  if (s.scene_of != scene) upload_scene(s, this);
  upload_environment(s, this);
  check(s, lf_set_direct_hemisphere_sample(s.ctx, direct_hemisphere_sample ? 1 : 0), "lf_set_direct_hemisphere_sample");
  check(s, lf_render_scene_term(s.ctx), "lf_render_scene_term");
  check(s, lf_generate_ghost_buffer(s.ctx), "lf_generate_ghost_buffer");
  check(s, lf_render_flare_layer(s.ctx), "lf_render_flare_layer");
  const size_t W = sampleBuffer.w, H = sampleBuffer.h;
  ghost_buffer.resize(W, H);
  // Vector3D is 24 bytes, or 32 in the AVX build (CGL/include/CGL/vector3D.h:31-43)
  check(s, lf_read_tile(s.ctx, 1, 0, 0, (int)W, (int)H, (double*)&ghost_buffer.data[0],
                        sizeof(Vector3D) / sizeof(double)), "lf_read_tile(ghost)");
  s.sample.resize(W * H * 3);
  s.star.resize(W * H * 3);
  check(s, lf_read_tile(s.ctx, 0, 0, 0, (int)W, (int)H, s.sample.data(), 3), "lf_read_tile(sample)");
  check(s, lf_read_tile(s.ctx, 2, 0, 0, (int)W, (int)H, s.star.data(), 3), "lf_read_tile(starburst)");
  s.frame_ready = true;
}

Vector3D PathTracer::raytrace_starburst(size_t x, size_t y) {     // pathtracer.cpp:947-1004
  DeviceState& s = state(this);
  if (!s.frame_ready) { fprintf(stderr, "[PathTracer/MI355X] raytrace_starburst before generate_ghost_buffer\n"); exit(1); }
  const double* v = &s.star[3 * (x + y * sampleBuffer.w)];
  return Vector3D(v[0], v[1], v[2]);
}

void PathTracer::raytrace_pixel(size_t x, size_t y) {             // pathtracer.cpp:819-899
  // sampleBuffer = total_radiance + ghost_color + starburst_radiance (:891), composed on the device
  DeviceState& s = state(this);
  if (!s.frame_ready) { fprintf(stderr, "[PathTracer/MI355X] raytrace_pixel before generate_ghost_buffer\n"); exit(1); }
  const double* v = &s.sample[3 * (x + y * sampleBuffer.w)];
  sampleBuffer.update_pixel(Vector3D(v[0], v[1], v[2]), x, y);
}

}  // namespace CGL
