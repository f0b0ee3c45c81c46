// pathtracer_amd.cpp -- the drop-in: the reference's OWN `class CGL::PathTracer`
// (src/pathtracer/pathtracer.h:25-143, compiled against that header UNCHANGED) with every member on
// the MI355X through the C ABI of include/lensflare.h.  This translation unit takes the place of
// src/pathtracer/pathtracer.cpp in libpt31 (CMakeLists.txt:20-33) -- the whole file, it is removed,
// not split: every out-of-line member the header declares is defined here.  Nothing else in the
// reference tree changes: RaytracedRenderer (raytraced_renderer.cpp:58, :287-355, :622-647, :678)
// keeps calling
//     new PathTracer; set_frame_size(); clear(); find_sun_pos(); generate_ghost_buffer();   (main thread)
//     raytrace_pixel(x, y) ...; write_to_framebuffer(tile);                               (worker threads)
//     autofocus(loc);                                                                       (UI thread)
//
// What runs where: find_sun_pos, the ghosts (paraxial trace + quad rasteriser, or the geometric
// march), the starburst, the irradiance falloff, the sample loop of raytrace_pixel (camera rays,
// closest hit, direct lighting) and the tonemap all run on the device, once per frame, inside
// generate_ghost_buffer -- the place the reference already reserves for its per-frame pre-pass;
// raytrace_pixel / write_to_framebuffer are the per-pixel / per-tile hand-over of the finished
// values (plain host copies: callable concurrently, like the reference's).  The helper members the
// header also publishes (draw_ghost, rasterize_textured_triangle, ..., the single-ray integrator
// queries, autofocus) are single launches of the same device code.  The device context hangs off the
// PathTracer in a side table keyed by `this` (the class has no spare member and its header must not
// change).  No CPU fallback: without a gfx950 device the constructor ends the program.
//
// The geometric lens march (the north star's kernel) is selected WITHOUT touching the header:
//   * the camera handed to the renderer is a CGL::LensCamera (lens_camera_amd.h), or
//   * LF_LENS_FILE=<prescription.lens> is set in the environment of the unchanged host application;
//     LF_GEOMETRIC_SPP (default 64), LF_SUN_ANGULAR_RADIUS (radians, default 0.05) and
//     LF_GEOMETRIC_KEY (counter-RNG key) tune it.
// generate_ghost_buffer then fills ghost_buffer with lf_trace_ghosts (sun = the in-frame
// DirectionalLight find_sun_pos found, lf_set_sun_from_flares) instead of the paraxial quads.
//
// Built and exercised as test infrastructure by oracle/Makefile (targets `dropin`, `app`: the
// reference's other objects -- raytraced_renderer.o included -- linked unmodified, pathtracer.o
// replaced by this file) and tests/test_gpu_dropin.py, tests/test_dropin_symbols.py.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "pathtracer/pathtracer.h"   // the reference's header, as it is

#include "pathtracer/bsdf.h"
#include "pathtracer/camera.h"
#include "scene/environment_light.h"
#include "scene/light.h"
#include "scene/object.h"
#include "scene/sphere.h"
#include "scene/triangle.h"
#include "util/random_util.h"

#include "lens_camera_amd.h"
#include "lensflare.h"
#include "lf_frame_sequence.h"

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
  std::vector<double> sample;      // host mirror of the composed sensor buffer (what raytrace_pixel hands over)
  std::vector<uint32_t> rgba;      // ... of the tonemapped frame (what write_to_framebuffer hands over)
  std::vector<double> star;        // ... of raytrace_starburst(x, y): fetched when first asked for
  bool star_ready = false;
  std::mutex star_mu;
  const void* textures_of = nullptr;   // camera whose aperture textures are on the device
  const void* scene_of = nullptr;      // scene that is on the device
  const void* env_of = nullptr;        // EnvironmentLight whose map is on the device
  bool frame_ready = false;
  // the geometric march: prescription file, samples per pixel, lobe of the sun, RNG key
  std::string lens_file;
  int geo_spp = 64;
  float sun_radius = 0.05f;
  uint64_t geo_key = 0x1e45f1a4eULL;
  std::string lens_loaded;             // file that is on the device ...
  size_t lens_w = 0, lens_h = 0;       // ... for this frame size (the pixel pitch depends on it)
  bool mirror_ghost = false;           // LF_MIRROR_GHOST_BUFFER: fill the public ghost_buffer field every frame
  double lens_focus_mm = 0.0;          // object distance the loaded lens was last focused at (0: as the file says)
  float scene_aim = 0.0f;              // LF_LENS_CAMERA_AIM: margin of lf_set_lens_camera_aim (0: the march's disc)
  double world_per_mm = 0.001;         // LF_WORLD_PER_MM: scene units per lens millimetre (lens camera; 1 unit = 1 m)
  bool log_frame = true;               // LF_QUIET unset: one stdout line per frame for the march / the lens camera
  // what the device's scene kernel counted for the frame (BVHAccel::total_rays / total_isects, bvh.h:136)
  uint64_t frame_rays = 0, frame_isects = 0;
  std::atomic<bool> counters_pushed{true};
  std::atomic<uint64_t> probe_seq{0};  // counter-RNG stream of the single-ray integrator members
};

std::mutex g_mu;
std::map<const PathTracer*, DeviceState> g_state;

DeviceState& state(const PathTracer* pt) {
  // (the workers call raytrace_pixel once per pixel: each thread remembers the entry it looked up last;
  // map nodes do not move, and an entry lives as long as its PathTracer)
  static thread_local const PathTracer* last_pt = nullptr;
  static thread_local DeviceState* last_state = nullptr;
  if (pt != last_pt || !last_state) {
    std::lock_guard<std::mutex> lock(g_mu);
    last_state = &g_state[pt];
    last_pt = pt;
  }
  return *last_state;
}

void check(const DeviceState& s, lf_status st, const char* what) {
  if (st == LF_OK) return;
  fprintf(stderr, "[PathTracer/MI355X] %s: %s\n", what, lf_last_error(s.ctx));
  exit(1);
}

// a BSDF as the device's material record {kind, r, g, b} (include/lensflare.h, lf_set_scene)
void material_of(BSDF* b, double m[4]) {
  if (dynamic_cast<EmissionBSDF*>(b)) {
    Vector3D e = b->get_emission();
    m[0] = 1; m[1] = e.x; m[2] = e.y; m[3] = e.z;
  } else {
    // every other BSDF through its public f(): the diffuse one returns reflectance / pi whatever the
    // directions (bsdf.cpp:52-60); the Mirror / Refraction / Glass / Microfacet stubs return 0
    // (advanced_bsdf.cpp:17-133), i.e. under this integrator they are black occluders
    Vector3D f = b->f(Vector3D(0, 0, 1), Vector3D(0, 0, 1));
    m[0] = 2; m[1] = f.x; m[2] = f.y; m[3] = f.z;
  }
}

// the static scene, flattened the way lf_set_scene takes it (include/lensflare.h)
void upload_scene(DeviceState& s, PathTracer* pt) {
  std::vector<double> sph, tp, tn, mats, lights;
  std::vector<int> sph_m, tri_m;
  std::map<BSDF*, int> mat_of;
  auto material = [&](BSDF* b) -> int {
    auto it = mat_of.find(b);
    if (it != mat_of.end()) return it->second;
    double m[4];
    material_of(b, m);
    const int id = (int)(mats.size() / 4);
    mats.insert(mats.end(), m, m + 4);
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

// what the scene kernels need besides the geometry: the public sampling fields as they are now
void sync_scene(DeviceState& s, PathTracer* pt) {
  if (s.scene_of != pt->scene) upload_scene(s, pt);
  upload_environment(s, pt);
  check(s, lf_set_light_samples(s.ctx, (int)std::max<size_t>(1, pt->ns_area_light)), "lf_set_light_samples");
  check(s, lf_set_direct_hemisphere_sample(s.ctx, pt->direct_hemisphere_sample ? 1 : 0),
        "lf_set_direct_hemisphere_sample");
}

void sync_textures(DeviceState& s, PathTracer* pt) {
  if (s.textures_of == pt->camera) return;   // CameraApertureTexture::init already decoded the PNGs (camera.h:26-83)
  CameraApertureTexture* t[2] = {pt->camera->aperture_texture, pt->camera->ghost_aperture_texture};
  for (int k = 0; k < 2; k++)
    check(s, lf_set_aperture(s.ctx, (lf_aperture_slot)k, t[k]->aperture.data(), (int)t[k]->width,
                             (int)t[k]->height), "lf_set_aperture");
  s.textures_of = pt->camera;
}

// flare_origins, flare_radiance, axis_ray, angle_to_sun as they are now (the host may have edited them)
void sync_flares(DeviceState& s, PathTracer* pt) {
  std::vector<double> o, r;
  for (auto& f : pt->flare_origins) { o.push_back(f.x); o.push_back(f.y); }
  for (auto& f : pt->flare_radiance) { r.push_back(f.x); r.push_back(f.y); r.push_back(f.z); }
  double ax[2] = {pt->axis_ray.x, pt->axis_ray.y};
  check(s, lf_set_flares(s.ctx, (int)pt->flare_origins.size(), o.data(), r.data(), ax, pt->angle_to_sun),
        "lf_set_flares");
}

// rows [y0, y1) x columns [x0, x1) of the device ghost buffer into the public field
// (Vector3D is 24 bytes, or 32 in the AVX build: CGL/include/CGL/vector3D.h:31-43)
void fetch_ghost_rect(DeviceState& s, PathTracer* pt, int x0, int y0, int x1, int y1) {
  const size_t W = pt->sampleBuffer.w, H = pt->sampleBuffer.h;
  if (pt->ghost_buffer.w != W || pt->ghost_buffer.h != H) {
    pt->ghost_buffer.resize(W, H);
    x0 = y0 = 0; x1 = (int)W; y1 = (int)H;   // the field held another frame: take all of this one
  }
  if (x1 <= x0 || y1 <= y0) return;
  const size_t stride = sizeof(Vector3D) / sizeof(double), tw = (size_t)(x1 - x0);
  std::vector<double> tile(tw * (size_t)(y1 - y0) * stride);
  check(s, lf_read_tile(s.ctx, 1, x0, y0, x1, y1, tile.data(), stride), "lf_read_tile(ghost)");
  for (int y = y0; y < y1; y++)
    std::copy(tile.begin() + (size_t)(y - y0) * tw * stride, tile.begin() + (size_t)(y - y0 + 1) * tw * stride,
              (double*)&pt->ghost_buffer.data[(size_t)x0 + (size_t)y * W]);
}

// the helper members work on the ghost buffer outside a frame too: make sure the device has the
// frame size, the textures and the flare state they read
void helper_ready(DeviceState& s, PathTracer* pt, const char* who) {
  if (!pt->camera || pt->sampleBuffer.w == 0) {
    fprintf(stderr, "[PathTracer/MI355X] %s needs a camera and a frame size\n", who);
    exit(1);
  }
  sync_textures(s, pt);
  sync_flares(s, pt);
}

std::atomic<size_t> g_host_glue_calls{0};

}  // namespace

// how often the only member that is host glue around the reference's own BSDF / BVH objects
// (at_least_one_bounce_radiance) ran: the frame path never calls it (tests assert 0)
extern "C" size_t lf_dropin_host_glue_calls(void) { return g_host_glue_calls.load(); }

// fill the public field PathTracer::ghost_buffer from the device now (a host that reads the field;
// the reference's own host application never does, so the frame path does not pay for the copy
// unless LF_MIRROR_GHOST_BUFFER=1)
extern "C" void lf_dropin_fetch_ghost_buffer(PathTracer* pt) {
  DeviceState& s = state(pt);
  pt->ghost_buffer.resize(0, 0);   // forces the whole frame
  fetch_ghost_rect(s, pt, 0, 0, (int)pt->sampleBuffer.w, (int)pt->sampleBuffer.h);
}

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
  // (an entry outlives its PathTracer -- see the destructor -- so one at a recycled address starts afresh)
  s.sample.clear(); s.rgba.clear(); s.star.clear();
  s.star_ready = s.frame_ready = false;
  s.textures_of = s.scene_of = s.env_of = nullptr;
  s.lens_loaded.clear(); s.lens_w = s.lens_h = 0; s.lens_focus_mm = 0.0;
  const char* dev = getenv("LF_DEVICE");
  if (lf_create(&s.ctx, dev ? atoi(dev) : 0) != LF_OK) {
    fprintf(stderr, "[PathTracer/MI355X] no gfx950 device: this build has no CPU path\n");
    exit(1);
  }
  if (const char* f = getenv("LF_LENS_FILE")) s.lens_file = f;
  if (const char* v = getenv("LF_GEOMETRIC_SPP")) s.geo_spp = std::max(1, atoi(v));
  if (const char* v = getenv("LF_SUN_ANGULAR_RADIUS")) s.sun_radius = (float)atof(v);
  if (const char* v = getenv("LF_GEOMETRIC_KEY")) s.geo_key = strtoull(v, nullptr, 0);
  s.mirror_ghost = getenv("LF_MIRROR_GHOST_BUFFER") != nullptr;
  if (const char* v = getenv("LF_WORLD_PER_MM")) s.world_per_mm = atof(v);
  s.scene_aim = getenv("LF_LENS_CAMERA_AIM") ? (float)atof(getenv("LF_LENS_CAMERA_AIM")) : 0.0f;
  s.log_frame = getenv("LF_QUIET") == nullptr;
  s.frame_rays = s.frame_isects = 0;
  s.counters_pushed.store(true);
}

PathTracer::~PathTracer() {
  delete gridSampler;
  delete hemisphereSampler;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_state.find(this);
  if (it != g_state.end()) {
    if (it->second.ctx) lf_destroy(it->second.ctx);
    it->second.ctx = nullptr;   // (the entry itself stays: another thread may still hold its address)
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
  // the frame was tonemapped on the device (k_tonemap = HDRImageBuffer::toColor) when it was
  // rendered; a tile is a plain copy, callable from every worker at once
  DeviceState& s = state(this);
  if (!s.frame_ready) { fprintf(stderr, "[PathTracer/MI355X] write_to_framebuffer before generate_ghost_buffer\n"); exit(1); }
  const size_t W = sampleBuffer.w;
  for (size_t y = y0; y < y1; y++)
    std::copy(s.rgba.begin() + x0 + y * W, s.rgba.begin() + x1 + y * W, &framebuffer.data[x0 + y * framebuffer.w]);
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
  sync_flares(s, this);
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
  sync_textures(s, this);
  sync_flares(s, this);
  sync_scene(s, this);
  const size_t W = sampleBuffer.w, H = sampleBuffer.h;
  // which ghosts: a LensCamera brings its prescription, LF_LENS_FILE selects one for any camera
  const LensCamera* lens_cam = dynamic_cast<const LensCamera*>(camera);
  const std::string lens_file = lens_cam && !lens_cam->lens_file().empty() ? lens_cam->lens_file() : s.lens_file;
  lf_frame_plan plan;
  memset(&plan, 0, sizeof(plan));
  plan.ns_aa = (int)ns_aa; plan.flare_radius = flare_radius; plan.flare_intensity = flare_intensity;
  // the reference's shared std::mt19937 in its visit order (32x32 tiles, one worker); a host that
  // runs several workers has no reproducible order anyway and may switch to the counter RNG
  // (LF_COUNTER_JITTER=<key>: that key; any other value: the default one)
  plan.jitter = getenv("LF_COUNTER_JITTER") ? LF_FRAME_JITTER_COUNTER : LF_FRAME_JITTER_MT19937;
  plan.mt_seed = 5489; plan.counter_key = 0x1e45f1a4eULL;
  if (const char* v = getenv("LF_COUNTER_JITTER")) {
    const unsigned long long k = strtoull(v, nullptr, 0);
    if (k > 1) plan.counter_key = k;
  }
  plan.scene = LF_FRAME_SCENE_DEVICE;
  plan.ghosts = LF_FRAME_GHOSTS_PARAXIAL;   // the reference's paraxial quads
  if (!lens_file.empty()) {
    // the geometric march: every ghost pair of the prescription + the primary path, per sensor sample
    if (s.lens_loaded != lens_file || s.lens_w != W || s.lens_h != H) {
      check(s, lf_load_lens_file(s.ctx, lens_file.c_str()), "lf_load_lens_file");
      s.lens_loaded = lens_file; s.lens_w = W; s.lens_h = H;
      s.lens_focus_mm = 0.0;
    }
    // no sun in the frame: nothing to march towards (:724-726)
    plan.ghosts = flare_origins.empty() ? LF_FRAME_GHOSTS_NONE : LF_FRAME_GHOSTS_MARCH;
    plan.sun_from_flares = 1;
    plan.sun_angular_radius = lens_cam ? lens_cam->sun_angular_radius() : s.sun_radius;
    plan.geo_spp = lens_cam ? lens_cam->samples_per_pixel() : s.geo_spp;
    plan.geo_key = s.geo_key;
    // ... and the SCENE is imaged through the same prescription: the sample loop's camera->generate_ray
    // (pathtracer.cpp:848) becomes the primary path of the march's sensor sample (lf_set_lens_camera).
    // Camera::generate_ray is not virtual (camera.h:166), so the reference's loop could never reach a
    // lens; the device's loop does.  LF_LENS_PINHOLE_SCENE / LensCamera::image_scene = false keep the
    // reference's pinhole for the scene term (ghosts through the lens, scene through a pinhole: rounds 1-3).
    // Camera::lensRadius / focalDistance (camera.h:171-172; the -b / -d flags, autofocus): lensRadius > 0 is the
    // reference's switch for its (stubbed) thin lens.  With a real lens it asks for the focus to FOLLOW
    // focalDistance -- the sensor moves to the paraxial image of that distance (lf_focus_lens); otherwise the
    // prescription's own last thickness stands (the shipped files: focus at infinity).
    {
      const double wpm = lens_cam ? lens_cam->world_per_mm : s.world_per_mm;
      const double want = (camera->lensRadius > 0 && std::isfinite(camera->focalDistance) && camera->focalDistance > 0)
                              ? camera->focalDistance / wpm : 0.0;
      if (want != s.lens_focus_mm) {
        // (focalDistance counts from the camera position = the entrance pupil's centre, not from the first vertex)
        if (want > 0) check(s, lf_focus_lens_from_pupil(s.ctx, want, nullptr), "lf_focus_lens_from_pupil");
        else check(s, lf_load_lens_file(s.ctx, lens_file.c_str()), "lf_load_lens_file");
        s.lens_focus_mm = want;
      }
    }
    const bool through_lens = lens_cam ? lens_cam->image_scene : getenv("LF_LENS_PINHOLE_SCENE") == nullptr;
    if (through_lens) {
      plan.lens_camera_mode = (lens_cam ? lens_cam->chromatic : getenv("LF_LENS_CHROMATIC") != nullptr) ? 2 : 1;
      plan.world_per_mm = lens_cam ? lens_cam->world_per_mm : s.world_per_mm;
      plan.exposure = 0.0;                      // calibrated on the axis
      // (where the scene samples aim: the march's disc, or -- LensCamera::scene_aim_margin / LF_LENS_CAMERA_AIM > 0
      // -- the exit pupil's image, under which more of the samples leave the lens)
      check(s, lf_set_lens_camera_aim(s.ctx, lens_cam ? lens_cam->scene_aim_margin : s.scene_aim), "lf_set_lens_camera_aim");
      plan.jitter = LF_FRAME_JITTER_COUNTER;    // the lens camera's samples are the march's
      plan.counter_key = s.geo_key;
    }
  }
  if (s.log_frame) { check(s, lf_timing_reset(s.ctx), "lf_timing_reset"); check(s, lf_timing_enable(s.ctx, 1), "lf_timing_enable"); }
  check(s, lf_reset_scene_counters(s.ctx), "lf_reset_scene_counters");
  check(s, lf_reset_counters(s.ctx), "lf_reset_counters");
  const char* failed = "lf_run_frame";
  check(s, lf_run_frame(s.ctx, &plan, &failed), failed);
  // the numbers behind the reference's end-of-frame log (raytraced_renderer.cpp:706-709): the renderer
  // zeroes bvh->total_rays / total_isects AFTER this pre-pass (:349), so they are handed over by the
  // first raytrace_pixel of the frame (push_bvh_counters)
  {
    uint64_t sc[4] = {0, 0, 0, 0};
    check(s, lf_get_scene_counters(s.ctx, sc), "lf_get_scene_counters");
    s.frame_rays = sc[0]; s.frame_isects = sc[1];
    s.counters_pushed.store(false);
    if (s.log_frame) {
      // the march has no counterpart in the reference's log: one line of its own, same style
      uint64_t ev = 0;
      int launches = 0; double ms = 0.0;
      check(s, lf_get_executed_events(s.ctx, &ev), "lf_get_executed_events");
      check(s, lf_synchronize(s.ctx), "lf_synchronize");
      (void)lf_timing_get(s.ctx, "march", &launches, &ms);
      if (plan.ghosts == LF_FRAME_GHOSTS_MARCH && launches > 0 && ms > 0.0)
        fprintf(stdout, "[PathTracer/MI355X] Lens march executed %llu ray-surface intersections in %.3f ms "
                        "(%.4f million per second).\n", (unsigned long long)ev, ms, (double)ev / ms * 1e-3);
      fprintf(stdout, "[PathTracer/MI355X] Scene kernel traced %llu rays with %llu primitive tests (device counters).\n",
              (unsigned long long)sc[0], (unsigned long long)sc[1]);
      if (plan.lens_camera_mode)
        fprintf(stdout, "[PathTracer/MI355X] Lens camera marched %llu sensor samples, %llu left the front element.\n",
                (unsigned long long)sc[2], (unsigned long long)sc[3]);
      check(s, lf_timing_enable(s.ctx, 0), "lf_timing_enable");
    }
  }
  // read back what the host's per-pixel / per-tile calls hand over: the composed sensor values and
  // their tonemapped form.  ghost_buffer (a public field no caller of the reference reads) and the
  // starburst alone (raytrace_starburst) come on demand.
  s.sample.resize(W * H * 3);
  check(s, lf_read_tile(s.ctx, 0, 0, 0, (int)W, (int)H, s.sample.data(), 3), "lf_read_tile(sample)");
  s.rgba.resize(W * H);
  check(s, lf_write_to_framebuffer(s.ctx, 0, 0, (int)W, (int)H, s.rgba.data(), W), "lf_write_to_framebuffer");
  s.star_ready = false;
  if (s.mirror_ghost) {
    ghost_buffer.resize(0, 0);
    fetch_ghost_rect(s, this, 0, 0, (int)W, (int)H);
  } else {
    ghost_buffer.resize(W, H);   // :720 (its content lives on the device: lf_dropin_fetch_ghost_buffer)
  }
  s.frame_ready = true;
}

Vector3D PathTracer::raytrace_starburst(size_t x, size_t y) {     // pathtracer.cpp:947-1004
  DeviceState& s = state(this);
  if (!s.frame_ready) { fprintf(stderr, "[PathTracer/MI355X] raytrace_starburst before generate_ghost_buffer\n"); exit(1); }
  const size_t W = sampleBuffer.w, H = sampleBuffer.h;
  {
    std::lock_guard<std::mutex> lock(s.star_mu);
    if (!s.star_ready) {
      s.star.resize(W * H * 3);
      check(s, lf_read_tile(s.ctx, 2, 0, 0, (int)W, (int)H, s.star.data(), 3), "lf_read_tile(starburst)");
      s.star_ready = true;
    }
  }
  const double* v = &s.star[3 * (x + y * W)];
  return Vector3D(v[0], v[1], v[2]);
}

void PathTracer::raytrace_pixel(size_t x, size_t y) {             // pathtracer.cpp:819-899
  // sampleBuffer = total_radiance + ghost_color + starburst_radiance (:891), composed on the device
  DeviceState& s = state(this);
  if (!s.frame_ready) { fprintf(stderr, "[PathTracer/MI355X] raytrace_pixel before generate_ghost_buffer\n"); exit(1); }
  // BVHAccel::total_rays / total_isects for the reference's end-of-frame log (raytraced_renderer.cpp:706-709):
  // the renderer zeroes them between the pre-pass and the workers (:349), so the first pixel hands them over
  if (!s.counters_pushed.load(std::memory_order_acquire) && !s.counters_pushed.exchange(true) && bvh) {
    bvh->total_rays = s.frame_rays;
    bvh->total_isects = s.frame_isects;
  }
  const double* v = &s.sample[3 * (x + y * sampleBuffer.w)];
  sampleBuffer.update_pixel(Vector3D(v[0], v[1], v[2]), x, y);
}

// ---- the ghost / starburst helper members (pathtracer.h:42-57, :95-101): one device launch each ----

void PathTracer::fill_textured_pixel(float x0, float y0, float u0, float v0, float x1, float y1, float u1,
                                     float v1, float x2, float y2, float u2, float v2, int x, int y,
                                     Vector3D ghost_color) {       // pathtracer.cpp:305-343
  DeviceState& s = state(this);
  helper_ready(s, this, "fill_textured_pixel");
  const float v[12] = {x0, y0, u0, v0, x1, y1, u1, v1, x2, y2, u2, v2};
  const double c[3] = {ghost_color.x, ghost_color.y, ghost_color.z};
  check(s, lf_fill_textured_pixel(s.ctx, v, x, y, c), "lf_fill_textured_pixel");
  fetch_ghost_rect(s, this, x, y, x + 1, y + 1);
}

void PathTracer::rasterize_textured_triangle(float x0, float y0, float u0, float v0, float x1, float y1,
                                             float u1, float v1, float x2, float y2, float u2, float v2,
                                             Vector3D ghost_color) {   // pathtracer.cpp:346-410
  DeviceState& s = state(this);
  helper_ready(s, this, "rasterize_textured_triangle");
  const float v[12] = {x0, y0, u0, v0, x1, y1, u1, v1, x2, y2, u2, v2};
  const double c[3] = {ghost_color.x, ghost_color.y, ghost_color.z};
  int bb[4];
  check(s, lf_rasterize_textured_triangle(s.ctx, v, c, bb), "lf_rasterize_textured_triangle");
  fetch_ghost_rect(s, this, bb[0], bb[1], bb[2], bb[3]);
}

Vector2D PathTracer::shift_vertex(float x, float y, float scale, float shift_amount) {   // pathtracer.cpp:412-430
  DeviceState& s = state(this);
  sync_flares(s, this);   // reads axis_ray
  double out[2];
  check(s, lf_shift_vertex(s.ctx, x, y, scale, shift_amount, out), "lf_shift_vertex");
  return Vector2D(out[0], out[1]);
}

void PathTracer::draw_ghost(string color, float r1, float r2) {   // pathtracer.cpp:433-508
  DeviceState& s = state(this);
  helper_ready(s, this, "draw_ghost");
  const int channel = color == "red" ? 0 : color == "green" ? 1 : 2;   // :482-488
  int bb[4];
  check(s, lf_draw_ghost(s.ctx, channel, r1, r2, bb), "lf_draw_ghost");
  fetch_ghost_rect(s, this, bb[0], bb[1], bb[2], bb[3]);
}

std::complex<double> PathTracer::compute_phase(int flare, double u, double v, Vector2D& screen_pos) {   // :917-931
  DeviceState& s = state(this);
  if (flare < 0 || (size_t)flare >= flare_origins.size()) {   // the reference indexes its vector unchecked
    fprintf(stderr, "[PathTracer/MI355X] compute_phase: no flare %d in the frame\n", flare);
    exit(1);
  }
  sync_flares(s, this);
  double e[2], p[2];
  check(s, lf_compute_phase(s.ctx, flare, u, v, e, p), "lf_compute_phase");
  screen_pos = Vector2D(p[0], p[1]);
  return std::complex<double>(e[0], e[1]);
}

Vector3D PathTracer::calculate_irradiance_falloff(size_t x, size_t y, double radius) {   // pathtracer.cpp:1043-1063
  // the 32 draws are pixel (x, y)'s own: those the reference's shared generator hands that pixel when
  // the frame is visited in tile order (lf_set_jitter_mt19937 in generate_ghost_buffer), or the
  // counter RNG's
  DeviceState& s = state(this);
  sync_flares(s, this);
  double rgb[3];
  check(s, lf_irradiance_falloff(s.ctx, (int)x, (int)y, radius, rgb), "lf_irradiance_falloff");
  return Vector3D(rgb[0], rgb[1], rgb[2]);
}

// ---- the integrator members (pathtracer.h:62-77): single-ray queries on the device scene ----------
namespace {
void pack_ray(const Ray& r, double out[8]) {
  out[0] = r.o.x; out[1] = r.o.y; out[2] = r.o.z; out[3] = r.d.x; out[4] = r.d.y; out[5] = r.d.z;
  out[6] = r.min_t; out[7] = r.max_t;
}
Vector3D shade_on_device(PathTracer* pt, int what, const Ray& r, const Intersection& isect) {
  DeviceState& s = state(pt);
  if (!pt->scene) { fprintf(stderr, "[PathTracer/MI355X] integrator query without a scene\n"); exit(1); }
  sync_scene(s, pt);
  double ray[8], m[4], rgb[3];
  pack_ray(r, ray);
  material_of(isect.bsdf, m);
  const double n[3] = {isect.n.x, isect.n.y, isect.n.z};
  check(s, lf_scene_shade(s.ctx, what, ray, isect.t, n, m, s.probe_seq++, rgb), "lf_scene_shade");
  return Vector3D(rgb[0], rgb[1], rgb[2]);
}
}  // namespace

Vector3D PathTracer::zero_bounce_radiance(const Ray& r, const Intersection& isect) {          // :215-220
  return shade_on_device(this, 0, r, isect);
}
Vector3D PathTracer::one_bounce_radiance(const Ray& r, const Intersection& isect) {           // :222-232
  return shade_on_device(this, 1, r, isect);
}
Vector3D PathTracer::estimate_direct_lighting_hemisphere(const Ray& r, const Intersection& isect) {   // :86-138
  return shade_on_device(this, 2, r, isect);
}
Vector3D PathTracer::estimate_direct_lighting_importance(const Ray& r, const Intersection& isect) {   // :142-213
  return shade_on_device(this, 3, r, isect);
}

Vector3D PathTracer::est_radiance_global_illumination(const Ray& r) {                         // :282-302
  DeviceState& s = state(this);
  if (!scene) { fprintf(stderr, "[PathTracer/MI355X] est_radiance_global_illumination without a scene\n"); exit(1); }
  sync_scene(s, this);
  double ray[8], out[8];
  pack_ray(r, ray);
  check(s, lf_scene_trace_ray(s.ctx, ray, s.probe_seq++, out), "lf_scene_trace_ray");
  if (out[0] != 0.0) r.max_t = out[1];   // BVHAccel::intersect shortens the (mutable) segment to the hit
  return Vector3D(out[5], out[6], out[7]);
}

Vector3D PathTracer::at_least_one_bounce_radiance(const Ray& r, const Intersection& isect) {  // :234-280
  // Dead code in the reference (its only call is commented out, :299), kept callable: the direct
  // term is the device's (one_bounce_radiance above); the recursion is the Russian-roulette walk of
  // the reference's own objects -- BSDF::sample_f, BVHAccel::intersect, random_uniform -- which are
  // the host's, not this library's.
  g_host_glue_calls++;
  Vector3D L_out(0, 0, 0);
  if (r.depth <= 0) return L_out;
  L_out = one_bounce_radiance(r, isect);
  const double continue_p = 0.7;
  if (r.depth <= 1 || random_uniform() < 1 - continue_p) return L_out;
  Matrix3x3 o2w;
  make_coord_space(o2w, isect.n);
  const Matrix3x3 w2o = o2w.T();
  const Vector3D w_out = w2o * (-r.d);
  Vector3D w_in;
  double pdf;
  isect.bsdf->sample_f(w_out, &w_in, &pdf);
  Ray next(r.o + r.d * isect.t, o2w * w_in);
  next.depth = r.depth - 1;
  next.min_t = EPS_F;
  Intersection next_isect;
  if (bvh && bvh->intersect(next, &next_isect)) {
    const double cos_theta = dot(w_in.unit(), Vector3D(0, 0, 1));
    const Vector3D f = isect.bsdf->f(-1 * w_in, w_out);
    L_out += ((at_least_one_bounce_radiance(next, next_isect) * cos_theta * f) / pdf) / continue_p;
  }
  return L_out;
}

void PathTracer::autofocus(Vector2D loc) {                         // pathtracer.cpp:1065-1072
  // UI thread (raytraced_renderer.cpp:677-679): the closest hit of the camera ray through the clicked
  // pixel becomes the focal distance; a miss leaves INF_D there, like Intersection's default t
  DeviceState& s = state(this);
  if (!camera || !scene) { fprintf(stderr, "[PathTracer/MI355X] autofocus needs a camera and a scene\n"); return; }
  sync_scene(s, this);
  const Ray r = camera->generate_ray(loc.x / sampleBuffer.w, loc.y / sampleBuffer.h);
  double ray[8], out[8];
  pack_ray(r, ray);
  check(s, lf_scene_trace_ray(s.ctx, ray, s.probe_seq++, out), "lf_scene_trace_ray");
  camera->focalDistance = out[0] != 0.0 ? out[1] : INF_D;
}

}  // namespace CGL
