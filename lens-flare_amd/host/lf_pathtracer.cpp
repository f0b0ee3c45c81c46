// lf_pathtracer.cpp -- see lf_pathtracer.h.  Host logic only (call order, mirrors, errors); all
// flare arithmetic happens behind the C ABI on the GPU.
#include "lf_pathtracer.h"

#include <cstring>

#include "lf_frame_sequence.h"

#include <cmath>
#include <stdexcept>

namespace lfamd {

namespace {
const double kPi = 3.14159265358979323;  // CGL/include/CGL/misc.h:11

Vector3D mul(const double m[9], const Vector3D& v) {
  return Vector3D(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z,
                  m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
}  // namespace

Ray Camera::generate_ray(double x, double y) const {
  // camera.cpp:278-305
  double edge_x = std::tan(0.5 * (hFov * (kPi / 180.0)));
  double edge_y = std::tan(0.5 * (vFov * (kPi / 180.0)));
  Vector3D d(edge_x * (2 * x - 1), edge_y * (2 * y - 1), -1);
  double rn = 1. / std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
  d = Vector3D(d.x * rn, d.y * rn, d.z * rn);
  Ray r;
  r.o = pos;
  r.d = mul(c2w, d);
  r.min_t = nClip;
  r.max_t = fClip;
  return r;
}

Ray Camera::generate_ray_for_thin_lens(double x, double y, double rndR, double rndTheta) const {
  // The reference leaves this unimplemented (camera_lens.cpp:22-30 returns a constant ray); this
  // is the standard thin-lens construction its comment describes: the pinhole ray fixes the point
  // of perfect focus at distance focalDistance, the ray starts on a uniformly sampled lens point.
  double edge_x = std::tan(0.5 * (hFov * (kPi / 180.0)));
  double edge_y = std::tan(0.5 * (vFov * (kPi / 180.0)));
  Vector3D p_focus(edge_x * (2 * x - 1) * focalDistance, edge_y * (2 * y - 1) * focalDistance,
                   -focalDistance);
  double rr = lensRadius * std::sqrt(rndR), th = 2.0 * kPi * rndTheta;
  Vector3D p_lens(rr * std::cos(th), rr * std::sin(th), 0);
  Vector3D d(p_focus.x - p_lens.x, p_focus.y - p_lens.y, p_focus.z - p_lens.z);
  double rn = 1. / std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
  d = Vector3D(d.x * rn, d.y * rn, d.z * rn);
  Ray r;
  Vector3D ol = mul(c2w, p_lens);
  r.o = Vector3D(pos.x + ol.x, pos.y + ol.y, pos.z + ol.z);
  r.d = mul(c2w, d);
  r.min_t = nClip;
  r.max_t = fClip;
  return r;
}

void LensCamera::generate_rays(size_t n, const double* xy_pupil, std::vector<Ray>* rays,
                               std::vector<double>* weights, int lambda) const {
  std::vector<float> xy(2 * n), uv(2 * n), out(8 * n);
  for (size_t i = 0; i < n; i++) {
    // the lens inverts the image: normalised (x, y) -> sensor millimetres
    xy[2 * i] = (float)(-(xy_pupil[4 * i] - 0.5) * sw_);
    xy[2 * i + 1] = (float)(-(xy_pupil[4 * i + 1] - 0.5) * sh_);
    uv[2 * i] = (float)(2.0 * xy_pupil[4 * i + 2] - 1.0);
    uv[2 * i + 1] = (float)(2.0 * xy_pupil[4 * i + 3] - 1.0);
  }
  lf_status st = lf_generate_lens_rays(pt_->context(), lambda, n, xy.data(), uv.data(), out.data());
  if (st != LF_OK) throw std::runtime_error(std::string("lf_generate_lens_rays: ") + pt_->last_error());
  rays->resize(n);
  if (weights) weights->resize(n);
  for (size_t i = 0; i < n; i++) {
    const float* o = &out[8 * i];
    Ray r;
    Vector3D ol = mul(c2w, Vector3D(o[0], o[1], o[2]));
    r.o = Vector3D(pos.x + ol.x, pos.y + ol.y, pos.z + ol.z);
    r.d = mul(c2w, Vector3D(o[3], o[4], o[5]));
    r.min_t = nClip; r.max_t = fClip;
    r.depth = o[7] != 0.0f ? 1 : 0;
    (*rays)[i] = r;
    if (weights) (*weights)[i] = o[6];
  }
}

bool LensCamera::generate_ray(double x, double y, double pu, double pv, Ray* out, double* weight,
                              int lambda) const {
  const double in[4] = {x, y, pu, pv};
  std::vector<Ray> rays;
  std::vector<double> w;
  generate_rays(1, in, &rays, &w, lambda);
  if (out) *out = rays[0];
  if (weight) *weight = w[0];
  return rays[0].depth != 0;
}

PathTracer::PathTracer(int device) {
  lf_status st = lf_create(&ctx_, device);
  if (st != LF_OK) throw std::runtime_error("lf_create failed: no MI355X device (there is no CPU fallback)");
}

PathTracer::~PathTracer() {
  if (ctx_) lf_destroy(ctx_);
}

std::string PathTracer::last_error() const { return lf_last_error(ctx_); }

void PathTracer::check(lf_status st, const char* what) {
  if (st != LF_OK) throw std::runtime_error(std::string(what) + ": " + lf_last_error(ctx_));
}

void PathTracer::set_frame_size(size_t width, size_t height) {
  // pathtracer.cpp:66-69
  sampleBuffer.resize(width, height);
  check(lf_set_frame(ctx_, (int)width, (int)height), "lf_set_frame");
  frame_ready_ = false;
}

void PathTracer::clear() {
  // pathtracer.cpp:71-79
  camera = nullptr;
  sampleBuffer.resize(0, 0);
  frame_ready_ = false;
  textures_uploaded_ = false;
}

void PathTracer::upload_textures() {
  if (textures_uploaded_ || !camera) return;
  CameraApertureTexture* t[2] = {camera->aperture_texture, camera->ghost_aperture_texture};
  for (int s = 0; s < 2; s++) {
    if (!t[s] || t[s]->aperture.empty()) throw std::runtime_error("camera aperture texture missing");
    check(lf_set_aperture(ctx_, (lf_aperture_slot)s, t[s]->aperture.data(), (int)t[s]->width,
                          (int)t[s]->height), "lf_set_aperture");
    lf_aperture_stats st;
    check(lf_get_aperture_stats(ctx_, (lf_aperture_slot)s, &st), "lf_get_aperture_stats");
    t[s]->total_value = st.total_value;   // the fields CameraApertureTexture::init fills
    t[s]->min_x = st.min_x; t[s]->min_y = st.min_y; t[s]->max_x = st.max_x; t[s]->max_y = st.max_y;
  }
  textures_uploaded_ = true;
}

void PathTracer::find_sun_pos() {
  // pathtracer.cpp:32-64 on the device; results copied back into the public fields
  if (!camera) throw std::runtime_error("find_sun_pos: camera not set");
  // the fields may have been cleared by the caller (raytraced_renderer.cpp:306-307)
  std::vector<double> o, r;
  for (auto& f : flare_origins) { o.push_back(f.x); o.push_back(f.y); }
  for (auto& f : flare_radiance) { r.push_back(f.x); r.push_back(f.y); r.push_back(f.z); }
  double ax[2] = {axis_ray.x, axis_ray.y};
  check(lf_set_flares(ctx_, (int)flare_origins.size(), o.data(), r.data(), ax, angle_to_sun), "lf_set_flares");
  double pos[3] = {camera->pos.x, camera->pos.y, camera->pos.z};
  check(lf_set_camera(ctx_, camera->c2w, pos, camera->hFov, camera->vFov), "lf_set_camera");
  std::vector<double> L;
  for (auto& l : lights) {
    L.insert(L.end(), {l.posLight.x, l.posLight.y, l.posLight.z, l.radiance.x, l.radiance.y, l.radiance.z});
  }
  check(lf_find_sun_pos(ctx_, L.data(), (int)lights.size()), "lf_find_sun_pos");
  int n = 0;
  double oo[2 * LF_MAX_FLARES], rr[3 * LF_MAX_FLARES], a2[2];
  float ang;
  check(lf_get_flares(ctx_, &n, oo, rr, a2, &ang), "lf_get_flares");
  // find_sun_pos APPENDS to the vectors (emplace_back, :42-43); lf_find_sun_pos starts from empty
  for (int k = 0; k < n; k++) {
    flare_origins.emplace_back(oo[2 * k], oo[2 * k + 1]);
    flare_radiance.emplace_back(rr[3 * k], rr[3 * k + 1], rr[3 * k + 2]);
  }
  axis_ray = Vector2D(a2[0], a2[1]);
  angle_to_sun = ang;
  frame_ready_ = false;
}

void PathTracer::use_geometric_ghosts(int n_surfaces, int stop_index, int n_lambda,
                                      const float* radius, const float* thickness, const float* ior,
                                      const float* semi_aperture, float sensor_width_mm,
                                      const float sun_dir[3], float sun_angular_radius, int spp) {
  check(lf_set_lens(ctx_, n_surfaces, stop_index, n_lambda, radius, thickness, ior, semi_aperture,
                    sensor_width_mm), "lf_set_lens");
  float rad[3] = {1, 1, 1};
  if (!flare_radiance.empty()) {
    rad[0] = (float)flare_radiance[0].x; rad[1] = (float)flare_radiance[0].y; rad[2] = (float)flare_radiance[0].z;
  }
  check(lf_set_sun(ctx_, sun_dir, rad, sun_angular_radius), "lf_set_sun");
  geometric_ = true;
  geo_spp_ = spp;
  frame_ready_ = false;
}

void PathTracer::set_scene(int n_spheres, const double* spheres, const int* sphere_material,
                           int n_triangles, const double* tri_positions, const double* tri_normals,
                           const int* tri_material, int n_materials, const double* materials,
                           int n_lights, const double* scene_lights) {
  check(lf_set_scene(ctx_, n_spheres, spheres, sphere_material, n_triangles, tri_positions,
                     tri_normals, tri_material, n_materials, materials, n_lights, scene_lights),
        "lf_set_scene");
  device_scene_ = true;
  frame_ready_ = false;
}

lf_collada_camera PathTracer::load_collada(const std::string& path) {
  lf_collada_camera cam;
  double suns[6 * LF_MAX_FLARES];
  int n = 0;
  check(lf_load_collada(ctx_, path.c_str(), &cam, suns, LF_MAX_FLARES, &n), "lf_load_collada");
  lights.clear();
  for (int k = 0; k < n && k < LF_MAX_FLARES; k++)
    lights.push_back({Vector3D(suns[6 * k], suns[6 * k + 1], suns[6 * k + 2]),
                      Vector3D(suns[6 * k + 3], suns[6 * k + 4], suns[6 * k + 5])});
  device_scene_ = true;
  frame_ready_ = false;
  return cam;
}

void PathTracer::generate_ghost_buffer() {
  // pathtracer.cpp:714-817.  The reference fills ghost_buffer here and evaluates the starburst
  // later, pixel by pixel, inside raytrace_pixel; the device renders the whole flare layer now
  // (same place in the frame: main thread, before the workers start) and raytrace_pixel reads it.
  if (!camera) throw std::runtime_error("generate_ghost_buffer: camera not set");
  upload_textures();
  // push the public fields (the host may have edited them after find_sun_pos)
  std::vector<double> o, r;
  for (auto& f : flare_origins) { o.push_back(f.x); o.push_back(f.y); }
  for (auto& f : flare_radiance) { r.push_back(f.x); r.push_back(f.y); r.push_back(f.z); }
  double ax[2] = {axis_ray.x, axis_ray.y};
  check(lf_set_flares(ctx_, (int)flare_origins.size(), o.data(), r.data(), ax, angle_to_sun), "lf_set_flares");
  {
    if (starburst_scale.size() != starburst_weight.size())
      throw std::runtime_error("generate_ghost_buffer: starburst_scale / starburst_weight sizes differ");
    std::vector<double> w;
    for (auto& v : starburst_weight) { w.push_back(v.x); w.push_back(v.y); w.push_back(v.z); }
    check(lf_set_starburst_spectrum(ctx_, (int)starburst_scale.size(), starburst_scale.data(), w.data()),
          "lf_set_starburst_spectrum");
  }
  const size_t W = sampleBuffer.w, H = sampleBuffer.h;
  static_assert(sizeof(Vector3D) == 3 * sizeof(double), "Vector3D must be 3 packed doubles");
  // the frame itself: the one definition of the call order both host mirrors share (lf_frame_sequence.h)
  lf_frame_plan plan;
  std::memset(&plan, 0, sizeof(plan));
  plan.ns_aa = (int)ns_aa; plan.flare_radius = flare_radius; plan.flare_intensity = flare_intensity;
  plan.jitter = counter_jitter ? LF_FRAME_JITTER_COUNTER : LF_FRAME_JITTER_MT19937;
  plan.mt_seed = jitter_seed; plan.counter_key = 0x1e45f1a4eULL;
  plan.ghosts = geometric_ ? LF_FRAME_GHOSTS_MARCH : LF_FRAME_GHOSTS_PARAXIAL;
  plan.sun_from_flares = 0;             // use_geometric_ghosts set the sun explicitly
  plan.geo_spp = geo_spp_; plan.geo_key = 0x1e45f1a4eULL;
  plan.lens_camera_mode = geometric_ ? lens_camera_mode : 0;
  plan.world_per_mm = lens_world_per_mm; plan.exposure = 0.0;
  if (plan.lens_camera_mode) plan.jitter = LF_FRAME_JITTER_COUNTER;   // the lens camera's samples are the march's
  if (scene_radiance) {
    // a scene term the HOST evaluates (est_radiance_global_illumination on the CPU, a callback): handed
    // to the device, which composes (scene + ghost) + starburst exactly like pathtracer.cpp:891
    std::vector<double> scene(W * H * 3);
    for (size_t y = 0; y < H; y++)
      for (size_t x = 0; x < W; x++) {
        Vector3D t = scene_radiance(x, y);
        double* d = &scene[3 * (x + y * W)];
        d[0] = t.x; d[1] = t.y; d[2] = t.z;
      }
    check(lf_set_scene_term(ctx_, scene.data()), "lf_set_scene_term");
    plan.scene = LF_FRAME_SCENE_HOST;
  } else {
    // the sample loop of raytrace_pixel (:841-875) on the device (BVH + direct lighting), or no scene
    plan.scene = device_scene_ ? LF_FRAME_SCENE_DEVICE : LF_FRAME_SCENE_NONE;
  }
  const char* failed = "lf_run_frame";
  check(lf_run_frame(ctx_, &plan, &failed), failed);
  ghost_buffer.resize(W, H);
  check(lf_read_tile(ctx_, 1, 0, 0, (int)W, (int)H, &ghost_buffer.data[0].x, 3), "lf_read_tile(ghost)");
  star_.resize(W * H * 3);
  check(lf_read_tile(ctx_, 2, 0, 0, (int)W, (int)H, star_.data(), 3), "lf_read_tile(starburst)");
  sample_.resize(W * H * 3);
  check(lf_read_tile(ctx_, 0, 0, 0, (int)W, (int)H, sample_.data(), 3), "lf_read_tile(sample)");
  frame_ready_ = true;
}

Vector3D PathTracer::raytrace_starburst(size_t x, size_t y) {
  if (!frame_ready_) throw std::runtime_error("raytrace_starburst before generate_ghost_buffer");
  const double* s = &star_[3 * (x + y * sampleBuffer.w)];
  return Vector3D(s[0], s[1], s[2]);
}

void PathTracer::raytrace_pixel(size_t x, size_t y) {
  // pathtracer.cpp:819-899: sampleBuffer = total_radiance + ghost_color + starburst_radiance
  // (composed on the device by generate_ghost_buffer; this is the per-pixel hand-over)
  if (!frame_ready_) throw std::runtime_error("raytrace_pixel before generate_ghost_buffer");
  const double* s = &sample_[3 * (x + y * sampleBuffer.w)];
  sampleBuffer.update_pixel(Vector3D(s[0], s[1], s[2]), x, y);
}

void PathTracer::write_to_framebuffer(ImageBuffer& fb, size_t x0, size_t y0, size_t x1, size_t y1) {
  // HDRImageBuffer::toColor (util/image.h:208-223) on the device
  std::lock_guard<std::mutex> lock(mu_);
  check(lf_write_to_framebuffer(ctx_, (int)x0, (int)y0, (int)x1, (int)y1, &fb.data[x0 + y0 * fb.w],
                                fb.w), "lf_write_to_framebuffer");
}

}  // namespace lfamd
