// lens_camera_amd.cpp -- see lens_camera_amd.h.  Host logic only (coordinate hand-over, the context's
// life cycle); the march through the prescription is lf_generate_lens_rays on the GPU.
#include "lens_camera_amd.h"

#include <cstdio>
#include <cstdlib>

#include "util/random_util.h"

#include "lensflare.h"

namespace CGL {

struct LensCamera::Device {
  lf_ctx* ctx = nullptr;
  const void* mask_of = nullptr;   // aperture texture that is on the device
  float sensor_w = 36.0f;
  int n_lambda = 1;
  double z_ep = 0.0;               // the paraxial entrance pupil's z (mm): what sits at the camera position
};

namespace {
void must(lf_ctx* ctx, lf_status st, const char* what) {
  if (st == LF_OK) return;
  fprintf(stderr, "[LensCamera/MI355X] %s: %s\n", what, ctx ? lf_last_error(ctx) : "no gfx950 device (there is no CPU path)");
  exit(1);
}
}  // namespace

LensCamera::LensCamera() : Camera() {
  aperture_texture = nullptr;
  ghost_aperture_texture = nullptr;
}

LensCamera::LensCamera(const Camera& placed) : Camera(placed) {}

LensCamera::~LensCamera() {
  if (dev_) {
    if (dev_->ctx) lf_destroy(dev_->ctx);
    delete dev_;
  }
}

void LensCamera::set_lens(const std::string& lens_file, int samples_per_pixel, float sun_angular_radius, int device) {
  lens_file_ = lens_file;
  spp_ = samples_per_pixel > 0 ? samples_per_pixel : 1;
  sun_radius_ = sun_angular_radius;
  device_ = device;
  if (dev_) {   // a new prescription: the context re-loads it at the next ray
    if (dev_->ctx) lf_destroy(dev_->ctx);
    delete dev_;
    dev_ = nullptr;
  }
}

LensCamera::Device* LensCamera::dev() const {
  if (lens_file_.empty()) {
    fprintf(stderr, "[LensCamera/MI355X] generate_ray before set_lens\n");
    exit(1);
  }
  if (!dev_) {
    dev_ = new Device();
    must(nullptr, lf_create(&dev_->ctx, device_), "lf_create");
    must(dev_->ctx, lf_set_frame(dev_->ctx, 64, 64), "lf_set_frame");   // (ray generation works in millimetres)
    must(dev_->ctx, lf_load_lens_file(dev_->ctx, lens_file_.c_str()), "lf_load_lens_file");
    int n = 0, stop = 0;
    double efl = 0;
    must(dev_->ctx, lf_get_lens_info(dev_->ctx, &n, &stop, &dev_->n_lambda, &dev_->sensor_w, &efl), "lf_get_lens_info");
  }
  if (dev_->mask_of != (aperture_texture ? (const void*)aperture_texture : (const void*)dev_)) {
    // the stop's mask: the aperture PNG the reference loads for its starburst (camera.h:171), or an open stop
    if (aperture_texture && !aperture_texture->aperture.empty()) {
      must(dev_->ctx, lf_set_aperture(dev_->ctx, LF_APERTURE_STARBURST, aperture_texture->aperture.data(),
                                      (int)aperture_texture->width, (int)aperture_texture->height), "lf_set_aperture");
      dev_->mask_of = aperture_texture;
    } else {
      const float open = 1.0f;
      must(dev_->ctx, lf_set_aperture(dev_->ctx, LF_APERTURE_STARBURST, &open, 1, 1), "lf_set_aperture");
      dev_->mask_of = dev_;
    }
    // where the lens sits relative to the camera position: as the device's lens camera places it
    must(dev_->ctx, lf_set_lens_camera(dev_->ctx, 1, 1.0, 1.0), "lf_set_lens_camera");
    must(dev_->ctx, lf_get_lens_camera(dev_->ctx, nullptr, nullptr, nullptr, &dev_->z_ep), "lf_get_lens_camera");
  }
  return dev_;
}

void LensCamera::generate_rays(size_t n, const double* xy_pupil, std::vector<Ray>* rays,
                               std::vector<double>* weights, int lambda) const {
  Device* d = dev();
  if (lambda < 0) lambda = d->n_lambda / 2;
  const double sw = d->sensor_w, sh = sw / aspect_ratio();
  std::vector<float> xy(2 * n), uv(2 * n), out(8 * n);
  for (size_t i = 0; i < n; i++) {
    // the lens inverts the image: normalised (x, y) -> sensor millimetres
    xy[2 * i] = (float)(-(xy_pupil[4 * i] - 0.5) * sw);
    xy[2 * i + 1] = (float)(-(xy_pupil[4 * i + 1] - 0.5) * sh);
    uv[2 * i] = (float)(2.0 * xy_pupil[4 * i + 2] - 1.0);
    uv[2 * i + 1] = (float)(2.0 * xy_pupil[4 * i + 3] - 1.0);
  }
  must(d->ctx, lf_generate_lens_rays(d->ctx, lambda, n, xy.data(), uv.data(), out.data()), "lf_generate_lens_rays");
  rays->resize(n);
  if (weights) weights->resize(n);
  const Vector3D eye = position();
  for (size_t i = 0; i < n; i++) {
    const float* o = &out[8 * i];
    // lens space = camera space: optical axis z, the scene at z < 0 (Camera looks down -z, camera.cpp:294)
    Ray r(eye + c2w * (world_per_mm * Vector3D(o[0], o[1], (double)o[2] - d->z_ep)), c2w * Vector3D(o[3], o[4], o[5]));
    r.min_t = near_clip();
    r.max_t = far_clip();
    r.depth = o[7] != 0.0f ? 1 : 0;
    (*rays)[i] = r;
    if (weights) (*weights)[i] = o[6];
  }
}

Ray LensCamera::generate_ray(double x, double y, double pu, double pv, bool* alive, double* weight, int lambda) const {
  const double in[4] = {x, y, pu, pv};
  std::vector<Ray> rays;
  std::vector<double> w;
  generate_rays(1, in, &rays, &w, lambda);
  if (alive) *alive = rays[0].depth != 0;
  if (weight) *weight = w[0];
  return rays[0];
}

Ray LensCamera::generate_ray(double x, double y) const {
  Ray r;
  for (int attempt = 0; attempt < 64; attempt++) {
    bool alive = false;
    const double pu = random_uniform(), pv = random_uniform();
    r = generate_ray(x, y, pu, pv, &alive);
    if (alive) break;
  }
  return r;
}

}  // namespace CGL
