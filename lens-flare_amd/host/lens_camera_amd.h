// lens_camera_amd.h -- `class CGL::LensCamera : public CGL::Camera`: the north star's
// LensCamera::generate_ray on the reference's own plug-in surface.  The reference has only the
// pinhole Camera::generate_ray (src/pathtracer/camera.h:166, camera.cpp:278-305) and a stub
// Camera::generate_ray_for_thin_lens (camera.h:168, camera_lens.cpp:22-30); a comment
// (`// CameraLensStructure* ...`, camera.h:174) marks where a lens structure was planned.  This class is
// that structure: a Camera -- compiled against the reference's unchanged camera.h, so it goes wherever
// a `Camera*` goes (RaytracedRenderer::set_camera, PathTracer::camera) -- that carries a lens
// prescription and whose generate_ray marches the sensor sample through it on the MI355X
// (lf_generate_lens_rays, include/lensflare.h).
//
// Two things happen when the renderer is handed a LensCamera instead of a Camera:
//   * PathTracer::generate_ghost_buffer (pathtracer_amd.cpp) sees it (dynamic_cast: Camera is
//     polymorphic) and fills ghost_buffer with the geometric march of THIS prescription instead of the
//     paraxial quads -- no environment variable, no header change;
//   * code that wants camera rays through the real lens calls generate_ray / generate_rays below
//     (Camera::generate_ray is not virtual: a `Camera*` still gets the pinhole ray, by design of the
//     reference's header).
#pragma once

#include <string>
#include <vector>

#include "pathtracer/camera.h"   // the reference's header, as it is

namespace CGL {

class LensCamera : public Camera {
 public:
  LensCamera();
  // a configured / placed reference camera (Camera::configure, place, load_settings) becomes a lens camera
  explicit LensCamera(const Camera& placed);
  LensCamera(const LensCamera&) = delete;
  LensCamera& operator=(const LensCamera&) = delete;
  ~LensCamera();

  // the prescription (a .lens file: lens-flare_amd/data/dgauss11.lens documents the format) and how
  // the ghost march samples it; device = HIP ordinal used for ray generation
  void set_lens(const std::string& lens_file, int samples_per_pixel = 64, float sun_angular_radius = 0.05f,
                int device = 0);
  const std::string& lens_file() const { return lens_file_; }
  int samples_per_pixel() const { return spp_; }
  float sun_angular_radius() const { return sun_radius_; }
  // world units per millimetre of the prescription: the lens has a size in the scene (parallax,
  // depth of field).  The centre of the paraxial entrance pupil sits at the camera position -- with the
  // stop closed to a point the lens camera IS the reference's pinhole.  Default: 1 scene unit = 1 m.
  double world_per_mm = 0.001;
  // the renderer's sample loop images the scene THROUGH this lens (pathtracer_amd.cpp, lf_set_lens_camera:
  // depth of field, vignetting, distortion, transmission); false: ghosts through the lens, the scene
  // through the reference's pinhole (Camera::generate_ray, camera.cpp:278-305)
  bool image_scene = true;
  // one ray per wavelength instead of one at the reference wavelength (lateral / axial colour in the scene)
  bool chromatic = false;
  // > 0: the scene samples aim at the exit pupil's image x this margin (lf_set_lens_camera_aim: more of the
  // samples then leave the lens); 0: at the march's disc, so that a scene ray IS the march's primary path
  float scene_aim_margin = 0.0f;

  // LensCamera::generate_ray: (x, y) normalised sensor coordinates as for Camera::generate_ray;
  // (pu, pv) in [0,1)^2 samples the rear pupil.  The primary path is marched through the prescription
  // at wavelength index lambda (-1: the middle one).  The ray starts on the front element
  // (camera position + c2w * front-element point) and carries Camera's clip range; a sample that is
  // vignetted, clipped by the aperture mask (aperture_texture, if set) or totally reflected comes back
  // with *alive = false (its depth field is 0, a live ray's is 1).  *weight = transmitted fraction
  // (Fresnel losses and the mask's texel).
  Ray generate_ray(double x, double y, double pu, double pv, bool* alive = nullptr, double* weight = nullptr,
                   int lambda = -1) const;
  // the two-argument form hides Camera::generate_ray: the pupil point is drawn from the reference's
  // random_uniform() (util/random_util.h), re-drawn (at most 64 times) while the sample is blocked
  Ray generate_ray(double x, double y) const;
  // batched: n samples {x, y, pu, pv}; one launch
  void generate_rays(size_t n, const double* xy_pupil, std::vector<Ray>* rays, std::vector<double>* weights,
                     int lambda = -1) const;

 private:
  struct Device;
  Device* dev() const;          // the ray-generation context, created at first use
  std::string lens_file_;
  int spp_ = 64, device_ = 0;
  float sun_radius_ = 0.05f;
  mutable Device* dev_ = nullptr;
};

}  // namespace CGL
