// lf_pathtracer.h -- C++ host mirror of the reference's plug-in surface for the lens-flare path:
// `class PathTracer` (src/pathtracer/pathtracer.h:25-143) and the `Camera` methods the path uses
// (src/pathtracer/camera.h:166-180).  Same member names, argument meaning and call order, so the
// CGL host application (RaytracedRenderer::start_raytracing / raytrace_tile,
// src/pathtracer/raytraced_renderer.cpp:300-311, :622-647) drives it unchanged; behind every
// method sits the C ABI of include/lensflare.h (hand-written gfx950 kernels).  See INTEGRATION.md
// for the few lines a maintainer changes in the reference tree.
//
// Types are minimal stand-alone equivalents of the CGL ones (Vector2D/Vector3D are doubles,
// CGL/include/CGL/vector2D.h, vector3D.h) so this header needs nothing from the reference.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "lensflare.h"

namespace lfamd {

struct Vector2D { double x = 0, y = 0; Vector2D() {} Vector2D(double x, double y) : x(x), y(y) {} };
struct Vector3D {
  double x = 0, y = 0, z = 0;
  Vector3D() {}
  Vector3D(double x, double y, double z) : x(x), y(y), z(z) {}
};

// src/pathtracer/ray.h:20-71 (the fields generate_ray fills)
struct Ray {
  Vector3D o, d;
  double min_t = 0.0, max_t = 1e300;
  size_t depth = 0;
};

// src/util/image.h:105-242
struct HDRImageBuffer {
  size_t w = 0, h = 0;
  std::vector<Vector3D> data;
  void resize(size_t w_, size_t h_) { w = w_; h = h_; data.assign(w * h, Vector3D()); }
  void clear() { data.assign(w * h, Vector3D()); }
  void update_pixel(const Vector3D& s, size_t x, size_t y) { data[x + y * w] = s; }
  Vector3D get_pixel_value(size_t x, size_t y) const { return data[x + y * w]; }
};

// src/util/image.h:20-99
struct ImageBuffer {
  size_t w = 0, h = 0;
  std::vector<uint32_t> data;
  ImageBuffer() {}
  ImageBuffer(size_t w_, size_t h_) : w(w_), h(h_), data(w_ * h_, 0xFF000000u) {}
};

// src/pathtracer/camera.h:18-88; the PNG decode stays with the host (lodepng in the reference),
// init_from_texels takes the floats CameraApertureTexture::init derives from the red channel.
struct CameraApertureTexture {
  size_t width = 0, height = 0;
  std::vector<float> aperture;
  double total_value = 0;
  int min_x = 0, min_y = 0, max_x = -1, max_y = -1;
  void init_from_texels(const float* texels, size_t w, size_t h) {
    width = w; height = h; aperture.assign(texels, texels + w * h);
  }
};

// the slice of `Camera` (src/pathtracer/camera.h:93-199) the flare path reads
class Camera {
 public:
  double hFov = 50, vFov = 35, nClip = 0.01, fClip = 100;  // degrees, as in the reference
  Vector3D pos;
  double c2w[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};              // row-major c2w(i,j)
  double lensRadius = 0, focalDistance = 4.7;
  CameraApertureTexture* aperture_texture = nullptr;
  CameraApertureTexture* ghost_aperture_texture = nullptr;
  // Camera::generate_ray (camera.cpp:278-305): pinhole ray through normalised sensor (x, y)
  Ray generate_ray(double x, double y) const;
  // Camera::generate_ray_for_thin_lens (declared camera.h:168; a stub in the reference,
  // camera_lens.cpp:22-30): thin lens of radius lensRadius focused at focalDistance,
  // (rndR, rndTheta) in [0,1) sample the lens disc uniformly
  Ray generate_ray_for_thin_lens(double x, double y, double rndR, double rndTheta) const;
};

// the DirectionalLight fields find_sun_pos reads (src/scene/light.h:16-29)
struct DirectionalLight { Vector3D posLight, radiance; };

class PathTracer {
 public:
  explicit PathTracer(int device = 0);
  ~PathTracer();
  PathTracer(const PathTracer&) = delete;
  PathTracer& operator=(const PathTracer&) = delete;

  // ---- the reference's methods (pathtracer.h:38-96), same names ------------------------------
  void set_frame_size(size_t width, size_t height);
  void clear();
  void find_sun_pos();
  void generate_ghost_buffer();   // main thread, once per frame: renders the whole flare layer
  void raytrace_pixel(size_t x, size_t y);          // callable concurrently for distinct pixels
  Vector3D raytrace_starburst(size_t x, size_t y);  // starburst + falloff of the prepared frame
  void write_to_framebuffer(ImageBuffer& framebuffer, size_t x0, size_t y0, size_t x1, size_t y1);

  // ---- the reference's public fields (pathtracer.h:93-135) ----------------------------------
  size_t ns_aa = 1;
  double flare_radius = 20, flare_intensity = 1;
  // spectral starburst (row f4, not in the reference): empty = the reference's monochrome one;
  // otherwise one entry per wavelength, scale = lambda_ref / lambda, weight = its share of R, G, B
  std::vector<double> starburst_scale;
  std::vector<Vector3D> starburst_weight;
  HDRImageBuffer sampleBuffer, ghost_buffer;
  Camera* camera = nullptr;
  std::vector<DirectionalLight> lights;   // scene->lights filtered to DirectionalLight
  std::vector<Vector2D> flare_origins;
  std::vector<Vector3D> flare_radiance;
  Vector2D axis_ray;
  float angle_to_sun = 0;

  // ---- additions ----------------------------------------------------------------------------
  // scene radiance of one pixel, already averaged like pathtracer.cpp:841-875 (sum / (ns_aa+1));
  // default: nothing hit.  Evaluated for the whole frame inside generate_ghost_buffer().
  std::function<Vector3D(size_t, size_t)> scene_radiance;
  // use the geometric lens march for the ghosts instead of the paraxial quads
  void use_geometric_ghosts(int n_surfaces, int stop_index, int n_lambda, const float* radius,
                            const float* thickness, const float* ior, const float* semi_aperture,
                            float sensor_width_mm, const float sun_dir[3], float sun_angular_radius,
                            int spp);
  // replaces PathTracer::bvh / scene (pathtracer.h:116-124): hand the static scene to the device
  // so that the sample loop of raytrace_pixel (BVH closest hit + direct lighting) runs there too;
  // argument layout as lf_set_scene (include/lensflare.h)
  void set_scene(int n_spheres, const double* spheres, const int* sphere_material, int n_triangles,
                 const double* tri_positions, const double* tri_normals, const int* tri_material,
                 int n_materials, const double* materials, int n_lights, const double* scene_lights);
  // the same from a COLLADA file: replaces ColladaParser::load + Application::load
  // for the static scene; fills `lights` (the DirectionalLights find_sun_pos projects) and returns
  // the file's camera block (present = 0 if it has none)
  lf_collada_camera load_collada(const std::string& path);
  uint32_t jitter_seed = 5489;     // std::mt19937 default, reference visit order (32x32 tiles)
  bool counter_jitter = false;     // order-free Philox jitter instead
  // with use_geometric_ghosts: the device scene term images the scene through the same prescription
  // (lf_set_lens_camera: 0 = the reference's pinhole, 1 = reference wavelength, 2 = one ray per wavelength)
  int lens_camera_mode = 0;
  double lens_world_per_mm = 0.001;
  lf_ctx* context() { return ctx_; }
  std::string last_error() const;

 private:
  void check(lf_status st, const char* what);
  void upload_textures();
  lf_ctx* ctx_ = nullptr;
  bool frame_ready_ = false, textures_uploaded_ = false, geometric_ = false, device_scene_ = false;
  int geo_spp_ = 0;
  std::vector<double> star_;    // host mirror of raytrace_starburst for every pixel
  std::vector<double> sample_;  // host mirror of the composed sensor buffer
  std::mutex mu_;
};

// LensCamera::generate_ray of the north star: a camera ray that really went through the lens
// prescription handed to PathTracer::use_geometric_ghosts (the reference only has the pinhole
// Camera::generate_ray and a stub generate_ray_for_thin_lens, camera_lens.cpp:22-30).
// (x, y) normalised sensor coordinates as for Camera::generate_ray; (pu, pv) in [0,1)^2 samples the
// rear pupil.  The ray is marched on the GPU (lf_generate_lens_rays); a vignetted / clipped sample
// returns false.  The lens' front vertex sits at the camera position, looking down camera -z.
class LensCamera : public Camera {
 public:
  explicit LensCamera(PathTracer* pt, float sensor_width_mm = 36.0f, float sensor_height_mm = 24.0f)
      : pt_(pt), sw_(sensor_width_mm), sh_(sensor_height_mm) {}
  bool generate_ray(double x, double y, double pu, double pv, Ray* out, double* weight = nullptr,
                    int lambda = 1) const;
  // batched form: n samples {x, y, pu, pv}; rays[i].depth = 1 if alive else 0
  void generate_rays(size_t n, const double* xy_pupil, std::vector<Ray>* rays,
                     std::vector<double>* weights, int lambda = 1) const;

 private:
  PathTracer* pt_;
  float sw_, sh_;
};

}  // namespace lfamd
