// shim_demo.cpp -- drives lfamd::PathTracer exactly the way the reference's
// RaytracedRenderer::start_raytracing / raytrace_tile do (src/pathtracer/raytraced_renderer.cpp
// :300-311, :314-328, :622-647) and dumps the buffers.  tests/test_gpu_host_shim.py feeds it the
// golden cases and compares against the real reference's output.
//
//   shim_demo <case.txt> <aperture.f32> <ghost.f32> <outprefix> [n_threads]
// case.txt: W H ns_aa flare_radius flare_intensity hFov vFov  pos(3)  c2w(9)  aw ah gw gh
//           n_lights  then n_lights x (posLight xyz, radiance rgb)
#include <cstdio>
#include <cstdlib>
#include <string>
#include <fstream>
#include <thread>
#include <vector>

#include "lf_pathtracer.h"

using namespace lfamd;

static std::vector<float> read_f32(const char* path, size_t n) {
  std::vector<float> v(n);
  FILE* f = fopen(path, "rb");
  if (!f || fread(v.data(), sizeof(float), n, f) != n) { perror(path); exit(2); }
  fclose(f);
  return v;
}

template <typename T>
static void dump(const std::string& path, const T* p, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  fwrite(p, sizeof(T), n, f);
  fclose(f);
}

// shim_demo geo <lens.txt> <mask.f32> <mw> <mh> <W> <H> <spp> <sun x y z> <angular radius> <outprefix>
// lens.txt: n stop n_lambda sensor_w, then n rows {radius thickness semi_aperture ior[0..n_lambda)}
// The north star's plug-in surface through the C++ mirror: PathTracer::use_geometric_ghosts +
// generate_ghost_buffer (the geometric march fills ghost_buffer) and LensCamera::generate_rays
// (camera rays that really went through the prescription).
static int geo_main(int argc, char** argv) {
  if (argc < 14) return 1;
  std::ifstream in(argv[2]);
  int n, stop, nl;
  float sensor_w;
  in >> n >> stop >> nl >> sensor_w;
  std::vector<float> radius(n), thick(n), semi(n), ior((size_t)n * nl);
  for (int k = 0; k < n; k++) {
    in >> radius[k] >> thick[k] >> semi[k];
    for (int l = 0; l < nl; l++) in >> ior[(size_t)l * n + k];
  }
  const size_t mw = atoi(argv[4]), mh = atoi(argv[5]), W = atoi(argv[6]), H = atoi(argv[7]);
  const int spp = atoi(argv[8]);
  const float sun[3] = {(float)atof(argv[9]), (float)atof(argv[10]), (float)atof(argv[11])};
  const float alpha = (float)atof(argv[12]);
  const std::string out = argv[13];
  try {
    PathTracer pt(0);
    Camera cam;
    CameraApertureTexture ap;
    ap.init_from_texels(read_f32(argv[3], mw * mh).data(), mw, mh);
    cam.aperture_texture = &ap;
    cam.ghost_aperture_texture = &ap;
    pt.clear();
    pt.set_frame_size(W, H);
    pt.camera = &cam;
    pt.counter_jitter = true;
    pt.flare_radiance.emplace_back(1.0, 0.9, 0.5);       // the radiance the march takes for the sun
    pt.flare_origins.emplace_back(0.5, 0.5);
    pt.axis_ray = Vector2D(0.5, 0.5);
    pt.use_geometric_ghosts(n, stop, nl, radius.data(), thick.data(), ior.data(), semi.data(), sensor_w, sun,
                            alpha, spp);
    pt.generate_ghost_buffer();
    dump(out + ".ghost.f64", &pt.ghost_buffer.data[0].x, W * H * 3);
    // LensCamera: a grid of sensor positions x pupil samples
    LensCamera lc(&pt, sensor_w, sensor_w * (float)H / (float)W);
    std::vector<double> q;
    for (int i = 0; i < 16; i++)
      for (int j = 0; j < 16; j++) {
        q.push_back(0.1 + 0.8 * i / 15.0); q.push_back(0.1 + 0.8 * j / 15.0);
        q.push_back(0.05 + 0.9 * ((i * 7 + j * 3) % 16) / 15.0); q.push_back(0.05 + 0.9 * ((i * 5 + j * 11) % 16) / 15.0);
      }
    std::vector<Ray> rays;
    std::vector<double> w;
    lc.generate_rays(q.size() / 4, q.data(), &rays, &w, nl / 2);
    std::vector<double> flat;
    for (size_t i = 0; i < rays.size(); i++)
      flat.insert(flat.end(), {rays[i].o.x, rays[i].o.y, rays[i].o.z, rays[i].d.x, rays[i].d.y, rays[i].d.z, w[i],
                               (double)rays[i].depth});
    dump(out + ".rays.f64", flat.data(), flat.size());
    dump(out + ".query.f64", q.data(), q.size());
    Ray one;
    double w1 = 0;
    const bool alive = lc.generate_ray(0.5, 0.5, 0.5, 0.5, &one, &w1, nl / 2);   // the chief ray
    printf("chief ray alive=%d d=(%g, %g, %g) w=%g\n", (int)alive, one.d.x, one.d.y, one.d.z, w1);
  } catch (const std::exception& e) {
    fprintf(stderr, "shim_demo geo: %s\n", e.what());
    return 3;
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "geo") return geo_main(argc, argv);
  if (argc < 5) return 1;
  std::ifstream in(argv[1]);
  size_t W, H, ns_aa, aw, ah, gw, gh, n_lights;
  double radius, intensity;
  Camera cam;
  in >> W >> H >> ns_aa >> radius >> intensity >> cam.hFov >> cam.vFov;
  in >> cam.pos.x >> cam.pos.y >> cam.pos.z;
  for (int i = 0; i < 9; i++) in >> cam.c2w[i];
  in >> aw >> ah >> gw >> gh >> n_lights;
  std::string out = argv[4];
  int n_threads = argc > 5 ? atoi(argv[5]) : 4;
  try {
    PathTracer pt(0);
    for (size_t l = 0; l < n_lights; l++) {
      DirectionalLight d;
      in >> d.posLight.x >> d.posLight.y >> d.posLight.z >> d.radiance.x >> d.radiance.y >> d.radiance.z;
      pt.lights.push_back(d);
    }
    CameraApertureTexture ap, gh_tex;
    ap.init_from_texels(read_f32(argv[2], aw * ah).data(), aw, ah);
    gh_tex.init_from_texels(read_f32(argv[3], gw * gh).data(), gw, gh);
    cam.aperture_texture = &ap;
    cam.ghost_aperture_texture = &gh_tex;

    // ---- start_raytracing (raytraced_renderer.cpp:300-311) ----
    pt.clear();
    pt.set_frame_size(W, H);
    pt.camera = &cam;
    pt.ns_aa = ns_aa;
    pt.flare_radius = radius;
    pt.flare_intensity = intensity;
    pt.flare_origins.clear();
    pt.flare_radiance.clear();
    pt.find_sun_pos();
    pt.generate_ghost_buffer();

    // ---- tile queue + worker threads (:314-328, :352-354, :681-715) ----
    ImageBuffer fb(W, H);
    struct Tile { size_t x0, y0, x1, y1; };
    std::vector<Tile> tiles;
    const size_t T = 32;
    for (size_t y = 0; y < H; y += T)
      for (size_t x = 0; x < W; x += T) tiles.push_back({x, y, std::min(x + T, W), std::min(y + T, H)});
    std::vector<std::thread> workers;
    for (int t = 0; t < n_threads; t++)
      workers.emplace_back([&, t]() {
        for (size_t i = t; i < tiles.size(); i += n_threads) {
          const Tile& tl = tiles[i];
          for (size_t y = tl.y0; y < tl.y1; y++)
            for (size_t x = tl.x0; x < tl.x1; x++) pt.raytrace_pixel(x, y);   // raytrace_tile :637-641
          pt.write_to_framebuffer(fb, tl.x0, tl.y0, tl.x1, tl.y1);            // :646
        }
      });
    for (auto& w : workers) w.join();

    dump(out + ".sample.f64", &pt.sampleBuffer.data[0].x, W * H * 3);
    dump(out + ".ghost.f64", &pt.ghost_buffer.data[0].x, W * H * 3);
    dump(out + ".rgba.u32", fb.data.data(), W * H);
    FILE* f = fopen((out + ".flares.txt").c_str(), "w");
    fprintf(f, "%zu %a %a %a\n", pt.flare_origins.size(), pt.axis_ray.x, pt.axis_ray.y, (double)pt.angle_to_sun);
    for (size_t k = 0; k < pt.flare_origins.size(); k++)
      fprintf(f, "%a %a %a %a %a\n", pt.flare_origins[k].x, pt.flare_origins[k].y, pt.flare_radiance[k].x,
              pt.flare_radiance[k].y, pt.flare_radiance[k].z);
    fprintf(f, "aperture_bbox %d %d %d %d %a\n", ap.min_x, ap.min_y, ap.max_x, ap.max_y, ap.total_value);
    fclose(f);
    Ray r = cam.generate_ray(0.5, 0.5);
    printf("centre ray d = (%g, %g, %g)\n", r.d.x, r.d.y, r.d.z);
  } catch (const std::exception& e) {
    fprintf(stderr, "shim_demo: %s\n", e.what());
    return 3;
  }
  return 0;
}
