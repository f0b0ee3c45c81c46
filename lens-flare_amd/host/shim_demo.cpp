// shim_demo.cpp -- drives lfamd::PathTracer exactly the way the reference's
// RaytracedRenderer::start_raytracing / raytrace_tile do (src/pathtracer/raytraced_renderer.cpp
// :300-311, :314-328, :622-647) and dumps the buffers.  tests/test_gpu_host_shim.py feeds it the
// golden cases and compares against the real reference's output.
//
//   shim_demo <case.txt> <aperture.f32> <ghost.f32> <outprefix> [n_threads]
// case.txt: W H ns_aa flare_radius flare_intensity hFov vFov  pos(3)  c2w(9)  aw ah gw gh
//           n_lights  then n_lights x (posLight xyz, radiance rgb)
#include <cstdio>
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

int main(int argc, char** argv) {
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
