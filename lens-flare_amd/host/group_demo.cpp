// group_demo.cpp -- a C++ host rendering one frame on several devices through the C ABI's lf_group_*
// (the shape a CGL-style application takes: one process, a context + stream per device, the
// per-frame sequence issued from one host thread per device, ONE all-gather of finished tile rows),
// checked against the same frame rendered by a single context.
//
//   group_demo <lens.txt> <mask.f32> <mw> <mh> <W> <H> <spp> <device> [<device> ...]
// lens.txt as for `shim_demo geo`.  Listing a device twice rehearses the sharding on a one-GPU box
// (such a group exchanges with peer copies instead of an RCCL communicator).  Exit code 0 = every
// device ends up with exactly the single-context frame and the shares' counters add up to its.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "lensflare.h"

struct Job {
  int n, stop, nl;
  float sensor_w;
  std::vector<float> radius, thick, semi, ior, mask;
  int mw, mh, spp;
  double light[6];
};

static int fail(const char* what, const char* why) { fprintf(stderr, "group_demo: %s: %s\n", what, why); return 2; }

static lf_status setup(lf_ctx* c, const Job& j) {
  lf_status st;
  if ((st = lf_set_params(c, 1, 25.0, 1.0)) != LF_OK) return st;
  if ((st = lf_set_aperture(c, LF_APERTURE_STARBURST, j.mask.data(), j.mw, j.mh)) != LF_OK) return st;
  if ((st = lf_set_aperture(c, LF_APERTURE_GHOST, j.mask.data(), j.mw, j.mh)) != LF_OK) return st;
  if ((st = lf_set_lens(c, j.n, j.stop, j.nl, j.radius.data(), j.thick.data(), j.ior.data(), j.semi.data(),
                        j.sensor_w)) != LF_OK) return st;
  if ((st = lf_set_jitter_counter(c, 42)) != LF_OK) return st;
  const double c2w[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pos[3] = {0, 0, 0};
  return lf_set_camera(c, c2w, pos, 40.0, 30.0);
}

// the per-frame sequence of one device's share (RaytracedRenderer::start_raytracing + its workers)
static lf_status render_share(lf_ctx* c, int /*rank*/, void* user) {
  const Job& j = *static_cast<const Job*>(user);
  lf_status st;
  if ((st = lf_find_sun_pos(c, j.light, 1)) != LF_OK) return st;
  if ((st = lf_set_sun_from_flares(c, 0, 0.0, 0.05f)) != LF_OK) return st;
  if ((st = lf_trace_ghosts(c, j.spp, 9)) != LF_OK) return st;
  return lf_render_flare_layer(c);
}

int main(int argc, char** argv) {
  if (argc < 9) return 1;
  Job j;
  std::ifstream in(argv[1]);
  in >> j.n >> j.stop >> j.nl >> j.sensor_w;
  j.radius.resize(j.n); j.thick.resize(j.n); j.semi.resize(j.n); j.ior.resize((size_t)j.n * j.nl);
  for (int k = 0; k < j.n; k++) {
    in >> j.radius[k] >> j.thick[k] >> j.semi[k];
    for (int l = 0; l < j.nl; l++) in >> j.ior[(size_t)l * j.n + k];
  }
  j.mw = atoi(argv[3]); j.mh = atoi(argv[4]);
  const int W = atoi(argv[5]), H = atoi(argv[6]);
  j.spp = atoi(argv[7]);
  j.mask.resize((size_t)j.mw * j.mh);
  FILE* f = fopen(argv[2], "rb");
  if (!f || fread(j.mask.data(), sizeof(float), j.mask.size(), f) != j.mask.size()) return fail(argv[2], "cannot read");
  fclose(f);
  const double light[6] = {0.4, 0.3, -10.0, 1.0, 0.9, 0.5};
  memcpy(j.light, light, sizeof(light));
  std::vector<int> devices;
  for (int a = 8; a < argc; a++) devices.push_back(atoi(argv[a]));
  const size_t n_px = (size_t)W * H * 3;

  // ---- the whole frame on one context ----------------------------------------------------------
  lf_ctx* one = nullptr;
  if (lf_create(&one, devices[0]) != LF_OK) return fail("lf_create", "no device");
  std::vector<double> want(n_px);
  lf_counters want_cnt;
  if (lf_set_frame(one, W, H) != LF_OK || setup(one, j) != LF_OK || lf_reset_counters(one) != LF_OK ||
      render_share(one, 0, &j) != LF_OK || lf_read_tile(one, 0, 0, 0, W, H, want.data(), 3) != LF_OK ||
      lf_get_counters(one, &want_cnt) != LF_OK)
    return fail("single context", lf_last_error(one));
  lf_destroy(one);

  // ---- the same frame on the group -------------------------------------------------------------
  lf_group* g = nullptr;
  if (lf_group_create(&g, (int)devices.size(), devices.data()) != LF_OK) return fail("lf_group_create", "failed");
  if (lf_group_set_frame(g, W, H) != LF_OK) return fail("lf_group_set_frame", lf_group_last_error(g));
  for (int r = 0; r < lf_group_size(g); r++)
    if (setup(lf_group_ctx(g, r), j) != LF_OK || lf_reset_counters(lf_group_ctx(g, r)) != LF_OK)
      return fail("setup", lf_last_error(lf_group_ctx(g, r)));
  // the cull pre-pass of this launch, shared between the devices (each builds 1 / n of the table, one all-gather)
  if (lf_group_share_cull(g, j.spp) != LF_OK) return fail("lf_group_share_cull", lf_group_last_error(g));
  if (lf_group_for_each(g, render_share, &j) != LF_OK) return fail("lf_group_for_each", lf_group_last_error(g));
  if (lf_group_gather(g, 0) != LF_OK) return fail("lf_group_gather", lf_group_last_error(g));
  lf_counters sum;
  memset(&sum, 0, sizeof(sum));
  std::vector<double> got(n_px);
  for (int r = 0; r < lf_group_size(g); r++) {
    lf_ctx* c = lf_group_ctx(g, r);
    lf_counters cnt;
    if (lf_read_tile(c, 0, 0, 0, W, H, got.data(), 3) != LF_OK || lf_get_counters(c, &cnt) != LF_OK)
      return fail("read back", lf_last_error(c));
    if (memcmp(got.data(), want.data(), n_px * sizeof(double)) != 0) { fprintf(stderr, "rank %d: frame differs\n", r); return 3; }
    sum.rays_launched += cnt.rays_launched; sum.surface_events += cnt.surface_events;
    sum.rays_clipped_stop += cnt.rays_clipped_stop; sum.rays_vignetted += cnt.rays_vignetted;
    sum.rays_tir += cnt.rays_tir; sum.rays_reached_scene += cnt.rays_reached_scene;
    sum.rays_hit_light += cnt.rays_hit_light;
  }
  if (memcmp(&sum, &want_cnt, sizeof(sum)) != 0) { fprintf(stderr, "counters of the shares do not add up\n"); return 4; }
  double peak = 0;
  for (double v : want) peak = v > peak ? v : peak;
  printf("group_demo: %d devices, %dx%d, %d spp: every device holds the single-context frame (peak %g), "
         "%llu rays\n", lf_group_size(g), W, H, j.spp, peak, (unsigned long long)sum.rays_launched);
  lf_group_destroy(g);
  return peak > 0 ? 0 : 5;
}
