// lf_frame_sequence.h -- ONE definition of the per-frame call sequence over the C ABI.
//
// The reference renders a frame as  find_sun_pos(); generate_ghost_buffer();  on the main thread and
// then raytrace_pixel(x, y) for every pixel on its workers (raytraced_renderer.cpp:300-311, :637-641;
// pathtracer.cpp:819-899).  On the device the same work is: [the flare state is in place: lf_find_sun_pos
// / lf_set_flares] -> parameters and jitter -> the sample loop (scene term: pinhole, or through the lens)
// -> the ghosts (paraxial quads, or sun hand-over + geometric march) -> the flare layer (starburst,
// falloff, composition).  Both host mirrors of the reference's PathTracer surface --
// pathtracer_amd.cpp (the reference's own class against its unchanged header) and lf_pathtracer.cpp
// (stand-alone types) -- spell that order out HERE, once, and nowhere else.  Header-only, plain C ABI
// calls, no CGL declarations.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "lensflare.h"

enum lf_frame_jitter { LF_FRAME_JITTER_MT19937 = 0, LF_FRAME_JITTER_COUNTER = 1 };
enum lf_frame_ghosts {
  LF_FRAME_GHOSTS_PARAXIAL = 0,   // the reference's quads (lf_generate_ghost_buffer)
  LF_FRAME_GHOSTS_MARCH = 1,      // the geometric march of the loaded prescription (lf_trace_ghosts)
  LF_FRAME_GHOSTS_NONE = 2        // nothing to march towards (no sun in the frame): an empty ghost buffer
};
enum lf_frame_scene {
  LF_FRAME_SCENE_NONE = 0,        // no scene term (lf_set_scene_term(NULL))
  LF_FRAME_SCENE_DEVICE = 1,      // the sample loop on the device (lf_render_scene_term)
  LF_FRAME_SCENE_HOST = 2         // the host evaluated it and has handed it over (lf_set_scene_term(rgb))
};

typedef struct lf_frame_plan {
  int ns_aa;
  double flare_radius, flare_intensity;
  int jitter;                     // lf_frame_jitter
  uint32_t mt_seed;               // MT19937 table: the reference's engine starts at 5489
  uint64_t counter_key;
  int scene;                      // lf_frame_scene
  int lens_camera_mode;           // 0 pinhole; 1 / 2: the scene through the prescription (needs the counter RNG)
  double world_per_mm, exposure;
  int ghosts;                     // lf_frame_ghosts
  int sun_from_flares;            // the march's sun = the in-frame light of the flare state (flare 0)
  float sun_angular_radius;
  int geo_spp;
  uint64_t geo_key;
} lf_frame_plan;

// runs the frame; on failure *failed names the call that failed (a string literal)
static inline lf_status lf_run_frame(lf_ctx* ctx, const lf_frame_plan* p, const char** failed) {
  lf_status st;
#define LF_FRAME_STEP(call)                           \
  do {                                                \
    st = (call);                                      \
    if (st != LF_OK) {                                \
      if (failed) *failed = #call;                    \
      return st;                                      \
    }                                                 \
  } while (0)
  LF_FRAME_STEP(lf_set_params(ctx, p->ns_aa, p->flare_radius, p->flare_intensity));
  if (p->jitter == LF_FRAME_JITTER_COUNTER) LF_FRAME_STEP(lf_set_jitter_counter(ctx, p->counter_key));
  else LF_FRAME_STEP(lf_set_jitter_mt19937(ctx, p->mt_seed, NULL, 0));
  // the sample loop of raytrace_pixel (pathtracer.cpp:841-875)
  LF_FRAME_STEP(lf_set_lens_camera(ctx, p->scene == LF_FRAME_SCENE_DEVICE ? p->lens_camera_mode : 0,
                                   p->world_per_mm, p->exposure));
  if (p->scene == LF_FRAME_SCENE_DEVICE) LF_FRAME_STEP(lf_render_scene_term(ctx));
  else if (p->scene == LF_FRAME_SCENE_NONE) LF_FRAME_STEP(lf_set_scene_term(ctx, NULL));
  // generate_ghost_buffer (pathtracer.cpp:714-817)
  if (p->ghosts == LF_FRAME_GHOSTS_MARCH) {
    if (p->sun_from_flares)
      LF_FRAME_STEP(lf_set_sun_from_flares(ctx, 0, 0.0, p->sun_angular_radius));
    LF_FRAME_STEP(lf_trace_ghosts(ctx, p->geo_spp, p->geo_key));
  } else if (p->ghosts == LF_FRAME_GHOSTS_NONE) {
    LF_FRAME_STEP(lf_clear_ghost_buffer(ctx));
  } else {
    LF_FRAME_STEP(lf_generate_ghost_buffer(ctx));
  }
  // raytrace_starburst + the composition of :891
  LF_FRAME_STEP(lf_render_flare_layer(ctx));
#undef LF_FRAME_STEP
  return LF_OK;
}
