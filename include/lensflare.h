/* lensflare.h -- C ABI of liblensflare_hip.so, the MI355X (gfx950) implementation of the
 * lens-flare hot path of aatifjiwani/lens-flare (src/pathtracer).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ types, no exceptions, every
 * entry point returns an lf_status.  Each function names the reference interface it replaces
 * (file:line relative to the reference checkout).  The reference itself has no C ABI -- its
 * boundary is the public surface of `class PathTracer` inside libpt31.so
 * (CMakeLists.txt:20-33,139-142; src/pathtracer/pathtracer.h:25-143) -- so the C++ shim in
 * lens-flare_amd/host/ maps that surface 1:1 onto these calls.
 *
 * Threading contract (mirrors src/pathtracer/raytraced_renderer.cpp:300-311, :637-646): the
 * set-up and render calls are made from one thread per context; lf_read_* may then be called
 * concurrently from any number of threads.  One context drives one GPU; multi-GPU = one context
 * (and normally one process) per GPU, each rendering its own band of sensor rows.
 */
#ifndef LENSFLARE_H
#define LENSFLARE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LF_ABI_VERSION 1
#define LF_MAX_SURFACES 16   /* interfaces per prescription (incl. the stop) */
#define LF_MAX_LAMBDA 8      /* wavelengths per prescription */
#define LF_MAX_FLARES 8      /* in-frame directional lights */
#define LF_MAX_PAIRS 128     /* ghost pairs per trace call */

typedef struct lf_ctx lf_ctx;

typedef enum {
  LF_OK = 0,
  LF_ERR_INVALID = 1,   /* bad argument (null pointer, size out of range, ...) */
  LF_ERR_NO_DEVICE = 2, /* no gfx950 device / device index out of range */
  LF_ERR_HIP = 3,       /* a HIP runtime call failed; see lf_last_error */
  LF_ERR_STATE = 4,     /* call order violated (e.g. render before set_frame) */
  LF_ERR_OOM = 5
} lf_status;

typedef enum { LF_APERTURE_STARBURST = 0, LF_APERTURE_GHOST = 1 } lf_aperture_slot;

/* CameraApertureTexture (src/pathtracer/camera.h:18-88) */
typedef struct {
  int width, height;
  int min_x, min_y, max_x, max_y; /* bbox of texels > 0 (camera.h:54-72) */
  double total_value;             /* camera.h:61 */
} lf_aperture_stats;

/* device-side counters of the geometric march */
typedef struct {
  uint64_t rays_launched;   /* (sample, wavelength, pair) rays started */
  uint64_t surface_events;  /* ray-surface events (intersect + refract|reflect) of all paths, each path
                               counted on its own -- the legs that paths of one sample share are
                               computed once on the device, see lf_get_executed_events */
  uint64_t rays_clipped_stop;   /* killed by the aperture mask at the stop */
  uint64_t rays_vignetted;      /* outside a semi-aperture or missed a surface */
  uint64_t rays_tir;            /* total internal reflection */
  uint64_t rays_reached_scene;  /* left the front element */
  uint64_t rays_hit_light;      /* ... inside the light's angular lobe (non-zero radiance) */
} lf_counters;

/* ---------------------------------------------------------------- life cycle ------------- */
/* replaces: PathTracer::PathTracer / ~PathTracer (pathtracer.cpp:14-30).  device = HIP ordinal. */
lf_status lf_create(lf_ctx** out, int device);
lf_status lf_destroy(lf_ctx* ctx);
/* text of the last error on this context (never NULL); valid until the next call */
const char* lf_last_error(const lf_ctx* ctx);
int lf_abi_version(void);
/* optional: run every launch on a caller-owned hipStream_t (e.g. torch's current stream) */
lf_status lf_set_stream(lf_ctx* ctx, void* hip_stream);
lf_status lf_synchronize(lf_ctx* ctx);

/* ---------------------------------------------------------------- frame ------------------ */
/* replaces: PathTracer::set_frame_size (pathtracer.cpp:66-69) + clear (:71-79) */
lf_status lf_set_frame(lf_ctx* ctx, int width, int height);
/* multi-GPU sharding: this context renders sensor rows [y0, y1) only (default: all rows).
 * Mirrors the tile queue of raytraced_renderer.cpp:314-328, one band of tile rows per GPU. */
lf_status lf_set_band(lf_ctx* ctx, int y0, int y1);
/* finer multi-GPU sharding for the geometric march: of the band's 8-row sensor tile rows t = y/8
 * this context marches only those with t % period == phase (default period 1 = all).  Dealing
 * tile rows round-robin to the GPUs balances the vignetting-dependent ray survival. */
lf_status lf_set_row_interleave(lf_ctx* ctx, int phase, int period);
/* ... or by BLOCKS of 64 x 64 pixels (round 6): block b (row-major over the frame) belongs to rank b % nranks; a frame
 * narrower than a multiple of 64 has partial blocks at its right and lower edges.  The block is the path cull's: a rank
 * that owns whole blocks builds, audits and reads only its own rows of the cull table -- pre-pass, audit and march all
 * shrink with the number of ranks and the table never crosses a link (where lf_set_row_interleave needs
 * lf_comm_share_cull / lf_set_cull_share and a second collective per frame).  lf_comm_gather / lf_group_gather
 * exchange blocks then.  Pixels, counters summed over the ranks and the gathered frame are the single-GPU frame's.
 * lf_set_row_interleave switches back. */
lf_status lf_set_block_deal(lf_ctx* ctx, int rank, int nranks);
/* replaces the public fields PathTracer::ns_aa, flare_radius, flare_intensity
 * (pathtracer.h:93-94,107) */
lf_status lf_set_params(lf_ctx* ctx, int ns_aa, double flare_radius, double flare_intensity);

/* ---------------------------------------------------------------- inputs ----------------- */
/* replaces: CameraApertureTexture::init (camera.h:26-83) minus the PNG decode: texels are the
 * float values the reference derives from the red channel (CGL/src/color.cpp:16-21).  bbox and
 * total_value are computed on the device. */
lf_status lf_set_aperture(lf_ctx* ctx, lf_aperture_slot slot, const float* texels, int width,
                          int height);
lf_status lf_get_aperture_stats(lf_ctx* ctx, lf_aperture_slot slot, lf_aperture_stats* out);

/* replaces the lens-table globals of pathtracer.cpp:541-556 (Ts, red/green/blue_refr,
 * curvatures) and the constants of trace_ray_auto_* (:619-633): n interfaces, ior[3][n] row-major.
 * Passing NULL arrays restores the reference's hard-coded table. */
lf_status lf_set_paraxial_lens(lf_ctx* ctx, int n, int stop_index, const float* thickness,
                               const float* curvature, const float* ior_rgb);

/* replaces Camera state read by find_sun_pos (camera.h:171-180; camera.cpp:245-273):
 * c2w row-major 3x3, position, fields of view in DEGREES (already fitted to the aspect ratio) */
lf_status lf_set_camera(lf_ctx* ctx, const double c2w[9], const double pos[3], double hfov_deg,
                        double vfov_deg);
/* replaces: PathTracer::find_sun_pos (pathtracer.cpp:32-64).  lights: n x {posLight xyz,
 * radiance rgb} of the scene's DirectionalLights (src/scene/light.h:16-29). */
lf_status lf_find_sun_pos(lf_ctx* ctx, const double* lights, int n_lights);
/* direct write / read of the public fields flare_origins, flare_radiance, axis_ray,
 * angle_to_sun (pathtracer.h:128-135).  origins: n x 2, radiance: n x 3. */
lf_status lf_set_flares(lf_ctx* ctx, int n, const double* origins, const double* radiance,
                        const double axis_ray[2], float angle_to_sun);
lf_status lf_get_flares(lf_ctx* ctx, int* n, double* origins, double* radiance,
                        double axis_ray[2], float* angle_to_sun);

/* sub-pixel jitter of calculate_irradiance_falloff (pathtracer.cpp:1043-1063).
 * MT19937: reproduce the reference's shared std::mt19937 (util/random_util.h:10-22) for a single
 * worker visiting `order` (pixel index x + y*W; NULL = the reference's 32x32 tile order,
 * raytraced_renderer.cpp:314-328,:637-641).  COUNTER: order-free Philox4x32-10 keyed by `key`. */
lf_status lf_set_jitter_mt19937(lf_ctx* ctx, uint32_t seed, const uint32_t* order, size_t n_order);
lf_status lf_set_jitter_counter(lf_ctx* ctx, uint64_t key);

/* optional scene-radiance term (est_radiance_global_illumination averaged as in
 * pathtracer.cpp:841-875) computed by the host: W*H*3 doubles, NULL = zero */
lf_status lf_set_scene_term(lf_ctx* ctx, const double* rgb);

/* ---------------------------------------------------------------- scene term (row f2) ---- */
/* The static scene the sample loop of raytrace_pixel shades (pathtracer.cpp:841-875): replaces
 * PathTracer::bvh / scene (pathtracer.h:116-124) for spheres (scene/sphere.h), triangles
 * (scene/triangle.h) with DiffuseBSDF / EmissionBSDF materials (pathtracer/bsdf.h:122-153,
 * :271-288) and DirectionalLight / PointLight (scene/light.h:16-29, :47-58).
 *   spheres        n x {cx, cy, cz, r}
 *   tri_positions  n x 9 (p1 p2 p3), tri_normals n x 9 (vertex normals n1 n2 n3)
 *   materials      n x {kind, r, g, b}: kind 0 = diffuse reflectance, 1 = emitted radiance,
 *                  2 = diffuse given as the BSDF value f = reflectance / pi (what DiffuseBSDF::f
 *                  returns, bsdf.cpp:52-60: for hosts that can call f() but not read `reflectance`)
 *   lights         n x {type, x, y, z, r, g, b}: type 0 = directional (xyz = dirToLight, unit),
 *                  1 = point (xyz = position); rgb = radiance.  Order = order in scene->lights.
 * Other light and BSDF types are refused (LF_ERR_INVALID): see lf_scene.hip. */
lf_status lf_set_scene(lf_ctx* ctx, int n_spheres, const double* spheres, const int* sphere_material,
                       int n_triangles, const double* tri_positions, const double* tri_normals,
                       const int* tri_material, int n_materials, const double* materials,
                       int n_lights, const double* lights);
/* The lights of the current scene in their general form (replaces the lights lf_set_scene took; the
 * order is the order of scene->lights): n x 16 doubles
 *   {type, r, g, b,  v0 xyz,  v1 xyz,  v2 xyz,  v3 xyz}
 *   type 0 DirectionalLight          v0 = dirToLight                       (scene/light.h:16-29)
 *        1 PointLight                v0 = position                         (:47-58)
 *        2 InfiniteHemisphereLight   (no vectors)                          (:31-44, light.cpp:26-48)
 *        3 AreaLight                 v0 = position, v1 = direction, v2 = dim_x, v3 = dim_y  (:78-97, light.cpp:74-101)
 *        4 EnvironmentLight          (no vectors, rgb unused: the map of lf_set_environment_map;
 *                                    scene/environment_light.cpp:140-171 -- the renderer appends
 *                                    it to scene->lights, raytraced_renderer.cpp:127-128)
 * Types 2, 3 and 4 are SAMPLED lights: lf_set_light_samples(ns_area_light) samples each per hit
 * (PathTracer::ns_area_light, pathtracer.h:108; estimate_direct_lighting_importance,
 * pathtracer.cpp:143-213), drawn from the counter RNG -- they need lf_set_jitter_counter (the
 * reference's shared MT19937 is consumed in hit order, which no parallel schedule reproduces) and are
 * validated statistically against reference frames, not bit for bit. */
/* the box around the scene's primitives (what Application::load asks GLScene::Scene::get_bbox for to
 * place its camera, application.cpp:277-300) and how many there are */
lf_status lf_scene_bounds(lf_ctx* ctx, double bmin[3], double bmax[3], int* n_primitives);
lf_status lf_set_scene_lights(lf_ctx* ctx, int n_lights, const double* rows);
lf_status lf_set_light_samples(lf_ctx* ctx, int ns_area_light);
/* PathTracer::envLight (pathtracer.h:119; `new EnvironmentLight(envmap)`, raytraced_renderer.cpp:
 * 80-84): the environment map.  rgb = HDRImageBuffer::data, w*h texels of 3 doubles, row-major (the
 * .exr decoder stays with the host).  A camera ray that hits nothing returns
 * EnvironmentLight::sample_dir (pathtracer.cpp:291-292; environment_light.cpp:173-182, bilinear
 * in (theta, phi) with its edge rules) -- deterministic, so it is also served in MT19937 parity mode;
 * listed as a light of type 4 it is importance-sampled through the tables of
 * EnvironmentLight::init (:19-59), built here in the reference's order of operations (the
 * probability_debug.png the reference writes as a side effect is not written).  w = h = 0 removes
 * the map.  Maps smaller than 2 x 2 or without any light are refused. */
lf_status lf_set_environment_map(lf_ctx* ctx, int w, int h, const double* rgb);
/* PathTracer::direct_hemisphere_sample (pathtracer.h:114, the -H flag): one_bounce_radiance
 * (pathtracer.cpp:222-232) uses estimate_direct_lighting_hemisphere (:86-138) -- uniform directions
 * over the hemisphere, lights.size() * ns_area_light per hit, collecting the emission of the
 * surfaces they reach -- instead of sampling the lights.  Counter RNG only, statistical parity. */
lf_status lf_set_direct_hemisphere_sample(lf_ctx* ctx, int on);
/* Row f3: a COLLADA file -> the static scene, in one call.  Replaces
 * Collada::ColladaParser::load (src/scene/collada/collada.cpp:131-218) + Application::load
 * (src/application/application.cpp:232-365) + the GLScene -> SceneObjects conversion behind them
 * (gl_scene/mesh.cpp, util/halfEdgeMesh.cpp, scene/object.cpp, gl_scene/ *_light.h): parses the file
 * on the host (lens-flare_amd/host/lf_collada.cpp; triangles, vertex normals, lights and camera come
 * out bit-identical to the reference's, tests/test_collada_loader.py) and hands the result to
 * lf_set_scene / lf_set_scene_lights.  Lights are uploaded in node order: directional, point, area
 * and ambient (= infinite hemisphere) lights.
 *   camera          (may be NULL) what Application::load derives from the file's camera node(s)
 *   sun_lights      (may be NULL) for lf_find_sun_pos: 6 doubles per DirectionalLight, posLight
 *                   xyz + radiance rgb; at most max_sun_lights are written, *n_sun_lights = found
 * Mirror / glass / refraction / microfacet BSDFs are unfilled stubs in the reference (f() = 0, no
 * emission, advanced_bsdf.cpp:17-133) under a direct-lighting-only integrator (pathtracer.cpp:
 * 282-302): such surfaces are black occluders there and are uploaded as exactly that.
 * LF_ERR_INVALID with a message if the file has a spot light (a stub in the reference whose
 * sample_L leaves its outputs uninitialised, light.cpp:64-72) or what the reference itself cannot
 * load (it exits or reads uninitialised memory there).  lf_collada_check answers the
 * same question for a file without a device (0 = renderable; msg receives the reason otherwise). */
typedef struct {
  int present;
  double hfov, vfov, nclip, fclip;
  double pos[3], dir[3], up[3];
} lf_collada_camera;
lf_status lf_load_collada(lf_ctx* ctx, const char* path, lf_collada_camera* camera,
                          double* sun_lights, int max_sun_lights, int* n_sun_lights);
lf_status lf_collada_check(const char* path, char* msg, size_t msg_cap);
/* replaces the public fields samplesPerBatch / maxTolerance (pathtracer.h:112-113) and
 * Camera::nClip / fClip (camera.h:188) */
lf_status lf_set_sampling(lf_ctx* ctx, int samples_per_batch, double max_tolerance, double n_clip,
                          double f_clip);
/* replaces the sample loop of PathTracer::raytrace_pixel (pathtracer.cpp:831-875) for every pixel
 * of the band: ns_aa jittered pinhole rays -> closest hit -> emission + direct lighting, adaptive
 * early-out, divided by the loop variable (ns_aa + 1 without early-out).  The result becomes the
 * scene term lf_render_flare_layer composes.  Pixel jitter follows lf_set_jitter_*. */
lf_status lf_render_scene_term(lf_ctx* ctx);

/* ---------------------------------------------------------------- render ----------------- */
/* replaces: PathTracer::generate_ghost_buffer (pathtracer.cpp:714-817): paraxial trace of the
 * 13 reflection pairs x 3 colours and rasterisation of the 39 textured quads */
lf_status lf_generate_ghost_buffer(lf_ctx* ctx);
/* replaces: PathTracer::raytrace_pixel (pathtracer.cpp:819-899) for every pixel of the band:
 * sample = scene + ghost_buffer + raytrace_starburst (incl. calculate_irradiance_falloff) */
lf_status lf_render_flare_layer(lf_ctx* ctx);

/* ---------------------------------------------------------------- helper members ---------- */
/* The reference declares the steps of generate_ghost_buffer / raytrace_starburst as public members
 * ("Testing functions", pathtracer.h:42-57, :95-101).  A replacement of pathtracer.cpp defines every
 * one of them; these are their device forms (one small launch each; the frame-level calls above do
 * not go through them).  The ghost helpers ADD to the device ghost buffer, triangle by triangle like
 * HDRImageBuffer::update_pixel_additive (util/image.h:145); bbox (may be NULL) receives the pixel
 * rectangle [x0, x1) x [y0, y1) the call can have touched, for a partial read-back.
 *
 * replaces: ghost_buffer.clear() (generate_ghost_buffer, pathtracer.cpp:719; util/image.h:233) */
lf_status lf_clear_ghost_buffer(lf_ctx* ctx);
/* replaces: PathTracer::draw_ghost(string color, float r1, float r2) (pathtracer.cpp:433-508):
 * channel 0 / 1 / 2 = "red" / "green" / any other string; uses axis_ray of the flare state */
lf_status lf_draw_ghost(lf_ctx* ctx, int channel, float r1, float r2, int bbox[4]);
/* replaces: PathTracer::rasterize_textured_triangle (pathtracer.cpp:346-410);
 * v = {x0, y0, u0, v0,  x1, y1, u1, v1,  x2, y2, u2, v2}, colour = ghost_color */
lf_status lf_rasterize_textured_triangle(lf_ctx* ctx, const float v[12], const double colour[3], int bbox[4]);
/* replaces: PathTracer::fill_textured_pixel (pathtracer.cpp:305-343): vertices as given (already
 * y-sorted and shifted by the caller), pixel (x, y) inside the frame */
lf_status lf_fill_textured_pixel(lf_ctx* ctx, const float v[12], int x, int y, const double colour[3]);
/* replaces: PathTracer::shift_vertex (pathtracer.cpp:412-430) */
lf_status lf_shift_vertex(lf_ctx* ctx, float x, float y, float scale, float shift_amount, double out_xy[2]);
/* replaces: PathTracer::compute_phase (pathtracer.cpp:917-931) with complex_exp (:901-915);
 * screen_pos (may be NULL) = the flare origin in pixels the member returns through its reference */
lf_status lf_compute_phase(lf_ctx* ctx, int flare, double u, double v, double out_re_im[2], double screen_pos[2]);
/* replaces: PathTracer::calculate_irradiance_falloff(x, y, radius) (pathtracer.cpp:1043-1063); the
 * 32 jitter draws are pixel (x, y)'s own (lf_set_jitter_*) */
lf_status lf_irradiance_falloff(lf_ctx* ctx, int x, int y, double radius, double rgb[3]);
/* Single-ray forms of the integrator members (pathtracer.h:66-77) on the device scene of
 * lf_set_scene.  ray = {origin xyz, direction xyz, min_t, max_t}.  Sampled lights draw from the
 * counter RNG, stream `seq` (any number; distinct calls should pass distinct values).
 * replaces: PathTracer::est_radiance_global_illumination (pathtracer.cpp:282-302) and the closest-hit
 * query of PathTracer::autofocus (:1065-1072): out = {hit (1/0), t, normal xyz, radiance rgb};
 * without a hit the radiance is the environment's (or 0). */
lf_status lf_scene_trace_ray(lf_ctx* ctx, const double ray[8], uint64_t seq, double out[8]);
/* replaces, for an intersection (t, n, material) the host found itself: what = 0
 * PathTracer::zero_bounce_radiance (:215-220), 1 one_bounce_radiance (:222-232, by
 * lf_set_direct_hemisphere_sample), 2 estimate_direct_lighting_hemisphere (:86-138),
 * 3 estimate_direct_lighting_importance (:142-213).  material = {kind, r, g, b} as lf_set_scene. */
lf_status lf_scene_shade(lf_ctx* ctx, int what, const double ray[8], double t, const double n[3],
                         const double material[4], uint64_t seq, double rgb[3]);

/* ---------------------------------------------------------------- read back -------------- */
/* replaces reads of PathTracer::sampleBuffer / ghost_buffer (util/image.h:139-151).
 * which: 0 = sampleBuffer, 1 = ghost_buffer, 2 = the value of raytrace_starburst(x,y)
 * (pathtracer.cpp:947-1004: starburst + irradiance falloff) on its own, so that a host which
 * computes its scene radiance per pixel can form (scene + ghost) + starburst exactly like
 * pathtracer.cpp:891; 3 = the scene-radiance term itself (what lf_render_scene_term or
 * lf_set_scene_term left: the average of est_radiance_global_illumination over the pixel's camera
 * rays, pathtracer.cpp:841-875; LF_ERR_STATE when there is none; under a row interleave only this
 * context's tile rows are filled -- the term is consumed by the flare layer, not exchanged).  dst receives (x1-x0)*(y1-y0) pixels, each
 * `pixel_stride` doubles apart (3 = packed Vector3D, 4 = the AVX build's 32-byte Vector3D). */
lf_status lf_read_tile(lf_ctx* ctx, int which, int x0, int y0, int x1, int y1, double* dst,
                       size_t pixel_stride);
lf_status lf_read_pixel(lf_ctx* ctx, int which, int x, int y, double rgb[3]);
/* replaces: PathTracer::write_to_framebuffer -> HDRImageBuffer::toColor (util/image.h:208-223,
 * :53-62): RGBA8 of the tile, dst rows `row_stride` uint32 apart */
lf_status lf_write_to_framebuffer(lf_ctx* ctx, int x0, int y0, int x1, int y1, uint32_t* dst,
                                  size_t row_stride);
/* replaces the pixel preparation of RaytracedRenderer::save_image
 * (raytraced_renderer.cpp:739-746): the whole tonemapped frame with rows flipped (PNG is top-down)
 * and alpha forced to 0xFF, W*H uint32 ready for lodepng::encode (the encoder stays with the host) */
lf_status lf_save_image_rgba(lf_ctx* ctx, uint32_t* dst);
/* device pointer of a buffer (for RCCL gathers over xGMI without a host hop);
 * which as in lf_read_tile; layout W*H*3 doubles row-major.  The allocation is padded to a
 * multiple of 64 rows (bytes reports the padded size) so that in-place all-gathers of whole
 * tile-row groups never run past the end. */
lf_status lf_device_buffer(lf_ctx* ctx, int which, void** dptr, size_t* bytes);

/* ---------------------------------------------------------------- multi-GPU -------------- */
/* Replaces the reference's scale-out, which lives in the host: a mutex-guarded queue of 32x32 tiles
 * drained by std::threads (src/pathtracer/raytraced_renderer.cpp:314-328, :352-354, :681-715;
 * src/util/work_queue.h:11-51).  The 8-row sensor tile rows are dealt round-robin to the GPUs
 * (lf_set_row_interleave), every GPU renders its tile rows, and ONE RCCL all-gather per frame (over
 * xGMI) completes the frame on every GPU; there is no other data-path collective.  RCCL is loaded at
 * first use (dlopen): LF_ERR_STATE if it is not installed.
 *
 * One process per GPU: rank 0 calls lf_comm_get_unique_id, the host shares the id (MPI,
 * torch.distributed, a file ...), every rank calls lf_comm_init_rank after lf_set_frame -- it also
 * sets the row interleave (rank, nranks).  lf_comm_gather(which) after rendering: buffer `which`
 * (as lf_read_tile) then holds the whole frame on every rank; it is enqueued on the context's stream. */
#define LF_COMM_ID_BYTES 128
lf_status lf_comm_get_unique_id(unsigned char id[LF_COMM_ID_BYTES]);
lf_status lf_comm_init_rank(lf_ctx* ctx, int nranks, int rank, const unsigned char id[LF_COMM_ID_BYTES]);
lf_status lf_comm_gather(lf_ctx* ctx, int which);
/* The same exchange on the context's second stream, so that it overlaps what the host queues next
 * (a sequence of frames: frame k + 1 is marched while frame k is exchanged).  The main stream waits
 * only until this rank's rows have been copied out; the rows the exchange fills in belong to other
 * ranks.  lf_comm_wait makes the main stream wait for the exchange; lf_synchronize and the read
 * functions (lf_read_tile / _pixel, lf_write_to_framebuffer, lf_save_image_rgba) do so themselves.
 * Until then the buffer's rows of the OTHER ranks are undefined; do not change the frame size, band or
 * interleave while an exchange is pending (those calls drain it first). */
lf_status lf_comm_gather_async(lf_ctx* ctx, int which);
/* What a rank hands to the collective for a communicator of `world` ranks, computed where lf_comm_gather,
 * lf_comm_gather_async and lf_group_gather (RCCL and the peer-copy stand-in alike) compute it: out =
 * {sendcount of the ncclAllGather in elements, bytes per element, byte offset of the receive area in the
 * staging buffer, staging bytes, groups of `world` consecutive tile rows, elements per tile row}.  Rank q's
 * sendcount elements arrive at receive area + q * sendcount.  Inspection for tests and hosts that bring
 * their own transport; nothing is allocated or sent. */
lf_status lf_comm_exchange_plan(lf_ctx* ctx, int world, uint64_t out[6]);
lf_status lf_comm_wait(lf_ctx* ctx);
lf_status lf_comm_destroy(lf_ctx* ctx);
/* LF_OK if RCCL can be loaded here (dlopen + every entry point): what every rank checks, and agrees
 * on over its control plane, BEFORE the first blocking RCCL call */
lf_status lf_comm_available(void);
/* what RCCL reports for the communicator the context is attached to (ncclCommCount,
 * ncclCommUserRank); nranks = 0, rank = -1 without one */
lf_status lf_comm_info(lf_ctx* ctx, int* nranks, int* rank);
/* non-blocking: has the last exchange (lf_comm_gather or _async) finished on the device?  Lets a host
 * wait for the first collective of a new communicator with a deadline instead of blocking in
 * lf_synchronize behind a peer that never joined */
lf_status lf_comm_test(lf_ctx* ctx, int* done);
/* ncclCommAbort: ends the collectives in flight on this rank and drops the communicator, so that the
 * context's streams drain; the recovery from a failed first exchange (every rank calls it) */
lf_status lf_comm_abort(lf_ctx* ctx);
/* bits = 64 (default): tile rows travel as the doubles they are (every rank holds the same bits);
 * 32: as floats -- half the bytes on the wire (SURVEY 8e budgets f32: 12.4 MB per rank at 4K), the
 * rows a rank receives are rounded to float, its own stay as rendered.  Same value on every rank. */
lf_status lf_comm_set_exchange_precision(lf_ctx* ctx, int bits);
/* Giving up on a communicator call that is BLOCKED in another host thread (ncclCommInitRank and a communicator's
 * first collective block until every peer has joined; a peer that died never does).  The thread that waited for
 * the deadline calls lf_comm_poison: from then on the blocked call -- should it ever return -- publishes nothing
 * into the context (a communicator that arrives late is aborted where it stands), every lf_comm_* call on the
 * context returns LF_ERR_STATE, lf_comm_abort still ends what can be ended, and lf_destroy returns LF_ERR_STATE
 * WITHOUT freeing anything while the blocked call has not come back (the context is leaked on purpose: the
 * blocked thread still stands on it).  There is no way back: the process falls back to another exchange or exits. */
lf_status lf_comm_poison(lf_ctx* ctx);
int lf_comm_is_poisoned(lf_ctx* ctx);
/* One process, n devices (a C++ host such as the CGL application): one context + stream per device
 * and one communicator over them (ncclCommInitAll).  Set-up calls go to every context
 * (lf_group_ctx); lf_group_set_frame = lf_set_frame + the round-robin deal on every context;
 * lf_group_for_each runs fn(ctx, rank, user) on one host thread per device, concurrently -- the
 * per-frame sequence (lf_find_sun_pos, lf_trace_ghosts, lf_render_flare_layer ...) goes there, like
 * the reference's worker threads; lf_group_gather exchanges the finished tile rows and returns when
 * every device holds the whole frame.  A device listed twice (a rehearsal on a one-GPU box) cannot
 * join an RCCL communicator: such a group exchanges with peer copies instead. */
typedef struct lf_group lf_group;
typedef lf_status (*lf_group_fn)(lf_ctx* ctx, int rank, void* user);
lf_status lf_group_create(lf_group** out, int n, const int* devices);
lf_status lf_group_destroy(lf_group* g);
int lf_group_size(const lf_group* g);
lf_ctx* lf_group_ctx(lf_group* g, int rank);
const char* lf_group_last_error(const lf_group* g);
lf_status lf_group_set_frame(lf_group* g, int width, int height);
lf_status lf_group_for_each(lf_group* g, lf_group_fn fn, void* user);
lf_status lf_group_gather(lf_group* g, int which);
/* the cull pre-pass of the group's next lf_trace_ghosts(spp) shared between its devices (see lf_set_cull_share): every
 * context builds its slab, one all-gather, every context takes the table over.  Call it before the lf_group_for_each
 * that renders, once per launch; without it every device builds the whole table (until lf_group_share_cull has been
 * used once: from then on a launch without it is refused, like any launch of a host-shared table). */
lf_status lf_group_share_cull(lf_group* g, int spp);
/* the group's frame dealt by blocks of 64 x 64 pixels (lf_set_block_deal on every context; 0: back to tile rows, the
 * default of lf_group_set_frame).  lf_group_share_cull has nothing to do then. */
lf_status lf_group_set_block_deal(lf_group* g, int on);

/* ---------------------------------------------------------------- geometric lens --------- */
/* The north-star path: real ray march through spherical interfaces.  The reference has no
 * counterpart (its lens is paraxial, pathtracer.cpp:511-689; Camera::generate_ray_for_thin_lens
 * is a stub, camera_lens.cpp:22-30), so this replaces `generate_ghost_buffer` when selected.
 *   radius[k]      signed radius of curvature in mm (0 = flat; the stop is flat)
 *   thickness[k]   axial distance vertex k -> vertex k+1 (last: to the sensor)
 *   ior[l*n + k]   index of the medium BEHIND interface k at wavelength l
 *   semi_ap[k]     clear semi-aperture in mm
 * Interface order: scene side first.  */
lf_status lf_set_lens(lf_ctx* ctx, int n_surfaces, int stop_index, int n_lambda,
                      const float* radius, const float* thickness, const float* ior,
                      const float* semi_aperture, float sensor_width_mm);
/* The same from a prescription file (lens-flare_amd/data/ *.lens: `sensor_width_mm w`, then one row
 * per interface `radius thickness n_1 .. n_L semi_aperture`, '#' comments; radius 0 with index 0 is
 * the stop).  Replaces the hard-coded table of pathtracer.cpp:541-556 as an INPUT, which is what
 * lets a host that cannot change the reference's header select a lens (LF_LENS_FILE). */
lf_status lf_load_lens_file(lf_ctx* ctx, const char* path);
/* what lf_set_lens / lf_load_lens_file installed (any pointer may be NULL); efl_mm = lf_paraxial_efl at
 * the middle wavelength (0 for an afocal prescription) */
lf_status lf_get_lens_info(lf_ctx* ctx, int* n_surfaces, int* stop_index, int* n_lambda, float* sensor_width_mm,
                           double* efl_mm);
/* RGB weight of each wavelength (n_lambda x 3); default: identity for n_lambda == 3.
 * RANGE CONTRACT of the march's accumulators (lf_set_lambda_rgb, lf_set_sun, lf_trace_ghosts): a pixel
 * channel is the sum of UNSIGNED 64-bit fixed-point contributions (u64)(v 2^bits), so every weight and every
 * radiance must be finite and >= 0 -- anything else is refused with LF_ERR_INVALID (a host whose colour
 * matrix has negative lobes renders the positive and the negative weights as two frames and subtracts).
 * The magnitude is free: lf_trace_ghosts chooses `bits` per launch -- 36 (a grid of 1.5e-11 per sample)
 * unless the largest sum the launch could produce, spp x paths x (pupil solid angle) x
 * max_c sum_l radiance[c] weights[l][c], would then reach 2^62; in that case the largest exponent that
 * keeps it below (a sun of radiance 1e9 at 1024 spp x 8 wavelengths: 2^17; on the fixed 2^-36 grid of rounds
 * 1-4 such a frame wrapped from a radiance of ~6e7 on).  No sum can wrap, and a
 * frame at 2^k x the radiance is 2^k x the frame, bit for bit, once both leave the default grid.
 * lf_get_march_fix_bits reports the last launch's exponent. */
lf_status lf_set_lambda_rgb(lf_ctx* ctx, const float* weights);
/* the light: unit direction from the lens towards the sun in lens space (z < 0), radiance (finite, >= 0: see
 * the range contract above), angular radius of its (smooth) lobe in radians */
lf_status lf_set_sun(lf_ctx* ctx, const float dir[3], const float radiance[3],
                     float angular_radius);
/* Sun hand-over from the scene to the march: turn in-frame flare `flare` of lf_find_sun_pos /
 * lf_set_flares (normalised screen position flare_origins[flare], radiance flare_radiance[flare];
 * src/pathtracer/pathtracer.cpp:32-64, camera.cpp:245-273) into the lens-space light of
 * lf_set_sun, so that the geometric ghosts and the starburst agree about where the sun is: the
 * direction is the one a lens of focal length efl_mm images at that sensor point,
 *   (  (nx - 1/2) sensor_w / efl,  (ny - 1/2) sensor_w (H/W) / efl,  -1  )  normalised.
 * efl_mm <= 0 takes the paraxial image scale of the prescription at the middle wavelength: where the chief
 * ray of a distant point lands on the sensor per unit field angle -- the focal length (lf_paraxial_efl) while
 * the sensor sits in the focal plane, as in the shipped files, and what a sensor moved by lf_focus_lens really
 * sees otherwise.  Needs lf_set_frame, lf_set_lens and a flare state.  No reference counterpart (the
 * reference's ghosts take only `angle_to_sun`, pathtracer.cpp:50, :735-762). */
lf_status lf_set_sun_from_flares(lf_ctx* ctx, int flare, double efl_mm, float angular_radius);
/* Paraxial effective focal length of a prescription at one wavelength (host arithmetic, no
 * device): the system matrix from the reference's own T / R operators (pathtracer.cpp:527-533);
 * ior_row = the n indices of that wavelength (media BEHIND each interface).  Arguments as
 * lf_set_lens. */
lf_status lf_paraxial_efl(int n_surfaces, int stop_index, const float* radius, const float* thickness,
                          const float* ior_row, double* efl_mm);
/* Paraxial image scale of a prescription AS ITS SENSOR SITS (host arithmetic, no device): the height at
 * which the chief ray of a distant point lands on the sensor per unit field angle -- what
 * lf_set_sun_from_flares(efl_mm <= 0) divides by.  Equal to lf_paraxial_efl when the last thickness puts the
 * sensor in the paraxial focal plane; a file focused elsewhere, or a sensor moved by lf_focus_lens, differs. */
lf_status lf_paraxial_image_scale(int n_surfaces, int stop_index, const float* radius, const float* thickness,
                                  const float* ior_row, double* scale_mm);
/* ghost pairs to enumerate: pairs = n x {i, j} interface indices (i < j, neither the stop);
 * i = j = -1 is the primary (no reflection) path.  n = 0 / NULL = all glass pairs. */
lf_status lf_set_ghost_pairs(lf_ctx* ctx, const int* pairs, int n_pairs, int include_primary);
/* Part of the sampling specification of lf_trace_ghosts (no reference
 * counterpart): each pupil stratum is split into 2^bits x 2^bits sub-cells and all 64 pixels of an
 * 8x8 sensor tile aim sample s at the same, randomly drawn sub-cell (coherent fate at the aperture
 * mask).  bits = 0 draws every pixel's pupil point independently inside its stratum; every value
 * is an unbiased estimator of the same image (tests/test_gpu_march_f64.py).  0..8; default 6 (64 x 64
 * sub-cells) since round 4 -- rounds 1-3 shipped 2 (4 x 4): a frame of those rounds is reproduced by
 * lf_set_pupil_subcells(2) + lf_set_tile_stride(1). */
lf_status lf_set_pupil_subcells(lf_ctx* ctx, int bits);
/* Part of the sampling specification (no reference counterpart): the disc every sensor sample aims its
 * pupil point at -- radius and axial position z in the prescription's coordinates (z = 0 at the first
 * vertex, growing towards the sensor).  Default (radius <= 0): the rear element's clear aperture at its
 * vertex plane, which is valid for EVERY path but wastes the samples that miss the exit pupil.  A
 * smaller disc is an unbiased estimator only for paths whose first crossing of the stop precedes their
 * first reflection (the primary path and the pairs (i, j) with i in front of the stop): the start
 * weight is the disc's solid angle, so nothing else changes.  Pairs with both mirrors behind the stop
 * must keep the default.  lf_aim_at_exit_pupil sets the disc to the paraxial image of the stop's open
 * part (the circle around the mask's non-zero texels) through the rear group, times `margin` (> 1
 * leaves room for pupil aberration off the axis).  lf_paraxial_exit_pupil: that image for a
 * prescription and one wavelength (host arithmetic with the reference's T / R operators,
 * pathtracer.cpp:527-533): its z and its lateral magnification. */
lf_status lf_set_pupil_target(lf_ctx* ctx, float radius_mm, float z_mm);
lf_status lf_get_pupil_target(lf_ctx* ctx, float* radius_mm, float* z_mm, float* z_sensor_mm);
lf_status lf_aim_at_exit_pupil(lf_ctx* ctx, float margin);
lf_status lf_paraxial_exit_pupil(int n_surfaces, int stop_index, const float* radius, const float* thickness,
                                 const float* ior_row, double* z_mm, double* magnification);
/* on: lf_trace_ghosts ADDS its pixels to ghost_buffer instead of replacing them (a frame composed of
 * launches with different pair sets / pupil targets).  Default off. */
lf_status lf_set_ghost_accumulate(lf_ctx* ctx, int on);
/* Which 64 pixels a wave of the march takes (part of the sampling specification, like the pupil
 * sub-cells: it decides which pixels share a sub-cell draw): 8 rows x 8 columns that are `stride` (1, 2, 4
 * or 8) apart; `stride` such waves interleave inside a block of 8 * stride columns.  stride 1 (the default
 * of rounds 1-3) = an 8 x 8 block of adjacent pixels, whose shared sub-cell correlates the noise of
 * neighbouring pixels (variance of an 8 x 8 tile mean = 38x independent pixels at 4 x 4 sub-cells);
 * stride 8 (THE DEFAULT since round 4) spreads a wave's pixels over 64 columns, so that the correlated noise
 * lands on pixels 8 apart (7.4x with the 64 x 64 sub-cells that are the default beside it).  Per-pixel
 * expectation and variance do not depend on it; the tile rows (multi-GPU deal) stay 8 rows.  The same key
 * gives DIFFERENT pixels under a different stride / sub-cell setting: a host or an oracle that wants the
 * frames of rounds 1-3 calls lf_set_tile_stride(1) + lf_set_pupil_subcells(2). */
lf_status lf_set_tile_stride(lf_ctx* ctx, int stride);
/* march `spp` sensor samples per pixel of the band through every selected pair and wavelength and
 * accumulate into ghost_buffer (replacing its content).  key seeds the counter RNG. */
lf_status lf_trace_ghosts(lf_ctx* ctx, int spp, uint64_t key);
/* PATH CULLING (round 5; no reference counterpart -- the reference enumerates 13 fixed pairs per channel and
 * draws each as one quad, pathtracer.cpp:735-762, :452-508).  lf_trace_ghosts does not start a path where it
 * cannot carry light: a pre-pass bounds, for every block of 64 x 64 sensor pixels (128 x 128 where that is still
 * <= 1.25 mm on the sensor and the launch has few samples: 4K below 256 spp x 3 wavelengths; 32 or 16 where 64 pixels
 * are more than 1.8 mm: frames narrower than 1280 pixels on 36 mm), every cell of the pupil square and every selected
 * path, where the rays of that 4-D box can go -- on each diaphragm of the path and in direction space at the exit --
 * and the march starts only the paths whose box may end inside the sun's lobe.  A path that is not started adds
 * exactly 0 IF the bound is right, and then ghost_buffer is the buffer of the full enumeration bit for bit.
 * WHAT STANDS BEHIND THE BOUND: it is a second-order estimate from 15 marched rays per box (DESIGN.md section 5 states the
 * assumption: the third order of the bundle's map is small against the measured second order away from the edges
 * where a ray ends, and boxes touching such an edge are never dropped), checked against the full enumeration on whole
 * frames and on 39 000 random frames of three design families (double Gauss, Cooke triplet, their 8-wavelength forms:
 * scaled 0.5 .. 2, bent 8 %, refocused, suns of 0.17 .. 17 degrees over the field, blocks of 0.6 .. 1.8 mm) -- and, on
 * EVERY launch, audited: lf_set_cull_audit below.  A host that wants the enumeration itself: mode 0.
 * lf_counters / lf_get_executed_events count the rays that were started.
 *   mode 1 (default): on; the table is reused while lens, pairs, sun, frame, pupil disc, mask and sample
 *                     count are unchanged.   2: on, rebuilt at every lf_trace_ghosts.
 *   mode 0: off -- every sample marches every selected path (rounds 1-4; the shared-leg path tree).
 * Applies to at most 128 paths and 4096 samples per pixel; beyond, lf_trace_ghosts marches everything (lf_get_cull_reason says why).
 * lf_get_cull_info: {mode, did the last lf_trace_ghosts cull, blocks_x, blocks_y, cells per block (the pupil
 * strata G x G), G, pre-pass cells per axis, block size in pixels}.  lf_get_cull_table: the masks the last
 * launch used, [blocks_y * blocks_x][cells + 1] (block side = info[7] pixels; bit q = path q of the selection in lf_set_ghost_pairs order,
 * the primary path first; entry `cells` = the union, used by the unstratified samples s >= G * G), so that a
 * checker can march exactly what the device marched. */
lf_status lf_set_march_culling(lf_ctx* ctx, int mode);
lf_status lf_get_cull_info(lf_ctx* ctx, int info[8]);
lf_status lf_get_cull_table(lf_ctx* ctx, uint64_t* out, size_t n_entries);
/* of all (block, cell, path) combinations, the fraction the last pre-pass found able to carry light.  Above
 * 0.10 + 1.6 / paths (0.135 with 46 paths, 0.18 with 21) lf_trace_ghosts marches everything after all (the path tree
 * shares legs and lets rays die early; the culled march starts each path alone): a very wide sun or a handful of
 * samples per pixel. */
lf_status lf_get_cull_started_fraction(lf_ctx* ctx, double* fraction);
/* WHY the last lf_trace_ghosts did, or did not, cull (informational: lf_trace_ghosts returns LF_OK either way and the
 * pixels are the same -- what changes is the time: the path tree marches every path of every sample). */
typedef enum {
  LF_CULL_APPLIED = 0,              /* the culled march ran */
  LF_CULL_OFF = 1,                  /* lf_set_march_culling(0) */
  LF_CULL_NO_STOP = 2,              /* the prescription has no stop: nothing bounds a pupil cell */
  LF_CULL_TOO_MANY_PATHS = 3,       /* more than 128 selected paths */
  LF_CULL_TOO_MANY_SAMPLES = 4,     /* more than 4096 samples per pixel (64 x 64 strata) */
  LF_CULL_BLOCK_TOO_LARGE = 5,      /* even a block of 16 x 16 pixels is more than 1.8 mm on this sensor */
  LF_CULL_TABLE_TOO_FULL = 6,       /* the table starts more than 0.10 + 1.6 / paths of everything: the path tree is faster */
  LF_CULL_AUDIT_REFUTED = 7,        /* an audit ray of a dropped box reached the light: the table was not used (see below) */
  LF_CULL_DISPERSION = 8            /* the index columns are not monotonic in the wavelength: the two ends of the spectrum
                                       do not bracket what lies between, the pre-pass's dispersion bound does not hold */
} lf_cull_reason;
lf_status lf_get_cull_reason(lf_ctx* ctx, int* reason);
/* THE AUDIT.  The pre-pass's bounds are second-order estimates from 15 rays per box: conservative on every
 * frame anyone has compared with the full enumeration, but not a proof.  Every table is therefore checked where it is
 * used: for each (block, cell, path) combination it does NOT start, `rays_per_dropped_box` rays of that box (a random
 * point of the block, a random point of the pupil cell, one of the launch's wavelengths) are marched with the march's
 * own events; one that ends inside the sun's lobe refutes the table -- the launch marches every path of every sample
 * instead (same pixels as always, the path tree's time), lf_get_cull_reason says LF_CULL_AUDIT_REFUTED and the counters
 * below say how often.  Default 1 ray (0.9 ms on the 1080p bench frame); 0 switches the audit off.
 * lf_get_cull_audit: rays marched / rays that reached the light / launches refuted, since lf_reset_counters. */
lf_status lf_set_cull_audit(lf_ctx* ctx, int rays_per_dropped_box);
lf_status lf_get_cull_audit(lf_ctx* ctx, uint64_t* rays, uint64_t* lit, int* launches_refuted);
/* TEST HOOK -- not for hosts.  The switches the test suite and bench.py's A/B legs need, by name (no environment
 * variable changes what this library computes):
 *   "cull_force" 0/1            keep the culled march whatever the table starts (small test frames)
 *   "cull_weights_first" 0/1    the Fresnel / mask weight on EVERY executed event (SURVEY 8d's unit), same pixels
 *   "cull_no_prefix" 0/1        every started path marched alone from the sensor (round 5's culled march), same pixels
 *   "cull_general_kernel" 0/1   build the table with the kernel that takes its rules as arguments (must give the shipped one's table)
 *   "cull_strict", "cull_strict_lost", "cull_slack", "cull_keep_partial", "cull_disable", "cull_margin", "cull_lobe_k",
 *   "cull_lost_rel", "cull_lost_abs"   the pre-pass rules round 5 REPLACED (they lose lit rays on some prescriptions:
 *                               what the audit is shown to catch, tests/test_gpu_cull.py)
 *   "scene_compact" -1/0/1      k_scene_lens's ray compaction: by the tree's size / off / on
 *   "scene_lens_strided" -1/0/1 k_scene_lens's wave tile: by the tree's size / 8 x 8 neighbours / the march's strided tile
 *   "bvh_median" 0/1, "bvh_leaf" 1..4   the scene tree of the next lf_set_scene: median splits (round 2), primitives per leaf
 *   "comm_force_exchange" 0/1   run the collectives with a single rank as well
 *   "march_tail_tiles" n, "march_tail_groups" g   the culled march splits its LAST n wave tiles over g workgroups each so that a
 *                               launch ends on short workgroups (-1 / -1: the default, half a round of resident workgroups x 4;
 *                               0 / 1: no split); same pixels and counters whatever the values
 * Unknown names are refused.  ctx == NULL: the value becomes the default of every context created afterwards (a
 * test that cannot reach the contexts a helper creates); 0 takes the default away. */
lf_status lf_test_knob(lf_ctx* ctx, const char* name, double value);
/* THE PRE-PASS OF A MULTI-GPU FRAME, SHARED.  The table covers the whole frame (a block of 64 sensor rows holds tile
 * rows of every rank), so N ranks that each build all of it spend the same time on it as one GPU does: the part of a
 * frame that does not shrink with N.  Shared, rank r builds the rows of the blocks b with b % N == r (dealt round
 * robin: what a block costs depends on the ghosts that cross it), the rows of one rank lie together (its slab; equal
 * slabs, the last ones padded) and ONE all-gather per table completes it on every rank -- the second and last
 * collective of a frame.  Every rank must make the same launches with the same inputs (lens, pairs, sun, frame,
 * mask, sample count): the exchange is collective.  ghost_buffer, counters and lf_get_cull_table (handed out in block
 * order whatever the layout) are those of a table built alone, bit for bit.
 *   lf_comm_share_cull(ctx, 1)     after lf_comm_init_rank: lf_trace_ghosts completes the table itself, by an
 *                                  ncclAllGather on the communicator's stream (in place, between pre-pass and march).
 *   lf_set_cull_share(rank, N)     the HOST owns the exchange (no usable RCCL communicator: torch.distributed, MPI,
 *                                  a rehearsal on one device): per launch lf_cull_prepare(spp) builds this rank's slab,
 *                                  lf_cull_table_view hands out the device table {pointer, entries, entries per rank}
 *                                  (all zero if this launch does not cull) for an in-place all-gather of slabs in rank
 *                                  order, lf_cull_commit takes the completed table over, and the lf_trace_ghosts that
 *                                  follows (same spp, same inputs) marches with it; a launch without them is refused.
 *                                  (1 rank: off.) */
lf_status lf_comm_share_cull(lf_ctx* ctx, int on);
lf_status lf_set_cull_share(lf_ctx* ctx, int rank, int nranks);
lf_status lf_cull_prepare(lf_ctx* ctx, int spp);
lf_status lf_cull_table_view(lf_ctx* ctx, void** device_ptr, uint64_t* entries, uint64_t* entries_per_rank);
lf_status lf_cull_commit(lf_ctx* ctx);
/* the fixed-point exponent the last lf_trace_ghosts used (36 unless the range contract above lowered it) */
lf_status lf_get_march_fix_bits(lf_ctx* ctx, int* bits);
/* replaces: LensCamera::generate_ray of the north star / Camera::generate_ray_for_thin_lens
 * (declared camera.h:168, a stub in camera_lens.cpp:22-30), batched: for n sensor samples -- sensor
 * position in mm (x, y; the lens inverts the image) and a rear-pupil sample in [-1,1]^2 -- march the
 * primary path through the prescription at wavelength index `lambda`.  out: n x 8 floats
 * {origin xyz on the front element, unit direction xyz towards the scene, transmitted weight,
 * alive (1/0)} in lens space (optical axis = z, light travels +z, the scene lies at z < 0). */
lf_status lf_generate_lens_rays(lf_ctx* ctx, int lambda, size_t n, const float* sensor_xy_mm,
                                const float* pupil_uv, float* out);
/* How the flare layer evaluates the two powers of the reference's starburst / falloff code --
 * `radiance / pow(r, 1.5)` (calculate_irradiance_falloff, pathtracer.cpp:1056) and `pow(factor, 8.0)`
 * (raytrace_starburst, :982):
 *   1  exact: the double-precision pow the reference calls (the device's libm, <= 1 ulp like glibc's)
 *   2  fast:  rsqrt(r^3) and three squarings (~3 and ~4 ulp from exact; 0.34 -> 0.15 ms per 1080p frame)
 *   0  auto (default): exact in MT19937 parity mode -- the mode that exists to reproduce the reference's
 *      frames, where agreement must hold by construction -- fast with the counter RNG
 * Either way the result is far inside the north star's 1e-4 (and the parity tests' 1e-9). */
lf_status lf_set_flare_arithmetic(lf_ctx* ctx, int mode);
/* ---- the lens camera of the scene term (round 4) -----------------------------------------------
 * replaces: the call `camera->generate_ray(x, y)` in the sample loop of PathTracer::raytrace_pixel
 * (src/pathtracer/pathtracer.cpp:841-850) -- a pinhole in the reference (camera.cpp:278-305; its
 * thin-lens variant is a stub, camera_lens.cpp:22-30, declared camera.h:168) -- by the north star's
 * "for each sensor sample, march a ray through the lens prescription ... accumulate radiance into
 * the sensor buffer": with mode != 0 lf_render_scene_term starts sample s of pixel (x, y) exactly as
 * lf_trace_ghosts does (the same counter-RNG block, sensor point, pupil stratum and sub-cell; key =
 * lf_set_jitter_counter's, strata for ns_aa samples), marches its primary path N-1 .. 0 through the
 * prescription with the Fresnel / aperture weight (the arithmetic of lf_generate_lens_rays), carries
 * the exit ray into the scene by the camera's pose (lens space = camera space: x right, y up, scene
 * at -z; lens millimetres times world_per_mm; the centre of the paraxial entrance pupil sits at the
 * camera position) and weights the radiance it finds by exposure x transmitted weight.  A sample the
 * lens blocks contributes 0 and still counts in the mean (the division by the loop variable,
 * pathtracer.cpp:875, is unchanged).  The loop visits the march's samples 0 .. ns_aa-1 in a scattered
 * order (iteration i takes sample (i * step) mod ns_aa, step = the integer nearest ns_aa / 1.618 that is
 * coprime to ns_aa) so that the reference's adaptive early-out (:862-868) stops on a prefix that covers
 * the pupil, not on its rim.
 *   mode 0  off: the reference's pinhole camera (default)
 *   mode 1  one ray per sample at the reference wavelength (index n_lambda / 2) carries R, G and B
 *   mode 2  one ray per wavelength; channel c collects lambda_rgb[l][c] of wavelength l's radiance
 *           (lateral and axial colour show; n_lambda closest-hit searches per sample)
 * exposure <= 0 calibrates: 1 / (mean transmitted weight of the on-axis sensor point over a 64 x 64
 * grid of pupil points, reference wavelength), so that a scene of uniform radiance L renders as L at
 * the frame's centre -- what the reference's pinhole returns everywhere.  The calibration and the
 * interface table follow later changes of the lens, the stop mask and the pupil target.
 * Needs lf_set_lens / lf_load_lens_file, the stop mask (LF_APERTURE_STARBURST slot), lf_set_camera
 * and the counter RNG (MT19937 parity mode is refused: its table has no pupil draws).
 * lf_get_lens_camera: the settings in force (after calibration) and the entrance pupil's z (mm). */
lf_status lf_set_lens_camera(lf_ctx* ctx, int mode, double world_per_mm, double exposure);
lf_status lf_get_lens_camera(lf_ctx* ctx, int* mode, double* world_per_mm, double* exposure,
                             double* entrance_pupil_z_mm);
/* Where the lens camera's samples aim.  margin <= 0 (default): at the march's disc (the rear element's clear
 * aperture, or lf_set_pupil_target's) -- the scene ray of a sample IS the march's primary path, and with a
 * pentagon stop about a quarter of the samples leaves the lens.  margin > 0: at the paraxial image of the
 * stop's open part through the rear group, times margin (as lf_aim_at_exit_pupil computes it; > 1 leaves room
 * for the pupil's aberration off the axis) -- an unbiased estimator for the primary path, the only path the
 * lens camera marches, under which more of the samples pass (1.2-1.8x with a pentagon stop, whose area is 76 %
 * of its circumscribed circle); the march's own sampling is not touched.  Same expectation; the exposure
 * calibration follows. */
lf_status lf_set_lens_camera_aim(lf_ctx* ctx, float margin);
/* the paraxial image of the stop's centre through the interfaces IN FRONT of it (host arithmetic with
 * the reference's T / R operators, pathtracer.cpp:527-533): its z in lens space (the front vertex is
 * z = 0, the scene at z < 0; typically a few mm > 0, inside the lens) and its lateral magnification */
lf_status lf_paraxial_entrance_pupil(int n_surfaces, int stop_index, const float* radius, const float* thickness,
                                     const float* ior_row, double* z_mm, double* magnification);
/* replaces: Camera::focalDistance (camera.h:174; the -d flag) for a real lens: moves the sensor -- the
 * prescription's last thickness -- to the paraxial image (reference wavelength) of an axial object
 * object_distance_mm in front of the first vertex; <= 0 or infinite: focus at infinity.  The pair
 * selection, wavelength weights and a still-valid pupil target survive; march program, lens-camera
 * table and calibration are rebuilt on next use.  sensor_distance_mm: the new last thickness. */
lf_status lf_focus_lens(lf_ctx* ctx, double object_distance_mm, float* sensor_distance_mm);
/* the same with the distance measured from the CAMERA POSITION of the lens camera -- the centre of the paraxial
 * entrance pupil (lf_paraxial_entrance_pupil; 19.95 mm behind the double Gauss's first vertex) -- which is what
 * Camera::focalDistance means (camera.h:174: the drop-in passes it here) */
lf_status lf_focus_lens_from_pupil(lf_ctx* ctx, double distance_from_entrance_pupil_mm, float* sensor_distance_mm);
/* replaces: BVHAccel::total_rays / total_isects (src/scene/bvh.h:85,105,136; bvh.cpp:211), the numbers
 * behind the reference's end-of-frame log (raytraced_renderer.cpp:706-709), as the device's scene
 * kernel counts them since the last reset: out = {rays handed to the closest-hit / occlusion search
 * (camera + shadow + hemisphere rays), primitive tests of closest-hit searches, lens-camera samples
 * started (one per sample and traced wavelength), of those that left the front element} */
lf_status lf_get_scene_counters(lf_ctx* ctx, uint64_t out[4]);
lf_status lf_reset_scene_counters(lf_ctx* ctx);
/* Host logic of the march, inspectable WITHOUT a device (used by the CPU tests): builds what
 * lf_trace_ghosts uploads for a prescription and a pair selection (arguments as lf_set_lens /
 * lf_set_ghost_pairs; pairs = NULL selects every glass pair).
 *   info   int[8 + 4 (LF_MAX_PAIRS + 1)]: {n_paths, flat rows per wavelength, first program row,
 *          program rows per wavelength, total rows, jump-table entries, 0, 0} then per path
 *          {i, j, first flat row, flat rows}
 *   rows   8 x 4 bytes per row: dzv (vertex z of the interface the ray comes from, or of the sensor,
 *          minus this interface's), curv, h2, eta, sgn (floats), flags (int), radius, eta^2 (floats);
 *          n_lambda x flat sequences, then n_lambda x the path-tree program, then one spare row
 *   skip   one int per program row: (rows to jump << 2) | state to restore, for a wave that is dead
 * flags: 1 mirror, 2 stop, 4 flat, 8 restore slot 1 after END, 0x10 / 0x20 park in slot 0 / 1
 * before the event, 0x40 END of a path, 0x80 restore slot 0 after END; bits 8-15 run length,
 * 16-23 number of paths sharing the row, 24-31 the path an END row completes.
 * rows / skip may be NULL to query the sizes (info[4], info[5]) first. */
lf_status lf_march_tables(int n_surfaces, int stop_index, int n_lambda, const float* radius,
                          const float* thickness, const float* ior, const float* semi_aperture,
                          const int* pairs, int n_pairs, int include_primary, int* info,
                          float* rows, size_t rows_cap, int* skip, size_t skip_cap);
/* Spectral starburst (no reference counterpart -- the reference's
 * starburst, pathtracer.cpp:947-1000, is monochrome).  n = 0 restores the reference behaviour.
 * Wavelength l sees the reference's diffraction pattern magnified by 1 / scale[l]
 * (scale = lambda_ref / lambda_l), shaped like the reference's value and added with the weights
 * rgb_weights[3 l + {0,1,2}]; n <= LF_MAX_LAMBDA.  One wavelength with scale 1 and weights
 * (1,1,1) is the reference formula.  Takes effect at the next lf_render_flare_layer. */
lf_status lf_set_starburst_spectrum(lf_ctx* ctx, int n, const double* scale, const double* rgb_weights);
/* y[k] = the square root of x[k] exactly as the march computes it (the CDNA4 v_sqrt_f32
 * instruction, 1 ulp).  Host pointers.  No reference counterpart: the parity tests measure the
 * instruction's deviation from the correctly rounded root with this call and hand it to the CPU
 * oracle, which then follows the device bit for bit (oracle/lf_geo_oracle.c, geo_set_sqrt_table). */
lf_status lf_native_sqrt(lf_ctx* ctx, const float* x, float* y, size_t n);
/* y[k] = the reciprocal of x[k] exactly as the march computes it (v_rcp_f32, 1 ulp): its divisions on
 * the per-event and per-sample path are multiplications by this reciprocal (round 4).  As for the root,
 * the parity tests measure its deviation from the correctly rounded reciprocal -- a function of the
 * significand alone -- and hand it to the CPU oracle (geo_set_rcp_table). */
lf_status lf_native_rcp(lf_ctx* ctx, const float* x, float* y, size_t n);
lf_status lf_get_counters(lf_ctx* ctx, lf_counters* out);
lf_status lf_reset_counters(lf_ctx* ctx);
/* Ray-surface events the device actually computed since the last reset: the paths of one sensor
 * sample start with the same backward leg and the pairs (i, .) share the forward leg after the
 * reflection at i, and the march computes every shared leg once (no reference counterpart). */
lf_status lf_get_executed_events(lf_ctx* ctx, uint64_t* out);
/* What the counted events cost in full (no reference counterpart): the march's first pass computes
 * intersection + refraction / reflection for every event; the Fresnel / aperture WEIGHT of an event is
 * computed by marching a finished path a second time, and only for waves in which a lane reached the
 * light's lobe.  out = {executed events (= lf_get_executed_events), events of those second marches
 * counted for the lanes that needed them (= events that ever had a Fresnel factor evaluated), rows of
 * those second marches (each row is one event for all 64 lanes of a wave), 0}.  The second marches are
 * NOT part of `executed events`. */
lf_status lf_get_march_stats(lf_ctx* ctx, uint64_t out[4]);

/* ---------------------------------------------------------------- measurement ------------ */
/* HIP-event timing of the kernels launched since the last reset, on the context's stream.
 * kernel: "march", "flare_layer", "ghost_raster", "dft", "frame_setup", "tonemap", "scene_term",
 * "exchange" (pack -> all-gather -> unpack of lf_comm_gather / _async, on the stream it runs on). */
lf_status lf_timing_enable(lf_ctx* ctx, int on);
lf_status lf_timing_reset(lf_ctx* ctx);
lf_status lf_timing_get(lf_ctx* ctx, const char* kernel, int* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* LENSFLARE_H */
