"""Path culling of the geometric march (lens-flare_amd/csrc/lf_cull.hip, round 5; no reference counterpart -- the
reference draws 13 fixed pairs per channel as quads, src/pathtracer/pathtracer.cpp:735-762).  lf_trace_ghosts starts
only the paths a pre-pass found able to carry light from the sun to a block of the sensor through a cell of the
pupil.  What must hold, and is held here:

  * ghost_buffer = the buffer of the FULL enumeration, bit for bit -- against the path-tree kernel that marches
    every path of every sample (lf_set_march_culling(0)) and against the float32 oracle, which always marches
    everything; the oracle also reports how many rays the device's table skips that did reach the light: none;
  * counters = the oracle's, restricted to the rays the device started (it reads the device's own table);
  * on frames where the table would start most of everything the launch takes the path tree by itself.

The frames the oracle can afford are small.  A cull block must be small ON THE SENSOR (<= 1.8 mm: lf_cull_block_log2), so
the small frames here are CROPS: the bench's pixel pitch (36 mm / 1920) on a narrower sensor around the axis.  There the
table starts more than the 12 % above which a launch takes the path tree by itself; the test knob cull_force
(lf_test_knob, a hook of lfk_march) keeps the culled kernel.

Round 6: the pre-pass kernel that ships (one rule set, a footprint only where a test can fire) builds the table of the
kernel that takes its rules as arguments, bit for bit; every table is AUDITED (a ray per dropped box: lf_set_cull_audit)
-- shown on the frames that lost light under the rules round 5 replaced; frames of 960 and 640 pixels cull with blocks of
32 and 16; no environment variable changes what the library computes."""
import numpy as np
import pytest

from goldenlib import load_texels, ray_budget
from oracle import lfo

pytestmark = pytest.mark.gpu
RAD = [1.0, 0.9, 0.5]


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="module")
def lf(pkg):
    ctx = pkg.LensFlare(0)
    lfo.geo_follow_device(ctx)
    yield ctx
    lfo.geo_follow_device(None)
    ctx.close()


@pytest.fixture()
def forced(pkg, lf):
    """the culled march whatever the table starts: on the module's context and on every context created meanwhile"""
    lf.test_knob("cull_force", 1)
    pkg.test_knob_default("cull_force", 1)
    yield
    lf.test_knob("cull_force", 0)
    pkg.test_knob_default("cull_force", 0)


def _crop(lens, W, pitch_of=1920):
    """the prescription on a sensor W pixels wide at the pixel pitch of a 36 mm sensor `pitch_of` pixels wide"""
    c = dict(lens)
    c["sensor_width_mm"] = 36.0 * W / pitch_of
    return c


def _setup(pkg, lf, lens, W, H, sun, alpha, mask, pairs=None, primary=True, lambda_rgb=None):
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    if lambda_rgb is not None:
        lf.set_lambda_rgb(lambda_rgb)
    lf.set_sun(sun, RAD, alpha)
    lf.set_ghost_pairs(pairs, primary)
    lf.set_band(0, H)
    lf.set_row_interleave(0, 1)


def _both(pkg, lf, spp, key):
    """-> culled pixels + counters + table info, full-enumeration pixels + counters"""
    lf.set_march_culling(2)
    lf.reset_counters()
    lf.trace_ghosts(spp, key)
    info = lf.cull_info()
    g1, c1 = lf.read_buffer(pkg.GHOST_BUFFER), lf.counters()
    lf.set_march_culling(0)
    lf.reset_counters()
    lf.trace_ghosts(spp, key)
    g0, c0 = lf.read_buffer(pkg.GHOST_BUFFER), lf.counters()
    assert not lf.cull_info()["culled"]
    lf.set_march_culling(1)
    return g1, c1, info, g0, c0


@pytest.mark.parametrize("W,H,spp,sun,alpha", [
    (96, 64, 64, [0.03, 0.02, -1.0], 0.05), (130, 70, 16, [-0.06, 0.04, -1.0], 0.05), (64, 64, 256, [0.0, 0.0, -1.0], 0.02),
    (48, 32, 5, [-0.02, 0.01, -1.0], 0.05), (40, 24, 300, [0.001, -0.0015, -1.0], 0.004), (72, 40, 1, [0.03, 0.02, -1.0], 0.1),
    (96, 64, 36, [0.1, 0.02, -1.0], 0.05)])
def test_culled_march_is_the_full_enumeration_and_the_oracle(pkg, lf, forced, W, H, spp, sun, alpha):
    lens = _crop(pkg.load_lens_file("dgauss11.lens"), W)
    mask = load_texels("pentbig500_14.png")
    _setup(pkg, lf, lens, W, H, sun, alpha, mask)
    key = 0xC011 + spp
    g1, c1, info, g0, c0 = _both(pkg, lf, spp, key)
    assert info["culled"] and info["block_px"] == 64 and info["blocks_x"] == (W + 63) // 64
    assert np.array_equal(g1, g0)                                   # nothing that was skipped could have contributed
    assert c0["rays_launched"] == W * H * spp * 3 * 46 and 0 < c1["rays_launched"] <= c0["rays_launched"]
    assert c1["rays_hit_light"] == c0["rays_hit_light"]             # every lit ray was started
    assert c1["rays_launched"] == c1["rays_clipped_stop"] + c1["rays_vignetted"] + c1["rays_tir"] + c1["rays_reached_scene"]
    assert lf.executed_events() > 0
    # the oracle: pixels of the full enumeration, counters of what the device's table starts
    lf.set_march_culling(2)
    lf.reset_counters()
    lf.trace_ghosts(spp, key)
    assert 0 < lf.executed_events() <= c1["surface_events"]         # (the started paths of a sample share their common leg)
    # ... and every started path marched alone from the sensor (round 5's march, lf_test_knob): the same frame and counters
    lf.test_knob("cull_no_prefix", 1)
    try:
        lf.reset_counters()
        lf.trace_ghosts(spp, key)
        assert np.array_equal(lf.read_buffer(pkg.GHOST_BUFFER), g1) and lf.counters() == c1
        assert lf.executed_events() == c1["surface_events"]
    finally:
        lf.test_knob("cull_no_prefix", 0)
    lf.reset_counters()
    lf.trace_ghosts(spp, key)
    og, oc = lfo.geo_trace(lens, W, H, 0, H, spp, key, None, True, mask, sun, RAD, alpha, n_threads=16)
    assert lfo.last_culled_lit == 0
    assert np.array_equal(g1, og) and oc == c1
    og0, oc0 = lfo.geo_trace(lens, W, H, 0, H, spp, key, None, True, mask, sun, RAD, alpha, n_threads=16, cull=None)
    assert np.array_equal(og0, og) and oc0 == c0                    # (and the path tree's counters are the full oracle's)
    lf.set_march_culling(1)


@pytest.mark.parametrize("stride,bits", [(1, 2), (8, 0), (4, 1), (2, 8)])
def test_every_sampling_specification(pkg, lf, forced, stride, bits):
    """the table cell a wave looks up follows the sub-cell it aims at: 4 x 4 cells per stratum where the
    specification has at least that many sub-cells, fewer otherwise (independent pixels: the whole stratum)"""
    W, H, spp, sun = 80, 48, 64, [0.05, -0.03, -1.0]
    lens = _crop(pkg.load_lens_file("dgauss11.lens"), W)
    mask = load_texels("pentbig500_14.png")
    _setup(pkg, lf, lens, W, H, sun, 0.05, mask)
    lf.set_tile_stride(stride)
    lf.set_pupil_subcells(bits)
    lfo.set_tile_stride(stride)
    lfo.lib().geo_set_sub_bits(bits)
    try:
        g1, c1, info, g0, c0 = _both(pkg, lf, spp, 77)
        assert info["culled"] and info["P"] == 32          # 4 x 4 table cells per stratum whatever the sub-cells
        assert np.array_equal(g1, g0) and c1["rays_hit_light"] == c0["rays_hit_light"] > 0
        lf.set_march_culling(2)
        lf.reset_counters()
        lf.trace_ghosts(spp, 77)
        og, oc = lfo.geo_trace(lens, W, H, 0, H, spp, 77, None, True, mask, sun, RAD, 0.05, n_threads=16)
        assert np.array_equal(g1, og) and oc == c1 and lfo.last_culled_lit == 0
    finally:
        lf.set_tile_stride(pkg.DEFAULT_TILE_STRIDE)
        lf.set_pupil_subcells(pkg.DEFAULT_SUBCELL_BITS)
        lfo.set_tile_stride(pkg.DEFAULT_TILE_STRIDE)
        lfo.lib().geo_set_sub_bits(pkg.DEFAULT_SUBCELL_BITS)
        lf.set_march_culling(1)


def test_eight_wavelengths_pair_subsets_bands_and_a_pupil_target(pkg, lf, forced):
    mask = load_texels("pentbig500_14.png")
    W, H, spp, sun = 96, 72, 64, [0.04, 0.03, -1.0]
    lens8 = _crop(pkg.load_lens_file("dgauss11_8lambda.lens"), W)
    w8, _ = pkg.spectral_weights(lens8["lambda_nm"])
    _setup(pkg, lf, lens8, W, H, sun, 0.05, mask, lambda_rgb=w8)
    g1, c1, info, g0, c0 = _both(pkg, lf, spp, 5)
    assert info["culled"] and np.array_equal(g1, g0) and c1["rays_hit_light"] == c0["rays_hit_light"] > 0
    # a subset of the pairs without the primary path: the bits follow the selection's order
    lens = _crop(pkg.load_lens_file("dgauss11.lens"), W)
    pairs = [(1, 3), (0, 2), (6, 8), (2, 9)]
    _setup(pkg, lf, lens, W, H, sun, 0.05, mask, pairs=pairs, primary=False)
    g1, c1, info, g0, c0 = _both(pkg, lf, spp, 6)
    assert np.array_equal(g1, g0) and g0.max() > 0 and c0["rays_launched"] == W * H * spp * 3 * 4
    lf.set_march_culling(2)
    lf.trace_ghosts(spp, 6)
    t = lf.cull_table()
    assert t.shape == (2, 2, info["cells"] + 1) and int(t.max()) < 16      # four paths, four bits
    # one rank's share of the frame (tile rows dealt round-robin) and a band: the same pixels where they render
    _setup(pkg, lf, lens, W, H, sun, 0.05, mask)
    whole, _, _, _, _ = _both(pkg, lf, spp, 9)
    for rank in range(3):
        lf.set_row_interleave(rank, 3)
        lf.set_march_culling(2)
        lf.trace_ghosts(spp, 9)
        part = lf.read_buffer(pkg.GHOST_BUFFER)
        rows = [y for y in range(H) if (y // 8) % 3 == rank]
        assert np.array_equal(part[rows], whole[rows])
    lf.set_row_interleave(0, 1)
    lf.set_band(16, 40)
    lf.trace_ghosts(spp, 9)
    assert np.array_equal(lf.read_buffer(pkg.GHOST_BUFFER)[16:40], whole[16:40])
    lf.set_band(0, H)
    # samples aimed at the exit pupil's image (primary + the pairs with a mirror in front of the stop)
    n, stop = lens["n"], lens["stop"]
    front = [(i, j) for i in range(n) for j in range(i + 1, n) if i != stop and j != stop and i < stop]
    _setup(pkg, lf, lens, W, H, sun, 0.05, mask, pairs=front)
    lf.aim_at_exit_pupil(1.1)
    try:
        g1, c1, info, g0, c0 = _both(pkg, lf, spp, 10)
        assert info["culled"] and np.array_equal(g1, g0) and g0.max() > 0
    finally:
        lf.set_pupil_target(0.0, 0.0)
        lf.set_ghost_pairs(None, True)
        lf.set_march_culling(1)


def test_table_reuse_rebuild_and_the_automatic_choice(pkg, lf):
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 1920, 1080, 256
    _setup(pkg, lf, lens, W, H, [0.03, 0.02, -1.0], 0.05, mask)
    lf.timing_reset()
    lf.timing_enable(True)
    lf.set_march_culling(1)
    for _ in range(3):
        lf.trace_ghosts(spp, 3)
    lf.synchronize()
    assert lf.cull_info()["culled"] and lf.timing_get("cull_prepass")[0] == 1       # built once, reused twice
    frac = lf.cull_started_fraction()
    assert 0.0 < frac < 0.12
    first = lf.read_buffer(pkg.GHOST_BUFFER)
    lf.set_sun([0.031, 0.02, -1.0], RAD, 0.05)                                      # the sun moved: a new table
    lf.trace_ghosts(spp, 3)
    lf.synchronize()
    assert lf.timing_get("cull_prepass")[0] == 2
    lf.set_sun([0.03, 0.02, -1.0], RAD, 0.05)
    lf.set_march_culling(2)
    for _ in range(2):
        lf.trace_ghosts(spp, 3)
    lf.synchronize()
    assert lf.timing_get("cull_prepass")[0] == 4                                    # mode 2: at every launch
    assert np.array_equal(lf.read_buffer(pkg.GHOST_BUFFER), first)
    lf.timing_enable(False)
    # a sun as wide as the field: the table would start most of everything -- the launch marches the path tree
    lf.set_sun([0.0, 0.0, -1.0], RAD, 0.6)
    lf.reset_counters()
    lf.trace_ghosts(spp, 3)
    assert not lf.cull_info()["culled"] and lf.cull_started_fraction() > 0.12
    assert lf.counters()["rays_launched"] == W * H * spp * 3 * 46
    assert lf.cull_table() is None
    with pytest.raises(pkg.LensFlareError):
        lf.set_march_culling(3)
    assert lf.cull_reason() == "table_too_full"
    # a frame whose smallest blocks (16 pixels) are still large on the sensor (240 pixels on 36 mm: 2.4 mm): every path is marched
    _setup(pkg, lf, lens, 240, 136, [0.03, 0.02, -1.0], 0.05, mask)
    lf.reset_counters()
    lf.trace_ghosts(4, 1)
    assert not lf.cull_info()["culled"] and lf.counters()["rays_launched"] == 240 * 136 * 4 * 3 * 46
    assert lf.cull_reason() == "block_too_large"
    # no stop in the prescription: nothing to bound a pupil with -- every path is marched
    thin = pkg.load_lens_file("thinlens.lens")
    _setup(pkg, lf, thin, 64, 64, [0.0, 0.0, -1.0], 0.1, np.ones((8, 8), np.float32))
    lf.reset_counters()
    lf.trace_ghosts(4, 1)
    c = lf.counters()
    assert not lf.cull_info()["culled"] and ray_budget(lf, c, 64 * 64 * 4 * 3 * 2)
    assert lf.cull_reason() == "no_stop"
    lf.set_march_culling(0)
    lf.trace_ghosts(4, 1)
    assert lf.cull_reason() == "off"
    lf.set_march_culling(1)


def test_bench_frame_culled_equals_full_on_other_suns(pkg, lf):
    """1080p, all 46 paths, the automatic choice: four suns the bench does not use (off axis, near a corner, outside
    the frame, a small one) -- whole frames, culled = full enumeration bit for bit, and what it saves"""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 1920, 1080, 64
    for sun, alpha in (([0.12, 0.08, -1.0], 0.05), ([0.30, -0.17, -1.0], 0.05), ([0.45, 0.1, -1.0], 0.05), ([0.05, 0.02, -1.0], 0.004)):
        _setup(pkg, lf, lens, W, H, sun, alpha, mask)
        g1, c1, info, g0, c0 = _both(pkg, lf, spp, 0x5EED)
        assert info["culled"]
        assert np.array_equal(g1, g0) and c1["rays_hit_light"] == c0["rays_hit_light"]
        assert c1["rays_launched"] < 0.2 * c0["rays_launched"]
        print(f"sun {sun} alpha {alpha}: started {c1['rays_launched'] / c0['rays_launched']:.4f} of the rays, "
              f"reached the scene {c1['rays_reached_scene'] / c1['rays_launched']:.3f} (full: {c0['rays_reached_scene'] / c0['rays_launched']:.3f})")


@pytest.mark.parametrize("mask_name,refocus_mm,lens_name", [
    ("pentbiglines.png", 0.0, "dgauss11.lens"),       # a small pentagon with thin bright lines: the occupancy grid, not a bounding box
    ("octagonbokeh.png", 0.0, "dgauss11.lens"),       # a nearly full stop
    ("pentbig500_14.png", 500.0, "dgauss11.lens"),    # the lens refocused to half a metre (the sensor 5.7 mm further back)
    ("pentbig500_14.png", 0.0, "dgauss11_8lambda.lens"),
    ("synthetic_ring", 0.0, "dgauss11.lens")])        # an annular stop: open cells around a closed centre
def test_other_masks_focus_and_prescriptions_at_1080p(pkg, lf, mask_name, refocus_mm, lens_name):
    """whole 1080p frames, the automatic choice: culled = full enumeration bit for bit under other stop masks (the cull
    reads a 32 x 32 occupancy grid of the mask), a refocused lens and the 8-wavelength prescription"""
    lens = pkg.load_lens_file(lens_name)
    if mask_name == "synthetic_ring":
        yy, xx = np.mgrid[0:256, 0:256]
        rr = np.hypot(xx - 127.5, yy - 127.5) / 128.0
        mask = ((rr > 0.35) & (rr < 0.8)).astype(np.float32)
    else:
        mask = load_texels(mask_name)
    W, H, spp = 1920, 1080, 16
    lam = None
    if lens_name.endswith("8lambda.lens"):
        lam, _ = pkg.spectral_weights(lens["lambda_nm"])
    for sun, alpha in (([0.03, 0.02, -1.0], 0.05), ([-0.15, 0.2, -1.0], 0.03)):
        _setup(pkg, lf, lens, W, H, sun, alpha, mask, lambda_rgb=lam)
        if refocus_mm > 0:
            lf.focus_lens(refocus_mm)
        g1, c1, info, g0, c0 = _both(pkg, lf, spp, 0xF0C5)
        assert np.array_equal(g1, g0) and c1["rays_hit_light"] == c0["rays_hit_light"]
        if info["culled"]:
            assert c1["rays_launched"] < 0.5 * c0["rays_launched"]
        print(f"{mask_name} {lens_name} refocus {refocus_mm} sun {sun}: culled {info['culled']}, started "
              f"{c1['rays_launched'] / c0['rays_launched']:.4f}, lit rays {c0['rays_hit_light']}")


def test_the_split_tail_is_the_same_frame(pkg, lf):
    """k_march_cull splits the LAST tiles of a launch over several workgroups (MarchArgs::tail_from: the launch ends on short
    workgroups; their integer sums meet in a small buffer the last arrival converts and clears).  Whatever the tail's size
    and split (lf_test_knob march_tail_tiles / march_tail_groups; 0 / 1 = no tail): the same pixels and counters -- twice in
    a row (the buffer is left clean), accumulating, with columns 8 apart, and for a share of the block deal."""
    lens = pkg.load_lens_file("dgauss11.lens")
    W, H, spp = 1920, 1080, 64
    _setup(pkg, lf, lens, W, H, [0.12, 0.08, -1.0], 0.05, load_texels("pentbig500_14.png"))
    lf.set_march_culling(2)

    def frame(tiles, groups, accumulate=False):
        lf.test_knob("march_tail_tiles", tiles)
        lf.test_knob("march_tail_groups", groups)
        lf.reset_counters()
        if accumulate:
            lf.clear_ghost_buffer()
            lf.set_ghost_accumulate(True)
            lf.trace_ghosts(spp, 5)
        lf.trace_ghosts(spp, 7)
        lf.set_ghost_accumulate(False)
        assert lf.cull_info()["culled"]
        return lf.read_buffer(pkg.GHOST_BUFFER), lf.counters(), lf.executed_events()
    try:
        for deal in (None, (1, 3)):
            if deal:
                lf.set_block_deal(*deal)
            for stride in (1, 8):                # (8 = the default)
                lf.set_tile_stride(stride)
                g0, c0, e0 = frame(0, 1)
                assert g0.any()
                for tiles, groups in ((-1, -1), (-1, -1), (1024, 3), (4096, 2), (64, 2), (100000, 4)):
                    g1, c1, e1 = frame(tiles, groups)
                    assert np.array_equal(g1, g0) and c1 == c0 and e1 == e0, (deal, stride, tiles, groups)
            ga, ca, _ = frame(0, 1, accumulate=True)
            gb, cb, _ = frame(-1, -1, accumulate=True)
            assert np.array_equal(ga, gb) and ca == cb and not np.array_equal(ga, g0)
    finally:
        lf.test_knob("march_tail_tiles", -1)
        lf.test_knob("march_tail_groups", -1)
        lf.set_tile_stride(8)
        lf.set_block_deal(0, 1)
        lf.set_march_culling(1)


def test_weight_on_every_event_is_the_same_frame(pkg, lf):
    """k_march_cull<K, true> (lf_test_knob cull_weights_first, bench.py's `every_event_weighted` leg): the Fresnel / mask weight
    evaluated on EVERY executed event of every started ray -- SURVEY 8d's unit event -- instead of on a second march of
    the lanes that reach the lobe.  Same weights, same order of additions: the same pixels and counters."""
    lens = pkg.load_lens_file("dgauss11.lens")
    _setup(pkg, lf, lens, 1920, 64, [0.12, 0.08, -1.0], 0.05, load_texels("pentbiglines.png"))
    lf.set_march_culling(2)
    out = []
    for first in (False, True):
        lf.test_knob("cull_weights_first", first)
        lf.reset_counters()
        lf.trace_ghosts(64, 77)
        assert lf.cull_info()["culled"]
        c = lf.counters()
        out.append((lf.read_buffer(pkg.GHOST_BUFFER), {k: v for k, v in c.items() if not k.startswith("remarch")}))
    lf.test_knob("cull_weights_first", 0)
    lf.set_march_culling(1)
    assert out[0][0].any() and np.array_equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1], (out[0][1], out[1][1])


def test_random_frames_culled_equals_full(pkg, forced):
    """tests/cull_fuzz.py (the recorded draws: profiles/r05_cull_fuzz.json, r06_cull_fuzz.json), a fresh draw of 160 here:
    random masks, prescriptions, sensors, suns, pair subsets, sampling specifications and bands -- the culled kernel
    (forced) against the full enumeration, pixels and the count of rays that reached the light; and the launch's audit
    (one ray per dropped box, as shipped) finds nothing to refute."""
    import cull_fuzz
    r = cull_fuzz.run(160, 7, log=None, audit=1)
    s = r["summary"]
    print(s)
    assert s["compared"] == 160 and s["culled_kernel_ran"] == 160 and s["frames_with_light"] > 120
    assert s["frames_differing"] == 0, [c for c in r["cases"] if c.get("BAD")]
    assert s["audit_rays"] > 1e8 and s["audit_lit"] == 0 and s["launches_refuted_by_the_audit"] == 0


def test_the_drivers_smoke_entry():
    """__graft_entry__.smoke(): what the driver runs before the bench -- the flare layer against the real reference's
    golden frame, the march against the float32 oracle, the lens-imaged flat field, and the culled default launch of a
    1920-pixel-wide frame against the full enumeration."""
    import __graft_entry__ as g
    g.smoke()


ONCE_LOST = [191, 691, 1630, 1699, 1812, 2006, 2451, 3329, 3589, 3725, 3973, 4063, 4095, 4215, 4223, 4242, 4366, 4469, 4574, 4933,
             5085, 5234, 5416, 5455, 5687, 5722, 5913]


def test_the_frames_that_once_lost_lit_rays(pkg, forced):
    """tests/cull_fuzz.py 6000 424242 -- the draw that brought a second design family (a Cooke triplet) -- found 27 frames
    on which the pre-pass of that day dropped boxes that carried light (1 to 507 lit rays of 1e5 ... 1e9): boxes bounded by
    the samples left after total reflection took the others, 'every sample ends here' decided by a range rule fitted to the
    double Gauss, second order in four axes at once against a small lobe (profiles/r05_march_variants.txt).  The same 27
    frames, replayed from the same stream, under the strict rules: the full enumeration, bit for bit."""
    import cull_fuzz
    r = cull_fuzz.run(max(ONCE_LOST) + 1, 424242, log=None, only=set(ONCE_LOST))
    s = r["summary"]
    assert s["compared"] == len(ONCE_LOST) and s["culled_kernel_ran"] == len(ONCE_LOST)
    assert s["frames_differing"] == 0, [c for c in r["cases"] if c.get("BAD")]
    assert sum(1 for c in r["cases"] if c["lens"] == "triplet") == 23


def test_the_frames_past_a_spheres_rim(pkg, forced):
    """... and the two of the harsher draw (FUZZ_HARSH=1 tests/cull_fuzz.py 4000 1234567 5: suns of 10 and 15 degrees on
    perturbed 8-wavelength prescriptions, tables that start 27 %): boxes that had lost samples past a sphere's rim, bounded by
    the samples left, dropped with 213 and 19 lit rays.  A box that lost any sample is bounded by nothing now."""
    import cull_fuzz
    was = cull_fuzz.HARSH
    cull_fuzz.HARSH = True
    try:
        r = cull_fuzz.run(3331, 1234567, log=None, only={3328, 3330}, families=5)
    finally:
        cull_fuzz.HARSH = was
    s = r["summary"]
    assert s["compared"] == 2 and s["culled_kernel_ran"] == 2 and s["frames_differing"] == 0, r["cases"]
    assert all(c["lens"] == "dgauss11_8lambda.lens" and c["alpha"] > 0.15 and c["lit_rays_full"] > 5e8 for c in r["cases"])


# ---- round 6 ------------------------------------------------------------------------------------------------------------
# the rules round 5 REPLACED (range rule for "every sample ends here", boxes that lost samples bounded by the rest, second
# order not summed over the axes, no extra factor on the lobe test): installed through the test hook only
OLD_RULES = {"cull_strict": 0, "cull_strict_lost": 0, "cull_slack": 0, "cull_lobe_k": 1.0}


def _table(pkg, lf, spp, key=1):
    lf.set_march_culling(2)
    lf.trace_ghosts(spp, key)
    assert lf.cull_info()["culled"], lf.cull_reason()
    return lf.cull_table(), lf.cull_started_fraction()


@pytest.mark.parametrize("lens_name,W,H,spp,sun,alpha", [
    ("dgauss11.lens", 1920, 1080, 256, [0.01533, 0.0069, -1.0], 0.05),         # the bench frame
    ("dgauss11.lens", 1920, 1080, 64, [0.30, -0.17, -1.0], 0.05),
    ("dgauss11_8lambda.lens", 3840, 2160, 1024, [0.01533, 0.0069, -1.0], 0.05),  # C5's table (the pre-pass only)
    ("dgauss11_8lambda.lens", 1920, 1080, 36, [-0.15, 0.2, -1.0], 0.2),          # a wide sun: boxes lose samples
    ("dgauss11.lens", 3840, 2160, 16, [0.05, 0.02, -1.0], 0.004)])               # 128-pixel blocks, a small lobe
def test_the_shipped_kernel_is_the_general_one(pkg, lf, forced, lens_name, W, H, spp, sun, alpha):
    """k_cull_level (the shipped rules hard-wired: a footprint only where a test can fire, no scratch) builds the table of
    k_cull_level_general (the rules as arguments: round 5's kernel) bit for bit"""
    lens = pkg.load_lens_file(lens_name)
    lam = pkg.spectral_weights(lens["lambda_nm"])[0] if lens_name.endswith("8lambda.lens") else None
    _setup(pkg, lf, lens, W, H, sun, alpha, load_texels("pentbig500_14.png"), lambda_rgb=lam)
    lf.set_cull_audit(0)
    try:
        shipped, frac = _table_only(pkg, lf, spp)
        lf.test_knob("cull_general_kernel", 1)
        general, frac_g = _table_only(pkg, lf, spp)
    finally:
        lf.test_knob("cull_general_kernel", 0)
        lf.set_cull_audit(1)
        lf.set_march_culling(1)
    assert shipped.any() and np.array_equal(shipped, general) and frac == frac_g
    print(f"{lens_name} {W}x{H} {spp} spp: started {frac:.4f}")


def _table_only(pkg, lf, spp):
    """the table of the launch lf_trace_ghosts(spp) would make, without the march: one tile row is rendered"""
    lf.set_band(0, 8)
    try:
        return _table(pkg, lf, spp)
    finally:
        lf.set_band(0, lf.H)


def test_random_frames_shipped_kernel_is_the_general_one(pkg, forced):
    """... and on 60 random frames (three design families, every kind of mask): the same tables"""
    import cull_fuzz
    tables = {}

    def hook(lf, rec, case):
        lf.set_cull_audit(0)
        lf.set_march_culling(2)
        lf.set_band(0, 8)
        lf.set_row_interleave(0, 1)
        lf.trace_ghosts(case["spp"], case["key"])
        rec["culled"] = lf.cull_info()["culled"]
        tables.setdefault(rec["case"], []).append(lf.cull_table())

    cull_fuzz.run(60, 99, log=None, hook=hook, families=5)
    cull_fuzz.run(60, 99, log=None, hook=hook, families=5, knobs={"cull_general_kernel": 1})
    assert len(tables) == 60
    for case, (a, b) in tables.items():
        assert a is not None and np.array_equal(a, b), case


def test_the_audit_costs_little_and_finds_nothing_on_the_bench_frame(pkg, lf):
    """c3: 8.8e7 dropped (block, cell, path) combinations, one ray each, marched with the march's own events: none reaches
    the light; under 2 ms"""
    lens = pkg.load_lens_file("dgauss11.lens")
    _setup(pkg, lf, lens, 1920, 1080, [0.01533, 0.0069, -1.0], 0.05, load_texels("pentbig500_14.png"))
    lf.set_march_culling(2)
    lf.reset_counters()
    lf.timing_reset()
    lf.timing_enable(True)
    for k in range(3):
        lf.trace_ghosts(256, 11 + k)
    lf.synchronize()
    n, ms = lf.timing_get("cull_audit")
    lf.timing_enable(False)
    a = lf.cull_audit()
    print(f"audit: {a}, {ms / n:.3f} ms per frame; pre-pass {lf.timing_get('cull_prepass')}")
    assert lf.cull_info()["culled"] and lf.cull_reason() == "applied"
    assert n == 3 and a["lit"] == 0 and a["launches_refuted"] == 0
    assert abs(a["rays"] / 3 - (1.0 - lf.cull_started_fraction()) * 30 * 17 * 4096 * 46) < 1.0
    assert ms / n < 2.0
    lf.set_march_culling(1)


def test_the_audit_catches_the_rules_that_lost_light(pkg, forced):
    """The 27 frames of ONCE_LOST under the rules round 5 replaced (installed through the test hook), 32 keys each, 8 audit
    rays per dropped box: the audit refutes tables -- and a launch whose table it refuted is the full enumeration's frame
    (it took the path tree); what it does not notice is counted, not hidden."""
    import cull_fuzz
    r = cull_fuzz.run(max(ONCE_LOST) + 1, 424242, log=None, only=set(ONCE_LOST), knobs=OLD_RULES, audit=8, n_keys=32)
    s = r["summary"]
    print(s)
    lost = [c for c in r["cases"] if c["values_differing"] or c["lit_rays_full"] != c["lit_rays_culled"] or c["audit_refuted"]]
    print("frames that lost light or were refuted:", [(c["case"], c["audit_refuted"], c["differing_unnoticed"]) for c in lost])
    assert s["compared"] == len(ONCE_LOST)
    assert s["launches_refuted_by_the_audit"] > 0 and s["audit_lit"] > 0
    assert s["launches_differing_although_refuted"] == 0      # a refuted launch took the path tree: the enumeration's frame


@pytest.mark.parametrize("W,H,block", [(1280, 720, 64), (960, 540, 32), (640, 360, 32), (400, 224, 16)])
def test_small_frames_cull_with_smaller_blocks(pkg, lf, forced, W, H, block):
    """frames narrower than 1280 pixels on the 36 mm sensor: blocks of 32 / 16 pixels (<= 1.8 mm), a wave tile (64 x 8 pixels)
    spans several of them and its lanes look their rows up one by one -- culled = full enumeration, every sampling default"""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    for sun, alpha, spp in (([0.01533, 0.0069, -1.0], 0.05, 64), ([0.12, -0.08, -1.0], 0.03, 16)):
        _setup(pkg, lf, lens, W, H, sun, alpha, mask)
        g1, c1, info, g0, c0 = _both(pkg, lf, spp, 0xB10C + W)
        assert info["culled"] and info["block_px"] == block and info["blocks_x"] == (W + block - 1) // block, (info, lf.cull_reason())
        assert np.array_equal(g1, g0) and c1["rays_hit_light"] == c0["rays_hit_light"] > 0
        assert c1["rays_launched"] < 0.5 * c0["rays_launched"]
        print(f"{W}x{H} blocks of {block}: started {c1['rays_launched'] / c0['rays_launched']:.4f} of the rays, "
              f"table {lf.cull_started_fraction():.4f} (the launch's own choice without the test knob: "
              f"{'culled' if lf.cull_started_fraction() <= 0.10 + 1.6 / 46 else 'path tree, table too full'})")
    # ... and against the oracle on a crop with 16-pixel blocks (its table lookup follows the block size)
    try:
        Wc, Hc, spp = 80, 48, 16
        crop = dict(lens)
        crop["sensor_width_mm"] = 36.0 * Wc / 400       # (0.09 mm per pixel: 32 pixels would be 2.9 mm, 16 are 1.44)
        _setup(pkg, lf, crop, Wc, Hc, [0.03, 0.02, -1.0], 0.05, mask)
        g1, c1, info, g0, c0 = _both(pkg, lf, spp, 5)
        assert info["culled"] and info["block_px"] == 16 and np.array_equal(g1, g0)
        lf.set_march_culling(2)
        lf.reset_counters()
        lf.trace_ghosts(spp, 5)
        lfo.geo_follow_device(lf)          # (the smoke test above ends with the oracle following no device)
        og, oc = lfo.geo_trace(crop, Wc, Hc, 0, Hc, spp, 5, None, True, mask, [0.03, 0.02, -1.0], RAD, 0.05, n_threads=16)
        assert np.array_equal(g1, og) and oc == c1 and lfo.last_culled_lit == 0
    finally:
        lf.set_march_culling(1)


def test_a_prescription_whose_index_columns_are_out_of_order_marches_everything(pkg, lf):
    """the pre-pass brackets the spectrum by the first and the last index column: a table with a column out of order
    (lf_set_lens accepts any) is not culled -- and says why"""
    lens = dict(pkg.load_lens_file("dgauss11.lens"))
    ior = np.array(lens["ior"], np.float32)
    ior[[1, 2]] = ior[[2, 1]]                      # C, F, d: d lies BETWEEN the two ends in index, not at the end
    lens["ior"] = ior
    _setup(pkg, lf, lens, 1920, 64, [0.01533, 0.0069, -1.0], 0.05, load_texels("pentbig500_14.png"))
    lf.reset_counters()
    lf.trace_ghosts(4, 1)
    assert not lf.cull_info()["culled"] and lf.cull_reason() == "dispersion_not_monotonic"
    assert lf.counters()["rays_launched"] == 1920 * 64 * 4 * 3 * 46


def test_more_paths_than_a_mask_has_bits(pkg, lf, forced):
    """13 interfaces: 66 pairs + the primary path = 67 paths, more than the 64 bits of a table entry: the culled march goes in
    two launches over the halves of the selection, each with its own table, the integer sums of both in one accumulator --
    the frame and the counters of the one launch that marches everything"""
    lens = dict(pkg.load_lens_file("dgauss11.lens"))

    def split(k, t_first, radius, ior_first):
        """an interface inside element k (a cemented pair of slightly different glasses)"""
        for key, val in (("radius", radius), ("thickness", t_first), ("semi_aperture", lens["semi_aperture"][k])):
            lens[key] = np.insert(np.asarray(lens[key], np.float32), k, np.float32(val))
        lens["thickness"][k + 1] -= np.float32(t_first)
        lens["ior"] = np.insert(np.asarray(lens["ior"], np.float32), k, np.asarray(ior_first, np.float32), axis=1)
        # (row k keeps the element's front radius and gets the new glass; row k + 1 is the new interface into the old glass)
        lens["radius"][k], lens["radius"][k + 1] = lens["radius"][k + 1], np.float32(radius)
        lens["n"] += 1
        if lens["stop"] >= k:
            lens["stop"] += 1

    split(9, 1.4, 300.0, [1.6204 - 0.0031, 1.6204, 1.6204 + 0.0072])        # the rear element
    split(0, 1.6, -250.0, [1.6200 - 0.0051, 1.6200, 1.6200 + 0.0119])       # the front element
    assert lens["n"] == 13 and lens["radius"][lens["stop"]] == 0.0
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 1920, 256, 64
    lens["sensor_width_mm"] = 36.0
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    lf.focus_lens(0.0)
    lf.set_sun([0.02, 0.01, -1.0], RAD, 0.05)
    lf.set_ghost_pairs(None, True)
    lf.set_band(0, H)
    lf.set_row_interleave(0, 1)
    g1, c1, info, g0, c0 = _both(pkg, lf, spp, 67)
    assert c0["rays_launched"] == W * H * spp * 3 * 67
    assert info["culled"] and info["reason"] == "applied"
    assert np.array_equal(g1, g0) and g0.max() > 0 and c1["rays_hit_light"] == c0["rays_hit_light"] > 0
    assert c1["rays_launched"] < 0.5 * c0["rays_launched"]
    lf.set_march_culling(2)
    lf.trace_ghosts(spp, 67)
    with pytest.raises(pkg.LensFlareError):
        lf.cull_table()                      # two tables: not handed out as one
    # the selection is what it was: a subset of it marches in one launch again
    lf.set_ghost_pairs([(0, 1), (2, 3)], True)
    lf.trace_ghosts(spp, 67)
    assert lf.cull_table() is not None
    lf.set_ghost_pairs(None, True)
    lf.set_march_culling(1)
