"""Multi-GPU inside the C ABI (lens-flare_amd/csrc/lf_group.hip): tile rows dealt round-robin, ONE
all-gather of finished tile rows per frame.  The GPU box has one device, so what runs here is
  * the RCCL plumbing itself with a 1-rank communicator (dlopen, unique id, ncclCommInitRank, gather);
  * the whole sharded frame with 2 and 3 contexts on device 0 (a device listed twice cannot join an
    RCCL communicator: the group then exchanges with peer copies -- same packing, same result):
    every context must end up with the frame a single context renders, bit for bit.
The n-GPU RCCL path proper is exercised by the driver's scaling run (bench.py --gpus N)."""
import numpy as np
import pytest

from goldenlib import load_texels

pytestmark = pytest.mark.gpu
SUN = dict(direction=[0.03, 0.02, -1.0], radiance=[1.0, 0.9, 0.5], angular_radius=0.05)


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def _setup(pkg, lf, lens, mask):
    lf.set_params(1, 25.0, 1.0)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_aperture(pkg.APERTURE_GHOST, mask)
    lf.set_lens(lens)
    lf.set_sun(SUN["direction"], SUN["radiance"], SUN["angular_radius"])
    lf.set_ghost_pairs(None, True)
    lf.set_jitter_counter(42)
    lf.set_camera(np.eye(3), [0, 0, 0], 40.0, 30.0)


def _frame(lf, spp, key):
    lf.find_sun_pos([[0.4, 0.3, -10.0, 1.0, 0.9, 0.5]])
    lf.trace_ghosts(spp, key)
    lf.render_flare_layer()


def test_single_rank_communicator(pkg):
    """ncclGetUniqueId / ncclCommInitRank / the gather entry point on a communicator of one."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    lf = pkg.LensFlare(0)
    lf.set_frame(72, 40)
    _setup(pkg, lf, lens, mask)
    uid = pkg.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    lf.comm_init_rank(1, 0, uid)
    _frame(lf, 8, 5)
    before = lf.read_buffer(pkg.SAMPLE_BUFFER)
    lf.comm_gather(pkg.SAMPLE_BUFFER)
    assert np.array_equal(lf.read_buffer(pkg.SAMPLE_BUFFER), before) and before.max() > 0
    with pytest.raises(pkg.LensFlareError):
        lf.comm_init_rank(1, 0, uid)          # one communicator per context
    lf.comm_destroy()
    lf.close()


@pytest.fixture()
def force_exchange(pkg):
    """the collectives run with a single rank as well (lf_test_knob comm_force_exchange), for every context created meanwhile"""
    pkg.test_knob_default("comm_force_exchange", 1)
    yield
    pkg.test_knob_default("comm_force_exchange", 0)


@pytest.fixture()
def cull_forced(pkg):
    """the culled march whatever the table starts (lf_test_knob cull_force), for every context created meanwhile"""
    pkg.test_knob_default("cull_force", 1)
    yield
    pkg.test_knob_default("cull_force", 0)


def test_async_exchange_pipeline_of_frames(pkg, force_exchange):
    """lf_comm_gather_async: pack / ncclAllGather / unpack on the second stream while the next frame
    is rendered on the first.  With the test knob comm_force_exchange the whole path runs on a communicator of
    one (RCCL copies the rank's slots to itself, the unpack skips them): a sequence of DIFFERENT
    frames queued back to back, each exchanged asynchronously, must leave exactly the last frame --
    and every read in between must see the frame that was current, whole."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H = 96, 56
    ref = pkg.LensFlare(0)
    ref.set_frame(W, H)
    _setup(pkg, ref, lens, mask)
    want = []
    for k in range(4):
        ref.set_sun([0.03 - 0.02 * k, 0.02, -1.0], SUN["radiance"], SUN["angular_radius"])
        _frame(ref, 8, 100 + k)
        want.append(ref.read_buffer(pkg.SAMPLE_BUFFER))
    ref.close()
    assert not np.array_equal(want[0], want[3])

    lf = pkg.LensFlare(0)
    lf.set_frame(W, H)
    _setup(pkg, lf, lens, mask)
    lf.comm_init_rank(1, 0, pkg.comm_unique_id())
    for k in range(4):                       # nothing waits for anything in here
        lf.set_sun([0.03 - 0.02 * k, 0.02, -1.0], SUN["radiance"], SUN["angular_radius"])
        _frame(lf, 8, 100 + k)
        lf.comm_gather_async(pkg.SAMPLE_BUFFER)
    assert np.array_equal(lf.read_buffer(pkg.SAMPLE_BUFFER), want[3])     # the read joins the streams
    # ... interleaved with reads, synchronous gathers and an explicit wait
    for k in (1, 0, 2):
        lf.set_sun([0.03 - 0.02 * k, 0.02, -1.0], SUN["radiance"], SUN["angular_radius"])
        _frame(lf, 8, 100 + k)
        lf.comm_gather_async(pkg.SAMPLE_BUFFER)
        if k == 0:
            lf.comm_gather(pkg.SAMPLE_BUFFER)
        if k == 2:
            lf.comm_wait()
        assert np.array_equal(lf.read_buffer(pkg.SAMPLE_BUFFER), want[k])
    lf.comm_gather_async(pkg.GHOST_BUFFER)
    lf.synchronize()
    lf.set_frame(40, 24)                      # a resize drains a pending exchange before it frees the buffers
    lf.comm_destroy()
    lf.close()


@pytest.mark.parametrize("n,W,H", [(2, 72, 40), (3, 100, 52), (2, 64, 8)])
def test_group_renders_the_single_gpu_frame(pkg, n, W, H):
    """n contexts (rehearsal group on device 0): round-robin tile rows + gather == one context."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    one = pkg.LensFlare(0)
    one.set_frame(W, H)
    _setup(pkg, one, lens, mask)
    one.reset_counters()
    _frame(one, 16, 9)
    want_sample, want_ghost = one.read_buffer(pkg.SAMPLE_BUFFER), one.read_buffer(pkg.GHOST_BUFFER)
    want_cnt = one.counters()
    one.close()
    assert want_ghost.max() > 0

    grp = pkg.LensFlareGroup([0] * n)
    grp.set_frame(W, H)
    for r in grp.ranks:
        _setup(pkg, r, lens, mask)
        r.reset_counters()
    grp.for_each(lambda lf, rank: _frame(lf, 16, 9))      # one host thread per context
    # before the exchange a rank holds only its own tile rows
    part = grp.ranks[1].read_buffer(pkg.GHOST_BUFFER)
    own = (np.arange(H) // 8) % n == 1
    assert np.array_equal(part[own], want_ghost[own])
    grp.gather(pkg.SAMPLE_BUFFER)
    grp.gather(pkg.GHOST_BUFFER)
    for r in grp.ranks:
        assert np.array_equal(r.read_buffer(pkg.SAMPLE_BUFFER), want_sample)
        assert np.array_equal(r.read_buffer(pkg.GHOST_BUFFER), want_ghost)
    total = {}
    for r in grp.ranks:
        for k, v in r.counters().items():
            total[k] = total.get(k, 0) + v
    assert total == want_cnt                              # the shares add up to the single-GPU frame
    grp.close()


def test_cpp_host_drives_a_group(pkg, tmp_path):
    """lens-flare_amd/host/group_demo.cpp: a plain C++ host (no Python in the loop) renders a frame on a
    group of contexts through lf_group_create / _set_frame / _for_each / _gather and verifies it against
    a single context itself; here with device 0 listed three times."""
    import os
    import subprocess
    demo = os.path.join(os.path.dirname(pkg.__file__), "host", "group_demo")
    assert os.path.exists(demo), "run __graft_entry__.build() first"
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    mask.tofile(tmp_path / "mask.f32")
    n, nl = lens["n"], lens["ior"].shape[0]
    with open(tmp_path / "lens.txt", "w") as f:
        f.write(f"{n} {lens['stop']} {nl} {lens['sensor_width_mm']!r}\n")
        for k in range(n):
            row = [lens["radius"][k], lens["thickness"][k], lens["semi_aperture"][k]] + [lens["ior"][l, k] for l in range(nl)]
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
    r = subprocess.run([demo, str(tmp_path / "lens.txt"), str(tmp_path / "mask.f32"), str(mask.shape[1]),
                        str(mask.shape[0]), "120", "68", "16", "0", "0", "0"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "3 devices" in r.stdout


def test_communicator_introspection_deadline_test_and_abort(pkg):
    """What bench.py's bring-up relies on (round 3): lf_comm_available before any blocking call,
    lf_comm_info = what RCCL itself reports (ncclCommCount / ncclCommUserRank), lf_comm_test = the
    non-blocking completion query behind the first exchange's deadline, lf_comm_abort = the way out of
    a failed one, and the exchange's own timing entry."""
    import time
    assert pkg.comm_available()
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    lf = pkg.LensFlare(0)
    lf.set_frame(64, 48)
    _setup(pkg, lf, lens, mask)
    assert lf.comm_info() == (0, -1)                       # no communicator yet
    lf.comm_init_rank(1, 0, pkg.comm_unique_id())
    assert lf.comm_info() == (1, 0)
    _frame(lf, 8, 3)
    want = lf.read_buffer(pkg.SAMPLE_BUFFER)
    lf.test_knob("comm_force_exchange", 1)
    lf.timing_reset()
    lf.timing_enable(True)
    lf.comm_gather_async(pkg.SAMPLE_BUFFER)
    t0 = time.time()
    while not lf.comm_test():                              # polls, never blocks
        assert time.time() - t0 < 30
        time.sleep(0.001)
    lf.comm_gather(pkg.SAMPLE_BUFFER)
    lf.synchronize()
    assert lf.comm_test()
    n, ms = lf.timing_get("exchange")
    lf.timing_enable(False)
    assert n == 2 and 0 < ms < 1000                        # both forms are timed, on the stream they run on
    assert np.array_equal(lf.read_buffer(pkg.SAMPLE_BUFFER), want)
    lf.comm_abort()
    assert lf.comm_info() == (0, -1)
    lf.set_row_interleave(0, 1)
    _frame(lf, 8, 3)                                       # the context's streams are usable afterwards
    assert np.array_equal(lf.read_buffer(pkg.SAMPLE_BUFFER), want)
    lf.close()


def test_float_exchange_rounds_only_received_rows(pkg):
    """lf_comm_set_exchange_precision(32) (SURVEY 8e budgets an f32 exchange): a rank's own tile rows
    stay the doubles it rendered, the rows it receives are those doubles rounded to float."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, spp, key, n = 72, 52, 8, 21, 3
    ref = pkg.LensFlare(0)
    ref.set_frame(W, H)
    _setup(pkg, ref, lens, mask)
    _frame(ref, spp, key)
    want = ref.read_buffer(pkg.SAMPLE_BUFFER)
    ref.close()
    grp = pkg.LensFlareGroup([0] * n)
    grp.set_frame(W, H)
    for lf in grp.ranks:
        _setup(pkg, lf, lens, mask)
        lf.comm_set_exchange_precision(32)
    grp.for_each(lambda lf, r: _frame(lf, spp, key))
    grp.gather(pkg.SAMPLE_BUFFER)
    as_float = want.astype(np.float32).astype(np.float64)
    for r, lf in enumerate(grp.ranks):
        got = lf.read_buffer(pkg.SAMPLE_BUFFER)
        own = (np.arange(H) // 8) % n == r
        assert np.array_equal(got[own], want[own])             # rendered here: untouched
        assert np.array_equal(got[~own], as_float[~own])       # received: the sender's value as a float
        assert np.abs(got - want).max() <= 6e-8 * np.abs(want).max()
    grp.ranks[0].comm_set_exchange_precision(64)
    with pytest.raises(pkg.LensFlareError):                    # one precision per group
        grp.gather(pkg.SAMPLE_BUFFER)
    grp.close()


def test_tonemapped_frame_is_refreshed_by_an_exchange(pkg):
    """write_to_framebuffer caches the rows it has tonemapped; an exchange changes rows of the sensor
    buffer, so the cache must not survive it (round-2 advisor finding): before the gather a rank's
    framebuffer shows its own tile rows only, after it the whole frame -- not the stale copy."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, n = 72, 40, 2
    one = pkg.LensFlare(0)
    one.set_frame(W, H)
    _setup(pkg, one, lens, mask)
    _frame(one, 8, 13)
    want = one.write_to_framebuffer(0, 0, W, H)
    one.close()
    grp = pkg.LensFlareGroup([0] * n)
    grp.set_frame(W, H)
    for r in grp.ranks:
        _setup(pkg, r, lens, mask)
    grp.for_each(lambda lf, rank: _frame(lf, 8, 13))
    own = (np.arange(H) // 8) % n == 0
    before = grp.ranks[0].write_to_framebuffer(0, 0, W, H)       # fills the cache with the partial frame
    assert np.array_equal(before[own], want[own]) and not np.array_equal(before[~own], want[~own])
    grp.gather(pkg.SAMPLE_BUFFER)
    assert np.array_equal(grp.ranks[0].write_to_framebuffer(0, 0, W, H), want)
    grp.close()


@pytest.mark.parametrize("W,H,spp,bits", [(1920, 1080, 4, 64), (3840, 2160, 1, 64), (1920, 1080, 4, 32), (3840, 2160, 1, 32),
                                          # ... and at the configurations' own sample counts: c3 / C4's march as 8 ranks render it
                                          (1920, 1080, 256, 64), (3840, 2160, 256, 64),
                                          # C5: 8 wavelengths (the committed Cauchy-fit prescription) at 4K x 1024 spp
                                          (3840, 2160, 1024, 64)])
def test_eight_ranks_at_bench_sizes(pkg, W, H, spp, bits):
    """What only an 8-GPU node would otherwise reveal: the exchange's correctness hinges
    on the [groups][world][tile row] staging layout being what ncclAllGather delivers.  8 contexts on device 0
    (the peer-copy stand-in executes the SAME plan the RCCL call is given: lf_group.hip exchange_plan) at the
    two bench frame sizes -- 135 and 270 tile rows, neither divisible by 8, so the padded last group is
    exercised -- in the f64 and the f32 exchange: every rank ends with the single-context frame, bit for bit
    (f32: the received rows rounded), and the plan is the one lf_comm_gather hands to ncclAllGather."""
    n = 8
    lens = pkg.load_lens_file("dgauss11_8lambda.lens" if spp == 1024 else "dgauss11.lens")
    w8 = pkg.spectral_weights(lens["lambda_nm"])[0] if spp == 1024 else None
    mask = load_texels("pentbig500_14.png")
    one = pkg.LensFlare(0)
    one.set_frame(W, H)
    _setup(pkg, one, lens, mask)
    if w8 is not None:
        one.set_lambda_rgb(w8)
    one.reset_counters()
    _frame(one, spp, 9)
    want = one.read_buffer(pkg.SAMPLE_BUFFER)
    want_cnt = one.counters()
    one.close()
    assert want.max() > 0
    ntrows = (H + 7) // 8
    assert ntrows % n != 0                                    # the last group of tile rows is short
    groups = (ntrows + n - 1) // n
    grp = pkg.LensFlareGroup([0] * n)
    grp.set_frame(W, H)
    for r in grp.ranks:
        _setup(pkg, r, lens, mask)
        if w8 is not None:
            r.set_lambda_rgb(w8)
        r.reset_counters()
        r.comm_set_exchange_precision(bits)
        plan = r.comm_exchange_plan(n)
        # sendcount / offsets: rank q's `sendcount` elements at recv + q * sendcount, the ncclAllGather contract
        assert plan["groups"] == groups and plan["tile_row_elements"] == 8 * W * 3
        assert plan["sendcount"] == groups * 8 * W * 3
        assert plan["element_bytes"] == bits // 8
        assert plan["recv_offset_bytes"] == plan["sendcount"] * plan["element_bytes"]
        assert plan["staging_bytes"] >= (n + 1) * plan["sendcount"] * plan["element_bytes"]
    grp.for_each(lambda lf, rank: _frame(lf, spp, 9))
    grp.gather(pkg.SAMPLE_BUFFER)
    as_float = want.astype(np.float32).astype(np.float64)
    for r, lf in enumerate(grp.ranks):
        got = lf.read_buffer(pkg.SAMPLE_BUFFER)
        if bits == 64:
            assert np.array_equal(got, want)
        else:
            own = (np.arange(H) // 8) % n == r
            assert np.array_equal(got[own], want[own]) and np.array_equal(got[~own], as_float[~own])
        del got
    total = {}
    for r in grp.ranks:
        for k, v in r.counters().items():
            total[k] = total.get(k, 0) + v
    assert total == want_cnt
    grp.close()


def test_a_poisoned_context_refuses_communicator_calls_and_still_renders(pkg):
    """lf_comm_poison (what a host calls when it gives up on a bring-up call blocked in another thread):
    every lf_comm_* call is refused afterwards -- nothing a late-returning call produces can be published -- the
    rest of the context works, and with no call left inside lf_destroy frees it normally."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    lf = pkg.LensFlare(0)
    lf.set_frame(64, 32)
    _setup(pkg, lf, lens, mask)
    _frame(lf, 4, 5)
    before = lf.read_buffer(pkg.SAMPLE_BUFFER)
    uid = pkg.comm_unique_id()
    assert not lf.comm_is_poisoned()
    lf.comm_poison()
    assert lf.comm_is_poisoned()
    for call in (lambda: lf.comm_init_rank(1, 0, uid), lambda: lf.comm_gather(pkg.SAMPLE_BUFFER),
                 lambda: lf.comm_gather_async(pkg.SAMPLE_BUFFER)):
        with pytest.raises(pkg.LensFlareError) as e:
            call()
        assert e.value.status == 4          # LF_ERR_STATE
    assert lf.comm_info() == (0, -1)        # no communicator was published
    lf.comm_abort()                         # still allowed, nothing to end
    _frame(lf, 4, 5)
    assert np.array_equal(lf.read_buffer(pkg.SAMPLE_BUFFER), before)
    lf.close()
    assert getattr(lf, "leaked", None) is False     # no call was left inside: freed


@pytest.mark.parametrize("n,W,H,spp", [(8, 1920, 1080, 16), (4, 1920, 1080, 16), (3, 1920, 1080, 16), (8, 3840, 2160, 4), (7, 1900, 1000, 9)])
def test_shared_cull_prepass_is_the_table_built_alone(pkg, cull_forced, n, W, H, spp):
    """The pre-pass shared between the ranks (lf_set_cull_share): n contexts on device 0, each builds
    the slab of table rows of the blocks b with b % n == rank (lf_cull_prepare), the test plays the all-gather (slab r
    of rank r into everybody's table, what ncclAllGather / sharding.complete_cull_table deliver), lf_cull_commit, then
    every rank marches its tile rows.  The table every rank ends with (lf_get_cull_table, block order) is the one a
    single context builds alone, the started fraction too, the gathered frame and the summed counters are the
    single-context frame's -- bit for bit; block counts that n does not divide (the last slabs are padded)."""
    import torch
    # (cull_forced: few samples, a wide table -- the launch would take the path tree by itself)
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")

    class View:
        def __init__(self, ptr, count):
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<i8", "data": (ptr, False), "version": 2}

    def setup(lf):
        lf.set_frame(W, H)
        _setup(pkg, lf, lens, mask)
        lf.set_march_culling(2)
        lf.reset_counters()

    one = pkg.LensFlare(0)
    setup(one)
    one.trace_ghosts(spp, 9)
    assert one.cull_info()["culled"]
    want, want_cnt, want_tab, want_frac = one.read_buffer(pkg.GHOST_BUFFER), one.counters(), one.cull_table(), one.cull_started_fraction()
    info = one.cull_info()
    one.close()
    assert want.max() > 0 and ((info["blocks_x"] * info["blocks_y"]) % n != 0 or n == 3)   # (n = 3 divides the 510 blocks: no padding)
    ranks = [pkg.LensFlare(0) for _ in range(n)]
    views = []
    for r, lf in enumerate(ranks):
        setup(lf)
        lf.set_row_interleave(r, n)
        lf.set_cull_share(r, n)
        with pytest.raises(pkg.LensFlareError):          # a shared table must be completed before the launch
            lf.trace_ghosts(spp, 9)
        lf.cull_prepare(spp)
        ptr, total, per_rank = lf.cull_table_view()
        assert total == per_rank * n and per_rank % (info["cells"] + 1) == 0
        views.append(torch.as_tensor(View(ptr, total), device="cuda:0"))
    k = views[0].numel() // n
    full = torch.cat([views[r][r * k:(r + 1) * k] for r in range(n)])
    for r in range(n):                                     # (a rank's table holds its own slab and zeros before the exchange)
        own = views[r].clone(); own[r * k:(r + 1) * k] = 0
        assert not own.any()
        views[r].copy_(full)
    torch.cuda.synchronize()
    got = np.zeros_like(want)
    total_cnt = {}
    for r, lf in enumerate(ranks):
        lf.cull_commit()
        lf.trace_ghosts(spp, 9)
        assert lf.cull_info()["culled"] and lf.cull_started_fraction() == want_frac
        assert np.array_equal(lf.cull_table(), want_tab)
        own = (np.arange(H) // 8) % n == r
        got[own] = lf.read_buffer(pkg.GHOST_BUFFER)[own]
        for key, v in lf.counters().items():
            total_cnt[key] = total_cnt.get(key, 0) + v
    assert np.array_equal(got, want)
    assert total_cnt == want_cnt
    # the next launch needs its own prepare / commit (mode 2 rebuilds): refused without
    with pytest.raises(pkg.LensFlareError):
        ranks[0].trace_ghosts(spp, 9)
    ranks[0].set_cull_share(0, 1)                          # sharing off: the rank builds the whole table again
    ranks[0].set_row_interleave(0, 1)
    ranks[0].trace_ghosts(spp, 9)
    assert np.array_equal(ranks[0].read_buffer(pkg.GHOST_BUFFER), want) and np.array_equal(ranks[0].cull_table(), want_tab)
    for lf in ranks:
        lf.close()


def test_table_all_gather_through_the_communicator(pkg, force_exchange, cull_forced):
    """lf_comm_share_cull: the in-place ncclAllGather of table slabs on the communicator's stream, between pre-pass and
    march -- with the one rank a one-GPU box can form (the test knob comm_force_exchange runs the collective all the same, as for
    the frame's exchange): the RCCL call, the stream hand-over and the launch order are the multi-rank ones, the frame
    is the plain one; frames in a row with the frame's own asynchronous exchange in between (one stream carries every
    RCCL call); refused without a communicator, switched off by lf_comm_abort."""
    if not pkg.comm_available():
        pytest.skip("librccl.so.1 not loadable")
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 1920, 1080, 16
    plain = pkg.LensFlare(0)
    plain.set_frame(W, H)
    _setup(pkg, plain, lens, mask)
    plain.set_march_culling(2)
    _frame(plain, spp, 9)
    want, want_tab = plain.read_buffer(pkg.SAMPLE_BUFFER), plain.cull_table()
    plain.close()
    lf = pkg.LensFlare(0)
    lf.set_frame(W, H)
    _setup(pkg, lf, lens, mask)
    lf.set_march_culling(2)
    with pytest.raises(pkg.LensFlareError):
        lf.comm_share_cull(True)                       # no communicator yet
    lf.comm_init_rank(1, 0, pkg.comm_unique_id())
    lf.comm_share_cull(True)
    for _ in range(3):
        _frame(lf, spp, 9)
        lf.comm_gather_async(pkg.SAMPLE_BUFFER)
    lf.comm_wait()
    assert lf.cull_info()["culled"]
    assert np.array_equal(lf.read_buffer(pkg.SAMPLE_BUFFER), want) and np.array_equal(lf.cull_table(), want_tab)
    lf.comm_abort()
    _frame(lf, spp, 9)                                  # the communicator is gone: the rank builds its table alone
    assert np.array_equal(lf.read_buffer(pkg.SAMPLE_BUFFER), want)
    lf.close()


@pytest.mark.parametrize("n", [2, 7])
def test_group_shares_the_cull_prepass(pkg, cull_forced, n):
    """lf_group_share_cull (one process, n devices; here n contexts on device 0, whose all-gather is the peer-copy
    stand-in): every context builds its slab, the group completes the table, the gathered 1080p frame is the
    single-context frame, bit for bit; a frame later without the call is refused (mode 2: the table is per launch)."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 1920, 1080, 16
    one = pkg.LensFlare(0)
    one.set_frame(W, H)
    _setup(pkg, one, lens, mask)
    one.set_march_culling(2)
    _frame(one, spp, 9)
    want, want_tab = one.read_buffer(pkg.SAMPLE_BUFFER), one.cull_table()
    one.close()
    grp = pkg.LensFlareGroup([0] * n)
    grp.set_frame(W, H)
    for r in grp.ranks:
        _setup(pkg, r, lens, mask)
        r.set_march_culling(2)
    for _ in range(2):
        grp.share_cull(spp)
        grp.for_each(lambda lf, rank: _frame(lf, spp, 9))
        grp.gather(pkg.SAMPLE_BUFFER)
    for lf in grp.ranks:
        assert lf.cull_info()["culled"]
        assert np.array_equal(lf.read_buffer(pkg.SAMPLE_BUFFER), want) and np.array_equal(lf.cull_table(), want_tab)
    with pytest.raises(pkg.LensFlareError):
        grp.for_each(lambda lf, rank: _frame(lf, spp, 9))
    grp.close()


def test_bench_goes_through_the_multi_rank_bring_up_with_one_rank(pkg):
    """`bench.py --gpus N` as far as one GPU can take it (LF_BENCH_SOLO_COMM=1): gloo control plane, communicator id,
    ncclCommInitRank and the first exchange under their deadlines, lf_comm_share_cull, the first frame with a
    communicator-completed cull table under the deadline, the timed frames with table and frame all-gathers -- and the
    JSON line says so: the C ABI's exchange, the pre-pass shared through RCCL, the single-GPU frame's event count."""
    import json
    import os
    import subprocess
    import sys
    if not pkg.comm_available():
        pytest.skip("librccl.so.1 not loadable")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LF_BENCH_SOLO_COMM="1", LF_BENCH_COMM_TIMEOUT="120")
    lines = {}
    for deal in ("blocks", "rows"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu", "--config", "c2"],
                           capture_output=True, text=True, timeout=600, env=dict(env, LF_BENCH_DEAL=deal), cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["config"]["gather_mode"] == "cabi" and line["config"]["gather_note"] is None and line["config"]["deal"] == deal
        assert line["culling"]["culled"] and line["culling"]["audit"]["lit"] == 0
        assert line["rccl_nranks"] == 1 and line["n_gpus"] == 1
        lines[deal] = line
    # dealt by blocks (the default) nothing of the table is exchanged; dealt by rows the communicator completes it
    assert lines["blocks"]["culling"]["prepass_shared_between_ranks"].startswith("not needed")
    assert lines["rows"]["culling"]["prepass_shared_between_ranks"] == "rccl (C ABI)"
    line = lines["blocks"]
    plain = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu", "--config", "c2"],
                           capture_output=True, text=True, timeout=600, env={k: v for k, v in env.items() if k != "LF_BENCH_SOLO_COMM"}, cwd=root)
    assert plain.returncode == 0, plain.stderr[-3000:]
    want = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    for line in lines.values():
        assert line["config"]["events_executed_per_frame"] == want["config"]["events_executed_per_frame"]
        assert line["culling"]["started_fraction"] == want["culling"]["started_fraction"]
    # a bring-up call that does not come back within its deadline (here: a deadline nothing can meet): the helper thread is
    # abandoned INSIDE the context, which is poisoned and never used again -- the bench goes on with a fresh context and the
    # torch.distributed exchange, says so, and renders the same frame
    late = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu", "--config", "c2"],
                          capture_output=True, text=True, timeout=600, env=dict(env, LF_BENCH_COMM_TIMEOUT="0.000001"), cwd=root)
    assert late.returncode == 0, late.stderr[-3000:]
    line = json.loads([l for l in late.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["gather_mode"] == "torch" and "deadline" in line["config"]["gather_note"]
    assert line["config"]["events_executed_per_frame"] == want["config"]["events_executed_per_frame"]
    assert line["culling"]["culled"] and line["culling"]["audit"]["lit"] == 0


# ---- round 6: the frame dealt by BLOCKS of 64 x 64 pixels (lf_set_block_deal) -------------------------------------------
def _block_owner(W, H, n):
    """(H, W) array: the rank that owns each pixel under the block deal"""
    bx = (W + 63) // 64
    yy, xx = np.mgrid[0:H, 0:W]
    return ((yy // 64) * bx + xx // 64) % n


@pytest.mark.parametrize("n,W,H,spp", [(8, 1920, 1080, 16), (3, 1920, 1080, 16), (7, 1900, 1000, 9), (4, 3840, 2160, 4)])
def test_block_deal_is_the_single_context_frame_with_no_table_exchange(pkg, cull_forced, n, W, H, spp):
    """n contexts on device 0, each with the blocks b % n == rank: every context builds, audits and reads ONLY its own rows of
    the cull table (no prepare / view / commit, no all-gather: nothing is exchanged but the finished blocks), the frame put
    together from everybody's own blocks -- flare layer included -- is the single context's bit for bit, the counters and the
    audit's rays sum to its, a rank's table rows are the single table's rows of its blocks and nothing else."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")

    def setup(lf):
        lf.set_frame(W, H)
        _setup(pkg, lf, lens, mask)
        lf.set_march_culling(2)
        lf.reset_counters()

    one = pkg.LensFlare(0)
    setup(one)
    _frame(one, spp, 9)
    assert one.cull_info()["culled"]
    info = one.cull_info()
    want_sample, want_ghost = one.read_buffer(pkg.SAMPLE_BUFFER), one.read_buffer(pkg.GHOST_BUFFER)
    want_cnt, want_tab, want_audit = one.counters(), one.cull_table(), one.cull_audit()
    # (the single context may take blocks of 128 pixels at 4K with few samples; dealt by blocks the table's block is the deal's)
    one.close()
    owner = _block_owner(W, H, n)
    got_sample, got_ghost = np.zeros_like(want_sample), np.zeros_like(want_ghost)
    total_cnt, audit_rays, started = {}, 0, 0.0
    for r in range(n):
        lf = pkg.LensFlare(0)
        setup(lf)
        lf.set_block_deal(r, n)
        _frame(lf, spp, 9)
        ci = lf.cull_info()
        assert ci["culled"] and ci["block_px"] == 64, (ci, lf.cull_reason())
        mine = owner == r
        got_sample[mine] = lf.read_buffer(pkg.SAMPLE_BUFFER)[mine]
        got_ghost[mine] = lf.read_buffer(pkg.GHOST_BUFFER)[mine]
        for key, v in lf.counters().items():
            total_cnt[key] = total_cnt.get(key, 0) + v
        a = lf.cull_audit()
        assert a["lit"] == 0 and a["rays"] > 0
        audit_rays += a["rays"]
        tab = lf.cull_table().reshape(-1, ci["cells"] + 1)
        own_blocks = np.arange(tab.shape[0]) % n == r
        assert not tab[~own_blocks].any()                  # nobody else's rows were built
        if info["block_px"] == 64:
            assert np.array_equal(tab[own_blocks], want_tab.reshape(-1, ci["cells"] + 1)[own_blocks])
        started += lf.cull_started_fraction() * own_blocks.sum()
        lf.close()
    assert np.array_equal(got_ghost, want_ghost) and np.array_equal(got_sample, want_sample) and want_ghost.max() > 0
    if info["block_px"] == 64:
        assert total_cnt == want_cnt and audit_rays == want_audit["rays"]
    else:
        assert total_cnt["rays_hit_light"] == want_cnt["rays_hit_light"]


@pytest.mark.parametrize("n", [2, 5])
def test_group_dealt_by_blocks_gathers_the_single_context_frame(pkg, cull_forced, n):
    """lf_group_set_block_deal (one process, n devices; here n contexts on device 0 and the peer-copy stand-in of the
    all-gather): blocks packed, exchanged and unpacked -- every context ends with the single-context 1080p frame, bit for
    bit; lf_group_share_cull has nothing to do; back to tile rows: the same frame."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 1920, 1080, 16
    one = pkg.LensFlare(0)
    one.set_frame(W, H)
    _setup(pkg, one, lens, mask)
    one.set_march_culling(2)
    _frame(one, spp, 9)
    want = one.read_buffer(pkg.SAMPLE_BUFFER)
    one.close()
    grp = pkg.LensFlareGroup([0] * n)
    grp.set_frame(W, H)
    for r in grp.ranks:
        _setup(pkg, r, lens, mask)
        r.set_march_culling(2)
    grp.set_block_deal(True)
    for _ in range(2):
        grp.share_cull(spp)                      # (nothing to share: returns at once)
        grp.for_each(lambda lf, rank: _frame(lf, spp, 9))
        grp.gather(pkg.SAMPLE_BUFFER)
    for lf in grp.ranks:
        assert lf.cull_info()["culled"]
        assert np.array_equal(lf.read_buffer(pkg.SAMPLE_BUFFER), want)
    grp.set_block_deal(False)
    grp.share_cull(spp)
    grp.for_each(lambda lf, rank: _frame(lf, spp, 9))
    grp.gather(pkg.SAMPLE_BUFFER)
    for lf in grp.ranks:
        assert np.array_equal(lf.read_buffer(pkg.SAMPLE_BUFFER), want)
    grp.close()


def test_block_deal_exchange_through_the_communicator(pkg, force_exchange, cull_forced):
    """the blocks' exchange on the C ABI's RCCL path with the one rank a one-GPU box can form: pack, ncclAllGather, unpack
    of 64 x 64 blocks (synchronous and on the exchange's own stream), the plan's unit is a block"""
    if not pkg.comm_available():
        pytest.skip("librccl.so.1 not loadable")
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 1900, 1000, 16
    lf = pkg.LensFlare(0)
    lf.set_frame(W, H)
    _setup(pkg, lf, lens, mask)
    lf.set_march_culling(2)
    lf.comm_init_rank(1, 0, pkg.comm_unique_id())
    lf.set_block_deal(0, 1)
    _frame(lf, spp, 9)
    want = lf.read_buffer(pkg.SAMPLE_BUFFER)
    lf.comm_gather(pkg.SAMPLE_BUFFER)
    assert np.array_equal(lf.read_buffer(pkg.SAMPLE_BUFFER), want)
    for _ in range(2):
        _frame(lf, spp, 9)
        lf.comm_gather_async(pkg.SAMPLE_BUFFER)
    lf.comm_wait()
    assert np.array_equal(lf.read_buffer(pkg.SAMPLE_BUFFER), want) and want.max() > 0
    lf.close()
