"""The N>1 path on CPU: world_size 2 and 3 over gloo.  Each rank fills its round-robin tile rows of
a frame (with the CPU oracle standing in for the GPU march, which the -m gpu tests already pin to
it bit for bit), then the product's own exchange step (lens_flare_amd.sharding, the code bench.py
runs over RCCL) must rebuild the complete frame on every rank."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, spp, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as g
    pkg = g.load_package()
    from lens_flare_amd import sharding
    from oracle import lfo
    from goldenlib import load_texels
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    sun = ([0.03, 0.02, -1.0], [1.0, 0.9, 0.5], 0.05)
    rows = sharding.padded_rows(H, world)
    frame = torch.zeros(rows * W * 3, dtype=torch.float64)
    view = frame.numpy().reshape(rows, W, 3)
    events = 0
    for t in sharding.my_tile_rows(H, rank, world):
        y0, y1 = t * 8, min(H, t * 8 + 8)
        g_, c = lfo.geo_trace(lens, W, H, y0, y1, spp, 5, None, True, mask, *sun, n_threads=2)
        view[y0:y1] = g_[y0:y1]
        events += c["surface_events"]
    mine = frame.clone()
    sharding.gather_frame_inplace(frame, W, H, rank, world, dist)
    # the single-collective variant (what bench.py runs) must produce the same frame
    sharding.gather_frame(mine, W, H, rank, world, dist)
    assert torch.equal(mine, frame)
    ev = torch.tensor([float(events)], dtype=torch.float64)
    dist.all_reduce(ev)
    np.save(os.path.join(out_dir, f"frame_{rank}.npy"), view[:H].copy())
    np.save(os.path.join(out_dir, f"events_{rank}.npy"), ev.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_row_deal_and_inplace_gather(world, tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package()
    from oracle import lfo
    from goldenlib import load_texels
    W, H, spp = 24, 44, 4   # 6 tile rows, the last one partial (4 rows)
    mp.spawn(_worker, args=(world, _free_port(), W, H, spp, str(tmp_path)), nprocs=world, join=True)
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    full, cnt = lfo.geo_trace(lens, W, H, 0, H, spp, 5, None, True, mask, [0.03, 0.02, -1.0],
                              [1.0, 0.9, 0.5], 0.05, n_threads=4)
    assert full.max() > 0
    for r in range(world):
        got = np.load(tmp_path / f"frame_{r}.npy")
        assert np.array_equal(got, full), f"rank {r} did not end up with the complete frame"
        assert np.load(tmp_path / f"events_{r}.npy")[0] == cnt["surface_events"]


def test_sharding_helpers():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    from lens_flare_amd import sharding
    assert sharding.n_tile_rows(1080) == 135
    for world in (1, 2, 4, 8):
        rows = [sharding.my_tile_rows(1080, r, world) for r in range(world)]
        assert sorted(sum(rows, [])) == list(range(135))
        assert max(map(len, rows)) - min(map(len, rows)) <= 1
        assert sharding.padded_rows(1080, world) <= 1088  # fits the library's 64-row padding


# ---- a rank that fails alone must not leave its peers hanging ----------------------------------------
def _bringup_worker(rank, world, port, out_dir, fail_rank, stage):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    from lens_flare_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []
    # stage "preflight": rank `fail_rank` cannot load the communication library
    ok, bad = sharding.agree(dist, not (stage == "preflight" and rank == fail_rank), "librccl.so.1 not found")
    log.append(("preflight", ok, bad))
    if ok:
        # stage "exchange": rank `fail_rank` raises when it enqueues; the others' collective can then
        # never complete (their test() stays False) -- they must give up at the deadline, not block
        state = {"enqueued": False}

        def enqueue():
            if stage == "exchange" and rank == fail_rank:
                raise RuntimeError("LF_ERR_HIP: injected launch failure")
            if stage == "blocked" and rank != fail_rank:
                # what RCCL does when a peer died before joining: the call never returns
                import threading
                threading.Event().wait()
            state["enqueued"] = True

        def test():
            return stage not in ("exchange",)     # completes at once unless a peer is missing

        ok, bad = sharding.first_exchange(dist, enqueue, test, timeout_s=0.3, poll_s=0.01)
        log.append(("exchange", ok, bad))
    import json
    json.dump(log, open(os.path.join(out_dir, f"log_{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("stage", ["none", "preflight", "exchange", "blocked"])
def test_one_rank_failing_alone_is_noticed_by_all(stage, tmp_path):
    """bench.py's bring-up of the C ABI's RCCL communicator (lens_flare_amd.sharding.agree /
    first_exchange): whatever a single rank's local failure, every rank returns, with the same verdict
    and the failing rank's reason."""
    import json
    import torch.multiprocessing as mp
    world, fail_rank = 3, 1
    mp.spawn(_bringup_worker, args=(world, _free_port(), str(tmp_path), fail_rank, stage), nprocs=world, join=True)
    logs = [json.load(open(tmp_path / f"log_{r}.json")) for r in range(world)]
    assert all(l == logs[0] for l in logs)                      # one verdict
    # ... and one answer to "does a helper thread still stand in some rank's context?" (sharding.expired: bench.py then goes
    # on with a FRESH context on every rank instead of the one the abandoned call may still write to)
    expired = _sharding().expired
    if stage == "none":
        assert [x[:2] for x in logs[0]] == [["preflight", True], ["exchange", True]]
    elif stage == "preflight":
        assert logs[0] == [["preflight", False, ["rank 1: librccl.so.1 not found"]]]   # no exchange was attempted
        assert not expired(logs[0][0][2])
    elif stage == "blocked":
        assert expired(logs[0][1][2])
        # the two healthy ranks' enqueue blocks on the host for ever (their peer "died"): the helper
        # thread is abandoned at the deadline, every rank still reaches the verdict
        name, ok, bad = logs[0][1]
        assert name == "exchange" and not ok and len(bad) == world - 1
        assert all("blocked for more than" in b for b in bad)
    else:
        name, ok, bad = logs[0][1]
        assert name == "exchange" and not ok and len(bad) == world
        assert "rank 1: RuntimeError: LF_ERR_HIP: injected launch failure" in bad
        assert all("did not complete within" in b for b in bad if not b.startswith("rank 1"))
        assert expired(bad) and not expired(["rank 1: RuntimeError: LF_ERR_HIP: injected launch failure"])


# ---- `python bench.py --gpus N` starts its own ranks (round 5) ------------------------------------------
_PROBE = """
import json, os, sys
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
print("noise from rank", rank)                       # not JSON: must go to the launcher's stderr side
if rank == 0:
    print(json.dumps({"rank": rank, "world": world, "argv": sys.argv[1:], "local": os.environ["LOCAL_RANK"],
                      "master": os.environ["MASTER_ADDR"], "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))
sys.exit(int(os.environ.get("PROBE_FAIL_CODE", "3")) if rank == int(os.environ.get("PROBE_FAIL_RANK", "-1")) else 0)
"""


def _sharding():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    from lens_flare_amd import sharding
    return sharding


def test_self_launch_decision_and_command():
    sharding = _sharding()
    assert not sharding.needs_self_launch(1, {})
    assert sharding.needs_self_launch(2, {})
    assert sharding.needs_self_launch(8, {"HOME": "/root"})
    # under a launcher (torchrun sets both) behaviour is unchanged: the script is a rank, not a launcher
    assert not sharding.needs_self_launch(8, {"WORLD_SIZE": "8", "RANK": "3"})
    assert not sharding.needs_self_launch(2, {"RANK": "0"})
    cmd = sharding.launch_command("/x/bench.py", ["--gpus", 4, "--steps", "7", "--warmup", 2], 4, 29517, python="py")
    assert cmd[:3] == ["py", "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index("/x/bench.py")
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]   # the script's own arguments, in order, last


def test_self_launch_relays_arguments_and_json(tmp_path):
    import io
    import json
    sharding = _sharding()
    script = tmp_path / "probe.py"
    script.write_text(_PROBE)
    out, err = io.StringIO(), io.StringIO()
    env = {k: v for k, v in os.environ.items() if k not in ("PROBE_FAIL_RANK",)}
    env["WORLD_SIZE"] = "99"     # a stale variable of the caller must not reach the ranks
    rc = sharding.self_launch(str(script), ["--gpus", "2", "--steps", "3", "--config", "c3"], 2, environ=env, out=out, err=err)
    assert rc == 0, err.getvalue()
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1, out.getvalue()                      # ONE JSON line: rank 0's
    rec = json.loads(lines[0])
    assert rec["world"] == 2 and rec["rank"] == 0 and rec["local"] == "0" and rec["master"] == "127.0.0.1"
    assert rec["argv"] == ["--gpus", "2", "--steps", "3", "--config", "c3"]
    assert rec["ipc"] == "0"
    assert "noise from rank 1" in err.getvalue() and "noise" not in out.getvalue()


def test_self_launch_exit_code_when_a_rank_fails(tmp_path):
    import io
    sharding = _sharding()
    script = tmp_path / "probe.py"
    script.write_text(_PROBE)
    out, err = io.StringIO(), io.StringIO()
    env = dict(os.environ, PROBE_FAIL_RANK="1", PROBE_FAIL_CODE="5")
    rc = sharding.self_launch(str(script), ["--gpus", "2"], 2, environ=env, out=out, err=err)
    assert rc != 0                                              # a failed rank is a failed run


def test_bench_refuses_a_rank_count_that_disagrees_with_gpus():
    """under an existing launcher environment bench.py stays a rank: --gpus must equal WORLD_SIZE (checked
    before torch.cuda is touched)"""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


# ---- a blocked bring-up call that comes back AFTER the verdict ---------------------------------
def test_a_call_that_returns_after_the_deadline_publishes_nothing():
    """call_with_deadline gives up on a blocked call, tells the callee's side (on_expire = lf_comm_poison) BEFORE it
    reports the expiry, and ignores whatever the call returns or raises later: a late return must not look like a
    success to anybody."""
    import threading
    import time
    sharding = _sharding()
    release, came_back = threading.Event(), threading.Event()
    log = []

    class FakeCtx:                       # the protocol of lf_comm_init_rank / lf_comm_poison (lf_group.hip)
        def __init__(self):
            self.mu, self.poisoned, self.comm = threading.Lock(), False, None

        def init_rank(self):
            release.wait()               # ncclCommInitRank blocked on a peer that never joined ... until much later
            with self.mu:
                if self.poisoned:
                    log.append("late communicator aborted, not published")
                else:
                    self.comm = "communicator"
            came_back.set()

        def poison(self):
            with self.mu:
                self.poisoned = True
            log.append("poisoned")

    ctx = FakeCtx()
    t0 = time.monotonic()
    done, err = sharding.call_with_deadline(ctx.init_rank, 0.2, on_expire=ctx.poison)
    assert (done, err) == (False, None) and time.monotonic() - t0 < 5.0
    assert log == ["poisoned"]           # the callee knew before the caller went on
    # ... the caller takes its verdict, aborts, falls back; THEN the blocked call returns
    release.set()
    assert came_back.wait(5.0)
    assert ctx.comm is None and log == ["poisoned", "late communicator aborted, not published"]
    # a call that makes the deadline is reported as before, and on_expire stays out of it
    calls = []
    assert sharding.call_with_deadline(lambda: calls.append(1), 5.0, on_expire=lambda: calls.append("expired")) == (True, None)
    assert calls == [1]
    boom = RuntimeError("x")

    def raises():
        raise boom
    assert sharding.call_with_deadline(raises, 5.0) == (True, boom)


def _table_worker(rank, world, port, blocks, row_entries, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    from lens_flare_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nb = (blocks + world - 1) // world                     # rows per slab: equal slabs, the last ones padded
    table = torch.zeros(nb * world * row_entries, dtype=torch.int64)
    rows = table.view(nb * world, row_entries)
    for b in range(rank, blocks, world):                   # the blocks this rank builds, at lf_cull_row_of_block(b)
        rows[(b % world) * nb + b // world] = torch.arange(row_entries, dtype=torch.int64) + 1000 * (b + 1)
    sharding.complete_cull_table(table, rank, world, dist)
    np.save(os.path.join(out_dir, f"table_{rank}.npy"), table.numpy())
    empty = torch.zeros(0, dtype=torch.int64)              # a launch that does not cull: nothing to exchange, no collective
    sharding.complete_cull_table(empty, rank, world, dist)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,blocks", [(2, 7), (3, 10)])
def test_shared_cull_table_is_completed_by_one_all_gather(world, blocks, tmp_path):
    """The host-owned exchange of the shared pre-pass (lf_set_cull_share; bench.py's torch / rehearsal modes): every
    rank holds its slab of rows (blocks dealt round robin, rows of one rank together), ONE in-place all-gather of equal
    slabs -- the layout lf_cull_table_view hands out -- and every rank holds every block's row where
    lf_cull_row_of_block puts it, the padding rows of the short slabs still zero."""
    import torch.multiprocessing as mp
    row_entries = 5
    port = _free_port()
    mp.spawn(_table_worker, args=(world, port, blocks, row_entries, str(tmp_path)), nprocs=world, join=True)
    nb = (blocks + world - 1) // world
    want = np.zeros((nb * world, row_entries), np.int64)
    for b in range(blocks):
        want[(b % world) * nb + b // world] = np.arange(row_entries) + 1000 * (b + 1)
    assert (want.sum(axis=1) == 0).sum() == nb * world - blocks
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"table_{r}.npy").reshape(nb * world, row_entries), want)


# ---- round 6: the frame dealt by blocks of 64 x 64 pixels --------------------------------------------------------------
def _block_worker(rank, world, port, W, H, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    from lens_flare_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    want = np.arange(H * W * 3, dtype=np.float64).reshape(H, W, 3) + 1.0       # the frame every rank must end with
    frame = torch.zeros(H * W * 3 + 17, dtype=torch.float64)                   # (a buffer longer than the frame: padding untouched)
    view = frame.numpy()[:H * W * 3].reshape(H, W, 3)
    bx = (W + 63) // 64
    for b in sharding.my_blocks(W, H, rank, world):                           # this rank "renders" its own blocks
        y0, x0 = (b // bx) * 64, (b % bx) * 64
        view[y0:y0 + 64, x0:x0 + 64] = want[y0:y0 + 64, x0:x0 + 64]
    scratch = {}
    for _ in range(2):                                                         # (the staging is reused between frames)
        sharding.gather_blocks(frame, W, H, rank, world, dist, scratch=scratch)
    np.save(os.path.join(out_dir, f"blocks_{rank}.npy"), frame.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(2, 200, 150), (3, 130, 64), (3, 64, 200)])
def test_block_deal_and_its_gather(world, W, H, tmp_path):
    """lf_set_block_deal's exchange as bench.py's torch / rehearsal modes run it (sharding.gather_blocks; inside the C ABI:
    lf_group.hip k_pack_blocks / k_unpack_blocks, tests/test_gpu_multi.py): blocks b % world == rank rendered in place,
    packed (partial blocks at the right and lower edges, fewer blocks than world x groups), ONE all-gather, unpacked --
    every rank holds the whole frame, nothing outside it is touched; the deal covers every block exactly once."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    from lens_flare_amd import sharding
    owned = sorted(b for r in range(world) for b in sharding.my_blocks(W, H, r, world))
    assert owned == list(range(sharding.n_blocks(W, H)))
    mp.spawn(_block_worker, args=(world, _free_port(), W, H, str(tmp_path)), nprocs=world, join=True)
    want = np.arange(H * W * 3, dtype=np.float64) + 1.0
    for r in range(world):
        got = np.load(tmp_path / f"blocks_{r}.npy")
        assert np.array_equal(got[:H * W * 3], want) and not got[H * W * 3:].any(), r
