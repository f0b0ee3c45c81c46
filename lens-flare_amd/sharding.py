"""Sensor sharding across GPUs (one process per GPU, torch.distributed; backend nccl = RCCL over
xGMI on the GPU box, gloo on CPU in tests).

The reference parallelises a frame over 32x32 sensor tiles pulled from a mutex-guarded queue by
std::threads (src/pathtracer/raytraced_renderer.cpp:314-328, :681-715; src/util/work_queue.h).
Here the unit is the march kernel's 8-row sensor *tile row*: tile row t belongs to rank t % world
(round-robin, which evens out the vignetting-dependent cost between the frame's centre and edges),
every rank renders its tile rows into its own full-frame buffer, and the only data-path collective
is the exchange of finished tile rows.  A group of `world` consecutive tile rows is contiguous in the
row-major frame and rank r owns slot r of each group, so the frame viewed as
[groups][world][tile-row elements] is exchanged with ONE all-gather per frame (gather_frame: pack
this rank's slots, all_gather_into_tensor, unpack -- two device copies of 1/world and 1 frame) or,
without any staging, with one in-place all-gather per group (gather_frame_inplace: `groups` small
collectives, latency-bound on xGMI when the frame is split 8 ways).

Round 6: the frame can also be dealt by BLOCKS of 64 x 64 pixels (block b, row-major, belongs to rank b % world:
lf_set_block_deal) -- the block is the path cull's, so a rank builds, audits and reads only its own rows of the cull
table and nothing but finished blocks is exchanged (gather_blocks: pack this rank's blocks, ONE all-gather, unpack).
"""
TILE_ROWS = 8
BLOCK = 64


def n_tile_rows(H):
    return (H + TILE_ROWS - 1) // TILE_ROWS


def my_tile_rows(H, rank, world):
    return [t for t in range(n_tile_rows(H)) if t % world == rank]


def padded_rows(H, world):
    """Rows the frame buffer must hold so that every group of `world` tile rows is addressable."""
    groups = (n_tile_rows(H) + world - 1) // world
    return groups * world * TILE_ROWS


def gather_frame_inplace(frame, W, H, rank, world, dist, elems_per_pixel=3):
    """frame: flat tensor of >= padded_rows(H, world) * W * elems_per_pixel elements holding this
    rank's tile rows at their final position.  After the call rows [0, H) are complete everywhere."""
    if world == 1:
        return
    e = TILE_ROWS * W * elems_per_pixel
    groups = (n_tile_rows(H) + world - 1) // world
    assert frame.numel() >= groups * world * e, "frame buffer is not padded for in-place gathers"
    for grp in range(groups):
        out = frame[grp * world * e:(grp + 1) * world * e]
        dist.all_gather_into_tensor(out, out[rank * e:(rank + 1) * e])


def gather_frame(frame, W, H, rank, world, dist, elems_per_pixel=3, scratch=None):
    """Same result as gather_frame_inplace with a single collective.  scratch: optional dict that
    keeps the two staging tensors between frames."""
    if world == 1:
        return
    e = TILE_ROWS * W * elems_per_pixel
    groups = (n_tile_rows(H) + world - 1) // world
    assert frame.numel() >= groups * world * e, "frame buffer is not padded for the gather"
    v = frame[:groups * world * e].view(groups, world, e)
    if scratch is None:
        scratch = {}
    send = scratch.get("send")
    if send is None or send.shape != (groups, e) or send.device != frame.device:
        send = scratch["send"] = frame.new_empty((groups, e))
        scratch["recv"] = frame.new_empty((world, groups, e))
    recv = scratch["recv"]
    send.copy_(v[:, rank, :])
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
    v.copy_(recv.permute(1, 0, 2))


def n_blocks(W, H):
    return ((W + BLOCK - 1) // BLOCK) * ((H + BLOCK - 1) // BLOCK)


def my_blocks(W, H, rank, world):
    """the 64 x 64-pixel blocks (row-major indices) of rank `rank` under the block deal"""
    return list(range(rank, n_blocks(W, H), world))


def gather_blocks(frame, W, H, rank, world, dist, elems_per_pixel=3, scratch=None):
    """The block deal's exchange (what lf_comm_gather does inside the C ABI: lf_group.hip k_pack_blocks / k_unpack_blocks).
    frame: flat tensor of >= H * W * elems_per_pixel elements, this rank's blocks rendered in place.  Every rank packs its
    blocks (padded to 64 x 64 at the frame's edges, and to ceil(blocks / world) per rank), ONE all-gather, everybody's
    blocks unpacked.  After the call rows [0, H) are complete everywhere."""
    if world == 1:
        return
    bx = (W + BLOCK - 1) // BLOCK
    nblk = n_blocks(W, H)
    groups = (nblk + world - 1) // world
    img = frame[:H * W * elems_per_pixel].view(H, W, elems_per_pixel)
    if scratch is None:
        scratch = {}
    send = scratch.get("bsend")
    if send is None or send.shape != (groups, BLOCK, BLOCK, elems_per_pixel) or send.device != frame.device:
        send = scratch["bsend"] = frame.new_zeros((groups, BLOCK, BLOCK, elems_per_pixel))
        scratch["brecv"] = frame.new_empty((world, groups, BLOCK, BLOCK, elems_per_pixel))
    recv = scratch["brecv"]

    def box(b):
        y0, x0 = (b // bx) * BLOCK, (b % bx) * BLOCK
        return y0, min(H, y0 + BLOCK), x0, min(W, x0 + BLOCK)

    for g in range(groups):
        b = g * world + rank
        if b < nblk:
            y0, y1, x0, x1 = box(b)
            send[g, :y1 - y0, :x1 - x0] = img[y0:y1, x0:x1]
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
    for r in range(world):
        if r == rank:
            continue
        for g in range(groups):
            b = g * world + r
            if b < nblk:
                y0, y1, x0, x1 = box(b)
                img[y0:y1, x0:x1] = recv[r, g, :y1 - y0, :x1 - x0]


def complete_cull_table(table, rank, world, dist):
    """The cull pre-pass shared between the ranks (include/lensflare.h, lf_set_cull_share): `table` is the flat table
    of lf_cull_table_view -- `world` equal slabs, this rank's slab built (lf_cull_prepare), the others still zero.
    ONE in-place all-gather completes it everywhere; lf_cull_commit comes next.  An empty table (this launch does
    not cull: the same on every rank, the inputs are the same) needs nothing."""
    if world == 1 or table.numel() == 0:
        return
    k = table.numel() // world
    assert k * world == table.numel(), "the table is not made of equal slabs"
    dist.all_gather_into_tensor(table, table[rank * k:(rank + 1) * k])


# ---- one rank per GPU: who starts them --------------------------------------------------------------
# The reference scales out inside its host: start_raytracing() spawns numWorkerThreads std::threads over
# one tile queue (raytraced_renderer.cpp:352-354).  Here a rank is a PROCESS (one per GPU), so something has
# to start N of them: either the caller (python -m torch.distributed.run ... script --gpus N: RANK /
# WORLD_SIZE are in the environment) or, when the script is started plainly as `python script --gpus N`,
# the script itself -- through launch_command / self_launch below, BEFORE anything in it touches the GPU.
def launch_command(script, argv, n_ranks, port, python=None):
    """The torch.distributed.run command line that starts n_ranks copies of `script argv` on this node
    (static rendezvous on 127.0.0.1: the container's hostname may not resolve)."""
    import sys
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            f"--nproc-per-node={int(n_ranks)}", "--master-addr", "127.0.0.1", "--master-port", str(int(port)),
            script] + [str(a) for a in argv]


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def needs_self_launch(n_ranks, environ):
    """True when the script was asked for n_ranks > 1 and nobody has started the ranks yet."""
    return int(n_ranks) > 1 and "WORLD_SIZE" not in environ and "RANK" not in environ


def self_launch(script, argv, n_ranks, environ=None, out=None, err=None, timeout_s=None):
    """Start the n_ranks processes as CHILDREN (never an exec: the caller keeps running, and must not have
    touched the GPU), relay what they print -- lines that are JSON objects to `out`, everything else to `err`
    -- and return the launcher's exit code: non-zero as soon as any rank fails (torch.distributed.run ends the
    others), so that a failed rank is a failed run, never a restart."""
    import os
    import subprocess
    import sys
    out = out or sys.stdout
    err = err or sys.stderr
    env = dict(os.environ if environ is None else environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL across processes needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = launch_command(script, argv, n_ranks, free_port())
    print("self-launch: " + " ".join(cmd), file=err, flush=True)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None if err is sys.stderr else subprocess.STDOUT,
                         text=True, bufsize=1)
    try:
        for line in p.stdout:
            (out if line.lstrip().startswith("{") else err).write(line)
            (out if line.lstrip().startswith("{") else err).flush()
        return p.wait(timeout=timeout_s)
    except BaseException:
        p.kill()
        p.wait()
        raise


# ---- bringing up the data-path communicator without ever hanging --------------------------------
# One process per GPU: a rank that fails LOCALLY (library missing, bad argument, a launch error) while
# its peers are inside a blocking collective leaves them there for good.  Two rules avoid that:
# (1) nothing blocking is called before every rank has said, over the control plane (gloo), that it is
# able to; (2) the first collective on a new communicator is waited for with a deadline, and the
# verdict is again taken together.  bench.py follows both; tests/test_sharding_gloo.py injects a
# one-rank failure into each.
def agree(dist, ok, why=""):
    """Every rank passes its own verdict; returns (all ok?, reasons of the ranks that were not).
    Control-plane collective (gloo): every rank MUST reach it -- callers catch their local errors and
    pass ok=False instead of raising."""
    verdicts = [None] * dist.get_world_size()
    dist.all_gather_object(verdicts, (bool(ok), str(why)))
    bad = [f"rank {r}: {w or 'failed'}" for r, (k, w) in enumerate(verdicts) if not k]
    return not bad, bad


def call_with_deadline(fn, timeout_s, on_expire=None):
    """Run a HOST-BLOCKING call (ncclCommInitRank, or the first collective on a fresh communicator, whose
    transport set-up blocks the calling thread until every peer has joined) in a helper thread and wait for
    it at most timeout_s.  Returns (done, error): (True, None) when fn returned, (True, exception) when it
    raised, (False, None) when it is still blocked -- the thread is then abandoned (a daemon: it cannot keep
    the process alive) and the caller must treat the communicator as lost (lf_comm_abort, fall back or
    exit).  Never re-executes anything: a process that has touched the GPU must not exec.

    The abandoned thread is still INSIDE the call and may come back at any later time.  on_expire (the C ABI's
    lf_comm_poison: LensFlare.comm_poison) runs before this function reports the expiry and tells the callee's
    side that whatever the call still produces must not be published: a communicator that arrives late is
    aborted where it stands, the context is leaked rather than freed under the blocked thread.  On this side
    nothing the thread returns or raises after the deadline is looked at: its result box is dropped."""
    import threading
    box = {}
    gate = threading.Lock()       # the verdict: whoever takes it first decides whether the result counts
    state = {"expired": False}

    def run():
        try:
            fn()
            err = None
        except BaseException as e:  # noqa: BLE001 -- handed to the caller, whatever it is
            err = e
        with gate:
            if not state["expired"]:
                box["err"] = err
                box["done"] = True

    t = threading.Thread(target=run, daemon=True, name="lf-bringup")
    t.start()
    t.join(timeout_s)
    with gate:
        if box.get("done"):
            return True, box.get("err")
        state["expired"] = True
    if on_expire is not None:
        on_expire()
    return False, None


def expired(reasons):
    """Did any rank's bring-up step end on its DEADLINE (agree()'s reasons)?  Then a helper thread of that rank was abandoned
    inside its context (call_with_deadline): the context must not be used again -- every rank builds a fresh one."""
    return any(("within the deadline" in r) or ("blocked for more than" in r) or ("did not complete within" in r) for r in reasons)


def first_exchange(dist, enqueue, test, timeout_s=120.0, poll_s=0.01, clock=None, sleep=None, on_expire=None):
    """Run the first exchange of a fresh communicator so that NO rank can hang: `enqueue()` queues it
    (may raise, may block on the host: it runs under the deadline in a helper thread), `test()` says
    whether it has completed on the device (non-blocking).
    A rank whose peers never joined sees `test()` stay False and gives up after timeout_s.  Returns
    (all ranks completed?, reasons); on False every rank must abort its communicator (lf_comm_abort)
    before using its streams again."""
    import time
    clock = clock or time.monotonic
    sleep = sleep or time.sleep
    ok, why = True, ""
    try:
        # enqueue() itself may block on the HOST: the first ncclAllGather of a communicator connects its
        # channels in the calling thread and waits there for every peer.  Same deadline.
        done, err = call_with_deadline(enqueue, timeout_s, on_expire)
        if not done:
            raise TimeoutError(f"enqueueing the first exchange blocked for more than {timeout_s:g} s (a peer never joined)")
        if err is not None:
            raise err
        t0 = clock()
        while not test():
            if clock() - t0 > timeout_s:
                ok, why = False, f"the first exchange did not complete within {timeout_s:g} s"
                break
            sleep(poll_s)
    except Exception as e:  # noqa: BLE001 -- the verdict must reach agree() whatever went wrong
        ok, why = False, f"{type(e).__name__}: {e}"
    return agree(dist, ok, why)
