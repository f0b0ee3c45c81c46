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
"""
TILE_ROWS = 8


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
