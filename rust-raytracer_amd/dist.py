"""Multi-GPU frame: one process per GPU, image row blocks interleaved over ranks, one
gather of the RGB8 blocks to rank 0 (RCCL over xGMI when the backend is "nccl").

The render path shards by pixels (src/render.rs:163-170 renders pixels independently;
only the AA detector looks at neighbours, and each rank renders a one-row halo for it),
so there is no data-path collective apart from collecting the finished framebuffer.

`render_local` is the only piece that touches a renderer: it must fill a
(rows_capacity, W, 3) uint8 tensor with this rank's row blocks in compact order.  The
product passes the HIP path; the CPU tests pass a stand-in to exercise the split,
gather and re-assembly logic under gloo."""
import torch
import torch.distributed as dist


def owned_rows_capacity(height, nranks, tile_rows):
    if nranks <= 1:
        return height
    blocks = (height + tile_rows - 1) // tile_rows
    return ((blocks + nranks - 1) // nranks) * tile_rows


def row_map(height, nranks, tile_rows, device="cpu"):
    """For every image row: (owning rank, row inside that rank's compact buffer)."""
    y = torch.arange(height, device=device)
    block = y // tile_rows
    if nranks <= 1:
        return torch.zeros_like(y), y
    return block % nranks, (block // nranks) * tile_rows + y % tile_rows


def assemble(gathered, height, nranks, tile_rows):
    """gathered: (nranks, rows_capacity, W, 3) → (height, W, 3)."""
    owner, local = row_map(height, nranks, tile_rows, gathered.device)
    return gathered[owner, local]


def render_frame(render_local, width, height, tile_rows, device, group=None):
    """Render this rank's share and gather the frame on rank 0.

    Returns the assembled (H, W, 3) uint8 tensor on rank 0 and None elsewhere."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    cap = owned_rows_capacity(height, world, tile_rows)
    local = torch.zeros((cap, width, 3), dtype=torch.uint8, device=device)
    render_local(local, rank, world, tile_rows)
    if world == 1:
        return local[:height]
    if rank == 0:
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, parts, dst=0, group=group)
        return assemble(torch.stack(parts), height, world, tile_rows)
    dist.gather(local, None, dst=0, group=group)
    return None


def hip_render_local(device_scene, dim, cam, cfg, check=True):
    """render_local callback backed by rt_render_device on torch's current stream.

    With `check` the callback waits for the frame and renders it again if rt_frame_status reports lost
    rays; a caller that pipelines frames passes check=False and calls `device_scene.frame_dropped()`
    itself before it uses them."""
    def fn(local, rank, world, tile_rows):
        c = cfg.copy()
        c.rank, c.nranks, c.tile_rows = rank, world, tile_rows
        stream = torch.cuda.current_stream(local.device).cuda_stream
        for _ in range(8):
            device_scene.render_device(dim, cam, c, local.data_ptr(), stream)
            if check and device_scene.frame_dropped():
                continue          # a ray generation outgrew the queues: the library has enlarged them, render again
            return
        raise RuntimeError("the frame keeps overflowing the ray queues")
    return fn
