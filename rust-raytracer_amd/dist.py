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


class FrameGather:
    """The buffers of the multi-GPU frame loop, allocated once: this rank's compact row-block buffer, on rank 0 the
    landing area of the gather (one slice per rank), the row map and the assembled frame.  A frame then costs the
    render, one gather and one indexed copy; nothing is allocated inside the loop."""

    def __init__(self, width, height, tile_rows, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.height, self.tile_rows = height, tile_rows
        cap = owned_rows_capacity(height, self.world, tile_rows)
        self.local = torch.zeros((cap, width, 3), dtype=torch.uint8, device=device)
        self.parts = self.frame = self.index = None
        if self.world > 1 and self.rank == 0:
            self.landing = torch.zeros((self.world, cap, width, 3), dtype=torch.uint8, device=device)
            self.parts = [self.landing[r] for r in range(self.world)]          # views: the gather writes straight into `landing`
            owner, local = row_map(height, self.world, tile_rows, device)
            self.index = owner * cap + local                                    # row y of the frame = row index[y] of landing.view(world * cap, W, 3)
            self.frame = torch.empty((height, width, 3), dtype=torch.uint8, device=device)

    def render(self, render_local):
        """Render this rank's share and gather the frame on rank 0: the assembled (H, W, 3) uint8 tensor there
        (overwritten by the next call), None elsewhere."""
        render_local(self.local, self.rank, self.world, self.tile_rows)
        if self.world == 1:
            return self.local[:self.height]
        if self.rank == 0:
            dist.gather(self.local, self.parts, dst=0, group=self.group)
            torch.index_select(self.landing.view(-1, *self.landing.shape[2:]), 0, self.index, out=self.frame)
            return self.frame
        dist.gather(self.local, None, dst=0, group=self.group)
        return None


_gathers = {}


def render_frame(render_local, width, height, tile_rows, device, group=None):
    """Render this rank's share and gather the frame on rank 0.

    Returns the assembled (H, W, 3) uint8 tensor on rank 0 (valid until the next call with the same frame shape)
    and None elsewhere.  The buffers behind it are kept per frame shape (FrameGather)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    key = (width, height, tile_rows, str(device), world, id(group))
    g = _gathers.get(key)
    if g is None:
        g = _gathers[key] = FrameGather(width, height, tile_rows, device, group)
    return g.render(render_local)


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
