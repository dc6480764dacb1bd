"""The multi-GPU frame logic (row-block interleave, compact per-rank buffers, gather to
rank 0, re-assembly) under world_size 2 with the gloo backend on CPU.  The per-rank
renderer is a stand-in backed by the CPU oracle; the product passes the HIP path through
the same `render_frame`."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIDTH, HEIGHT, TILE_ROWS = 40, 37, 4   # ragged: 37 rows do not divide into 2 x 4-row blocks


def _oracle_local(scene, dim, cam, cfg):
    from oracle import binding as oracle

    def fn(local, rank, world, tile_rows):
        c = cfg.copy()
        c.rank, c.nranks, c.tile_rows = rank, world, tile_rows
        full = oracle.render(scene, dim, cam, c, want_aux=False)["rgb"]
        for y in range(dim.height):
            if (y // tile_rows) % world == rank:
                local[((y // tile_rows) // world) * tile_rows + y % tile_rows] = torch.from_numpy(full[y])
    return fn


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scenes = importlib.import_module("harness.scenes")
    rdist = importlib.import_module("rust-raytracer_amd.dist")
    s = scenes.Scene("glass_monkey")
    dim, cam, cfg = s.view(width=WIDTH, height=HEIGHT, anti_alias=True, seed=1)
    frame = rdist.render_frame(_oracle_local(s, dim, cam, cfg), WIDTH, HEIGHT, TILE_ROWS, torch.device("cpu"))
    if rank == 0:
        np.save(out_path, frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_frame_equals_single_rank(tmp_path, scene_cache, oracle):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    s = scene_cache("glass_monkey")
    dim, cam, cfg = s.view(width=WIDTH, height=HEIGHT, anti_alias=True, seed=1)
    want = oracle.render(s, dim, cam, cfg, want_aux=False)["rgb"]
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_row_map_and_capacity():
    rdist = importlib.import_module("rust-raytracer_amd.dist")
    render = importlib.import_module("rust-raytracer_amd.render")
    lib = render.hip_lib()
    for h, n, t in [(1080, 1, 16), (1080, 2, 16), (1080, 8, 16), (37, 2, 4), (5, 8, 16), (2160, 8, 32)]:
        cap = rdist.owned_rows_capacity(h, n, t)
        assert cap == lib.rt_owned_rows_capacity(h, n, t)
        owner, local = rdist.row_map(h, n, t)
        assert int(local.max()) < cap
        # (owner, local) is injective: no two rows share a slot
        assert len({(int(o), int(l)) for o, l in zip(owner, local)}) == h
        if n > 1:
            counts = torch.bincount(owner, minlength=n)
            assert int(counts.max()) <= cap
