"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups run the product's shard -> gather ->
scatter logic (raytracing-rust_amd/distributed.py); the pixels of each shard come from the oracle
standing in for the GPU kernel.  Rank 0's gathered frame must equal the unsharded render."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, width, height, tile, result_path):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import importlib
    import oracle as O
    import scenes
    D = importlib.import_module("raytracing-rust_amd.distributed")
    abi = scenes.abi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ls = scenes.load_ssml("overshadowed")
        s = O.Scene(ls.scene)
        cam = O.camera_new(**ls.camera_params)
        opts = abi.default_render_opts(width, height, 2, seed=9)
        opts.tile_width = opts.tile_height = tile
        so = D.shard_opts(opts, rank, world)
        # the oracle renders this rank's tiles in FRAME layout; pack them the way RT_LAYOUT_SHARD does
        fo = D.copy_opts(so); fo.output_layout = abi.RT_LAYOUT_FRAME
        frame, rays = s.render(cam, fo, n_threads=2)
        order = D.shard_order_numpy(so)
        g = D.ShardGather(opts, rank, world, torch.device("cpu"))
        shard = g.new_shard_buffer()
        valid = order >= 0
        shard[: len(order)][torch.from_numpy(valid)] = torch.from_numpy(frame.reshape(-1, 3)[order[valid]])
        out = g.gather(shard)
        rays_t = D.reduce_rays(torch.tensor([rays], dtype=torch.int64), world)
        if rank == 0:
            full, rays_full = s.render(cam, opts, n_threads=2)
            np.save(result_path, np.array([np.array_equal(out.numpy(), full), int(rays_t.item()) == rays_full]))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,width,height,tile", [(2, 50, 30, 8), (3, 33, 17, 4), (2, 16, 8, 8), (8, 70, 44, 8)])  # 8 = the node bench.py --gpus 8 runs on
def test_shard_gather_reassembles_the_frame(tmp_path, world, width, height, tile):
    result = str(tmp_path / "ok.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, width, height, tile, result), nprocs=world, join=True)
    ok = np.load(result)
    assert ok.all(), ok


def test_shard_order_numpy_matches_the_library():
    sys.path.insert(0, ROOT)
    import importlib
    hb = importlib.import_module("raytracing-rust_amd.hip_backend")
    D = importlib.import_module("raytracing-rust_amd.distributed")
    abi = D.abi
    for (w, h, tw, world) in [(50, 30, 0, 3), (1920, 1080, 0, 8), (33, 17, 4, 2), (64, 64, 16, 5)]:
        for rank in range(world):
            o = abi.default_render_opts(w, h, 1)
            o.tile_width = o.tile_height = tw
            so = D.shard_opts(o, rank, world)
            lib_order = hb.shard_pixel_order(so).astype(np.int64)  # UINT64_MAX -> -1
            assert np.array_equal(lib_order, D.shard_order_numpy(so))
            assert len(lib_order) <= D.max_shard_entries(o, world)
