"""Image tiles sharded across the GPUs of one node: replicated scene, disjoint tiles, ONE
end-of-render gather of the framebuffer shards to rank 0 (RCCL over xGMI via torch.distributed;
backend "nccl" is RCCL on ROCm).  No per-bounce or per-pass communication exists on this path:
the reference itself partitions by pixel chunk with no cross-chunk dependency
(samplers/random_sampler.rs:45-52), and the random stream is keyed by (seed, pixel, sample), so a
pixel's value does not depend on which GPU renders it.

Tile t (8x8 pixels by default, row-major tile order) belongs to shard t % world: the interleave
balances cheap sky tiles against geometry tiles.  Each rank renders its tiles into a packed
RT_LAYOUT_SHARD buffer; rank 0 receives the shards and scatters them into the frame with the index
table rt_shard_pixel_order() defines.  Per-GPU payload is W*H*12/world bytes (3.1 MB at 1080p on 8
GPUs); shards arrive on distinct xGMI links.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import abi


def copy_opts(opts):
    o = abi.RenderOpts()
    C.memmove(C.byref(o), C.byref(opts), C.sizeof(o))
    return o


def shard_opts(opts, rank, world):
    o = copy_opts(opts)
    o.shard_index, o.shard_count = rank, world
    o.output_layout = abi.RT_LAYOUT_SHARD
    return o


def shard_order_numpy(opts):
    """Pixel index of every packed shard entry (numpy restatement of rt_shard_pixel_order, used where
    librt_hip.so is not wanted, e.g. by CPU-only tests of the gather logic).  -1 marks padding."""
    tw = opts.tile_width or 8
    th = opts.tile_height or 8
    tiles_x = (opts.width + tw - 1) // tw
    tiles_y = (opts.height + th - 1) // th
    tiles = np.arange(opts.shard_index, tiles_x * tiles_y, opts.shard_count, dtype=np.int64)
    ty, tx = tiles // tiles_x, tiles % tiles_x
    iy, ix = np.divmod(np.arange(tw * th, dtype=np.int64), tw)
    x = tx[:, None] * tw + ix[None, :]
    y = ty[:, None] * th + iy[None, :]
    pix = np.where((x < opts.width) & (y < opts.height), y * opts.width + x, -1)
    return pix.reshape(-1)


def max_shard_entries(opts, world):
    tw = opts.tile_width or 8
    th = opts.tile_height or 8
    tiles = ((opts.width + tw - 1) // tw) * ((opts.height + th - 1) // th)
    return ((tiles + world - 1) // world) * tw * th


class ShardGather:
    """Pre-computes the scatter tables once; `gather()` is the per-frame collective + scatter."""

    def __init__(self, opts, rank, world, device):
        self.rank, self.world, self.device = rank, world, device
        self.width, self.height = int(opts.width), int(opts.height)
        self.entries = max_shard_entries(opts, world)  # equal-sized buffers for the collective
        self.local_entries = len(shard_order_numpy(shard_opts(opts, rank, world)))
        self.recv = None
        self.index = None
        if rank == 0:
            idx = []
            for k in range(world):
                order = shard_order_numpy(shard_opts(opts, k, world))
                pad = np.full(self.entries - len(order), -1, dtype=np.int64)
                idx.append(np.concatenate([order, pad]))
            idx = np.concatenate(idx)
            self.valid = torch.from_numpy(np.nonzero(idx >= 0)[0]).to(device)
            self.index = torch.from_numpy(idx[idx >= 0]).to(device)
            self.recv = [torch.empty(self.entries, 3, dtype=torch.float32, device=device) for _ in range(world)]

    def new_shard_buffer(self):
        return torch.zeros(self.entries, 3, dtype=torch.float32, device=self.device)

    def gather(self, shard, frame=None):
        """shard: [entries, 3] on every rank.  Returns the [H, W, 3] frame on rank 0, None elsewhere."""
        if self.world == 1 and not dist.is_initialized():
            parts = shard
        else:
            dist.gather(shard, self.recv if self.rank == 0 else None, dst=0)
            if self.rank != 0:
                return None
            parts = torch.cat(self.recv, dim=0)
        if frame is None:
            frame = torch.empty(self.height * self.width, 3, dtype=torch.float32, device=self.device)
        frame.view(-1, 3)[self.index] = parts[self.valid]
        return frame.view(self.height, self.width, 3)


def reduce_rays(rays_tensor, world):
    """SamplerProgress.rays_shot summed over shards (8 bytes)."""
    if world > 1 or dist.is_initialized():
        dist.reduce(rays_tensor, dst=0, op=dist.ReduceOp.SUM)
    return rays_tensor
