"""Shard planning and result gathering for multi-GPU runs (one process per GPU).

The hot path shards by receipt / key / block index with no data-path exchange:
every rank verifies the claims of its own contiguous index range against its own witness shard,
and ONE all-gather of the per-shard verdict bytes / bitmaps produces the global result.  The
collective is `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests); this module is plumbing and contains no compute.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_items: int, world: int):
    """Contiguous, near-equal index ranges [(lo, hi)] * world; the first n_items % world get one extra."""
    base, extra = divmod(n_items, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def gather_bytes(local: np.ndarray, n_items: int, dist, device="cpu"):
    """All-gather per-shard byte arrays laid out by shard_bounds(n_items, world) into the global array.
    Shards differ by at most one element, so every rank sends max-shard bytes (one collective)."""
    import torch

    world = dist.get_world_size()
    bounds = shard_bounds(n_items, world)
    width = max(hi - lo for lo, hi in bounds)
    buf = torch.zeros(width, dtype=torch.uint8, device=device)
    buf[: len(local)] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.uint8)).to(device)
    out = torch.empty(world * width, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy().reshape(world, width)
    return np.concatenate([out[r, : hi - lo] for r, (lo, hi) in enumerate(bounds)])


def pack_bits(flags: np.ndarray) -> np.ndarray:
    """bool/0-1 bytes → little-endian bitmap bytes (bit i of byte i//8), the form the engine's
    CID bitmap has."""
    return np.packbits(np.asarray(flags, dtype=np.uint8), bitorder="little")


class PaddedGather:
    """The per-step collective of bench.py: every rank contributes one message of ITS OWN length (shards are
    generated independently, so block counts differ slightly); messages are padded to the longest one,
    agreed once at setup, and all-gathered into a [world, width] tensor."""

    def __init__(self, local_len: int, dist, device="cpu"):
        import torch

        self.dist = dist
        self.world = dist.get_world_size()
        t = torch.tensor([int(local_len)], dtype=torch.int64, device=device)
        lens = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(self.world)]
        dist.all_gather(lens, t)
        self.lens = [int(x.item()) for x in lens]
        self.width = max(self.lens)
        self.payload = torch.zeros(self.width, dtype=torch.uint8, device=device)
        self.gathered = torch.empty(self.world * self.width, dtype=torch.uint8, device=device)

    def run(self):
        """One all-gather of `payload` (the caller fills payload[:local_len] beforehand)."""
        self.dist.all_gather_into_tensor(self.gathered, self.payload)

    def message(self, rank: int):
        """Rank `rank`'s unpadded message out of the last gather."""
        lo = rank * self.width
        return self.gathered[lo: lo + self.lens[rank]]
