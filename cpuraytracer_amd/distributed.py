"""Multi-GPU layer: one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).

The path shards by image rows: rank r renders the rows j with j % world == r of the SAME
image (cyclic, because cost is strongly row dependent; single rows, because cost also grows steadily towards the
bottom of the picture: with 4-row blocks the rank that always got the lowest rows of each band had 2 % more
work than the mean, with single rows 0.5 % -- tools/shard_balance_probe.py), every rank holds
the whole (tiny) scene, and the only exchange step is one gather of the finished strips to rank 0,
which de-interleaves them.  Each pixel depends only on (global pixel id, s, seed), so the assembled
image is byte-identical to the single-GPU image.  No other collective exists on the path.
"""
import numpy as np

from ._capi import BLOCK_ROWS, RtRowset


def shard_rowset(H, rank, world, block_rows=BLOCK_ROWS):
    return RtRowset(0, H, block_rows, rank, world)


def local_rows(H, rank, world, block_rows=BLOCK_ROWS):
    nblocks = (H + block_rows - 1) // block_rows
    rows = 0
    for b in range(rank, nblocks, world):
        rows += min(block_rows, H - b * block_rows)
    return rows


def global_row(lr, rank, world, block_rows=BLOCK_ROWS):
    return ((lr // block_rows) * world + rank) * block_rows + lr % block_rows


def max_local_rows(H, world, block_rows=BLOCK_ROWS):
    return max(local_rows(H, r, world, block_rows) for r in range(world))


def assemble(parts, H, world, block_rows=BLOCK_ROWS):
    """parts[r]: array [>= local_rows(r), W, C] (padding rows beyond local_rows are ignored)."""
    first = np.asarray(parts[0])
    out = np.zeros((H,) + first.shape[1:], dtype=first.dtype)
    for r in range(world):
        p = np.asarray(parts[r])
        for lr in range(local_rows(H, r, world, block_rows)):
            out[global_row(lr, r, world, block_rows)] = p[lr]
    return out


def gather_strip(strip, H, rank, world, dst=0, group=None):
    """Gather per-rank strips (torch tensors [rows_r, W, C] on the rank's device) to `dst`.

    Strips are padded to the largest shard so that one equal-count gather serves ragged heights.
    Returns the list of tensors on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    rows_max = max_local_rows(H, world)
    if strip.shape[0] < rows_max:
        pad = torch.zeros((rows_max - strip.shape[0],) + tuple(strip.shape[1:]), dtype=strip.dtype, device=strip.device)
        strip = torch.cat([strip, pad], 0)
    strip = strip.contiguous()
    if world == 1:
        return [strip]
    out = [torch.empty_like(strip) for _ in range(world)] if rank == dst else None
    dist.gather(strip, out, dst=dst, group=group)
    return out


class StripExchange:
    """One buffer per rank holding its HDR strip (float32) followed by its LDR strip (uint8), both padded to the largest shard,
    and ONE gather per step for the two of them (two collectives of a megabyte each cost twice the launch and rendezvous)."""

    def __init__(self, H, W, rank, world, device):
        import torch
        self.H, self.W, self.rank, self.world = H, W, rank, world
        self.rows = local_rows(H, rank, world)
        self.rows_max = max_local_rows(H, world)
        self.hdr_bytes = self.rows_max * W * 3 * 4
        self.buf = torch.zeros(self.hdr_bytes + self.rows_max * W * 3, dtype=torch.uint8, device=device)
        self.hdr = self.buf[:self.hdr_bytes].view(torch.float32).view(self.rows_max, W, 3)
        self.ldr = self.buf[self.hdr_bytes:].view(self.rows_max, W, 3)
        self.out = [torch.empty_like(self.buf) for _ in range(world)] if (rank == 0 and world > 1) else None

    def gather(self, through_host=False):
        """-> (hdr parts, ldr parts) on rank 0, (None, None) elsewhere.  through_host: CPU tensors over gloo (rehearsal)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return [self.hdr], [self.ldr]
        if through_host:
            src = self.buf.cpu()
            out = [torch.empty_like(src) for _ in range(self.world)] if self.rank == 0 else None
        else:
            src, out = self.buf, self.out
        dist.gather(src, out, dst=0)
        if self.rank != 0:
            return None, None
        W, rm, hb = self.W, self.rows_max, self.hdr_bytes
        return ([o[:hb].view(torch.float32).view(rm, W, 3) for o in out], [o[hb:].view(rm, W, 3) for o in out])
