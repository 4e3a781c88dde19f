"""Variant sharding across the GPUs of one node.

Every variant is independent (assoc.c:38-82, tdt.c:41-271 carry no state from
one variant to the next), so rank g scans the contiguous range
[g*V/G, (g+1)*V/G) with no data-path collective.  The only exchange is the
final gather of the per-variant result block to rank 0, done with
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  torch is plumbing here: device memory and the process group.
"""
import torch
import torch.distributed as dist

RESULT_BYTES_CHISQ = 40      # {A1,A2,U1,U2} int32 + odds, chisq, p f64


def variant_range(rank, world, n_variants):
    """Contiguous shard of rank `rank`: [lo, hi)."""
    lo = (n_variants * rank) // world
    hi = (n_variants * (rank + 1)) // world
    return lo, hi


def result_block_layout(n):
    """Byte offsets of the SoA pieces inside one shard's result block of n variants:
    counts int32[n][4] | odds f64[n] | chisq f64[n] | p f64[n]."""
    return {"counts": 0, "odds": 16 * n, "chisq": 24 * n, "p": 32 * n, "bytes": 40 * n}


def gather_blocks(block, sizes, dst=0, group=None, async_op=False, out_bufs=None):
    """Gathers one uint8 result block per rank on `dst`.

    block : 1-D uint8 tensor of this rank (length sizes[rank])
    sizes : list of block lengths of all ranks (known to every rank: it follows
            from variant_range), so ragged shards need no size exchange.
    out_bufs : optional preallocated receive buffers on dst (world tensors of
            max(sizes) bytes), so a steady-state loop allocates nothing.
    Returns (list_of_tensors_or_None, work): on dst the list holds every rank's
    block trimmed to its size, elsewhere None.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert block.dtype == torch.uint8 and block.dim() == 1 and block.numel() == sizes[rank]
    cap = max(sizes)
    if world == 1:
        return [block], None
    send = block
    if block.numel() != cap:                      # pad ragged shards to a common size
        send = torch.zeros(cap, dtype=torch.uint8, device=block.device)
        send[: block.numel()] = block
    bufs = None
    if rank == dst:
        bufs = out_bufs if out_bufs is not None else \
            [torch.empty(cap, dtype=torch.uint8, device=block.device) for _ in range(world)]
        assert len(bufs) == world and all(b.numel() == cap for b in bufs)
    work = dist.gather(send, gather_list=bufs, dst=dst, group=group, async_op=async_op)
    if rank != dst:
        return None, work
    return [b[: sizes[r]] for r, b in enumerate(bufs)], work
