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


def plan_tiles(rank, world, n_variants, row_bytes, tile_bytes, strong=True):
    """How rank `rank` of `world` walks its shard in tiles of at most `tile_bytes` (bench.py).

    strong: `n_variants` is the whole cohort and the rank's shard is variant_range(rank, world, n_variants);
    otherwise every rank has a shard of `n_variants` of its own.  Every rank walks the SAME number of tiles (the
    gathers of tile t are collective), sized for the largest shard; a shorter shard's last tiles may be empty.
    Returns dict(v_lo, n, n_tiles, per_tile, tiles=[(lo, hi) within the shard]) and, for every tile, the number of
    variants EVERY rank holds in it (block sizes are known to all ranks without an exchange)."""
    def shard(r):
        return variant_range(r, world, n_variants) if strong else (r * n_variants, (r + 1) * n_variants)
    v_lo, v_hi = shard(rank)
    n = v_hi - v_lo
    n_max = max(shard(r)[1] - shard(r)[0] for r in range(world))
    cap = max(1, int(tile_bytes) // max(1, int(row_bytes)))            # variants per tile buffer
    n_tiles = max(1, -(-n_max // cap))
    per_tile = max(1, -(-n_max // n_tiles))
    tiles = [(min(t * per_tile, n), min((t + 1) * per_tile, n)) for t in range(n_tiles)]
    counts = [[min((t + 1) * per_tile, shard(r)[1] - shard(r)[0]) - min(t * per_tile, shard(r)[1] - shard(r)[0])
               for r in range(world)] for t in range(n_tiles)]
    return dict(v_lo=v_lo, n=n, n_max=n_max, n_tiles=n_tiles, per_tile=per_tile, tiles=tiles, counts=counts)


def result_block_layout(n):
    """Byte offsets of the SoA pieces inside one shard's result block of n variants:
    counts int32[n][4] | odds f64[n] | chisq f64[n] | p f64[n]."""
    return {"counts": 0, "odds": 16 * n, "chisq": 24 * n, "p": 32 * n, "bytes": 40 * n}


def gather_blocks(block, sizes, dst=0, group=None, async_op=False, out_bufs=None, force=False):
    """Gathers one uint8 result block per rank on `dst`.

    block : 1-D uint8 tensor of this rank (length sizes[rank])
    sizes : list of block lengths of all ranks (known to every rank: it follows
            from variant_range), so ragged shards need no size exchange.
    out_bufs : optional preallocated receive buffers on dst (world tensors of
            max(sizes) bytes), so a steady-state loop allocates nothing.
    force : with one rank, still go through the collective (one real RCCL rank through the N > 1 path: a check).
    Returns (list_of_tensors_or_None, work): on dst the list holds every rank's
    block trimmed to its size, elsewhere None.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert block.dtype == torch.uint8 and block.dim() == 1 and block.numel() == sizes[rank]
    cap = max(sizes)
    if world == 1 and not force:
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


def reduce_sample_counters(counters, dst=0, group=None, async_op=False):
    """Per-sample counters of get_sample_stats (missing genotypes, Mendelian errors per sample: call site
    stats_runner.c:197-198) are sums over variants, so with variants sharded across ranks they are the one
    quantity that needs a reduction: an int32 sum of n_samples values onto `dst`
    (ncclReduce over xGMI on the GPU box, gloo in the CPU tests).  In place; returns the work handle."""
    assert counters.dtype == torch.int32 and counters.dim() == 1
    if dist.get_world_size(group) == 1:
        return None
    return dist.reduce(counters, dst=dst, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def pair_row_range(rank, world, n_variants):
    """Row band [lo, hi) of the epistasis pair scan for rank `rank`: the triangle of V(V-1)/2 pairs is cut into bands
    of whole 64-row blocks with (nearly) equal numbers of pairs, so every GPU scans the same amount of work; each
    rank ranks its band (hpgv_epi_rank_pairs over its rows) and the per-fold top lists are merged on rank 0."""
    blocks = (n_variants + 63) // 64
    total = n_variants * (n_variants - 1) // 2

    def pairs_before(row):
        row = min(row, n_variants)
        return row * (2 * n_variants - row - 1) // 2

    def cut(k):
        if k <= 0:
            return 0
        if k >= world:
            return n_variants
        target = total * k // world
        lo, hi = 0, blocks
        while lo < hi:                               # first block boundary with at least `target` pairs before it
            mid = (lo + hi) // 2
            if pairs_before(mid * 64) >= target:
                hi = mid
            else:
                lo = mid + 1
        return min(lo * 64, n_variants)

    return cut(rank), cut(rank + 1)
