"""Sample-index sharding of one render over the ranks of a node.

Paths are independent and the Philox counter is (pixel, sample, bounce,
attempt), so rank g of G traces samples [g*spp/G, (g+1)*spp/G) of EVERY pixel
into its own float sum image; one sum-reduce to rank 0 then yields the image a
single rank would have produced from samples [0, spp) (up to f32 summation
order).  No other collective is on the data path.
"""


def sample_shard(spp, rank, world_size):
    """Half-open sample range of `rank`: ranges are disjoint, ordered and cover [0, spp)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("rank %r of %r" % (rank, world_size))
    if spp < 0:
        raise ValueError("spp %r" % (spp,))
    return (rank * spp) // world_size, ((rank + 1) * spp) // world_size


def render_step(render_range, reduce_to_root, finalize, spp, rank, world_size):
    """One step of the multi-rank render: this rank's share, the one sum-reduce, rank 0's finalise.

    render_range(begin, end) -> this rank's sum image (any tensor type; a rank whose range is empty gets
    render_range(b, b) and must return zeros); reduce_to_root(t) sums t over ranks into rank 0's t
    (torch.distributed.reduce over RCCL, or over gloo in the CPU tests); finalize(t, spp) turns rank 0's
    full-sample sum into the image (mean, gamma, quantise) and is called on rank 0 only.  bench.py and the
    world-size-2 tests run this same function.  Returns (local tensor, what finalize returned or None).
    """
    begin, end = sample_shard(spp, rank, world_size)
    local = render_range(begin, end)
    if world_size > 1:
        reduce_to_root(local)
    image = finalize(local, spp) if rank == 0 else None
    return local, image
