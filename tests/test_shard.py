"""CPU: the multi-rank path — sample sharding and the sum-reduce — under gloo with world_size 2.

Ranks render their share with the CPU oracle (the checker) in place of the GPU; what is under test is the
host logic in raytracer_amd/shard.py that bench.py and a multi-GPU driver run unchanged over RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import HERE, ROOT


def test_sample_shards_partition_the_range(rt):
    for spp in (0, 1, 7, 100, 1000):
        for world in (1, 2, 3, 8):
            parts = [rt.sample_shard(spp, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == spp
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        rt.sample_shard(10, 2, 2)


def _worker(rank, world, port, out_path):
    sys.path[:0] = [ROOT, HERE]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import oracle_binding as ob
    from raytracer_amd.shard import render_step

    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = ob.Scene(ob.gen_scene(1))
    cam = ob.camera_build([0, 0, 0], [13, 1.5, 3], 1.0, 0.1, 24, 16)
    spp = 5

    def render_range(b, e):
        if e == b:
            return torch.zeros(16, 24, 3, dtype=torch.float64)
        s, _, _ = scene.render(cam, spp=e - b, depth=10, seed=7, sample_begin=b, threads=1)
        return torch.from_numpy(s)

    total, image = render_step(render_range, lambda t: dist.reduce(t, dst=0, op=dist.ReduceOp.SUM),
                               lambda t, n: ob.finalize(t.numpy(), n), spp, rank, world)
    dist.barrier()
    assert (image is None) == (rank != 0)  # only rank 0 finalises
    if rank == 0:
        np.save(out_path, total.numpy())
        np.save(out_path + ".rgb8.npy", image)
    dist.destroy_process_group()


def test_two_rank_gloo_reduce_equals_single_rank(ob, tmp_path):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    scene = ob.Scene(ob.gen_scene(1))
    cam = ob.camera_build([0, 0, 0], [13, 1.5, 3], 1.0, 0.1, 24, 16)
    want, _, _ = scene.render(cam, spp=5, depth=10, seed=7, threads=1)
    assert np.allclose(got, want, rtol=1e-12, atol=0)
    assert np.array_equal(ob.finalize(got, 5), ob.finalize(want, 5))
    assert np.array_equal(np.load(out + ".rgb8.npy"), ob.finalize(want, 5))  # what rank 0's finalise step produced
