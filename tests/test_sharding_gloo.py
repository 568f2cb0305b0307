"""N > 1 path on CPU: two processes over gloo shard one ray batch (BVH replicated, no
data-path collective), trace their shards with the oracle standing in for the device, and
rank 0 checks the gathered records against the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest

from tinybvh_amd.sharding import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_exactly_once():
    for n in (0, 1, 63, 64, 65, 1000, 16_777_216, 67_108_864 + 17):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            for a, b in zip(r, r[1:]):
                assert a[1] == b[0]
            for b, e in r:
                assert b % 64 == 0 or b == n
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) < 128 or n < 64 * world
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import tinybvh_amd as tb
    from tinybvh_amd import rays as R, scenes
    from tinybvh_amd.sharding import gather_hits, max_over_ranks, shard_range
    from oracle_lib import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        verts = scenes.soup(4096, seed=7)                     # "replicated BVH": every rank builds the same one
        h = tb.HostBVH(verts, tb.LAYOUT_BVH2_WALD)
        rays = R.random_rays(10_007, (0, 0, 0), (10, 10, 10), seed=3)
        b, e = shard_range(rays.shape[0], rank, world)
        orc = Oracle()
        dist.barrier()
        mine = orc.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays[b:e])
        t = max_over_ranks(0.5 + rank, dist)
        assert t == 0.5 + (world - 1)
        hits = np.ascontiguousarray(mine).view(np.uint32).reshape(-1, 16)[:, 12:16].copy()
        full = gather_hits(hits, rays.shape[0], dist)
        if rank == 0:
            ref = orc.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays)
            want = np.ascontiguousarray(ref).view(np.uint32).reshape(-1, 16)[:, 12:16]
            q.put(bool(np.array_equal(full, want)))
        dist.barrier()
        # the replicated BVH of a multi-process run: rank 0 builds once and writes the blob file, the others load it (bench.py --gpus N)
        from tinybvh_amd.sharding import build_once_load_everywhere
        path = os.path.join("/tmp", f"tbvh_gloo_{port}.cwbvh")
        host, _ = build_once_load_everywhere(verts, rank, world, dist, path)
        sums = [int(host.blob(k, np.uint32, 4).astype(np.uint64).sum()) for k in (0, 1)] + [int(host.blob(0, np.uint32, 4).shape[0]), int(host.n_tris)]
        import torch
        t = torch.tensor(sums, dtype=torch.int64)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        same = all(bool(torch.equal(g, got[0])) for g in got)
        has_bvh2 = host.blob(2, np.uint32, 8).shape[0] > 0
        if rank == 0:
            os.remove(path)
            q.put(bool(same and has_bvh2))
        else:
            assert not has_bvh2           # only the builder holds the BVH2 (the oracle runs on rank 0)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_over_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True      # sharded trace == single-process trace
    assert q.get(timeout=5) is True      # every rank holds the same blobs, built once
