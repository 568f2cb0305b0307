"""Multi-GPU sharding of ray batches: the BVH is replicated on every GPU,
the ray array is cut into contiguous shards (tile / bounce order is kept inside a shard, so
coherence survives), one process per GPU, and there is NO collective on the data path.
torch.distributed is used only for the barrier and the max-over-ranks of the elapsed time
(and, optionally, a final gather of hit records to rank 0)."""
from __future__ import annotations

import numpy as np

WAVE = 64


def shard_range(n_rays: int, rank: int, world: int, align: int = WAVE):
    """Contiguous [begin, end) of rank's shard; boundaries are multiples of `align` rays so a
    wave never straddles two shards; the shards cover [0, n_rays) exactly once."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    units = -(-n_rays // align)
    base, extra = divmod(units, world)
    b = rank * base + min(rank, extra)
    e = b + base + (1 if rank < extra else 0)
    return min(b * align, n_rays), min(e * align, n_rays)


def build_once_load_everywhere(verts, rank: int, world: int, dist, path: str, **build_kw):
    """The replicated BVH8_CWBVH of an N-process run without N concurrent host builds on one node's cores: rank 0 builds (all cores) and
    writes the blobs as a BVH8_CWBVH::Save-compatible file (tbvh_cwbvh_file_write), everyone meets at a barrier, ranks 1..N-1 read the file
    (tbvh_cwbvh_file_read).  Returns a HostBVH on every rank — rank 0's with the BVH2 it was encoded from (what the oracle needs), the others'
    with the two blobs only — and the seconds this rank spent.  `path` must be visible to all ranks (one node: /tmp)."""
    import os
    import time
    import tinybvh_amd as tb
    t0 = time.time()
    host = None
    if rank == 0:
        host = tb.HostBVH(verts, tb.LAYOUT_CWBVH, **build_kw)
        tmp = f"{path}.{os.getpid()}.tmp"
        host.save_cwbvh(tmp)
        os.replace(tmp, path)             # the file appears complete or not at all
    if dist is not None and world > 1:
        dist.barrier()
    if rank != 0:
        host = tb.HostBVH.from_cwbvh_file(path, verts.shape[0] // 3)
    if dist is not None and world > 1:
        dist.barrier()                    # everyone has read it: rank 0 may delete it
    return host, time.time() - t0


def max_over_ranks(seconds: float, dist=None, device=None) -> float:
    """MAX of a per-rank elapsed time (the bench contract's timing rule)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_hits(local_hits: np.ndarray, n_rays: int, dist, device=None):
    """Optional final gather of the 16-byte hit records (t,u,v,prim as 4 x u32) of every
    rank's shard to rank 0, outside any timed region.  Returns the full (n_rays, 4) array on
    rank 0, None elsewhere."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_rays, r, world)[1] - shard_range(n_rays, r, world)[0] for r in range(world)]
    mx = max(sizes)
    buf = torch.zeros((mx, 4), dtype=torch.int32, device=device or "cpu")
    buf[: local_hits.shape[0]] = torch.from_numpy(local_hits.view(np.int32).reshape(-1, 4)).to(buf.device)
    out = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    return np.concatenate([o[:s].cpu().numpy().view(np.uint32) for o, s in zip(out, sizes)])
