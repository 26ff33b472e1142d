"""Multi-GPU partitioning of the VMAF hot path (one process per GPU, torch.distributed over RCCL).

The reference has no multi-device code; its only parallelism is one process per ref/dis pair
(python/src/vmaf/tools/misc.py:246-276).  What the path allows:

* single video: contiguous frame ranges per rank.  VIF/ADM are per frame; motion needs the previous
  frame's blurred ref (wrapper/src/combo.c:309-326), so a shard first blurs frame start-1 (one-frame
  halo, ref only); motion2[i] = min(motion[i], motion[i+1]) (wrapper/src/vmaf.cpp:375-379) crosses the
  right edge of a shard, so it is computed after the gather, never per shard.
* batch mode: whole pairs, longest first; no exchange step, results concatenated in input order.

The only collective is one gather of the per-frame atom rows (20 doubles per frame) to rank 0.
"""
import numpy as np

N_ATOMS = 20


def frame_range(total_frames, world, rank):
    """Contiguous shard [start, end) of rank; sizes differ by at most one frame."""
    base, extra = divmod(total_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def assign_pairs(costs, world):
    """Batch mode: greedy longest-first assignment of pairs (cost = width*height*frames) to ranks.
    -> list of per-rank index lists; every index appears exactly once."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += costs[i]
    return [sorted(x) for x in out]


def extract_shard(extractor, ref, dis, start, end):
    """Atoms of frames [start, end) of one video.  `extractor` has motion_seed(ref_frame) and
    extract(ref, dis, continue_sequence) - vmaf_amd.Engine does.  ref/dis are indexable by frame."""
    if end <= start:
        return np.zeros((0, N_ATOMS))
    if start > 0:
        extractor.motion_seed(ref[start - 1])              # halo: exact motion for the shard's first frame
    return extractor.extract(ref[start:end], dis[start:end], continue_sequence=start > 0)


def gather_rows(local_rows, dist=None, device=None):
    """Gather per-frame rows [n_local, k] from every rank to rank 0, in rank order (= frame order for
    frame-range shards).  Returns the concatenated array on rank 0, None elsewhere.  `dist` is
    torch.distributed (initialised) or None for a single process."""
    local_rows = np.ascontiguousarray(local_rows, np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_rows
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    k = local_rows.shape[1]
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    padded = torch.zeros((nmax, k), dtype=torch.float64, device=dev)
    if local_rows.shape[0]:
        padded[:local_rows.shape[0]] = torch.from_numpy(local_rows).to(dev)
    bufs = [torch.zeros_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, bufs, dst=0)
    if rank != 0:
        return None
    return np.concatenate([bufs[r][:counts[r]].cpu().numpy() for r in range(world)], axis=0)
