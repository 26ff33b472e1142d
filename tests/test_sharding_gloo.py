"""CPU tests of the multi-GPU path (vmaf_amd/sharding.py) with world_size 2 and 3 over gloo.

The per-shard extractor here is the CPU oracle wrapped in the Engine interface, so what is proven is the
partitioning logic itself: frame ranges + one-frame motion halo + gather reproduce the unsharded result
bit for bit, including a scene cut placed exactly at a shard edge (BASELINE.json configs[4])."""
import os
import socket

import numpy as np
import pytest

import oracle
from vmaf_amd import sharding, synth

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


class OracleExtractor:
    """Engine-shaped wrapper over the oracle's per-frame loop (wrapper/src/combo.c:147-371)."""

    def __init__(self, w, h):
        self.prev = np.zeros((h, w), np.float32)
        self.have_prev = False

    def motion_seed(self, ref_frame):
        r = np.ascontiguousarray(ref_frame, np.float32) - np.float32(128)
        self.prev = oracle.conv_sep(oracle.binding.motion_filter(), r)
        self.have_prev = True

    def extract(self, ref, dis, continue_sequence=False):
        out = np.zeros((len(ref), 20))
        for i in range(len(ref)):
            first = (i == 0 and not continue_sequence)
            out[i] = oracle.frame(ref[i], dis[i], self.prev, first)
        return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r8, d8 = synth.make_clip(96, 64, 8, seed=77, cuts=(4,))      # cut at frame 4 = the edge of rank 1's shard for world 2
        ref, dis = synth.to_luma_f32(r8), synth.to_luma_f32(d8)
        start, end = sharding.frame_range(len(ref), world, rank)
        local = sharding.extract_shard(OracleExtractor(96, 64), ref, dis, start, end)
        full = sharding.gather_rows(local, dist)
        if rank == 0:
            q.put(full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_unsharded(world):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    r8, d8 = synth.make_clip(96, 64, 8, seed=77, cuts=(4,))
    want = oracle.clip_atoms(synth.to_luma_f32(r8), synth.to_luma_f32(d8))
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    # motion2 is computed after the gather (vmaf.cpp:375-379): identical by construction, checked anyway
    assert np.array_equal(oracle.derived_features(got)["motion2"], oracle.derived_features(want)["motion2"])


def test_frame_ranges_cover_every_frame_once():
    for total in (0, 1, 7, 600, 1201):
        for world in (1, 2, 3, 8):
            spans = [sharding.frame_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pair_assignment_balances_and_covers():
    costs = [1280 * 720 * f for f in (48, 48, 10, 100, 48, 3, 75, 48, 48, 20)]
    parts = sharding.assign_pairs(costs, 4)
    assert sorted(i for p in parts for i in p) == list(range(len(costs)))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) <= sum(costs) / 4 + max(costs)


def test_gather_rows_single_process_is_identity():
    a = np.arange(40.0).reshape(2, 20)
    assert np.array_equal(sharding.gather_rows(a, None), a)
