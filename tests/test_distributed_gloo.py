"""CPU, world_size 2, gloo: the sharding host logic (contiguous world blocks,
world-dimension all-gather, the episode-number exchange between the two phases
of a step) with a fake simulator that follows the two-phase protocol of
include/mrl_envs.h.  The same code runs over RCCL on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from madrona_rl_envs_playground_amd.distributed import ShardedSimulator, gather_worlds, shard_range


def test_shard_ranges_partition_the_batch():
    for total in (8, 10, 262144, 7):
        for ws in (1, 2, 3, 8):
            blocks = [shard_range(total, r, ws) for r in range(ws)]
            assert blocks[0][0] == 0 and sum(n for _, n in blocks) == total
            for (lo, n), (lo2, _) in zip(blocks, blocks[1:]):
                assert lo + n == lo2
    assert shard_range(262144, 3, 8) == (3 * 32768, 32768)


class _T:
    def __init__(self, t):
        self.t = t

    def to_torch(self):
        return self.t


class FakeEpisodeSim:
    """World w finishes at step t iff hash(global id, t) % 5 == 0; finished worlds
    get episode numbers in ascending world order.  Follows the two-phase protocol of
    include/mrl_envs.h: phase 1 leaves the shard's finished count in SHARD_COUNT,
    phase 2 (plain or gathered) hands out the episode numbers."""

    def __init__(self, n):
        self.n, self.lo, self.t = n, 0, 0
        self.done = torch.zeros(n, dtype=torch.int32)
        self.episode = torch.arange(n, dtype=torch.int64)
        self.counter = n
        self.shard_count = torch.zeros(1, dtype=torch.int32)

    def reseed_shard(self, lo, total):
        self.lo, self.counter = lo, total
        self.episode = torch.arange(lo, lo + self.n, dtype=torch.int64)

    def done_tensor(self):
        return _T(self.done)

    def shard_count_tensor(self):
        return _T(self.shard_count)

    def step_phase1(self, actions=None):
        gid = torch.arange(self.lo, self.lo + self.n)
        self.done = (((gid * 2654435761 + self.t * 40503) >> 3) % 5 == 0).to(torch.int32)
        self.shard_count.fill_(int(self.done.sum()))  # in place: the exported word never moves
        self.t += 1

    def step_phase2(self, base=None):
        b = self.counter if base is None else int(base.item())
        idx = torch.nonzero(self.done).flatten()
        self.episode[idx] = b + torch.arange(len(idx))
        self.counter = b + len(idx)

    def step_phase2_gathered(self, counts, rank):
        assert counts.dtype == torch.int32 and int(counts[rank]) == int(self.shard_count)
        b = self.counter + int(counts[:rank].sum())
        idx = torch.nonzero(self.done).flatten()
        self.episode[idx] = b + torch.arange(len(idx))
        self.counter += int(counts.sum())

    def step(self):
        self.step_phase1()
        self.step_phase2()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, ws, port, total, steps, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        sh = ShardedSimulator(lambda n: FakeEpisodeSim(n), total)
        history = []
        for _ in range(steps):
            sh.step()
            full = sh.gather(sh.sim.episode, world_dim=0)
            history.append(full.clone())
        # a (P, N, F)-style tensor gathered along dim 1, equal and ragged shards
        lo, n = sh.lo, sh.n
        local = torch.arange(lo, lo + n).view(1, n, 1).expand(2, n, 3).contiguous()
        got = sh.gather(local, world_dim=1)
        assert got.shape == (2, total, 3) and torch.equal(got[0, :, 0], torch.arange(total))
        if total % ws == 0:
            out = torch.empty(total, 4, dtype=torch.int64)
            gather_worlds(torch.full((n, 4), rank), 0, out=out)
            assert torch.equal(out[:, 0], torch.arange(ws).repeat_interleave(n))
        torch.save(torch.stack(history), os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [64, 37])
def test_sharded_episode_numbers_equal_single_process(total, tmp_path):
    steps, ws = 25, 2
    mp.start_processes(_worker, args=(ws, _free_port(), total, steps, str(tmp_path)), nprocs=ws, join=True,
                       start_method="spawn")
    single = FakeEpisodeSim(total)
    expect = []
    for _ in range(steps):
        single.step()
        expect.append(single.episode.clone())
    expect = torch.stack(expect)
    for rank in range(ws):
        got = torch.load(os.path.join(str(tmp_path), f"r{rank}.pt"))
        assert torch.equal(got, expect), f"rank {rank} disagrees with the single-process numbering"
    assert int(expect.max()) > total  # some worlds did finish
