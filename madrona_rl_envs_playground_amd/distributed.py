"""Sharding a world batch over the GPUs of one node (one process per GPU,
``torch.distributed`` with the ``nccl`` backend = RCCL over xGMI; ``gloo`` for
CPU rehearsal).  The reference is single-device; this
module is new functionality.

Worlds are independent, so each rank owns a contiguous block
``[lo, lo + n)`` of the global batch and steps it with its own simulator; no
collective is needed to *step*.  Two optional exchanges exist:

* ``gather_worlds``: all-gather of a per-rank tensor along its world dimension,
  for a consumer that wants the global observation batch on every rank
  (the Overcooked block is world-major, so the gathered buffer *is* the global
  ``(N, P, H, W, F)`` tensor);
* the episode-number exchange of Hanabi/Cartpole: the reference seeds every new
  episode from one global counter (src/hanabi_env/sim.cpp:449-451,
  src/cartpole_env/sim.cpp:51-53).  To give world ``w`` of a sharded run the same
  episode sequence as in a single-simulator run, ranks all-gather the number of
  worlds that finished in this step (one int32 each) between the two phases of
  the step; everything stays on the device, nothing synchronises the host.  Phase 1 leaves that
  number in the simulator's SHARD_COUNT word; the collective gathers the words, and phase 2
  (``mrl_step_phase2_gathered``) adds up the lower ranks' itself -- a sharded step is two launches
  and one collective, no torch arithmetic in between.

Since round 4 the episode-number exchange can also run WITHOUT a collective (``ShardedSimulator(..., exchange="mailbox")``):
every rank owns a small device mailbox that its peers map through IPC handles (exchanged once, at construction); the count
launch of a step stores the rank's word into every peer's mailbox over xGMI and phase 2 polls its own -- a sharded step is then
``mrl_step_exchanged``: three launches, no host call and no RCCL call in between (the 4-byte all-gather costs 10-15 us per
step, more than the step itself).  The collective stays the default: the mailbox has run on one GPU only (one rank, and two
processes sharing the card), never across xGMI.

A process group of ONE rank still runs every collective (RCCL at world_size 1 is how the ``nccl``
code path is exercised on a one-GPU box, tests/test_gpu_nccl.py); without a process group the
gathers are identities and no collective is issued.
"""
import torch
import torch.distributed as dist


def shard_range(total_worlds, rank, world_size):
    """Contiguous block of worlds owned by ``rank``: (first world, count).
    The first ``total % world_size`` ranks take one extra world."""
    base, extra = divmod(int(total_worlds), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, base + (1 if rank < extra else 0)


def _all_gather_into(buf, piece, group=None):
    """``all_gather_into_tensor``; device tensors under the ``gloo`` backend (one-GPU rehearsals of the
    rank protocol, never a measured configuration) are staged through the host, which gloo can gather."""
    if piece.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(buf.shape, dtype=buf.dtype)
        dist.all_gather_into_tensor(host, piece.cpu(), group=group)
        buf.copy_(host)
    else:
        dist.all_gather_into_tensor(buf, piece, group=group)


def gather_worlds(local, world_dim=0, group=None, out=None, sizes=None):
    """All-gather ``local`` along ``world_dim``; every rank gets the global tensor.

    With equal shards (``sizes`` None) this is one ``all_gather_into_tensor``
    straight into the output when ``world_dim`` is 0 (RCCL: each rank's slab
    crosses its xGMI links once).  ``sizes`` = per-rank world counts for ragged
    shards (``shard_range`` gives them without communication)."""
    if not dist.is_initialized():
        if out is not None:
            out.copy_(local)
            return out
        return local
    ws = dist.get_world_size(group)
    moved = local.movedim(world_dim, 0).contiguous()
    tail = tuple(moved.shape[1:])
    if sizes is None:
        direct = out is not None and world_dim == 0 and out.is_contiguous()
        buf = out if direct else torch.empty((ws * moved.shape[0],) + tail, dtype=moved.dtype, device=moved.device)
        _all_gather_into(buf, moved, group)
        if direct:
            return out
    else:
        # ragged shards: pad every slab to the largest, one collective, cut the padding out
        biggest = max(int(s) for s in sizes)
        padded = torch.zeros((biggest,) + tail, dtype=moved.dtype, device=moved.device)
        padded[:moved.shape[0]] = moved
        slabs = torch.empty((ws * biggest,) + tail, dtype=moved.dtype, device=moved.device)
        _all_gather_into(slabs, padded, group)
        buf = torch.cat([slabs[r * biggest:r * biggest + int(s)] for r, s in enumerate(sizes)], dim=0)
    result = buf.movedim(0, world_dim)
    if out is not None:
        out.copy_(result)
        return out
    return result


class ShardedSimulator:
    """This rank's slice of a ``total_worlds`` batch.

    ``factory(num_worlds)`` builds the rank-local simulator (any of the three
    games).  For Hanabi/Cartpole the shard is re-seeded so that local world ``i``
    is global world ``lo + i`` and episode numbering continues from
    ``total_worlds`` exactly as in a single simulator of the whole batch.
    """

    def __init__(self, factory, total_worlds, group=None, needs_episode_exchange=True, exchange="collective"):
        if exchange not in ("collective", "mailbox"):
            raise ValueError("exchange must be 'collective' or 'mailbox'")
        self.exchange = exchange
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.total_worlds = int(total_worlds)
        self.lo, self.n = shard_range(total_worlds, self.rank, self.world_size)
        self.sim = factory(self.n)
        self.needs_episode_exchange = needs_episode_exchange
        if needs_episode_exchange:
            self.sim.reseed_shard(self.lo, self.total_worlds)  # world i = global world lo + i; the counter starts at total_worlds
            self._mine = self.sim.shard_count_tensor().to_torch()  # one word, written by phase 1
            self._counts = torch.zeros((self.world_size,), dtype=self._mine.dtype, device=self._mine.device)
            if exchange == "mailbox":
                # one mailbox per rank, mapped by every peer: the only collective of this mode, once
                mine = self.sim.exchange_create(self.world_size, self.rank)
                handles = [mine]
                if dist.is_initialized() and self.world_size > 1:
                    handles = [None] * self.world_size
                    dist.all_gather_object(handles, mine, group=group)
                self.sim.exchange_connect(handles)

    def step(self, actions=None):
        """One step of this rank's worlds.  ``actions``: rank-local action tensor
        (or None to use the simulator's ACTION tensor)."""
        if self.needs_episode_exchange and self.exchange == "mailbox":
            self.sim.step_exchanged(actions)  # phase 1, count + publish into the peers' mailboxes, phase 2 polling its own
            return
        if not self.needs_episode_exchange or not dist.is_initialized():
            # no exchange to make: Overcooked has no episode counter, and without a process group this one shard IS the whole
            # batch, so the simulator's own counter is the global one and the library's best single-GPU step applies
            if actions is None:
                self.sim.step()
            else:
                self.sim.step_with_actions(actions)
            return
        self.sim.step_phase1(actions)
        _all_gather_into(self._counts, self._mine, self.group)  # one int32 per rank
        self.sim.step_phase2_gathered(self._counts, self.rank)

    def close(self):
        """Destroys the rank-local simulator (raises if one of its steps hit SCAN_TIMEOUT)."""
        self.sim.close()

    def gather(self, local, world_dim=0, out=None):
        sizes = None
        if self.total_worlds % self.world_size:
            sizes = [shard_range(self.total_worlds, r, self.world_size)[1] for r in range(self.world_size)]
        return gather_worlds(local, world_dim, self.group, out, sizes)
