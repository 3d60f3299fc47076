"""Python-visible simulator classes: drop-in for the reference's nanobind modules.

    reference module                            class here
    build.madrona_overcooked_example_python  -> OvercookedSimulator   (src/overcooked_env/bindings.cpp:11-84)
    build.madrona_simplecooked_example_python -> SimplecookedSimulator (src/overcooked2_env/bindings.cpp:11-84)
    build.madrona_hanabi_example_python      -> HanabiSimulator       (src/hanabi_env/bindings.cpp:8-48)
    build.madrona_cartpole_example_python    -> CartpoleSimulator     (src/cartpole_env/bindings.cpp:8-31)
    build.madrona_balance_example_python     -> BalanceBeamSimulator  (src/balance_beam_env/bindings.cpp:8-34)

Same constructor keywords, same method names; every ``*_tensor()`` returns an
object whose ``to_torch()`` yields a persistent zero-copy ``torch.Tensor`` on the
simulator's GPU (the reference's ``madrona.py.Tensor.to_torch()``).  ``madrona``
below mimics the ``<module>.madrona`` submodule the wrappers reach for
(envs/overcooked_env.py:31).

Everything is computed by libmrl_envs.so (HIP, gfx950).  ``ExecMode.CPU`` raises:
the reference's CPU TaskGraph executor is out of scope, and a
silent CPU path would defeat the parity tests.
"""
import ctypes
import enum

import torch

from . import _lib
from ._lib import MrlError  # noqa: F401  (re-export)



# The raw handle of torch's current stream on a device.  ``torch.cuda.current_stream(d).cuda_stream`` builds a Stream
# object on every call (1.1 us measured, a fifth of a step call's host cost -- small batches are bound by that cost,
# tools/host_overhead.py); torch's own raw getter returns the same pointer as an int.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device_index):
    if _raw_stream is not None:
        return _raw_stream(device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


class ExecMode(enum.Enum):
    CPU = 0
    CUDA = 1
    HIP = 1  # alias: on this engine the GPU mode is HIP on MI355X


class madrona:  # noqa: N801  (mirrors `<module>.madrona.ExecMode`)
    ExecMode = ExecMode


_TORCH_DTYPE = {
    _lib.MRL_INT8: (torch.int8, "|i1", 1),
    _lib.MRL_UINT8: (torch.uint8, "|u1", 1),
    _lib.MRL_INT32: (torch.int32, "<i4", 4),
    _lib.MRL_FLOAT32: (torch.float32, "<f4", 4),
    _lib.MRL_UINT32: (torch.int32, "<i4", 4),  # torch has no general uint32; same bits
}


class _DeviceBlob:
    """Carrier for ``__cuda_array_interface__``; keeps the simulator alive."""

    def __init__(self, owner, ptr, shape, strides_bytes, typestr):
        self._owner = owner
        self.__cuda_array_interface__ = {
            "shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2,
            "strides": tuple(strides_bytes),
        }


class Tensor:
    """What ``sim.*_tensor()`` returns (stands in for ``madrona::py::Tensor``)."""

    def __init__(self, sim, slot):
        desc = _lib.TensorDesc()
        _lib.check(_lib.lib().mrl_tensor(sim._handle, slot, ctypes.byref(desc)))
        self._sim = sim
        self.ptr = desc.data
        self.dtype_code = desc.dtype
        self.shape = tuple(desc.shape[i] for i in range(desc.ndim))
        self.strides = tuple(desc.strides[i] for i in range(desc.ndim))
        self.device = desc.device
        self._torch = None

    def to_torch(self):
        if self._torch is None:
            dtype, typestr, size = _TORCH_DTYPE[self.dtype_code]
            blob = _DeviceBlob(self._sim, self.ptr, self.shape, [s * size for s in self.strides], typestr)
            t = torch.as_tensor(blob, device=torch.device("cuda", self.device))
            if t.dtype != dtype:
                t = t.view(dtype)
            if t.data_ptr() != self.ptr:
                raise MrlError("torch.as_tensor copied the exported buffer instead of aliasing it")
            self._torch = t
        return self._torch


def random_hash(seed, step, world, player):
    """The counter-based hash behind the device-side random policies (``mrl_rollout_random``;
    csrc/random_policy.hpp), restated with numpy so a stream can be replayed.  uint32 array."""
    import numpy as np
    m = np.uint64(0xFFFFFFFF)
    seed = int(seed) & (2 ** 64 - 1)
    h = (np.uint64(seed & 0xFFFFFFFF) ^ (np.uint64(step) * np.uint64(0x9E3779B9) & m) ^
         (np.asarray(world, np.uint64) * np.uint64(0x85EBCA6B) & m) ^
         ((np.asarray(player, np.uint64) + np.uint64(1)) * np.uint64(0xC2B2AE35) & m) ^
         (np.uint64(seed >> 32) * np.uint64(0x27D4EB2F) & m))
    h ^= h >> np.uint64(16)
    h = h * np.uint64(0x7FEB352D) & m
    h ^= h >> np.uint64(15)
    h = h * np.uint64(0x846CA68B) & m
    h ^= h >> np.uint64(16)
    return h.astype(np.uint32)


def random_action(seed, step, world, player):
    """Overcooked: ``(hash * 6) >> 32`` for (step index, world, player)."""
    import numpy as np
    return ((random_hash(seed, step, world, player).astype(np.uint64) * np.uint64(6)) >> np.uint64(32)).astype(np.int32)


def random_cartpole_action(seed, step, world):
    """Cartpole: the hash's top bit (player 0)."""
    import numpy as np
    return (random_hash(seed, step, world, np.zeros_like(np.asarray(world))) >> np.uint32(31)).astype(np.int32)


def random_hanabi_action(seed, step, world, mover, legal_mask):
    """Hanabi: the k-th legal move of the mover, ``k = (hash * count) >> 32``.
    ``legal_mask``: (n, 20) 0/1 array of the mover's legal moves, ``mover``: (n,) 0/1."""
    import numpy as np
    legal_mask = np.asarray(legal_mask) != 0
    count = legal_mask.sum(-1).astype(np.uint64)
    k = (random_hash(seed, step, world, mover).astype(np.uint64) * count) >> np.uint64(32)
    rank = np.cumsum(legal_mask, axis=-1) - 1                      # rank of each legal move among the legal ones
    pick = legal_mask & (rank == k[:, None].astype(np.int64))
    return np.where(count > 0, pick.argmax(-1), 0).astype(np.int32)


class _Simulator:
    """Shared handle management for the three games."""

    _SLOTS = {}

    def __init__(self, exec_mode, gpu_id):
        mode = getattr(exec_mode, "name", str(exec_mode))
        if mode == "CPU" or exec_mode == 0:
            raise NotImplementedError(
                "ExecMode.CPU: this engine has only the HIP (MI355X) step kernels; the reference's CPU "
                "TaskGraph executor is not reproduced (the test-only CPU oracle lives in oracle/).")
        self._L = _lib.lib()
        self._handle = ctypes.c_void_p()
        self.gpu_id = int(gpu_id)
        self._tensors = {}

    # --- reference API -------------------------------------------------
    def step(self):
        """One environment step for all worlds, enqueued on torch's current stream
        (Manager::step; no host synchronisation)."""
        stream = _stream_ptr(self.gpu_id)
        rc = self._L.mrl_step(self._handle, stream)
        if rc:
            _lib.check(rc)

    # --- extensions ----------------------------------------------------
    def _action_pointer(self, actions, steps=None):
        """Every entry point that hands the library a caller-owned action array goes through here: the
        library reads raw int32 words, so dtype, layout, device and size are checked on this side."""
        if not isinstance(actions, torch.Tensor) or not actions.is_cuda or actions.device.index != self.gpu_id:
            raise ValueError(f"actions must be a tensor on cuda:{self.gpu_id} (the simulator's device)")
        if actions.dtype != torch.int32 or not actions.is_contiguous():
            raise ValueError("actions must be a contiguous int32 tensor on the simulator's device")
        expect = self._action_numel if steps is None else steps * self._action_numel
        if actions.numel() != expect:
            raise ValueError(f"actions has {actions.numel()} elements, expected {expect}")
        return actions.data_ptr()

    def step_with_actions(self, actions):
        """Step reading actions from ``actions`` (int32, the ACTION tensor's shape,
        contiguous, on this GPU) instead of the ACTION tensor."""
        ptr = self._action_pointer(actions)
        stream = _stream_ptr(self.gpu_id)
        rc = self._L.mrl_step_with_actions(self._handle, ptr, stream)
        if rc:
            _lib.check(rc)

    def step_with_actions_i64(self, actions):
        """Step reading ``actions`` as int64 (the ACTION tensor's shape, contiguous, on this GPU): the step kernel narrows
        them itself and mirrors them into the ACTION tensor (``mrl_step_with_actions_i64``; Overcooked, Simplecooked)."""
        if (not isinstance(actions, torch.Tensor) or not actions.is_cuda or actions.device.index != self.gpu_id or
                actions.dtype != torch.int64 or not actions.is_contiguous() or actions.numel() != self._action_numel):
            raise ValueError("actions must be a contiguous int64 tensor of the ACTION tensor's size on the simulator's device")
        self._step_i64_checked(actions.data_ptr())

    def _step_i64_checked(self, ptr):
        """``mrl_step_with_actions_i64`` on a pointer the caller has already validated (the env wrappers check the
        harness's tensor once, not twice: every microsecond of host work per call shows at small batches)."""
        rc = self._L.mrl_step_with_actions_i64(self._handle, ptr, _stream_ptr(self.gpu_id))
        if rc:
            _lib.check(rc)

    def step_sequence(self, actions):
        """One step per leading index of ``actions`` (int32, shape (K,) + ACTION tensor's shape, contiguous,
        on this GPU): same results as K ``step_with_actions`` calls (``mrl_step_sequence``)."""
        if not isinstance(actions, torch.Tensor) or actions.dim() < 1:
            raise ValueError("actions must hold K consecutive ACTION tensors")
        ptr = self._action_pointer(actions, steps=int(actions.shape[0]))
        stream = _stream_ptr(self.gpu_id)
        _lib.check(self._L.mrl_step_sequence(self._handle, ptr, int(actions.shape[0]), stream))

    def rollout_random(self, num_steps, seed=0, first_step=0):
        """``num_steps`` steps under the uniform random policy, actions drawn on the
        device (``mrl_rollout_random``; see ``random_action`` for the stream)."""
        stream = _stream_ptr(self.gpu_id)
        _lib.check(self._L.mrl_rollout_random(self._handle, int(num_steps), int(seed) & (2 ** 64 - 1), int(first_step),
                                              stream))

    def step_phase1(self, actions=None):
        ptr = self._action_pointer(actions) if actions is not None else None
        stream = _stream_ptr(self.gpu_id)
        _lib.check(self._L.mrl_step_phase1(self._handle, ptr, stream))

    def _word_pointer(self, words, count, what):
        """A caller-owned array of ``count`` 32-bit words the library reads on the device (episode base, gathered
        shard counts): checked here like the action arrays, the C ABI takes a raw pointer."""
        if (not isinstance(words, torch.Tensor) or not words.is_cuda or words.device.index != self.gpu_id or
                words.dtype not in (torch.int32, torch.uint32) or not words.is_contiguous() or words.numel() != count):
            raise ValueError(f"{what} must be a contiguous int32 tensor of {count} element(s) on cuda:{self.gpu_id}")
        return words.data_ptr()

    def step_phase2(self, episode_base=None):
        """``episode_base``: 1-element int32/uint32 CUDA tensor, or None for the
        simulator's own counter."""
        stream = _stream_ptr(self.gpu_id)
        ptr = self._word_pointer(episode_base, 1, "episode_base") if episode_base is not None else None
        _lib.check(self._L.mrl_step_phase2(self._handle, ptr, stream))

    def step_phase2_gathered(self, counts, rank):
        """Phase 2 of rank ``rank`` of a sharded batch: ``counts`` = every rank's SHARD_COUNT of this step (int32 CUDA
        tensor, one element per rank, what an all-gather of ``shard_count_tensor()`` delivers).  The re-seeding launch
        works out its own episode base and advances the simulator's counter (``mrl_step_phase2_gathered``)."""
        if not isinstance(counts, torch.Tensor):
            raise ValueError("counts must be a tensor")
        ptr = self._word_pointer(counts, counts.numel(), "counts")
        _lib.check(self._L.mrl_step_phase2_gathered(self._handle, ptr, int(counts.numel()), int(rank), _stream_ptr(self.gpu_id)))

    def exchange_create(self, num_ranks, rank):
        """This rank's mailbox of the collective-free shard exchange (``mrl_exchange_create``) -> its IPC handle (64 bytes),
        to be handed to every rank."""
        buf = ctypes.create_string_buffer(_lib.IPC_HANDLE_BYTES)
        _lib.check(self._L.mrl_exchange_create(self._handle, int(num_ranks), int(rank), buf))
        return buf.raw

    def exchange_connect(self, handles):
        """``handles``: the IPC handles of all ranks, in rank order (``mrl_exchange_connect``)."""
        blob = b"".join(handles)
        if len(blob) % _lib.IPC_HANDLE_BYTES:
            raise ValueError("handles must be 64 bytes each")
        _lib.check(self._L.mrl_exchange_connect(self._handle, blob))

    def step_exchanged(self, actions=None):
        """One step of a shard whose ranks exchange their finished counts through the mailboxes (``mrl_step_exchanged``):
        phase 1, count + publish, phase 2 polling -- no collective, no host call in between."""
        ptr = None if actions is None else self._action_pointer(actions)
        _lib.check(self._L.mrl_step_exchanged(self._handle, ptr, _stream_ptr(self.gpu_id)))

    def set_observation_output(self, out):
        """Later steps write their observations into ``out`` -- an int8 CUDA tensor of the world-major shape
        (N, P, H, W, F), contiguous, e.g. one slot of a rollout buffer -- instead of the simulator's own tensor; ``None``
        hands the output back (``mrl_set_observation_output``; Overcooked and Simplecooked).  The caller keeps ``out``
        alive while steps that write to it are in flight."""
        if out is None:
            _lib.check(self._L.mrl_set_observation_output(self._handle, None, 0))
            return
        if (not isinstance(out, torch.Tensor) or not out.is_cuda or out.device.index != self.gpu_id or
                out.dtype not in (torch.int8, torch.uint8) or not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous int8 tensor on cuda:{self.gpu_id}")
        _lib.check(self._L.mrl_set_observation_output(self._handle, out.data_ptr(), out.numel()))

    def prepare_graph_capture(self):
        """Makes this simulator's steps capturable in a HIP graph (``torch.cuda.graph``): Hanabi, Cartpole and the balance beam
        move their launch-to-launch counter state into device memory (``mrl_prepare_graph_capture``; one extra one-thread
        launch per step from then on); nothing to do for Overcooked and Simplecooked.  Call it outside the capture."""
        _lib.check(self._L.mrl_prepare_graph_capture(self._handle, _stream_ptr(self.gpu_id)))

    def set_observation_ring(self, ring):
        """``ring``: an int8 CUDA tensor (T, N, P, H, W, F) whose slots ``ring[s]`` are contiguous -- a rollout buffer.  Step
        number k from this call on writes its observations to ``ring[k % T]``, whether it is a call of its own or step k of
        ``rollout_random`` / ``step_sequence`` (``mrl_set_observation_ring``); ``None`` hands the output back."""
        if ring is None:
            _lib.check(self._L.mrl_set_observation_ring(self._handle, None, 0, 0))
            return
        if (not isinstance(ring, torch.Tensor) or not ring.is_cuda or ring.device.index != self.gpu_id or ring.dim() < 2 or
                ring.dtype not in (torch.int8, torch.uint8) or not ring[0].is_contiguous()):
            raise ValueError(f"ring must be an int8 tensor (T, N, P, H, W, F) on cuda:{self.gpu_id} with contiguous slots")
        _lib.check(self._L.mrl_set_observation_ring(self._handle, ring.data_ptr(), int(ring.stride(0)), int(ring.shape[0])))

    @property
    def scan_timed_out(self):
        """True once a bounded in-kernel wait has expired (``mrl_scan_timed_out``); every later step raises."""
        return bool(self._L.mrl_scan_timed_out(self._handle))

    def reseed_shard(self, world_offset, num_worlds_total):
        stream = _stream_ptr(self.gpu_id)
        _lib.check(self._L.mrl_reseed_shard(self._handle, int(world_offset), int(num_worlds_total), stream))

    def set_episode_counter(self, next_episode):
        stream = _stream_ptr(self.gpu_id)
        _lib.check(self._L.mrl_set_episode_counter(self._handle, int(next_episode), stream))

    @property
    def kernel_name(self):
        return self._L.mrl_kernel_name(self._handle).decode()

    @property
    def rollout_kernel_name(self):
        """The kernel the next ``rollout_random`` runs (a persistent one, or the step kernel once per step)."""
        return self._L.mrl_rollout_kernel_name(self._handle).decode()

    @property
    def bytes_per_world_step(self):
        return int(self._L.mrl_bytes_per_world_step(self._handle))

    @property
    def launch_shape(self):
        """(workgroups, threads per workgroup, LDS bytes per workgroup, worlds per wavefront) of the step kernel."""
        out = (ctypes.c_uint32 * 4)()
        _lib.check(self._L.mrl_launch_shape(self._handle, ctypes.byref(out)))
        return tuple(int(v) for v in out)

    @property
    def num_worlds(self):
        return int(self._L.mrl_num_worlds(self._handle))

    def _tensor(self, slot):
        if slot not in self._tensors:
            self._tensors[slot] = Tensor(self, slot)
        return self._tensors[slot]

    def close(self):
        """Destroys the simulator.  Raises ``MrlError`` if one of its steps ran into SCAN_TIMEOUT (the
        results since then carry unspecified episode numbers) -- after freeing it all the same."""
        if getattr(self, "_handle", None) is not None and self._handle.value:
            bad = bool(self._L.mrl_scan_timed_out(self._handle))
            self._L.mrl_destroy(self._handle)
            self._handle = ctypes.c_void_p()
            if bad:
                raise MrlError("an in-kernel wait expired during this simulator's life (SCAN_TIMEOUT): "
                               "episode numbers since then are unspecified")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OvercookedSimulator(_Simulator):
    """Signature of src/overcooked_env/bindings.cpp:14-71."""

    _create = "mrl_overcooked_create"

    def __init__(self, exec_mode, gpu_id, num_worlds, terrain, height, width, num_players, start_player_x,
                 start_player_y, placement_in_pot_rew, dish_pickup_rew, soup_pickup_rew, recipe_values, recipe_times,
                 horizon, debug_compile=True):
        super().__init__(exec_mode, gpu_id)
        if len(terrain) < height * width:
            raise ValueError("terrain shorter than height*width")
        if len(start_player_x) < num_players or len(start_player_y) < num_players:
            raise ValueError("fewer start positions than players")
        if len(recipe_values) < 16 or len(recipe_times) < 16:
            raise ValueError("recipe tables need 16 entries")
        keep = [_lib.i64_array(terrain), _lib.i64_array(start_player_x), _lib.i64_array(start_player_y),
                _lib.i64_array(recipe_values), _lib.i64_array(recipe_times)]
        cfg = _lib.OvercookedConfig(int(height), int(width), int(num_players), int(placement_in_pot_rew),
                                    int(dish_pickup_rew), int(soup_pickup_rew), int(horizon), *keep)
        _lib.check(getattr(self._L, self._create)(ctypes.byref(cfg), int(gpu_id), int(num_worlds), ctypes.byref(self._handle)))
        self.num_players, self.height, self.width = int(num_players), int(height), int(width)
        self._action_numel = int(num_players) * int(num_worlds)

    def done_tensor(self): return self._tensor(0)
    def active_agent_tensor(self): return self._tensor(1)
    def action_tensor(self): return self._tensor(2)
    def observation_tensor(self): return self._tensor(3)
    def agent_state_tensor(self): return self._tensor(3)  # alias, bindings.cpp:77
    def action_mask_tensor(self): return self._tensor(4)
    def reward_tensor(self): return self._tensor(5)
    def world_id_tensor(self): return self._tensor(6)
    def agent_id_tensor(self): return self._tensor(7)
    def location_world_id_tensor(self): return self._tensor(8)
    def location_id_tensor(self): return self._tensor(9)
    # world-major views of this engine (no reference counterpart)
    def observation_world_major_tensor(self): return self._tensor(10)
    def state_players_tensor(self): return self._tensor(11)
    def state_objects_tensor(self): return self._tensor(12)
    def state_timestep_tensor(self): return self._tensor(13)


STEP_MANY_MAX = 8  # simulators per launch (the kernel arguments hold that many parameter blocks: mrl_step_many)


def can_step_with_others(sim):
    """May ``sim`` take part in ``step_many``?  Overcooked simulators whose workgroups share one copy of a world's state
    (a large layout that does not fit one tile, with fewer than 8192 worlds: ``mrl_overcooked_step_team``) step alone."""
    return type(sim) is OvercookedSimulator and "step_team" not in sim.kernel_name


def step_many(sims, actions=None):
    """One launch for several ``OvercookedSimulator``s on one GPU -- any mix of layouts and world counts (``mrl_step_many``).
    ``actions``: None (every simulator's ACTION tensor) or one int32 tensor per simulator (an entry may be None)."""
    if not sims:
        return
    if any(type(s) is not OvercookedSimulator for s in sims):
        raise ValueError("step_many takes OvercookedSimulator instances")
    handles = (ctypes.c_void_p * len(sims))(*[s._handle for s in sims])
    ptrs = None
    if actions is not None:
        if len(actions) != len(sims):
            raise ValueError("one action tensor (or None) per simulator")
        ptrs = (ctypes.c_void_p * len(sims))(*[None if a is None else s._action_pointer(a) for s, a in zip(sims, actions)])
    _lib.check(sims[0]._L.mrl_step_many(handles, len(sims), ptrs, _stream_ptr(sims[0].gpu_id)))


class SimplecookedSimulator(OvercookedSimulator):
    """Signature of src/overcooked2_env/bindings.cpp:14-71 ("Simplecooked": the world the reference's trainer
    uses, train/env_utils.py:3).  Same keyword arguments and tensor getters as ``OvercookedSimulator``; terrain
    values follow overcooked2's enum (tomato source last), rows are 5P + 10 bytes, at most 2 players, 100 cells."""

    _create = "mrl_simplecooked_create"

    def dishes_out_tensor(self): return self._tensor(14)  # WorldState.num_dishes_out, int32 (N)


class HanabiSimulator(_Simulator):
    """Signature of src/hanabi_env/bindings.cpp:10-36."""

    def __init__(self, exec_mode, gpu_id, num_worlds, colors, ranks, players, max_information_tokens,
                 max_life_tokens, debug_compile=True):
        super().__init__(exec_mode, gpu_id)
        cfg = _lib.HanabiConfig(int(colors), int(ranks), int(players), int(max_information_tokens),
                                int(max_life_tokens))
        _lib.check(self._L.mrl_hanabi_create(ctypes.byref(cfg), int(gpu_id), int(num_worlds),
                                             ctypes.byref(self._handle)))
        self._action_numel = 2 * int(num_worlds)

    def done_tensor(self): return self._tensor(0)
    def active_agent_tensor(self): return self._tensor(1)
    def action_tensor(self): return self._tensor(2)
    def observation_tensor(self): return self._tensor(3)
    def action_mask_tensor(self): return self._tensor(4)
    def reward_tensor(self): return self._tensor(5)
    def world_id_tensor(self): return self._tensor(6)
    def agent_id_tensor(self): return self._tensor(7)
    def agent_state_tensor(self): return self._tensor(8)
    def game_tensor(self): return self._tensor(9)
    def reset_count_tensor(self): return self._tensor(10)
    def scan_timeout_tensor(self): return self._tensor(11)
    def shard_count_tensor(self): return self._tensor(12)


class CartpoleSimulator(_Simulator):
    """Signature of src/cartpole_env/bindings.cpp:11-24."""

    def __init__(self, exec_mode, gpu_id, num_worlds, debug_compile=True):
        super().__init__(exec_mode, gpu_id)
        _lib.check(self._L.mrl_cartpole_create(int(gpu_id), int(num_worlds), ctypes.byref(self._handle)))
        self._action_numel = int(num_worlds)

    def reset_tensor(self): return self._tensor(0)
    def action_tensor(self): return self._tensor(1)
    def observation_tensor(self): return self._tensor(2)
    def reward_tensor(self): return self._tensor(3)
    def world_id_tensor(self): return self._tensor(4)
    def reset_count_tensor(self): return self._tensor(5)
    def scan_timeout_tensor(self): return self._tensor(6)
    def shard_count_tensor(self): return self._tensor(7)


class BalanceBeamSimulator(_Simulator):
    """Signature of src/balance_beam_env/bindings.cpp:10-24."""

    def __init__(self, exec_mode, gpu_id, num_worlds, debug_compile=True):
        super().__init__(exec_mode, gpu_id)
        _lib.check(self._L.mrl_balance_create(int(gpu_id), int(num_worlds), ctypes.byref(self._handle)))
        self._action_numel = 2 * int(num_worlds)

    def done_tensor(self): return self._tensor(0)
    def active_agent_tensor(self): return self._tensor(1)
    def action_tensor(self): return self._tensor(2)
    def observation_tensor(self): return self._tensor(3)
    def agent_state_tensor(self): return self._tensor(3)  # alias, bindings.cpp:29
    def action_mask_tensor(self): return self._tensor(4)
    def reward_tensor(self): return self._tensor(5)
    def world_id_tensor(self): return self._tensor(6)
    def agent_id_tensor(self): return self._tensor(7)
    def reset_count_tensor(self): return self._tensor(8)
    def shard_count_tensor(self): return self._tensor(9)


def random_balance_action(seed, step, world, player):
    """Balance beam: ``(hash * 4) >> 32`` for (step index, world, player)."""
    import numpy as np
    return ((random_hash(seed, step, world, player).astype(np.uint64) * np.uint64(4)) >> np.uint64(32)).astype(np.int32)
