"""TEST INFRASTRUCTURE ONLY -- ctypes front end of the CPU oracle (oracle/*.c).

May be imported by tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg, and by nothing else: the product path (the HIP kernels
behind include/mrl_envs.h) never touches this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmrl_oracle.so")
_lib = None

MAX_CELLS = 255
MAX_PLAYERS = 64
NUM_RECIPES = 16
HANABI_OBS = 658
HANABI_STATE = 783
HANABI_MOVES = 20


class OvercookedConfig(ctypes.Structure):
    _fields_ = [
        ("height", ctypes.c_int64), ("width", ctypes.c_int64), ("num_players", ctypes.c_int64),
        ("placement_in_pot_rew", ctypes.c_int64), ("dish_pickup_rew", ctypes.c_int64),
        ("soup_pickup_rew", ctypes.c_int64), ("horizon", ctypes.c_int64),
        ("terrain", ctypes.c_int64 * MAX_CELLS),
        ("start_player_x", ctypes.c_int64 * MAX_PLAYERS),
        ("start_player_y", ctypes.c_int64 * MAX_PLAYERS),
        ("recipe_values", ctypes.c_int64 * NUM_RECIPES),
        ("recipe_times", ctypes.c_int64 * NUM_RECIPES),
    ]


class HanabiConfig(ctypes.Structure):
    _fields_ = [("colors", ctypes.c_uint32), ("ranks", ctypes.c_uint32), ("players", ctypes.c_uint32),
                ("max_information_tokens", ctypes.c_uint32), ("max_life_tokens", ctypes.c_uint32)]


def build(force=False):
    """Compile oracle/libmrl_oracle.so with the committed Makefile (gcc only)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile"]
    if not force and os.path.exists(_LIB_PATH):
        if os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in srcs):
            return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B", "libmrl_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, u32, i32p = ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int32)
        L.orc_overcooked_create.restype = vp
        L.orc_overcooked_create.argtypes = [ctypes.POINTER(OvercookedConfig), u32]
        L.orc_overcooked_destroy.argtypes = [vp]
        L.orc_overcooked_step.argtypes = [vp, i32p, ctypes.c_int]
        for name, rt in (("obs", ctypes.POINTER(ctypes.c_uint8)), ("reward", i32p), ("done", i32p)):
            fn = getattr(L, "orc_overcooked_" + name)
            fn.restype, fn.argtypes = rt, [vp]
        L.orc_overcooked_dump.argtypes = [vp, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint8), i32p]

        L.orc_simplecooked_create.restype = vp
        L.orc_simplecooked_create.argtypes = [ctypes.POINTER(OvercookedConfig), u32]
        L.orc_simplecooked_destroy.argtypes = [vp]
        L.orc_simplecooked_step.argtypes = [vp, i32p, ctypes.c_int]
        for name, rt in (("obs", ctypes.POINTER(ctypes.c_uint8)), ("reward", i32p), ("done", i32p)):
            fn = getattr(L, "orc_simplecooked_" + name)
            fn.restype, fn.argtypes = rt, [vp]
        L.orc_simplecooked_dump.argtypes = [vp, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint8), i32p, i32p]

        L.orc_balance_create.restype = vp
        L.orc_balance_create.argtypes = [u32]
        L.orc_balance_destroy.argtypes = [vp]
        L.orc_balance_step.argtypes = [vp, i32p]
        for name, rt in (("obs", i32p), ("loc", i32p), ("time", i32p), ("reward", ctypes.POINTER(ctypes.c_float)), ("done", i32p)):
            fn = getattr(L, "orc_balance_" + name)
            fn.restype, fn.argtypes = rt, [vp]
        L.orc_balance_episodes.restype = u32
        L.orc_balance_episodes.argtypes = [vp]

        L.orc_cartpole_create.restype = vp
        L.orc_cartpole_create.argtypes = [u32]
        L.orc_cartpole_destroy.argtypes = [vp]
        L.orc_cartpole_step.argtypes = [vp, i32p, ctypes.c_int]
        L.orc_cartpole_state.restype = ctypes.POINTER(ctypes.c_float)
        L.orc_cartpole_state.argtypes = [vp]
        L.orc_cartpole_reward.restype = ctypes.POINTER(ctypes.c_float)
        L.orc_cartpole_reward.argtypes = [vp]
        L.orc_cartpole_done.restype = i32p
        L.orc_cartpole_done.argtypes = [vp]
        L.orc_cartpole_episodes.restype = u32
        L.orc_cartpole_episodes.argtypes = [vp]
        L.orc_rng_seed.restype = u32
        L.orc_rng_seed.argtypes = [u32]
        L.orc_rng_next.restype = ctypes.c_float
        L.orc_rng_next.argtypes = [ctypes.POINTER(u32)]

        L.orc_hanabi_create.restype = vp
        L.orc_hanabi_create.argtypes = [ctypes.POINTER(HanabiConfig), u32]
        L.orc_hanabi_destroy.argtypes = [vp]
        L.orc_hanabi_step.argtypes = [vp, i32p, ctypes.c_int]
        for name, rt in (("obs", ctypes.POINTER(ctypes.c_uint8)), ("state", ctypes.POINTER(ctypes.c_uint8)),
                         ("mask", i32p), ("active", i32p), ("reward", ctypes.POINTER(ctypes.c_float)),
                         ("done", i32p)):
            fn = getattr(L, "orc_hanabi_" + name)
            fn.restype, fn.argtypes = rt, [vp]
        L.orc_hanabi_episodes.restype = u32
        L.orc_hanabi_episodes.argtypes = [vp]
        L.orc_hanabi_record_bytes.restype = u32
        L.orc_hanabi_dump.argtypes = [vp, ctypes.POINTER(ctypes.c_uint8)]
        _lib = L
    return _lib


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    arr = np.ctypeslib.as_array(ptr, shape=(n,))
    return arr.view(dtype).reshape(shape)


def _as_i32(actions, shape):
    a = np.ascontiguousarray(np.asarray(actions).reshape(shape), dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


class OvercookedOracle:
    """N worlds of the reference Overcooked step on the CPU.

    ``params`` is the dict the reference's ``get_base_layout_params`` returns
    (envs/overcooked_env.py:261-371)."""

    def __init__(self, params, num_worlds, num_threads=1):
        self.L = lib()
        cfg = OvercookedConfig()
        for k in ("height", "width", "num_players", "placement_in_pot_rew", "dish_pickup_rew",
                  "soup_pickup_rew", "horizon"):
            setattr(cfg, k, int(params[k]))
        for k in ("terrain", "start_player_x", "start_player_y", "recipe_values", "recipe_times"):
            arr = getattr(cfg, k)
            for i, v in enumerate(params[k]):
                arr[i] = int(v)
        self.P, self.H, self.W = int(params["num_players"]), int(params["height"]), int(params["width"])
        self.C, self.F, self.N = self.H * self.W, 5 * self.P + 16, int(num_worlds)
        self.num_threads = num_threads
        self.h = self.L.orc_overcooked_create(ctypes.byref(cfg), self.N)
        if not self.h:
            raise ValueError("oracle rejected the Overcooked config")
        self.obs = _view(self.L.orc_overcooked_obs(self.h), (self.N, self.P, self.C, self.F), np.uint8)
        self.reward = _view(self.L.orc_overcooked_reward(self.h), (self.P, self.N), np.int32)
        self.done = _view(self.L.orc_overcooked_done(self.h), (self.N,), np.int32)

    def step(self, actions):
        a, p = _as_i32(actions, (self.P, self.N))
        self.L.orc_overcooked_step(self.h, p, self.num_threads)

    def dump(self):
        pl = np.zeros((self.N, self.P, 6), np.uint8)
        ob = np.zeros((self.N, self.C, 4), np.uint8)
        ts = np.zeros((self.N,), np.int32)
        u8 = ctypes.POINTER(ctypes.c_uint8)
        self.L.orc_overcooked_dump(self.h, pl.ctypes.data_as(u8), ob.ctypes.data_as(u8),
                                   ts.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        return pl, ob, ts

    def close(self):
        if self.h:
            self.L.orc_overcooked_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SimplecookedOracle:
    """N worlds of the reference's overcooked2_env ("Simplecooked") step on the CPU.

    ``params``: what the reference's ``envs/overcooked2_env.py:get_base_layout_params`` returns
    (terrain ints in overcooked2's enum order, 16-entry recipe tables, shaping rewards, horizon)."""

    def __init__(self, params, num_worlds, num_threads=1):
        self.L = lib()
        cfg = OvercookedConfig()
        for k in ("height", "width", "num_players", "placement_in_pot_rew", "dish_pickup_rew",
                  "soup_pickup_rew", "horizon"):
            setattr(cfg, k, int(params[k]))
        for k in ("terrain", "start_player_x", "start_player_y", "recipe_values", "recipe_times"):
            arr = getattr(cfg, k)
            for i, v in enumerate(params[k]):
                arr[i] = int(v)
        self.P, self.H, self.W = int(params["num_players"]), int(params["height"]), int(params["width"])
        self.C, self.F, self.N = self.H * self.W, 5 * self.P + 10, int(num_worlds)
        self.num_threads = num_threads
        self.h = self.L.orc_simplecooked_create(ctypes.byref(cfg), self.N)
        if not self.h:
            raise ValueError("oracle rejected the Simplecooked config")
        self.obs = _view(self.L.orc_simplecooked_obs(self.h), (self.N, self.P, self.C, self.F), np.uint8)
        self.reward = _view(self.L.orc_simplecooked_reward(self.h), (self.P, self.N), np.int32)
        self.done = _view(self.L.orc_simplecooked_done(self.h), (self.N,), np.int32)

    def step(self, actions):
        a, p = _as_i32(actions, (self.P, self.N))
        self.L.orc_simplecooked_step(self.h, p, self.num_threads)

    def dump(self):
        pl = np.zeros((self.N, self.P, 6), np.uint8)
        ob = np.zeros((self.N, self.C, 4), np.uint8)
        ts = np.zeros((self.N,), np.int32)
        dishes = np.zeros((self.N,), np.int32)
        u8, i32 = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int32)
        self.L.orc_simplecooked_dump(self.h, pl.ctypes.data_as(u8), ob.ctypes.data_as(u8), ts.ctypes.data_as(i32),
                                     dishes.ctypes.data_as(i32))
        return pl, ob, ts, dishes

    def close(self):
        if self.h:
            self.L.orc_simplecooked_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BalanceOracle:
    """N worlds of the reference's balance-beam step on the CPU (worlds in order: episode numbering)."""

    def __init__(self, num_worlds):
        self.L = lib()
        self.N = int(num_worlds)
        self.h = self.L.orc_balance_create(self.N)
        self.obs = _view(self.L.orc_balance_obs(self.h), (2, self.N, 7), np.int32)
        self.loc = _view(self.L.orc_balance_loc(self.h), (2, self.N), np.int32)
        self.time = _view(self.L.orc_balance_time(self.h), (self.N,), np.int32)
        self.reward = _view(self.L.orc_balance_reward(self.h), (2, self.N), np.float32)
        self.done = _view(self.L.orc_balance_done(self.h), (self.N,), np.int32)

    def step(self, actions):
        a, p = _as_i32(actions, (2, self.N))
        self.L.orc_balance_step(self.h, p)

    def plant(self, obs):
        """Overwrite the state with observation rows (2, N, 7): positions and time are read back from them."""
        self.obs[...] = obs
        self.loc[...] = obs[:, :, 0] - 2
        self.time[...] = obs[0, :, 6]

    @property
    def episodes(self):
        return int(self.L.orc_balance_episodes(self.h))

    def close(self):
        if self.h:
            self.L.orc_balance_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CartpoleOracle:
    def __init__(self, num_worlds, num_threads=1):
        self.L = lib()
        self.N = int(num_worlds)
        self.num_threads = num_threads
        self.h = self.L.orc_cartpole_create(self.N)
        self.state = _view(self.L.orc_cartpole_state(self.h), (self.N, 4), np.float32)
        self.reward = _view(self.L.orc_cartpole_reward(self.h), (self.N, 1), np.float32)
        self.done = _view(self.L.orc_cartpole_done(self.h), (self.N, 1), np.int32)

    def step(self, actions):
        a, p = _as_i32(actions, (self.N,))
        self.L.orc_cartpole_step(self.h, p, self.num_threads)

    @property
    def episodes(self):
        return int(self.L.orc_cartpole_episodes(self.h))

    def close(self):
        if self.h:
            self.L.orc_cartpole_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HanabiOracle:
    def __init__(self, config, num_worlds, num_threads=1):
        self.L = lib()
        self.N = int(num_worlds)
        self.num_threads = num_threads
        cfg = HanabiConfig(int(config["colors"]), int(config["ranks"]), int(config["players"]),
                           int(config["max_information_tokens"]), int(config["max_life_tokens"]))
        self.h = self.L.orc_hanabi_create(ctypes.byref(cfg), self.N)
        if not self.h:
            raise ValueError("oracle rejected the Hanabi config")
        N = self.N
        self.obs = _view(self.L.orc_hanabi_obs(self.h), (2, N, HANABI_OBS), np.uint8)
        self.state = _view(self.L.orc_hanabi_state(self.h), (2, N, HANABI_STATE), np.uint8)
        self.mask = _view(self.L.orc_hanabi_mask(self.h), (2, N, HANABI_MOVES), np.int32)
        self.active = _view(self.L.orc_hanabi_active(self.h), (2, N), np.int32)
        self.reward = _view(self.L.orc_hanabi_reward(self.h), (2, N), np.float32)
        self.done = _view(self.L.orc_hanabi_done(self.h), (N,), np.int32)

    def step(self, actions):
        a, p = _as_i32(actions, (2, self.N))
        self.L.orc_hanabi_step(self.h, p, self.num_threads)

    @property
    def episodes(self):
        return int(self.L.orc_hanabi_episodes(self.h))

    def dump(self):
        nb = int(self.L.orc_hanabi_record_bytes())
        rec = np.zeros((self.N, nb), np.uint8)
        self.L.orc_hanabi_dump(self.h, rec.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
        return rec

    def close(self):
        if self.h:
            self.L.orc_hanabi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rng_seed(episode):
    return int(lib().orc_rng_seed(int(episode)))


def rng_stream(episode, count):
    """First ``count`` floats the reference generator yields for an episode index."""
    L = lib()
    st = ctypes.c_uint32(L.orc_rng_seed(int(episode)))
    return np.array([L.orc_rng_next(ctypes.byref(st)) for _ in range(count)], dtype=np.float32)
