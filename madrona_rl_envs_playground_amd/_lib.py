"""ctypes binding of libmrl_envs.so -- the C ABI declared in include/mrl_envs.h.

There is no fallback: if the library is missing or cannot be loaded the import
of any simulator raises.  ``build()`` compiles it with hipcc for gfx950
(cross-compiles without a GPU).
"""
import ctypes
import os
import subprocess

# torch first: its wheel bundles its own HIP runtime (torch/lib/libamdhip64.so).
# If libmrl_envs.so pulled in /opt/rocm's copy before torch is imported, the
# process would hold two runtimes and torch's would find no GPU.
import torch  # noqa: F401

_PKG = os.path.dirname(os.path.abspath(__file__))
# MRL_ENVS_LIB: load another build of the same ABI (the diagnostic build of `make diag`, diag/libmrl_envs_diag.so)
LIB_PATH = os.environ.get("MRL_ENVS_LIB") or os.path.join(_PKG, "libmrl_envs.so")
CSRC = os.path.join(_PKG, "csrc")
HEADER = os.path.join(os.path.dirname(_PKG), "include", "mrl_envs.h")

MRL_OK, MRL_ERR_INVALID, MRL_ERR_DEVICE, MRL_ERR_SLOT = 0, 1, 2, 3
MRL_INT8, MRL_UINT8, MRL_INT32, MRL_FLOAT32, MRL_UINT32 = 0, 1, 2, 3, 4
MAX_DIMS = 6
MAX_RANKS, IPC_HANDLE_BYTES = 16, 64  # MRL_MAX_RANKS, MRL_IPC_HANDLE_BYTES

# every symbol include/mrl_envs.h declares
SYMBOLS = [
    "mrl_overcooked_create", "mrl_hanabi_create", "mrl_cartpole_create", "mrl_step", "mrl_step_with_actions",
    "mrl_step_phase1", "mrl_step_phase2", "mrl_set_episode_counter", "mrl_reseed_shard", "mrl_tensor", "mrl_game",
    "mrl_num_worlds", "mrl_kernel_name", "mrl_rollout_kernel_name", "mrl_bytes_per_world_step", "mrl_destroy", "mrl_last_error",
    "mrl_abi_version", "mrl_rollout_random", "mrl_step_sequence", "mrl_debug_set", "mrl_probe_stream",
    "mrl_scan_timed_out", "mrl_simplecooked_create", "mrl_launch_shape", "mrl_balance_create", "mrl_step_with_actions_i64",
    "mrl_step_phase2_gathered", "mrl_set_observation_output", "mrl_set_observation_ring", "mrl_prepare_graph_capture", "mrl_step_many",
    "mrl_build_hash", "mrl_exchange_create", "mrl_exchange_connect", "mrl_step_exchanged",
]
ABI_VERSION = 4  # MRL_ABI_VERSION of include/mrl_envs.h this binding was written against


class TensorDesc(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("dtype", ctypes.c_int32), ("ndim", ctypes.c_int32),
                ("shape", ctypes.c_int64 * MAX_DIMS), ("strides", ctypes.c_int64 * MAX_DIMS),
                ("device", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class OvercookedConfig(ctypes.Structure):
    _fields_ = [("height", ctypes.c_int64), ("width", ctypes.c_int64), ("num_players", ctypes.c_int64),
                ("placement_in_pot_rew", ctypes.c_int64), ("dish_pickup_rew", ctypes.c_int64),
                ("soup_pickup_rew", ctypes.c_int64), ("horizon", ctypes.c_int64),
                ("terrain", ctypes.POINTER(ctypes.c_int64)), ("start_player_x", ctypes.POINTER(ctypes.c_int64)),
                ("start_player_y", ctypes.POINTER(ctypes.c_int64)), ("recipe_values", ctypes.POINTER(ctypes.c_int64)),
                ("recipe_times", ctypes.POINTER(ctypes.c_int64))]


class HanabiConfig(ctypes.Structure):
    _fields_ = [("colors", ctypes.c_uint32), ("ranks", ctypes.c_uint32), ("players", ctypes.c_uint32),
                ("max_information_tokens", ctypes.c_uint32), ("max_life_tokens", ctypes.c_uint32)]


class MrlError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile csrc/*.hip into libmrl_envs.so for gfx950 (needs hipcc, no GPU).  Up to date means: the hash compiled into
    the library (``mrl_build_hash``) is the hash of the sources lying beside it -- not file times."""
    if not force and embedded_hash(LIB_PATH) == source_hash():
        return LIB_PATH
    jobs = str(min(6, os.cpu_count() or 1))  # one object per kernel file (csrc/Makefile)
    proc = subprocess.run(["make", "-C", CSRC, "-j", jobs] + (["-B"] if force else []), capture_output=True, text=True)
    if proc.returncode == 0 and embedded_hash(LIB_PATH) != source_hash():
        # file times said "up to date" but the binary is of other sources (a stale copy with a newer time): everything again
        proc = subprocess.run(["make", "-C", CSRC, "-j", jobs, "-B"], capture_output=True, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout + proc.stderr)
    if proc.returncode != 0:
        raise MrlError("building libmrl_envs.so failed:\n" + proc.stdout + proc.stderr)
    if embedded_hash(LIB_PATH) != source_hash():
        raise MrlError(f"{LIB_PATH} was rebuilt but carries hash {embedded_hash(LIB_PATH)}, the sources hash to {source_hash()}")
    return LIB_PATH


def source_hash():
    """sha256 (first 16 hex digits) over the sources libmrl_envs.so is built from: name NUL contents of csrc/*.hip, *.hpp and
    the Makefile in sorted order, then include/mrl_envs.h -- the value csrc/Makefile compiles into the library
    (``mrl_build_hash()``).  Measurements that cannot be taken inside a benchmark run (PMC counters:
    profiles/step_traffic.json) carry the LIBRARY's hash, and are only quoted for the build they were taken on."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")) or f == "Makefile")
    for path in [os.path.join(CSRC, f) for f in files] + [HEADER]:
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


_HASH_TAG = b"MRL_SOURCE_HASH="


def embedded_hash(path):
    """The source hash compiled into a library file, read from its bytes without loading it (None: no such file or no tag)."""
    try:
        blob = open(path, "rb").read()
    except OSError:
        return None
    at = blob.find(_HASH_TAG)
    if at < 0:
        return None
    value = blob[at + len(_HASH_TAG):at + len(_HASH_TAG) + 16]
    return value.decode() if len(value) == 16 and all(c in b"0123456789abcdef" for c in value) else None


def build_hash():
    """``mrl_build_hash()`` of the loaded library."""
    return lib().mrl_build_hash().decode()


_lib = None


def lib():
    """The loaded library.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MrlError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback for the step kernels.")
    if os.path.isdir(CSRC) and embedded_hash(LIB_PATH) != source_hash():
        # the binary must prove which sources it was built from: rebuild (hipcc cross-compiles anywhere) or refuse
        stale = embedded_hash(LIB_PATH)
        try:
            build()
        except (MrlError, OSError) as exc:
            raise MrlError(f"{LIB_PATH} was built from other sources than those beside it (library {stale}, sources "
                           f"{source_hash()}) and could not be rebuilt: {exc}") from None
    L = ctypes.CDLL(LIB_PATH)
    vp, u32, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int
    L.mrl_overcooked_create.argtypes = [ctypes.POINTER(OvercookedConfig), i32, u32, ctypes.POINTER(vp)]
    L.mrl_simplecooked_create.argtypes = [ctypes.POINTER(OvercookedConfig), i32, u32, ctypes.POINTER(vp)]
    L.mrl_launch_shape.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32 * 4)]
    L.mrl_hanabi_create.argtypes = [ctypes.POINTER(HanabiConfig), i32, u32, ctypes.POINTER(vp)]
    L.mrl_cartpole_create.argtypes = [i32, u32, ctypes.POINTER(vp)]
    L.mrl_balance_create.argtypes = [i32, u32, ctypes.POINTER(vp)]
    L.mrl_step.argtypes = [vp, vp]
    L.mrl_step_with_actions.argtypes = [vp, vp, vp]
    L.mrl_step_with_actions_i64.argtypes = [vp, vp, vp]
    L.mrl_step_many.argtypes = [ctypes.POINTER(vp), u32, ctypes.POINTER(vp), vp]
    L.mrl_step_phase1.argtypes = [vp, vp, vp]
    L.mrl_step_phase2.argtypes = [vp, vp, vp]
    L.mrl_step_phase2_gathered.argtypes = [vp, vp, u32, u32, vp]
    L.mrl_exchange_create.argtypes = [vp, u32, u32, ctypes.c_char_p]
    L.mrl_exchange_connect.argtypes = [vp, ctypes.c_char_p]
    L.mrl_step_exchanged.argtypes = [vp, vp, vp]
    L.mrl_set_observation_output.argtypes = [vp, vp, ctypes.c_uint64]
    L.mrl_set_observation_ring.argtypes = [vp, vp, ctypes.c_uint64, u32]
    L.mrl_prepare_graph_capture.argtypes = [vp, vp]
    L.mrl_set_episode_counter.argtypes = [vp, u32, vp]
    L.mrl_reseed_shard.argtypes = [vp, u32, u32, vp]
    L.mrl_rollout_random.argtypes = [vp, u32, ctypes.c_uint64, u32, vp]
    L.mrl_step_sequence.argtypes = [vp, vp, u32, vp]
    L.mrl_tensor.argtypes = [vp, i32, ctypes.POINTER(TensorDesc)]
    L.mrl_game.argtypes = [vp]
    L.mrl_num_worlds.argtypes = [vp]
    L.mrl_num_worlds.restype = u32
    L.mrl_kernel_name.argtypes = [vp]
    L.mrl_kernel_name.restype = ctypes.c_char_p
    L.mrl_rollout_kernel_name.argtypes = [vp]
    L.mrl_rollout_kernel_name.restype = ctypes.c_char_p
    L.mrl_bytes_per_world_step.argtypes = [vp]
    L.mrl_bytes_per_world_step.restype = ctypes.c_uint64
    L.mrl_destroy.argtypes = [vp]
    L.mrl_destroy.restype = None
    L.mrl_last_error.restype = ctypes.c_char_p
    L.mrl_abi_version.restype = i32
    L.mrl_build_hash.restype = ctypes.c_char_p
    L.mrl_debug_set.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    L.mrl_probe_stream.argtypes = [vp, vp, ctypes.c_uint64, i32, i32, vp]
    L.mrl_scan_timed_out.argtypes = [vp]
    if L.mrl_abi_version() != ABI_VERSION:
        raise MrlError(f"{LIB_PATH} implements ABI version {L.mrl_abi_version()}, this binding expects {ABI_VERSION}: "
                       "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
    if os.path.isdir(CSRC) and L.mrl_build_hash().decode() != source_hash():
        raise MrlError(f"{LIB_PATH} reports build hash {L.mrl_build_hash().decode()}, the sources beside it hash to {source_hash()}")
    _lib = L
    return L


def debug_set(key, value):
    """Test / measurement knob for the NEXT simulator created (``mrl_debug_set``); ``debug_set(None, 0)`` clears all."""
    check(lib().mrl_debug_set(key.encode() if key is not None else None, int(value)))


class debug_knobs:
    """``with debug_knobs({"fused_step": 1}): sim = ...`` -- knobs apply to simulators created inside."""

    def __init__(self, knobs):
        self.knobs = dict(knobs)

    def __enter__(self):
        for k, v in self.knobs.items():
            debug_set(k, v)
        return self

    def __exit__(self, *exc):
        debug_set(None, 0)
        return False


def check(rc):
    if rc != MRL_OK:
        msg = lib().mrl_last_error().decode() or f"error code {rc}"
        raise MrlError(msg)


def i64_array(values):
    arr = (ctypes.c_int64 * len(values))(*[int(v) for v in values])
    return arr
