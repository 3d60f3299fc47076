// extern "C" entry points of libmrl_envs.so (include/mrl_envs.h).
#include "common.hpp"
#include "episode_scan.hpp"

#include <algorithm>
#include <cstring>
#include <exception>
#include <map>
#include <mutex>
#include <string>

namespace mrl {

static thread_local char g_error[512] = "";

// mrl_debug_set: the knobs tests and measurement tools may turn (the library reads no environment variable)
static const char *const kDebugKeys[] = {
    "overcooked.wpw",        // worlds per wave (0 = chosen by the library)
    "overcooked.whole_max",  // largest single-pass observation tile, bytes
    "overcooked.lds_max",    // LDS budget per workgroup, bytes
    "overcooked.share_max_players",  // experiment: up to how many players the four waves of a workgroup share one world of a large layout
    "overcooked.share_private",  // 1: waves that share a world keep a private copy of its state each (step_body) instead of one per workgroup (team_body)
    "overcooked.no_share",   // 1: never let the waves of a workgroup share one world
    "overcooked.lds_pad",    // experiment: extra LDS bytes per workgroup (limits how many are resident per CU)
    "overcooked.no_fixed",   // 1: never use the kernels specialised for one layout size
    "overcooked.no_direct",  // 1: the single-pass encode looks for its dynamic cells (cell -> player map + ballot compaction) instead of
                             // taking them from the player lanes and the holder-cell table
    "overcooked.whole_store",   // single-pass stream-out stores: 0 by slab size and group alignment (default), 1 write-through, 2 plain
    "overcooked.store_policy",  // multi-pass stream-out stores: 0 by slab size (default), 1 sc1 write-through, 2 plain, 3 nt
    "overcooked.wide_rollout",  // 1: the multi-step launches keep the single step's group size (default: twice as wide where it fits)
    "overcooked.groups",     // groups of worlds a wave steps one after the other in the single step of the standard layouts: 0 by batch size, 1, 2
    "overcooked.shared_consts",  // 1: constants through the workgroup-shared LDS block + barrier even where a private copy would do
    "overcooked.variant",    // 0: the library's choice; 1: force the generic (lane = world) transition
    "hanabi.variant",        // cap on the encoder variant (0 = the generic encoders)
    "hanabi.pairing",        // phase A of the single-launch step: 4 (default) four leader waves step their own and wave w + 4's worlds with all
                             // 64 lanes, 1 the pairs are (2k, 2k + 1), 0 every wave steps its own 32 worlds
    "hanabi.no_persistent",  // 1: mrl_rollout_random as one launch per step
    "cartpole.no_persistent",
    "cartpole.persistent_max",  // largest batch mrl_rollout_random runs as ONE persistent launch (above: one single-launch step per step)
    "cartpole.variant",      // arithmetic of the transition: 0 the library's default, 1 typed and rounded as the reference writes it (four double
                             // divisions), 2 the same float roundings around fused double intermediates and reciprocals, 3 = 2 with sin/cos
                             // evaluated without range reduction while |theta| <= pi/4 (the default), 4 float throughout (csrc/cartpole.hip)
    "fused_step",            // mrl_step of Hanabi / Cartpole as ONE launch with the in-kernel look-back (episode_scan.hpp) or as
                             // phase 1 + phase 2 launches: 0 the library's choice, 1 one launch where the kernel exists, 2 always two
    "fused_heal_test",       // m > 0: in the single-launch step, workgroups whose index is a multiple of m act as if dispatched late, so
                             // that higher workgroups take the recount path of the healing look-back (tests)
    "inject_scan_timeout",   // 1: the simulator's SCAN_TIMEOUT alarm is raised right after construction (tests of the error path)
    "ablate",                // diagnostic build only: phase ablation mask
    "stamps",                // diagnostic build only: in-kernel time stamps
};
static std::mutex g_debug_mutex;
static std::map<std::string, int64_t> g_debug;

int64_t debug_get(const char *key, int64_t fallback)
{
    std::lock_guard<std::mutex> lock(g_debug_mutex);
    auto it = g_debug.find(key);
    return it == g_debug.end() ? fallback : it->second;
}

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

void bind_device(int gpu_id)
{
    int count = 0;
    hipError_t err = hipGetDeviceCount(&count);
    if (err != hipSuccess || count == 0) {
        set_error("no HIP device available (%s); this engine has no CPU execution mode",
                  err == hipSuccess ? "device count is 0" : hipGetErrorString(err));
        throw HipError{MRL_ERR_DEVICE};
    }
    if (gpu_id < 0 || gpu_id >= count) {
        set_error("gpu_id %d out of range (%d device(s))", gpu_id, count);
        throw HipError{MRL_ERR_INVALID};
    }
    MRL_HIP(hipSetDevice(gpu_id));
    hipDeviceProp_t prop;
    MRL_HIP(hipGetDeviceProperties(&prop, gpu_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libmrl_envs.so carries gfx950 (MI355X) code objects only", gpu_id,
                  prop.gcnArchName);
        throw HipError{MRL_ERR_DEVICE};
    }
}

template <typename Fn> static int guarded(Fn &&fn)
{
    try {
        g_error[0] = '\0';
        fn();
        return MRL_OK;
    } catch (const HipError &e) {
        return e.code;
    } catch (const std::exception &e) {
        set_error("%s", e.what());
        return MRL_ERR_INVALID;
    }
}

static int need(const mrl_sim *sim)
{
    if (!sim) {
        set_error("null simulator handle");
        return MRL_ERR_INVALID;
    }
    return MRL_OK;
}

// Entry of every call that advances the simulation: a bounded in-kernel wait that expired in an
// earlier call (SCAN_TIMEOUT, episode_scan.hpp) left the episode numbering unspecified -- refuse to go on.
static int need_healthy(const mrl_sim *sim)
{
    if (int rc = need(sim)) return rc;
    if (sim->scan_timed_out()) {
        set_error("an in-kernel wait of an earlier step expired (SCAN_TIMEOUT): episode numbers of this simulator are "
                  "unspecified from that step on; destroy it and create a new one");
        return MRL_ERR_DEVICE;
    }
    return MRL_OK;
}

// Hanabi, Cartpole and the balance beam keep launch-to-launch state on the HOST by default (which half of the
// double-buffered episode counter is current, the epoch tag of the single-launch step): their launches bake it into kernel
// arguments, so a captured launch replayed later would run with stale values (repeated episode seeds, a look-back that
// accepts the previous replay's counts).  They refuse to be captured until mrl_prepare_graph_capture has moved that state
// into device memory, where every step advances it itself.  Overcooked and Simplecooked have no such state.
static int need_not_capturing(const mrl_sim *sim, void *hip_stream, const char *what)
{
    if (sim->capturable() || !hip_stream) return MRL_OK;
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)hip_stream, &status) != hipSuccess) {
        (void)hipGetLastError();
        return MRL_OK;
    }
    if (status == hipStreamCaptureStatusNone) return MRL_OK;
    set_error("%s: this game's launches carry host-side episode-counter state and cannot be captured in a HIP graph as they are; "
              "call mrl_prepare_graph_capture on the simulator first (outside the capture)", what);
    return MRL_ERR_INVALID;
}

// ---- roofline.peak_measured of bench.py: float4 streams over caller buffers ----
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int kMode>
__global__ void __launch_bounds__(256) mrl_probe_stream_kernel(f32x4 *__restrict__ dst, const f32x4 *__restrict__ src, size_t chunks)
{
    // every workgroup streams through ONE contiguous range (like a wave of the step kernels writing its group's
    // slab): 4 x 4 KB in flight per workgroup and trip
    const size_t per_block = (chunks + gridDim.x - 1) / gridDim.x;
    const size_t first = (size_t)blockIdx.x * per_block, last = first + per_block < chunks ? first + per_block : chunks;
    const f32x4 fill = {1.f, 2.f, 3.f, 4.f};
    size_t i = first + threadIdx.x;
    for (; i + 3 * 256 < last; i += 4 * 256) {
        f32x4 a = fill, b = fill, c = fill, d = fill;
        if (kMode == 0) {
            a = src[i];
            b = src[i + 256];
            c = src[i + 512];
            d = src[i + 768];
        }
        if (kMode == 2) {  // write-through, like the observation stores of the step kernels
            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst + i), "v"(a) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst + i + 256), "v"(b) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst + i + 512), "v"(c) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst + i + 768), "v"(d) : "memory");
        } else {
            dst[i] = a;
            dst[i + 256] = b;
            dst[i + 512] = c;
            dst[i + 768] = d;
        }
    }
    for (; i < last; i += 256) dst[i] = kMode == 0 ? src[i] : fill;
}

}  // namespace mrl

using mrl::guarded;

extern "C" {

int mrl_abi_version(void) { return MRL_ABI_VERSION; }

#ifndef MRL_SOURCE_HASH
#error "build through csrc/Makefile: it passes -DMRL_SOURCE_HASH (the hash of the sources, see the Makefile)"
#endif
// "MRL_SOURCE_HASH=" in front so that the value can also be found in the file without loading it (_lib.embedded_hash)
static const char g_build_hash[] = "MRL_SOURCE_HASH=" MRL_SOURCE_HASH;
const char *mrl_build_hash(void) { return g_build_hash + sizeof("MRL_SOURCE_HASH=") - 1; }
const char *mrl_last_error(void) { return mrl::g_error; }

int mrl_overcooked_create(const mrl_overcooked_config *cfg, int gpu_id, uint32_t num_worlds, mrl_sim **out)
{
    if (!out) return MRL_ERR_INVALID;
    *out = nullptr;
    mrl::DeviceGuard on(gpu_id);  // the caller's current device is restored on return
    return guarded([&] { *out = mrl::create_overcooked(cfg, gpu_id, num_worlds); });
}

int mrl_simplecooked_create(const mrl_overcooked_config *cfg, int gpu_id, uint32_t num_worlds, mrl_sim **out)
{
    if (!out) return MRL_ERR_INVALID;
    *out = nullptr;
    mrl::DeviceGuard on(gpu_id);  // the caller's current device is restored on return
    return guarded([&] { *out = mrl::create_simplecooked(cfg, gpu_id, num_worlds); });
}

int mrl_hanabi_create(const mrl_hanabi_config *cfg, int gpu_id, uint32_t num_worlds, mrl_sim **out)
{
    if (!out) return MRL_ERR_INVALID;
    *out = nullptr;
    mrl::DeviceGuard on(gpu_id);  // the caller's current device is restored on return
    return guarded([&] { *out = mrl::create_hanabi(cfg, gpu_id, num_worlds); });
}

int mrl_cartpole_create(int gpu_id, uint32_t num_worlds, mrl_sim **out)
{
    if (!out) return MRL_ERR_INVALID;
    *out = nullptr;
    mrl::DeviceGuard on(gpu_id);  // the caller's current device is restored on return
    return guarded([&] { *out = mrl::create_cartpole(gpu_id, num_worlds); });
}

int mrl_balance_create(int gpu_id, uint32_t num_worlds, mrl_sim **out)
{
    if (!out) return MRL_ERR_INVALID;
    *out = nullptr;
    mrl::DeviceGuard on(gpu_id);  // the caller's current device is restored on return
    return guarded([&] { *out = mrl::create_balance(gpu_id, num_worlds); });
}

int mrl_step(mrl_sim *sim, void *hip_stream)
{
    if (int rc = mrl::need_healthy(sim)) return rc;
    if (int rc = mrl::need_not_capturing(sim, hip_stream, "mrl_step")) return rc;
    mrl::DeviceGuard on(sim->device);
    return guarded([&] { sim->step(nullptr, (hipStream_t)hip_stream); });
}

int mrl_step_with_actions(mrl_sim *sim, const int32_t *actions_dev, void *hip_stream)
{
    if (int rc = mrl::need_healthy(sim)) return rc;
    if (int rc = mrl::need_not_capturing(sim, hip_stream, "mrl_step_with_actions")) return rc;
    mrl::DeviceGuard on(sim->device);
    return guarded([&] { sim->step(actions_dev, (hipStream_t)hip_stream); });
}

int mrl_step_many(mrl_sim *const *sims, uint32_t count, const int32_t *const *actions_dev_or_null, void *hip_stream)
{
    if (!sims && count) {
        mrl::set_error("mrl_step_many: null simulator list");
        return MRL_ERR_INVALID;
    }
    for (uint32_t k = 0; k < count; k++)
        if (int rc = mrl::need_healthy(sims[k])) return rc;
    if (count == 0) return MRL_OK;
    mrl::DeviceGuard on(sims[0]->device);
    return guarded([&] { mrl::step_many_overcooked(sims, count, actions_dev_or_null, (hipStream_t)hip_stream); });
}

int mrl_step_with_actions_i64(mrl_sim *sim, const int64_t *actions_dev, void *hip_stream)
{
    if (int rc = mrl::need_healthy(sim)) return rc;
    if (!actions_dev) {
        mrl::set_error("mrl_step_with_actions_i64: null action array");
        return MRL_ERR_INVALID;
    }
    mrl::DeviceGuard on(sim->device);
    int rc = MRL_OK;
    const int g = guarded([&] {
        if (!sim->step_i64(reinterpret_cast<const long long *>(actions_dev), (hipStream_t)hip_stream)) {
            mrl::set_error("mrl_step_with_actions_i64: not available for game %d; convert to int32 and use mrl_step_with_actions", sim->game);
            rc = MRL_ERR_INVALID;
        }
    });
    return g != MRL_OK ? g : rc;
}

int mrl_step_phase1(mrl_sim *sim, const int32_t *actions_dev_or_null, void *hip_stream)
{
    if (int rc = mrl::need_healthy(sim)) return rc;
    if (int rc = mrl::need_not_capturing(sim, hip_stream, "mrl_step_phase1")) return rc;
    mrl::DeviceGuard on(sim->device);
    return guarded([&] {
        sim->phase1(actions_dev_or_null, (hipStream_t)hip_stream);
        sim->publish_shard_count((hipStream_t)hip_stream);
    });
}

int mrl_step_phase2(mrl_sim *sim, const uint32_t *episode_base_dev, void *hip_stream)
{
    if (int rc = mrl::need_healthy(sim)) return rc;
    if (int rc = mrl::need_not_capturing(sim, hip_stream, "mrl_step_phase2")) return rc;
    mrl::DeviceGuard on(sim->device);
    return guarded([&] { sim->phase2(episode_base_dev, (hipStream_t)hip_stream); });
}

int mrl_step_phase2_gathered(mrl_sim *sim, const uint32_t *counts_dev, uint32_t num_ranks, uint32_t rank, void *hip_stream)
{
    if (int rc = mrl::need_healthy(sim)) return rc;
    if (int rc = mrl::need_not_capturing(sim, hip_stream, "mrl_step_phase2_gathered")) return rc;
    if (!counts_dev || num_ranks == 0 || num_ranks > 1024 || rank >= num_ranks) {
        mrl::set_error("mrl_step_phase2_gathered: need the gathered counts, 1..1024 ranks and rank < num_ranks (got %u of %u)", rank, num_ranks);
        return MRL_ERR_INVALID;
    }
    mrl::DeviceGuard on(sim->device);
    return guarded([&] { sim->phase2_gathered(counts_dev, num_ranks, rank, (hipStream_t)hip_stream); });
}

int mrl_exchange_create(mrl_sim *sim, uint32_t num_ranks, uint32_t rank, uint8_t *ipc_handle_out)
{
    if (int rc = mrl::need(sim)) return rc;
    static_assert(sizeof(hipIpcMemHandle_t) == MRL_IPC_HANDLE_BYTES, "MRL_IPC_HANDLE_BYTES is hipIpcMemHandle_t's size");
    if (!ipc_handle_out || num_ranks == 0 || num_ranks > MRL_MAX_RANKS || rank >= num_ranks) {
        mrl::set_error("mrl_exchange_create: need a handle buffer, 1..%d ranks and rank < num_ranks (got %u of %u)", MRL_MAX_RANKS, rank, num_ranks);
        return MRL_ERR_INVALID;
    }
    if (sim->exchange.mine) {
        mrl::set_error("mrl_exchange_create: this simulator already has a mailbox");
        return MRL_ERR_INVALID;
    }
    mrl::DeviceGuard on(sim->device);
    return guarded([&] {
        mrl::ShardExchange &x = sim->exchange;
        const size_t bytes = sizeof(unsigned long long) * mrl::kMailSlots * MRL_MAX_RANKS;
        void *block = nullptr;
        // fine-grained: the peers' stores must become visible to a kernel of this device that is already running
        if (hipExtMallocWithFlags(&block, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            MRL_HIP(hipMalloc(&block, bytes));
        }
        x.mine = static_cast<unsigned long long *>(block);
        MRL_HIP(hipMemset(x.mine, 0, bytes));
        MRL_HIP(hipDeviceSynchronize());
        hipIpcMemHandle_t handle;
        MRL_HIP(hipIpcGetMemHandle(&handle, x.mine));
        memcpy(ipc_handle_out, &handle, sizeof(handle));
        x.num_ranks = num_ranks;
        x.rank = rank;
        x.step = 0;
    });
}

int mrl_exchange_connect(mrl_sim *sim, const uint8_t *ipc_handles_of_all_ranks)
{
    if (int rc = mrl::need(sim)) return rc;
    mrl::ShardExchange &x = sim->exchange;
    if (!ipc_handles_of_all_ranks || !x.mine || x.connected) {
        mrl::set_error("mrl_exchange_connect: call mrl_exchange_create first, once, and pass the handles of all ranks");
        return MRL_ERR_INVALID;
    }
    mrl::DeviceGuard on(sim->device);
    return guarded([&] {
        for (uint32_t p = 0; p < x.num_ranks; p++) {
            if (p == x.rank) {
                x.peer[p] = x.mine;  // (a process cannot open its own handle)
                continue;
            }
            hipIpcMemHandle_t handle;
            memcpy(&handle, ipc_handles_of_all_ranks + (size_t)p * sizeof(handle), sizeof(handle));
            void *mapped = nullptr;
            MRL_HIP(hipIpcOpenMemHandle(&mapped, handle, hipIpcMemLazyEnablePeerAccess));
            x.peer[p] = static_cast<unsigned long long *>(mapped);
        }
        x.connected = true;
    });
}

int mrl_step_exchanged(mrl_sim *sim, const int32_t *actions_dev_or_null, void *hip_stream)
{
    if (int rc = mrl::need_healthy(sim)) return rc;
    if (int rc = mrl::need_not_capturing(sim, hip_stream, "mrl_step_exchanged")) return rc;
    if (!sim->exchange.connected) {
        mrl::set_error("mrl_step_exchanged: no connected mailbox (mrl_exchange_create, mrl_exchange_connect)");
        return MRL_ERR_INVALID;
    }
    mrl::DeviceGuard on(sim->device);
    return guarded([&] {
        mrl::ShardExchange &x = sim->exchange;
        x.step += 1;  // this step's tag: the words of step k live in slot k % kMailSlots
        x.publishing = true;
        struct Done {
            mrl::ShardExchange &x;
            ~Done() { x.publishing = false; }
        } done{x};
        sim->step_exchanged(actions_dev_or_null, (hipStream_t)hip_stream);
    });
}

int mrl_set_observation_output(mrl_sim *sim, void *obs_dev_or_null, uint64_t bytes)
{
    if (int rc = mrl::need(sim)) return rc;
    const uint64_t want = sim->observation_bytes();
    if (want == 0) {
        mrl::set_error("mrl_set_observation_output: game %d writes no redirectable observation slab (Overcooked and Simplecooked do)", sim->game);
        return MRL_ERR_INVALID;
    }
    if (obs_dev_or_null && bytes != want) {
        mrl::set_error("mrl_set_observation_output: need a device buffer of exactly %llu bytes (N x P x H x W x F int8), got %llu at %p",
                       (unsigned long long)want, (unsigned long long)bytes, obs_dev_or_null);
        return MRL_ERR_INVALID;
    }
    return guarded([&] { sim->set_observation_output(obs_dev_or_null); });
}

int mrl_set_observation_ring(mrl_sim *sim, void *base_dev_or_null, uint64_t slot_stride_bytes, uint32_t num_slots)
{
    if (int rc = mrl::need(sim)) return rc;
    const uint64_t want = sim->observation_bytes();
    if (want == 0) {
        mrl::set_error("mrl_set_observation_ring: game %d writes no redirectable observation slab (Overcooked and Simplecooked do)", sim->game);
        return MRL_ERR_INVALID;
    }
    if (base_dev_or_null && (num_slots == 0 || slot_stride_bytes < want)) {
        mrl::set_error("mrl_set_observation_ring: need at least one slot and a slot stride >= %llu bytes (N x P x H x W x F int8); got stride "
                       "%llu, %u slot(s) at %p",
                       (unsigned long long)want, (unsigned long long)slot_stride_bytes, num_slots, base_dev_or_null);
        return MRL_ERR_INVALID;
    }
    return guarded([&] { sim->set_observation_ring(base_dev_or_null, slot_stride_bytes, num_slots); });
}

int mrl_prepare_graph_capture(mrl_sim *sim, void *hip_stream)
{
    if (int rc = mrl::need_healthy(sim)) return rc;
    mrl::DeviceGuard on(sim->device);
    return guarded([&] { sim->prepare_graph_capture((hipStream_t)hip_stream); });
}

int mrl_set_episode_counter(mrl_sim *sim, uint32_t next_episode, void *hip_stream)
{
    if (int rc = mrl::need(sim)) return rc;
    mrl::DeviceGuard on(sim->device);
    return guarded([&] { sim->set_episode_counter(next_episode, (hipStream_t)hip_stream); });
}

int mrl_reseed_shard(mrl_sim *sim, uint32_t world_offset, uint32_t num_worlds_total, void *hip_stream)
{
    if (int rc = mrl::need(sim)) return rc;
    mrl::DeviceGuard on(sim->device);
    return guarded([&] { sim->reseed_shard(world_offset, num_worlds_total, (hipStream_t)hip_stream); });
}

int mrl_step_sequence(mrl_sim *sim, const int32_t *actions_dev, uint32_t num_steps, void *hip_stream)
{
    if (int rc = mrl::need_healthy(sim)) return rc;
    if (int rc = mrl::need_not_capturing(sim, hip_stream, "mrl_step_sequence")) return rc;
    mrl::DeviceGuard on(sim->device);
    if (!actions_dev && num_steps) {
        mrl::set_error("mrl_step_sequence: null action array");
        return MRL_ERR_INVALID;
    }
    return guarded([&] { sim->step_sequence(actions_dev, num_steps, (hipStream_t)hip_stream); });
}

int mrl_rollout_random(mrl_sim *sim, uint32_t num_steps, uint64_t seed, uint32_t first_step, void *hip_stream)
{
    if (int rc = mrl::need_healthy(sim)) return rc;
    if (int rc = mrl::need_not_capturing(sim, hip_stream, "mrl_rollout_random")) return rc;
    mrl::DeviceGuard on(sim->device);
    return guarded([&] { sim->rollout_random(num_steps, seed, first_step, (hipStream_t)hip_stream); });
}

int mrl_tensor(mrl_sim *sim, int slot, mrl_tensor_desc *out)
{
    if (int rc = mrl::need(sim)) return rc;
    mrl::DeviceGuard on(sim->device);
    if (!out) return MRL_ERR_INVALID;
    int rc = MRL_OK;
    int g = guarded([&] {
        memset(out, 0, sizeof(*out));
        if (!sim->tensor(slot, out)) {
            mrl::set_error("tensor slot %d is not exported by game %d", slot, sim->game);
            rc = MRL_ERR_SLOT;
        }
    });
    return g != MRL_OK ? g : rc;
}

int mrl_game(const mrl_sim *sim) { return sim ? sim->game : 0; }
uint32_t mrl_num_worlds(const mrl_sim *sim) { return sim ? sim->num_worlds : 0; }
const char *mrl_kernel_name(const mrl_sim *sim) { return sim ? sim->kernel_name() : ""; }
const char *mrl_rollout_kernel_name(const mrl_sim *sim) { return sim ? sim->rollout_kernel_name() : ""; }
uint64_t mrl_bytes_per_world_step(const mrl_sim *sim) { return sim ? sim->bytes_per_world_step() : 0; }

void mrl_destroy(mrl_sim *sim)
{
    if (!sim) return;
    mrl::DeviceGuard on(sim->device);
    (void)hipDeviceSynchronize();
    delete sim;
}

int mrl_launch_shape(const mrl_sim *sim, uint32_t out[4])
{
    if (!sim || !out) return MRL_ERR_INVALID;
    sim->launch_shape(out);
    return MRL_OK;
}

int mrl_scan_timed_out(const mrl_sim *sim) { return sim && sim->scan_timed_out() ? 1 : 0; }

int mrl_debug_set(const char *key, int64_t value)
{
    if (!key) {  // forget everything
        std::lock_guard<std::mutex> lock(mrl::g_debug_mutex);
        mrl::g_debug.clear();
        return MRL_OK;
    }
    for (const char *known : mrl::kDebugKeys)
        if (strcmp(known, key) == 0) {
            std::lock_guard<std::mutex> lock(mrl::g_debug_mutex);
            mrl::g_debug[key] = value;
            return MRL_OK;
        }
    mrl::set_error("mrl_debug_set: unknown key '%s'", key);
    return MRL_ERR_INVALID;
}

int mrl_probe_stream(void *dst_dev, const void *src_dev, uint64_t bytes, int mode, int gpu_id, void *hip_stream)
{
    if (!dst_dev || (mode == 0 && !src_dev) || mode < 0 || mode > 2 || bytes < 16 || (bytes & 15u) ||
        (reinterpret_cast<uintptr_t>(dst_dev) & 15u) || (reinterpret_cast<uintptr_t>(src_dev) & 15u)) {
        mrl::set_error("mrl_probe_stream: need 16-byte aligned device buffers, a multiple of 16 bytes and mode 0..2");
        return MRL_ERR_INVALID;
    }
    mrl::DeviceGuard on(gpu_id);
    return guarded([&] {
        const size_t chunks = bytes / 16;
        const unsigned grid = (unsigned)std::min<size_t>((chunks + 1023) / 1024, 256 * 32);
        auto *dst = static_cast<mrl::f32x4 *>(dst_dev);
        auto *src = static_cast<const mrl::f32x4 *>(src_dev);
        hipStream_t stream = (hipStream_t)hip_stream;
        if (mode == 0)
            hipLaunchKernelGGL((mrl::mrl_probe_stream_kernel<0>), dim3(grid), dim3(256), 0, stream, dst, src, chunks);
        else if (mode == 1)
            hipLaunchKernelGGL((mrl::mrl_probe_stream_kernel<1>), dim3(grid), dim3(256), 0, stream, dst, src, chunks);
        else
            hipLaunchKernelGGL((mrl::mrl_probe_stream_kernel<2>), dim3(grid), dim3(256), 0, stream, dst, src, chunks);
        MRL_HIP(hipGetLastError());
    });
}

}  // extern "C"
