// Shared host-side plumbing of libmrl_envs.so (C ABI in include/mrl_envs.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mrl_envs.h"

namespace mrl {

void set_error(const char *fmt, ...);

struct HipError {
    int code;
};

#define MRL_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t err__ = (call);                                                                 \
        if (err__ != hipSuccess) {                                                                 \
            ::mrl::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__,   \
                             __LINE__);                                                            \
            throw ::mrl::HipError{MRL_ERR_DEVICE};                                                 \
        }                                                                                          \
    } while (0)

// Device allocation owned by a simulator; freed in the destructor.
class DeviceArena {
  public:
    ~DeviceArena()
    {
        for (void *p : blocks_) (void)hipFree(p);
    }
    template <typename T> T *alloc(size_t count, bool zero = true)
    {
        void *p = nullptr;
        size_t bytes = count * sizeof(T);
        if (bytes == 0) bytes = sizeof(T);
        MRL_HIP(hipMalloc(&p, bytes));
        blocks_.push_back(p);
        if (zero) MRL_HIP(hipMemset(p, 0, bytes));
        return static_cast<T *>(p);
    }

  private:
    std::vector<void *> blocks_;
};

// Restores the caller's current device when a C-ABI call returns: launches go to the simulator's
// device whatever the caller's current device is (one process may hold simulators on several
// GPUs), and torch's idea of the current device must not change behind its back.
class DeviceGuard {
  public:
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&previous_) != hipSuccess) previous_ = -1;
        if (previous_ != device) (void)hipSetDevice(device);
        else previous_ = -1;  // nothing to restore
    }
    ~DeviceGuard()
    {
        if (previous_ >= 0) (void)hipSetDevice(previous_);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;

  private:
    int previous_ = -1;
};

// Test / measurement knobs set through mrl_debug_set (include/mrl_envs.h) and consulted by the NEXT
// mrl_*_create.  The shipped library reads no environment variable.
int64_t debug_get(const char *key, int64_t fallback);

// Raised by a kernel whose bounded wait expired (episode_scan.hpp): one word in HBM (the exported
// SCAN_TIMEOUT tensor) and one in host-mapped pinned memory, which the host looks at on every
// later call without touching the device.
struct Alarm {
    uint32_t *dev = nullptr;
    uint32_t *host = nullptr;
    __device__ __forceinline__ void raise() const
    {
        *dev = 1u;
        *host = 1u;
    }
};

class AlarmOwner {
  public:
    ~AlarmOwner()
    {
        if (host_) (void)hipHostFree(host_);
    }
    void init(DeviceArena &arena)
    {
        alarm_.dev = arena.alloc<uint32_t>(1);
        MRL_HIP(hipHostMalloc(reinterpret_cast<void **>(&host_), sizeof(uint32_t), hipHostMallocMapped));
        *host_ = 0u;
        void *mapped = nullptr;
        MRL_HIP(hipHostGetDevicePointer(&mapped, host_, 0));
        alarm_.host = static_cast<uint32_t *>(mapped);
    }
    const Alarm &alarm() const { return alarm_; }
    bool raised() const { return host_ && *reinterpret_cast<volatile uint32_t *>(host_) != 0u; }

  private:
    Alarm alarm_{};
    uint32_t *host_ = nullptr;
};

// Launch-to-launch state of the games that number episodes (which half of the double-buffered episode counter is
// current, the look-back's epoch).  By default it lives on the HOST and travels in kernel arguments -- which is why those
// launches cannot be replayed from a HIP graph.  After mrl_prepare_graph_capture it lives HERE, in device memory: every step
// enqueues a one-thread launch that advances it and the step's kernels read it, so a captured sequence replays correctly.
struct LaunchState {
    uint32_t flips;  // steps so far: the current counter is counter[flips & 1]
    uint32_t epoch;  // tag of the single-launch step's status words
};

struct DeviceCounter {
    const LaunchState *state = nullptr;  // nullptr: host mode, the arguments below are used as they are
    uint32_t *counter = nullptr;         // the two halves of the episode counter
    uint32_t external_base = 0;          // the base comes from the caller (sharded phase 2): only `next` and the epoch are derived
    // in device mode, AFTER the step's advance launch: this step reads counter[(flips - 1) & 1] and writes counter[flips & 1]
    __device__ __forceinline__ void apply(const uint32_t *&base, uint32_t *&next, uint32_t &epoch) const
    {
        if (state) {
            const uint32_t flips = state->flips;
            if (!external_base) base = counter + ((flips - 1u) & 1u);
            next = counter + (flips & 1u);
            epoch = state->epoch;
        }
    }
};

inline mrl_tensor_desc make_desc(void *data, int dtype, int device, std::initializer_list<int64_t> shape,
                                 std::initializer_list<int64_t> strides = {})
{
    mrl_tensor_desc d{};
    d.data = data;
    d.dtype = dtype;
    d.device = device;
    d.ndim = (int32_t)shape.size();
    int i = 0;
    for (int64_t s : shape) d.shape[i++] = s;
    if (strides.size() == shape.size()) {
        i = 0;
        for (int64_t s : strides) d.strides[i++] = s;
    } else {
        int64_t run = 1;
        for (int k = d.ndim - 1; k >= 0; k--) {
            d.strides[k] = run;
            run *= d.shape[k];
        }
    }
    return d;
}

}  // namespace mrl

namespace mrl {
// The mailbox of the collective-free shard exchange (mrl_exchange_*; the protocol is in episode_scan.hpp).
struct ShardExchange {
    uint32_t num_ranks = 0, rank = 0;
    uint32_t step = 0;                            // tag of the current step's words
    unsigned long long *mine = nullptr;           // kMailSlots x MRL_MAX_RANKS words, word r of a slot written by rank r
    unsigned long long *peer[MRL_MAX_RANKS] = {}; // rank p's block as mapped here (peer[rank] == mine)
    bool connected = false;
    bool publishing = false;                      // inside mrl_step_exchanged: the count launch also writes the mailboxes
    ~ShardExchange()
    {
        for (uint32_t p = 0; p < num_ranks; p++)
            if (connected && p != rank && peer[p]) (void)hipIpcCloseMemHandle(peer[p]);
        if (mine) (void)hipFree(mine);
    }
};
}  // namespace mrl

// The opaque handle of the C ABI.
struct mrl_sim {
    int game = 0;
    int device = 0;
    uint32_t num_worlds = 0;
    mrl::DeviceArena arena;
    mrl::ShardExchange exchange;

    virtual ~mrl_sim() {}
    // actions == nullptr -> read the simulator's own ACTION tensor
    virtual void phase1(const int32_t *actions, hipStream_t stream) = 0;
    virtual void phase2(const uint32_t *episode_base_dev, hipStream_t stream) = 0;
    // phase 2 of a sharded batch: `counts` holds every rank's SHARD_COUNT of this step (all-gathered on the device);
    // the re-seeding launch sums the lower ranks' itself and advances the simulator's own counter by the sum of all.
    // Games without an episode counter have nothing to do.
    virtual void phase2_gathered(const uint32_t *, uint32_t, uint32_t, hipStream_t) {}
    // mrl_step_phase1 only (mrl_step does not pay for it): the shard's finished worlds of the phase 1 just enqueued -> SHARD_COUNT
    // (and, for mrl_step_exchanged, into the peers' mailboxes: `exchange`)
    virtual void publish_shard_count(hipStream_t) {}
    // phase 2 of mrl_step_exchanged: like phase2_gathered, the counts polled from this rank's mailbox
    virtual void phase2_exchanged(hipStream_t) {}
    // mrl_step_exchanged: three launches by default; a game whose single-launch step does the exchange itself overrides it
    virtual void step_exchanged(const int32_t *actions, hipStream_t stream)
    {
        phase1(actions, stream);
        publish_shard_count(stream);
        phase2_exchanged(stream);
    }
    // whole step; games whose step is two launches may override it with a single fused launch
    virtual void step(const int32_t *actions, hipStream_t stream)
    {
        phase1(actions, stream);
        phase2(nullptr, stream);
    }
    // whole step reading the caller's int64 action tensor (what the reference's harness hands its wrapper); false:
    // this game has no such path and the caller should convert
    virtual bool step_i64(const long long *, hipStream_t) { return false; }
    // num_steps steps driven by a caller-provided action array of num_steps consecutive ACTION tensors;
    // default: one launch per step (`action_elems` = elements of one ACTION tensor)
    virtual size_t action_elems() const = 0;
    virtual void step_sequence(const int32_t *actions, uint32_t num_steps, hipStream_t stream)
    {
        for (uint32_t k = 0; k < num_steps; k++) step(actions + (size_t)k * action_elems(), stream);
    }
    // num_steps, seed, first_step: the uniform random policy on the device (include/mrl_envs.h)
    virtual void rollout_random(uint32_t, uint64_t, uint32_t, hipStream_t) { throw std::runtime_error("this game has no device-side random-policy rollout"); }
    // mrl_set_observation_output: later steps write their observation slab to `out` (nullptr: the simulator's own buffer
    // again).  Returns the slab's size in bytes, 0 if the game's observation cannot be redirected.
    virtual uint64_t set_observation_output(void *) { return 0; }
    virtual uint64_t observation_bytes() const { return 0; }
    // mrl_set_observation_ring: step number k after this call writes its slab to base + (k % num_slots) * stride_bytes
    virtual void set_observation_ring(void *, uint64_t, uint32_t) {}
    // mrl_prepare_graph_capture: from now on the launch-to-launch state lives in device memory (see LaunchState); games
    // without such state have nothing to do.  capturable(): may this simulator's launches be captured right now?
    virtual void prepare_graph_capture(hipStream_t) {}
    virtual bool capturable() const { return true; }
    virtual void set_episode_counter(uint32_t, hipStream_t) {}
    virtual void reseed_shard(uint32_t, uint32_t, hipStream_t) {}
    virtual bool tensor(int slot, mrl_tensor_desc *out) = 0;
    virtual const char *kernel_name() const = 0;
    virtual const char *rollout_kernel_name() const { return kernel_name(); }
    virtual uint64_t bytes_per_world_step() const = 0;
    // a bounded in-kernel wait expired in an earlier call: episode numbers are unspecified from there on
    virtual bool scan_timed_out() const { return false; }
    // launch shape of the step kernel: workgroups, threads per workgroup, LDS bytes per workgroup, worlds per wave
    virtual void launch_shape(uint32_t out[4]) const { out[0] = out[1] = out[2] = out[3] = 0; }
};

namespace mrl {
// selects gpu_id and checks it is a gfx950 part; throws HipError
void bind_device(int gpu_id);
mrl_sim *create_overcooked(const mrl_overcooked_config *cfg, int gpu_id, uint32_t num_worlds);
mrl_sim *create_simplecooked(const mrl_overcooked_config *cfg, int gpu_id, uint32_t num_worlds);
mrl_sim *create_hanabi(const mrl_hanabi_config *cfg, int gpu_id, uint32_t num_worlds);
mrl_sim *create_cartpole(int gpu_id, uint32_t num_worlds);
mrl_sim *create_balance(int gpu_id, uint32_t num_worlds);
// several Overcooked simulators' steps as one launch (mrl_step_many); actions_or_null[k] == nullptr: simulator k's ACTION tensor
void step_many_overcooked(mrl_sim *const *sims, uint32_t count, const int32_t *const *actions_or_null, hipStream_t stream);
}  // namespace mrl
