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

// The opaque handle of the C ABI.
struct mrl_sim {
    int game = 0;
    int device = 0;
    uint32_t num_worlds = 0;
    mrl::DeviceArena arena;

    virtual ~mrl_sim() {}
    // actions == nullptr -> read the simulator's own ACTION tensor
    virtual void phase1(const int32_t *actions, hipStream_t stream) = 0;
    virtual void phase2(const uint32_t *episode_base_dev, hipStream_t stream) = 0;
    // whole step; games whose step is two launches may override it with a single fused launch
    virtual void step(const int32_t *actions, hipStream_t stream)
    {
        phase1(actions, stream);
        phase2(nullptr, stream);
    }
    // num_steps steps driven by a caller-provided action array of num_steps consecutive ACTION tensors;
    // default: one launch per step (`action_elems` = elements of one ACTION tensor)
    virtual size_t action_elems() const = 0;
    virtual void step_sequence(const int32_t *actions, uint32_t num_steps, hipStream_t stream)
    {
        for (uint32_t k = 0; k < num_steps; k++) step(actions + (size_t)k * action_elems(), stream);
    }
    // num_steps, seed, first_step: the uniform random policy on the device (include/mrl_envs.h)
    virtual void rollout_random(uint32_t, uint64_t, uint32_t, hipStream_t) { throw std::runtime_error("this game has no device-side random-policy rollout"); }
    virtual void set_episode_counter(uint32_t, hipStream_t) {}
    virtual void reseed_shard(uint32_t, uint32_t, hipStream_t) {}
    virtual bool tensor(int slot, mrl_tensor_desc *out) = 0;
    virtual const char *kernel_name() const = 0;
    virtual uint64_t bytes_per_world_step() const = 0;
};

namespace mrl {
// selects gpu_id and checks it is a gfx950 part; throws HipError
void bind_device(int gpu_id);
mrl_sim *create_overcooked(const mrl_overcooked_config *cfg, int gpu_id, uint32_t num_worlds);
mrl_sim *create_hanabi(const mrl_hanabi_config *cfg, int gpu_id, uint32_t num_worlds);
mrl_sim *create_cartpole(int gpu_id, uint32_t num_worlds);
}  // namespace mrl
