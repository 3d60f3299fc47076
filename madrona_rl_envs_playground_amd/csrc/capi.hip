// extern "C" entry points of libmrl_envs.so (include/mrl_envs.h).
#include "common.hpp"

#include <cstring>
#include <exception>

namespace mrl {

static thread_local char g_error[512] = "";

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
    // launches go to the simulator's device whatever the caller's current device is
    // (one process may hold simulators on several GPUs)
    int current = -1;
    if (hipGetDevice(&current) == hipSuccess && current != sim->device) (void)hipSetDevice(sim->device);
    return MRL_OK;
}

}  // namespace mrl

using mrl::guarded;

extern "C" {

int mrl_abi_version(void) { return MRL_ABI_VERSION; }
const char *mrl_last_error(void) { return mrl::g_error; }

int mrl_overcooked_create(const mrl_overcooked_config *cfg, int gpu_id, uint32_t num_worlds, mrl_sim **out)
{
    if (!out) return MRL_ERR_INVALID;
    *out = nullptr;
    return guarded([&] { *out = mrl::create_overcooked(cfg, gpu_id, num_worlds); });
}

int mrl_hanabi_create(const mrl_hanabi_config *cfg, int gpu_id, uint32_t num_worlds, mrl_sim **out)
{
    if (!out) return MRL_ERR_INVALID;
    *out = nullptr;
    return guarded([&] { *out = mrl::create_hanabi(cfg, gpu_id, num_worlds); });
}

int mrl_cartpole_create(int gpu_id, uint32_t num_worlds, mrl_sim **out)
{
    if (!out) return MRL_ERR_INVALID;
    *out = nullptr;
    return guarded([&] { *out = mrl::create_cartpole(gpu_id, num_worlds); });
}

int mrl_step(mrl_sim *sim, void *hip_stream)
{
    if (int rc = mrl::need(sim)) return rc;
    return guarded([&] { sim->step(nullptr, (hipStream_t)hip_stream); });
}

int mrl_step_with_actions(mrl_sim *sim, const int32_t *actions_dev, void *hip_stream)
{
    if (int rc = mrl::need(sim)) return rc;
    return guarded([&] { sim->step(actions_dev, (hipStream_t)hip_stream); });
}

int mrl_step_phase1(mrl_sim *sim, const int32_t *actions_dev_or_null, void *hip_stream)
{
    if (int rc = mrl::need(sim)) return rc;
    return guarded([&] { sim->phase1(actions_dev_or_null, (hipStream_t)hip_stream); });
}

int mrl_step_phase2(mrl_sim *sim, const uint32_t *episode_base_dev, void *hip_stream)
{
    if (int rc = mrl::need(sim)) return rc;
    return guarded([&] { sim->phase2(episode_base_dev, (hipStream_t)hip_stream); });
}

int mrl_set_episode_counter(mrl_sim *sim, uint32_t next_episode, void *hip_stream)
{
    if (int rc = mrl::need(sim)) return rc;
    return guarded([&] { sim->set_episode_counter(next_episode, (hipStream_t)hip_stream); });
}

int mrl_reseed_shard(mrl_sim *sim, uint32_t world_offset, uint32_t num_worlds_total, void *hip_stream)
{
    if (int rc = mrl::need(sim)) return rc;
    return guarded([&] { sim->reseed_shard(world_offset, num_worlds_total, (hipStream_t)hip_stream); });
}

int mrl_step_sequence(mrl_sim *sim, const int32_t *actions_dev, uint32_t num_steps, void *hip_stream)
{
    if (int rc = mrl::need(sim)) return rc;
    if (!actions_dev && num_steps) {
        mrl::set_error("mrl_step_sequence: null action array");
        return MRL_ERR_INVALID;
    }
    return guarded([&] { sim->step_sequence(actions_dev, num_steps, (hipStream_t)hip_stream); });
}

int mrl_rollout_random(mrl_sim *sim, uint32_t num_steps, uint64_t seed, uint32_t first_step, void *hip_stream)
{
    if (int rc = mrl::need(sim)) return rc;
    return guarded([&] { sim->rollout_random(num_steps, seed, first_step, (hipStream_t)hip_stream); });
}

int mrl_tensor(mrl_sim *sim, int slot, mrl_tensor_desc *out)
{
    if (int rc = mrl::need(sim)) return rc;
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
uint64_t mrl_bytes_per_world_step(const mrl_sim *sim) { return sim ? sim->bytes_per_world_step() : 0; }

void mrl_destroy(mrl_sim *sim)
{
    if (!sim) return;
    (void)hipSetDevice(sim->device);
    (void)hipDeviceSynchronize();
    delete sim;
}

}  // extern "C"
