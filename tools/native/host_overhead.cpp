// Time per mrl_step_with_actions call issued from C++ (no Python, no ctypes) on a 32-world batch: 3.35 us on MI355X / ROCm 7.2
// against 4.1 us through ctypes (tools/host_overhead.py) -- the binding is 0.7 us of a call, the HIP launch path the rest.
// Build: hipcc -O2 -I include tools/native/host_overhead.cpp -L madrona_rl_envs_playground_amd -lmrl_envs -o gpurun_out/host_overhead_c
// Run:   LD_LIBRARY_PATH=madrona_rl_envs_playground_amd gpurun_out/host_overhead_c
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>

#include "mrl_envs.h"

int main()
{
    // cramped_room (layouts.py): XXPXX / O  XO... the exact cells do not matter for the launch cost; use a legal 5x4 kitchen
    const int64_t terrain[20] = {2, 2, 1, 2, 2, 3, 0, 0, 0, 3, 2, 0, 0, 0, 2, 2, 5, 2, 6, 2};
    const int64_t sx[2] = {1, 3}, sy[2] = {1, 1};
    int64_t values[16], times[16];
    for (int i = 0; i < 16; i++) values[i] = 20, times[i] = 20;
    mrl_overcooked_config cfg{4, 5, 2, 3, 3, 5, 400, terrain, sx, sy, values, times};
    const uint32_t n = 32;
    mrl_sim *sim = nullptr;
    if (mrl_overcooked_create(&cfg, 0, n, &sim) != 0) {
        printf("create failed: %s\n", mrl_last_error());
        return 1;
    }
    int32_t *actions = nullptr;
    if (hipMalloc(&actions, 2 * n * sizeof(int32_t)) != hipSuccess || hipMemset(actions, 0, 2 * n * sizeof(int32_t)) != hipSuccess) return 1;
    for (int i = 0; i < 200; i++) mrl_step_with_actions(sim, actions, nullptr);
    (void)hipDeviceSynchronize();
    const int K = 20000;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < K; i++) mrl_step_with_actions(sim, actions, nullptr);
    const auto t1 = std::chrono::steady_clock::now();
    (void)hipDeviceSynchronize();
    printf("mrl_step_with_actions from C++: %.2f us of host time per call (%u worlds)\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / K, n);
    mrl_destroy(sim);
    return 0;
}
