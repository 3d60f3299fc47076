// The random policies of the reference's benchmark harnesses, drawn on the device
// (C ABI: mrl_rollout_random in include/mrl_envs.h).
//
//   Overcooked  torch.randint(high=6) per agent         scripts/overcooked_example.py:99-106
//   Cartpole    torch.randint(high=2)                    scripts/cartpole_example.py:53-87
//   Hanabi      argmax(rand * mask): a uniformly random  scripts/hanabi_example.py:53-82
//               LEGAL move of the player to move
//
// All three come from one counter-based hash of (seed, step index, world, player), so any step of
// a stream can be recomputed on the host (tests replay it through the ordinary step):
//   Overcooked  action = (h * 6) >> 32
//   Cartpole    action = h >> 31
//   Hanabi      action = position of the k-th set bit of the mover's 20-bit legal-move mask,
//               k = (h * popcount(mask)) >> 32            (player = the mover)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace mrl {

__host__ __device__ __forceinline__ uint32_t policy_hash(uint64_t seed, uint32_t step, uint32_t world, uint32_t player)
{
    uint32_t h = (uint32_t)seed ^ (step * 0x9E3779B9u) ^ (world * 0x85EBCA6Bu) ^ ((player + 1u) * 0xC2B2AE35u) ^
                 ((uint32_t)(seed >> 32) * 0x27D4EB2Fu);
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

__host__ __device__ __forceinline__ uint32_t scale(uint32_t h, uint32_t n) { return (uint32_t)(((uint64_t)h * n) >> 32); }

// position of the k-th (0-based) set bit of a mask of at most 32 bits, k < popcount(mask)
__device__ __forceinline__ uint32_t nth_set_bit(uint32_t mask, uint32_t k)
{
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t width = 16; width > 0; width >>= 1) {
        const uint32_t low = mask & ((1u << width) - 1u);
        const uint32_t c = (uint32_t)__popc(low);
        const bool upper = k >= c;
        k -= upper ? c : 0u;
        mask = upper ? mask >> width : low;
        pos += upper ? width : 0u;
    }
    return pos;
}

}  // namespace mrl
