// Device helpers shared by the grid-world kernels (overcooked2 / "Simplecooked" uses them; overcooked.hip
// keeps its own copies of the older ones in its anonymous namespace).  Everything here is about ONE wave:
// in-order LDS hand-offs, pair exchange through DPP, write-through streaming stores, packed byte tables.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace mrl_grid {

constexpr int kWave = 64;

enum : uint32_t { A_NORTH = 0, A_SOUTH = 1, A_EAST = 2, A_WEST = 3, A_STAY = 4, A_INTERACT = 5 };
enum : uint32_t { O_NONE = 0, O_TOMATO, O_ONION, O_DISH, O_SOUP };
constexpr uint32_t kItemNone = 0xFF000000u;  // name NONE, no ingredients, cooking_tick -1
constexpr uint32_t kMaxIngredients = 3;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Cross-lane hand-off through LDS inside ONE wave: DS instructions of a wave execute in issue order, so a
// later ds_read sees an earlier ds_write of another lane without any wait; only the compiler must keep the
// order.  (A wavefront-scope fence also does, but hipcc lowers it with s_waitcnt vmcnt(0).)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)reinterpret_cast<uintptr_t>(p);  // low 32 bits of a shared pointer = LDS offset
}

// Zero-fill of a 256-byte-granular LDS region with ds_write_addtid_b32 (LDS address = M0 + offset + 4 * lane, no
// address register): 256 bytes per instruction at twice the rate of ds_write_b32 and 1.6x that of ds_write_b128
// (MI355X_MICROARCH.md, LDS).  M0 is written and restored inside each statement (hipcc reserves it and does not
// preserve it around asm).  `tile` is wave-uniform.
__device__ __forceinline__ void tile_zero_addtid(uint8_t *tile, uint32_t nbytes)
{
    const uint32_t total = (nbytes + 255u) >> 8;  // 256-byte pieces
    uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(tile));
    uint32_t keep;
    const uint32_t zero = 0;
    uint32_t done = 0;
    for (; done + 8 <= total; done += 8, base += 2048) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
                     "ds_write_addtid_b32 %2\n\t"
                     "ds_write_addtid_b32 %2 offset:256\n\t"
                     "ds_write_addtid_b32 %2 offset:512\n\t"
                     "ds_write_addtid_b32 %2 offset:768\n\t"
                     "ds_write_addtid_b32 %2 offset:1024\n\t"
                     "ds_write_addtid_b32 %2 offset:1280\n\t"
                     "ds_write_addtid_b32 %2 offset:1536\n\t"
                     "ds_write_addtid_b32 %2 offset:1792\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(base), "v"(zero)
                     : "memory");
    }
    for (; done < total; done += 1, base += 256) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
                     "ds_write_addtid_b32 %2\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(base), "v"(zero)
                     : "memory");
    }
}

// Value of the neighbouring lane of a pair (lanes 2k and 2k+1 swap): DPP quad_perm [1,0,3,2], no LDS.
// Call with all lanes enabled.
__device__ __forceinline__ uint32_t swap_pair(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}

// 16-byte write-through (sc1) store through a buffer descriptor: the observation slab is written once and
// not read by the kernel; write-through streams it out while the waves still work instead of leaving dirty
// lines for the end-of-kernel write-back (measured in overcooked.hip).  Out-of-range offsets are dropped.
// kPlain: ordinary stores -- for groups whose slab is not whole 128-byte lines, in the multi-step launches and wherever
// the slab is larger than the Infinity Cache (two write-through halves of a line cost a read-modify-write each there;
// the L2 merges ordinary ones: overcooked.hip, stream_store_rsrc).
template <bool kPlain = false>
__device__ __forceinline__ void stream_store_rsrc(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_offset, const uint4 &v)
{
    u32x4 r;
    r.x = v.x;
    r.y = v.y;
    r.z = v.z;
    r.w = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(r, rsrc, (int)byte_offset, 0, kPlain ? 0 : 16);  // aux bit 4 = sc1
}

// Cell-index delta of a move: NORTH -W, SOUTH +W, EAST +1, WEST -1, STAY / INTERACT 0, packed as signed
// bytes (byte k = direction k): one 64-bit shift instead of a compare ladder (which hipcc lowers to
// exec-masked branch trees on divergent lanes).
__device__ __forceinline__ int32_t step_of(uint32_t dir, uint64_t deltas)
{
    return (int32_t)(int8_t)(deltas >> (8u * dir));
}
__host__ __device__ __forceinline__ uint64_t pack_deltas(int64_t width)
{
    return (uint64_t)(uint8_t)(int8_t)(-width) | ((uint64_t)(uint8_t)(int8_t)width << 8) | (1ull << 16) | (0xFFull << 24);
}

// items are packed name | onions << 8 | tomatoes << 16 | tick << 24
__device__ __forceinline__ uint32_t recipe_of(uint32_t item)
{
    return ((kMaxIngredients + 1) * ((item >> 8) & 0xFF) + ((item >> 16) & 0xFF)) & 15u;
}
__device__ __forceinline__ uint32_t count_of(uint32_t item) { return (((item >> 8) & 0xFF) + ((item >> 16) & 0xFF)) & 0xFF; }

// 16-entry byte table held in four scalar registers
__device__ __forceinline__ uint32_t lookup16(const uint32_t (&w)[4], uint32_t idx)
{
    const uint64_t lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32), hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    const uint64_t half = (idx & 8u) ? hi : lo;
    return (uint32_t)(half >> ((idx & 7u) * 8u)) & 0xFFu;
}

}  // namespace mrl_grid
