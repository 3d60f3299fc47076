// Overcooked world step for gfx950: the whole step fused in one kernel (action
// apply -> interactions -> movement/collisions -> pot ticks -> horizon reset ->
// reward/done -> observation encode), one wavefront per GROUP of worlds.
//
// Semantics follow the reference task graph
//   /root/reference/src/overcooked_env/sim.cpp:498-537
// (systems :199-495, observation rows :68-167, init :556-659) with the component
// widths of sim.hpp:59-184.  Nothing of Madrona's ECS/taskgraph is reproduced:
// the 20 graph nodes and their per-cell scratch (past/current/future_player,
// interacting_players) collapse into
//   - a loop over the interacting players in ascending id, which is what the
//     reference's four rank phases serialise to (sim.cpp:259-358),
//   - a pairwise proposal test for the all-or-nothing collision rule
//     (sim.cpp:363-426): same target or swapped cells => nobody moves,
//   - a from-scratch observation encode (the reference updates rows in place
//     and clears the player channels through past_player; every vacated cell is
//     cleared, so the row is a pure function of the state).
//
// Mapping (profiles/r01_a_*: the first version, one wave per world, moved exactly
// the algorithmic 41 MB per launch but spent ~540 instructions per world, mostly
// wave-uniform transition logic running on 64 lanes for one world, and was
// issue/latency bound at 21 us).  Now a wave owns `wpw` consecutive worlds:
//   load     all lanes copy the group's state slab HBM -> LDS (contiguous, coalesced); while the
//            loads are in flight the observation tile in LDS is filled with the group's static
//            BACKGROUND (terrain one-hot bytes, everything else zero), an image built on the host
//   step     lane = (world, player): every player's interaction, move proposal and collision test
//            at once; same-cell interactions in ascending player id by rank, like the reference's
//            four do_counter_pot_int phases; pairs exchange through DPP (P = 2) or LDS (any P)
//   observe  only DYNAMIC cells (an object on them or a player standing there; typically a few per
//            world) are compacted into a list with wave ballots, and one lane per (dynamic cell,
//            viewer) patches that row of the tile; the tile is streamed out with 16-byte
//            write-through stores (the group's rows are one contiguous range of HBM)
//   store    state slab, rewards, flags
// so the transition costs one pass of straight-line code per group whatever the player count, the
// encode touches what changed and the stream-out runs with all lanes busy.  wpw is picked on the
// host so that a launch still has thousands of waves (wpw = 1 for very large layouts), wpw*P <= 64.
// Layouts whose group slab does not fit one LDS tile keep the multi-pass row assembly below.
//
// HBM layout, all world-major so a group's loads and
// stores are contiguous and a shard of worlds is one contiguous slab:
//   cell_obj [N][C]  u32  name | onions<<8 | tomatoes<<16 | tick<<24
//   players  [N][P]  2xu32 {pos | orientation<<8, held item (same packing)}
//   timestep [N]     i32
//   action   [P][N]  i32  (the reference's exported shape, mgr.cpp:214-218)
//   reward   [P][N]  i32, done [N] i32
//   obs      [N][P][C][F] u8, F = 5P+16: one contiguous P*C*F block per world
#include "common.hpp"
#include "random_policy.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

constexpr int kWave = 64;
#ifndef MRL_OVERCOOKED_WPB
#define MRL_OVERCOOKED_WPB 4
#endif
constexpr int kWavesPerBlock = MRL_OVERCOOKED_WPB;
constexpr int kBlock = kWave * kWavesPerBlock;
constexpr int kRowsPerPass = 128;  // observation rows assembled per LDS tile

enum : uint32_t { A_NORTH = 0, A_SOUTH = 1, A_EAST = 2, A_WEST = 3, A_STAY = 4, A_INTERACT = 5 };
enum : uint32_t { T_AIR = 0, T_POT, T_COUNTER, T_ONION_SRC, T_TOMATO_SRC, T_DISH_SRC, T_SERVING };
enum : uint32_t { O_NONE = 0, O_TOMATO, O_ONION, O_DISH, O_SOUP };
constexpr uint32_t kItemNone = 0xFF000000u;  // name NONE, tick -1 (sim.hpp:59-64)
constexpr uint32_t kMaxIngredients = 3;

// layout of the constant block (copied into LDS by every workgroup)
constexpr uint32_t kConstTerrain = 0;    // 256 bytes
constexpr uint32_t kConstTimes = 256;    // 16 bytes
constexpr uint32_t kConstValues = 272;   // 16 bytes
constexpr uint32_t kConstStart = 288;    // 64 bytes: start cell of each player
constexpr uint32_t kConstPots = 352;     // 256 bytes: cells holding a pot
constexpr uint32_t kConstBytes = 608;

struct StepParams {
    uint32_t num_worlds;
    uint32_t P, C, W, F;
    uint64_t deltas;       // move deltas as signed bytes, see step_of
    uint32_t rows;         // P*C
    uint32_t block_bytes;  // P*C*F
    // x / d == umulhi(x, floor(2^32/d)+1) while x*d < 2^32 (checked on the host)
    uint32_t inv_c, inv_p, inv_rows;
    uint32_t placement_rew, soup_pickup_rew;
    uint32_t times_w[4], values_w[4];  // the 16-entry recipe tables, 4 bytes per word (scalar registers)
    uint32_t pots_w;       // cells of the first four pots, one byte each
    uint32_t wpw;          // worlds per wave
    uint32_t per_xcd;      // workgroups of the launch / 8 (set per launch: the XCD-aware block -> world mapping)
    uint32_t wpp;          // whole worlds per observation pass (0: a world spans several passes)
    uint32_t num_pots;
#ifdef MRL_DIAG
    uint32_t ablate;             // diagnostic build only: phase ablation mask
    unsigned long long *stamps;  // diagnostic build only: per-wave s_memtime stamps
#endif
    uint32_t whole;        // a group's whole observation slab fits one LDS tile: single-pass encode
    uint32_t share;        // the waves of a workgroup share ONE world (few worlds x very large observations): each
                           // steps it redundantly, takes every kWavesPerBlock-th pass of its rows; wave 0 stores the state
    uint32_t team;         // share, with ONE copy of the world's state in LDS for the workgroup (team_body); off_tile / lds_wave_stride
                           // then place the four waves' row tiles behind that copy
    uint32_t steady;       // all passes of a wave share one alignment and cover whole worlds: zero-fill once
    uint32_t tail_even;    // rows' 16-byte tails are 2-byte aligned in the LDS tile (P even)
    uint32_t private_consts;  // <= 4 players and <= 4 pots: start cells and pots travel in kernel arguments, the terrain
                              // in a per-wave LDS copy -> no workgroup barrier in the kernel (the waves never meet)
    uint32_t starts_w;        // start cells of the first four players, one byte each
    uint32_t off_terr;        // per-wave terrain copy (private_consts)
    uint32_t store_policy; // multi-pass stream-out: 0 sc1 write-through, 1 plain, 2 nt (chosen by slab size; mrl_debug_set overcooked.store_policy)
    uint32_t patch;        // single-pass encode that only touches dynamic cells (zero-filled tile + terrain bytes + patches)
    uint32_t unaligned;    // ... whose group slabs do not all start on 16-byte boundaries (one or two big worlds per wave): the tile
                           // is kept as misaligned as the slab, so 16-byte chunks line up; a few head bytes go out one by one
    const uint16_t *terr_off;  // device, [terr_entries]: per row of a GROUP, tile offset of its terrain one-hot byte, 0 = none
    uint32_t terr_entries;     // wpw * rows
    // p.direct (see patch_direct): the encode needs no search for what is dynamic.  Players only ever stand on AIR cells and
    // objects only ever lie on counters / pots a player can face (HOLDER cells); the host checks the start cells and builds
    // the table of a group's holder cells: entry [k * 64 + lane], lanes below wpw * P of round 0 left free for the players
    uint32_t direct;
    const uint32_t *hold_tab;  // device, [hold_entries]: tile offset of the cell's viewer-0 row | cell index in the group << 16 | 1 << 30 | is_pot << 31
    uint32_t hold_entries;
    uint32_t off_pl, off_x, off_sum, off_cur, off_flags, off_list, off_tail, off_tile;  // byte offsets inside a wave's LDS region
    uint32_t lds_wave_stride;
    int64_t horizon;
    const uint32_t *consts;  // kConstBytes, device
    uint32_t *cell_obj;
    uint2 *players;
    int32_t *timestep;
    const int32_t *actions;
    const long long *actions64;  // mrl_step_with_actions_i64: the caller's int64 action tensor, read directly (NULL otherwise)
    int32_t *action_mirror;      // ... and mirrored into the ACTION tensor, which the reference's wrapper fills itself
    int32_t *reward;
    int32_t *done;
    uint8_t *obs;
    // mrl_set_observation_ring (multi-step launches): step k of the launch writes slot (ring_first + k) % ring_slots,
    // ring_stride bytes apart from `obs`; ring_slots <= 1: every step writes `obs`
    uint64_t ring_stride;
    uint32_t ring_slots, ring_first;
    uint32_t pair_exchange;  // mrl_step_many: two players and the pair-exchange (DPP) transition is this simulator's (not forced generic)
};

__device__ __forceinline__ void wave_lds_sync()
{
    // Cross-lane hand-off through LDS inside ONE wave: DS instructions of a wave are executed
    // in issue order, so a later ds_read sees an earlier ds_write of another lane without any
    // wait.  All that is needed is that the compiler keeps the order -- a compiler-only
    // barrier.  (A wavefront-scope release/acquire fence also does that, but hipcc lowers it
    // with s_waitcnt vmcnt(0): the wave then sits out the full latency of its state stores
    // before it starts the observation passes.)
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// Storing a row's 16 viewer-independent bytes into the LDS tile.  Rows are F = 5P+16 bytes
// apart, so these 16 bytes are in general not 16-byte aligned, and a DS store off its natural
// alignment is replayed (cdna_hip_programming.md G17; measured here: one misaligned b128 per
// row cost 2.8 us of the step, four misaligned b32 8.5 us).  So the store is split into
// naturally aligned pieces chosen by the (wave-uniform) alignment class of the address.
// Written as asm so the compiler cannot fuse the pieces back into one misaligned b128.  LDS
// operations of a wave execute in order, so later ds_reads of the same wave see these stores.
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)reinterpret_cast<uintptr_t>(p);  // low 32 bits of a shared pointer = LDS offset
}

__device__ __forceinline__ void lds_store_tail_a4(uint8_t *ptr, const uint4 &t)  // address % 4 == 0
{
    asm volatile("ds_write_b32 %0, %1\n\t"
                 "ds_write_b32 %0, %2 offset:4\n\t"
                 "ds_write_b32 %0, %3 offset:8\n\t"
                 "ds_write_b32 %0, %4 offset:12"
                 :
                 : "v"(lds_addr(ptr)), "v"(t.x), "v"(t.y), "v"(t.z), "v"(t.w)
                 : "memory");
}

__device__ __forceinline__ void lds_store_tail_a2(uint8_t *ptr, const uint4 &t)  // address % 4 == 2
{
    const uint32_t m1 = __builtin_amdgcn_alignbit(t.y, t.x, 16);
    const uint32_t m2 = __builtin_amdgcn_alignbit(t.z, t.y, 16);
    const uint32_t m3 = __builtin_amdgcn_alignbit(t.w, t.z, 16);
    asm volatile("ds_write_b16 %0, %1\n\t"
                 "ds_write_b32 %0, %2 offset:2\n\t"
                 "ds_write_b32 %0, %3 offset:6\n\t"
                 "ds_write_b32 %0, %4 offset:10\n\t"
                 "ds_write_b16_d16_hi %0, %5 offset:14"
                 :
                 : "v"(lds_addr(ptr)), "v"(t.x), "v"(m1), "v"(m2), "v"(m3), "v"(t.w)
                 : "memory");
}

// Both even classes without a branch: every lane issues the same five naturally aligned stores
// (three dwords at the next 4-byte boundaries, two halfwords at the ends), with data and
// addresses selected per lane.  In the single-pass encode the class follows the lane's parity,
// so the branchy version executes both classes in every iteration.
__device__ __forceinline__ void lds_store_tail_even(uint8_t *ptr, const uint4 &t)  // address % 2 == 0
{
    const uint32_t addr = lds_addr(ptr);
    const bool odd2 = (addr & 2u) != 0;       // address % 4 == 2
    const uint32_t base4 = addr & ~3u;
    const uint32_t m1 = __builtin_amdgcn_alignbit(t.y, t.x, 16);
    const uint32_t m2 = __builtin_amdgcn_alignbit(t.z, t.y, 16);
    const uint32_t m3 = __builtin_amdgcn_alignbit(t.w, t.z, 16);
    const uint32_t d1 = odd2 ? m1 : t.y, d2 = odd2 ? m2 : t.z, d3 = odd2 ? m3 : t.w;
    // halfword A: aligned: t.x low at base4;  shifted: t.x low at base4 + 2
    // halfword B: aligned: t.x high at base4 + 2;  shifted: t.w high at base4 + 16
    const uint32_t a_addr = odd2 ? base4 + 2u : base4;
    const uint32_t b_addr = odd2 ? base4 + 16u : base4 + 2u;
    const uint32_t b_data = odd2 ? t.w : t.x;
    asm volatile("ds_write_b32 %0, %1 offset:4\n\t"
                 "ds_write_b32 %0, %2 offset:8\n\t"
                 "ds_write_b32 %0, %3 offset:12\n\t"
                 "ds_write_b16 %4, %5\n\t"
                 "ds_write_b16_d16_hi %6, %7"
                 :
                 : "v"(base4), "v"(d1), "v"(d2), "v"(d3), "v"(a_addr), "v"(t.x), "v"(b_addr), "v"(b_data)
                 : "memory");
}

__device__ __forceinline__ void lds_store_tail(uint8_t *ptr, const uint4 &t, uint32_t align_class)
{
    if (align_class == 0) {
        lds_store_tail_a4(ptr, t);
    } else if (align_class == 2) {
        lds_store_tail_a2(ptr, t);
    } else {  // odd address: bytes
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ptr[k] = (uint8_t)(t.x >> (8 * k));
            ptr[4 + k] = (uint8_t)(t.y >> (8 * k));
            ptr[8 + k] = (uint8_t)(t.z >> (8 * k));
            ptr[12 + k] = (uint8_t)(t.w >> (8 * k));
        }
    }
}

// 16-byte store of observation bytes.  The observation slab (34 MB per launch at 32768
// worlds) is written once and not read again by this kernel, and it is larger than the L2s:
// with plain stores the dirty lines pile up in L2 and are written back in the end-of-kernel
// release, which the next launch waits for.  sc1 (write-through, line not kept) streams them
// out while the waves are still working.  Measured on MI355X, us per launch at 32768 worlds:
// plain 14.05, nt 13.44, sc1 11.78, sc0 sc1 11.79.  MRL_STORE_POLICY (0 plain, 1 nt, 2 sc1,
// 3 sc0 sc1) exists to re-measure.
#ifndef MRL_STORE_POLICY
#define MRL_STORE_POLICY 2
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_store(uint4 *dst, const uint4 &v)
{
#if MRL_STORE_POLICY == 0
    *dst = v;
#else
    u32x4 r;
    r.x = v.x;
    r.y = v.y;
    r.z = v.z;
    r.w = v.w;
#if MRL_STORE_POLICY == 1
    asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(dst), "v"(r) : "memory");
#elif MRL_STORE_POLICY == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(r) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(r) : "memory");
#endif
#endif
}

__device__ __forceinline__ void nt_store(uint4 *dst, const uint4 &v)
{
    u32x4 r;
    r.x = v.x;
    r.y = v.y;
    r.z = v.z;
    r.w = v.w;
    __builtin_nontemporal_store(r, reinterpret_cast<u32x4 *>(dst));
}

// Cell-index delta of a move (sim.cpp:185-197): NORTH -W, SOUTH +W, EAST +1, WEST -1, STAY and
// INTERACT 0.  `deltas` packs them as signed bytes (|W| <= 85 since H >= 3 and H*W <= 255),
// byte k = delta of direction k; one 64-bit shift instead of a compare ladder -- hipcc lowers
// such ladders to exec-masked branch trees, which cost ~20 instructions each on divergent lanes.
__device__ __forceinline__ int32_t step_of(uint32_t dir, uint64_t deltas)
{
    return (int32_t)(int8_t)(deltas >> (8u * dir));
}

__device__ __forceinline__ uint32_t recipe_of(uint32_t item)
{
    return ((kMaxIngredients + 1) * ((item >> 8) & 0xFF) + ((item >> 16) & 0xFF)) & 15u;
}

__device__ __forceinline__ uint32_t count_of(uint32_t item) { return (((item >> 8) & 0xFF) + ((item >> 16) & 0xFF)) & 0xFF; }

// The counter/pot/source/serving interaction of one player (sim.cpp:208-358), shared by the
// generic and the register-resident transition.  `there` is the object on the faced cell
// (only meaningful for counters and pots), `need` its cooking time and `value` the delivery
// value of the held soup; returns the player's new held item.
// 16-entry byte table held in four scalar registers: one select and a 64-bit shift instead of
// an LDS round trip (and no compare ladder, see step_of)
__device__ __forceinline__ uint32_t lookup16(const uint32_t (&w)[4], uint32_t idx)
{
    const uint64_t lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32), hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    const uint64_t half = (idx & 8u) ? hi : lo;
    return (uint32_t)(half >> ((idx & 7u) * 8u)) & 0xFFu;
}

__device__ __forceinline__ uint32_t interact(const StepParams &p, int32_t need, int32_t value, uint32_t terr, uint32_t held,
                                             uint32_t &there, int32_t &reward)
{
    // Straight-line selects instead of the if/else ladder of sim.cpp:229-253,299-335: lanes of
    // one wave are different worlds taking different branches, so a ladder costs the sum of
    // all its arms anyway; this form is shorter than that sum.
    const uint32_t hname = held & 0xFF, oname = there & 0xFF;
    const int32_t tick = (int8_t)(there >> 24);
    const bool counter = terr == T_COUNTER, pot = terr == T_POT;
    const bool empty_handed = hname == O_NONE, nothing_there = oname == O_NONE, soup_there = oname == O_SOUP;
    // counter: put down / pick up
    const bool put = counter & !empty_handed & nothing_there;
    const bool take = counter & empty_handed & !nothing_there;
    // pot: start cooking / plate a ready soup / add an ingredient
    const bool start = pot & empty_handed & soup_there & (tick < 0) & (count_of(there) > 0);
    const bool plate = pot & (hname == O_DISH) & soup_there & (tick >= 0) & (tick >= need);
    const bool ingredient = pot & ((hname == O_ONION) | (hname == O_TOMATO));
    const uint32_t soup = nothing_there ? (O_SOUP | kItemNone) : there;
    const bool add = ingredient & !(((int8_t)(soup >> 24) >= 0) | (count_of(soup) == kMaxIngredients));
    // sources and serving window
    const bool source = (terr == T_ONION_SRC) | (terr == T_TOMATO_SRC) | (terr == T_DISH_SRC);
    // ONION_SRC(3) -> ONION(2), TOMATO_SRC(4) -> TOMATO(1), DISH_SRC(5) -> DISH(3), as a packed table
    const uint32_t fresh = ((0x00030102u >> (((terr - T_ONION_SRC) & 3u) * 8u)) & 0xFFu) | kItemNone;
    const bool grab = source & empty_handed;
    const bool serve = (terr == T_SERVING) & (hname == O_SOUP);

    uint32_t new_there = there;
    new_there = start ? (there & 0x00FFFFFFu) : new_there;
    new_there = ingredient ? (add ? soup + (hname == O_ONION ? 0x100u : 0x10000u) : soup) : new_there;
    new_there = put ? held : new_there;
    new_there = (take | plate) ? kItemNone : new_there;
    uint32_t new_held = held;
    new_held = (take | plate) ? there : new_held;
    new_held = grab ? fresh : new_held;
    new_held = (put | add | serve) ? kItemNone : new_held;
    reward += (plate ? (int32_t)p.soup_pickup_rew : 0) + (add ? (int32_t)p.placement_rew : 0) + (serve ? value : 0);
    there = new_there;
    return new_held;
}

// The 16 viewer-independent bytes of a cell's observation rows, row[5P .. 5P+16): terrain
// one-hot (6), idle pot onions/tomatoes, soup onions/tomatoes, remaining time, ready, dish,
// onion, tomato, urgency (sim.cpp:79-120,151-164; terrain bytes :642-645).  `o` is the object on
// the cell, `h` what the player standing there holds (kItemNone if nobody).  Straight-line selects.
__device__ __forceinline__ uint4 cell_tail(const StepParams &p, uint32_t terr, uint32_t o, uint32_t h, uint32_t urgent)
{
    const int32_t need = (int32_t)lookup16(p.times_w, recipe_of(o));
    const uint32_t oname = o & 0xFF, on = (o >> 8) & 0xFF, tom = (o >> 16) & 0xFF;
    const int32_t tick = (int8_t)(o >> 24);
    const uint32_t hname = h & 0xFF;
    const bool is_soup = oname == O_SOUP, in_pot = terr == T_POT;
    const bool idle = is_soup & in_pot & (tick < 0);
    const bool hot = is_soup & in_pot & (tick >= 0);
    const bool plated = is_soup & !in_pot;
    const bool hsoup = hname == O_SOUP;  // a soup in hand overrides the cell's soup channels
    const uint32_t idle_on = idle ? on : 0u, idle_tom = idle ? tom : 0u;
    const uint32_t soup_on = hsoup ? (h >> 8) & 0xFF : ((hot | plated) ? on : 0u);
    const uint32_t soup_tom = hsoup ? (h >> 16) & 0xFF : ((hot | plated) ? tom : 0u);
    const uint32_t remaining = (hot & !hsoup) ? (uint32_t)(need - tick) & 0xFF : 0u;
    const uint32_t ready = (hsoup | plated | (hot & (tick >= need))) ? 1u : 0u;
    const uint32_t dish = ((oname == O_DISH) | (hname == O_DISH)) ? 1u : 0u;
    const uint32_t onion = ((oname == O_ONION) | (hname == O_ONION)) ? 1u : 0u;
    const uint32_t tomato = ((oname == O_TOMATO) | (hname == O_TOMATO)) ? 1u : 0u;
    const uint32_t tbit = terr == T_AIR ? 0u : 1u;
    const uint32_t tsh = ((terr - 1u) & 3u) * 8u;
    uint4 t;
    t.x = (terr >= 1 && terr <= 4) ? (tbit << tsh) : 0u;
    t.y = ((terr >= 5) ? (tbit << tsh) : 0u) | (idle_on << 16) | (idle_tom << 24);
    t.z = soup_on | (soup_tom << 8) | (remaining << 16) | (ready << 24);
    t.w = dish | (onion << 8) | (tomato << 16) | ((urgent ? 1u : 0u) << 24);
    return t;
}

// Value of the neighbouring lane of a pair (lanes 2k and 2k+1 swap): DPP quad_perm [1,0,3,2], no LDS.
// Called with all lanes enabled.
__device__ __forceinline__ uint32_t swap_pair(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}

// The transition of a whole group at once: lane = (world wl of the group, player q), lane = wl*P + q
// (sim.cpp:199-438).  `posori` = pos | orientation << 8 and `held` are the lane's own player in
// registers; the cells' objects are in LDS (obj_w = the C cells of the lane's world).  kP == 2: pairs exchange through DPP;
// kP == 0: any player count, through the scratch arrays s_x (>= 128 words), s_sum, s_blk (one word per world).
//   interactions  sources and the serving window touch nobody else's state.  Counter / pot
//                 interactions on the same cell happen in ascending player id: a player's RANK is the
//                 number of lower ids facing its cell (setup_interact_time, sim.cpp:259-283) and
//                 round r runs the players of rank r (do_counter_pot_int0..3, sim.cpp:285-358); rounds
//                 nobody takes part in are skipped with one ballot.
//   movement      every lane proposes (sim.cpp:363-379); any two equal proposals or any swapping pair
//                 in a world and nobody in it moves (sim.cpp:383-426).
// Inactive lanes (beyond the group's players) run along with harmless values.  Returns the
// reward of the lane's WORLD (summed over its players) through reward_world.
template <int kP>
__device__ __forceinline__ void transition_lanes(const StepParams &p, const uint8_t *s_terrain, uint32_t *obj_w, uint32_t *s_x,
                                                 uint32_t *s_sum, uint32_t *s_blk, uint32_t P, uint32_t lane, bool active, uint32_t wl,
                                                 uint32_t q, uint32_t a, uint32_t &posori, uint32_t &held, int32_t &reward_world)
{
    const uint32_t pos = posori & 0xFFu, ori = (posori >> 8) & 0xFFu;
    const uint32_t tgt = pos + (uint32_t)step_of(ori, p.deltas);
    const uint32_t terr = s_terrain[tgt];
    const uint32_t ahead = s_terrain[pos + (uint32_t)step_of(a, p.deltas)];  // STAY / INTERACT: own cell (AIR)
    const bool inter = active && a == A_INTERACT;
    const bool touches = inter && (terr == T_COUNTER || terr == T_POT);
    const uint32_t key = touches ? tgt : 0xFFFFu;
    const uint32_t base = wl * P;
    uint32_t rank = 0;
    if constexpr (kP == 2) {
        const uint32_t other = swap_pair(key);
        rank = (q == 1u && touches && other == tgt) ? 1u : 0u;
    } else {
        s_x[lane] = key;
        wave_lds_sync();
        for (uint32_t q2 = 0; q2 + 1 < P; q2++) rank += (q2 < q && touches && s_x[base + q2] == key) ? 1u : 0u;
        wave_lds_sync();
    }
    int32_t mine = 0;
    const int32_t value = (int32_t)lookup16(p.values_w, recipe_of(held));
    uint32_t *cell = obj_w + (touches ? tgt : 0u);  // obj_w: the cells of the lane's world
    constexpr uint32_t kRounds = kP == 2 ? 2u : 4u;  // at most four players face one cell
    // one copy of the interaction code for all rounds (rounds after the first are rare and the launch is sensitive to
    // the size of its straight-line code: DESIGN.md 4.1)
#pragma clang loop unroll(disable)
    for (uint32_t r = 0; r < kRounds; r++) {
        const bool todo = inter && rank == r;
        if (r > 0 && __ballot(todo) == 0ull) break;
        if (todo) {
            uint32_t there = touches ? *cell : kItemNone;
            const int32_t need = (int32_t)lookup16(p.times_w, recipe_of(there));
            held = interact(p, need, value, terr, held, there, mine);
            if (touches) *cell = there;
        }
        wave_lds_sync();
    }
    // movement proposal; orientation := action unless STAY / INTERACT
    uint32_t prop = pos, pori = ori;
    {
        const bool moves = a != A_INTERACT;
        const uint32_t np = pos + (uint32_t)step_of(a, p.deltas);
        pori = (moves && a != A_STAY) ? a : ori;
        prop = (moves && ahead == T_AIR) ? np : pos;
    }
    bool blocked;
    if constexpr (kP == 2) {
        reward_world = mine + (int32_t)swap_pair((uint32_t)mine);
        const uint32_t o = swap_pair(pos | (prop << 8));
        const uint32_t opos = o & 0xFFu, oprop = o >> 8;
        blocked = (prop == oprop) | ((prop == opos) & (pos == oprop));
    } else {
        s_x[lane] = pos | (prop << 8);
        if (active && q == 0) {
            s_sum[wl] = 0u;
            s_blk[wl] = 0u;
        }
        wave_lds_sync();
        if (mine != 0) atomicAdd(&s_sum[wl], (uint32_t)mine);
        bool hit = false;
        for (uint32_t q2 = 0; q2 < P; q2++) {
            const uint32_t v = s_x[base + q2];
            hit |= (q2 != q) & ((prop == (v >> 8)) | ((prop == (v & 0xFFu)) & (pos == (v >> 8))));
        }
        if (hit && active) s_blk[wl] = 1u;
        wave_lds_sync();
        reward_world = active ? (int32_t)s_sum[wl] : 0;
        blocked = active && s_blk[wl] != 0u;
        wave_lds_sync();
    }
    posori = (blocked ? pos : prop) | (pori << 8);
}

// Pots (sim.cpp:430-438), after the interactions: a pot started in this step is already at 1.  Lane = world.
__device__ __forceinline__ void tick_pots_world(const StepParams &p, const uint8_t *s_pots, uint32_t *obj, bool valid)
{
    if (valid) {
        for (uint32_t k = 0; k < p.num_pots; k++) {
            const uint32_t c = k < 4 ? (p.pots_w >> (8 * k)) & 0xFFu : (uint32_t)s_pots[k];
            const uint32_t o = obj[c];
            const int32_t tick = (int8_t)(o >> 24);
            if ((o & 0xFF) == O_SOUP && tick >= 0 && tick < (int32_t)lookup16(p.times_w, recipe_of(o)))
                obj[c] = (o & 0x00FFFFFFu) | ((uint32_t)(uint8_t)(tick + 1) << 24);
        }
    }
}
__device__ __forceinline__ void tick_pots(const StepParams &p, const uint8_t *s_pots, uint32_t *s_obj, uint32_t nw, uint32_t lane)
{
    tick_pots_world(p, s_pots, s_obj + lane * p.C, lane < nw);
}
// Diagnostics (make diag -> diag/libmrl_envs_diag.so, never the shipped library): in-kernel stamps
// for tools/stamps.py and phase ablation.  In the normal build these expand to nothing.
// kPlain: ordinary stores instead of write-through ones.  Write-through wins while the slab fits the 256 MiB Infinity Cache
// and whenever a group's slab is whole 128-byte lines (cramped_room, counter_circuit); a group slab that is NOT (1300- or
// 2340-byte worlds) ends in a line it shares with the next group's wave, and two write-through partial lines that must go out
// to HBM cost a read-modify-write each: coordination_ring at 262144 worlds 121 us per step write-through, 71 us plain
// (1 M worlds 534 / 305; asymmetric_advantages 210 / 133 and 894 / 570) -- the L2 merges the halves before it writes back.
// The host picks the instantiation (overcooked.whole_store: 0 by slab size and alignment, 1 write-through, 2 plain).
template <bool kPlain = false>
__device__ __forceinline__ void stream_store_rsrc(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_offset, const uint4 &v)
{
    u32x4 r;
    r.x = v.x;
    r.y = v.y;
    r.z = v.z;
    r.w = v.w;
#ifndef MRL_WHOLE_STORE_AUX
#define MRL_WHOLE_STORE_AUX 16
#endif
    __builtin_amdgcn_raw_buffer_store_b128(r, rsrc, (int)byte_offset, 0, kPlain ? 0 : MRL_WHOLE_STORE_AUX);  // aux bit 4 = sc1 (write-through)
}

// Single-pass encode of a group's observation slab (small layouts; see the call site).
template <bool kPlain = false>
__device__ __forceinline__ void observe_whole(const StepParams &p, const uint8_t *s_terrain, const uint32_t *s_obj,
                                              const uint32_t *s_pl, const uint8_t *s_cur, const uint8_t *s_flags,
                                              uint8_t *s_tile, uint32_t P, uint32_t w0, uint32_t nw, uint32_t lane,
                                              bool prezeroed = false)
{
    const uint32_t C = p.C, F = p.F, shift = 5 * P, ncells = nw * C;
#ifdef MRL_DIAG
    if (p.ablate & 8u) return;  // diagnostic build: no encode at all
#endif
    // The group's whole observation slab fits one LDS tile (small layouts): zero it, let
    // each cell-lane drop its 16 bytes into the rows of all P viewers plus the two player
    // bytes of whoever stands there, then stream the slab out.  Three LDS round trips per
    // launch in the dependency chain instead of three per 128-row pass.
    uint8_t *gobs = p.obs + (size_t)w0 * p.block_bytes;
    // (backstop: whatever world count the caller derived for the group, nothing is stored past the end of the whole slab)
    const uint32_t nbytes = min(nw, w0 < p.num_worlds ? p.num_worlds - w0 : 0u) * p.block_bytes;
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(gobs) & 15u);
    uint8_t *tile = s_tile + mis;
    if (!prezeroed) {
        const uint32_t nchunks = (mis + nbytes + 15u) >> 4;
        for (uint32_t k = lane; k < nchunks; k += kWave) reinterpret_cast<uint4 *>(s_tile)[k] = make_uint4(0, 0, 0, 0);
    }
    wave_lds_sync();
    const uint32_t plane = __umul24(C, F);  // bytes of one viewer's rows
    // (Giving a lane the cell pair 2i, 2i+1 so that each store has a wave-uniform alignment
    // class was tried: fewer LDS stores but worse lane utilisation, 4% slower overall.)
    for (uint32_t i = lane; i < ncells; i += kWave) {
        const uint32_t l = __umulhi(i, p.inv_c), c = i - __umul24(l, C);
        const uint32_t terr = s_terrain[c];
        const uint32_t o = s_obj[i];
        const uint32_t who = s_cur[i];
        const uint32_t urgent = s_flags[l];
        const bool occupied = who != 0xFF;
        const uint32_t pidx = (__umul24(l, P) + (occupied ? who : 0u)) * 2;
        const uint32_t w_ori = (s_pl[pidx] >> 8) & 0xFF;
        const uint32_t h = occupied ? s_pl[pidx + 1] : kItemNone;
        const uint4 t = cell_tail(p, terr, o, h, urgent);
        uint32_t off = __umul24(l, p.block_bytes) + __umul24(c, F);  // row (l, viewer 0, c)
        for (uint32_t v = 0; v < P; v++, off += plane) {
            uint8_t *row = tile + off;
            lds_store_tail_even(row + shift, t);  // this path is only taken for even P: every tail is 2-byte aligned
            if (occupied) {
                const uint32_t rel = who == v ? 0u : (who < v ? who + 1u : who);
                row[rel] = 1;
                row[P + 4 * rel + w_ori] = 1;
            }
        }
    }
    wave_lds_sync();
#ifdef MRL_DIAG
    if (p.ablate & 2u) return;  // diagnostic build: encode in LDS, no stores to HBM
#endif
    const uint32_t head = min((16u - mis) & 15u, nbytes);
    if (lane < head) gobs[lane] = tile[lane];
    const uint32_t body = (nbytes - head) >> 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(tile + head);
    // 16-byte body as raw buffer stores over exactly the body: chunks past its end are dropped by the
    // bounds check, so the four-deep batches need no per-lane branches
    const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(gobs + head, 0, (int)(body << 4), 0x00020000);
    // (reads past the body stay inside the workgroup's LDS or return zero; their stores are dropped by the bounds check)
    uint32_t k0 = lane;
    for (; k0 + 3 * kWave < body + lane; k0 += 4 * kWave) {
        const uint32_t ka = k0, kb = k0 + kWave, kc = k0 + 2 * kWave, kd = k0 + 3 * kWave;
        const uint4 va = src[ka], vb = src[kb], vc = src[kc], vd = src[kd];
        stream_store_rsrc<kPlain>(out, ka << 4, va);
        stream_store_rsrc<kPlain>(out, kb << 4, vb);
        stream_store_rsrc<kPlain>(out, kc << 4, vc);
        stream_store_rsrc<kPlain>(out, kd << 4, vd);
    }
    for (; k0 < body + lane; k0 += kWave) stream_store_rsrc<kPlain>(out, k0 << 4, src[k0]);
    const uint32_t done_bytes = head + (body << 4);
    if (lane < nbytes - done_bytes) gobs[done_bytes + lane] = tile[done_bytes + lane];
}

// Single-pass encode over a BACKGROUND image (p.patch; see the header of this file).  `tile` already
// holds the group's static rows (terrain one-hot bytes, zeros elsewhere) and starts on a 16-byte
// boundary like the group's slab in HBM.  What changes from step to step is confined to the cells
// with an object on them or a player standing there: those are compacted into s_list with wave
// ballots, and one lane per (dynamic cell, viewer) overwrites that row's 16-byte tail and sets the
// two player bytes; rows of worlds in their last 40 steps get the urgency byte in a pass of their own.
// the group's dynamic cells, compacted in ascending order into s_list; returns how many (wave-uniform).
// All LDS reads are issued before the first ballot (a loop that reads, votes and writes per 64 cells paid
// one LDS round trip per iteration: 0.5 us for three iterations with four waves per SIMD).
__device__ __forceinline__ uint32_t find_dynamic(const StepParams &p, const uint32_t *s_obj, const uint8_t *s_cur, uint16_t *s_list,
                                                 uint32_t l0, uint32_t nl, uint32_t lane)
{
    const uint32_t ncells = (l0 + nl) * p.C;  // cells [l0 * C, ncells): worlds l0 .. l0 + nl - 1 of the group
    constexpr int kBatch = 4;
    uint32_t ndyn = 0;
    for (uint32_t i0 = l0 * p.C; i0 < ncells; i0 += kBatch * kWave) {
        uint32_t o[kBatch], who[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const uint32_t i = min(i0 + (uint32_t)k * kWave + lane, ncells - 1u);
            o[k] = s_obj[i];
            who[k] = s_cur[i];
        }
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            if (i0 + (uint32_t)k * kWave >= ncells) break;  // wave-uniform
            const uint32_t i = i0 + (uint32_t)k * kWave + lane;
            const bool dyn = i < ncells && (((o[k] & 0xFFu) != O_NONE) | (who[k] != 0xFFu));
            const unsigned long long m = __ballot(dyn);
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (dyn) s_list[ndyn + before] = (uint16_t)i;
            ndyn += (uint32_t)__popcll(m);
        }
    }
    wave_lds_sync();
    return ndyn;
}

// p.direct: what is dynamic in a group is known without looking for it.  A player only ever stands on an AIR cell (moves go to AIR
// cells, sim.cpp:363-379; the host checks the start cells) and an object only ever lies on a counter or in a pot that a player can
// face (put / add are the only ways an object reaches a cell, sim.cpp:229-253,299-335) -- so a group's dynamic rows are the rows of
// its players' cells, encoded by the PLAYER lanes straight from their registers (terrain AIR, no object, what the player holds), and
// the rows of the HOLDER cells that hold something, one lane per holder cell from a table built on the host.  No cell -> player map,
// no ballot compaction, no list: one LDS read (the holder's object) between the transition and the patch stores.
constexpr int kHoldPerLane = 4;  // 256 table entries per group
struct HoldTab {
    uint32_t e[kHoldPerLane];
};
__device__ __forceinline__ void hold_request(const StepParams &p, uint32_t lane, HoldTab &h)
{
    const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.hold_tab), 0, (int)(p.hold_entries * 4u), 0x00020000);
#pragma unroll
    for (int k = 0; k < kHoldPerLane; k++) {
        if ((uint32_t)k * kWave >= p.hold_entries) {  // wave-uniform
            h.e[k] = 0;
            continue;
        }
        h.e[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(tab, (int)((lane + (uint32_t)k * kWave) * 4u), 0, 0);
    }
}
// kUndo: put the same rows back to their static content (after the stream-out of a tile that outlives the step)
template <int kP, bool kUndo>
__device__ __forceinline__ void patch_direct(const StepParams &p, uint32_t *s_obj, const uint8_t *s_flags, const HoldTab &hold,
                                             uint8_t *tile, uint32_t P, uint32_t nw, bool active, uint32_t wl, uint32_t q, uint32_t posori,
                                             uint32_t held)
{
    const uint32_t C = p.C, F = p.F, shift = 5 * P, ncells = nw * C;
    const uint32_t plane = __umul24(C, F);
    const uint32_t ori = (posori >> 8) & 0xFFu;
#pragma unroll
    for (int k = 0; k < kHoldPerLane; k++) {
        if (k > 0 && (uint32_t)k * kWave >= p.hold_entries) break;  // wave-uniform
        const uint32_t e = hold.e[k];
        const uint32_t i = (e >> 16) & 0x3FFFu;
        const bool holder = ((e >> 30) & 1u) != 0u && i < ncells;
        uint32_t o = s_obj[holder ? i : 0u];
        if (!kUndo && holder && (e >> 31)) {
            // the pot's tick (sim.cpp:430-438) rides on this read instead of a pass of its own (tick_pots) with its own LDS
            // round trip in front of the first store: every pot that can hold a soup is a holder cell; a pot of a world
            // that reset in this step has been emptied by now and does not tick, like an empty one
            const int32_t tick = (int8_t)(o >> 24);
            if ((o & 0xFFu) == O_SOUP && tick >= 0 && tick < (int32_t)lookup16(p.times_w, recipe_of(o))) {
                o = (o & 0x00FFFFFFu) | ((uint32_t)(uint8_t)(tick + 1) << 24);
                s_obj[i] = o;
            }
        }
        const bool player = k == 0 && active;
        if (player || (holder && (o & 0xFFu) != O_NONE)) {
            const uint32_t l = player ? wl : __umulhi(i, p.inv_c);
            const uint32_t terr = player ? (uint32_t)T_AIR : ((e >> 31) ? (uint32_t)T_POT : (uint32_t)T_COUNTER);
            const uint4 t = cell_tail(p, terr, (player || kUndo) ? (uint32_t)kItemNone : o, (player && !kUndo) ? held : (uint32_t)kItemNone, s_flags[l]);
            uint8_t *row0 = tile + (player ? __umul24(l, p.block_bytes) + __umul24(posori & 0xFFu, F) : (e & 0xFFFFu));
            auto viewer_row = [&](uint32_t v, uint8_t *row) {
                lds_store_tail_even(row + shift, t);
                if (player) {
                    const uint32_t rel = q == v ? 0u : (q < v ? q + 1u : q);
                    row[rel] = kUndo ? 0 : 1;
                    row[P + 4 * rel + ori] = kUndo ? 0 : 1;
                }
            };
            if constexpr (kP == 2) {
                viewer_row(0, row0);
                viewer_row(1, row0 + plane);
            } else {
                for (uint32_t v = 0; v < P; v++) viewer_row(v, row0 + __umul24(v, plane));
            }
        }
    }
}

// kRestore (persistent rollouts, where the tile outlives the step): after the stream-out the patched rows are
// put back to their static content, so the next step again only touches what is dynamic then; s_prev remembers
// each world's urgency flag as the tile has it.
template <int kP, bool kRestore, bool kPlain = false>
__device__ __forceinline__ void observe_patch(const StepParams &p, const uint8_t *s_terrain, uint32_t *s_obj,
                                              const uint32_t *s_pl, const uint8_t *s_cur, const uint8_t *s_flags, uint8_t *s_prev,
                                              const uint16_t *s_list, uint32_t ndyn, uint8_t *tile, uint32_t P, uint32_t w0, uint32_t l0,
                                              uint32_t nl, uint32_t lane, const HoldTab &hold, bool active, uint32_t wl, uint32_t q,
                                              uint32_t posori, uint32_t held)
{
    // worlds l0 .. l0 + nl - 1 of the group (the single step encodes a group in two halves so that the first
    // half's stores are on their way while the second is still being patched); `tile` is the whole group's
    const uint32_t C = p.C, F = p.F, shift = 5 * P;
#ifdef MRL_DIAG
    if (p.ablate & 8u) return;  // diagnostic build: no encode at all
#endif
    const uint32_t plane = __umul24(C, F);  // bytes of one viewer's rows
    // lane = dynamic cell; its 16-byte tail is worked out once and dropped into the rows of all P viewers
    if (p.direct) patch_direct<kP, false>(p, s_obj, s_flags, hold, tile, P, nl, active, wl, q, posori, held);
    for (uint32_t j = lane; j < (p.direct ? 0u : ndyn); j += kWave) {
        const uint32_t i = s_list[j];
        const uint32_t l = __umulhi(i, p.inv_c), c = i - __umul24(l, C);
        const uint32_t terr = s_terrain[c];
        const uint32_t o = s_obj[i];
        const uint32_t who = s_cur[i];
        const uint32_t urgent = s_flags[l];
        const bool occupied = who != 0xFF;
        const uint32_t pidx = (__umul24(l, P) + (occupied ? who : 0u)) * 2;
        const uint32_t w_ori = (s_pl[pidx] >> 8) & 0xFF;
        const uint32_t h = occupied ? s_pl[pidx + 1] : kItemNone;
        const uint4 t = cell_tail(p, terr, o, h, urgent);
        uint8_t *row0 = tile + __umul24(l, p.block_bytes) + __umul24(c, F);
        auto viewer_row = [&](uint32_t v, uint8_t *row) {
            lds_store_tail_even(row + shift, t);  // this path is only taken for even P: every tail is 2-byte aligned
            if (occupied) {
                const uint32_t rel = who == v ? 0u : (who < v ? who + 1u : who);
                row[rel] = 1;
                row[P + 4 * rel + w_ori] = 1;
            }
        };
        if constexpr (kP == 2) {
            viewer_row(0, row0);
            viewer_row(1, row0 + plane);
        } else {
            for (uint32_t v = 0; v < P; v++) viewer_row(v, row0 + __umul24(v, plane));
        }
    }
    // urgency channel (sim.cpp:79-83) of the rows that were not patched: only where a world's flag differs from
    // what the tile holds (a fresh tile holds 0); the flag changes twice per episode
    {
        const uint32_t mine = lane < nl ? s_flags[l0 + lane] : 0u;
        const uint32_t had = (kRestore && lane < nl) ? s_prev[l0 + lane] : 0u;
        if (__builtin_expect(__ballot(mine != had) != 0ull, 0)) {
            const uint32_t nrows = (l0 + nl) * p.rows;
            for (uint32_t r = l0 * p.rows + lane; r < nrows; r += kWave) {
                const uint32_t l = __umulhi(r, p.inv_rows);
                const uint32_t f = s_flags[l];
                if (f != (kRestore ? (uint32_t)s_prev[l] : 0u)) tile[__umul24(r, F) + F - 1u] = (uint8_t)f;
            }
            wave_lds_sync();
            if (kRestore && lane < nl) s_prev[l0 + lane] = (uint8_t)mine;
        }
    }
    wave_lds_sync();
#ifdef MRL_DIAG
    if (p.ablate & 2u) return;  // diagnostic build: encode in LDS, no stores to HBM
#endif
    // stream the slab out: 16-byte body as raw buffer stores over exactly the body (chunks past its end are
    // dropped by the bounds check, so the four-deep batches need no per-lane branches), then the odd tail bytes
    uint8_t *gobs = p.obs + (size_t)(w0 + l0) * p.block_bytes;
    // (backstop: whatever world count the caller derived for the group -- the ragged-batch fault of round 2 was a wrong one --
    // the descriptor below never reaches past the end of the whole slab)
    const uint32_t nbytes = min(nl, w0 + l0 < p.num_worlds ? p.num_worlds - (w0 + l0) : 0u) * p.block_bytes;
    const uint8_t *from = tile + __umul24(l0, p.block_bytes);
    const uint32_t head = p.unaligned ? min((16u - ((uint32_t)reinterpret_cast<uintptr_t>(gobs) & 15u)) & 15u, nbytes) : 0u;
    if (lane < head) gobs[lane] = from[lane];
    const uint32_t body = (nbytes - head) >> 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(from + head);
    const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(gobs + head, 0, (int)(body << 4), 0x00020000);
    // (reads past the body stay inside the workgroup's LDS or return zero; their stores are dropped by the bounds check)
    // A store instruction covers 64 consecutive chunks = 1 KB.  Where a group's slab does not start on a 128-byte line
    // (coordination_ring / forced_coordination: 5200-byte groups, asymmetric_advantages 9360), rounds counted from the slab's
    // first chunk would make EVERY instruction straddle nine lines, two of them partial -- and a line's two parts are two
    // write-through transactions, which is what the drain is made of (section 4.1 of DESIGN.md).  So the rounds are counted
    // from the line boundary in front of the slab: the first `lshift` lanes of round 0 fall in front of it (their offset
    // wraps, the descriptor drops them; their LDS read lands in the wave's own region), every other instruction is eight
    // whole lines, and only the slab's first and last line are shared with the neighbouring groups.
    // Same box, us per launch with / without (profiles/r04_ae_overcooked_line_rounds_ab.txt): coordination_ring 10.14-10.17 / 10.49-10.50,
    // forced_coordination 10.47 / 10.67, asymmetric_advantages 15.63-15.69 / 15.75-15.80; the aligned layouts are untouched (lshift = 0).
    // Not in the multi-step launches: their ordinary stores meet in the L2 anyway, and the extra round cost them 0.1 us per step.
#ifndef MRL_NO_LINE_ROUNDS
    const uint32_t lshift = kRestore ? 0u : (uint32_t)(reinterpret_cast<uintptr_t>(gobs + head) >> 4) & 7u;
#else
    const uint32_t lshift = 0u;
#endif
    // (signed chunk indices: a lane in front of the slab reads up to 112 bytes below the tile -- the wave's own state -- and its
    // store offset wraps to 0xFFFFFF90 and more, which the descriptor drops)
    int32_t b0 = -(int32_t)lshift;  // wave-uniform
    for (; b0 + 3 * kWave < (int32_t)body; b0 += 4 * kWave) {
        const int32_t ka = b0 + (int32_t)lane, kb = ka + kWave, kc = ka + 2 * kWave, kd = ka + 3 * kWave;
        const uint4 va = src[ka], vb = src[kb], vc = src[kc], vd = src[kd];
        stream_store_rsrc<kPlain>(out, (uint32_t)ka << 4, va);
        stream_store_rsrc<kPlain>(out, (uint32_t)kb << 4, vb);
        stream_store_rsrc<kPlain>(out, (uint32_t)kc << 4, vc);
        stream_store_rsrc<kPlain>(out, (uint32_t)kd << 4, vd);
    }
    for (; b0 < (int32_t)body; b0 += kWave) {
        const int32_t k = b0 + (int32_t)lane;
        stream_store_rsrc<kPlain>(out, (uint32_t)k << 4, src[k]);
    }
    const uint32_t done_bytes = head + (body << 4);
    if (lane < nbytes - done_bytes) gobs[done_bytes + lane] = from[done_bytes + lane];
    if constexpr (kRestore) {
        wave_lds_sync();
        if (p.direct) patch_direct<kP, true>(p, s_obj, s_flags, hold, tile, P, nl, active, wl, q, posori, held);
        for (uint32_t j = lane; j < (p.direct ? 0u : ndyn); j += kWave) {
            const uint32_t i = s_list[j];
            const uint32_t l = __umulhi(i, p.inv_c), c = i - __umul24(l, C);
            const uint32_t who = s_cur[i];
            const bool occupied = who != 0xFF;
            const uint32_t w_ori = (s_pl[(__umul24(l, P) + (occupied ? who : 0u)) * 2] >> 8) & 0xFF;
            const uint4 t = cell_tail(p, s_terrain[c], kItemNone, kItemNone, s_flags[l]);  // the row's static tail
            uint8_t *row0 = tile + __umul24(l, p.block_bytes) + __umul24(c, F);
            auto viewer_row = [&](uint32_t v, uint8_t *row) {
                lds_store_tail_even(row + shift, t);
                if (occupied) {
                    const uint32_t rel = who == v ? 0u : (who < v ? who + 1u : who);
                    row[rel] = 0;
                    row[P + 4 * rel + w_ori] = 0;
                }
            };
            if constexpr (kP == 2) {
                viewer_row(0, row0);
                viewer_row(1, row0 + plane);
            } else {
                for (uint32_t v = 0; v < P; v++) viewer_row(v, row0 + __umul24(v, plane));
            }
        }
        wave_lds_sync();
    }
}

// The static part of the tile: zeros plus one terrain one-hot byte per row of a non-AIR cell
// (sim.cpp:642-645).  A lane takes rows i = lane, lane + 64, ... of the GROUP and ends up with the tile offsets
// of their terrain bytes in registers.
// (Copying a ready-made background image from global memory instead was measured: 10 KB per wave through
// the vector memory pipe, 64 B/clk per CU, cost ~1 us per step.)
constexpr int kTerrPosPerLane = 10;  // 64 x 10 rows: room for 16-world groups of the smallest layout
struct TerrPos {
    uint32_t off[kTerrPosPerLane];  // byte offset in the tile, 0 = nothing to set (no terrain byte sits at offset 0)
};
// p.terr_off is that table for a whole GROUP (entry i = row i of the group, worlds in order): one bounds-checked
// 2-byte load per 64 rows, no division, requested together with the state loads
__device__ __forceinline__ void terrain_request(const StepParams &p, uint32_t lane, TerrPos &r)
{
    const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.terr_off), 0, (int)(p.terr_entries * 2u), 0x00020000);
#pragma unroll
    for (int k = 0; k < kTerrPosPerLane; k++) {
        if ((uint32_t)k * kWave >= p.terr_entries) {  // wave-uniform
            r.off[k] = 0;
            continue;
        }
        r.off[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(tab, (int)((lane + (uint32_t)k * kWave) * 2u), 0, 0);
    }
}
__device__ __forceinline__ void terrain_deliver(const StepParams &p, const TerrPos &r, uint8_t *tile, uint32_t nw)
{
    const uint32_t limit = nw * p.block_bytes;  // a ragged last group holds fewer worlds
#pragma unroll
    for (int k = 0; k < kTerrPosPerLane; k++) {
        if ((uint32_t)k * kWave >= p.terr_entries) break;
        if (r.off[k] != 0u && r.off[k] < limit) tile[r.off[k]] = 1;
    }
}
__device__ __forceinline__ void tile_zero(const StepParams &p, uint32_t lane, uint8_t *tile, uint32_t nw)
{
    const uint32_t nchunks = (nw * p.block_bytes + 31u) >> 4;  // covers any start misalignment
    for (uint32_t k = lane; k < nchunks; k += kWave) reinterpret_cast<uint4 *>(tile)[k] = make_uint4(0, 0, 0, 0);
}
// Zero-fill of a 256-byte-granular tile with ds_write_addtid_b32 (LDS address = M0 + offset + 4 * lane, no address
// register): 256 bytes per instruction at twice the rate of ds_write_b32 and 1.6x that of ds_write_b128
// (MI355X_MICROARCH.md, LDS).  The tile is the hot LDS-write traffic of the kernel: 16 waves per CU x 8.3 KB.
// M0 is written and restored inside each statement (hipcc reserves it and does not preserve it around asm).
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

#ifdef MRL_DIAG
#define STAMP(k)                                                                                               \
    do {                                                                                                       \
        if (p.stamps && lane == 0)                                                                             \
            p.stamps[(size_t)(blockIdx.x * kWavesPerBlock + wib) * 16 + (k)] = __builtin_amdgcn_s_memtime();   \
    } while (0)
#define STAMP_REALTIME(k)                                                                                      \
    do {                                                                                                       \
        if (p.stamps && lane == 0)                                                                             \
            p.stamps[(size_t)(blockIdx.x * kWavesPerBlock + wib) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define ABLATED(bit) (p.ablate & (bit))
#else
#define STAMP(k) ((void)0)
#define STAMP_REALTIME(k) ((void)0)
#define ABLATED(bit) false
#endif

// Action `at` of the caller's array: int32, or int64 narrowed to its low word -- only that word is loaded (an 8-byte
// load whose upper half is dead made hipcc wait for ALL outstanding loads before reusing the register: 0.45 us).
// One base pointer and a shift rather than a select between two element loads.
__device__ __forceinline__ uint32_t load_action(const StepParams &p, size_t at)
{
    const bool wide = p.actions64 != nullptr;
    const char *base = wide ? reinterpret_cast<const char *>(p.actions64) : reinterpret_cast<const char *>(p.actions);
    return *reinterpret_cast<const uint32_t *>(base + (at << (wide ? 3 : 2)));
}

// `block`: the workgroup's index among those of THIS simulator (blockIdx.x, except under mrl_overcooked_step_many, where
// one grid covers the workgroups of several simulators)
template <bool kInit, int kP, bool kPlain = false>
__device__ __forceinline__ void step_body(const StepParams &p, const uint32_t block)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    // wave-uniform values are forced into SGPRs so the address math around them is scalar
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));

    STAMP(0);
    STAMP_REALTIME(13);
#ifdef MRL_DIAG
    // where the wave runs: HW_ID (wave / SIMD / CU / SE) and XCC_ID, for tools/hwid_map.py
    if (p.stamps && lane == 0)
        p.stamps[(size_t)(blockIdx.x * kWavesPerBlock + wib) * 16 + 12] =
            (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif
    if (ABLATED(16)) return;  // diagnostic build: the empty launch
    // the constants and the group's state slab are fetched together: one HBM/L2 latency.  Small configurations
    // (p.private_consts) need only the terrain in LDS, one private copy per wave: no barrier, the waves never meet.
    constexpr int kConstWordsPerThread = (kConstBytes / 4 + kBlock - 1) / kBlock;
    uint32_t const_word[kConstWordsPerThread];
    const bool private_consts = p.private_consts != 0;  // wave-uniform
    if (private_consts) {
        const __amdgpu_buffer_rsrc_t r_terr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.consts), 0, (int)((p.C + 3u) & ~3u), 0x00020000);
        const_word[0] = __builtin_amdgcn_raw_buffer_load_b32(r_terr, (int)(lane * 4u), 0, 0);
    } else {
#pragma unroll
        for (int j = 0; j < kConstWordsPerThread; j++) const_word[j] = tid + j * kBlock < kConstBytes / 4 ? p.consts[tid + j * kBlock] : 0u;
    }
    const uint8_t *s_terrain = private_consts ? smem + kConstBytes + wib * p.lds_wave_stride + p.off_terr : smem + kConstTerrain;
    const uint8_t *s_start = smem + kConstStart;
    const uint8_t *s_pots = smem + kConstPots;

    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs, so
    // give each XCD one contiguous range of worlds (neighbouring groups share
    // cache lines of the state arrays and of the observation slab; keep them in one L2).
    const uint32_t per_xcd = p.per_xcd;
    const uint32_t logical_block = (block & 7u) * per_xcd + (block >> 3);
    const uint32_t w0 = p.share ? logical_block : (logical_block * kWavesPerBlock + wib) * p.wpw;
    const uint32_t nw = w0 < p.num_worlds ? min(p.wpw, p.num_worlds - w0) : 0u;

    // per-wave LDS: the group's state in the same dense order as in HBM
    uint8_t *wbase = smem + kConstBytes + wib * p.lds_wave_stride;
    uint32_t *s_obj = reinterpret_cast<uint32_t *>(wbase);                 // [wpw][C]
    uint32_t *s_pl = reinterpret_cast<uint32_t *>(wbase + p.off_pl);       // [wpw][P][2]
    uint32_t *s_x = reinterpret_cast<uint32_t *>(wbase + p.off_x);         // [128] scratch of the transition
    uint32_t *s_sum = reinterpret_cast<uint32_t *>(wbase + p.off_sum);     // [2][wpw]
    uint32_t *s_blk = s_sum + p.wpw;
    uint8_t *s_cur = wbase + p.off_cur;                                    // [wpw][C] cell -> player
    uint8_t *s_flags = wbase + p.off_flags;                                // [64] urgency per world
    uint16_t *s_list = reinterpret_cast<uint16_t *>(wbase + p.off_list);   // [wpw][C] dynamic cells
    uint4 *s_tail = reinterpret_cast<uint4 *>(wbase + p.off_tail);         // [wpw][C]
    uint8_t *s_tile = wbase + p.off_tile;

    const uint32_t P = kP > 0 ? (uint32_t)kP : p.P, C = p.C, N = p.num_worlds;
    const uint32_t ncells = nw * C, nplayers = nw * P;
    // lane = (world of the group, player)
    const uint32_t wl = kP == 2 ? lane >> 1 : (P == 1u ? lane : __umulhi(lane, p.inv_p));
    const uint32_t q = lane - wl * P;
    const bool active = lane < nplayers;

    uint32_t posori = 0, held = kItemNone, act = A_STAY;
    int32_t t_loaded = 0;
    TerrPos tpos;
    HoldTab hold{};
    // p.unaligned: the tile of the patch encode is as misaligned as the group's slab in HBM
    const uint32_t patch_mis = p.unaligned ? (uint32_t)(reinterpret_cast<uintptr_t>(p.obs + (size_t)w0 * p.block_bytes) & 15u) : 0u;
    // ---------------- load: HBM slab -> LDS (straight copies) ----------------
    // All global loads of the group are issued before the first one is consumed (explicitly
    // batched: a plain copy loop waits for each load before issuing the next, which measured
    // 2.1 us -- three to four dependent HBM/L2 latencies -- for a 1.2 KB slab).
    if (!kInit) {
        const uint32_t *g_obj = p.cell_obj + (size_t)w0 * C;
        constexpr int kBatch = 4;
        uint32_t cell_reg[kBatch];
        // bounds-checked buffer loads: lanes beyond the group's slab read zero, no branch around any load (behind a
        // branch hipcc consumes a load on the spot, `s_waitcnt vmcnt(0)` before the others are even issued)
        const __amdgpu_buffer_rsrc_t r_obj = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(g_obj), 0, (int)(ncells * 4u), 0x00020000);
#pragma unroll
        for (int k = 0; k < kBatch; k++) cell_reg[k] = __builtin_amdgcn_raw_buffer_load_b32(r_obj, (int)((lane + (uint32_t)k * kWave) * 4u), 0, 0);
        const __amdgpu_buffer_rsrc_t r_pl = __builtin_amdgcn_make_buffer_rsrc(p.players + (size_t)w0 * P, 0, (int)(nplayers * 8u), 0x00020000);
        const auto pl_raw = __builtin_amdgcn_raw_buffer_load_b64(r_pl, (int)(lane * 8u), 0, 0);
        const uint2 pl_reg = make_uint2(pl_raw[0], pl_raw[1]);
        const size_t a_at = (size_t)(active ? q : 0u) * N + min(w0 + wl, N - 1u);
        const uint32_t a_raw = load_action(p, a_at);
        t_loaded = p.timestep[min(w0 + wl, N - 1u)];
        if (p.patch) terrain_request(p, lane, tpos);
        // while the loads are in flight: the cell -> player map starts empty, the tile of the single-pass encode zeroed
        if (!p.direct)
            for (uint32_t i = lane; i < (p.wpw * C + 3u) >> 2; i += kWave) reinterpret_cast<uint32_t *>(s_cur)[i] = 0xFFFFFFFFu;
        if (p.patch)
            tile_zero_addtid(s_tile, nw * p.block_bytes + patch_mis);
        else if (p.whole)
            tile_zero(p, lane, s_tile, nw);
        if (p.direct) hold_request(p, lane, hold);  // not among the preloaded arguments: asked for behind the fill
        STAMP(6);
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const uint32_t i = lane + k * kWave;
            if (i < ncells) s_obj[i] = cell_reg[k];
        }
        for (uint32_t i = lane + kBatch * kWave; i < ncells; i += kWave) s_obj[i] = g_obj[i];
        posori = pl_reg.x & 0xFFFFu;
        held = pl_reg.y;
        act = (active && a_raw <= A_INTERACT) ? a_raw : (uint32_t)A_STAY;  // outside the enum = outside the contract
        if (p.actions64 && active && (!p.share || wib == 0)) p.action_mirror[a_at] = (int32_t)a_raw;
        if (!active) {
            posori = 0;
            held = kItemNone;
        }
    } else {
        if (p.patch) terrain_request(p, lane, tpos);
        if (!p.direct)
            for (uint32_t i = lane; i < (p.wpw * C + 3u) >> 2; i += kWave) reinterpret_cast<uint32_t *>(s_cur)[i] = 0xFFFFFFFFu;
        if (p.patch)
            tile_zero_addtid(s_tile, nw * p.block_bytes + patch_mis);
        else if (p.whole)
            tile_zero(p, lane, s_tile, nw);
        if (p.direct) hold_request(p, lane, hold);
    }
    if (p.patch) terrain_deliver(p, tpos, s_tile + patch_mis, nw);
    if (private_consts) {
        if (lane * 4u < p.C) reinterpret_cast<uint32_t *>(const_cast<uint8_t *>(s_terrain))[lane] = const_word[0];
        STAMP(7);
        wave_lds_sync();
    } else {
#pragma unroll
        for (int j = 0; j < kConstWordsPerThread; j++)
            if (tid + j * kBlock < kConstBytes / 4) reinterpret_cast<uint32_t *>(smem)[tid + j * kBlock] = const_word[j];
        STAMP(7);
        __syncthreads();
    }
    if (nw == 0) return;
    if (ABLATED(32)) return;  // diagnostic build: launch + loads + barrier
    STAMP(1);

    // ---------------- step: lane = (world, player) ----------------
    int32_t reward_world = 0, t = 0;
    bool reset_now = kInit;
    if (!kInit) {
        if (!ABLATED(4)) {
            transition_lanes<kP>(p, s_terrain, s_obj + wl * C, s_x, s_sum, s_blk, P, lane, active, wl, q, act, posori, held, reward_world);
            if (!p.direct) tick_pots(p, s_pots, s_obj, nw, lane);  // direct: the holder lanes of the encode tick the pots
        }
        // horizon (sim.cpp:485-489)
        t = t_loaded + 1;
        reset_now = (int64_t)t >= p.horizon;
    }
    // reset (sim.cpp:441-482): rare, and then usually every world of the group at once
    if (__builtin_expect(__ballot(active && reset_now) != 0ull, 0)) {
        if (reset_now) {
            t = 0;
            posori = (private_consts ? (p.starts_w >> (8u * (q & 3u))) & 0xFFu : (uint32_t)s_start[active ? q : 0u]) | (A_NORTH << 8);
            held = kItemNone;
        }
        if (active && q == 0) s_sum[wl] = reset_now ? 1u : 0u;
        wave_lds_sync();
        for (uint32_t i = lane; i < ncells; i += kWave)
            if (s_sum[__umulhi(i, p.inv_c)] != 0u) s_obj[i] = kItemNone;
        wave_lds_sync();
    }
    STAMP(2);
    // what the encode reads: player records, cell -> player map, urgency channel (sim.cpp:79-83)
    if (active) {
        if (!p.direct) {
            reinterpret_cast<uint2 *>(s_pl)[lane] = make_uint2(posori, held);
            s_cur[wl * C + (posori & 0xFFu)] = (uint8_t)q;
        }
        if (q == 0) s_flags[wl] = (p.horizon - (int64_t)t < 40) ? 1 : 0;
    }
    wave_lds_sync();

    // ---------------- store: LDS -> HBM slab, rewards, flags ----------------
    // The LAST thing a wave does: issued before the observation stream-out these stores sit in front
    // of it in the wave's vmcnt order, and hipcc parks the wave on `s_waitcnt vmcnt(0)` -- a store
    // round trip -- as soon as one of their address registers is reused.  Nothing waits for them here.
    auto store_state = [&]() {
        if (p.share && wib != 0) return;  // the siblings computed the same state
        uint32_t *g_obj = p.cell_obj + (size_t)w0 * C;
#ifdef MRL_STATE_SC1
        for (uint32_t i = lane; i < ncells; i += kWave) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(g_obj + i), "v"(s_obj[i]) : "memory");
        if (active) {
            const uint32_t world = w0 + wl;
            asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p.players + (size_t)w0 * P + lane), "v"(make_uint2(posori, held)) : "memory");
            asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p.reward + (size_t)q * N + world), "v"(reward_world) : "memory");
            if (q == 0) {
                asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p.timestep + world), "v"(t) : "memory");
                asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p.done + world), "v"(kInit ? 0 : (int32_t)reset_now) : "memory");
            }
        }
#else
        for (uint32_t i = lane; i < ncells; i += kWave) g_obj[i] = s_obj[i];
        if (active) {
            const uint32_t world = w0 + wl;
            p.players[(size_t)w0 * P + lane] = make_uint2(posori, held);
            p.reward[(size_t)q * N + world] = reward_world;
            if (q == 0) {
                p.timestep[world] = t;
                p.done[world] = kInit ? 0 : (int32_t)reset_now;
            }
        }
#endif
    };

    STAMP(3);
    // ---------------- observe (sim.cpp:68-167, 642-645) ----------------
    // A row is [5P viewer-relative player channels][16 viewer-independent bytes].  The 16 bytes
    // (terrain one-hot, pot / soup / item channels incl. what the player standing there holds,
    // urgency) are computed once per CELL, straight-line, and reused by the P viewers' rows.
    const uint32_t F = p.F, shift = 5 * P;
    if (p.whole) {
        if (p.patch) {
            // (Encoding and streaming the group in two halves, so that the first half's stores leave while the second is
            // patched, measured slower: 8.6 vs 8.25 us.)
            const uint32_t ndyn = p.direct ? 0u : find_dynamic(p, s_obj, s_cur, s_list, 0, nw, lane);
            STAMP(4);
            observe_patch<kP, false, kPlain>(p, s_terrain, s_obj, s_pl, s_cur, s_flags, nullptr, s_list, ndyn, s_tile + patch_mis, P, w0, 0, nw, lane, hold, active,
                                     wl, q, posori, held);
            STAMP(5);
        } else
            observe_whole<kPlain>(p, s_terrain, s_obj, s_pl, s_cur, s_flags, s_tile, P, w0, nw, lane, true);
        store_state();
        STAMP(15);
        STAMP_REALTIME(14);
        return;
    }
    for (uint32_t i = lane; i < ncells; i += kWave) {
        const uint32_t l = __umulhi(i, p.inv_c), c = i - __umul24(l, C);
        const uint32_t terr = s_terrain[c];
        const uint32_t o = s_obj[i];
        const uint32_t who = s_cur[i];
        const uint32_t urgent = s_flags[l];
        const uint32_t h = who != 0xFF ? s_pl[(__umul24(l, P) + who) * 2 + 1] : kItemNone;
        const uint4 t = cell_tail(p, terr, o, h, urgent);
        s_tail[i] = t;
    }
    wave_lds_sync();

    STAMP(4);
    // Rows are assembled kRowsPerPass at a time in an LDS tile and streamed out.  When a world
    // has no more rows than a tile, a pass covers whole worlds (p.wpp of them), so which
    // (world-in-pass, viewer, cell) a lane's rows are is the same in every pass and is worked
    // out once, outside the loop; larger layouts take the generic per-pass indexing.
    // Lane i assembles rows 2i and 2i+1 of the tile: each of the two has one alignment class
    // for the whole wave (F is even for even P), which picks the LDS store sequence.
    constexpr uint32_t kRowsPerLane = kRowsPerPass / kWave;
    const uint32_t rows = p.rows;
    const uint32_t pass_rows = p.wpp ? p.wpp * rows : (uint32_t)kRowsPerPass;
    uint32_t k_wl[kRowsPerLane], k_viewer[kRowsPerLane], k_c[kRowsPerLane];
#pragma unroll
    for (uint32_t j = 0; j < kRowsPerLane; j++) {
        const uint32_t tr = lane * kRowsPerLane + j;
        const uint32_t wl = __umulhi(tr, p.inv_rows), r = tr - __umul24(wl, rows);
        k_wl[j] = wl;
        k_viewer[j] = __umulhi(r, p.inv_c);
        k_c[j] = r - __umul24(k_viewer[j], C);
    }
    const uint32_t total_rows = nw * rows;
    uint8_t *gobs = p.obs + (size_t)w0 * p.block_bytes;
    // p.steady: every pass of this wave starts at the same address mod 16 and covers whole
    // worlds, so the tile is zero-filled once; a pass overwrites all 16-byte tails, sets the
    // player bytes, streams the tile out and takes the player bytes back out.
    if (p.steady) {
        const uint32_t mis0 = (uint32_t)(reinterpret_cast<uintptr_t>(gobs) & 15u);
        const uint32_t nchunks = (mis0 + min(pass_rows, total_rows) * F + 15u) >> 4;
        for (uint32_t k = lane; k < nchunks; k += kWave) reinterpret_cast<uint4 *>(s_tile)[k] = make_uint4(0, 0, 0, 0);
    }
    uint32_t first_world = 0;  // of the current pass (aligned passes only)
    const uint32_t pass_step = p.share ? kWavesPerBlock * pass_rows : pass_rows;
    for (uint32_t r0 = p.share ? wib * pass_rows : 0u; r0 < (ABLATED(8) ? 0u : total_rows); r0 += pass_step, first_world += p.wpp) {
        const uint32_t nrows = min(pass_rows, total_rows - r0);
        const uint32_t nbytes = nrows * F;
        uint8_t *g = gobs + (size_t)r0 * F;
        // keep LDS and global addresses congruent mod 16 so aligned 16-byte
        // chunks line up on both sides
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(g) & 15u);
        uint8_t *tile = s_tile + mis;

        if (!p.steady) {
            const uint32_t nchunks = (mis + nbytes + 15u) >> 4;
            for (uint32_t k = lane; k < nchunks; k += kWave) reinterpret_cast<uint4 *>(s_tile)[k] = make_uint4(0, 0, 0, 0);
        }
        wave_lds_sync();

        // a lane's rows: indices, then all LDS reads together, then the stores
        bool valid[kRowsPerLane], occupied[kRowsPerLane];
        uint32_t cell[kRowsPerLane], pidx[kRowsPerLane], viewer[kRowsPerLane], who[kRowsPerLane], ori[kRowsPerLane];
        uint4 t[kRowsPerLane];
#pragma unroll
        for (uint32_t j = 0; j < kRowsPerLane; j++) {
            const uint32_t tr = lane * kRowsPerLane + j;
            valid[j] = tr < nrows && !ABLATED(1);
            uint32_t l, c;
            if (p.wpp) {
                l = first_world + k_wl[j];
                viewer[j] = k_viewer[j];
                c = k_c[j];
            } else {
                const uint32_t gr = r0 + tr;
                l = __umulhi(gr, p.inv_rows);
                const uint32_t r = gr - __umul24(l, rows);
                viewer[j] = __umulhi(r, p.inv_c);
                c = r - __umul24(viewer[j], C);
            }
            cell[j] = valid[j] ? __umul24(l, C) + c : 0u;
            pidx[j] = valid[j] ? __umul24(l, P) : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < kRowsPerLane; j++) {
            t[j] = s_tail[cell[j]];
            who[j] = s_cur[cell[j]];
        }
#pragma unroll
        for (uint32_t j = 0; j < kRowsPerLane; j++) {
            occupied[j] = valid[j] && who[j] != 0xFF;
            ori[j] = (s_pl[(pidx[j] + (occupied[j] ? who[j] : 0u)) * 2] >> 8) & 0xFF;
        }
        uint32_t mark_a[kRowsPerLane], mark_b[kRowsPerLane];  // tile offsets of the two player bytes
#pragma unroll
        for (uint32_t j = 0; j < kRowsPerLane; j++) {
            const uint32_t row = __umul24(lane * kRowsPerLane + j, F);
            // alignment class of this row's tail; wave-uniform when F is even (the row index has the parity of j)
            const uint32_t cls = p.tail_even ? ((mis + j * F + shift) & 3u) : 1u;
            if (valid[j]) lds_store_tail(tile + row + shift, t[j], cls);
            // the viewer-relative player channels: two bytes, only on occupied cells
            const uint32_t rel = who[j] == viewer[j] ? 0u : (who[j] < viewer[j] ? who[j] + 1u : who[j]);
            mark_a[j] = row + rel;
            mark_b[j] = row + P + 4 * rel + ori[j];
            if (occupied[j]) {
                tile[mark_a[j]] = 1;
                tile[mark_b[j]] = 1;
            }
        }
        wave_lds_sync();

        // stream the tile out: unaligned head/tail bytes, 16-byte body (LDS reads batched four deep)
        const uint32_t head = min((16u - mis) & 15u, nbytes);
        if (lane < head) g[lane] = tile[lane];
        const uint32_t body = (nbytes - head) >> 4;
        const uint4 *src = reinterpret_cast<const uint4 *>(tile + head);
        uint4 *dst = reinterpret_cast<uint4 *>(g + head);
        if (!ABLATED(2)) {
            for (uint32_t k0 = lane; k0 < body + lane; k0 += 4 * kWave) {
                const uint32_t ka = k0, kb = k0 + kWave, kc = k0 + 2 * kWave, kd = k0 + 3 * kWave;
                const bool ba = ka < body, bb = kb < body, bc = kc < body, bd = kd < body;
                const uint4 va = src[ba ? ka : 0u], vb = src[bb ? kb : 0u], vc = src[bc ? kc : 0u], vd = src[bd ? kd : 0u];
                if (p.store_policy == 0) {
                    if (ba) stream_store(dst + ka, va);
                    if (bb) stream_store(dst + kb, vb);
                    if (bc) stream_store(dst + kc, vc);
                    if (bd) stream_store(dst + kd, vd);
                } else if (p.store_policy == 1) {
                    if (ba) dst[ka] = va;
                    if (bb) dst[kb] = vb;
                    if (bc) dst[kc] = vc;
                    if (bd) dst[kd] = vd;
                } else {
                    if (ba) nt_store(dst + ka, va);
                    if (bb) nt_store(dst + kb, vb);
                    if (bc) nt_store(dst + kc, vc);
                    if (bd) nt_store(dst + kd, vd);
                }
            }
        }
        const uint32_t done_bytes = head + (body << 4);
        if (lane < nbytes - done_bytes) g[done_bytes + lane] = tile[done_bytes + lane];
        wave_lds_sync();
        if (p.steady) {
#pragma unroll
            for (uint32_t j = 0; j < kRowsPerLane; j++) {
                if (occupied[j]) {
                    tile[mark_a[j]] = 0;
                    tile[mark_b[j]] = 0;
                }
            }
        }
        STAMP(5 + min(first_world / max(p.wpp, 1u), 8u));
    }
    store_state();
    STAMP(15);
    STAMP_REALTIME(14);
}

// ---------------------------------------------------------------------------------------------
// Few worlds of a very large layout (p.share): the four waves of a workgroup work on ONE world.  In step_body each of them
// keeps its own copy of the world in LDS and steps it redundantly; that copy (cells, players, cell -> player map, the
// 16-byte tail of every cell: 6-7 KB) times four is what decides how many workgroups fit a CU, and at the reference's
// 1000-environment table the launch then needs a second generation of workgroups: many_player_layout with 4 players
// takes 12.7 us per step at 768 worlds and 18.3 at 800 (three 45 KB workgroups per CU); with 8 players 15.9 at 512
// and 25.7 at 600.  Here the workgroup keeps ONE copy: all 256 threads bring the cells in, wave 0 steps the world (lane =
// player), all threads work out the cells' tails, then every wave assembles and streams every fourth pass of rows from
// its own tile as before.  Three workgroup barriers; 25 KB (4 players) / 36 KB (8) of LDS per workgroup.
// ---------------------------------------------------------------------------------------------
template <bool kInit>
__device__ __forceinline__ void team_body(const StepParams &p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    constexpr int kConstWordsPerThread = (kConstBytes / 4 + kBlock - 1) / kBlock;
    uint32_t const_word[kConstWordsPerThread];
#pragma unroll
    for (int j = 0; j < kConstWordsPerThread; j++) const_word[j] = tid + j * kBlock < kConstBytes / 4 ? p.consts[tid + j * kBlock] : 0u;
    const uint8_t *s_terrain = smem + kConstTerrain;
    const uint8_t *s_start = smem + kConstStart;
    const uint8_t *s_pots = smem + kConstPots;
    const uint32_t logical_block = (blockIdx.x & 7u) * p.per_xcd + (blockIdx.x >> 3);
    const uint32_t w0 = logical_block;  // the workgroup's world
    const uint32_t P = p.P, C = p.C, N = p.num_worlds, F = p.F, shift = 5 * P;
    if (w0 >= N) return;  // the whole workgroup (the grid is rounded up to a multiple of eight)
    uint8_t *base = smem + kConstBytes;
    uint32_t *s_obj = reinterpret_cast<uint32_t *>(base);              // [C]
    uint32_t *s_pl = reinterpret_cast<uint32_t *>(base + p.off_pl);    // [P][2]
    uint32_t *s_x = reinterpret_cast<uint32_t *>(base + p.off_x);      // transition scratch (wave 0)
    uint32_t *s_sum = reinterpret_cast<uint32_t *>(base + p.off_sum);
    uint32_t *s_blk = s_sum + 1;
    uint8_t *s_cur = base + p.off_cur;                                 // [C] cell -> player
    uint8_t *s_flags = base + p.off_flags;
    uint4 *s_tail = reinterpret_cast<uint4 *>(base + p.off_tail);      // [C]
    uint8_t *s_tile = base + p.off_tile + wib * p.lds_wave_stride;     // this wave's row tile

    // ---- load: the cells by all threads, the players by wave 0 ----
    const bool mine = wib == 0 && lane < P;  // lane = player
    uint32_t posori = 0, held = kItemNone, act = A_STAY;
    int32_t t_loaded = 0;
    if (!kInit) {
        const uint32_t *g_obj = p.cell_obj + (size_t)w0 * C;
        const uint32_t cell_reg = tid < C ? g_obj[tid] : 0u;  // C <= 255 < kBlock
        uint2 pl_reg = make_uint2(0, 0);
        uint32_t a_raw = A_STAY;
        if (mine) {
            pl_reg = p.players[(size_t)w0 * P + lane];
            a_raw = load_action(p, (size_t)lane * N + w0);
        }
        if (wib == 0) t_loaded = p.timestep[w0];
        if (tid < C) s_obj[tid] = cell_reg;
        if (mine) {
            posori = pl_reg.x & 0xFFFFu;
            held = pl_reg.y;
            act = a_raw <= A_INTERACT ? a_raw : (uint32_t)A_STAY;
            if (p.actions64) p.action_mirror[(size_t)lane * N + w0] = (int32_t)a_raw;
        }
    }
    if (tid < (C + 3u) >> 2) reinterpret_cast<uint32_t *>(s_cur)[tid] = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < kConstWordsPerThread; j++)
        if (tid + j * kBlock < kConstBytes / 4) reinterpret_cast<uint32_t *>(smem)[tid + j * kBlock] = const_word[j];
    __syncthreads();

    // ---- step: wave 0, lane = player ----
    int32_t reward_world = 0, t = 0;
    bool reset_now = kInit;
    if (wib == 0) {
        if (!kInit) {
            transition_lanes<0>(p, s_terrain, s_obj, s_x, s_sum, s_blk, P, lane, mine, 0u, lane, act, posori, held, reward_world);
            tick_pots(p, s_pots, s_obj, 1u, lane);
            t = t_loaded + 1;
            reset_now = (int64_t)t >= p.horizon;
        }
        if (reset_now) {  // wave-uniform: one world
            t = 0;
            posori = (uint32_t)s_start[mine ? lane : 0u] | (A_NORTH << 8);
            held = kItemNone;
            for (uint32_t i = lane; i < C; i += kWave) s_obj[i] = kItemNone;
        }
        if (mine) {
            reinterpret_cast<uint2 *>(s_pl)[lane] = make_uint2(posori, held);
            s_cur[posori & 0xFFu] = (uint8_t)lane;
        }
        if (lane == 0) s_flags[0] = (p.horizon - (int64_t)t < 40) ? 1 : 0;
    }
    __syncthreads();

    // ---- the cells' viewer-independent tails: all threads ----
    if (tid < C) {
        const uint32_t who = s_cur[tid];
        const uint32_t h = who != 0xFF ? s_pl[who * 2 + 1] : kItemNone;
        s_tail[tid] = cell_tail(p, s_terrain[tid], s_obj[tid], h, s_flags[0]);
    }
    __syncthreads();

    // ---- rows: every fourth pass of kRowsPerPass rows per wave (as step_body's multi-pass loop, one world) ----
    constexpr uint32_t kRowsPerLane = kRowsPerPass / kWave;
    const uint32_t total_rows = p.rows;
    uint8_t *gobs = p.obs + (size_t)w0 * p.block_bytes;
    for (uint32_t r0 = wib * kRowsPerPass; r0 < total_rows; r0 += kWavesPerBlock * kRowsPerPass) {
        const uint32_t nrows = min((uint32_t)kRowsPerPass, total_rows - r0);
        const uint32_t nbytes = nrows * F;
        uint8_t *g = gobs + (size_t)r0 * F;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(g) & 15u);  // tile and slab congruent mod 16
        uint8_t *tile = s_tile + mis;
        const uint32_t nchunks = (mis + nbytes + 15u) >> 4;
        for (uint32_t k = lane; k < nchunks; k += kWave) reinterpret_cast<uint4 *>(s_tile)[k] = make_uint4(0, 0, 0, 0);
        wave_lds_sync();
        bool valid[kRowsPerLane], occupied[kRowsPerLane];
        uint32_t cell[kRowsPerLane], viewer[kRowsPerLane], who[kRowsPerLane], ori[kRowsPerLane];
        uint4 tl[kRowsPerLane];
#pragma unroll
        for (uint32_t j = 0; j < kRowsPerLane; j++) {
            const uint32_t tr = lane * kRowsPerLane + j;
            valid[j] = tr < nrows;
            const uint32_t r = r0 + tr;
            viewer[j] = __umulhi(r, p.inv_c);
            cell[j] = valid[j] ? r - __umul24(viewer[j], C) : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < kRowsPerLane; j++) {
            tl[j] = s_tail[cell[j]];
            who[j] = s_cur[cell[j]];
        }
#pragma unroll
        for (uint32_t j = 0; j < kRowsPerLane; j++) {
            occupied[j] = valid[j] && who[j] != 0xFF;
            ori[j] = (s_pl[(occupied[j] ? who[j] : 0u) * 2] >> 8) & 0xFF;
        }
#pragma unroll
        for (uint32_t j = 0; j < kRowsPerLane; j++) {
            const uint32_t row = __umul24(lane * kRowsPerLane + j, F);
            const uint32_t cls = p.tail_even ? ((mis + j * F + shift) & 3u) : 1u;
            if (valid[j]) lds_store_tail(tile + row + shift, tl[j], cls);
            const uint32_t rel = who[j] == viewer[j] ? 0u : (who[j] < viewer[j] ? who[j] + 1u : who[j]);
            if (occupied[j]) {
                tile[row + rel] = 1;
                tile[row + P + 4 * rel + ori[j]] = 1;
            }
        }
        wave_lds_sync();
        const uint32_t head = min((16u - mis) & 15u, nbytes);
        if (lane < head) g[lane] = tile[lane];
        const uint32_t body = (nbytes - head) >> 4;
        const uint4 *src = reinterpret_cast<const uint4 *>(tile + head);
        uint4 *dst = reinterpret_cast<uint4 *>(g + head);
        for (uint32_t k0 = lane; k0 < body + lane; k0 += 4 * kWave) {
            const uint32_t ka = k0, kb = k0 + kWave, kc = k0 + 2 * kWave, kd = k0 + 3 * kWave;
            const bool ba = ka < body, bb = kb < body, bc = kc < body, bd = kd < body;
            const uint4 va = src[ba ? ka : 0u], vb = src[bb ? kb : 0u], vc = src[bc ? kc : 0u], vd = src[bd ? kd : 0u];
            if (p.store_policy == 0) {
                if (ba) stream_store(dst + ka, va);
                if (bb) stream_store(dst + kb, vb);
                if (bc) stream_store(dst + kc, vc);
                if (bd) stream_store(dst + kd, vd);
            } else if (p.store_policy == 1) {
                if (ba) dst[ka] = va;
                if (bb) dst[kb] = vb;
                if (bc) dst[kc] = vc;
                if (bd) dst[kd] = vd;
            } else {
                if (ba) nt_store(dst + ka, va);
                if (bb) nt_store(dst + kb, vb);
                if (bc) nt_store(dst + kc, vc);
                if (bd) nt_store(dst + kd, vd);
            }
        }
        const uint32_t done_bytes = head + (body << 4);
        if (lane < nbytes - done_bytes) g[done_bytes + lane] = tile[done_bytes + lane];
        wave_lds_sync();
    }

    // ---- store: cells by all threads, the rest by wave 0 ----
    if (tid < C) p.cell_obj[(size_t)w0 * C + tid] = s_obj[tid];
    if (mine) {
        p.players[(size_t)w0 * P + lane] = make_uint2(posori, held);
        p.reward[(size_t)lane * N + w0] = reward_world;
        if (lane == 0) {
            p.timestep[w0] = t;
            p.done[w0] = kInit ? 0 : (int32_t)reset_now;
        }
    }
}

template <bool kInit>
__global__ void __launch_bounds__(kBlock) mrl_overcooked_step_team(const StepParams p)
{
    team_body<kInit>(p);
}

template <bool kInit, int kP, bool kPlain = false>
__global__ void __launch_bounds__(kBlock) mrl_overcooked_step(const StepParams p)
{
    step_body<kInit, kP, kPlain>(p, blockIdx.x);
}

// Several simulators -- any mix of layouts, sizes and player counts -- stepped by ONE launch (mrl_step_many): the grid is
// the concatenation of their grids, a workgroup finds the simulator it belongs to and runs that simulator's step on
// that simulator's parameters, which travel by value in the kernel arguments (328 bytes each: no table in device memory
// to keep up to date).  The step itself is the generic one -- a kernel specialised for one layout size cannot serve
// several -- so a big single-layout batch is better off with its own launch; this is for many small sub-batches, where
// the launches, not the worlds, are what costs.
constexpr int kMaxMany = 8;
struct ManyParams {
    StepParams sim[kMaxMany];
    uint32_t first_block[kMaxMany + 1];  // workgroups of simulator k: first_block[k] .. first_block[k + 1]
    uint32_t count;
};
static_assert(sizeof(ManyParams) <= 3072, "the parameters of a many-simulator launch travel in the kernel argument segment (4 KB)");

__global__ void __launch_bounds__(kBlock) mrl_overcooked_step_many(const ManyParams m)
{
    uint32_t k = 0;
#pragma unroll
    for (int j = 1; j < kMaxMany; j++) k += ((uint32_t)j < m.count && blockIdx.x >= m.first_block[j]) ? 1u : 0u;
    const StepParams &p = m.sim[k];
    const uint32_t block = blockIdx.x - m.first_block[k];
    if (p.P == 2u && p.pair_exchange)
        step_body<false, 2, false>(p, block);
    else
        step_body<false, 0, false>(p, block);
}

// Per-wave LDS offsets of the single-pass, two-player, private-constants configuration: the same formulas as
// the host's layout() below, as a constexpr so that a kernel specialised for one layout size can fold them.
struct FixedLayout {
    uint32_t off_pl, off_x, off_sum, off_cur, off_flags, off_terr, off_list, off_tail, off_tile, stride;
};
constexpr uint32_t up16c(uint32_t v) { return (v + 15u) & ~15u; }
constexpr FixedLayout fixed_layout(uint32_t C, uint32_t wpw)
{
    FixedLayout f{};
    const uint32_t P = 2, F = 5 * P + 16;
    f.off_pl = up16c(wpw * C * 4);
    f.off_x = f.off_pl + up16c(wpw * P * 8);
    f.off_sum = f.off_x;
    f.off_cur = f.off_sum + up16c(2u * wpw * 4u);
    f.off_flags = f.off_cur + up16c(wpw * C);
    f.off_terr = f.off_flags + 64u;
    f.off_list = f.off_terr + up16c(C);
    f.off_tail = f.off_list + up16c(wpw * C * 2u);
    f.off_tile = f.off_tail;
    f.stride = f.off_tile + ((wpw * P * C * F + 255u) & ~255u);
    return f;
}

// size of a group's holder table (host: hold_table): kHold holder cells per world, round 0 keeps 2 * kW lanes for the players
constexpr uint32_t fixed_hold_entries(uint32_t wpw, uint32_t holders)
{
    const uint32_t free0 = 64u - 2u * wpw, total = wpw * holders;
    return total == 0 ? 1u : (total <= free0 ? 2u * wpw + total : 64u + (total - free0));
}

// The step for ONE layout size known at compile time (two players, kC cells, kW worlds per wave, single-pass encode,
// private constants): every size-dependent kernel argument is replaced by a constant, so divisions by the cell count,
// loop trip counts and LDS offsets fold (the kernel is bound by instruction issue, not by bytes: section 4.1 of
// DESIGN.md).  The host launches it only when the simulator's parameters are exactly these; results are identical.
template <int kC, int kW, int kWidth, int kPots, int kHold>
__device__ __forceinline__ StepParams fixed_params(const StepParams &p)
{
    constexpr FixedLayout f = fixed_layout(kC, kW);
    StepParams q = p;
    q.P = 2;
    q.C = kC;
    q.W = kWidth;
    q.F = 26;
    q.deltas = (uint64_t)(uint8_t)(int8_t)(-kWidth) | ((uint64_t)(uint8_t)(int8_t)kWidth << 8) | (1ull << 16) | (0xFFull << 24);
    q.num_pots = kPots;
    q.rows = 2 * kC;
    q.block_bytes = 2 * kC * 26;
    q.inv_c = (uint32_t)((1ull << 32) / (uint64_t)kC) + 1u;
    q.inv_p = (uint32_t)((1ull << 32) / 2ull) + 1u;
    q.inv_rows = (uint32_t)((1ull << 32) / (uint64_t)(2 * kC)) + 1u;
    q.wpw = kW;
    q.wpp = 0;
    q.whole = 1;
    q.patch = 1;
    q.share = 0;
    q.steady = 0;
    q.tail_even = 1;
    q.private_consts = 1;
    q.terr_entries = kW * 2 * kC;
    q.direct = 1;
    q.unaligned = 0;
    q.hold_entries = fixed_hold_entries(kW, kHold);
    q.off_pl = f.off_pl;
    q.off_x = f.off_x;
    q.off_sum = f.off_sum;
    q.off_cur = f.off_cur;
    q.off_flags = f.off_flags;
    q.off_terr = f.off_terr;
    q.off_list = f.off_list;
    q.off_tail = f.off_tail;
    q.off_tile = f.off_tile;
    q.lds_wave_stride = f.stride;
    return q;
}

// kI64: the launch of mrl_step_with_actions_i64 (the int32 one drops the branch on the action pointer too)
// The first fourteen dwords of the argument list (sixteen user SGPRs less the argument-segment pointer) are what a
// wave needs to find its worlds and request its loads; written as separate scalar arguments they are PRELOADED into
// SGPRs by the command processor (-mllvm -amdgpu-kernarg-preload-count=16 in the Makefile; a by-value struct is not
// eligible), so the first global loads do not wait for a scalar load of the argument segment.  The struct carries
// everything else (and the same fields again, unused).
#define MRL_HOT_ARGS                                                                                                      \
    uint32_t *hot_cell_obj, uint2 *hot_players, int32_t *hot_timestep, const void *hot_actions, const uint32_t *hot_consts, \
        const uint16_t *hot_terr_off, uint32_t hot_num_worlds, uint32_t hot_per_xcd
template <bool kI64>
__device__ __forceinline__ void take_hot_args(StepParams &q, MRL_HOT_ARGS)
{
    q.cell_obj = hot_cell_obj;
    q.players = hot_players;
    q.timestep = hot_timestep;
    q.actions = kI64 ? nullptr : static_cast<const int32_t *>(hot_actions);
    q.actions64 = kI64 ? static_cast<const long long *>(hot_actions) : nullptr;
    q.consts = hot_consts;
    q.terr_off = hot_terr_off;
    q.num_worlds = hot_num_worlds;
    q.per_xcd = hot_per_xcd;
}
#define MRL_HOT_PASS hot_cell_obj, hot_players, hot_timestep, hot_actions, hot_consts, hot_terr_off, hot_num_worlds, hot_per_xcd

template <int kC, int kW, int kWidth, int kPots, int kHold, bool kI64 = false, bool kPlain = false>
__global__ void __launch_bounds__(kBlock) mrl_overcooked_step_fixed(MRL_HOT_ARGS, const StepParams p)
{
    StepParams q = fixed_params<kC, kW, kWidth, kPots, kHold>(p);
    take_hot_args<kI64>(q, MRL_HOT_PASS);
    step_body<false, 2, kPlain>(q, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Random-policy rollout on the device: num_steps steps in ONE launch.
// The reference's random-policy harness draws `randint(high=6)` per agent per step with torch and
// copies it into the action tensor (scripts/overcooked_example.py:99-106); here the draw happens
// in the kernel and the group's state stays in LDS between steps, so a step costs neither a launch
// nor a state round trip through HBM -- every step still writes its full observation slab,
// rewards and done flags (the next step overwrites them, as in the reference's harness).
//
// Action of (world w, player q) at global step index k:  mrl_random_action(seed, k, w, q) below --
// a counter-based hash, so any step of the stream can be recomputed (tests replay the stream
// through the ordinary step and compare).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t mrl_random_action(uint64_t seed, uint32_t step, uint32_t world, uint32_t player)
{
    return mrl::scale(mrl::policy_hash(seed, step, world, player), 6u);  // uniform over the six actions
}

template <int kP, bool kPlain = false>
__device__ __forceinline__ void rollout_body(const StepParams &p, uint32_t num_steps, uint64_t seed, uint32_t first_step,
                                             int32_t *action_out, const int32_t *action_seq)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    constexpr int kConstWordsPerThread = (kConstBytes / 4 + kBlock - 1) / kBlock;
    uint32_t const_word[kConstWordsPerThread];
    const bool private_consts = p.private_consts != 0;  // see mrl_overcooked_step
    if (private_consts) {
        const __amdgpu_buffer_rsrc_t r_terr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.consts), 0, (int)((p.C + 3u) & ~3u), 0x00020000);
        const_word[0] = __builtin_amdgcn_raw_buffer_load_b32(r_terr, (int)(lane * 4u), 0, 0);
    } else {
#pragma unroll
        for (int j = 0; j < kConstWordsPerThread; j++) const_word[j] = tid + j * kBlock < kConstBytes / 4 ? p.consts[tid + j * kBlock] : 0u;
    }
    const uint8_t *s_terrain = private_consts ? smem + kConstBytes + wib * p.lds_wave_stride + p.off_terr : smem + kConstTerrain;
    const uint8_t *s_start = smem + kConstStart;
    const uint8_t *s_pots = smem + kConstPots;
    const uint32_t per_xcd = p.per_xcd;
    const uint32_t logical_block = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const uint32_t w0 = (logical_block * kWavesPerBlock + wib) * p.wpw;
    const uint32_t nw = w0 < p.num_worlds ? min(p.wpw, p.num_worlds - w0) : 0u;
    uint8_t *wbase = smem + kConstBytes + wib * p.lds_wave_stride;
    uint32_t *s_obj = reinterpret_cast<uint32_t *>(wbase);
    uint32_t *s_pl = reinterpret_cast<uint32_t *>(wbase + p.off_pl);
    uint32_t *s_x = reinterpret_cast<uint32_t *>(wbase + p.off_x);
    uint32_t *s_sum = reinterpret_cast<uint32_t *>(wbase + p.off_sum);
    uint32_t *s_blk = s_sum + p.wpw;
    uint8_t *s_cur = wbase + p.off_cur;
    uint8_t *s_flags = wbase + p.off_flags;
    uint16_t *s_list = reinterpret_cast<uint16_t *>(wbase + p.off_list);
    uint8_t *s_tile = wbase + p.off_tile;
    const uint32_t P = kP > 0 ? (uint32_t)kP : p.P, C = p.C, N = p.num_worlds;
    const uint32_t ncells = nw * C, nplayers = nw * P;
    const uint32_t wl = kP == 2 ? lane >> 1 : (P == 1u ? lane : __umulhi(lane, p.inv_p));
    const uint32_t q = lane - wl * P;
    const bool active = lane < nplayers;
    const uint32_t world = min(w0 + wl, N - 1u);

    // the lane's player and its world's clock live in registers for the whole rollout; cell objects in LDS
    uint32_t posori = 0, held = kItemNone;
    int32_t t = 0;
    uint8_t *s_prev = s_flags + 32;  // urgency flags as the tile holds them (wpw * 2 <= 64 players: wpw <= 32)
    HoldTab hold{};
    const uint32_t patch_mis = p.unaligned ? (uint32_t)(reinterpret_cast<uintptr_t>(p.obs + (size_t)w0 * p.block_bytes) & 15u) : 0u;
    {
        const uint32_t *g_obj = p.cell_obj + (size_t)w0 * C;
        const uint2 pl_reg = p.players[(size_t)w0 * P + (active ? lane : 0u)];
        t = p.timestep[world];
        TerrPos tpos;
        if (p.direct) hold_request(p, lane, hold);
        if (p.patch) {
            terrain_request(p, lane, tpos);
            tile_zero_addtid(s_tile, nw * p.block_bytes + patch_mis);
            terrain_deliver(p, tpos, s_tile + patch_mis, nw);
            if (lane < 32) s_prev[lane] = 0;
        }
        for (uint32_t i = lane; i < ncells; i += kWave) s_obj[i] = g_obj[i];
        if (!p.direct)
            for (uint32_t i = lane; i < (p.wpw * C + 3u) >> 2; i += kWave) reinterpret_cast<uint32_t *>(s_cur)[i] = 0xFFFFFFFFu;
        if (active) {
            posori = pl_reg.x & 0xFFFFu;
            held = pl_reg.y;
        }
    }
    if (private_consts) {
        if (lane * 4u < p.C) reinterpret_cast<uint32_t *>(const_cast<uint8_t *>(s_terrain))[lane] = const_word[0];
        wave_lds_sync();
    } else {
#pragma unroll
        for (int j = 0; j < kConstWordsPerThread; j++)
            if (tid + j * kBlock < kConstBytes / 4) reinterpret_cast<uint32_t *>(smem)[tid + j * kBlock] = const_word[j];
        __syncthreads();
    }
    if (nw == 0) return;
    if (active && !p.direct) s_cur[wl * C + (posori & 0xFFu)] = (uint8_t)q;

    // mrl_step_sequence: the next step's action is requested before this step's encode, so its latency is
    // not in the step-to-step chain
    int32_t ahead = 0;
    if (action_seq) ahead = action_seq[(size_t)(active ? q : 0u) * N + world];
    StepParams ps = p;  // the step's view of the parameters: its observation slot when the output is a ring of slots
    uint32_t slot = p.ring_first;
    for (uint32_t k = 0; k < num_steps; k++) {
        if (p.ring_slots > 1u) {
            ps.obs = p.obs + (size_t)slot * p.ring_stride;
            slot = slot + 1u == p.ring_slots ? 0u : slot + 1u;
        }
        uint32_t a;
        if (action_seq) {  // step k's actions from the caller's (num_steps, P, N) array
            a = (uint32_t)ahead;
            a = a <= A_INTERACT ? a : (uint32_t)A_STAY;
            if (k + 1 < num_steps) ahead = action_seq[((size_t)(k + 1) * P + (active ? q : 0u)) * N + world];
        } else {
            a = mrl_random_action(seed, first_step + k, world, q);
            if (active && k + 1 == num_steps) action_out[(size_t)q * N + world] = (int32_t)a;  // the ACTION tensor shows the last draw
        }
        if (!active) a = A_STAY;
        const uint32_t old_cell = wl * C + (posori & 0xFFu);
        int32_t reward_world = 0;
        transition_lanes<kP>(p, s_terrain, s_obj + wl * C, s_x, s_sum, s_blk, P, lane, active, wl, q, a, posori, held, reward_world);
        if (!p.direct) tick_pots(p, s_pots, s_obj, nw, lane);  // direct: the holder lanes of the encode tick the pots
        t += 1;
        const bool reset_now = (int64_t)t >= p.horizon;
        if (__builtin_expect(__ballot(active && reset_now) != 0ull, 0)) {
            if (reset_now) {
                t = 0;
                posori = (private_consts ? (p.starts_w >> (8u * (q & 3u))) & 0xFFu : (uint32_t)s_start[active ? q : 0u]) | (A_NORTH << 8);
                held = kItemNone;
            }
            if (active && q == 0) s_sum[wl] = reset_now ? 1u : 0u;
            wave_lds_sync();
            for (uint32_t i = lane; i < ncells; i += kWave)
                if (s_sum[__umulhi(i, p.inv_c)] != 0u) s_obj[i] = kItemNone;
            wave_lds_sync();
        }
        // every lane clears its old cell before any lane marks its new one (two DS instructions, in order)
        if (active && !p.direct) s_cur[old_cell] = 0xFF;
        wave_lds_sync();
        if (active) {
            if (!p.direct) {
                reinterpret_cast<uint2 *>(s_pl)[lane] = make_uint2(posori, held);
                s_cur[wl * C + (posori & 0xFFu)] = (uint8_t)q;
            }
            if (q == 0) s_flags[wl] = (p.horizon - (int64_t)t < 40) ? 1 : 0;
        }
        wave_lds_sync();
        if (p.patch) {
            // the tile lives as long as the rollout: patch what is dynamic now, stream, put the static rows back
            const uint32_t ndyn = p.direct ? 0u : find_dynamic(p, s_obj, s_cur, s_list, 0, nw, lane);
            observe_patch<kP, true, kPlain>(ps, s_terrain, s_obj, s_pl, s_cur, s_flags, s_prev, s_list, ndyn, s_tile + patch_mis, P, w0, 0, nw, lane, hold,
                                             active, wl, q, posori, held);
        } else {
            observe_whole<kPlain>(ps, s_terrain, s_obj, s_pl, s_cur, s_flags, s_tile, P, w0, nw, lane);
        }
        if (active) {  // after the stream-out, like the state stores of the single step
            p.reward[(size_t)q * N + world] = reward_world;
            if (q == 0) p.done[world] = (int32_t)reset_now;
        }
        wave_lds_sync();
    }
    // state back to HBM once, after the last step
    uint32_t *g_obj = p.cell_obj + (size_t)w0 * C;
    for (uint32_t i = lane; i < ncells; i += kWave) g_obj[i] = s_obj[i];
    if (active) {
        p.players[(size_t)w0 * P + lane] = make_uint2(posori, held);
        if (q == 0) p.timestep[world] = t;
    }
}

template <int kP, bool kPlain = false>
__global__ void __launch_bounds__(kWavesPerBlock * kWave) mrl_overcooked_rollout(const StepParams p, uint32_t num_steps, uint64_t seed,
                                                                                 uint32_t first_step, int32_t *action_out,
                                                                                 const int32_t *action_seq)
{
    rollout_body<kP, kPlain>(p, num_steps, seed, first_step, action_out, action_seq);
}

// the multi-step launches for one layout size known at compile time (see mrl_overcooked_step_fixed)
template <int kC, int kW, int kWidth, int kPots, int kHold, bool kPlain = false>
__global__ void __launch_bounds__(kWavesPerBlock * kWave) mrl_overcooked_rollout_fixed(const StepParams p, uint32_t num_steps, uint64_t seed,
                                                                                       uint32_t first_step, int32_t *action_out,
                                                                                       const int32_t *action_seq)
{
    StepParams q = fixed_params<kC, kW, kWidth, kPots, kHold>(p);
    // with the tile and the cell -> player map alive across steps the search costs a multi-step launch little: the table
    // only where it is one round (measured, cramped_room 16 worlds per wave: 4.46 us per step searched, 4.72 with two rounds)
    q.direct = fixed_hold_entries(kW, kHold) <= 64u ? 1u : 0u;
    rollout_body<2, kPlain>(q, num_steps, seed, first_step, action_out, action_seq);
}

// ---------------------------------------------------------------------------------------------
// One step of kG consecutive groups per wave (single-pass, two-player, private-constants configurations).
// The ordinary launch gives every group its own wave: 4096 waves at 32768 worlds, all resident at once, all in the
// same phase at the same time, and nothing leaves the chip before a wave has gone through load latency, transition
// and patching.  Here a wave steps group A, streams it out, and steps group B while A's stores drain: B's state was
// requested together with A's (one latency for both), the observation tile is zeroed and given its terrain bytes
// once -- after A's stream-out the patched rows are put back to their static content, as in the multi-step
// launches -- and with half as many waves per SIMD a wave's phases take half as long, so the first stores leave
// earlier.  Same state and output arrays, same results.
// ---------------------------------------------------------------------------------------------
template <int kP, int kG, bool kPlain = false>
__device__ __forceinline__ void groups_body(const StepParams &p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    STAMP(0);
    STAMP_REALTIME(13);
    const __amdgpu_buffer_rsrc_t r_terr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.consts), 0, (int)((p.C + 3u) & ~3u), 0x00020000);
    const uint32_t terr_word = __builtin_amdgcn_raw_buffer_load_b32(r_terr, (int)(lane * 4u), 0, 0);
    uint8_t *wbase = smem + kConstBytes + wib * p.lds_wave_stride;
    const uint8_t *s_terrain = wbase + p.off_terr;
    const uint8_t *s_pots = smem + kConstPots;
    const uint32_t per_xcd = p.per_xcd;
    const uint32_t logical_block = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const uint32_t first_world = (logical_block * kWavesPerBlock + wib) * p.wpw * kG;  // this wave's kG groups are consecutive
    uint32_t *s_obj = reinterpret_cast<uint32_t *>(wbase);
    uint32_t *s_pl = reinterpret_cast<uint32_t *>(wbase + p.off_pl);
    uint32_t *s_x = reinterpret_cast<uint32_t *>(wbase + p.off_x);
    uint32_t *s_sum = reinterpret_cast<uint32_t *>(wbase + p.off_sum);
    uint32_t *s_blk = s_sum + p.wpw;
    uint8_t *s_cur = wbase + p.off_cur;
    uint8_t *s_flags = wbase + p.off_flags;
    uint8_t *s_prev = s_flags + 32;  // urgency flags as the tile holds them
    uint16_t *s_list = reinterpret_cast<uint16_t *>(wbase + p.off_list);
    uint8_t *s_tile = wbase + p.off_tile;
    const uint32_t P = (uint32_t)kP, C = p.C, N = p.num_worlds;
    const uint32_t wl = kP == 2 ? lane >> 1 : (P == 1u ? lane : __umulhi(lane, p.inv_p));
    const uint32_t q = lane - wl * P;

    // every group's state slab is requested up front: one HBM / Infinity Cache latency for all of them
    constexpr int kBatch = 4;
    uint32_t cell_reg[kG][kBatch];
    uint2 pl_reg[kG];
    uint32_t a_raw[kG];
    int32_t t_reg[kG];
    // Worlds in each of the wave's groups, worked out ONCE and kept opaque (readfirstlane): re-deriving them from N
    // inside the unrolled group loop let hipcc fold "this group is empty" (w0 >= N) into w0 == N and run a wave's
    // second group past the end of the batch when the batch ends inside its first one (found by the ragged-batch test).
    uint32_t nw_of[kG];
#pragma unroll
    for (int g = 0; g < kG; g++) {
        const uint32_t w0 = first_world + (uint32_t)g * p.wpw;
        nw_of[g] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(w0 < N ? min(p.wpw, N - w0) : 0u));
    }
#pragma unroll
    for (int g = 0; g < kG; g++) {
        const uint32_t w0 = first_world + (uint32_t)g * p.wpw;
        const uint32_t nw = nw_of[g];
        const uint32_t wc = min(w0, N - 1u);  // empty groups read nothing (zero-sized descriptors) from a valid address
        const __amdgpu_buffer_rsrc_t r_obj = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.cell_obj + (size_t)wc * C), 0, (int)(nw * C * 4u), 0x00020000);
#pragma unroll
        for (int k = 0; k < kBatch; k++) cell_reg[g][k] = __builtin_amdgcn_raw_buffer_load_b32(r_obj, (int)((lane + (uint32_t)k * kWave) * 4u), 0, 0);
        const __amdgpu_buffer_rsrc_t r_pl = __builtin_amdgcn_make_buffer_rsrc(p.players + (size_t)wc * P, 0, (int)(nw * P * 8u), 0x00020000);
        const auto raw = __builtin_amdgcn_raw_buffer_load_b64(r_pl, (int)(lane * 8u), 0, 0);
        pl_reg[g] = make_uint2(raw[0], raw[1]);
        const size_t a_at = (size_t)(lane < nw * P ? q : 0u) * N + min(w0 + wl, N - 1u);
        a_raw[g] = load_action(p, a_at);
        t_reg[g] = p.timestep[min(w0 + wl, N - 1u)];
    }
    HoldTab hold{};
    {
        const uint32_t nw0 = nw_of[0];  // the first group is the wave's fullest
        TerrPos tpos;
        terrain_request(p, lane, tpos);
        if (p.direct) hold_request(p, lane, hold);
        if (!p.direct)
            for (uint32_t i = lane; i < (p.wpw * C + 3u) >> 2; i += kWave) reinterpret_cast<uint32_t *>(s_cur)[i] = 0xFFFFFFFFu;
        tile_zero_addtid(s_tile, nw0 * p.block_bytes);
        if (lane < 32) s_prev[lane] = 0;
        STAMP(6);
        terrain_deliver(p, tpos, s_tile, nw0);
        if (lane * 4u < p.C) reinterpret_cast<uint32_t *>(const_cast<uint8_t *>(s_terrain))[lane] = terr_word;
        STAMP(7);
        wave_lds_sync();
    }
#pragma unroll
    for (int g = 0; g < kG; g++) {
        const uint32_t w0 = first_world + (uint32_t)g * p.wpw;
        const uint32_t nw = nw_of[g];
        if (nw == 0) break;  // wave-uniform; later groups are empty too
        const uint32_t ncells = nw * C, nplayers = nw * P;
        const bool active = lane < nplayers;
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const uint32_t i = lane + k * kWave;
            if (i < ncells) s_obj[i] = cell_reg[g][k];
        }
        for (uint32_t i = lane + kBatch * kWave; i < ncells; i += kWave) s_obj[i] = p.cell_obj[(size_t)w0 * C + i];
        uint32_t posori = active ? pl_reg[g].x & 0xFFFFu : 0u, held = active ? pl_reg[g].y : (uint32_t)kItemNone;
        const uint32_t act = (active && a_raw[g] <= A_INTERACT) ? a_raw[g] : (uint32_t)A_STAY;
        if (p.actions64 && active) p.action_mirror[(size_t)q * N + w0 + wl] = (int32_t)a_raw[g];
        wave_lds_sync();
        if (g == 0) STAMP(1);
        int32_t reward_world = 0;
        transition_lanes<kP>(p, s_terrain, s_obj + wl * C, s_x, s_sum, s_blk, P, lane, active, wl, q, act, posori, held, reward_world);
        if (!p.direct) tick_pots(p, s_pots, s_obj, nw, lane);  // direct: the holder lanes of the encode tick the pots
        int32_t t = t_reg[g] + 1;
        const bool reset_now = (int64_t)t >= p.horizon;
        if (__builtin_expect(__ballot(active && reset_now) != 0ull, 0)) {
            if (reset_now) {
                t = 0;
                posori = ((p.starts_w >> (8u * (q & 3u))) & 0xFFu) | (A_NORTH << 8);
                held = kItemNone;
            }
            if (active && q == 0) s_sum[wl] = reset_now ? 1u : 0u;
            wave_lds_sync();
            for (uint32_t i = lane; i < ncells; i += kWave)
                if (s_sum[__umulhi(i, p.inv_c)] != 0u) s_obj[i] = kItemNone;
            wave_lds_sync();
        }
        if (g == 0) STAMP(2);
        if (active) {
            if (!p.direct) {
                reinterpret_cast<uint2 *>(s_pl)[lane] = make_uint2(posori, held);
                s_cur[wl * C + (posori & 0xFFu)] = (uint8_t)q;
            }
            if (q == 0) s_flags[wl] = (p.horizon - (int64_t)t < 40) ? 1 : 0;
        }
        wave_lds_sync();
        if (g == 0) STAMP(3);
        const uint32_t ndyn = p.direct ? 0u : find_dynamic(p, s_obj, s_cur, s_list, 0, nw, lane);
        if (g == 0) STAMP(4);
        observe_patch<kP, true, kPlain>(p, s_terrain, s_obj, s_pl, s_cur, s_flags, s_prev, s_list, ndyn, s_tile, P, w0, 0, nw, lane, hold, active, wl,
                                        q, posori, held);
        if (g == 0) STAMP(5);
        // the group's state, rewards and flags: behind its stream-out, like the ordinary step
        uint32_t *g_obj = p.cell_obj + (size_t)w0 * C;
        for (uint32_t i = lane; i < ncells; i += kWave) g_obj[i] = s_obj[i];
        if (active) {
            const uint32_t world = w0 + wl;
            p.players[(size_t)w0 * P + lane] = make_uint2(posori, held);
            p.reward[(size_t)q * N + world] = reward_world;
            if (q == 0) {
                p.timestep[world] = t;
                p.done[world] = (int32_t)reset_now;
            }
            if (g + 1 < kG && !p.direct) s_cur[wl * C + (posori & 0xFFu)] = 0xFF;  // the cell -> player map starts the next group empty
        }
        wave_lds_sync();
    }
    STAMP(15);
    STAMP_REALTIME(14);
}

template <int kC, int kW, int kWidth, int kPots, int kHold, bool kI64, int kG, bool kPlain = false>
__global__ void __launch_bounds__(kBlock) mrl_overcooked_step_groups_fixed(MRL_HOT_ARGS, const StepParams p)
{
    StepParams q = fixed_params<kC, kW, kWidth, kPots, kHold>(p);
    take_hot_args<kI64>(q, MRL_HOT_PASS);
    groups_body<2, kG, kPlain>(q);
}

// fallback for layouts without the single-pass encode: draw into the ACTION tensor, then an ordinary step
__global__ void mrl_overcooked_draw_actions(int32_t *action, uint32_t players, uint32_t n, uint64_t seed, uint32_t step)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)players * n) action[i] = (int32_t)mrl_random_action(seed, step, (uint32_t)(i % n), (uint32_t)(i / n));
}

__global__ void fill_ids(int32_t *world_id, int32_t *row_id, uint32_t rows, uint32_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)rows * n) {
        world_id[i] = (int32_t)(i % n);
        row_id[i] = (int32_t)(i / n);
    }
}

__global__ void fill_i32(int32_t *dst, int32_t value, size_t count)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = value;
}

struct OvercookedSim final : mrl_sim {
    StepParams params{};
    uint32_t H = 0;
    uint32_t grid = 0, lds_bytes = 0;
    bool generic = false;  // tests: two-player layouts through the any-player-count transition as well
    using StepKernel = void (*)(const StepParams);
    using RolloutKernel = void (*)(const StepParams, uint32_t, uint64_t, uint32_t, int32_t *, const int32_t *);
    StepKernel generic_step = nullptr;        // mrl_overcooked_step<false, 2 or 0, store flavour>
    RolloutKernel generic_rollout = nullptr;  // mrl_overcooked_rollout<2 or 0, store flavour>
    using FixedKernel = void (*)(MRL_HOT_ARGS, const StepParams);
    FixedKernel fixed_kernel = nullptr;  // mrl_overcooked_step_fixed<...> when the parameters are exactly its
    FixedKernel fixed_kernel_i64 = nullptr;
    FixedKernel groups_kernel = nullptr;      // mrl_overcooked_step_groups_fixed<...>: kGroups groups per wave
    FixedKernel groups_kernel_i64 = nullptr;
    uint32_t groups_grid = 0;
    const char *groups_name = nullptr;
    void (*fixed_rollout)(const StepParams, uint32_t, uint64_t, uint32_t, int32_t *, const int32_t *) = nullptr;
    // the multi-step launches with twice as many worlds per wave (half as many waves: the state lives in LDS / registers for
    // the whole launch, so there is no load phase to hide behind other waves and fewer, fatter waves issue fewer instructions)
    void (*wide_rollout)(const StepParams, uint32_t, uint64_t, uint32_t, int32_t *, const int32_t *) = nullptr;
    StepParams wide_params{};
    uint32_t wide_grid = 0, wide_lds = 0;
    const char *fixed_name = nullptr;
    int32_t *action = nullptr, *active = nullptr, *mask = nullptr;
    int32_t *world_id = nullptr, *agent_id = nullptr, *loc_world_id = nullptr, *loc_id = nullptr;
    uint8_t *own_obs = nullptr;  // the OBS_WORLD_MAJOR buffer; params.obs points elsewhere while the output is redirected

    // The kernels take the slab's address from the launch arguments and never read it back, so writing a step's
    // observations into a caller's slot (a rollout buffer) instead of the exported tensor is a different pointer in
    // the same launch: same bytes, same stores.
    uint64_t observation_bytes() const override { return (uint64_t)num_worlds * params.block_bytes; }
    uint64_t set_observation_output(void *out) override
    {
        set_observation_ring(out, 0, 1);
        return observation_bytes();
    }
    // a ring of slots: the step number `ring_pos` since this call writes slot ring_pos % slots (host-side count: a
    // launch captured in a HIP graph keeps the slot it was captured with)
    uint8_t *ring_base = nullptr;
    uint64_t ring_stride = 0;
    uint32_t ring_slots = 1;
    uint64_t ring_pos = 0;
    // Slots that do not start on 16-byte boundaries (a dense (T, N, P, H, W, F) buffer whose N x P x H x W x F is not a
    // multiple of 16: coordination_ring at 1001 worlds) are STAGED: the kernels stream a slab out in 16-byte chunks from a
    // 16-byte aligned base, so the step writes a slab of the simulator's (`staging`, allocated at the first such call; the
    // exported tensor stays untouched) and a device-to-device copy behind the launch moves it to the slot -- one more pass
    // over the slab per step, and the multi-step launches run one launch per step.  Aligned slots cost nothing.
    bool staged = false;
    uint8_t *staging = nullptr, *pending_dest = nullptr;
    void set_observation_ring(void *base, uint64_t stride_bytes, uint32_t slots) override
    {
        ring_base = base ? static_cast<uint8_t *>(base) : own_obs;
        ring_stride = base ? stride_bytes : 0;
        ring_slots = base && slots > 1 ? slots : 1;
        ring_pos = 0;
        staged = base && ((reinterpret_cast<uintptr_t>(base) & 15u) != 0 || (ring_slots > 1 && (ring_stride & 15u) != 0));
        if (staged && !staging) staging = arena.alloc<uint8_t>(observation_bytes(), false);
        for (StepParams *q : {&params, &wide_params}) {
            q->obs = staged ? staging : ring_base;
            q->ring_stride = staged ? 0 : ring_stride;
            q->ring_slots = staged ? 1 : ring_slots;
            q->ring_first = 0;
        }
    }
    // staged slots: the slab just written -> the caller's slot, behind the launch on the same stream
    void deliver(uint8_t *dest, hipStream_t stream)
    {
        if (staged && dest) MRL_HIP(hipMemcpyAsync(dest, staging, observation_bytes(), hipMemcpyDeviceToDevice, stream));
    }
    // the slot(s) of the next `steps` steps: single-step launches get the slot as their `obs`, multi-step ones the first index
    uint8_t *take_slots(uint32_t steps, uint32_t *first)
    {
        const uint32_t at = (uint32_t)(ring_pos % ring_slots);
        ring_pos += steps;
        if (first) *first = at;
        return ring_base + (size_t)at * ring_stride;
    }

    // this step's parameters as the generic step kernel takes them (mrl_step_many)
    StepParams generic_step_params(const int32_t *actions)
    {
        StepParams a = params;
        a.actions = actions ? actions : action;
        pending_dest = take_slots(1, nullptr);  // (staged: delivered by step_many_overcooked behind the shared launch)
        a.obs = staged ? staging : pending_dest;
        a.ring_slots = 1;
        a.per_xcd = grid >> 3;
        a.pair_exchange = (a.P == 2 && !generic) ? 1u : 0u;
        return a;
    }

    void launch(bool init, const int32_t *actions, hipStream_t stream)
    {
        StepParams a = params;
        a.actions = actions ? actions : action;
        uint8_t *const dest = init ? nullptr : take_slots(1, nullptr);
        if (!init) a.obs = staged ? staging : dest;
        a.ring_slots = 1;  // a single step writes exactly its `obs`
        a.per_xcd = ((!init && groups_kernel) ? groups_grid : grid) >> 3;
        const void *hot_actions = a.actions64 ? static_cast<const void *>(a.actions64) : static_cast<const void *>(a.actions);
        // two-player layouts (all five standard ones) exchange through DPP instead of LDS
        if (a.team && init)
            hipLaunchKernelGGL((mrl_overcooked_step_team<true>), dim3(grid), dim3(kBlock), lds_bytes, stream, a);
        else if (a.team)
            hipLaunchKernelGGL((mrl_overcooked_step_team<false>), dim3(grid), dim3(kBlock), lds_bytes, stream, a);
        else if (init)
            hipLaunchKernelGGL((mrl_overcooked_step<true, 0>), dim3(grid), dim3(kBlock), lds_bytes, stream, a);
        else if (groups_kernel)
            hipLaunchKernelGGL(a.actions64 ? groups_kernel_i64 : groups_kernel, dim3(groups_grid), dim3(kBlock), lds_bytes, stream, a.cell_obj,
                               a.players, a.timestep, hot_actions, a.consts, a.terr_off, a.num_worlds, a.per_xcd, a);
        else if (fixed_kernel)
            hipLaunchKernelGGL(a.actions64 ? fixed_kernel_i64 : fixed_kernel, dim3(grid), dim3(kBlock), lds_bytes, stream, a.cell_obj, a.players,
                               a.timestep, hot_actions, a.consts, a.terr_off, a.num_worlds, a.per_xcd, a);
        else
            hipLaunchKernelGGL(generic_step, dim3(grid), dim3(kBlock), lds_bytes, stream, a);
        MRL_HIP(hipGetLastError());
        deliver(dest, stream);
    }

    void phase1(const int32_t *actions, hipStream_t stream) override { launch(false, actions, stream); }
    bool step_i64(const long long *actions, hipStream_t stream) override
    {
        const long long *keep = params.actions64;
        params.actions64 = actions;
        params.action_mirror = action;
        launch(false, nullptr, stream);
        params.actions64 = keep;
        return true;
    }
    void phase2(const uint32_t *, hipStream_t) override {}

    void rollout_random(uint32_t num_steps, uint64_t seed, uint32_t first_step, hipStream_t stream) override
    {
        if (num_steps == 0) return;
        if (params.whole && !staged) {
            take_slots(num_steps, &params.ring_first);
            wide_params.ring_first = params.ring_first;
            if (wide_rollout)
                hipLaunchKernelGGL(wide_rollout, dim3(wide_grid), dim3(kWavesPerBlock * kWave), wide_lds, stream, wide_params, num_steps,
                                   seed, first_step, action, (const int32_t *)nullptr);
            else if (fixed_rollout)
                hipLaunchKernelGGL(fixed_rollout, dim3(grid), dim3(kWavesPerBlock * kWave), lds_bytes, stream, params, num_steps, seed,
                                   first_step, action, (const int32_t *)nullptr);
            else
                hipLaunchKernelGGL(generic_rollout, dim3(grid), dim3(kWavesPerBlock * kWave), lds_bytes, stream, params, num_steps, seed,
                                   first_step, action, (const int32_t *)nullptr);
            MRL_HIP(hipGetLastError());
            return;
        }
        const size_t count = (size_t)params.P * num_worlds;
        for (uint32_t k = 0; k < num_steps; k++) {
            hipLaunchKernelGGL(mrl_overcooked_draw_actions, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream,
                               action, params.P, num_worlds, seed, first_step + k);
            launch(false, nullptr, stream);
        }
    }

    void step_sequence(const int32_t *actions, uint32_t num_steps, hipStream_t stream) override
    {
        if (num_steps == 0) return;
        if (params.whole && !staged) {
            take_slots(num_steps, &params.ring_first);
            wide_params.ring_first = params.ring_first;
            if (wide_rollout)
                hipLaunchKernelGGL(wide_rollout, dim3(wide_grid), dim3(kWavesPerBlock * kWave), wide_lds, stream, wide_params, num_steps,
                                   0ull, 0u, action, actions);
            else if (fixed_rollout)
                hipLaunchKernelGGL(fixed_rollout, dim3(grid), dim3(kWavesPerBlock * kWave), lds_bytes, stream, params, num_steps, 0ull, 0u,
                                   action, actions);
            else
                hipLaunchKernelGGL(generic_rollout, dim3(grid), dim3(kWavesPerBlock * kWave), lds_bytes, stream, params, num_steps, 0ull, 0u,
                                   action, actions);
            MRL_HIP(hipGetLastError());
            return;
        }
        mrl_sim::step_sequence(actions, num_steps, stream);
    }

    void ensure_ids()
    {
        if (world_id) return;
        const uint32_t P = params.P, N = num_worlds, rows = params.rows;
        world_id = arena.alloc<int32_t>((size_t)P * N, false);
        agent_id = arena.alloc<int32_t>((size_t)P * N, false);
        loc_world_id = arena.alloc<int32_t>((size_t)rows * N, false);
        loc_id = arena.alloc<int32_t>((size_t)rows * N, false);
        const size_t a = (size_t)P * N, b = (size_t)rows * N;
        hipLaunchKernelGGL(fill_ids, dim3((unsigned)((a + 255) / 256)), dim3(256), 0, 0, world_id, agent_id, P, N);
        hipLaunchKernelGGL(fill_ids, dim3((unsigned)((b + 255) / 256)), dim3(256), 0, 0, loc_world_id, loc_id, rows, N);
        MRL_HIP(hipGetLastError());
        MRL_HIP(hipDeviceSynchronize());
    }

    bool tensor(int slot, mrl_tensor_desc *out) override
    {
        const int64_t P = params.P, N = num_worlds, C = params.C, F = params.F, W = params.W;
        switch (slot) {
        case MRL_OVERCOOKED_DONE: *out = mrl::make_desc(params.done, MRL_INT32, device, {N}); return true;
        case MRL_OVERCOOKED_ACTIVE_AGENT: *out = mrl::make_desc(active, MRL_INT32, device, {P, N}); return true;
        case MRL_OVERCOOKED_ACTION: *out = mrl::make_desc(action, MRL_INT32, device, {P, N, 1}); return true;
        case MRL_OVERCOOKED_OBSERVATION:
            *out = mrl::make_desc(own_obs, MRL_INT8, device, {P * C, N, F}, {F, P * C * F, 1});
            return true;
        case MRL_OVERCOOKED_ACTION_MASK: *out = mrl::make_desc(mask, MRL_INT32, device, {P, N, 6}); return true;
        case MRL_OVERCOOKED_REWARD: *out = mrl::make_desc(params.reward, MRL_INT32, device, {P, N}); return true;
        case MRL_OVERCOOKED_WORLD_ID: ensure_ids(); *out = mrl::make_desc(world_id, MRL_INT32, device, {P, N}); return true;
        case MRL_OVERCOOKED_AGENT_ID: ensure_ids(); *out = mrl::make_desc(agent_id, MRL_INT32, device, {P, N}); return true;
        case MRL_OVERCOOKED_LOCATION_WORLD_ID:
            ensure_ids();
            *out = mrl::make_desc(loc_world_id, MRL_INT32, device, {P * C, N});
            return true;
        case MRL_OVERCOOKED_LOCATION_ID: ensure_ids(); *out = mrl::make_desc(loc_id, MRL_INT32, device, {P * C, N}); return true;
        case MRL_OVERCOOKED_OBS_WORLD_MAJOR:
            *out = mrl::make_desc(own_obs, MRL_INT8, device, {N, P, (int64_t)H, W, F});
            return true;
        case MRL_OVERCOOKED_STATE_PLAYERS: *out = mrl::make_desc(params.players, MRL_UINT8, device, {N, P, 8}); return true;
        case MRL_OVERCOOKED_STATE_OBJECTS: *out = mrl::make_desc(params.cell_obj, MRL_UINT8, device, {N, C, 4}); return true;
        case MRL_OVERCOOKED_STATE_TIMESTEP: *out = mrl::make_desc(params.timestep, MRL_INT32, device, {N}); return true;
#ifdef MRL_DIAG
        case 14:
            if (!params.stamps) return false;
            *out = mrl::make_desc(params.stamps, MRL_UINT8, device, {(int64_t)grid * kWavesPerBlock * 16 * 8});
            return true;
#endif
        default: return false;
        }
    }

    size_t action_elems() const override { return (size_t)params.P * num_worlds; }
    void launch_shape(uint32_t out[4]) const override
    {
        out[0] = groups_kernel ? groups_grid : grid;
        out[1] = kBlock;
        out[2] = lds_bytes;
        out[3] = params.wpw * (groups_kernel ? 2u : 1u);  // worlds a wavefront steps (two groups one after the other, or one)
    }
    const char *kernel_name() const override
    {
        if (params.team) return "mrl_overcooked_step_team<false>";
        if (groups_kernel) return groups_name;
        if (fixed_kernel) return fixed_name;
        return params.P == 2 && !generic ? "mrl_overcooked_step<false, 2>" : "mrl_overcooked_step<false, 0>";
    }

    uint64_t bytes_per_world_step() const override
    {
        // actions 4P + player state r/w 2*8P + cell objects
        // r/w 2*4C + timestep r/w 8 + obs out P*C*F + reward 4P + done 4
        const uint64_t P = params.P, C = params.C;
        return 4 * P + 16 * P + 8 * C + 8 + params.block_bytes + 4 * P + 4;
    }
};

}  // namespace

mrl_sim *mrl::create_overcooked(const mrl_overcooked_config *cfg, int gpu_id, uint32_t num_worlds)
{
    if (!cfg || !cfg->terrain || !cfg->start_player_x || !cfg->start_player_y || !cfg->recipe_values ||
        !cfg->recipe_times) {
        set_error("overcooked: null config field");
        throw HipError{MRL_ERR_INVALID};
    }
    const int64_t H = cfg->height, W = cfg->width, P = cfg->num_players;
    if (H < 3 || W < 3 || H * W > 255) {
        set_error("overcooked: height*width must be 9..255 (the reference stores the cell count in a uint8, "
                  "src/overcooked_env/sim.hpp:86), got %lldx%lld",
                  (long long)H, (long long)W);
        throw HipError{MRL_ERR_INVALID};
    }
    if (P < 1 || P > 64) {
        set_error("overcooked: num_players must be 1..64 (MAX_NUM_PLAYERS), got %lld", (long long)P);
        throw HipError{MRL_ERR_INVALID};
    }
    if (num_worlds == 0) {
        set_error("overcooked: num_worlds must be > 0");
        throw HipError{MRL_ERR_INVALID};
    }
    const int64_t C = H * W;
    alignas(4) uint8_t consts[kConstBytes];
    memset(consts, 0, sizeof(consts));
    uint32_t num_pots = 0;
    for (int64_t c = 0; c < C; c++) {
        const int64_t t = cfg->terrain[c];
        if (t < 0 || t > 6) {
            set_error("overcooked: terrain[%lld] = %lld is not a TerrainT value", (long long)c, (long long)t);
            throw HipError{MRL_ERR_INVALID};
        }
        const int64_t x = c % W, y = c / W;
        if (t == T_AIR && (x == 0 || y == 0 || x == W - 1 || y == H - 1)) {
            set_error("overcooked: walkable cell on the grid border at (%lld,%lld); the step indexes neighbours "
                      "without bounds checks (src/overcooked_env/sim.cpp:185-197)",
                      (long long)x, (long long)y);
            throw HipError{MRL_ERR_INVALID};
        }
        consts[kConstTerrain + c] = (uint8_t)t;
        if (t == T_POT) consts[kConstPots + num_pots++] = (uint8_t)c;
    }
    for (int r = 0; r < 16; r++) {
        consts[kConstTimes + r] = (uint8_t)cfg->recipe_times[r];
        consts[kConstValues + r] = (uint8_t)cfg->recipe_values[r];
    }
    for (int64_t q = 0; q < P; q++) {
        const int64_t x = cfg->start_player_x[q], y = cfg->start_player_y[q];
        if (x < 1 || y < 1 || x >= W - 1 || y >= H - 1) {
            set_error("overcooked: start position of player %lld (%lld,%lld) is not an interior cell", (long long)q,
                      (long long)x, (long long)y);
            throw HipError{MRL_ERR_INVALID};
        }
        consts[kConstStart + q] = (uint8_t)(y * W + x);
    }

    bind_device(gpu_id);
    auto *sim = new OvercookedSim();
    try {
        sim->game = MRL_GAME_OVERCOOKED;
        sim->device = gpu_id;
        sim->num_worlds = num_worlds;
        sim->H = (uint32_t)H;
        sim->generic = mrl::debug_get("overcooked.variant", 0) == 1;
        StepParams &a = sim->params;
        const uint32_t N = num_worlds;
        a.num_worlds = N;
        a.P = (uint32_t)P;
        a.C = (uint32_t)C;
        a.W = (uint32_t)W;
        a.F = 5 * a.P + 16;
        a.rows = a.P * a.C;
        a.block_bytes = a.rows * a.F;
        a.inv_c = (uint32_t)((1ull << 32) / (uint64_t)C) + 1u;
        a.placement_rew = (uint8_t)cfg->placement_in_pot_rew;  // uint8 like WorldState (sim.hpp:95-97)
        a.soup_pickup_rew = (uint8_t)cfg->soup_pickup_rew;
        a.horizon = cfg->horizon;
        a.num_pots = num_pots;
        a.deltas = (uint64_t)(uint8_t)(int8_t)(-W) | ((uint64_t)(uint8_t)(int8_t)W << 8) | (1ull << 16) | (0xFFull << 24);
        memcpy(a.times_w, consts + kConstTimes, 16);
        memcpy(a.values_w, consts + kConstValues, 16);
        memcpy(&a.pots_w, consts + kConstPots, 4);
        a.tail_even = (a.P % 2 == 0) ? 1u : 0u;
        {
            // Multi-pass stream-out: write-through (sc1) stores stream the slab out while the waves still work, which pays
            // as long as the slab fits the 256 MiB Infinity Cache; larger slabs go out as plain stores through the L2
            // write-back (same rate in general, and 30 players x 1000 worlds -- 1.27 GB in 1.27 MB strides -- ran at half
            // the rate with write-through: 490 vs 258 us per step).
            const int64_t knob = mrl::debug_get("overcooked.store_policy", 0);  // 0 auto, 1 sc1, 2 plain, 3 nt
            const uint64_t slab = (uint64_t)N * a.P * a.C * a.F;
            a.store_policy = knob ? (uint32_t)(knob - 1) : (slab > (256ull << 20) ? 1u : 0u);
        }
        a.private_consts = (P <= 4 && num_pots <= 4 && !mrl::debug_get("overcooked.shared_consts", 0)) ? 1u : 0u;
        memcpy(&a.starts_w, consts + kConstStart, 4);
        a.wpp = a.rows <= (uint32_t)kRowsPerPass ? (uint32_t)kRowsPerPass / a.rows : 0u;
        a.steady = (a.wpp > 0 && ((uint64_t)a.wpp * a.block_bytes) % 16 == 0) ? 1u : 0u;
#ifdef MRL_DIAG
        a.ablate = (uint32_t)mrl::debug_get("ablate", 0);
        a.stamps = nullptr;
#endif
        a.inv_p = P == 1 ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / (uint64_t)P) + 1u;  // x/1: umulhi(x, 2^32-1) == x-1 for x>0; handled in-kernel
        a.inv_rows = (uint32_t)((1ull << 32) / (uint64_t)a.rows) + 1u;
        const uint32_t tile_bytes = (((uint32_t)kRowsPerPass * a.F + 15u) & ~15u) + 32u;
        // Per-wave LDS.  If the group's whole observation slab fits a tile of kWholeTileMax bytes
        // the encode is single-pass and needs no per-cell tail buffer; otherwise rows go through
        // a kRowsPerPass-row tile fed from a 16-byte-per-cell tail buffer.
        // 9400 admits four worlds of a 9x5 layout (asymmetric_advantages: 25.4 -> 21.7 us per launch at 32768
        // worlds against two worlds per wave); larger tiles were measured and lose (8 worlds of a 5x5 layout:
        // 15.5 vs 12.4 us, 8 of counter_circuit 20.0 vs 16.7, 16 of cramped_room 12.8 vs 10.8)
        uint32_t kWholeTileMax = (uint32_t)mrl::debug_get("overcooked.whole_max", 9400);  // knobs: experiments and tests only
        const uint32_t lds_max = (uint32_t)mrl::debug_get("overcooked.lds_max", 65536);
        auto layout = [&](uint32_t wpw) {
            auto up16 = [](uint32_t v) { return (v + 15u) & ~15u; };
            a.wpw = wpw;
            a.off_pl = up16(wpw * a.C * 4);
            a.off_x = a.off_pl + up16(wpw * a.P * 8);
            // scratch of the any-player-count transition only: the two-player kernels exchange through DPP, and every
            // byte counts -- four workgroups of a 32768-world launch must fit a CU's 160 KB together
            a.off_sum = a.off_x + ((a.P == 2 && !sim->generic) ? 0u : 128u * 4u);
            a.off_cur = a.off_sum + up16(2u * wpw * 4u);
            a.off_flags = a.off_cur + up16(wpw * a.C);
            a.off_terr = a.off_flags + 64u;
            a.off_list = a.off_terr + (a.private_consts ? up16(a.C) : 0u);
            a.off_tail = a.off_list + up16(wpw * a.C * 2u);
            const uint32_t whole_tile = up16(wpw * a.block_bytes) + 32u;
            a.whole = (a.tail_even && whole_tile <= kWholeTileMax) ? 1u : 0u;
            // (group slabs that do not start on 16-byte boundaries -- one or two big worlds per wave -- keep tile and slab congruent)
            a.patch = (a.whole && wpw * a.rows <= (uint32_t)kTerrPosPerLane * kWave && wpw * a.block_bytes < 65536u - 16u) ? 1u : 0u;
            a.unaligned = (a.patch && ((uint64_t)wpw * a.block_bytes) % 16u != 0) ? 1u : 0u;  // (block_bytes is even here: tail_even)
            if (a.whole) {
                a.off_tile = a.off_tail;
                // with a.patch the tile is zeroed in whole 256-byte pieces (tile_zero_addtid)
                a.lds_wave_stride = a.off_tile + (a.patch ? ((wpw * a.block_bytes + (a.unaligned ? 15u : 0u) + 255u) & ~255u) : whole_tile);
            } else {
                a.off_tile = a.off_tail + wpw * a.C * 16;
                a.lds_wave_stride = a.off_tile + tile_bytes;
            }
            return kConstBytes + kWavesPerBlock * a.lds_wave_stride;
        };
        // worlds per wave: as many as keep >= 4096 waves in the launch, fit 40 KB of LDS per
        // workgroup (>= 4 workgroups per CU) and keep the reciprocal divisions exact
        uint32_t wpw = 64;
        while (wpw > 1 && wpw * a.P > 64u) wpw >>= 1;  // the transition runs one lane per (world, player)
        const uint32_t wpw_cap = wpw;
        if (const int64_t forced = mrl::debug_get("overcooked.wpw", 0)) {
            wpw = (forced < 1 || forced > 64) ? 1u : std::min((uint32_t)forced, wpw_cap);
        } else {
            while (wpw > 1 && (N + wpw - 1) / wpw < 4096) wpw >>= 1;
            // the single-pass encode is worth more than wider groups (measured, 32768 worlds,
            // us per launch at wpw 4 / 8: coordination_ring 14.4 / 17.6, counter_circuit 19.4 / 25.8,
            // asymmetric_advantages 31.5 / 36.3): shrink the group until its slab fits one tile
            uint32_t cand = wpw;
            while (cand > 1 && (layout(cand), !a.whole)) cand >>= 1;
            if ((layout(cand), a.whole)) wpw = cand;
            // A world that does not fit the 9400-byte tile but fits 16 KB (two players on up to 255 cells, four on up to 113)
            // still does better as ONE wave's single-pass tile than through the multi-pass row assembly once there are a few
            // hundred worlds: many_player_layout with 2 players, us per step multi-pass (the waves of a workgroup share a world,
            // team_body) / single-pass -- 256 worlds 5.8 / 6.0, 512 6.6 / 6.2, 768 7.6 / 6.6, 1000 8.5 / 6.9, 2000 14.5 / 9.1,
            // 4000 22.9 / 16.0
            if (!a.whole && N >= 384 && !mrl::debug_get("overcooked.whole_max", 0)) {
                const uint32_t keep = kWholeTileMax;
                kWholeTileMax = 16384;
                if ((layout(1), a.whole))
                    wpw = 1;
                else
                    kWholeTileMax = keep;
            }
        }
        while (wpw > 1 && (layout(wpw) > lds_max || (uint64_t)wpw * a.rows * a.rows >= (1ull << 32) ||
                           (uint64_t)wpw * a.C * a.C >= (1ull << 32)))
            wpw >>= 1;
        sim->lds_bytes = layout(wpw) + (uint32_t)mrl::debug_get("overcooked.lds_pad", 0);  // the pad: residency experiments
        // p.direct (patch_direct): holder cells = counters / pots next to a walkable cell; players must start on walkable cells
        std::vector<uint32_t> holders;  // cell | is_pot << 8
        for (int64_t c = 0; c < C; c++) {
            const uint32_t t = consts[kConstTerrain + c];
            if (t != T_COUNTER && t != T_POT) continue;
            const int64_t x = c % W, y = c / W;
            bool faced = false;
            if (x > 0 && consts[kConstTerrain + c - 1] == T_AIR) faced = true;
            if (x + 1 < W && consts[kConstTerrain + c + 1] == T_AIR) faced = true;
            if (y > 0 && consts[kConstTerrain + c - W] == T_AIR) faced = true;
            if (y + 1 < H && consts[kConstTerrain + c + W] == T_AIR) faced = true;
            if (faced) holders.push_back((uint32_t)c | (t == T_POT ? 0x100u : 0u));
        }
        // table of a group of `gw` worlds: round 0 keeps lanes [0, gw * P) for the players
        auto hold_table = [&](uint32_t gw) {
            std::vector<uint32_t> tab;
            const uint32_t free0 = (uint32_t)kWave - std::min<uint32_t>((uint32_t)kWave, gw * a.P);
            uint32_t s = 0;
            for (uint32_t l = 0; l < gw; l++)
                for (const uint32_t h : holders) {
                    const uint32_t c = h & 0xFFu;
                    const uint32_t slot = s < free0 ? gw * a.P + s : (uint32_t)kWave + (s - free0);
                    if (tab.size() <= slot) tab.resize(slot + 1, 0u);
                    tab[slot] = (l * a.block_bytes + c * a.F) | ((l * a.C + c) << 16) | (1u << 30) | ((h >> 8) << 31);
                    s++;
                }
            if (tab.empty()) tab.resize(1, 0u);
            return tab;
        };
        {
            bool starts_walkable = true;
            for (int64_t q = 0; q < P; q++) starts_walkable = starts_walkable && consts[kConstTerrain + consts[kConstStart + q]] == T_AIR;
            a.direct = (a.patch && starts_walkable && hold_table(a.wpw).size() <= (size_t)kHoldPerLane * kWave &&
                        !mrl::debug_get("overcooked.no_direct", 0))
                           ? 1u
                           : 0u;
        }
        // Stores of the single-pass stream-out (stream_store_rsrc): write-through, except where the slab is larger than the
        // Infinity Cache AND a group's slab is not whole 128-byte lines -- then ordinary stores, which the L2 merges
        // (the multi-step launches rewrite the same lines step after step and do better with ordinary stores for such groups
        // at every size: asymmetric_advantages 32768 worlds 10.6 -> 9.3 us per step, 65536 21.2 -> 18.6, coordination_ring
        // 6.61 -> 6.44; the single step inside the cache does not: coordination_ring 10.5 vs 12.0, asymmetric_advantages 15.6 vs 17.3)
        const auto plain_for = [&](uint32_t group_worlds, bool multi_step) {
            const int64_t knob = mrl::debug_get("overcooked.whole_store", 0);  // 0 by slab size and alignment, 1 write-through, 2 plain
            const uint64_t slab = (uint64_t)N * a.block_bytes;
            const bool whole_lines = ((uint64_t)group_worlds * a.block_bytes) % 64u == 0;  // (64: Simplecooked random0's 8000-byte groups, half a 128-byte line off, do not care)
            return knob ? knob == 2 : (!whole_lines && (multi_step || slab > (256ull << 20)));
        };
        const bool plain = plain_for(wpw, false), plain_multi = plain_for(wpw, true);
        {
            const bool pairs = a.P == 2 && !sim->generic;  // two-player layouts (all five standard ones) exchange through DPP instead of LDS
            sim->generic_step = pairs ? (plain ? &mrl_overcooked_step<false, 2, true> : &mrl_overcooked_step<false, 2, false>)
                                      : (plain ? &mrl_overcooked_step<false, 0, true> : &mrl_overcooked_step<false, 0, false>);
            sim->generic_rollout = pairs ? (plain_multi ? &mrl_overcooked_rollout<2, true> : &mrl_overcooked_rollout<2, false>)
                                         : (plain_multi ? &mrl_overcooked_rollout<0, true> : &mrl_overcooked_rollout<0, false>);
        }
        if (sim->lds_bytes > 65536) {
            MRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sim->generic_step), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sim->lds_bytes));
            MRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mrl_overcooked_step<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)sim->lds_bytes));
            MRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sim->generic_rollout), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)sim->lds_bytes));
        }
        // Few worlds of a large layout leave most of the GPU idle at one wave per world: the four waves of a
        // workgroup then share a world.  Measured on many_player_layout (15x17) at 1000 worlds, us per step
        // shared / not: 2 players 9.4 / 10.3, 4 players 14.6 / 18.4; with round 1's transition (serial in the player
        // count on one lane, redundant in every sibling wave) more players lost (8: 36.8 / 32.6, 30: 564 / 521), with
        // the lane-per-player transition they no longer do (8: 30.7 / 30.8, 16: 82.7 / 86.7, 30: 253 / 260), so any count.
        a.share = (wpw == 1 && !a.whole && a.wpp == 0 && N < 8192 && P <= (uint32_t)mrl::debug_get("overcooked.share_max_players", 64) && !mrl::debug_get("overcooked.no_share", 0)) ? 1u : 0u;
        if (a.share && !mrl::debug_get("overcooked.share_private", 0)) {
            // team_body: one copy of the world's state per workgroup, the four row tiles behind it
            auto up16 = [](uint32_t v) { return (v + 15u) & ~15u; };
            a.team = 1;
            a.off_pl = up16(a.C * 4);
            a.off_x = a.off_pl + up16(a.P * 8);
            a.off_sum = a.off_x + 128u * 4u;
            a.off_cur = a.off_sum + 16u;
            a.off_flags = a.off_cur + up16(a.C);
            a.off_tail = a.off_flags + 64u;
            a.off_tile = a.off_tail + a.C * 16u;
            a.lds_wave_stride = tile_bytes;
            sim->lds_bytes = kConstBytes + a.off_tile + kWavesPerBlock * tile_bytes + (uint32_t)mrl::debug_get("overcooked.lds_pad", 0);
            if (sim->lds_bytes > 65536) {
                MRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mrl_overcooked_step_team<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)sim->lds_bytes));
                MRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mrl_overcooked_step_team<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)sim->lds_bytes));
            }
        }
        {
            // a kernel specialised for this layout size, if there is one and the parameters are exactly what it assumes
            auto matches = [&](uint32_t C_, uint32_t wpw_, uint32_t width_, uint32_t pots_, uint32_t hold_) {
                const FixedLayout f = fixed_layout(C_, wpw_);
                return a.P == 2 && !sim->generic && a.C == C_ && a.W == width_ && a.num_pots == pots_ && a.wpw == wpw_ && a.whole &&
                       a.patch && !a.unaligned && a.direct && holders.size() == hold_ && !a.share && a.private_consts && a.off_pl == f.off_pl && a.off_sum == f.off_sum &&
                       a.off_cur == f.off_cur && a.off_flags == f.off_flags && a.off_terr == f.off_terr && a.off_list == f.off_list &&
                       a.off_tile == f.off_tile && a.lds_wave_stride == f.stride && !mrl::debug_get("overcooked.no_fixed", 0);
            };
#define MRL_FIXED(C_, WPW_, WIDTH_, POTS_, HOLD_)                                                                    \
    if (!sim->fixed_kernel && matches(C_, WPW_, WIDTH_, POTS_, HOLD_)) {                                            \
        sim->fixed_kernel = plain ? &mrl_overcooked_step_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, false, true>         \
                                  : &mrl_overcooked_step_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, false, false>;       \
        sim->fixed_kernel_i64 = plain ? &mrl_overcooked_step_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, true, true>      \
                                      : &mrl_overcooked_step_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, true, false>;    \
        sim->fixed_rollout = plain_multi ? &mrl_overcooked_rollout_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, true>      \
                                         : &mrl_overcooked_rollout_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, false>;    \
        sim->fixed_name = "mrl_overcooked_step_fixed<" #C_ ", " #WPW_ ", " #WIDTH_ ", " #POTS_ ", " #HOLD_ ", false>";      \
        if (groups == 2) {                                                                                          \
            sim->groups_kernel = plain ? &mrl_overcooked_step_groups_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, false, 2, true>     \
                                       : &mrl_overcooked_step_groups_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, false, 2, false>;   \
            sim->groups_kernel_i64 = plain ? &mrl_overcooked_step_groups_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, true, 2, true>  \
                                           : &mrl_overcooked_step_groups_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, true, 2, false>; \
            sim->groups_name = "mrl_overcooked_step_groups_fixed<" #C_ ", " #WPW_ ", " #WIDTH_ ", " #POTS_ ", " #HOLD_ ", false, 2>"; \
        }                                                                                                           \
    }
            // Groups of worlds a wave steps one after the other in the single step (mrl_overcooked_step_groups_fixed): two
            // as soon as the launch has more groups than waves fit the GPU at once (then a second generation of waves
            // would start behind the first anyway, and a wave that requests both groups' state up front and steps the second
            // while the first one's stores drain does better); one below that, where two would halve the waves in flight.
            // Measured on MI355X with the r02_e kernels, us per launch one / two groups per wave -- cramped_room (4096 waves
            // fit): 32768 worlds 7.99 / 8.14, 36864 10.27 / 10.12, 45056 11.2 / 10.8, 65536 14.1 / 13.1, 131072 25.7 / 24.1,
            // 262144 54.8 / 48.6, 524288 120 / 112, 1 M 239 / 240; at 32768 worlds (8192 groups) asymmetric_advantages (3072
            // fit) 16.7 / 15.7, counter_circuit (3072) 13.7 / 13.1, coordination_ring (6144) 10.49 / 10.48, forced_coordination
            // 10.60 / 10.68; coordination_ring 16640 worlds (4160 groups) 6.4 / 7.7.
            // mrl_debug_set overcooked.groups: 0 = that rule, 1 = always one, 2 = always two.
            const int64_t groups_knob = mrl::debug_get("overcooked.groups", 0);
            const uint64_t ngroups = ((uint64_t)N + wpw - 1) / wpw;
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, gpu_id);
            const uint64_t wg_per_cu = std::min<uint64_t>(8, std::max<uint64_t>(1, (160u * 1024u) / std::max<uint32_t>(sim->lds_bytes, 1u)));
            const uint64_t waves_at_once = (uint64_t)cus * wg_per_cu * kWavesPerBlock;
            const int64_t groups = groups_knob ? groups_knob : (ngroups > waves_at_once + waves_at_once / 8 ? 2 : 1);
            // the five standard layouts: cells, worlds per wave, grid width, pots
            // the five standard layouts: cells, worlds per wave, grid width, pots, holder cells
            MRL_FIXED(20, 8, 5, 1, 6)    // cramped_room
            MRL_FIXED(45, 4, 9, 2, 14)   // asymmetric_advantages
            MRL_FIXED(25, 4, 5, 2, 9)    // coordination_ring, forced_coordination
            MRL_FIXED(40, 4, 8, 2, 18)   // counter_circuit
#undef MRL_FIXED
        }
        const uint32_t waves = a.share ? N * kWavesPerBlock : (N + wpw - 1) / wpw;
        const uint32_t blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
        sim->grid = (blocks + 7u) & ~7u;
        a.per_xcd = sim->grid >> 3;  // launch() sets it again per launch (the two-groups kernels have their own grid)
        if (sim->groups_kernel) {
            const uint32_t gwaves = (waves + 1u) / 2u;
            sim->groups_grid = ((gwaves + kWavesPerBlock - 1) / kWavesPerBlock + 7u) & ~7u;
        }

#ifdef MRL_DIAG
        if (mrl::debug_get("stamps", 0)) a.stamps = sim->arena.alloc<unsigned long long>((size_t)sim->grid * kWavesPerBlock * 16);
#endif
        uint32_t *d_consts = sim->arena.alloc<uint32_t>(kConstBytes / 4, false);
        MRL_HIP(hipMemcpy(d_consts, consts, kConstBytes, hipMemcpyHostToDevice));
        a.consts = d_consts;
        a.cell_obj = sim->arena.alloc<uint32_t>((size_t)N * C);
        a.players = sim->arena.alloc<uint2>((size_t)N * P);
        a.timestep = sim->arena.alloc<int32_t>(N);
        a.reward = sim->arena.alloc<int32_t>((size_t)N * P);
        a.done = sim->arena.alloc<int32_t>(N);
        a.obs = sim->arena.alloc<uint8_t>((size_t)N * a.block_bytes, false);
        sim->own_obs = a.obs;
        sim->ring_base = a.obs;
        a.ring_stride = 0;
        a.ring_slots = 1;
        a.ring_first = 0;
        {
            // per row of a group: where its terrain one-hot byte goes in the tile (channel 5P + t - 1, sim.cpp:642-645)
            a.terr_entries = a.wpw * a.rows;
            std::vector<uint16_t> off(a.terr_entries, 0);
            if (a.patch)
                for (uint32_t l = 0; l < a.wpw; l++)
                    for (uint32_t v = 0; v < a.P; v++)
                        for (uint32_t c = 0; c < a.C; c++) {
                            const uint32_t t = consts[kConstTerrain + c];
                            if (t != T_AIR) off[l * a.rows + v * a.C + c] = (uint16_t)(l * a.block_bytes + (v * a.C + c) * a.F + 5 * a.P + t - 1);
                        }
            uint16_t *d_off = sim->arena.alloc<uint16_t>(a.terr_entries, false);
            MRL_HIP(hipMemcpy(d_off, off.data(), a.terr_entries * sizeof(uint16_t), hipMemcpyHostToDevice));
            a.terr_off = d_off;
        }
        auto upload_hold = [&](StepParams &dst, uint32_t gw) {
            const std::vector<uint32_t> tab = hold_table(gw);
            uint32_t *d_tab = sim->arena.alloc<uint32_t>(tab.size(), false);
            MRL_HIP(hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            dst.hold_tab = d_tab;
            dst.hold_entries = (uint32_t)tab.size();
        };
        upload_hold(a, a.wpw);
        // Multi-step launches of the standard layouts: groups twice as wide where the table of terrain offsets and the
        // LDS of two workgroups per CU allow (measured in DESIGN.md 4.1; mrl_debug_set overcooked.wide_rollout 1 = never)
        if (sim->fixed_rollout && mrl::debug_get("overcooked.wide_rollout", 0) != 1) {
            const uint32_t rw = 2u * a.wpw;
            uint32_t stride = 0;
#define MRL_WIDE(C_, W2_, WIDTH_, POTS_, HOLD_)                                                                             \
    if (!sim->wide_rollout && a.C == C_ && rw == W2_ && a.W == WIDTH_ && a.num_pots == POTS_ && holders.size() == HOLD_) {                      \
        sim->wide_rollout = plain_for(rw, true) ? &mrl_overcooked_rollout_fixed<C_, W2_, WIDTH_, POTS_, HOLD_, true>       \
                                                : &mrl_overcooked_rollout_fixed<C_, W2_, WIDTH_, POTS_, HOLD_, false>;     \
        stride = fixed_layout(C_, W2_).stride;                                                                      \
    }
            MRL_WIDE(20, 16, 5, 1, 6)  // cramped_room
            MRL_WIDE(25, 8, 5, 2, 9)   // coordination_ring, forced_coordination
            MRL_WIDE(40, 8, 8, 2, 18)  // counter_circuit
#undef MRL_WIDE
            const uint32_t wide_waves = (N + rw - 1) / rw;
            if (sim->wide_rollout && (rw * a.rows > (uint32_t)kTerrPosPerLane * kWave || wide_waves < 2048u ||
                                      hold_table(rw).size() > (size_t)kHoldPerLane * kWave))
                sim->wide_rollout = nullptr;
            if (sim->wide_rollout) {
                sim->wide_params = a;
                sim->wide_lds = kConstBytes + kWavesPerBlock * stride;
                sim->wide_grid = (((wide_waves + kWavesPerBlock - 1) / kWavesPerBlock) + 7u) & ~7u;
                sim->wide_params.per_xcd = sim->wide_grid >> 3;
                std::vector<uint16_t> off((size_t)rw * a.rows, 0);
                for (uint32_t l = 0; l < rw; l++)
                    for (uint32_t v = 0; v < a.P; v++)
                        for (uint32_t c = 0; c < a.C; c++) {
                            const uint32_t t = consts[kConstTerrain + c];
                            if (t != T_AIR) off[l * a.rows + v * a.C + c] = (uint16_t)(l * a.block_bytes + (v * a.C + c) * a.F + 5 * a.P + t - 1);
                        }
                uint16_t *d_off = sim->arena.alloc<uint16_t>(off.size(), false);
                MRL_HIP(hipMemcpy(d_off, off.data(), off.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
                sim->wide_params.terr_off = d_off;
                upload_hold(sim->wide_params, rw);
                if (sim->wide_lds > 65536)
                    MRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sim->wide_rollout), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)sim->wide_lds));
            }
        }
        sim->action = sim->arena.alloc<int32_t>((size_t)N * P);
        sim->active = sim->arena.alloc<int32_t>((size_t)N * P, false);
        sim->mask = sim->arena.alloc<int32_t>((size_t)N * P * 6, false);
        const size_t na = (size_t)N * P, nm = na * 6;
        hipLaunchKernelGGL(fill_i32, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, 0, sim->active, 1, na);
        hipLaunchKernelGGL(fill_i32, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, 0, sim->mask, 1, nm);
        MRL_HIP(hipGetLastError());
        // Sim::Sim (sim.cpp:556-659): reset state + first observation
        sim->launch(true, nullptr, 0);
        MRL_HIP(hipDeviceSynchronize());
    } catch (...) {
        delete sim;
        throw;
    }
    return sim;
}

// mrl_step_many (include/mrl_envs.h): the simulators' steps as ONE launch of mrl_overcooked_step_many
void mrl::step_many_overcooked(mrl_sim *const *sims, uint32_t count, const int32_t *const *actions_or_null, hipStream_t stream)
{
    if (count == 0) return;
    if (count > (uint32_t)kMaxMany) {
        set_error("mrl_step_many: at most %d simulators per launch, got %u", kMaxMany, count);
        throw HipError{MRL_ERR_INVALID};
    }
    ManyParams m{};
    uint32_t blocks = 0, lds = 0;
    for (uint32_t k = 0; k < count; k++) {
        auto *sim = static_cast<OvercookedSim *>(sims[k]);
        if (sims[k]->game != MRL_GAME_OVERCOOKED || sims[k]->device != sims[0]->device || sim->params.team) {
            set_error("mrl_step_many: simulator %u is not an Overcooked simulator on device %d that the generic step kernel can run "
                      "(few worlds of a large layout share one state copy per workgroup: step those on their own)", k, sims[0]->device);
            throw HipError{MRL_ERR_INVALID};
        }
        for (uint32_t j = 0; j < k; j++)
            if (sims[j] == sims[k]) {
                set_error("mrl_step_many: simulator %u is listed twice", k);
                throw HipError{MRL_ERR_INVALID};
            }
        m.first_block[k] = blocks;
        blocks += sim->grid;
        lds = std::max(lds, sim->lds_bytes);
    }
    m.first_block[count] = blocks;
    m.count = count;
    for (uint32_t k = 0; k < count; k++)
        m.sim[k] = static_cast<OvercookedSim *>(sims[k])->generic_step_params(actions_or_null ? actions_or_null[k] : nullptr);
    // hipFuncSetAttribute applies to the current device only and this entry point may be called for several devices and from
    // several threads: no process-wide cache of "already raised" -- the call costs little next to the launch
    if (lds > 65536)
        MRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mrl_overcooked_step_many), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mrl_overcooked_step_many, dim3(blocks), dim3(kBlock), lds, stream, m);
    MRL_HIP(hipGetLastError());
    for (uint32_t k = 0; k < count; k++) {
        auto *sim = static_cast<OvercookedSim *>(sims[k]);
        sim->deliver(sim->pending_dest, stream);  // (simulators whose slots are staged: OvercookedSim::set_observation_ring)
    }
}
