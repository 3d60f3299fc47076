// "Simplecooked" (the reference's overcooked2_env) world step for gfx950: the whole step fused in one
// kernel, one wavefront per GROUP of worlds, built like overcooked.hip's single-pass path.
//
// Semantics follow the reference task graph
//   /root/reference/src/overcooked2_env/sim.cpp:422-451
// (sequential resolve_interacts :199-287 with get_pot_states :174-185 and is_dish_pickup_useful
// :187-197, movement :289-348, pots :350-360, reset :362-420, observation rows :62-148, init :470-575)
// with the component widths of sim.hpp:55-189.  What differs from overcooked_env and shapes this file:
//   * terrain enum AIR, POT, COUNTER, ONION_SOURCE, DISH_SOURCE, SERVING, TOMATO_SOURCE (sim.hpp:40);
//   * rows are F = 5P + 10 bytes; at most 2 players and 100 cells (sim.hpp:12-13), so a group's whole
//     observation slab always fits one LDS tile: there is only the single-pass encode;
//   * interactions are strictly sequential in player order (one graph node), a pot starts cooking by
//     itself with its third ingredient, there is no "start cooking" interaction, and picking up a dish
//     pays dish_pickup_rew when no dish lies on a counter (WorldState.num_dishes_out == 0) and fewer
//     players hold a dish than there are pots that could use one;
//   * observation channel 5P+5 is zeroed on every pass (sim.cpp:74), which wipes the TOMATO_SOURCE
//     terrain bit that shares it: that bit is never visible; no urgency channel.
//
// Mapping: lane = (world of the group, player), pairs exchange through DPP (swap_pair).  Player 1's
// interaction depends on player 0's in three ways, all carried by a few exchanged words:
//   same counter / pot      -> player 1 runs in a second round (rank 1), exactly like overcooked.hip;
//   dishes on counters      -> player 0's put/take of a dish changes num_dishes_out before player 1's
//                              dish pickup is rated: the delta is exchanged;
//   dishes in hands         -> the rating counts dishes held "now": player 0 sees player 1's hand as it
//                              was, player 1 sees player 0's hand after its interaction.
// With one player there is nobody to exchange with and, as in the reference's numpy twin
// (envs/overcooked2_reimplement.py:243-245), a dish pickup is never rated useful.
//
// HBM layout, world-major like overcooked.hip:
//   cell_obj [N][C] u32 name | onions<<8 | tomatoes<<16 | tick<<24;  players [N][P] 2xu32;
//   clock [N] 2xi32 {timestep, num_dishes_out};  action [P][N] i32;  reward [P][N] i32;  done [N] i32;
//   obs [N][P][C][F] u8, one contiguous block per world.
#include "common.hpp"
#include "grid_common.hpp"
#include "random_policy.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

namespace {

using namespace mrl_grid;

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;

enum : uint32_t { T_AIR = 0, T_POT, T_COUNTER, T_ONION_SRC, T_DISH_SRC, T_SERVING, T_TOMATO_SRC };  // sim.hpp:40

constexpr uint32_t kMaxCells = 100, kMaxPlayers = 2;  // MAX_SIZE, MAX_NUM_PLAYERS (sim.hpp:12-13)
// constant block copied into LDS by every workgroup
constexpr uint32_t kConstTerrain = 0;   // 112 bytes
constexpr uint32_t kConstPots = 112;    // 112 bytes: cells holding a pot
constexpr uint32_t kConstBytes = 224;

struct SimpleParams {
    uint32_t num_worlds;
    uint32_t P, C, W, F;
    uint64_t deltas;
    uint32_t rows;         // P*C
    uint32_t block_bytes;  // P*C*F
    uint32_t inv_c, inv_rows;
    uint32_t placement_rew, dish_rew, soup_pickup_rew;
    uint32_t times_w[4], values_w[4];
    uint32_t starts;       // start cell of player 0 | player 1 << 8
    uint32_t wpw, num_pots;
    uint32_t off_pl, off_cur, off_list, off_tile, lds_wave_stride;
    int64_t horizon;
    const uint32_t *consts;
    const uint16_t *terr_pos;  // [rows]: offset of a row's terrain one-hot byte inside a world's block, 0xFFFF = none
    // flat: every group's slab starts on a 16-byte boundary and its rows fit kGroupTerrPerLane x 64, so the static
    // part of the tile is built like overcooked.hip's: ds_write_addtid zero fill + one table of tile offsets per GROUP
    uint32_t flat;
    const uint16_t *terr_off;  // [wpw * rows]: tile offset of a group row's terrain byte, 0 = none
    uint32_t terr_entries;
    // direct (the single step of flat two-player groups): the dynamic rows come from the player lanes and a host-built
    // table of the group's HOLDER cells (counters and pots a player can face) instead of a search -- overcooked.hip,
    // patch_direct.  Entry [k * 64 + lane]: tile offset of the cell's viewer-0 row | cell index in the group << 16 |
    // 1 << 30 | is_pot << 31; the lanes below wpw * P of round 0 are left to the players.
    uint32_t direct;
    const uint32_t *hold_tab;
    uint32_t hold_entries;
    uint32_t per_xcd;  // workgroups of the launch / 8
    uint32_t *cell_obj;
    uint2 *players;
    int2 *clock;  // {timestep, num_dishes_out}
    const int32_t *actions;
    const long long *actions64;  // mrl_step_with_actions_i64 (NULL otherwise); mirrored into the ACTION tensor through action_out
    int32_t *reward;
    int32_t *done;
    uint8_t *obs;
    // mrl_set_observation_ring (multi-step launches): step k of the launch writes slot (ring_first + k) % ring_slots,
    // ring_stride bytes apart from `obs`; ring_slots <= 1: every step writes `obs`
    uint64_t ring_stride;
    uint32_t ring_slots, ring_first;
    // device-side random policy (mrl_rollout_random): actions drawn in the kernel when sample != 0
    uint32_t sample, sample_step;
    uint64_t sample_seed;
    int32_t *action_out;
};

// One player's interaction (sim.cpp:214-283) as straight-line selects.  `there` is the object on the faced
// cell (meaningful for counters and pots); returns the new held item.  d_dishes: change of num_dishes_out;
// grabbed_dish: a dish was taken from the dish source (rated by the caller, which knows the other hand).
__device__ __forceinline__ uint32_t interact(const SimpleParams &p, uint32_t terr, uint32_t held, uint32_t &there, int32_t &reward,
                                             int32_t &d_dishes, bool &grabbed_dish)
{
    const uint32_t hname = held & 0xFF, oname = there & 0xFF;
    const int32_t tick = (int8_t)(there >> 24);
    const int32_t need = (int32_t)lookup16(p.times_w, recipe_of(there));
    const bool counter = terr == T_COUNTER, pot = terr == T_POT;
    const bool empty_handed = hname == O_NONE, nothing_there = oname == O_NONE;
    const bool put = counter & !empty_handed & nothing_there;
    const bool take = counter & empty_handed & !nothing_there;
    const bool plate = pot & (hname == O_DISH) & (oname == O_SOUP) & (tick >= 0) & (tick >= need);
    const bool ingredient = pot & ((hname == O_ONION) | (hname == O_TOMATO));
    const uint32_t soup = nothing_there ? (O_SOUP | kItemNone) : there;
    const bool add = ingredient & !(((int8_t)(soup >> 24) >= 0) | (count_of(soup) == kMaxIngredients));
    uint32_t soup2 = add ? soup + (hname == O_ONION ? 0x100u : 0x10000u) : soup;
    // the third ingredient starts the pot (soup_to_be_cooked_at_location && full, sim.cpp:268-270)
    const bool autocook = ingredient & ((int8_t)(soup2 >> 24) < 0) & (count_of(soup2) == kMaxIngredients);
    soup2 = autocook ? (soup2 & 0x00FFFFFFu) : soup2;
    const bool grab_onion = (terr == T_ONION_SRC) & empty_handed, grab_tomato = (terr == T_TOMATO_SRC) & empty_handed;
    grabbed_dish = (terr == T_DISH_SRC) & empty_handed;
    const bool serve = (terr == T_SERVING) & (hname == O_SOUP);
    const int32_t value = (int32_t)lookup16(p.values_w, recipe_of(held));

    uint32_t new_there = there;
    new_there = ingredient ? soup2 : new_there;
    new_there = put ? held : new_there;
    new_there = (take | plate) ? kItemNone : new_there;
    uint32_t new_held = held;
    new_held = (take | plate) ? there : new_held;
    new_held = grab_onion ? (O_ONION | kItemNone) : new_held;
    new_held = grab_tomato ? (O_TOMATO | kItemNone) : new_held;
    new_held = grabbed_dish ? (O_DISH | kItemNone) : new_held;
    new_held = (put | add | serve) ? kItemNone : new_held;
    reward += (plate ? (int32_t)p.soup_pickup_rew : 0) + (add ? (int32_t)p.placement_rew : 0) + (serve ? value : 0);
    d_dishes = (put & (hname == O_DISH)) ? 1 : ((take & (oname == O_DISH)) ? -1 : 0);
    there = new_there;
    return new_held;
}

// The whole transition of a group: lane = wl * P + q.  Returns the world's reward and the new
// num_dishes_out to every lane of the world.
template <int kP>
__device__ __forceinline__ void transition(const SimpleParams &p, const uint8_t *s_terrain, const uint8_t *s_pots, uint32_t *s_obj,
                                           bool active, uint32_t wl, uint32_t q, uint32_t a, uint32_t &posori, uint32_t &held,
                                           int32_t &dishes_out, int32_t &reward_world)
{
    const uint32_t C = p.C;
    const uint32_t pos = posori & 0xFFu, ori = (posori >> 8) & 0xFFu;
    const uint32_t tgt = pos + (uint32_t)step_of(ori, p.deltas);
    const uint32_t terr = s_terrain[tgt];
    const uint32_t ahead = s_terrain[pos + (uint32_t)step_of(a, p.deltas)];
    const bool inter = active && a == A_INTERACT;
    const bool touches = inter && (terr == T_COUNTER || terr == T_POT);
    // get_pot_states (sim.cpp:174-185), before any interaction of this step; the lanes of a world agree
    int32_t pot_states = 0;
    if (active)
        for (uint32_t k = 0; k < p.num_pots; k++) {
            const uint32_t o = s_obj[wl * C + s_pots[k]];
            pot_states += ((o & 0xFF) != O_NONE && ((int8_t)(o >> 24) >= 0 || count_of(o) < kMaxIngredients)) ? 1 : 0;
        }
    const uint32_t held_before = held;
    uint32_t rank = 0, other_before = kItemNone;
    if constexpr (kP == 2) {
        const uint32_t other_key = swap_pair(touches ? tgt : 0xFFFFu);
        other_before = swap_pair(held_before);
        rank = (q == 1u && touches && other_key == tgt) ? 1u : 0u;
    }
    int32_t mine = 0, d_dishes = 0;
    bool grabbed = false;
    uint32_t *cell = s_obj + (touches ? wl * C + tgt : 0u);
#pragma unroll
    for (uint32_t r = 0; r < (uint32_t)kP; r++) {
        const bool todo = inter && rank == r;
        if (r > 0 && __ballot(todo) == 0ull) break;
        if (todo) {
            uint32_t there = touches ? *cell : kItemNone;
            held = interact(p, terr, held, there, mine, d_dishes, grabbed);
            if (touches) *cell = there;
        }
        wave_lds_sync();
    }
    // dish pickup shaping (sim.cpp:241-246): rated with num_dishes_out and the hands as they are when the
    // player's turn comes: player 1 after player 0's interaction, player 0 before player 1's
    if constexpr (kP == 2) {
        const uint32_t other_after = swap_pair(held);
        const int32_t other_d = (int32_t)swap_pair((uint32_t)d_dishes);
        const uint32_t other_hand = q == 0 ? other_before : other_after;
        const int32_t out_before_me = dishes_out + (q == 0 ? 0 : other_d);
        const int32_t dishes_held = (other_hand & 0xFF) == O_DISH ? 1 : 0;  // the grabbing hand itself was empty
        if (grabbed && out_before_me == 0 && dishes_held < pot_states) mine += (int32_t)p.dish_rew;
        reward_world = mine + (int32_t)swap_pair((uint32_t)mine);
        dishes_out += d_dishes + other_d;
    } else {
        reward_world = mine;  // one player: never rated useful (overcooked2_reimplement.py:243-245)
        dishes_out += d_dishes;
    }
    (void)held_before;
    // movement proposal (sim.cpp:291-309); orientation := action unless STAY / INTERACT
    const bool moves = a != A_INTERACT;
    const uint32_t pori = (moves && a != A_STAY) ? a : ori;
    const uint32_t prop = (moves && ahead == T_AIR) ? pos + (uint32_t)step_of(a, p.deltas) : pos;
    bool blocked = false;
    if constexpr (kP == 2) {  // same target or swapped cells -> nobody moves (sim.cpp:311-348)
        const uint32_t o = swap_pair(pos | (prop << 8));
        const uint32_t opos = o & 0xFFu, oprop = o >> 8;
        blocked = (prop == oprop) | ((prop == opos) & (pos == oprop));
    }
    posori = (blocked ? pos : prop) | (pori << 8);
}

// pots (sim.cpp:350-360), after the interactions; lane = world
__device__ __forceinline__ void tick_pots(const SimpleParams &p, const uint8_t *s_pots, uint32_t *s_obj, uint32_t nw, uint32_t lane)
{
    if (lane < nw) {
        uint32_t *obj = s_obj + lane * p.C;
        for (uint32_t k = 0; k < p.num_pots; k++) {
            const uint32_t c = s_pots[k];
            const uint32_t o = obj[c];
            const int32_t tick = (int8_t)(o >> 24);
            if ((o & 0xFF) == O_SOUP && tick >= 0 && tick < (int32_t)lookup16(p.times_w, recipe_of(o)))
                obj[c] = (o & 0x00FFFFFFu) | ((uint32_t)(uint8_t)(tick + 1) << 24);
        }
    }
}

// The 10 viewer-independent bytes of a cell's rows, row[5P .. 5P+10) (sim.cpp:62-148; terrain bytes :553-558):
// terrain one-hot (5; channel 5 is the zeroed one), pot soup onions, cooking tick, soup not in a pot,
// dish, onion -- the last three also for what the player standing there holds.  Returned as bytes 0-3, 4-7, 8-9.
struct Tail10 {
    uint32_t a, b, c;
};
__device__ __forceinline__ Tail10 cell_tail(uint32_t terr, uint32_t o, uint32_t h)
{
    const uint32_t oname = o & 0xFF, on = (o >> 8) & 0xFF, hname = h & 0xFF;
    const int32_t tick = (int8_t)(o >> 24);
    const bool is_soup = oname == O_SOUP, in_pot = terr == T_POT;
    const uint32_t pot_on = (is_soup & in_pot) ? on : 0u;
    const uint32_t pot_tick = (is_soup & in_pot & (tick >= 0)) ? (uint32_t)tick & 0xFFu : 0u;
    const uint32_t soup = ((is_soup & !in_pot) | (hname == O_SOUP)) ? 1u : 0u;
    const uint32_t dish = ((oname == O_DISH) | (hname == O_DISH)) ? 1u : 0u;
    const uint32_t onion = ((oname == O_ONION) | (hname == O_ONION)) ? 1u : 0u;
    Tail10 t;
    t.a = (terr >= 1 && terr <= 4) ? (1u << ((terr - 1u) * 8u)) : 0u;
    t.b = (terr == 5 ? 1u : 0u) | (pot_on << 8) | (pot_tick << 16) | (soup << 24);
    t.c = dish | (onion << 8);
    return t;
}

// row + 5P for P = 2 is always 2 (mod 4) in the tile (rows 20 bytes apart, worlds 40 C apart, the tile as
// misaligned as the slab in HBM, i.e. by 0 or 8): halfword, dword, dword -- all naturally aligned.
__device__ __forceinline__ void lds_store_tail10_p2(uint8_t *ptr, const Tail10 &t)
{
    const uint32_t mid = __builtin_amdgcn_alignbit(t.b, t.a, 16);  // bytes 2..5
    const uint32_t end = __builtin_amdgcn_alignbit(t.c, t.b, 16);  // bytes 6..9
    asm volatile("ds_write_b16 %0, %1\n\t"
                 "ds_write_b32 %0, %2 offset:2\n\t"
                 "ds_write_b32 %0, %3 offset:6"
                 :
                 : "v"(lds_addr(ptr)), "v"(t.a), "v"(mid), "v"(end)
                 : "memory");
}
__device__ __forceinline__ void lds_store_tail10_bytes(uint8_t *ptr, const Tail10 &t)
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ptr[k] = (uint8_t)(t.a >> (8 * k));
        ptr[4 + k] = (uint8_t)(t.b >> (8 * k));
    }
    ptr[8] = (uint8_t)t.c;
    ptr[9] = (uint8_t)(t.c >> 8);
}

constexpr int kHoldPerLane = 4;     // 256 holder-table entries per group (direct)
constexpr int kTerrPosPerLane = 4;  // rows of one world <= 2 * 100
constexpr int kGroupTerrPerLane = 7;  // rows of one GROUP in the flat mode (the standard layouts need 320..400)

template <bool kInit, int kP, bool kPlain = false>
__device__ __forceinline__ void step_body(const SimpleParams &p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    // constants and the group's state are requested together: one HBM/L2 latency
    const uint32_t const_word = tid < kConstBytes / 4 ? p.consts[tid] : 0u;
    const uint8_t *s_terrain = smem + kConstTerrain;
    const uint8_t *s_pots = smem + kConstPots;

    // XCD-aware mapping: one contiguous world range per XCD (workgroups are dealt round-robin)
    const uint32_t per_xcd = p.per_xcd;
    const uint32_t logical_block = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const uint32_t w0 = (logical_block * kWavesPerBlock + wib) * p.wpw;
    const uint32_t nw = w0 < p.num_worlds ? min(p.wpw, p.num_worlds - w0) : 0u;

    uint8_t *wbase = smem + kConstBytes + wib * p.lds_wave_stride;
    uint32_t *s_obj = reinterpret_cast<uint32_t *>(wbase);                 // [wpw][C]
    uint32_t *s_pl = reinterpret_cast<uint32_t *>(wbase + p.off_pl);       // [wpw][P][2]
    uint8_t *s_cur = wbase + p.off_cur;                                    // [wpw][C] cell -> player
    uint16_t *s_list = reinterpret_cast<uint16_t *>(wbase + p.off_list);   // [wpw][C] dynamic cells
    uint8_t *s_tile = wbase + p.off_tile;

    constexpr uint32_t P = (uint32_t)kP;
    const uint32_t C = p.C, N = p.num_worlds, F = p.F;
    const uint32_t ncells = nw * C, nplayers = nw * P;
    const uint32_t wl = kP == 2 ? lane >> 1 : lane;
    const uint32_t q = kP == 2 ? lane & 1u : 0u;
    const bool active = lane < nplayers;
    const uint32_t world = min(w0 + wl, N - 1u);

    uint32_t posori = 0, held = kItemNone, act = A_STAY;
    int2 clock = make_int2(0, 0);
    uint32_t tpos[kTerrPosPerLane];
    uint32_t goff[kGroupTerrPerLane];
    uint32_t hold[kHoldPerLane] = {};
    const bool direct = kP == 2 && p.direct != 0;
    auto hold_request = [&]() {
        const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.hold_tab), 0, (int)(p.hold_entries * 4u), 0x00020000);
#pragma unroll
        for (int k = 0; k < kHoldPerLane; k++)
            hold[k] = (uint32_t)k * kWave < p.hold_entries ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(tab, (int)((lane + (uint32_t)k * kWave) * 4u), 0, 0) : 0u;
    };
    auto static_request = [&]() {
        if (p.flat) {
            const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.terr_off), 0, (int)(p.terr_entries * 2u), 0x00020000);
#pragma unroll
            for (int k = 0; k < kGroupTerrPerLane; k++)
                goff[k] = (uint32_t)k * kWave < p.terr_entries ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(tab, (int)((lane + (uint32_t)k * kWave) * 2u), 0, 0) : 0u;
        } else {
#pragma unroll
            for (int k = 0; k < kTerrPosPerLane; k++) tpos[k] = p.terr_pos[min(lane + (uint32_t)k * kWave, p.rows - 1u)];
        }
    };
    auto static_zero = [&]() {
        if (p.flat)
            tile_zero_addtid(s_tile, nw * p.block_bytes);
        else
            for (uint32_t k = lane; k < (nw * p.block_bytes + 31u) >> 4; k += kWave) reinterpret_cast<uint4 *>(s_tile)[k] = make_uint4(0, 0, 0, 0);
    };
    // ---------------- load ----------------
    if (!kInit) {
        const uint32_t *g_obj = p.cell_obj + (size_t)w0 * C;
        constexpr int kBatch = 4;
        uint32_t cell_reg[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const uint32_t i = lane + k * kWave;
            cell_reg[k] = i < ncells ? g_obj[i] : 0u;
        }
        // unconditional, clamped indices: behind a branch hipcc consumes a load on the spot
        const uint2 pl_reg = p.players[(size_t)min(w0, N - 1u) * P + (active ? lane : 0u)];  // idle waves (w0 >= N: the grid is rounded up) stay in bounds
        uint32_t a_raw;
        if (p.sample)  // uniform over the six actions (include/mrl_envs.h: mrl_rollout_random)
            a_raw = mrl::scale(mrl::policy_hash(p.sample_seed, p.sample_step, world, q), 6u);
        else {
            // int32, or int64 narrowed to its low word -- only that word is loaded; one base pointer and a shift rather
            // than two element loads to choose from (overcooked.hip, load_action)
            const bool wide = p.actions64 != nullptr;
            const char *base = wide ? reinterpret_cast<const char *>(p.actions64) : reinterpret_cast<const char *>(p.actions);
            a_raw = *reinterpret_cast<const uint32_t *>(base + (((size_t)q * N + world) << (wide ? 3 : 2)));
        }
        clock = p.clock[world];
        static_request();
        // while the loads are in flight: empty cell -> player map, zeroed tile
        if (!direct)
            for (uint32_t i = lane; i < (p.wpw * C + 3u) >> 2; i += kWave) reinterpret_cast<uint32_t *>(s_cur)[i] = 0xFFFFFFFFu;
        static_zero();
        if (direct) hold_request();
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const uint32_t i = lane + k * kWave;
            if (i < ncells) s_obj[i] = cell_reg[k];
        }
        for (uint32_t i = lane + kBatch * kWave; i < ncells; i += kWave) s_obj[i] = g_obj[i];
        posori = active ? pl_reg.x & 0xFFFFu : 0u;
        held = active ? pl_reg.y : kItemNone;
        act = (active && a_raw <= A_INTERACT) ? a_raw : (uint32_t)A_STAY;  // outside the enum = outside the contract
        if ((p.sample || p.actions64) && active) p.action_out[(size_t)q * N + world] = (int32_t)(p.sample ? act : a_raw);
    } else {
        static_request();
        if (!direct)
            for (uint32_t i = lane; i < (p.wpw * C + 3u) >> 2; i += kWave) reinterpret_cast<uint32_t *>(s_cur)[i] = 0xFFFFFFFFu;
        static_zero();
        if (direct) hold_request();
    }
    // the group's slab in HBM and its image in the tile are equally misaligned, so 16-byte chunks line up
    uint8_t *gobs = p.obs + (size_t)w0 * p.block_bytes;
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(gobs) & 15u);
    uint8_t *tile = s_tile + mis;
    // static part of the tile: the terrain one-hot byte of every row of a non-AIR, non-tomato-source cell
    if (p.flat) {
        const uint32_t limit = nw * p.block_bytes;  // a ragged last group holds fewer worlds
#pragma unroll
        for (int k = 0; k < kGroupTerrPerLane; k++)
            if ((uint32_t)k * kWave < p.terr_entries && goff[k] != 0u && goff[k] < limit) tile[goff[k]] = 1;
    } else {
#pragma unroll
        for (int k = 0; k < kTerrPosPerLane; k++) {
            const uint32_t i = lane + (uint32_t)k * kWave;
            if (i < p.rows && tpos[k] != 0xFFFFu)
                for (uint32_t l = 0; l < nw; l++) tile[__umul24(l, p.block_bytes) + tpos[k]] = 1;
        }
    }
    if (tid < kConstBytes / 4) reinterpret_cast<uint32_t *>(smem)[tid] = const_word;
    __syncthreads();
    if (nw == 0) return;

    // ---------------- step: lane = (world, player) ----------------
    int32_t reward_world = 0, t = 0, dishes_out = 0;
    bool reset_now = kInit;
    if (!kInit) {
        dishes_out = clock.y;
        transition<kP>(p, s_terrain, s_pots, s_obj, active, wl, q, act, posori, held, dishes_out, reward_world);
        if (!direct) tick_pots(p, s_pots, s_obj, nw, lane);  // direct: the holder lanes of the encode tick the pots
        t = clock.x + 1;  // sim.cpp:415-420
        reset_now = (int64_t)t >= p.horizon;
    }
    if (__builtin_expect(__ballot(active && reset_now) != 0ull, 0)) {  // sim.cpp:362-413
        if (reset_now) {
            t = 0;
            dishes_out = 0;
            posori = ((p.starts >> (8u * q)) & 0xFFu) | (A_NORTH << 8);
            held = kItemNone;
        }
        // the flags travel through the (not yet used) dynamic-cell list
        if (active && q == 0) s_list[wl] = reset_now ? 1 : 0;
        wave_lds_sync();
        for (uint32_t i = lane; i < ncells; i += kWave)
            if (s_list[__umulhi(i, p.inv_c)] != 0) s_obj[i] = kItemNone;
        wave_lds_sync();
    }
    if (active && !direct) {
        reinterpret_cast<uint2 *>(s_pl)[lane] = make_uint2(posori, held);
        s_cur[wl * C + (posori & 0xFFu)] = (uint8_t)q;
    }
    wave_lds_sync();

    // ---------------- observe: only the dynamic cells (sim.cpp:62-148) ----------------
    const uint32_t plane = __umul24(C, F), shift = 5 * P;
    if constexpr (kP == 2) {
        if (direct) {
            // players stand on AIR cells only and objects lie on holder cells only: lane = a player's cell (round 0) or a
            // holder cell; its 10-byte tail goes into the rows of both viewers
            const uint32_t ori = (posori >> 8) & 0xFFu;
#pragma unroll
            for (int k = 0; k < kHoldPerLane; k++) {
                if (k > 0 && (uint32_t)k * kWave >= p.hold_entries) break;  // wave-uniform
                const uint32_t e = hold[k];
                const uint32_t i = (e >> 16) & 0x3FFFu;
                const bool holder = ((e >> 30) & 1u) != 0u && i < ncells;
                uint32_t o = s_obj[holder ? i : 0u];
                if (holder && (e >> 31)) {
                    // the pot's tick (sim.cpp:350-360) rides on this read instead of tick_pots' own LDS round trip: every pot
                    // that can hold a soup is a holder cell; a pot emptied by this step's reset does not tick, like an empty one
                    const int32_t tick = (int8_t)(o >> 24);
                    if ((o & 0xFFu) == O_SOUP && tick >= 0 && tick < (int32_t)lookup16(p.times_w, recipe_of(o))) {
                        o = (o & 0x00FFFFFFu) | ((uint32_t)(uint8_t)(tick + 1) << 24);
                        s_obj[i] = o;
                    }
                }
                const bool player = k == 0 && active;
                if (player || (holder && (o & 0xFFu) != O_NONE)) {
                    const Tail10 t10 = cell_tail(player ? (uint32_t)T_AIR : ((e >> 31) ? (uint32_t)T_POT : (uint32_t)T_COUNTER),
                                                 player ? (uint32_t)kItemNone : o, player ? held : (uint32_t)kItemNone);
                    uint8_t *row0 = tile + (player ? __umul24(wl, p.block_bytes) + __umul24(posori & 0xFFu, F) : (e & 0xFFFFu));
#pragma unroll
                    for (uint32_t v = 0; v < 2; v++) {
                        uint8_t *row = row0 + v * plane;
                        lds_store_tail10_p2(row + shift, t10);
                        if (player) {
                            const uint32_t rel = q == v ? 0u : 1u;
                            row[rel] = 1;
                            row[P + 4 * rel + ori] = 1;
                        }
                    }
                }
            }
        }
    }
    uint32_t ndyn = 0;
    for (uint32_t i0 = 0; i0 < (direct ? 0u : ncells); i0 += kWave) {
        const uint32_t i = i0 + lane;
        const bool valid = i < ncells;
        const uint32_t o = s_obj[valid ? i : 0u];
        const uint32_t who = s_cur[valid ? i : 0u];
        const bool dyn = valid && (((o & 0xFFu) != O_NONE) | (who != 0xFFu));
        const unsigned long long m = __ballot(dyn);
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (dyn) s_list[ndyn + before] = (uint16_t)i;
        ndyn += (uint32_t)__popcll(m);
    }
    wave_lds_sync();
    const uint32_t nent = ndyn * P;
    for (uint32_t j = lane; j < nent; j += kWave) {
        const uint32_t k = kP == 2 ? j >> 1 : j, v = kP == 2 ? j & 1u : 0u;
        const uint32_t i = s_list[k];
        const uint32_t l = __umulhi(i, p.inv_c), c = i - __umul24(l, C);
        const uint32_t who = s_cur[i];
        const bool occupied = who != 0xFF;
        const uint32_t pidx = (__umul24(l, P) + (occupied ? who : 0u)) * 2;
        const uint32_t w_ori = (s_pl[pidx] >> 8) & 0xFF;
        const uint32_t h = occupied ? s_pl[pidx + 1] : kItemNone;
        const Tail10 t10 = cell_tail(s_terrain[c], s_obj[i], h);
        uint8_t *row = tile + __umul24(l, p.block_bytes) + __umul24(v, plane) + __umul24(c, F);
        if constexpr (kP == 2)
            lds_store_tail10_p2(row + shift, t10);
        else
            lds_store_tail10_bytes(row + shift, t10);
        if (occupied) {
            const uint32_t rel = who == v ? 0u : (who < v ? who + 1u : who);
            row[rel] = 1;
            row[P + 4 * rel + w_ori] = 1;
        }
    }
    wave_lds_sync();
    {   // stream the slab out: unaligned head bytes, 16-byte body as bounds-checked buffer stores, tail bytes
        const uint32_t nbytes = nw * p.block_bytes;
        const uint32_t head = min((16u - mis) & 15u, nbytes);
        if (lane < head) gobs[lane] = tile[lane];
        const uint32_t body = (nbytes - head) >> 4;
        const uint4 *src = reinterpret_cast<const uint4 *>(tile + head);
        const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(gobs + head, 0, (int)(body << 4), 0x00020000);
        for (uint32_t k0 = lane; k0 < body + lane; k0 += 4 * kWave) {
            const uint32_t ka = k0, kb = k0 + kWave, kc = k0 + 2 * kWave, kd = k0 + 3 * kWave;
            const uint32_t last = body - 1u;
            const uint4 va = src[min(ka, last)], vb = src[min(kb, last)], vc = src[min(kc, last)], vd = src[min(kd, last)];
            stream_store_rsrc<kPlain>(out, ka << 4, va);
            stream_store_rsrc<kPlain>(out, kb << 4, vb);
            stream_store_rsrc<kPlain>(out, kc << 4, vc);
            stream_store_rsrc<kPlain>(out, kd << 4, vd);
        }
        const uint32_t done_bytes = head + (body << 4);
        if (lane < nbytes - done_bytes) gobs[done_bytes + lane] = tile[done_bytes + lane];
    }

    // ---------------- store: the LAST thing a wave does (see overcooked.hip) ----------------
    uint32_t *g_obj = p.cell_obj + (size_t)w0 * C;
    for (uint32_t i = lane; i < ncells; i += kWave) g_obj[i] = s_obj[i];
    if (active) {
        p.players[(size_t)w0 * P + lane] = make_uint2(posori, held);
        p.reward[(size_t)q * N + world] = reward_world;
        if (q == 0) {
            p.clock[world] = make_int2(t, dishes_out);
            p.done[world] = kInit ? 0 : (int32_t)reset_now;
        }
    }
}

template <bool kInit, int kP, bool kPlain = false>
__global__ void __launch_bounds__(kBlock) mrl_simplecooked_step(const SimpleParams p)
{
    step_body<kInit, kP, kPlain>(p);
}

// The two-player step for ONE layout size known at compile time (kC cells in rows of kWidth, kPots pots, kW worlds
// per wave): every size-dependent kernel argument becomes a constant, so divisions by the cell count, loop trip counts
// and LDS offsets fold -- the kernel is bound by instruction issue, not by bytes.  Launched only when
// the simulator's parameters are exactly these; results are identical.
constexpr uint32_t up16c(uint32_t v) { return (v + 15u) & ~15u; }
// size of a group's holder table: kHold holder cells per world, round 0 keeps 2 * kW lanes for the players
constexpr uint32_t fixed_hold_entries(uint32_t wpw, uint32_t holders)
{
    const uint32_t free0 = 64u - 2u * wpw, total = wpw * holders;
    return total == 0 ? 1u : (total <= free0 ? 2u * wpw + total : 64u + (total - free0));
}
template <int kC, int kW, int kWidth, int kPots, int kHold>
__device__ __forceinline__ SimpleParams fixed_simple_params(const SimpleParams &p)
{
    SimpleParams q = p;
    q.P = 2;
    q.C = kC;
    q.W = kWidth;
    q.F = 20;
    q.deltas = pack_deltas(kWidth);
    q.rows = 2 * kC;
    q.block_bytes = 2 * kC * 20;
    q.inv_c = (uint32_t)((1ull << 32) / (uint64_t)kC) + 1u;
    q.inv_rows = (uint32_t)((1ull << 32) / (uint64_t)(2 * kC)) + 1u;
    q.wpw = kW;
    q.num_pots = kPots;
    q.off_pl = up16c(kW * kC * 4);
    q.off_cur = q.off_pl + up16c(kW * 2 * 8);
    q.off_list = q.off_cur + up16c(kW * kC);
    q.off_tile = q.off_list + up16c(kW * kC * 2 > 128 ? kW * kC * 2 : 128);
    q.flat = 1;  // (kW * 2 * kC * 20) % 16 == 0 for the four sizes below
    q.terr_entries = kW * 2 * kC;
    q.direct = 1;
    q.hold_entries = fixed_hold_entries(kW, kHold);
    q.lds_wave_stride = q.off_tile + ((kW * 2 * kC * 20 + 255u) & ~255u) + 48u;
    return q;
}

// kSource: where the actions come from -- 0 the int32 array, 1 the caller's int64 tensor, 2 drawn in the kernel
// The first fourteen argument dwords are separate scalars so that the command processor preloads them into SGPRs
// (-amdgpu-kernarg-preload-count in the Makefile; overcooked.hip, MRL_HOT_ARGS): what a wave needs to find its worlds
// and request its loads.  The struct carries everything else.
template <int kC, int kW, int kWidth, int kPots, int kHold, int kSource = 0, bool kPlain = false>
__global__ void __launch_bounds__(kBlock)
    mrl_simplecooked_step_fixed(uint32_t *hot_cell_obj, uint2 *hot_players, int2 *hot_clock, const void *hot_actions, const uint32_t *hot_consts,
                                const uint16_t *hot_terr_off, uint32_t hot_num_worlds, uint32_t hot_per_xcd, const SimpleParams p)
{
    SimpleParams q = fixed_simple_params<kC, kW, kWidth, kPots, kHold>(p);
    q.cell_obj = hot_cell_obj;
    q.players = hot_players;
    q.clock = hot_clock;
    q.actions = kSource == 0 ? static_cast<const int32_t *>(hot_actions) : nullptr;
    q.actions64 = kSource == 1 ? static_cast<const long long *>(hot_actions) : nullptr;
    q.consts = hot_consts;
    q.terr_off = hot_terr_off;
    q.num_worlds = hot_num_worlds;
    q.per_xcd = hot_per_xcd;
    if (kSource != 2) q.sample = 0;
    step_body<false, 2, kPlain>(q);
}

// ---------------------------------------------------------------------------------------------
// num_steps steps in ONE launch (two players, flat tile): mrl_rollout_random draws the uniform random policy in the
// kernel (the same stream as the per-step path: policy_hash(seed, first_step + k, world, player) scaled to six
// actions), mrl_step_sequence reads an open-loop (num_steps, P, N) action array.  Like overcooked.hip's multi-step
// launches: cell objects in LDS, players and clocks in registers, the observation tile zeroed and given its terrain
// bytes once -- a step patches what is dynamic now, streams the tile out and puts the patched rows back.
// ---------------------------------------------------------------------------------------------
template <bool kPlain>
__device__ __forceinline__ void rollout_body(const SimpleParams &p, uint32_t num_steps, uint64_t seed, uint32_t first_step,
                                             const int32_t *action_seq)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t P = 2;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t const_word = tid < kConstBytes / 4 ? p.consts[tid] : 0u;
    const uint8_t *s_terrain = smem + kConstTerrain;
    const uint8_t *s_pots = smem + kConstPots;
    const uint32_t per_xcd = p.per_xcd;
    const uint32_t logical_block = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const uint32_t w0 = (logical_block * kWavesPerBlock + wib) * p.wpw;
    const uint32_t nw = w0 < p.num_worlds ? min(p.wpw, p.num_worlds - w0) : 0u;
    uint8_t *wbase = smem + kConstBytes + wib * p.lds_wave_stride;
    uint32_t *s_obj = reinterpret_cast<uint32_t *>(wbase);
    uint32_t *s_pl = reinterpret_cast<uint32_t *>(wbase + p.off_pl);
    uint8_t *s_cur = wbase + p.off_cur;
    uint16_t *s_list = reinterpret_cast<uint16_t *>(wbase + p.off_list);
    uint8_t *tile = wbase + p.off_tile;  // flat: every group's slab starts on a 16-byte boundary
    const uint32_t C = p.C, N = p.num_worlds, F = p.F;
    const uint32_t ncells = nw * C, nplayers = nw * P;
    const uint32_t wl = lane >> 1, q = lane & 1u;
    const bool active = lane < nplayers;
    const uint32_t world = min(w0 + wl, N - 1u);

    uint32_t posori = 0, held = kItemNone;
    int32_t t = 0, dishes_out = 0;
    {
        const uint32_t *g_obj = p.cell_obj + (size_t)w0 * C;
        const uint2 pl_reg = p.players[(size_t)min(w0, N - 1u) * P + (active ? lane : 0u)];  // idle waves (w0 >= N: the grid is rounded up) stay in bounds
        const int2 clock = p.clock[world];
        const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.terr_off), 0, (int)(p.terr_entries * 2u), 0x00020000);
        uint32_t goff[kGroupTerrPerLane];
#pragma unroll
        for (int k = 0; k < kGroupTerrPerLane; k++)
            goff[k] = (uint32_t)k * kWave < p.terr_entries ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(tab, (int)((lane + (uint32_t)k * kWave) * 2u), 0, 0) : 0u;
        for (uint32_t i = lane; i < (p.wpw * C + 3u) >> 2; i += kWave) reinterpret_cast<uint32_t *>(s_cur)[i] = 0xFFFFFFFFu;
        tile_zero_addtid(tile, nw * p.block_bytes);
        for (uint32_t i = lane; i < ncells; i += kWave) s_obj[i] = g_obj[i];
        const uint32_t limit = nw * p.block_bytes;
#pragma unroll
        for (int k = 0; k < kGroupTerrPerLane; k++)
            if ((uint32_t)k * kWave < p.terr_entries && goff[k] != 0u && goff[k] < limit) tile[goff[k]] = 1;
        if (active) {
            posori = pl_reg.x & 0xFFFFu;
            held = pl_reg.y;
        }
        t = clock.x;
        dishes_out = clock.y;
    }
    if (tid < kConstBytes / 4) reinterpret_cast<uint32_t *>(smem)[tid] = const_word;
    __syncthreads();
    if (nw == 0) return;
    if (active) s_cur[wl * C + (posori & 0xFFu)] = (uint8_t)q;
    wave_lds_sync();

    const uint32_t plane = __umul24(C, F), shift = 5 * P;
    uint8_t *gobs = p.obs + (size_t)w0 * p.block_bytes;
    uint32_t slot = p.ring_first;  // the output may be a ring of slots: step k writes slot (ring_first + k) % ring_slots
    int32_t ahead = 0;  // mrl_step_sequence: the next step's action is requested before this step's encode
    if (action_seq) ahead = action_seq[(size_t)(active ? q : 0u) * N + world];
    for (uint32_t k = 0; k < num_steps; k++) {
        if (p.ring_slots > 1u) {
            gobs = p.obs + (size_t)slot * p.ring_stride + (size_t)w0 * p.block_bytes;
            slot = slot + 1u == p.ring_slots ? 0u : slot + 1u;
        }
        uint32_t a;
        if (action_seq) {
            a = (uint32_t)ahead;
            if (k + 1 < num_steps) ahead = action_seq[((size_t)(k + 1) * P + (active ? q : 0u)) * N + world];
        } else {
            a = mrl::scale(mrl::policy_hash(seed, first_step + k, world, q), 6u);
        }
        if (!action_seq && active && k + 1 == num_steps) p.action_out[(size_t)q * N + world] = (int32_t)a;  // the ACTION tensor shows the last draw
        a = (active && a <= A_INTERACT) ? a : (uint32_t)A_STAY;
        const uint32_t old_cell = wl * C + (posori & 0xFFu);
        int32_t reward_world = 0;
        transition<2>(p, s_terrain, s_pots, s_obj, active, wl, q, a, posori, held, dishes_out, reward_world);
        tick_pots(p, s_pots, s_obj, nw, lane);
        t += 1;
        const bool reset_now = (int64_t)t >= p.horizon;
        if (__builtin_expect(__ballot(active && reset_now) != 0ull, 0)) {
            if (reset_now) {
                t = 0;
                dishes_out = 0;
                posori = ((p.starts >> (8u * q)) & 0xFFu) | (A_NORTH << 8);
                held = kItemNone;
            }
            if (active && q == 0) s_list[wl] = reset_now ? 1 : 0;
            wave_lds_sync();
            for (uint32_t i = lane; i < ncells; i += kWave)
                if (s_list[__umulhi(i, p.inv_c)] != 0) s_obj[i] = kItemNone;
            wave_lds_sync();
        }
        // every lane clears its old cell before any lane marks its new one (two DS instructions, in order)
        if (active) s_cur[old_cell] = 0xFF;
        wave_lds_sync();
        if (active) {
            reinterpret_cast<uint2 *>(s_pl)[lane] = make_uint2(posori, held);
            s_cur[wl * C + (posori & 0xFFu)] = (uint8_t)q;
        }
        wave_lds_sync();
        uint32_t ndyn = 0;
        for (uint32_t i0 = 0; i0 < ncells; i0 += kWave) {
            const uint32_t i = i0 + lane;
            const bool valid = i < ncells;
            const uint32_t o = s_obj[valid ? i : 0u];
            const uint32_t who = s_cur[valid ? i : 0u];
            const bool dyn = valid && (((o & 0xFFu) != O_NONE) | (who != 0xFFu));
            const unsigned long long m = __ballot(dyn);
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (dyn) s_list[ndyn + before] = (uint16_t)i;
            ndyn += (uint32_t)__popcll(m);
        }
        wave_lds_sync();
        const uint32_t nent = ndyn * P;
        auto rows = [&](bool restore) {
            for (uint32_t j = lane; j < nent; j += kWave) {
                const uint32_t kk = j >> 1, v = j & 1u;
                const uint32_t i = s_list[kk];
                const uint32_t l = __umulhi(i, p.inv_c), c = i - __umul24(l, C);
                const uint32_t who = s_cur[i];
                const bool occupied = who != 0xFF;
                const uint32_t pidx = (__umul24(l, P) + (occupied ? who : 0u)) * 2;
                const uint32_t w_ori = (s_pl[pidx] >> 8) & 0xFF;
                const uint32_t h = occupied ? s_pl[pidx + 1] : kItemNone;
                const Tail10 t10 = restore ? cell_tail(s_terrain[c], kItemNone, kItemNone) : cell_tail(s_terrain[c], s_obj[i], h);
                uint8_t *row = tile + __umul24(l, p.block_bytes) + __umul24(v, plane) + __umul24(c, F);
                lds_store_tail10_p2(row + shift, t10);
                if (occupied) {
                    const uint32_t rel = who == v ? 0u : (who < v ? who + 1u : who);
                    row[rel] = restore ? 0 : 1;
                    row[P + 4 * rel + w_ori] = restore ? 0 : 1;
                }
            }
        };
        rows(false);
        wave_lds_sync();
        {
            const uint32_t nbytes = nw * p.block_bytes;
            const uint32_t body = nbytes >> 4;
            const uint4 *src = reinterpret_cast<const uint4 *>(tile);
            const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(gobs, 0, (int)(body << 4), 0x00020000);
            for (uint32_t k0 = lane; k0 < body + lane; k0 += 4 * kWave) {
                const uint32_t ka = k0, kb = k0 + kWave, kc = k0 + 2 * kWave, kd = k0 + 3 * kWave;
                const uint32_t last = body - 1u;
                const uint4 va = src[min(ka, last)], vb = src[min(kb, last)], vc = src[min(kc, last)], vd = src[min(kd, last)];
                stream_store_rsrc<kPlain>(out, ka << 4, va);
                stream_store_rsrc<kPlain>(out, kb << 4, vb);
                stream_store_rsrc<kPlain>(out, kc << 4, vc);
                stream_store_rsrc<kPlain>(out, kd << 4, vd);
            }
            const uint32_t done_bytes = body << 4;
            if (lane < nbytes - done_bytes) gobs[done_bytes + lane] = tile[done_bytes + lane];
        }
        wave_lds_sync();
        rows(true);  // the patched rows go back to their static content
        if (active) {
            p.reward[(size_t)q * N + world] = reward_world;
            if (q == 0) p.done[world] = (int32_t)reset_now;
        }
        wave_lds_sync();
    }
    uint32_t *g_obj = p.cell_obj + (size_t)w0 * C;
    for (uint32_t i = lane; i < ncells; i += kWave) g_obj[i] = s_obj[i];
    if (active) {
        p.players[(size_t)w0 * P + lane] = make_uint2(posori, held);
        if (q == 0) p.clock[world] = make_int2(t, dishes_out);
    }
}

template <bool kPlain>
__global__ void __launch_bounds__(kBlock) mrl_simplecooked_rollout(const SimpleParams p, uint32_t num_steps, uint64_t seed, uint32_t first_step,
                                                                   const int32_t *action_seq)
{
    rollout_body<kPlain>(p, num_steps, seed, first_step, action_seq);
}

template <int kC, int kW, int kWidth, int kPots, int kHold, bool kPlain>
__global__ void __launch_bounds__(kBlock) mrl_simplecooked_rollout_fixed(const SimpleParams p, uint32_t num_steps, uint64_t seed,
                                                                         uint32_t first_step, const int32_t *action_seq)
{
    rollout_body<kPlain>(fixed_simple_params<kC, kW, kWidth, kPots, kHold>(p), num_steps, seed, first_step, action_seq);
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

struct SimplecookedSim final : mrl_sim {
    SimpleParams params{};
    uint8_t *own_obs = nullptr;  // the OBS_WORLD_MAJOR buffer; params.obs points elsewhere while the output is redirected
    uint64_t observation_bytes() const override { return (uint64_t)num_worlds * params.block_bytes; }
    uint64_t set_observation_output(void *out) override  // see OvercookedSim::set_observation_output
    {
        set_observation_ring(out, 0, 1);
        return observation_bytes();
    }
    uint8_t *ring_base = nullptr;
    uint64_t ring_stride = 0, ring_pos = 0;
    uint32_t ring_slots = 1;
    void set_observation_ring(void *base, uint64_t stride_bytes, uint32_t slots) override  // see OvercookedSim::set_observation_ring
    {
        ring_base = base ? static_cast<uint8_t *>(base) : own_obs;
        ring_stride = base ? stride_bytes : 0;
        ring_slots = base && slots > 1 ? slots : 1;
        ring_pos = 0;
        // slots off 16-byte boundaries are staged (OvercookedSim::set_observation_ring): a slab of the simulator's + one copy
        staged = base && ((reinterpret_cast<uintptr_t>(base) & 15u) != 0 || (ring_slots > 1 && (ring_stride & 15u) != 0));
        if (staged && !staging) staging = arena.alloc<uint8_t>(observation_bytes(), false);
        params.obs = staged ? staging : ring_base;
        params.ring_stride = staged ? 0 : ring_stride;
        params.ring_slots = staged ? 1 : ring_slots;
        params.ring_first = 0;
    }
    bool staged = false;
    uint8_t *staging = nullptr;
    uint8_t *take_slots(uint32_t steps, uint32_t *first)
    {
        const uint32_t at = (uint32_t)(ring_pos % ring_slots);
        ring_pos += steps;
        if (first) *first = at;
        return ring_base + (size_t)at * ring_stride;
    }
    uint32_t H = 0, grid = 0, lds_bytes = 0;
    using FixedKernel = void (*)(uint32_t *, uint2 *, int2 *, const void *, const uint32_t *, const uint16_t *, uint32_t, uint32_t, const SimpleParams);
    FixedKernel fixed_kernel[3] = {};  // mrl_simplecooked_step_fixed<...> per action source, when the parameters are exactly its
    void (*fixed_rollout)(const SimpleParams, uint32_t, uint64_t, uint32_t, const int32_t *) = nullptr;
    void (*generic_rollout)(const SimpleParams, uint32_t, uint64_t, uint32_t, const int32_t *) = nullptr;
    void (*generic_step)(const SimpleParams) = nullptr;  // mrl_simplecooked_step<false, P, store flavour>
    const char *fixed_name = nullptr;
    int32_t *action = nullptr, *active = nullptr, *mask = nullptr;
    int32_t *world_id = nullptr, *agent_id = nullptr, *loc_world_id = nullptr, *loc_id = nullptr;

    void launch(bool init, const SimpleParams &given, hipStream_t stream)
    {
        SimpleParams a = given;
        uint8_t *const dest = init ? nullptr : take_slots(1, nullptr);
        if (!init) a.obs = staged ? staging : dest;
        a.ring_slots = 1;  // a single step writes exactly its `obs`
        if (init) {
            if (a.P == 2)
                hipLaunchKernelGGL((mrl_simplecooked_step<true, 2>), dim3(grid), dim3(kBlock), lds_bytes, stream, a);
            else
                hipLaunchKernelGGL((mrl_simplecooked_step<true, 1>), dim3(grid), dim3(kBlock), lds_bytes, stream, a);
        } else if (fixed_kernel[0]) {
            const int source = a.sample ? 2 : (a.actions64 ? 1 : 0);
            const void *hot_actions = source == 1 ? static_cast<const void *>(a.actions64) : static_cast<const void *>(a.actions);
            hipLaunchKernelGGL(fixed_kernel[source], dim3(grid), dim3(kBlock), lds_bytes, stream, a.cell_obj, a.players, a.clock, hot_actions,
                               a.consts, a.terr_off, a.num_worlds, a.per_xcd, a);
        } else {
            hipLaunchKernelGGL(generic_step, dim3(grid), dim3(kBlock), lds_bytes, stream, a);
        }
        MRL_HIP(hipGetLastError());
        if (staged && dest) MRL_HIP(hipMemcpyAsync(dest, staging, observation_bytes(), hipMemcpyDeviceToDevice, stream));
    }

    void phase1(const int32_t *actions, hipStream_t stream) override
    {
        SimpleParams a = params;
        a.actions = actions ? actions : action;
        launch(false, a, stream);
    }
    void phase2(const uint32_t *, hipStream_t) override {}
    bool step_i64(const long long *actions, hipStream_t stream) override
    {
        SimpleParams a = params;
        a.actions = action;
        a.actions64 = actions;
        a.action_out = action;
        launch(false, a, stream);
        return true;
    }

    // all steps of a call in one launch (two players, flat tile; see rollout_body)
    bool launch_rollout(uint32_t num_steps, uint64_t seed, uint32_t first_step, const int32_t *action_seq, hipStream_t stream)
    {
        if (params.P != 2 || !params.flat || staged) return false;  // (staged slots: one launch + one copy per step)
        SimpleParams a = params;
        a.action_out = action;
        take_slots(num_steps, &a.ring_first);
        if (fixed_rollout)
            hipLaunchKernelGGL(fixed_rollout, dim3(grid), dim3(kBlock), lds_bytes, stream, a, num_steps, seed, first_step, action_seq);
        else
            hipLaunchKernelGGL(generic_rollout, dim3(grid), dim3(kBlock), lds_bytes, stream, a, num_steps, seed, first_step, action_seq);
        MRL_HIP(hipGetLastError());
        return true;
    }

    void step_sequence(const int32_t *actions, uint32_t num_steps, hipStream_t stream) override
    {
        if (num_steps == 0) return;
        if (!launch_rollout(num_steps, 0, 0, actions, stream)) mrl_sim::step_sequence(actions, num_steps, stream);
    }

    // uniform random policy on the device: one launch for all steps where the layout allows, else the draw happens in the
    // step kernel, one launch per step
    void rollout_random(uint32_t num_steps, uint64_t seed, uint32_t first_step, hipStream_t stream) override
    {
        if (num_steps == 0) return;
        if (launch_rollout(num_steps, seed, first_step, nullptr, stream)) return;
        SimpleParams a = params;
        a.actions = action;
        a.sample = 1;
        a.sample_seed = seed;
        a.action_out = action;
        for (uint32_t k = 0; k < num_steps; k++) {
            a.sample_step = first_step + k;
            launch(false, a, stream);
        }
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
        case MRL_OVERCOOKED_STATE_TIMESTEP: *out = mrl::make_desc(params.clock, MRL_INT32, device, {N}, {2}); return true;
        case MRL_SIMPLECOOKED_STATE_DISHES_OUT:
            *out = mrl::make_desc(reinterpret_cast<int32_t *>(params.clock) + 1, MRL_INT32, device, {N}, {2});
            return true;
        default: return false;
        }
    }

    size_t action_elems() const override { return (size_t)params.P * num_worlds; }
    void launch_shape(uint32_t out[4]) const override
    {
        out[0] = grid;
        out[1] = kBlock;
        out[2] = lds_bytes;
        out[3] = params.wpw;
    }
    const char *kernel_name() const override
    {
        if (fixed_kernel[0]) return fixed_name;
        return params.P == 2 ? "mrl_simplecooked_step<false, 2>" : "mrl_simplecooked_step<false, 1>";
    }

    uint64_t bytes_per_world_step() const override
    {
        // like SURVEY.md section 8d for Overcooked: actions 4P + player state r/w 2*8P + cell objects r/w 2*4C +
        // clock {timestep, num_dishes_out} r/w 16 + obs out P*C*F + reward 4P + done 4
        const uint64_t P = params.P, C = params.C;
        return 4 * P + 16 * P + 8 * C + 16 + params.block_bytes + 4 * P + 4;
    }
};

}  // namespace

mrl_sim *mrl::create_simplecooked(const mrl_overcooked_config *cfg, int gpu_id, uint32_t num_worlds)
{
    if (!cfg || !cfg->terrain || !cfg->start_player_x || !cfg->start_player_y || !cfg->recipe_values || !cfg->recipe_times) {
        set_error("simplecooked: null config field");
        throw HipError{MRL_ERR_INVALID};
    }
    const int64_t H = cfg->height, W = cfg->width, P = cfg->num_players;
    if (H < 3 || W < 3 || H * W > (int64_t)kMaxCells) {
        set_error("simplecooked: height*width must be 9..100 (MAX_SIZE, src/overcooked2_env/sim.hpp:12), got %lldx%lld", (long long)H,
                  (long long)W);
        throw HipError{MRL_ERR_INVALID};
    }
    if (P < 1 || P > (int64_t)kMaxPlayers) {
        set_error("simplecooked: num_players must be 1..2 (MAX_NUM_PLAYERS, src/overcooked2_env/sim.hpp:13), got %lld", (long long)P);
        throw HipError{MRL_ERR_INVALID};
    }
    if (num_worlds == 0) {
        set_error("simplecooked: num_worlds must be > 0");
        throw HipError{MRL_ERR_INVALID};
    }
    const int64_t C = H * W;
    alignas(4) uint8_t consts[kConstBytes];
    memset(consts, 0, sizeof(consts));
    uint32_t num_pots = 0;
    for (int64_t c = 0; c < C; c++) {
        const int64_t t = cfg->terrain[c];
        if (t < 0 || t > 6) {
            set_error("simplecooked: terrain[%lld] = %lld is not a TerrainT value", (long long)c, (long long)t);
            throw HipError{MRL_ERR_INVALID};
        }
        const int64_t x = c % W, y = c / W;
        if (t == T_AIR && (x == 0 || y == 0 || x == W - 1 || y == H - 1)) {
            set_error("simplecooked: walkable cell on the grid border at (%lld,%lld); the step indexes neighbours without bounds checks "
                      "(src/overcooked2_env/sim.cpp:160-172)",
                      (long long)x, (long long)y);
            throw HipError{MRL_ERR_INVALID};
        }
        consts[kConstTerrain + c] = (uint8_t)t;
        if (t == T_POT) consts[kConstPots + num_pots++] = (uint8_t)c;
    }
    uint32_t starts = 0;
    for (int64_t q = 0; q < P; q++) {
        const int64_t x = cfg->start_player_x[q], y = cfg->start_player_y[q];
        if (x < 1 || y < 1 || x >= W - 1 || y >= H - 1) {
            set_error("simplecooked: start position of player %lld (%lld,%lld) is not an interior cell", (long long)q, (long long)x,
                      (long long)y);
            throw HipError{MRL_ERR_INVALID};
        }
        starts |= (uint32_t)(y * W + x) << (8 * q);
    }

    bind_device(gpu_id);
    auto *sim = new SimplecookedSim();
    try {
        sim->game = MRL_GAME_SIMPLECOOKED;
        sim->device = gpu_id;
        sim->num_worlds = num_worlds;
        sim->H = (uint32_t)H;
        SimpleParams &a = sim->params;
        const uint32_t N = num_worlds;
        a.num_worlds = N;
        a.P = (uint32_t)P;
        a.C = (uint32_t)C;
        a.W = (uint32_t)W;
        a.F = 5 * a.P + 10;
        a.rows = a.P * a.C;
        a.block_bytes = a.rows * a.F;
        a.inv_c = (uint32_t)((1ull << 32) / (uint64_t)C) + 1u;
        a.inv_rows = (uint32_t)((1ull << 32) / (uint64_t)a.rows) + 1u;
        a.placement_rew = (uint8_t)cfg->placement_in_pot_rew;  // uint8 like WorldState (sim.hpp:96-98)
        a.dish_rew = (uint8_t)cfg->dish_pickup_rew;
        a.soup_pickup_rew = (uint8_t)cfg->soup_pickup_rew;
        a.horizon = cfg->horizon;
        a.num_pots = num_pots;
        a.starts = starts;
        a.deltas = pack_deltas(W);
        for (int r = 0; r < 16; r++) {
            reinterpret_cast<uint8_t *>(a.times_w)[r] = (uint8_t)cfg->recipe_times[r];
            reinterpret_cast<uint8_t *>(a.values_w)[r] = (uint8_t)cfg->recipe_values[r];
        }
        // worlds per wave: as many as keep >= 4096 waves in the launch, one lane per (world, player), and a
        // tile of at most 9400 bytes (the limit measured in overcooked.hip); powers of two
        uint32_t wpw = 64 / a.P;
        while (wpw > 1 && (N + wpw - 1) / wpw < 4096) wpw >>= 1;
        while (wpw > 1 && wpw * a.block_bytes + 48u > 9400u) wpw >>= 1;
        if (const int64_t forced = mrl::debug_get("overcooked.wpw", 0)) wpw = std::min<uint32_t>((uint32_t)std::max<int64_t>(forced, 1), 64 / a.P);
        auto up16 = [](uint32_t v) { return (v + 15u) & ~15u; };
        a.wpw = wpw;
        a.off_pl = up16(wpw * a.C * 4);
        a.off_cur = a.off_pl + up16(wpw * a.P * 8);
        a.off_list = a.off_cur + up16(wpw * a.C);
        a.off_tile = a.off_list + up16(std::max(wpw * a.C * 2u, 128u));
        a.flat = ((wpw * a.block_bytes) % 16u == 0 && wpw * a.rows <= (uint32_t)kGroupTerrPerLane * kWave) ? 1u : 0u;
        a.terr_entries = wpw * a.rows;
        a.lds_wave_stride = a.off_tile + (a.flat ? ((wpw * a.block_bytes + 255u) & ~255u) : up16(wpw * a.block_bytes)) + 48u;
        sim->lds_bytes = kConstBytes + kWavesPerBlock * a.lds_wave_stride;
        // direct encode (see SimpleParams): holder cells = counters / pots next to a walkable cell; players start on walkable cells
        std::vector<uint32_t> holders;  // cell | is_pot << 8
        for (int64_t c = 0; c < C; c++) {
            const uint32_t t = consts[kConstTerrain + c];
            if (t != T_COUNTER && t != T_POT) continue;
            const int64_t x = c % W, y = c / W;
            const bool faced = (x > 0 && consts[kConstTerrain + c - 1] == T_AIR) || (x + 1 < W && consts[kConstTerrain + c + 1] == T_AIR) ||
                               (y > 0 && consts[kConstTerrain + c - W] == T_AIR) || (y + 1 < H && consts[kConstTerrain + c + W] == T_AIR);
            if (faced) holders.push_back((uint32_t)c | (t == T_POT ? 0x100u : 0u));
        }
        std::vector<uint32_t> hold_tab;
        {
            const uint32_t free0 = (uint32_t)kWave - std::min<uint32_t>((uint32_t)kWave, wpw * a.P);
            uint32_t k = 0;
            for (uint32_t l = 0; l < wpw; l++)
                for (const uint32_t h : holders) {
                    const uint32_t c = h & 0xFFu;
                    const uint32_t slot = k < free0 ? wpw * a.P + k : (uint32_t)kWave + (k - free0);
                    if (hold_tab.size() <= slot) hold_tab.resize(slot + 1, 0u);
                    hold_tab[slot] = (l * a.block_bytes + c * a.F) | ((l * a.C + c) << 16) | (1u << 30) | ((h >> 8) << 31);
                    k++;
                }
            if (hold_tab.empty()) hold_tab.resize(1, 0u);
            bool starts_walkable = true;
            for (int64_t q = 0; q < P; q++) starts_walkable = starts_walkable && consts[kConstTerrain + ((starts >> (8 * q)) & 0xFFu)] == T_AIR;
            a.direct = (a.P == 2 && a.flat && starts_walkable && hold_tab.size() <= (size_t)kHoldPerLane * kWave && wpw * a.block_bytes < 65536u &&
                        !mrl::debug_get("overcooked.no_direct", 0))
                           ? 1u
                           : 0u;
            a.hold_entries = (uint32_t)hold_tab.size();
        }
        // store flavour of the stream-out (grid_common.hpp, stream_store_rsrc): ordinary stores iff a group's slab is not whole
        // 128-byte lines and either the launch is a multi-step one or the slab exceeds the 256 MiB Infinity Cache
        const auto plain_for = [&](bool multi_step) {
            const int64_t knob = mrl::debug_get("overcooked.whole_store", 0);  // 0 that rule, 1 write-through, 2 plain
            const bool whole_lines = ((uint64_t)wpw * a.block_bytes) % 64u == 0;  // (64: Simplecooked random0's 8000-byte groups, half a 128-byte line off, do not care)
            return knob ? knob == 2 : (!whole_lines && (multi_step || (uint64_t)N * a.block_bytes > (256ull << 20)));
        };
        const bool plain = plain_for(false), plain_multi = plain_for(true);
        sim->generic_step = a.P == 2 ? (plain ? &mrl_simplecooked_step<false, 2, true> : &mrl_simplecooked_step<false, 2, false>)
                                     : (plain ? &mrl_simplecooked_step<false, 1, true> : &mrl_simplecooked_step<false, 1, false>);
        sim->generic_rollout = plain_multi ? &mrl_simplecooked_rollout<true> : &mrl_simplecooked_rollout<false>;
#define MRL_FIXED(C_, WPW_, WIDTH_, POTS_, HOLD_)                                                                                   \
    if (!sim->fixed_kernel[0] && a.P == 2 && a.C == C_ && a.W == WIDTH_ && a.num_pots == POTS_ && a.wpw == WPW_ && a.flat && a.direct && \
        holders.size() == HOLD_ && !mrl::debug_get("overcooked.no_fixed", 0)) {                                                      \
        sim->fixed_kernel[0] = plain ? &mrl_simplecooked_step_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, 0, true>                         \
                                     : &mrl_simplecooked_step_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, 0, false>;                       \
        sim->fixed_kernel[1] = plain ? &mrl_simplecooked_step_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, 1, true>                         \
                                     : &mrl_simplecooked_step_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, 1, false>;                       \
        sim->fixed_kernel[2] = plain ? &mrl_simplecooked_step_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, 2, true>                         \
                                     : &mrl_simplecooked_step_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, 2, false>;                       \
        sim->fixed_rollout = plain_multi ? &mrl_simplecooked_rollout_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, true>                     \
                                         : &mrl_simplecooked_rollout_fixed<C_, WPW_, WIDTH_, POTS_, HOLD_, false>;                   \
        sim->fixed_name = "mrl_simplecooked_step_fixed<" #C_ ", " #WPW_ ", " #WIDTH_ ", " #POTS_ ", " #HOLD_ ", 0>";                 \
    }
        // the five standard old-style layouts at the batch sizes where they get 8 worlds per wave: cells, worlds per wave,
        // grid width, pots, holder cells
        MRL_FIXED(20, 8, 5, 1, 6)    // simple
        MRL_FIXED(45, 4, 9, 2, 14)   // unident_s
        MRL_FIXED(25, 8, 5, 2, 9)    // random0, random1
        MRL_FIXED(40, 4, 8, 2, 18)   // random3
#undef MRL_FIXED
        if (sim->lds_bytes > 65536) {
            set_error("simplecooked: internal: %u bytes of LDS per workgroup", sim->lds_bytes);
            throw HipError{MRL_ERR_INVALID};
        }
        const uint32_t waves = (N + wpw - 1) / wpw;
        const uint32_t blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
        sim->grid = (blocks + 7u) & ~7u;
        a.per_xcd = sim->grid >> 3;
        {
            uint32_t *d_tab = sim->arena.alloc<uint32_t>(hold_tab.size(), false);
            MRL_HIP(hipMemcpy(d_tab, hold_tab.data(), hold_tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            a.hold_tab = d_tab;
        }

        uint32_t *d_consts = sim->arena.alloc<uint32_t>(kConstBytes / 4, false);
        MRL_HIP(hipMemcpy(d_consts, consts, kConstBytes, hipMemcpyHostToDevice));
        a.consts = d_consts;
        {
            // per row of one world: where its terrain one-hot byte goes (channel 5P + t - 1, sim.cpp:553-558); the
            // tomato source's falls on channel 5P+5, which observationSystem zeroes on every pass (sim.cpp:74)
            std::vector<uint16_t> pos(a.rows, 0xFFFFu);
            for (uint32_t v = 0; v < a.P; v++)
                for (uint32_t c = 0; c < a.C; c++) {
                    const uint32_t t = consts[kConstTerrain + c];
                    if (t != T_AIR && t != T_TOMATO_SRC) pos[v * a.C + c] = (uint16_t)((v * a.C + c) * a.F + 5 * a.P + t - 1);
                }
            uint16_t *d_pos = sim->arena.alloc<uint16_t>(a.rows, false);
            MRL_HIP(hipMemcpy(d_pos, pos.data(), a.rows * sizeof(uint16_t), hipMemcpyHostToDevice));
            a.terr_pos = d_pos;
            std::vector<uint16_t> off(a.terr_entries, 0);
            for (uint32_t l = 0; l < a.wpw; l++)
                for (uint32_t r = 0; r < a.rows; r++)
                    if (pos[r] != 0xFFFFu) off[l * a.rows + r] = (uint16_t)(l * a.block_bytes + pos[r]);
            uint16_t *d_off = sim->arena.alloc<uint16_t>(a.terr_entries, false);
            MRL_HIP(hipMemcpy(d_off, off.data(), a.terr_entries * sizeof(uint16_t), hipMemcpyHostToDevice));
            a.terr_off = d_off;
        }
        a.cell_obj = sim->arena.alloc<uint32_t>((size_t)N * C);
        a.players = sim->arena.alloc<uint2>((size_t)N * P);
        a.clock = sim->arena.alloc<int2>(N);
        a.reward = sim->arena.alloc<int32_t>((size_t)N * P);
        a.done = sim->arena.alloc<int32_t>(N);
        a.obs = sim->arena.alloc<uint8_t>((size_t)N * a.block_bytes, false);
        sim->own_obs = a.obs;
        sim->ring_base = a.obs;
        a.ring_stride = 0;
        a.ring_slots = 1;
        a.ring_first = 0;
        sim->action = sim->arena.alloc<int32_t>((size_t)N * P);
        sim->active = sim->arena.alloc<int32_t>((size_t)N * P, false);
        sim->mask = sim->arena.alloc<int32_t>((size_t)N * P * 6, false);
        a.actions = sim->action;
        a.action_out = sim->action;
        const size_t na = (size_t)N * P, nm = na * 6;
        hipLaunchKernelGGL(fill_i32, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, 0, sim->active, 1, na);
        hipLaunchKernelGGL(fill_i32, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, 0, sim->mask, 1, nm);
        MRL_HIP(hipGetLastError());
        // Sim::Sim (sim.cpp:470-575): reset state + first observation
        sim->launch(true, a, 0);
        MRL_HIP(hipDeviceSynchronize());
    } catch (...) {
        delete sim;
        throw;
    }
    return sim;
}
