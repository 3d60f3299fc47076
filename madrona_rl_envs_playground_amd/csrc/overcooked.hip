// Overcooked world step for gfx950: one wavefront per world, whole step fused
// in one kernel (action apply -> interactions -> movement/collisions -> pot
// ticks -> horizon reset -> reward/done -> observation encode).
//
// Semantics follow the reference task graph
//   /root/reference/src/overcooked_env/sim.cpp:498-537
// (systems :199-495, observation rows :68-167, init :556-659) with the component
// widths of sim.hpp:59-184.  Nothing of Madrona's ECS/taskgraph is reproduced:
// the 20 graph nodes and their per-cell scratch (past/current/future_player,
// interacting_players) collapse into
//   - a wave-uniform loop over the interacting players in ascending id, which is
//     what the reference's four rank phases serialise to (sim.cpp:259-358),
//   - a pairwise proposal test for the all-or-nothing collision rule
//     (sim.cpp:363-426): same target or swapped cells => nobody moves,
//   - a from-scratch observation encode (the reference updates rows in place
//     and clears the player channels through past_player; every vacated cell is
//     cleared, so the row is a pure function of the state).
//
// HBM layout, all world-major so a wave's loads and
// stores are contiguous and a shard of worlds is one contiguous slab:
//   cell_obj [N][C]  u32  name | onions<<8 | tomatoes<<16 | tick<<24
//   players  [N][P]  2xu32 {pos | orientation<<8, held item (same packing)}
//   timestep [N]     i32
//   action   [P][N]  i32  (the reference's exported shape, mgr.cpp:214-218)
//   reward   [P][N]  i32, done [N] i32
//   obs      [N][P][C][F] u8, F = 5P+16: one contiguous P*C*F block per world
// Per-wave LDS: the cell objects, a cell->player map, the players' orientation
// and held item, and a tile of up to 64 observation rows that is zero-filled,
// patched with the few non-zero bytes, and streamed out with 16-byte stores.
#include "common.hpp"

#include <cstring>

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;

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
constexpr uint32_t kConstBytes = 352;

struct StepParams {
    uint32_t num_worlds;
    uint32_t P, C, W, F;
    uint32_t rows;         // P*C
    uint32_t block_bytes;  // P*C*F
    uint32_t inv_c;        // floor(2^32/C)+1: r / C == umulhi(r, inv_c) for r < 2^16
    uint32_t placement_rew, soup_pickup_rew;
    uint32_t c_pad;        // C rounded up to 16
    uint32_t p_pad;        // P rounded up to 2
    uint32_t lds_wave_stride;
    int64_t horizon;
    const uint32_t *consts;  // kConstBytes, device
    uint32_t *cell_obj;
    uint2 *players;
    int32_t *timestep;
    const int32_t *actions;
    int32_t *reward;
    int32_t *done;
    uint8_t *obs;
};

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS instructions of one wave execute in issue order; this only stops the
    // compiler from moving LDS accesses across the point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ int32_t step_of(uint32_t dir, uint32_t width)
{
    // sim.cpp:185-197
    return dir == A_NORTH ? -(int32_t)width : dir == A_SOUTH ? (int32_t)width : dir == A_EAST ? 1 : dir == A_WEST ? -1 : 0;
}

__device__ __forceinline__ uint32_t recipe_of(uint32_t item)
{
    return ((kMaxIngredients + 1) * ((item >> 8) & 0xFF) + ((item >> 16) & 0xFF)) & 15u;
}

template <bool kInit>
__global__ void __launch_bounds__(kBlock) mrl_overcooked_step(const StepParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t wib = tid >> 6;

    if (tid < kConstBytes / 4) reinterpret_cast<uint32_t *>(smem)[tid] = p.consts[tid];
    __syncthreads();
    const uint8_t *s_terrain = smem + kConstTerrain;
    const uint8_t *s_times = smem + kConstTimes;
    const uint8_t *s_values = smem + kConstValues;
    const uint8_t *s_start = smem + kConstStart;

    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs, so
    // give each XCD one contiguous range of worlds (neighbouring worlds share
    // cache lines of every state array; keep them in one L2).
    const uint32_t per_xcd = gridDim.x >> 3;
    const uint32_t logical_block = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const uint32_t world = logical_block * kWavesPerBlock + wib;
    if (world >= p.num_worlds) return;

    uint8_t *wbase = smem + kConstBytes + wib * p.lds_wave_stride;
    uint32_t *s_obj = reinterpret_cast<uint32_t *>(wbase);
    uint8_t *s_cur = wbase + p.c_pad * 4;
    uint32_t *s_plr = reinterpret_cast<uint32_t *>(s_cur + p.c_pad);
    uint8_t *s_tile = reinterpret_cast<uint8_t *>(s_plr + 2 * p.p_pad);

    const uint32_t P = p.P, C = p.C, N = p.num_worlds;
    const bool is_player = lane < P;

    // ---------------- load ----------------
    uint32_t posori = 0, held = kItemNone, act = A_STAY;
    int32_t t = 0;
    if (!kInit) {
        for (uint32_t c = lane; c < C; c += kWave) s_obj[c] = p.cell_obj[(size_t)world * C + c];
        if (is_player) {
            const uint2 rec = p.players[(size_t)world * P + lane];
            posori = rec.x;
            held = rec.y;
            const uint32_t a = (uint32_t)p.actions[(size_t)lane * N + world];
            act = a <= A_INTERACT ? a : (uint32_t)A_STAY;  // values outside the enum are outside the contract
        }
        t = p.timestep[world];
    }
    uint32_t pos = posori & 0xFF, ori = (posori >> 8) & 0xFF;
    int32_t reward = 0;
    bool reset_now = kInit;

    if (!kInit) {
        wave_lds_sync();
        // ---------------- interactions (sim.cpp:208-358) ----------------
        const uint32_t facing = pos + (uint32_t)step_of(ori, p.W);
        const uint32_t facing_terrain = is_player ? (uint32_t)s_terrain[facing] : (uint32_t)T_AIR;
        unsigned long long todo = __ballot(is_player && act == A_INTERACT);
        while (todo) {
            const int q = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t tgt = (uint32_t)__builtin_amdgcn_readlane((int)facing, q);
            const uint32_t terr = (uint32_t)__builtin_amdgcn_readlane((int)facing_terrain, q);
            const uint32_t q_held = (uint32_t)__builtin_amdgcn_readlane((int)held, q);
            const uint32_t hname = q_held & 0xFF;
            uint32_t new_held = q_held;
            if (terr == T_COUNTER || terr == T_POT) {
                const uint32_t obj = rfl(s_obj[tgt]);
                const uint32_t oname = obj & 0xFF;
                uint32_t new_obj = obj;
                if (terr == T_COUNTER) {
                    if (hname != O_NONE && oname == O_NONE) {
                        new_obj = q_held;
                        new_held = kItemNone;
                    } else if (hname == O_NONE && oname != O_NONE) {
                        new_held = obj;
                        new_obj = kItemNone;
                    }
                } else {
                    const int32_t tick = (int8_t)(obj >> 24);
                    const uint32_t count = (((obj >> 8) & 0xFF) + ((obj >> 16) & 0xFF)) & 0xFF;
                    if (hname == O_NONE) {
                        // idle soup with something in it starts cooking
                        if (oname == O_SOUP && tick < 0 && count > 0) new_obj = obj & 0x00FFFFFFu;
                    } else if (hname == O_DISH && oname == O_SOUP && tick >= 0 &&
                               tick >= (int32_t)rfl(s_times[recipe_of(obj)])) {
                        new_held = obj;
                        new_obj = kItemNone;
                        reward += (int32_t)p.soup_pickup_rew;
                    } else if (hname == O_ONION || hname == O_TOMATO) {
                        uint32_t soup = oname == O_NONE ? (O_SOUP | kItemNone) : obj;
                        const int32_t stick = (int8_t)(soup >> 24);
                        const uint32_t scount = (((soup >> 8) & 0xFF) + ((soup >> 16) & 0xFF)) & 0xFF;
                        if (!(stick >= 0 || scount == kMaxIngredients)) {
                            soup += hname == O_ONION ? 0x100u : 0x10000u;
                            new_held = kItemNone;
                            reward += (int32_t)p.placement_rew;
                        }
                        new_obj = soup;
                    }
                }
                if (new_obj != obj) s_obj[tgt] = new_obj;
                wave_lds_sync();
            } else if (terr == T_ONION_SRC) {
                if (hname == O_NONE) new_held = O_ONION | kItemNone;
            } else if (terr == T_TOMATO_SRC) {
                if (hname == O_NONE) new_held = O_TOMATO | kItemNone;
            } else if (terr == T_DISH_SRC) {
                if (hname == O_NONE) new_held = O_DISH | kItemNone;
            } else if (terr == T_SERVING) {
                if (hname == O_SOUP) {
                    reward += (int32_t)rfl(s_values[recipe_of(q_held)]);
                    new_held = kItemNone;
                }
            }
            if ((int)lane == q) held = new_held;
        }

        // ---------------- movement (sim.cpp:363-426) ----------------
        uint32_t prop = pos, pori = ori;
        if (is_player && act != A_INTERACT) {
            const uint32_t np = pos + (uint32_t)step_of(act, p.W);
            pori = act == A_STAY ? ori : act;
            prop = s_terrain[np] != T_AIR ? pos : np;
        }
        bool conflict = false;
        for (uint32_t q = 0; q < P; q++) {
            const uint32_t pq = (uint32_t)__builtin_amdgcn_readlane((int)prop, (int)q);
            const uint32_t oq = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)q);
            conflict |= (q != lane) & ((prop == pq) | ((prop == oq) & (pos == pq)));
        }
        const bool blocked = __ballot(is_player && conflict) != 0ull;
        if (!blocked) pos = prop;
        ori = pori;

        // ---------------- horizon (sim.cpp:485-489) ----------------
        t += 1;
        reset_now = (int64_t)t >= p.horizon;
    }

    // ---------------- reset (sim.cpp:441-482) ----------------
    if (reset_now) {
        t = 0;
        if (is_player) {
            pos = s_start[lane];
            ori = A_NORTH;
            held = kItemNone;
        }
    }

    // ---------------- pots (sim.cpp:430-438), object reset, state write-back ----------------
    for (uint32_t c = lane; c < C; c += kWave) {
        uint32_t o = kInit ? kItemNone : s_obj[c];
        if (!kInit && s_terrain[c] == T_POT && (o & 0xFF) == O_SOUP) {
            const int32_t tick = (int8_t)(o >> 24);
            if (tick >= 0 && tick < (int32_t)s_times[recipe_of(o)]) o = (o & 0x00FFFFFFu) | ((uint32_t)(uint8_t)(tick + 1) << 24);
        }
        if (reset_now) o = kItemNone;
        s_obj[c] = o;
        s_cur[c] = 0xFF;
        p.cell_obj[(size_t)world * C + c] = o;
    }
    if (is_player) {
        p.players[(size_t)world * P + lane] = make_uint2(pos | (ori << 8), held);
        p.reward[(size_t)lane * N + world] = reward;
    }
    if (lane == 0) {
        p.timestep[world] = t;
        p.done[world] = kInit ? 0 : (int32_t)reset_now;
    }
    wave_lds_sync();
    if (is_player) {
        s_cur[pos] = (uint8_t)lane;
        s_plr[2 * lane] = ori;
        s_plr[2 * lane + 1] = held;
    }
    wave_lds_sync();

    // ---------------- observation (sim.cpp:68-167, 642-645) ----------------
    const bool urgent = p.horizon - (int64_t)t < 40;
    const uint32_t F = p.F, shift = 5 * P;
    uint8_t *gobs = p.obs + (size_t)world * p.block_bytes;
    for (uint32_t r0 = 0; r0 < p.rows; r0 += kWave) {
        const uint32_t nrows = min((uint32_t)kWave, p.rows - r0);
        const uint32_t nbytes = nrows * F;
        uint8_t *g = gobs + (size_t)r0 * F;
        // keep LDS and global addresses congruent mod 16 so aligned 16-byte
        // chunks line up on both sides
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(g) & 15u);
        uint8_t *tile = s_tile + mis;

        const uint32_t nchunks = (mis + nbytes + 15u) >> 4;
        for (uint32_t k = lane; k < nchunks; k += kWave) reinterpret_cast<uint4 *>(s_tile)[k] = make_uint4(0, 0, 0, 0);
        wave_lds_sync();

        if (lane < nrows) {
            const uint32_t r = r0 + lane;
            const uint32_t viewer = __umulhi(r, p.inv_c);
            const uint32_t c = r - viewer * C;
            uint8_t *row = tile + lane * F;
            const uint32_t terr = s_terrain[c];
            const uint32_t o = s_obj[c];
            const uint32_t who = s_cur[c];

            if (terr != T_AIR) row[shift + terr - 1] = 1;
            if (urgent) row[shift + 15] = 1;

            // channels shift+6 .. shift+14, computed branch-free then stored
            const uint32_t oname = o & 0xFF, on = (o >> 8) & 0xFF, tom = (o >> 16) & 0xFF;
            const int32_t tick = (int8_t)(o >> 24);
            uint32_t idle_on = 0, idle_tom = 0, soup_on = 0, soup_tom = 0, remaining = 0, ready = 0;
            uint32_t dish = oname == O_DISH, onion = oname == O_ONION, tomato = oname == O_TOMATO;
            if (oname == O_SOUP) {
                if (terr == T_POT) {
                    if (tick < 0) {
                        idle_on = on;
                        idle_tom = tom;
                    } else {
                        const int32_t need = (int32_t)s_times[recipe_of(o)];
                        soup_on = on;
                        soup_tom = tom;
                        remaining = (uint32_t)(need - tick) & 0xFF;
                        ready = tick >= need;
                    }
                } else {
                    soup_on = on;
                    soup_tom = tom;
                    ready = 1;
                }
            }
            if (who != 0xFF) {
                const uint32_t rel = who == viewer ? 0u : (who < viewer ? who + 1u : who);
                const uint32_t w_ori = s_plr[2 * who], w_held = s_plr[2 * who + 1];
                row[rel] = 1;
                row[P + 4 * rel + w_ori] = 1;
                const uint32_t hname = w_held & 0xFF;
                if (hname == O_SOUP) {
                    soup_on = (w_held >> 8) & 0xFF;
                    soup_tom = (w_held >> 16) & 0xFF;
                    remaining = 0;
                    ready = 1;
                } else if (hname == O_DISH) {
                    dish = 1;
                } else if (hname == O_ONION) {
                    onion = 1;
                } else if (hname == O_TOMATO) {
                    tomato = 1;
                }
            }
            uint8_t *tail = row + shift + 6;
            tail[0] = (uint8_t)idle_on;
            tail[1] = (uint8_t)idle_tom;
            tail[2] = (uint8_t)soup_on;
            tail[3] = (uint8_t)soup_tom;
            tail[4] = (uint8_t)remaining;
            tail[5] = (uint8_t)ready;
            tail[6] = (uint8_t)dish;
            tail[7] = (uint8_t)onion;
            tail[8] = (uint8_t)tomato;
        }
        wave_lds_sync();

        // stream the tile out: unaligned head/tail bytes, 16-byte body
        const uint32_t head = min((16u - mis) & 15u, nbytes);
        if (lane < head) g[lane] = tile[lane];
        const uint32_t body = (nbytes - head) >> 4;
        const uint4 *src = reinterpret_cast<const uint4 *>(tile + head);
        uint4 *dst = reinterpret_cast<uint4 *>(g + head);
        for (uint32_t k = lane; k < body; k += kWave) dst[k] = src[k];
        const uint32_t done_bytes = head + (body << 4);
        if (lane < nbytes - done_bytes) g[done_bytes + lane] = tile[done_bytes + lane];
        wave_lds_sync();
    }
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
    int32_t *action = nullptr, *active = nullptr, *mask = nullptr;
    int32_t *world_id = nullptr, *agent_id = nullptr, *loc_world_id = nullptr, *loc_id = nullptr;

    void launch(bool init, const int32_t *actions, hipStream_t stream)
    {
        StepParams a = params;
        a.actions = actions ? actions : action;
        if (init)
            hipLaunchKernelGGL(mrl_overcooked_step<true>, dim3(grid), dim3(kBlock), lds_bytes, stream, a);
        else
            hipLaunchKernelGGL(mrl_overcooked_step<false>, dim3(grid), dim3(kBlock), lds_bytes, stream, a);
        MRL_HIP(hipGetLastError());
    }

    void phase1(const int32_t *actions, hipStream_t stream) override { launch(false, actions, stream); }
    void phase2(const uint32_t *, hipStream_t) override {}

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
            *out = mrl::make_desc(params.obs, MRL_INT8, device, {P * C, N, F}, {F, P * C * F, 1});
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
            *out = mrl::make_desc(params.obs, MRL_INT8, device, {N, P, (int64_t)H, W, F});
            return true;
        case MRL_OVERCOOKED_STATE_PLAYERS: *out = mrl::make_desc(params.players, MRL_UINT8, device, {N, P, 8}); return true;
        case MRL_OVERCOOKED_STATE_OBJECTS: *out = mrl::make_desc(params.cell_obj, MRL_UINT8, device, {N, C, 4}); return true;
        case MRL_OVERCOOKED_STATE_TIMESTEP: *out = mrl::make_desc(params.timestep, MRL_INT32, device, {N}); return true;
        default: return false;
        }
    }

    const char *kernel_name() const override { return "mrl_overcooked_step<false>"; }

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
        a.c_pad = (a.C + 15u) & ~15u;
        a.p_pad = (a.P + 1u) & ~1u;
        const uint32_t tile_rows = a.rows < (uint32_t)kWave ? a.rows : (uint32_t)kWave;
        const uint32_t tile_bytes = ((tile_rows * a.F + 15u) & ~15u) + 32u;
        a.lds_wave_stride = a.c_pad * 4 + a.c_pad + a.p_pad * 8 + tile_bytes;
        sim->lds_bytes = kConstBytes + kWavesPerBlock * a.lds_wave_stride;
        const uint32_t blocks = (N + kWavesPerBlock - 1) / kWavesPerBlock;
        sim->grid = (blocks + 7u) & ~7u;

        uint32_t *d_consts = sim->arena.alloc<uint32_t>(kConstBytes / 4, false);
        MRL_HIP(hipMemcpy(d_consts, consts, kConstBytes, hipMemcpyHostToDevice));
        a.consts = d_consts;
        a.cell_obj = sim->arena.alloc<uint32_t>((size_t)N * C);
        a.players = sim->arena.alloc<uint2>((size_t)N * P);
        a.timestep = sim->arena.alloc<int32_t>(N);
        a.reward = sim->arena.alloc<int32_t>((size_t)N * P);
        a.done = sim->arena.alloc<int32_t>(N);
        a.obs = sim->arena.alloc<uint8_t>((size_t)N * a.block_bytes, false);
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
