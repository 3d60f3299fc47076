// "Balance beam" world step for gfx950 (the reference's balance_beam_env): two agents on a line of five
// spaces, three steps per episode; one lane per world.
//
// Semantics: /root/reference/src/balance_beam_env/sim.cpp:76-92 (actionSystem: moves -2, -1, +1, +2),
// :94-97 (timeSystem), :99-112 (observationSystem: a three-deep history of own and partner positions,
// shifted by one per step), :114-151 (checkDone: reward, out-of-range and time-out termination),
// :45-74 (resetWorld from the episode-seeded generator, rng.hpp:5-40), graph :155-171.
//
// The whole world state is IN the observation the reference exports: obs.x[0] = own position + BUFFER,
// obs.time = steps left (sim.cpp:109-111), so the kernel keeps no other per-world array.  The history
// shift of the reference runs `for (i = 2*TIME; i > 0; i--) x[i] = x[i-1]` over an array of 2*TIME
// entries, i.e. it also writes x[6], which is the `time` field that follows it in the struct
// (sim.hpp:52-55) and is assigned right after; and x[TIME] is assigned right after as well.  Net effect,
// reproduced here: x[5] = x[4], x[4] = x[3], x[2] = x[1], x[1] = x[0], x[3] = partner + BUFFER,
// x[0] = own + BUFFER.
//
// Episode indices are taken in ascending world order within a step (see cartpole.hip):
//   mrl_step_phase1  mrl_balance_step : moves, history, reward, done flag, per-workgroup finished counts
//   mrl_step_phase2  mrl_balance_reset: exclusive prefix over the counts, re-seed finished worlds
// and mrl_step is the two in a row.
#include "common.hpp"
#include "episode_scan.hpp"
#include "random_policy.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kTime = 3, kSpaces = 5, kBuffer = 2, kRow = 2 * kTime + 1;  // sim.cpp:9-13, sim.hpp:8

__device__ __forceinline__ uint32_t seed_of(uint32_t episode)
{
    // rng.hpp:7-26
    uint32_t v0 = episode, v1 = 0, sum = 0;
#pragma unroll
    for (int round = 0; round < 8; round++) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

__device__ __forceinline__ float next_uniform(uint32_t &g)
{
    // rng.hpp:28-36
    g = 1664525u * g + 1013904223u;
    return (float)(g & 0x00FFFFFFu) / (float)0x01000000;
}

struct Row {
    int32_t x[kRow];  // x[0..5] history, x[6] = time left
};

__device__ __forceinline__ Row load_row(const int32_t *obs, uint32_t n, uint32_t agent, uint32_t w)
{
    Row r;
    const int32_t *src = obs + ((size_t)agent * n + w) * kRow;
#pragma unroll
    for (int k = 0; k < kRow; k++) r.x[k] = src[k];
    return r;
}

__device__ __forceinline__ void store_row(int32_t *obs, uint32_t n, uint32_t agent, uint32_t w, const Row &r)
{
    int32_t *dst = obs + ((size_t)agent * n + w) * kRow;
#pragma unroll
    for (int k = 0; k < kRow; k++) dst[k] = r.x[k];
}

// resetWorld (sim.cpp:45-74): positions from the episode's generator, empty history
__device__ __forceinline__ void fresh_rows(uint32_t episode, Row &r0, Row &r1)
{
    uint32_t g = seed_of(episode);
    const int32_t loc0 = (int32_t)(kSpaces * next_uniform(g));
    const int32_t loc1 = (int32_t)(kSpaces * next_uniform(g));
#pragma unroll
    for (int k = 0; k < 2 * kTime; k++) r0.x[k] = r1.x[k] = 0;
    r0.x[0] = loc0 + kBuffer;
    r0.x[kTime] = loc1 + kBuffer;
    r1.x[0] = loc1 + kBuffer;
    r1.x[kTime] = loc0 + kBuffer;
    r0.x[2 * kTime] = r1.x[2 * kTime] = kTime - 1;
}

__device__ __forceinline__ int32_t move_of(int32_t choice)
{
    // sim.cpp:78-91; anything else leaves the agent where it is
    return choice == 0 ? -2 : choice == 1 ? -1 : choice == 2 ? 1 : choice == 3 ? 2 : 0;
}

__device__ __forceinline__ void shift_history(Row &r, int32_t own, int32_t partner, int32_t time)
{
    // sim.cpp:104-111, see the header of this file
    r.x[5] = r.x[4];
    r.x[4] = r.x[3];
    r.x[2] = r.x[1];
    r.x[1] = r.x[0];
    r.x[kTime] = partner + kBuffer;
    r.x[0] = own + kBuffer;
    r.x[2 * kTime] = time;
}

// One world's step on its two rows: moves, history, reward, termination (sim.cpp:76-151).  Pure in (rows, actions):
// the single-launch step's healing look-back re-runs it on another workgroup's inputs to learn how many of its worlds finish.
__device__ __forceinline__ bool move_world(Row &r0, Row &r1, int32_t a0, int32_t a1, float &rew)
{
    const int32_t loc0 = r0.x[0] - kBuffer + move_of(a0), loc1 = r1.x[0] - kBuffer + move_of(a1);
    const int32_t time = r0.x[2 * kTime] - 1;
    shift_history(r0, loc0, loc1, time);
    shift_history(r1, loc1, loc0, time);
    // checkDone (sim.cpp:114-151): double arithmetic, rounded to float once
    const int32_t gap = loc0 > loc1 ? loc0 - loc1 : loc1 - loc0;
    rew = (float)(loc0 == loc1 ? 1.0 : -gap * 0.2);
    bool over = false;
    if (loc0 < 0 || loc0 >= kSpaces || loc1 < 0 || loc1 >= kSpaces) {
        over = true;
        rew = (float)(-kSpaces * (time + 1) * 0.2);
    }
    return over || time == 0;
}

__device__ __forceinline__ void actions_of(const int32_t *action, uint32_t n, uint32_t w, bool sampled, uint64_t seed, uint32_t step, int32_t &a0, int32_t &a1)
{
    if (sampled) {  // uniform over the four moves, drawn here (include/mrl_envs.h: mrl_rollout_random)
        a0 = (int32_t)mrl::scale(mrl::policy_hash(seed, step, w, 0), 4u);
        a1 = (int32_t)mrl::scale(mrl::policy_hash(seed, step, w, 1), 4u);
    } else {
        a0 = action[w];
        a1 = action[(size_t)n + w];
    }
}

// workgroup b owns worlds [b*chunk, (b+1)*chunk), chunk a multiple of kBlock
__global__ void __launch_bounds__(kBlock) mrl_balance_step(uint32_t n, uint32_t chunk, const int32_t *action,  // (no __restrict__: mrl_rollout_random passes the ACTION tensor as action_out too)
                                                           int32_t *__restrict__ obs, float *__restrict__ reward,
                                                           int32_t *__restrict__ done, uint32_t *__restrict__ block_counts,
                                                           unsigned long long *__restrict__ finished_mask,
                                                           int32_t *action_out, uint64_t sample_seed, uint32_t sample_step)
{
    // Besides the int32 done flags of the DONE tensor every wave stores the ballot of its 64 flags as one word of
    // finished_mask (world w = bit w % 64 of word w / 64; chunk and kBlock are multiples of 64): the reset launch reads
    // those words -- 128 bytes per 1024 worlds -- instead of walking the flags round by round.
    __shared__ uint32_t s_wave[kBlock / 64];
    const uint32_t first = blockIdx.x * chunk, last = min(n, first + chunk);
    uint32_t finished = 0;  // wave-uniform
    for (uint32_t w0 = first; w0 < last; w0 += kBlock) {  // uniform trip count
        const uint32_t w = w0 + threadIdx.x;
        bool over = false;
        if (w < last) {
        Row r0 = load_row(obs, n, 0, w), r1 = load_row(obs, n, 1, w);
        int32_t a0, a1;
        actions_of(action, n, w, action_out != nullptr, sample_seed, sample_step, a0, a1);
        if (action_out) {
            action_out[w] = a0;
            action_out[(size_t)n + w] = a1;
        }
        float rew;
        over = move_world(r0, r1, a0, a1, rew);
        if (!over) {  // a finished world's rows come from the reset
            store_row(obs, n, 0, w, r0);
            store_row(obs, n, 1, w, r1);
        }
        reward[w] = rew;
        reward[(size_t)n + w] = rew;
        done[w] = over ? 1 : 0;
        }
        const unsigned long long votes = __ballot(over);
        if ((threadIdx.x & 63u) == 0 && w < last) finished_mask[w >> 6] = votes;
        finished += (uint32_t)__popcll(votes);
    }
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = finished;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < kBlock / 64; w++) total += s_wave[w];
        block_counts[blockIdx.x] = total;
    }
}

// kAll: (re)initialise every world as episode world_offset + world (construction / mrl_reseed_shard)
constexpr uint32_t kTripWords = 64;  // mask words (64 worlds each) the reset launch compacts per trip
template <bool kAll>
__global__ void __launch_bounds__(kBlock) mrl_balance_reset(uint32_t n, uint32_t chunk, const unsigned long long *__restrict__ finished_mask,
                                                            int32_t *__restrict__ obs, const uint32_t *__restrict__ block_counts,
                                                            const uint32_t *episode_base, uint32_t world_offset,
                                                            uint32_t *next_counter, uint32_t *__restrict__ reset_count,
                                                            const mrl::GatheredCounts gathered, const mrl::DeviceCounter device_counter)
{
    __shared__ uint32_t s_red[2 * kBlock / 64];
    __shared__ unsigned long long s_word[kTripWords];
    __shared__ uint32_t s_before[kTripWords];
    __shared__ uint32_t s_total;
    __shared__ uint16_t s_list[kTripWords * 64];
    const bool last_block = blockIdx.x == gridDim.x - 1;
    const uint32_t first = blockIdx.x * chunk, last = min(n, first + chunk);
    if (kAll) {
        for (uint32_t w = first + threadIdx.x; w < last; w += kBlock) {
            Row r0, r1;
            fresh_rows(world_offset + w, r0, r1);
            store_row(obs, n, 0, w, r0);
            store_row(obs, n, 1, w, r1);
        }
        return;
    }
    const uint32_t words = (last - first + 63u) >> 6;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t unused_epoch = 0;
    device_counter.apply(episode_base, next_counter, unused_epoch);  // (the launch state may live in device memory: common.hpp)
    // the first trip's mask words are requested before the prefix is summed
    unsigned long long word = threadIdx.x < min(words, kTripWords) ? finished_mask[(first >> 6) + threadIdx.x] : 0ull;
    if (block_counts[blockIdx.x] == 0 && !last_block) return;  // nothing finished here (uniform per workgroup)
    uint32_t grand_total = 0;
    uint32_t running = mrl::scan_prefix(block_counts, gridDim.x, blockIdx.x, s_red, last_block, &grand_total);
    uint32_t base = *episode_base, all_ranks = grand_total;
    const uint32_t counter_now = base;
    if (gathered.counts) base += mrl::lower_ranks(gathered, &all_ranks);  // sharded batch: the ranks below come first
    // finished worlds of a trip are compacted into s_list in ascending world order (entry e is the e-th finished
    // world: episode base + running + e), then re-seeded one per thread on dense lanes
    for (uint32_t w0 = 0; w0 < words; w0 += kTripWords) {  // uniform trip count
        const uint32_t here = min(words - w0, kTripWords);
        if (wave == 0) {
            if (w0 > 0) word = lane < here ? finished_mask[(first >> 6) + w0 + lane] : 0ull;
            const uint32_t c = (uint32_t)__popcll(word);
            uint32_t x = c;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t y = __shfl_up(x, off, 64);
                x += lane >= (uint32_t)off ? y : 0u;
            }
            s_word[lane] = word;
            s_before[lane] = x - c;
            if (lane == 63) s_total = x;
        }
        __syncthreads();
        const uint32_t total = s_total;
        for (uint32_t k = wave; k < here; k += kBlock / 64) {  // one wave per word, lane = bit
            const unsigned long long m = s_word[k];
            if ((m >> lane) & 1ull) s_list[s_before[k] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((k << 6) + lane);
        }
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < total; e += kBlock) {
            const uint32_t w = first + (w0 << 6) + s_list[e];
            Row r0, r1;
            fresh_rows(base + running + e, r0, r1);
            store_row(obs, n, 0, w, r0);
            store_row(obs, n, 1, w, r1);
        }
        running += total;
        __syncthreads();  // s_word / s_before / s_list are rewritten by the next trip
    }
    if (last_block && threadIdx.x == 0) {
        *reset_count = grand_total;
        *next_counter = gathered.counts ? counter_now + all_ranks : base + grand_total;
    }
}

// The whole step in ONE launch (mrl_step on one GPU).  A third of the worlds finish in every step (episodes last at most
// three), so the two-launch pair writes a third of the rows twice over -- stepped rows it then skips, fresh rows as 28-byte
// pieces scattered by the re-seeding launch -- and that launch costs 15 us of the 36 at 1 M worlds.  Here workgroup b owns
// worlds [1024 b, 1024 b + 1024), four per thread in registers; a finished world's rows are replaced by the fresh episode's
// BEFORE they are stored, so every row is written exactly once, in the step's own coalesced stream.  Episode indices come
// from the single-launch look-back of episode_scan.hpp (a count that does not appear is recounted from that workgroup's rows
// and actions; rows are stored only behind the __syncthreads that makes the workgroup's own count globally visible).
constexpr int kFusedWorlds = 4;  // worlds per thread
__device__ __forceinline__ uint32_t recount_chunk(uint32_t n, const int32_t *action, const int32_t *obs, uint32_t j, bool sampled, uint64_t seed,
                                                   uint32_t step)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t first = j * (kFusedWorlds * kBlock), last = min(n, first + kFusedWorlds * kBlock);
    uint32_t count = 0;
    for (uint32_t w0 = first; w0 < last; w0 += 64u) {
        const uint32_t w = w0 + lane, wc = w < last ? w : first;
        Row r0 = load_row(obs, n, 0, wc), r1 = load_row(obs, n, 1, wc);
        int32_t a0, a1;
        actions_of(action, n, wc, sampled, seed, step, a0, a1);
        float rew;
        const bool over = move_world(r0, r1, a0, a1, rew) && w < last;
        count += (uint32_t)__popcll(__ballot(over));
    }
    return count;
}

__global__ void __launch_bounds__(kBlock) mrl_balance_step_fused(uint32_t n, const int32_t *action, int32_t *__restrict__ obs,
                                                                 float *__restrict__ reward, int32_t *__restrict__ done,
                                                                 uint32_t *status, unsigned long long *group_total, uint32_t epoch,
                                                                 const uint32_t *episode_base,
                                                                 uint32_t *next_counter, uint32_t *__restrict__ reset_count, int32_t *action_out,
                                                                 uint64_t sample_seed, uint32_t sample_step, const mrl::HealTest heal,
                                                                 const mrl::DeviceCounter device_counter,
                                                                 const mrl::FusedExchange fx)  // sharded batch: the other ranks' counts (episode_scan.hpp)
{
    __shared__ uint32_t s_wave[kFusedWorlds][kBlock / 64];
    __shared__ uint32_t s_prefix, s_lower, s_all;
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t first = b * (kFusedWorlds * kBlock), last = min(n, first + kFusedWorlds * kBlock);
    const bool last_block = b == gridDim.x - 1, sampled = action_out != nullptr;
    device_counter.apply(episode_base, next_counter, epoch);  // (the launch state may live in device memory: common.hpp)
    mrl::heal_test_delay(heal, b, gridDim.x, epoch);          // test hook only
    const uint32_t base = *episode_base;
    Row r0[kFusedWorlds], r1[kFusedWorlds];
    float rew[kFusedWorlds];
    bool over[kFusedWorlds];
    uint32_t rank[kFusedWorlds];  // among the workgroup's finished worlds, in ascending world order (round u covers worlds first + 256 u ...)
#pragma unroll
    for (int u = 0; u < kFusedWorlds; u++) {
        const uint32_t w = first + u * kBlock + threadIdx.x, wc = w < last ? w : first;
        r0[u] = load_row(obs, n, 0, wc);
        r1[u] = load_row(obs, n, 1, wc);
        int32_t a0, a1;
        actions_of(action, n, wc, sampled, sample_seed, sample_step, a0, a1);
        if (sampled && w < last) {
            action_out[w] = a0;
            action_out[(size_t)n + w] = a1;
        }
        over[u] = move_world(r0[u], r1[u], a0, a1, rew[u]) && w < last;
        const unsigned long long votes = __ballot(over[u]);
        rank[u] = (uint32_t)__popcll(votes & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[u][wave] = (uint32_t)__popcll(votes);
    }
    __syncthreads();
    uint32_t block_total = 0;
#pragma unroll
    for (int u = 0; u < kFusedWorlds; u++)
        for (uint32_t v = 0; v < kBlock / 64; v++) {
            const uint32_t c = s_wave[u][v];
            block_total += c;
#pragma unroll
            for (int u2 = 0; u2 < kFusedWorlds; u2++)  // everything in front of (round u2, wave `wave`) in world order
                rank[u2] += (u < u2 || (u == u2 && v < wave)) ? c : 0u;
        }
    if (threadIdx.x == 0) mrl::publish_count(status, b, epoch, block_total);
    __syncthreads();  // the count is globally visible (vmcnt(0) in front of the barrier) before any row of this workgroup changes
    const bool needs_prefix = block_total != 0 || last_block;
    if (threadIdx.x < 64) {  // (two levels: mrl::grouped_prefix, episode_scan.hpp; a workgroup that closes a group looks back whatever its count)
        const uint32_t before = mrl::grouped_prefix(status, group_total, b, epoch, block_total, needs_prefix, heal, [&](uint32_t j) {
            return recount_chunk(n, action, obs, j, sampled, sample_seed, sample_step);
        });
        uint32_t lower_ranks = 0, all_counts = before + block_total;
        if (fx.mail.num_ranks && needs_prefix) {  // the last workgroup tells every rank the shard's total; the ranks below come first in the numbering
            lower_ranks = mrl::fused_exchange(fx, last_block, before + block_total, fx.mail.rank);
            if (last_block) all_counts = mrl::fused_exchange(fx, false, 0u, fx.mail.num_ranks);
        }
        if (threadIdx.x == 0 && needs_prefix) {
            s_prefix = before;
            s_lower = lower_ranks;
            s_all = all_counts;
        }
    }
    mrl::lds_barrier();
    const uint32_t own_before = needs_prefix ? s_prefix : 0u;
    const uint32_t before = own_before + (needs_prefix ? s_lower : 0u);
    // (Storing the rows of the worlds that go on while the look-back is under way, and the fresh rows behind it, was measured:
    // 32.6 against 30.4 us per step at 1 M worlds -- the 28-byte rows of neighbouring worlds share cache lines, and writing a
    // line in two passes costs more than the 2 us of waiting.)
#pragma unroll
    for (int u = 0; u < kFusedWorlds; u++) {
        const uint32_t w = first + u * kBlock + threadIdx.x;
        if (w < last) {
            if (over[u]) fresh_rows(base + before + rank[u], r0[u], r1[u]);  // the finished world's next episode, stored in its place
            store_row(obs, n, 0, w, r0[u]);
            store_row(obs, n, 1, w, r1[u]);
            reward[w] = rew[u];
            reward[(size_t)n + w] = rew[u];
            done[w] = over[u] ? 1 : 0;
        }
    }
    if (last_block && threadIdx.x == 0) {
        *reset_count = own_before + block_total;
        *next_counter = base + s_all;
    }
}

__global__ void fill_balance_ids(int32_t *world_id, int32_t *agent_id, int32_t *active, int32_t *mask, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * n) {
        world_id[i] = (int32_t)(i % n);
        agent_id[i] = (int32_t)(i / n);
        active[i] = 1;
#pragma unroll
        for (int k = 0; k < 4; k++) mask[(size_t)i * 4 + k] = 1;
    }
}

struct BalanceSim final : mrl_sim {
    uint32_t grid = 0, chunk = 0, parity = 0;
    int32_t *action = nullptr, *obs = nullptr, *done = nullptr, *world_id = nullptr, *agent_id = nullptr, *active = nullptr, *mask = nullptr;
    float *reward = nullptr;
    uint32_t *block_counts = nullptr, *counter = nullptr, *reset_count = nullptr;
    uint32_t *shard_count = nullptr;  // SHARD_COUNT: finished worlds of the last mrl_step_phase1
    mrl::LaunchStateOwner launch_state;  // parity in device memory once a caller wants to capture steps (common.hpp)
    mrl::AlarmOwner alarm;               // raised when the mailbox exchange of a sharded step waited in vain (mrl_step_exchanged)
    bool scan_timed_out() const override { return alarm.raised(); }
    bool capturable() const override { return launch_state.device_mode; }
    void prepare_graph_capture(hipStream_t stream) override { launch_state.to_device(parity, epoch, stream); }
    unsigned long long *finished_mask = nullptr;  // one bit per world: the done flags, as each wave's ballot
    // single-launch step (mrl_balance_step_fused)
    uint32_t *status = nullptr;                 // 32-bit status words, and per 256 workgroups their total (mrl::grouped_prefix)
    unsigned long long *group_total = nullptr;
    mrl::HealTest heal;
    uint32_t fused_grid = 0, epoch = 0;
    bool fused_step = false;

    void launch_fused(const int32_t *actions, int32_t *action_out, uint64_t seed, uint32_t sample_step, hipStream_t stream,
                      const mrl::FusedExchange &fx = mrl::FusedExchange{})
    {
        epoch += 1;
        if (launch_state.device_mode) launch_state.advance(stream);
        hipLaunchKernelGGL(mrl_balance_step_fused, dim3(fused_grid), dim3(kBlock), 0, stream, num_worlds, actions ? actions : action, obs, reward,
                           done, status, group_total, epoch, counter + parity, counter + (parity ^ 1u), reset_count, action_out, seed, sample_step,
                           heal, launch_state.counter_args(counter), fx);
        MRL_HIP(hipGetLastError());
        parity ^= 1u;
    }
    void step(const int32_t *actions, hipStream_t stream) override
    {
        if (fused_step)
            launch_fused(actions, nullptr, 0, 0, stream);
        else
            mrl_sim::step(actions, stream);
    }
    // a shard's step with the other ranks' counts taken from the mailboxes inside the single launch (episode_scan.hpp)
    void step_exchanged(const int32_t *actions, hipStream_t stream) override
    {
        if (fused_step)
            launch_fused(actions, nullptr, 0, 0, stream, mrl::fused_exchange_of(exchange, alarm.alarm()));
        else
            mrl_sim::step_exchanged(actions, stream);
    }

    void launch_step(const int32_t *actions, int32_t *action_out, uint64_t seed, uint32_t sample_step, hipStream_t stream)
    {
        hipLaunchKernelGGL(mrl_balance_step, dim3(grid), dim3(kBlock), 0, stream, num_worlds, chunk, actions ? actions : action, obs, reward,
                           done, block_counts, finished_mask, action_out, seed, sample_step);
        MRL_HIP(hipGetLastError());
    }
    void phase1(const int32_t *actions, hipStream_t stream) override { launch_step(actions, nullptr, 0, 0, stream); }
    void launch_reset(const uint32_t *base, const mrl::GatheredCounts &gathered, hipStream_t stream, bool external_base = false)
    {
        if (launch_state.device_mode) launch_state.advance(stream);
        hipLaunchKernelGGL((mrl_balance_reset<false>), dim3(grid), dim3(kBlock), 0, stream, num_worlds, chunk, finished_mask, obs, block_counts, base,
                           0u, counter + (parity ^ 1u), reset_count, gathered, launch_state.counter_args(counter, external_base));
        MRL_HIP(hipGetLastError());
        parity ^= 1u;
    }
    void publish_shard_count(hipStream_t stream) override
    {
        hipLaunchKernelGGL(mrl::sum_block_counts, dim3(1), dim3(256), 0, stream, block_counts, grid, shard_count, mrl::mail_of(exchange));
        MRL_HIP(hipGetLastError());
    }
    void phase2(const uint32_t *episode_base_dev, hipStream_t stream) override
    {
        launch_reset(episode_base_dev ? episode_base_dev : counter + parity, mrl::GatheredCounts{}, stream, episode_base_dev != nullptr);
    }
    void phase2_gathered(const uint32_t *counts, uint32_t num_ranks, uint32_t rank, hipStream_t stream) override
    {
        mrl::GatheredCounts g;
        g.counts = counts;
        g.num_ranks = num_ranks;
        g.rank = rank;
        launch_reset(counter + parity, g, stream);
    }
    void phase2_exchanged(hipStream_t stream) override { launch_reset(counter + parity, mrl::polled_counts(exchange, alarm.alarm()), stream); }
    void rollout_random(uint32_t num_steps, uint64_t seed, uint32_t first_step, hipStream_t stream) override
    {
        for (uint32_t k = 0; k < num_steps; k++) {
            if (fused_step) {
                launch_fused(action, action, seed, first_step + k, stream);
            } else {
                launch_step(action, action, seed, first_step + k, stream);
                phase2(nullptr, stream);
            }
        }
    }
    void set_episode_counter(uint32_t next_episode, hipStream_t stream) override
    {
        if (launch_state.device_mode) {  // which half is current is only known on the device
            hipLaunchKernelGGL(mrl::set_current_counter, dim3(1), dim3(1), 0, stream, counter, launch_state.dev, next_episode);
            MRL_HIP(hipGetLastError());
        } else {
            MRL_HIP(hipMemcpyAsync(counter + parity, &next_episode, sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        }
        MRL_HIP(hipStreamSynchronize(stream));
    }
    void reseed_shard(uint32_t world_offset, uint32_t num_worlds_total, hipStream_t stream) override
    {
        const uint32_t *none = nullptr;
        uint32_t *no_out = nullptr;
        hipLaunchKernelGGL((mrl_balance_reset<true>), dim3(grid), dim3(kBlock), 0, stream, num_worlds, chunk, finished_mask, obs, block_counts, none,
                           world_offset, no_out, no_out, mrl::GatheredCounts{}, mrl::DeviceCounter{});
        MRL_HIP(hipGetLastError());
        MRL_HIP(hipMemsetAsync(done, 0, sizeof(int32_t) * num_worlds, stream));
        MRL_HIP(hipMemsetAsync(reward, 0, sizeof(float) * 2 * num_worlds, stream));
        set_episode_counter(num_worlds_total, stream);
    }

    bool tensor(int slot, mrl_tensor_desc *out) override
    {
        const int64_t N = num_worlds;
        switch (slot) {
        case MRL_BALANCE_DONE: *out = mrl::make_desc(done, MRL_INT32, device, {N}); return true;
        case MRL_BALANCE_ACTIVE_AGENT: *out = mrl::make_desc(active, MRL_INT32, device, {2, N}); return true;
        case MRL_BALANCE_ACTION: *out = mrl::make_desc(action, MRL_INT32, device, {2, N, 1}); return true;
        case MRL_BALANCE_OBSERVATION: *out = mrl::make_desc(obs, MRL_INT32, device, {2, N, kRow}); return true;
        case MRL_BALANCE_ACTION_MASK: *out = mrl::make_desc(mask, MRL_INT32, device, {2, N, 4}); return true;
        case MRL_BALANCE_REWARD: *out = mrl::make_desc(reward, MRL_FLOAT32, device, {2, N}); return true;
        case MRL_BALANCE_WORLD_ID: *out = mrl::make_desc(world_id, MRL_INT32, device, {2, N}); return true;
        case MRL_BALANCE_AGENT_ID: *out = mrl::make_desc(agent_id, MRL_INT32, device, {2, N}); return true;
        case MRL_BALANCE_RESET_COUNT: *out = mrl::make_desc(reset_count, MRL_UINT32, device, {1}); return true;
        case MRL_BALANCE_SHARD_COUNT: *out = mrl::make_desc(shard_count, MRL_UINT32, device, {1}); return true;
        default: return false;
        }
    }
    size_t action_elems() const override { return (size_t)2 * num_worlds; }
    const char *kernel_name() const override { return fused_step ? "mrl_balance_step_fused" : "mrl_balance_step"; }
    // actions 8 + both agents' rows r/w 2 * 2 * 28 + reward 8 + done 4
    uint64_t bytes_per_world_step() const override { return 8 + 4 * kRow * 4 + 8 + 4; }
    void launch_shape(uint32_t out[4]) const override
    {
        out[0] = grid;
        out[1] = kBlock;
        out[2] = out[3] = 0;
    }
};

}  // namespace

mrl_sim *mrl::create_balance(int gpu_id, uint32_t num_worlds)
{
    if (num_worlds == 0) {
        set_error("balance: num_worlds must be > 0");
        throw HipError{MRL_ERR_INVALID};
    }
    bind_device(gpu_id);
    auto *sim = new BalanceSim();
    try {
        sim->game = MRL_GAME_BALANCE;
        sim->device = gpu_id;
        sim->num_worlds = num_worlds;
        const uint32_t groups = (num_worlds + kBlock - 1) / kBlock;
        const uint32_t blocks = groups < mrl::kMaxScanBlocks ? groups : mrl::kMaxScanBlocks;
        sim->chunk = ((groups + blocks - 1) / blocks) * kBlock;
        sim->grid = (num_worlds + sim->chunk - 1) / sim->chunk;
        const size_t N = num_worlds;
        sim->action = sim->arena.alloc<int32_t>(2 * N);
        sim->obs = sim->arena.alloc<int32_t>(2 * N * kRow);
        sim->done = sim->arena.alloc<int32_t>(N);
        sim->reward = sim->arena.alloc<float>(2 * N);
        sim->world_id = sim->arena.alloc<int32_t>(2 * N, false);
        sim->agent_id = sim->arena.alloc<int32_t>(2 * N, false);
        sim->active = sim->arena.alloc<int32_t>(2 * N, false);
        sim->mask = sim->arena.alloc<int32_t>(2 * N * 4, false);
        sim->block_counts = sim->arena.alloc<uint32_t>(sim->grid);
        sim->finished_mask = sim->arena.alloc<unsigned long long>(((size_t)sim->grid * sim->chunk + 63) / 64);
        sim->counter = sim->arena.alloc<uint32_t>(2);
        sim->reset_count = sim->arena.alloc<uint32_t>(1);
        sim->shard_count = sim->arena.alloc<uint32_t>(1);
        sim->launch_state.init(sim->arena);
        sim->alarm.init(sim->arena);
        {
            const uint32_t blocks = (num_worlds + kFusedWorlds * kBlock - 1) / (kFusedWorlds * kBlock);
            if (blocks <= mrl::kMaxFusedBlocks) {
                sim->fused_grid = blocks;
                sim->status = sim->arena.alloc<uint32_t>(blocks);
                sim->group_total = sim->arena.alloc<unsigned long long>((blocks + mrl::kGroup - 1) / mrl::kGroup);
                sim->heal.mod = (uint32_t)mrl::debug_get("fused_heal_test", 0);
                sim->heal.seen = sim->arena.alloc<uint32_t>(blocks);
                sim->fused_step = mrl::debug_get("fused_step", 0) != 2;  // 0 / 1: one launch (every row written once), 2: the two-launch pair
            }
        }
        hipLaunchKernelGGL(fill_balance_ids, dim3((unsigned)((2 * N + 255) / 256)), dim3(256), 0, 0, sim->world_id, sim->agent_id, sim->active,
                           sim->mask, num_worlds);
        MRL_HIP(hipGetLastError());
        sim->reseed_shard(0, num_worlds, 0);  // Sim::Sim (sim.cpp:173-202): world w starts as episode w
        MRL_HIP(hipDeviceSynchronize());
    } catch (...) {
        delete sim;
        throw;
    }
    return sim;
}
