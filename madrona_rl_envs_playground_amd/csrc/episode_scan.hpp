// Deterministic episode numbering for Hanabi and Cartpole.
//
// The reference draws each new episode's index from one process-wide atomic
// (src/hanabi_env/sim.cpp:449-451, src/cartpole_env/sim.cpp:51-53), i.e. in thread-arrival
// order.  Here finishing worlds take consecutive indices in ascending world order, which needs
// an exclusive prefix sum over "finished" counts between the two launches of a step:
//   launch 1 (step)  : block_counts[b] = finished worlds in workgroup b's contiguous chunk
//   launch 2 (reset) : prefix(b) = sum of block_counts[0..b)
// Both launches use the same grid of at most kMaxScanBlocks workgroups, each owning one
// contiguous chunk of worlds, so the prefix is at most kMaxScanBlocks L2-resident reads per
// workgroup and needs no atomics.  (Measured alternatives on MI355X, 1M Cartpole worlds:
// one workgroup per 256 worlds with an O(b) prefix 18.9 us/step; the same with coarse bins
// filled by atomicAdd 39 us/step -- 4096 adds onto 64 words serialise.)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"

namespace mrl {

constexpr uint32_t kMaxScanBlocks = 1024;  // 4 workgroups per CU on MI355X

// Workgroup barrier for hand-offs through LDS only.  __syncthreads() also waits for the wave's
// outstanding global stores (vmcnt(0)) -- microseconds when a wave has just streamed its output
// rows with write-through stores, and needed by none of the hand-offs in these kernels.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- batches sharded over several GPUs (mrl_step_phase1 / mrl_step_phase2_gathered) ----
// mrl_step_phase1 leaves the shard's finished worlds in one word (SHARD_COUNT): a one-workgroup launch behind the
// step kernel adds up its per-workgroup counts.  (One fire-and-forget atomic per workgroup onto that word from the step
// kernel itself was measured first: 1024 same-address atomics drain at ~8 ns each and the launch cannot end before
// they have -- Cartpole at 1 M worlds 14.9 -> 22.8 us per step, paid by unsharded runs too.)  Between the phases the
// ranks all-gather the word, and phase 2 of rank r takes base = own counter + counts of the ranks below r.
//
// The exchange itself has two forms.  (1) A collective: the caller all-gathers the ranks' SHARD_COUNT words (RCCL) and hands
// them to mrl_step_phase2_gathered.  (2) A MAILBOX (mrl_exchange_*, round 4): every rank owns a small block of
// fine-grained device memory that its peers have mapped through IPC handles; the count launch of rank r stores
// (step tag, count) into word r of EVERY rank's mailbox -- G plain stores over xGMI -- and phase 2 polls the G words of
// its own mailbox for this step's tag: no host call, no collective, nothing between the launches of a step.  A rank can
// be at most one step ahead of a peer that has not read yet (its next count needs that peer's next word), so the words
// live in a ring of kMailSlots step slots.  The wait is bounded like the persistent rollouts' (SCAN_TIMEOUT).
constexpr uint32_t kMailSlots = 4;

struct ShardMail {  // where the count launch publishes to (all zero: nowhere)
    unsigned long long *peer[MRL_MAX_RANKS];  // rank p's mailbox as mapped into this process (peer[rank] is this rank's own)
    uint32_t num_ranks, rank, tag;
};

__attribute__((unused)) static __global__ void __launch_bounds__(256) sum_block_counts(const uint32_t *__restrict__ block_counts,
                                                                                       uint32_t num_blocks, uint32_t *__restrict__ shard_count,
                                                                                       const ShardMail mail)
{
    __shared__ uint32_t s_wave[4];
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < num_blocks; i += 256) mine += block_counts[i];
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63u) == 0) s_wave[threadIdx.x >> 6] = mine;
    __syncthreads();
    const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (threadIdx.x == 0) *shard_count = total;
    if (threadIdx.x < mail.num_ranks)  // one store per peer, into that peer's memory
        __hip_atomic_store(&mail.peer[threadIdx.x][(mail.tag % kMailSlots) * MRL_MAX_RANKS + mail.rank],
                           ((unsigned long long)mail.tag << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct GatheredCounts {
    const uint32_t *counts = nullptr;  // nullptr: not a gathered phase 2
    uint32_t num_ranks = 0, rank = 0;
    // mailbox form: the words are polled for `tag` (then `counts` only says "sharded": it points at the mailbox too)
    const unsigned long long *mail = nullptr;
    uint32_t tag = 0;
    Alarm timed_out{};
};

// finished worlds of the ranks below this one (uniform loads); *all = of every rank
__device__ __forceinline__ uint32_t lower_ranks(const GatheredCounts &g, uint32_t *all)
{
    uint32_t below = 0, total = 0;
    for (uint32_t r = 0; r < g.num_ranks; r++) {
        uint32_t v;
        if (g.mail) {
            const unsigned long long *word = &g.mail[(g.tag % kMailSlots) * MRL_MAX_RANKS + r];
            unsigned long long w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (uint32_t polls = 0; (uint32_t)(w >> 32) != g.tag; polls++) {
                if (polls == (1u << 22)) {  // ~1 s: a peer never published this step
                    g.timed_out.raise();
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            v = (uint32_t)w;
        } else {
            v = g.counts[r];
        }
        below += r < g.rank ? v : 0u;
        total += v;
    }
    *all = total;
    return below;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t x)
{
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// The single-launch steps take part in the same exchange from INSIDE the kernel (mrl_step_exchanged where the game's
// single launch exists): the last workgroup, whose look-back yields the shard's total, stores the word into every peer's
// mailbox; the wave of every workgroup that needs a prefix then adds the words of the ranks below (lane r polls word r),
// and the last workgroup the words of all ranks for the counter.  One launch per step, one hop over xGMI behind the
// shard's own look-back.  Unlike the look-back among the workgroups of one GPU this wait cannot heal itself -- another
// rank's inputs are out of reach -- so it is bounded and raises the alarm, like a collective whose peer never arrives.
struct FusedExchange {
    ShardMail mail;                   // num_ranks == 0: not a sharded step
    const unsigned long long *mine;   // this rank's mailbox
    Alarm timed_out;
};

// (wave-uniform call) publishes `shard_total` if `publish`, then returns to every lane the sum of the words of ranks
// [0, upto) of this step
__device__ __forceinline__ uint32_t fused_exchange(const FusedExchange &fx, bool publish, uint32_t shard_total, uint32_t upto)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t slot = (fx.mail.tag % kMailSlots) * MRL_MAX_RANKS;
    if (publish && lane < fx.mail.num_ranks)
        __hip_atomic_store(&fx.mail.peer[lane][slot + fx.mail.rank], ((unsigned long long)fx.mail.tag << 32) | shard_total, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t v = 0;
    if (lane < upto) {
        unsigned long long w = __hip_atomic_load(&fx.mine[slot + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (uint32_t polls = 0; (uint32_t)(w >> 32) != fx.mail.tag; polls++) {
            if (polls == (1u << 22)) {
                fx.timed_out.raise();
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            w = __hip_atomic_load(&fx.mine[slot + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        v = (uint32_t)w;
    }
    return wave_sum(v);
}

// host side: the kernel arguments of the two launches from the simulator's ShardExchange (common.hpp)
inline ShardMail mail_of(const ShardExchange &x)
{
    ShardMail m{};
    if (x.connected && x.publishing) {
        for (uint32_t p = 0; p < x.num_ranks; p++) m.peer[p] = x.peer[p];
        m.num_ranks = x.num_ranks;
        m.rank = x.rank;
        m.tag = x.step;
    }
    return m;
}
inline FusedExchange fused_exchange_of(const ShardExchange &x, const Alarm &alarm)
{
    FusedExchange f{};
    f.mail = mail_of(x);
    f.mine = x.mine;
    f.timed_out = alarm;
    return f;
}
inline GatheredCounts polled_counts(const ShardExchange &x, const Alarm &alarm)
{
    GatheredCounts g;
    g.counts = reinterpret_cast<const uint32_t *>(x.mine);  // "a sharded phase 2"; the words themselves are read through `mail`
    g.num_ranks = x.num_ranks;
    g.rank = x.rank;
    g.mail = x.mine;
    g.tag = x.step;
    g.timed_out = alarm;
    return g;
}

// ---- device-resident launch state (mrl_prepare_graph_capture; common.hpp LaunchState) ----
// one thread, enqueued in front of the kernels of a step that read the state
__attribute__((unused)) static __global__ void advance_launch_state(LaunchState *st)
{
    st->flips += 1u;
    st->epoch += 1u;
}
// mrl_set_episode_counter in device mode: the current half is only known on the device
__attribute__((unused)) static __global__ void set_current_counter(uint32_t *counter, const LaunchState *st, uint32_t value)
{
    counter[st->flips & 1u] = value;
}

// Host side of that state, one per simulator.
struct LaunchStateOwner {
    LaunchState *dev = nullptr;
    bool device_mode = false;
    void init(DeviceArena &arena) { dev = arena.alloc<LaunchState>(1); }
    // host mirrors -> device, once; from here on every step advances the device copy itself
    void to_device(uint32_t parity, uint32_t epoch, hipStream_t stream)
    {
        if (device_mode) return;
        const LaunchState now{parity, epoch};
        MRL_HIP(hipMemcpyAsync(dev, &now, sizeof(now), hipMemcpyHostToDevice, stream));
        MRL_HIP(hipStreamSynchronize(stream));
        device_mode = true;
    }
    void advance(hipStream_t stream) const
    {
        hipLaunchKernelGGL(advance_launch_state, dim3(1), dim3(1), 0, stream, dev);
    }
    DeviceCounter counter_args(uint32_t *counter, bool external_base = false) const
    {
        return device_mode ? DeviceCounter{dev, counter, external_base ? 1u : 0u} : DeviceCounter{};
    }
};

// whole workgroup (blockDim.x a multiple of 64, <= 1024): exclusive prefix of workgroup `block`;
// with want_total also the sum over all workgroups.  s_red: 2 * blockDim.x / 64 words of LDS.
__device__ __forceinline__ uint32_t scan_prefix(const uint32_t *block_counts, uint32_t num_blocks, uint32_t block,
                                                uint32_t *s_red, bool want_total, uint32_t *grand_total)
{
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    uint32_t before = 0, all = 0;
    const uint32_t limit = want_total ? num_blocks : block;
    for (uint32_t i = tid; i < limit; i += nthreads) {
        const uint32_t v = block_counts[i];
        before += i < block ? v : 0u;
        all += v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        before += __shfl_down(before, off, 64);
        all += __shfl_down(all, off, 64);
    }
    const uint32_t nwaves = nthreads >> 6;
    if ((tid & 63u) == 0) {
        s_red[tid >> 6] = before;
        s_red[nwaves + (tid >> 6)] = all;
    }
    lds_barrier();
    uint32_t prefix = 0, total = 0;
    for (uint32_t w = 0; w < nwaves; w++) {
        prefix += s_red[w];
        total += s_red[nwaves + w];
    }
    lds_barrier();
    if (grand_total) *grand_total = total;
    return prefix;
}

// ---------------------------------------------------------------------------------------------
// Single-launch variant.  Instead of ending the kernel between "count" and "prefix", every
// workgroup publishes (epoch, count) in one 64-bit word and then looks back at the words of all
// lower-numbered workgroups for the same epoch (the host passes a fresh epoch per launch, so the
// status words are never cleared).
//
// Why the look-back cannot deadlock, whatever the occupancy, the dispatch order and whatever else
// runs on the GPU: it never DEPENDS on another workgroup.  A count that has not appeared after a
// short bounded wait is RECOMPUTED by the waiting wave from that workgroup's inputs -- the count of
// finishing worlds is a pure function of the worlds' state and actions before the step
// (wave_prefix_or_recount below; the game supplies the recount).  HIP promises nothing about
// dispatch order or co-residency (MI355X_MICROARCH.md, "Workgroup dispatch"), and nothing here
// assumes any: on an idle GPU all workgroups run side by side and publish within a microsecond of
// each other, so the recount is never taken; if a lower workgroup has not even started, the
// waiting one pays for its transition a second time and goes on.
//
// For the recount to read PRE-step inputs, a workgroup changes its worlds' state in HBM only after
// its own count is globally visible (publish, s_waitcnt vmcnt(0), then a flag in LDS the other
// waves look at before they store), and the healing wave re-reads the status word after the
// recount: if the count has appeared meanwhile it takes the published one (the inputs it read may
// have been overwritten), otherwise the inputs it read were still the old ones.
//
// (Round 2 made the wait safe with TICKETS instead -- a workgroup's index was the value one
// returning atomic gave it, so everything it waited for had started.  The 256..1024 atomics on one
// word serialise at ~11 ns each and every workgroup's first load waits for its ticket: 3 us on the
// front of a 23 us launch, which made the single launch slower than two.  And a cooperative launch,
// which does guarantee co-residency, costs 25-30 us per launch on this runtime: tools/coop_probe.py.)
//
// The persistent rollouts (mrl_*_rollout) are a different matter: there every workgroup waits for
// EVERY other one at each step, which needs the whole grid resident at once.  They are launched
// with hipLaunchCooperativeKernel, which refuses a grid the device cannot hold, and the host falls
// back to one launch per step when it does; their waits are bounded and raise the Alarm on expiry.
// ---------------------------------------------------------------------------------------------
// test hook (mrl_debug_set "inject_scan_timeout"): raises the alarm exactly as an expired wait would
__attribute__((unused)) static __global__ void raise_alarm_kernel(const Alarm alarm) { alarm.raise(); }

constexpr uint32_t kMaxFusedBlocks = 4096;
constexpr uint32_t kMaxPolls = 1u << 22;  // ~1 s of polling (persistent rollouts)
constexpr uint32_t kHealPolls = 96;       // ~100 us: then the single-launch look-back recounts instead of waiting

__device__ __forceinline__ void publish_count(unsigned long long *status, uint32_t block, uint32_t epoch, uint32_t count)
{
    __hip_atomic_store(&status[block], ((unsigned long long)epoch << 32) | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A status word in 32 bits, for kernels whose look-back keeps many of them in registers: the count in the low 12 bits
// (a workgroup of at most 4095 finishing worlds), the low 20 bits of the epoch above.  Twenty bits are enough BECAUSE EVERY
// WORKGROUP PUBLISHES IN EVERY LAUNCH: the word of a workgroup that has not published yet carries the previous launch's
// epoch, never one 2^20 launches old.
__device__ __forceinline__ void publish_count(uint32_t *status, uint32_t block, uint32_t epoch, uint32_t count)
{
    __hip_atomic_store(&status[block], (epoch << 12) | (count & 0xfffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool status_is(unsigned long long word, uint32_t epoch) { return (uint32_t)(word >> 32) == epoch; }
__device__ __forceinline__ bool status_is(uint32_t word, uint32_t epoch) { return (word >> 12) == (epoch & 0xfffffu); }
__device__ __forceinline__ uint32_t status_count(unsigned long long word) { return (uint32_t)word; }
__device__ __forceinline__ uint32_t status_count(uint32_t word) { return word & 0xfffu; }

// Sum of the counts status[first .. first + stride * kBatch) visible to this thread (index first + lane + stride j),
// waiting for each word's epoch tag.  All loads of the batch are issued before the first tag is looked
// at: a one-at-a-time loop pays a full memory round trip per status word (8..16 of them per lane).
template <int kBatch>
__device__ __forceinline__ uint32_t read_counts(const unsigned long long *status, uint32_t first, uint32_t limit, uint32_t epoch,
                                                uint32_t below, uint32_t *sum_below, const Alarm &timed_out,
                                                uint32_t lane = threadIdx.x & 63u, uint32_t stride = 64u)
{
    unsigned long long v[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; j++) {
        const uint32_t i = first + lane + stride * j;
        v[j] = i < limit ? __hip_atomic_load(&status[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
    uint32_t all = 0;
#pragma unroll
    for (int j = 0; j < kBatch; j++) {
        const uint32_t i = first + lane + stride * j;
        if (i < limit) {
            for (uint32_t polls = 0; (uint32_t)(v[j] >> 32) != epoch; polls++) {
                if (polls == kMaxPolls) {
                    timed_out.raise();
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                v[j] = __hip_atomic_load(&status[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            all += (uint32_t)v[j];
            *sum_below += i < below ? (uint32_t)v[j] : 0u;
        }
    }
    return all;
}

// ---- the self-healing look-back of the single-launch step (one wave) ----
// Test hook (mrl_debug_set "fused_heal_test" = m > 0): workgroups whose index is a multiple of m (but not the last)
// behave like a workgroup that was dispatched late -- they do nothing until a higher workgroup has recounted them
// (heal_seen[index] == epoch) or a bounded wait expires -- so the recount path runs against untouched inputs.
struct HealTest {
    uint32_t mod = 0;
    uint32_t *seen = nullptr;  // one word per workgroup
};

__device__ __forceinline__ void heal_test_delay(const HealTest &t, uint32_t block, uint32_t num_blocks, uint32_t epoch)
{
    if (t.mod == 0 || block % t.mod != 0 || block + 1 == num_blocks) return;  // uniform per workgroup
    if (threadIdx.x == 0) {
        for (uint32_t polls = 0; polls < (1u << 14); polls++) {
            if (__hip_atomic_load(&t.seen[block], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) break;
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
}

// The look-back in two halves, so that a kernel can put work between asking for the status words and needing them:
// lookback_issue requests the words of workgroups [first, min(limit, first + 64 K)) -- K per lane, all loads in flight
// together -- and lookback_finish turns them into this lane's share of the sum (to be added up over the wave by the
// caller, see wave_sum): words that have not appeared are asked for again TOGETHER (one round trip per poll, not one per
// word: under the other waves' row stores a round trip is ~2 us), and after kHealPolls polls recount(i) -- called by the
// WHOLE wave with a uniform i, returning workgroup i's count to every lane -- replaces the wait.
template <int K, typename Word>
__device__ __forceinline__ void lookback_issue(const Word *status, uint32_t first, uint32_t limit, Word (&v)[K])
{
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int j = 0; j < K; j++) {
        const uint32_t i = first + lane + 64u * j;
        v[j] = i < limit ? __hip_atomic_load(&status[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (Word)0;
    }
}

template <int K, typename Word, typename Recount>
__device__ __forceinline__ uint32_t lookback_finish(const Word *status, uint32_t first, uint32_t limit, uint32_t epoch,
                                                    const HealTest &test, Word (&v)[K], Recount &&recount)
{
    static_assert(K <= 32, "one bit per word in `missing`");
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t before = 0, missing = 0;
#pragma unroll
    for (int j = 0; j < K; j++) missing |= (first + lane + 64u * j < limit && !status_is(v[j], epoch)) ? 1u << j : 0u;
    for (uint32_t polls = 0; polls < kHealPolls && __ballot(missing != 0) != 0ull; polls++) {
        __builtin_amdgcn_s_sleep(2);
#pragma unroll
        for (int j = 0; j < K; j++)
            if ((missing >> j) & 1u) v[j] = __hip_atomic_load(&status[first + lane + 64u * j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < K; j++)
            if (status_is(v[j], epoch)) missing &= ~(1u << j);
    }
#pragma unroll
    for (int j = 0; j < K; j++)
        if (first + lane + 64u * j < limit && !((missing >> j) & 1u)) before += status_count(v[j]);
    if (__ballot(missing != 0) == 0ull) return before;
    for (int j = 0; j < K; j++) {
        unsigned long long todo = __ballot((missing >> j) & 1u);
        while (todo) {
            const uint32_t src = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t i = first + src + 64u * j;  // wave-uniform
            uint32_t c = recount(i);
            // A count that has appeared meanwhile wins: the inputs just read may already have been overwritten.  That
            // only works if the recount's loads are performed BEFORE the status word is read again -- the recount's result
            // depends on them, but nothing else orders two loads of one wave, so wait for them here (cold path).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const Word now = __hip_atomic_load(&status[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (status_is(now, epoch)) c = status_count(now);
            if (test.mod && lane == 0) __hip_atomic_store(&test.seen[i], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            before += lane == src ? c : 0u;
        }
    }
    return before;
}


// Sum of the counts of workgroups [from, block) for `epoch`, to every lane of the calling wave.
template <typename Word, typename Recount>
__device__ __forceinline__ uint32_t wave_prefix_or_recount(const Word *status, uint32_t block, uint32_t epoch,
                                                           const HealTest &test, Recount &&recount, uint32_t from = 0)
{
    uint32_t before = 0;
    for (uint32_t first = from; first < block; first += 64u * 8u) {
        Word v[8];
        lookback_issue<8>(status, first, block, v);
        before += lookback_finish<8>(status, first, block, epoch, test, v, recount);
    }
    return wave_sum(before);
}

// ---- the look-back in two levels (Cartpole, balance beam: up to 4096 workgroups) ----
// A workgroup's status word (32 bits) and, per GROUP of 256 consecutive workgroups, the group's total, published by the
// group's last workgroup once its own look-back over the other 255 is done.  A workgroup needs the words of the lower
// workgroups of its own group (up to four per lane) and the totals of the lower groups (up to sixteen in all).  Batches of
// up to 256 workgroups are one group: one hand-off through memory, ~1.5 us under the other waves' traffic; larger ones pay
// a second for the group totals.  (Round 4 measured the flat form at 1024 workgroups -- every workgroup reading all lower
// status words, up to sixteen per lane: the 4 KB those words occupy are one hot spot that a million uncached loads queue
// on, 3-5 us per look-back, profiles/r04_e_cartpole_fused_timeline_flat_lookback.txt -- and asking for the words BEFORE
// the work that hides their latency: they come back stale, the workgroups around publish at the same moment, and the
// second asking queues behind the first.)  A group total that does not appear is replaced by its 256 status words, a
// status word that does not appear by a recount: nothing waits on another workgroup.
constexpr uint32_t kGroup = 256;
constexpr int kGroupWords = kGroup / 64;  // status words per lane

// Called by ONE wave of workgroup `block`, which has published its own count, with wave-uniform arguments.  Publishes the
// group's total if this workgroup closes a group; returns to every lane the counts of workgroups [0, block) summed up
// (0 unless needs_prefix).
template <typename Recount>
__device__ __forceinline__ uint32_t grouped_prefix(const uint32_t *status, unsigned long long *group_total, uint32_t block, uint32_t epoch,
                                                   uint32_t block_total, bool needs_prefix, const HealTest &heal, Recount &&recount)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t group = block / kGroup, group_first = group * kGroup;
    const bool closes_group = block + 1 == group_first + kGroup;  // the group's last workgroup publishes the group's total
    if (!needs_prefix && !closes_group) return 0;
    uint32_t lower[kGroupWords];
    unsigned long long lower_groups[1];
    lookback_issue<kGroupWords>(status, group_first, block, lower);
    lookback_issue<1>(group_total, 0, needs_prefix ? group : 0u, lower_groups);
    const uint32_t in_group = wave_sum(lookback_finish<kGroupWords>(status, group_first, block, epoch, heal, lower, recount));
    if (closes_group && lane == 0) publish_count(group_total, group, epoch, in_group + block_total);
    if (!needs_prefix) return 0;
    auto regroup = [&](uint32_t g) {  // a group total that has not appeared: the group's 256 words instead
        uint32_t words[kGroupWords];
        lookback_issue<kGroupWords>(status, g * kGroup, g * kGroup + kGroup, words);
        return wave_sum(lookback_finish<kGroupWords>(status, g * kGroup, g * kGroup + kGroup, epoch, heal, words, recount));
    };
    return in_group + wave_sum(lookback_finish<1>(group_total, 0, group, epoch, HealTest{}, lower_groups, regroup));
}

}  // namespace mrl
