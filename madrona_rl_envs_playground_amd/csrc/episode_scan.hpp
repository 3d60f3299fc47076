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
__attribute__((unused)) static __global__ void __launch_bounds__(256) sum_block_counts(const uint32_t *__restrict__ block_counts,
                                                                                       uint32_t num_blocks, uint32_t *__restrict__ shard_count)
{
    __shared__ uint32_t s_wave[4];
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < num_blocks; i += 256) mine += block_counts[i];
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63u) == 0) s_wave[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) *shard_count = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

struct GatheredCounts {
    const uint32_t *counts = nullptr;  // nullptr: not a gathered phase 2
    uint32_t num_ranks = 0, rank = 0;
};

// finished worlds of the ranks below this one (uniform scalar loads); *all = of every rank
__device__ __forceinline__ uint32_t lower_ranks(const GatheredCounts &g, uint32_t *all)
{
    uint32_t below = 0, total = 0;
    for (uint32_t r = 0; r < g.num_ranks; r++) {
        const uint32_t v = g.counts[r];
        below += r < g.rank ? v : 0u;
        total += v;
    }
    *all = total;
    return below;
}

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
// workgroup publishes (epoch, count) in one 64-bit word and then waits until all lower-numbered
// workgroups have published theirs for the same epoch (the host passes a fresh epoch per launch,
// so the status words are never cleared).
//
// Why the wait cannot deadlock, whatever the occupancy, the dispatch order and whatever else runs
// on the GPU: a workgroup's index in this protocol is NOT blockIdx.x but a TICKET, the value one
// atomic increment returned when the workgroup started (take_ticket below; the same device rocPRIM's
// look-back scan uses).  A workgroup that holds ticket t therefore knows that tickets 0..t-1 were
// handed out before, i.e. those workgroups are running or done.  It publishes before it waits
// and waits only for lower tickets, so the lowest unfinished ticket never waits for anybody: by
// induction every wait ends.  HIP promises nothing about dispatch order (MI355X_MICROARCH.md,
// "Workgroup dispatch"), and nothing here assumes any.  The ticket also decides which worlds the
// workgroup owns, so "ascending world order" and "ascending ticket" are the same order.
// The wait is bounded all the same; on expiry the Alarm is raised (the launch then finishes with
// wrong episode numbers instead of hanging the GPU) and every later call on the simulator fails.
//
// The persistent rollouts (mrl_*_rollout) are a different matter: there every workgroup waits for
// EVERY other one at each step, which needs the whole grid resident at once.  They are launched
// with hipLaunchCooperativeKernel, which refuses a grid the device cannot hold, and the host falls
// back to one launch per step when it does.
// ---------------------------------------------------------------------------------------------
// test hook (mrl_debug_set "inject_scan_timeout"): raises the alarm exactly as an expired wait would
__attribute__((unused)) static __global__ void raise_alarm_kernel(const Alarm alarm) { alarm.raise(); }

constexpr uint32_t kMaxFusedBlocks = 4096;
constexpr uint32_t kMaxPolls = 1u << 22;  // ~1 s of polling

// One returning atomic per workgroup (thread 0), broadcast through LDS.  `ticket` counts up for
// the simulator's whole life; `ticket_base` is its value when this launch started (the host
// knows it: launches of one simulator never overlap and each hands out exactly gridDim.x
// tickets), so the difference is this workgroup's index in the launch.  Ends with a barrier.
__device__ __forceinline__ uint32_t take_ticket(uint32_t *ticket, uint32_t ticket_base, uint32_t *s_slot)
{
    if (threadIdx.x == 0) *s_slot = atomicAdd(ticket, 1u) - ticket_base;
    __syncthreads();
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)*s_slot);
}

__device__ __forceinline__ void publish_count(unsigned long long *status, uint32_t block, uint32_t epoch, uint32_t count)
{
    __hip_atomic_store(&status[block], ((unsigned long long)epoch << 32) | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

// whole workgroup; same contract as scan_prefix
__device__ __forceinline__ uint32_t wait_prefix(unsigned long long *status, uint32_t num_blocks, uint32_t block, uint32_t epoch,
                                                uint32_t *s_red, bool want_total, uint32_t *grand_total, const Alarm &timed_out)
{
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    uint32_t before = 0, all = 0;
    const uint32_t limit = want_total ? num_blocks : block;
    for (uint32_t first = 0; first < limit; first += nthreads * 4u)
        all += read_counts<4>(status, first, limit, epoch, block, &before, timed_out, tid, nthreads);
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

// One wave's version of wait_prefix (no workgroup barrier): lets ONE wave of a workgroup do the
// look-back while the others go on.  A vector load cannot return before the wave's older stores
// have been acknowledged (loads and stores share vmcnt, in issue order), so the wave that looks
// back should do so before it streams out its own results.
__device__ __forceinline__ uint32_t wave_wait_prefix(unsigned long long *status, uint32_t num_blocks, uint32_t block, uint32_t epoch,
                                                     bool want_total, uint32_t *grand_total, const Alarm &timed_out)
{
    uint32_t before = 0, all = 0;
    const uint32_t limit = want_total ? num_blocks : block;
    for (uint32_t first = 0; first < limit; first += 64u * 8u) all += read_counts<8>(status, first, limit, epoch, block, &before, timed_out);
    for (int off = 32; off > 0; off >>= 1) {
        before += __shfl_xor(before, off, 64);
        all += __shfl_xor(all, off, 64);
    }
    *grand_total = all;
    return before;
}

}  // namespace mrl
