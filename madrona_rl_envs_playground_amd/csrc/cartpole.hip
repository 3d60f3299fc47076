// Cartpole world step for gfx950: one lane per world (the state is four floats,
// so a wave moves 64 worlds with one 16-byte load and one 16-byte store per lane).
//
// Semantics: /root/reference/src/cartpole_env/sim.cpp:9-21 (constants are double
// literals, so most intermediates are double), :68-84 (explicit Euler step),
// :86-96 (termination), :48-66 (reset from the episode-seeded generator,
// rng.hpp:5-40).  This file is compiled with -ffp-contract=off: the reference's
// CPU executor rounds every product and sum separately.
//
// The reference hands out episode indices from one global atomic in whatever
// order worlds happen to reset (sim.cpp:51-53).  Here the order is fixed:
// ascending world index within a step.  That needs a prefix sum over the
// termination flags:
//   mrl_step             one launch (mrl_cartpole_step_fused, in-kernel prefix, episode_scan.hpp)
//   mrl_step_phase1 / 2  two launches for sharded batches:
//     mrl_cartpole_step : dynamics + done flag + per-workgroup reset counts
//     mrl_cartpole_reset: exclusive prefix over the counts, re-seed finished worlds
// HBM traffic per world-step: action 4 + state r/w 32 + reward 4 + done 4 = 44 B.
#include "common.hpp"
#include "episode_scan.hpp"
#include "random_policy.hpp"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr int kBlock = 256;

#define GRAVITY 9.8
#define MASSCART 1.0
#define MASSPOLE 0.1
#define TOTAL_MASS (MASSPOLE + MASSCART)
#define LENGTH 0.5
#define POLEMASS_LENGTH (MASSPOLE * LENGTH)
#define FORCE_MAG 10
#define TAU 0.02
#define X_THRESHOLD 2.4
#define PI_D 3.141592653589793238463
#define THETA_THRESHOLD (12 * 2 * PI_D / 360)

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

__device__ __forceinline__ float4 fresh_state(uint32_t episode)
{
    // sim.cpp:55-65
    uint32_t g = seed_of(episode);
    const float lo = -0.05f, span = 0.05f - (-0.05f);
    float4 s;
    s.x = lo + next_uniform(g) * span;
    s.y = lo + next_uniform(g) * span;
    s.z = lo + next_uniform(g) * span;
    s.w = lo + next_uniform(g) * span;
    return s;
}

// ---- the transition (sim.cpp:68-96), in four arithmetic variants (mrl_debug_set "cartpole.variant") ----
// The reference mixes float state with double literals, so nearly every intermediate is a double and the three
// quotients by TOTAL_MASS plus the one by the pole term are four IEEE double divisions (v_div_scale x2 / v_rcp /
// 7 fma / v_div_fmas / v_div_fixup each) -- ~85 double-precision instructions per world, and the step was bound by issuing
// them.  What the contract asks is 1e-5 on identical actions (BASELINE.json), the reference's own check 1e-6
// against a float64 numpy twin whose sin/cos are not sinf/cosf either (envs/cartpole_env.py:177-233,277).
//   kRefTyped   every expression typed and rounded as sim.cpp writes it (rounds 1-3)
//   kLean       the same float roundings (temp, costheta*temp, thetaacc, xacc and the four state components are rounded
//               to float exactly where the reference rounds them) around double intermediates that are computed with
//               fused multiply-adds, the constant quotients as products with the rounded reciprocal, and the one real
//               quotient as v_rcp_f64 + one Newton step + one residual correction.  A double intermediate differs from the
//               reference's by a few units in ITS last place, which survives the rounding to float with probability
//               ~2^-20 per world-step and is then one float ulp.
//   kLeanBounded  kLean, with sinf/cosf evaluated without range reduction while |theta| <= pi/4 (a live pole is within
//               12 degrees; the library call, which carries a Payne-Hanek path, remains for anything else)
//   kFloat      float throughout (fused multiply-adds, v_rcp_f32 + residual correction), bounded sin/cos
enum Variant : int { kRefTyped = 0, kLean = 1, kLeanBounded = 2, kFloat = 3, kNumVariants = 4 };
constexpr int kDefaultVariant = kLeanBounded;

// sin and cos of one float by minimax polynomials on [-pi/4, pi/4] (the classic single-precision kernels, < 1 ulp there);
// outside that range the caller takes the library's sincosf.  The choice is per world: a world's result never depends on
// its neighbours in the wave or in the thread.
constexpr float kQuarterPi = 0.78539816f;
__device__ __forceinline__ void sincos_poly(float x, float *s, float *c)
{
    const float z = x * x;
    float p = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = __builtin_fmaf(p, z, -1.6666654611e-1f);
    *s = __builtin_fmaf(p * z, x, x);
    float q = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    q = __builtin_fmaf(q, z, 4.166664568298827e-2f);
    *c = __builtin_fmaf(q * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
}

// termination (sim.cpp:88-91 compares the float state with double literals): for a float v, v > T (double) is
// v > largest float <= T, and v < -T likewise, so two float compares of |v| decide it (NaN: false on both sides, as there)
__device__ __forceinline__ bool out_of_bounds(float x, float theta)
{
    constexpr float kXLimit = 0x1.333332p+1f;      // largest float <= 2.4 (2.4f itself rounds up)
    constexpr float kThetaLimit = 0x1.acee9ep-3f;  // largest float <= 12 * 2 pi / 360
    static_assert((double)kXLimit <= X_THRESHOLD && (double)kThetaLimit <= THETA_THRESHOLD, "limits must round down");
    static_assert((double)(kXLimit + 0x1p-22f) > X_THRESHOLD && (double)(kThetaLimit + 0x1p-26f) > THETA_THRESHOLD, "and be the nearest such floats");
    return fabsf(x) > kXLimit || fabsf(theta) > kThetaLimit;
}

// The transition in two halves.  The new cart position and pole angle -- and with them the termination flag -- depend on
// the OLD velocities only (explicit Euler, sim.cpp:79,81,88-91): two multiply-adds per world.  The single-launch step works
// that half out first, publishes its workgroup's count of finished worlds, and runs the expensive half (sin, cos, the
// accelerations) while the count travels and the look-back reads the other workgroups' (mrl_cartpole_step_fused).
template <int V> __device__ __forceinline__ bool next_pose(const float4 &s, float &x, float &theta)
{
    if constexpr (V == kRefTyped) {
        x = s.x + TAU * s.y;
        theta = s.z + TAU * s.w;
        return x < -X_THRESHOLD || x > X_THRESHOLD || theta < -THETA_THRESHOLD || theta > THETA_THRESHOLD;  // sim.cpp:88-91
    } else if constexpr (V == kLean || V == kLeanBounded) {
        x = (float)__builtin_fma(TAU, (double)s.y, (double)s.x);
        theta = (float)__builtin_fma(TAU, (double)s.w, (double)s.z);
        return out_of_bounds(x, theta);
    } else {
        x = __builtin_fmaf((float)TAU, s.y, s.x);
        theta = __builtin_fmaf((float)TAU, s.w, s.z);
        return out_of_bounds(x, theta);
    }
}

template <int V>
__device__ __forceinline__ void rates_from(const float4 &s, int32_t action, float sintheta, float costheta, float &x_dot, float &theta_dot)
{
    x_dot = s.y;
    theta_dot = s.w;
    if constexpr (V == kRefTyped) {
        // expression types as written in sim.cpp:70-83
        const float force = (action == 1 ? FORCE_MAG : -FORCE_MAG);
        const float temp = (force + POLEMASS_LENGTH * theta_dot * theta_dot * sintheta) / TOTAL_MASS;
        const float thetaacc =
            (GRAVITY * sintheta - costheta * temp) / (LENGTH * (4.0 / 3.0 - MASSPOLE * costheta * costheta / TOTAL_MASS));
        const float xacc = temp - POLEMASS_LENGTH * thetaacc * costheta / TOTAL_MASS;
        x_dot = x_dot + TAU * xacc;
        theta_dot = theta_dot + TAU * thetaacc;
    } else if constexpr (V == kLean || V == kLeanBounded) {
        constexpr double kInvMass = 1.0 / TOTAL_MASS;                            // 1 / 1.1
        constexpr double kPoleOverMass = POLEMASS_LENGTH / TOTAL_MASS;           // 0.05 / 1.1
        constexpr double kFourThirdsLength = LENGTH * 4.0 / 3.0;                 // 0.5 * 4/3
        const double sn = sintheta, cs = costheta, td = theta_dot;
        const double force = action == 1 ? (double)FORCE_MAG : -(double)FORCE_MAG;
        const float temp = (float)(__builtin_fma((POLEMASS_LENGTH * td) * td, sn, force) * kInvMass);
        const float ct = costheta * temp;  // float * float in the reference too
        const double num = __builtin_fma(GRAVITY, sn, -(double)ct);
        const double den = __builtin_fma(-kPoleOverMass, cs * cs, kFourThirdsLength);  // in [0.62, 0.67]
        double r = __builtin_amdgcn_rcp(den);               // ~2^-23
        r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);  // ~2^-46
        double q = num * r;
        q = __builtin_fma(__builtin_fma(-den, q, num), r, q);  // residual correction: a unit or two in the last place
        const float thetaacc = (float)q;
        const double ta = thetaacc;
        const float xacc = (float)__builtin_fma(-kPoleOverMass * ta, cs, (double)temp);
        x_dot = (float)__builtin_fma(TAU, (double)xacc, (double)x_dot);
        theta_dot = (float)__builtin_fma(TAU, ta, td);
    } else {
        constexpr float kInvMass = (float)(1.0 / TOTAL_MASS);
        constexpr float kPoleOverMass = (float)(POLEMASS_LENGTH / TOTAL_MASS);
        constexpr float kFourThirdsLength = (float)(LENGTH * 4.0 / 3.0);
        const float force = action == 1 ? (float)FORCE_MAG : -(float)FORCE_MAG;
        const float temp = __builtin_fmaf(((float)POLEMASS_LENGTH * theta_dot) * theta_dot, sintheta, force) * kInvMass;
        const float num = __builtin_fmaf((float)GRAVITY, sintheta, -(costheta * temp));
        const float den = __builtin_fmaf(-kPoleOverMass, costheta * costheta, kFourThirdsLength);
        const float r = __builtin_amdgcn_rcpf(den);
        float q = num * r;
        q = __builtin_fmaf(__builtin_fmaf(-den, q, num), r, q);
        const float thetaacc = q;
        const float xacc = __builtin_fmaf(-kPoleOverMass * thetaacc, costheta, temp);
        x_dot = __builtin_fmaf((float)TAU, xacc, x_dot);
        theta_dot = __builtin_fmaf((float)TAU, thetaacc, theta_dot);
    }
}

// The expensive half for the K worlds of a thread.  With the bounded sin / cos all K polynomials run unconditionally and
// ONE branch per thread covers the (never taken, for a live pole) library call: the arithmetic of the K worlds is then one
// straight line of code the scheduler interleaves, with its double-precision constants set up once.
template <int V, int K>
__device__ __forceinline__ void next_rates(const float4 (&s)[K], const int32_t (&action)[K], float (&x_dot)[K], float (&theta_dot)[K])
{
    float sn[K], cs[K];
    if constexpr (V == kRefTyped || V == kLean) {
#pragma unroll
        for (int u = 0; u < K; u++) sincosf(s[u].z, &sn[u], &cs[u]);  // one range reduction for both (sim.cpp:71-72 calls cosf and sinf)
    } else {
        bool wide = false;
#pragma unroll
        for (int u = 0; u < K; u++) {
            sincos_poly(s[u].z, &sn[u], &cs[u]);
            wide |= !(fabsf(s[u].z) <= kQuarterPi);
        }
        if (wide) {
#pragma unroll
            for (int u = 0; u < K; u++)
                if (!(fabsf(s[u].z) <= kQuarterPi)) sincosf(s[u].z, &sn[u], &cs[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < K; u++) rates_from<V>(s[u], action[u], sn[u], cs[u], x_dot[u], theta_dot[u]);
}

// sim.cpp:68-96 for the K worlds of a thread; over[u] = world u's done flag
template <int V, int K> __device__ __forceinline__ void advance(float4 (&s)[K], const int32_t (&action)[K], bool (&over)[K])
{
    float x[K], theta[K], x_dot[K], theta_dot[K];
#pragma unroll
    for (int u = 0; u < K; u++) over[u] = next_pose<V>(s[u], x[u], theta[u]);
    next_rates<V, K>(s, action, x_dot, theta_dot);
#pragma unroll
    for (int u = 0; u < K; u++) s[u] = make_float4(x[u], x_dot[u], theta[u], theta_dot[u]);
}

constexpr int kUnroll = 4;  // worlds per thread whose loads are in flight together

// Both kernels run on the same grid: workgroup b owns worlds [b*chunk, (b+1)*chunk), chunk a
// multiple of kBlock, and walks it kBlock worlds at a time -- kUnroll rounds per trip with all
// their loads issued before the first is used (a one-round-per-trip loop keeps one 16-byte load
// per lane in flight and measured 10.9 us per launch at 1M worlds, i.e. latency-bound).
//
// Besides the int32 done flags of the RESET tensor, every wave stores the ballot of its 64 done
// flags as one word of `finished_mask` (world i is bit i % 64 of word i / 64; a wave's worlds are
// 64-aligned because chunk and kBlock are multiples of 64).  The reset launch reads those words
// -- 128 bytes per 1024 worlds instead of 4 KB of flags -- and ranks the set bits with popcounts.
template <int V>
__global__ void __launch_bounds__(kBlock) mrl_cartpole_step(uint32_t n, uint32_t chunk, const int32_t *__restrict__ action,
                                                            float4 *__restrict__ state, float *__restrict__ reward,
                                                            int32_t *__restrict__ done, uint32_t *__restrict__ block_counts,
                                                            unsigned long long *__restrict__ finished_mask)
{
    __shared__ uint32_t s_wave[kBlock / 64];
    const uint32_t first = blockIdx.x * chunk, last = min(n, first + chunk);
    uint32_t finished = 0;  // wave-uniform
    for (uint32_t i0 = first + threadIdx.x; i0 - threadIdx.x < last; i0 += kUnroll * kBlock) {  // uniform trip count
        float4 s[kUnroll];
        int32_t a[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const uint32_t i = i0 + u * kBlock;
            const uint32_t ic = i < last ? i : first;  // clamped: loads stay in bounds
            s[u] = state[ic];
            a[u] = action[ic];
        }
        bool finishes[kUnroll];
        advance<V, kUnroll>(s, a, finishes);  // (rounds past the end run on world `first`'s values and are dropped)
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const uint32_t i = i0 + u * kBlock;
            const bool over = finishes[u] && i < last;
            if (i < last) {
                state[i] = s[u];
                reward[i] = 1.f;
                done[i] = over ? 1 : 0;
            }
            const unsigned long long votes = __ballot(over);
            const uint32_t word = (i - (threadIdx.x & 63u)) >> 6;  // wave-uniform
            if ((threadIdx.x & 63u) == 0 && i < last) finished_mask[word] = votes;
            finished += (uint32_t)__popcll(votes);
        }
    }
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = finished;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < kBlock / 64; w++) total += s_wave[w];
        block_counts[blockIdx.x] = total;
    }
}

constexpr uint32_t kTripWords = 64;  // mask words (64 worlds each) the reset launch compacts per trip

// The finished worlds of a trip are first compacted into s_list in ascending world order (entry e
// is the e-th finished world, so its episode is base + running + e), then re-seeded one per
// thread: the seed hash is ~150 dependent instructions, so it runs once on dense lanes rather
// than once per mask word on the few lanes whose bit is set.
__global__ void __launch_bounds__(kBlock) mrl_cartpole_reset(uint32_t n, uint32_t chunk, float4 *__restrict__ state,
                                                             const uint32_t *__restrict__ block_counts,
                                                             const unsigned long long *__restrict__ finished_mask,
                                                             const uint32_t *episode_base, uint32_t *next_counter,
                                                             uint32_t *__restrict__ reset_count, const mrl::GatheredCounts gathered,
                                                             const mrl::DeviceCounter device_counter)
{
    __shared__ uint32_t s_red[2 * kBlock / 64];
    __shared__ unsigned long long s_word[kTripWords];
    __shared__ uint32_t s_before[kTripWords];
    __shared__ uint32_t s_total;
    __shared__ uint16_t s_list[kTripWords * 64];
    const bool last_block = blockIdx.x == gridDim.x - 1;
    uint32_t unused_epoch = 0;
    device_counter.apply(episode_base, next_counter, unused_epoch);  // (the launch state may live in device memory: common.hpp)
    const uint32_t mine = block_counts[blockIdx.x];
    const uint32_t first = blockIdx.x * chunk, last = min(n, first + chunk);
    const uint32_t words = (last - first + 63u) >> 6;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // the first trip's mask words are requested before the prefix is summed
    unsigned long long word = threadIdx.x < min(words, kTripWords) ? finished_mask[(first >> 6) + threadIdx.x] : 0ull;
    if (mine == 0 && !last_block) return;  // nothing finished here (uniform per workgroup)
    uint32_t grand_total = 0;
    uint32_t running = mrl::scan_prefix(block_counts, gridDim.x, blockIdx.x, s_red, last_block, &grand_total);
    uint32_t base = *episode_base, all_ranks = grand_total;
    const uint32_t counter_now = base;
    if (gathered.counts) base += mrl::lower_ranks(gathered, &all_ranks);  // sharded batch: the ranks below come first
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
        for (uint32_t e = threadIdx.x; e < total; e += kBlock)
            state[first + (w0 << 6) + s_list[e]] = fresh_state(base + running + e);
        running += total;
        __syncthreads();  // s_word / s_before / s_list are rewritten by the next trip
    }
    if (last_block && threadIdx.x == 0) {
        *reset_count = grand_total;
        *next_counter = gathered.counts ? counter_now + all_ranks : base + grand_total;
    }
}

// How many of workgroup j's worlds finish in this step, worked out by ONE wave of another workgroup from j's inputs in
// HBM: what the healing look-back of the single-launch step calls for a workgroup whose own count has not appeared
// (episode_scan.hpp).  Inlined: a call would give the kernel a stack in scratch memory.  It never runs on an idle GPU.
template <int V>
__device__ __forceinline__ uint32_t recount_chunk(uint32_t n, const float4 *state, uint32_t j)
{
    constexpr uint32_t kChunk = kUnroll * kBlock;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t first = j * kChunk, last = min(n, first + kChunk);
    uint32_t count = 0;
    for (uint32_t i0 = first; i0 < last; i0 += 64u) {
        const uint32_t i = i0 + lane, ic = i < last ? i : first;
        const float4 s = state[ic];
        float x, theta;
        const bool over = next_pose<V>(s, x, theta) && i < last;  // (the flag does not depend on the action)
        count += (uint32_t)__popcll(__ballot(over));
    }
    return count;
}

// The whole step in one launch (mrl_step / mrl_step_with_actions on one GPU): workgroup b owns worlds [1024 b, 1024 b + 1024),
// four per thread, all in registers from the first load to the last store.  Finished worlds get their episode index from
// the single-launch look-back of episode_scan.hpp (a count that does not appear is recounted from that workgroup's
// inputs: no workgroup ever depends on another one making progress) and are written once, already re-seeded; the
// two-launch pair above stays for the sharded path, whose episode base comes from the other ranks between the phases.
//
// Order of events in a workgroup (round 4): state loads, then action loads; new position and angle of its 1024 worlds (two
// multiply-adds each) -> who finishes -> the count is PUBLISHED ~30 instructions after the state has arrived; the first wave
// asks for the lower workgroups' counts right away (they were dispatched earlier and are ahead); then everybody runs the
// expensive half of the transition, under which the count's acknowledgement and the look-back's answers come in; one
// barrier; all stores.  A workgroup stores its worlds' state only after its count is globally visible (a healing
// workgroup that does not see the count reads the state as the step's input).  The finished worlds of a wave -- a dozen
// of its 256 under a random policy -- are re-seeded by as many lanes in ONE pass over a list in LDS (the seed hash is
// ~150 dependent instructions; per round it would run four times for three lanes each).
// (the look-back has two levels: mrl::grouped_prefix, episode_scan.hpp)
template <int V>
__global__ void __launch_bounds__(kBlock) mrl_cartpole_step_fused(uint32_t n, const int32_t *action,  // (no __restrict__: may be action_out)
                                                                  float4 *__restrict__ state, float *__restrict__ reward,
                                                                  int32_t *__restrict__ done, uint32_t *status,
                                                                  unsigned long long *group_total,
                                                                  uint32_t epoch, const uint32_t *episode_base,
                                                                  uint32_t *next_counter,
                                                                  uint32_t *__restrict__ reset_count,
                                                                  int32_t *action_out, uint64_t sample_seed, uint32_t sample_step,
                                                                  const mrl::HealTest heal, const mrl::DeviceCounter device_counter,
                                                                  const mrl::FusedExchange fx  // sharded batch: the other ranks' counts (episode_scan.hpp)
#ifdef MRL_DIAG
                                                                  , unsigned long long *stamps  // diagnostic build: s_memrealtime stamps per wave
#endif
)
{
#ifdef MRL_DIAG
#define CP_STAMP(k)                                                                                                        \
    do {                                                                                                                  \
        if (stamps && (threadIdx.x & 63u) == 0) stamps[(size_t)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define CP_STAMP(k) do { } while (0)
#endif
    CP_STAMP(0);
    // Four workgroups share a CU, one wave of each per SIMD, and the SIMD issues its oldest wave's vector instructions
    // first: without this a wave of a high workgroup whose state has arrived sits behind the lower workgroups' expensive
    // halves (~0.5 us each) before it can even count its finished worlds -- and everybody above waits for that count.
    __builtin_amdgcn_s_setprio(3);
    // action_out != nullptr: the reference harness's randint(high=2) drawn here (random_policy.hpp)
    __shared__ uint32_t s_votes[kUnroll][kBlock / 64];
    __shared__ uint32_t s_prefix, s_all_ranks;
    __shared__ uint32_t s_list[kBlock / 64][kUnroll * 64];  // per wave: its finished worlds, (rank in the workgroup << 16) | local index
    const uint32_t b = blockIdx.x;
    const uint32_t first = b * (kUnroll * kBlock), last = min(n, first + kUnroll * kBlock);
    const bool last_block = b == gridDim.x - 1;
    device_counter.apply(episode_base, next_counter, epoch);  // (the launch state may live in device memory: common.hpp)
    mrl::heal_test_delay(heal, b, gridDim.x, epoch);  // test hook only (uniform branch on a kernel argument)
    float4 s[kUnroll];
    int32_t a[kUnroll];
    const uint32_t base = *episode_base;
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
        const uint32_t i = first + u * kBlock + threadIdx.x;
        s[u] = state[i < last ? i : first];  // clamped: loads stay in bounds
    }
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {  // the actions are not needed before the expensive half: behind the state in the queue
        const uint32_t i = first + u * kBlock + threadIdx.x;
        const uint32_t ic = i < last ? i : first;
        if (action_out) {
            a[u] = (int32_t)(mrl::policy_hash(sample_seed, sample_step, ic, 0) >> 31);
            if (i < last) action_out[i] = a[u];
        } else {
            a[u] = action[ic];
        }
    }
    // First the cheap half of the transition: new position and angle, hence the finished worlds -- one ballot per round
    // and wave (round u covers worlds first + 256 u ...: ascending world order = round, then wave, then lane).  Every
    // hand-off below goes through LDS only: a __syncthreads would also wait for the wave's outstanding stores to be
    // acknowledged, microseconds while the whole GPU is storing.
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    bool over[kUnroll];
    unsigned long long votes[kUnroll];
    float nx[kUnroll], ntheta[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
        const uint32_t i = first + u * kBlock + threadIdx.x;
        over[u] = next_pose<V>(s[u], nx[u], ntheta[u]) && i < last;
        votes[u] = __ballot(over[u]);
        if (lane == 0) s_votes[u][wave] = (uint32_t)__popcll(votes[u]);
    }
    CP_STAMP(1);
    mrl::lds_barrier();
    uint32_t block_total = 0, wave_total = 0;
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
        uint32_t before_me = block_total;
        for (uint32_t w = 0; w < kBlock / 64; w++) {
            const uint32_t c = s_votes[u][w];
            before_me += w < wave ? c : 0u;
            block_total += c;
        }
        const uint32_t below = (uint32_t)__popcll(votes[u] & ((1ull << lane) - 1ull));
        if (over[u]) s_list[wave][wave_total + below] = ((before_me + below) << 16) | (uint32_t)(u * kBlock + threadIdx.x);
        wave_total += (uint32_t)__popcll(votes[u]);
    }
    // the count leaves ~30 instructions after the state has come in and travels while the expensive half runs
    if (wave == 0 && lane == 0) mrl::publish_count(status, b, epoch, block_total);
    const bool needs_prefix = block_total != 0 || last_block;  // uniform per workgroup
    CP_STAMP(2);
    __builtin_amdgcn_s_setprio(0);
    // (sin, cos, the accelerations: ~2 us of issue per SIMD at 1 M worlds)
    {
        float nxd[kUnroll], nthd[kUnroll];
        next_rates<V, kUnroll>(s, a, nxd, nthd);
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            asm volatile("" : "+v"(nxd[u]), "+v"(nthd[u]));  // here, not sunk into the store phase
            s[u] = make_float4(nx[u], nxd[u], ntheta[u], nthd[u]);
        }
    }
    CP_STAMP(3);
    __builtin_amdgcn_s_setprio(3);
    // the count is globally visible before any wave of this workgroup overwrites a world's state: the publishing wave waits
    // for ITS store, long acknowledged by now
    if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mrl::lds_barrier();
    CP_STAMP(4);
    if (wave == 0) {  // before its own stores: a load would sit out their acknowledgement
        uint32_t before = mrl::grouped_prefix(status, group_total, b, epoch, block_total, needs_prefix, heal,
                                              [&](uint32_t j) { return recount_chunk<V>(n, state, j); });
        if (needs_prefix) {
            if (fx.mail.num_ranks) {
                // a shard of a larger batch: the last workgroup knows the shard's total and tells every rank; the ranks
                // below this one come first in the numbering, and the counter moves on by the sum over all ranks
                const uint32_t shard_total = before + block_total;  // (meaningful in the last workgroup)
                before += mrl::fused_exchange(fx, last_block, shard_total, fx.mail.rank);
                if (last_block) {
                    const uint32_t all_ranks = mrl::fused_exchange(fx, false, 0u, fx.mail.num_ranks);
                    if (lane == 0) s_all_ranks = all_ranks;
                }
            }
            if (lane == 0) s_prefix = before;
        }
    }
    CP_STAMP(5);
    // everything that does not need the prefix goes out (the other three waves are here right after the barrier)
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
        const uint32_t i = first + u * kBlock + threadIdx.x;
        if (i < last) {
            // Non-temporal stores: the 25 MB a step writes at 1 M worlds are read again by the next launch, not by this one, and
            // what is still dirty in the L2 when the kernel ends has to be written back before the next one starts; with the
            // hint the lines leave while the kernel runs.  Same box, us per step at 1 M worlds, five runs each: 8.9-9.5 against
            // 9.3-9.7 with ordinary stores (profiles/r04_ap_cartpole_nt_ab.txt); write-through (sc1) stores instead: 10.9 (r04_an).
#ifndef MRL_CARTPOLE_OUT_PLAIN
            if (!over[u]) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                f32x4 v;
                v.x = s[u].x;
                v.y = s[u].y;
                v.z = s[u].z;
                v.w = s[u].w;
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(state + i));
            }
            __builtin_nontemporal_store(1.f, reward + i);
            __builtin_nontemporal_store(over[u] ? 1 : 0, done + i);
#else
            if (!over[u]) state[i] = s[u];
            reward[i] = 1.f;
            done[i] = over[u] ? 1 : 0;
#endif
        }
    }
    CP_STAMP(6);
    if (!needs_prefix) return;
    mrl::lds_barrier();
    const uint32_t prefix = s_prefix;
    for (uint32_t e = lane; e < wave_total; e += 64u) {  // this wave's finished worlds, one per lane
        const uint32_t entry = s_list[wave][e];
        state[first + (entry & 0xffffu)] = fresh_state(base + prefix + (entry >> 16));
    }
    CP_STAMP(7);
    if (last_block && threadIdx.x == 0) {
        if (fx.mail.num_ranks) {
            // prefix includes the ranks below: this shard's own total is what its look-back and its own count add up to
            uint32_t lower = 0;
            for (uint32_t r = 0; r < fx.mail.rank; r++) lower += (uint32_t)fx.mine[(fx.mail.tag % mrl::kMailSlots) * MRL_MAX_RANKS + r];
            *reset_count = prefix - lower + block_total;
            *next_counter = base + s_all_ranks;
        } else {
            const uint32_t grand_total = prefix + block_total;  // the whole GPU's
            *reset_count = grand_total;
            *next_counter = base + grand_total;
        }
    }
}

// mrl_rollout_random in ONE launch: the four worlds of a thread stay in registers for all steps;
// every step still writes state, reward, done and the drawn action.  Episode numbering needs the
// same two grid-wide hand-offs per step as mrl_hanabi_rollout (hanabi.hip): the lower workgroups'
// finished counts of this step (waited for) and everybody's counts of the previous step, through
// a ring of four epoch-tagged status arrays.  All workgroups must be resident at once: the host
// launches it cooperatively (the runtime refuses a grid the device cannot hold) and otherwise runs
// one launch per step; waits are bounded (SCAN_TIMEOUT).
constexpr int kRing = 4;
constexpr int64_t kPersistentMaxWorlds = 1 << 30;  // (set from the measurement below)

template <int V>
__global__ void __launch_bounds__(kBlock, 4) mrl_cartpole_rollout(uint32_t n, float4 *__restrict__ state, float *__restrict__ reward,
                                                               int32_t *__restrict__ done, int32_t *__restrict__ action_out,
                                                               unsigned long long *ring, uint32_t epoch0, uint32_t num_steps,
                                                               uint32_t first_step, uint64_t seed,
                                                               const uint32_t *__restrict__ episode_base,
                                                               uint32_t *__restrict__ next_counter,
                                                               uint32_t *__restrict__ reset_count, const mrl::Alarm timed_out)
{
    __shared__ uint32_t s_votes[kUnroll][kBlock / 64];
    __shared__ uint32_t s_red[2 * kBlock / 64];
    __shared__ uint32_t s_list[kBlock / 64][kUnroll * 64];  // per wave: its finished worlds, (rank in the workgroup << 16) | local index
    __shared__ float4 s_fresh[kBlock / 64][kUnroll * 64];   // ... and their next episodes' states, on their way back to the owners' registers
    const uint32_t G = gridDim.x, b = blockIdx.x;
    const uint32_t first = b * (kUnroll * kBlock), last = min(n, first + kUnroll * kBlock);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool last_block = b == G - 1;
    float4 s[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
        const uint32_t i = first + u * kBlock + threadIdx.x;
        s[u] = state[i < last ? i : first];
    }
    uint32_t base = *episode_base;
    for (uint32_t k = 0; k < num_steps; k++) {
        const uint32_t epoch = epoch0 + k;
        unsigned long long *now = ring + (size_t)(epoch % kRing) * G;
        const unsigned long long *before_step = ring + (size_t)((epoch - 1u) % kRing) * G;
        // the same order of events as in the single-launch step: who finishes (two multiply-adds per world), the count
        // published, then the expensive half while the count travels; the finished worlds of a wave re-seeded in one pass
        bool over[kUnroll];
        int32_t drawn[kUnroll];
        unsigned long long votes[kUnroll];
        float ntheta[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const uint32_t i = first + u * kBlock + threadIdx.x;
            drawn[u] = (int32_t)(mrl::policy_hash(seed, first_step + k, i < last ? i : first, 0) >> 31);
            float nx;
            over[u] = next_pose<V>(s[u], nx, ntheta[u]) && i < last;
            s[u].x = nx;  // (the expensive half does not look at the position: no register of its own)
            votes[u] = __ballot(over[u]);
            if (lane == 0) s_votes[u][wave] = (uint32_t)__popcll(votes[u]);
        }
        mrl::lds_barrier();
        uint32_t block_total = 0, wave_total = 0;
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            uint32_t before_me = block_total;
            for (uint32_t w = 0; w < kBlock / 64; w++) {
                const uint32_t c = s_votes[u][w];
                before_me += w < wave ? c : 0u;
                block_total += c;
            }
            const uint32_t below = (uint32_t)__popcll(votes[u] & ((1ull << lane) - 1ull));
            if (over[u]) s_list[wave][wave_total + below] = ((before_me + below) << 16) | (uint32_t)(u * kBlock + threadIdx.x);
            wave_total += (uint32_t)__popcll(votes[u]);
        }
        if (threadIdx.x == 0) mrl::publish_count(now, b, epoch, block_total);
        {
            float nxd[kUnroll], nthd[kUnroll];
            next_rates<V, kUnroll>(s, drawn, nxd, nthd);
#pragma unroll
            for (int u = 0; u < kUnroll; u++) s[u] = make_float4(s[u].x, nxd[u], ntheta[u], nthd[u]);
        }
        // lower workgroups of this step (only if somebody here finished) and everybody's previous step
        uint32_t lower = 0, prev_all = 0, unused = 0;
        if (k > 0)
            for (uint32_t f = 0; f < G; f += kBlock * 4u) prev_all += mrl::read_counts<4>(before_step, f, G, epoch - 1u, 0u, &unused, timed_out, threadIdx.x, kBlock);
        if (block_total != 0)
            for (uint32_t f = 0; f < b; f += kBlock * 4u) lower += mrl::read_counts<4>(now, f, b, epoch, 0u, &unused, timed_out, threadIdx.x, kBlock);
        for (int off = 32; off > 0; off >>= 1) {
            lower += __shfl_down(lower, off, 64);
            prev_all += __shfl_down(prev_all, off, 64);
        }
        if (lane == 0) {
            s_red[wave] = lower;
            s_red[kBlock / 64 + wave] = prev_all;
        }
        mrl::lds_barrier();
        lower = 0;
        prev_all = 0;
        for (uint32_t w = 0; w < kBlock / 64; w++) {
            lower += s_red[w];
            prev_all += s_red[kBlock / 64 + w];
        }
        base += prev_all;  // first episode index of this step
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const uint32_t i = first + u * kBlock + threadIdx.x;
            if (i < last) {
                action_out[i] = drawn[u];
                if (!over[u]) state[i] = s[u];
                reward[i] = 1.f;
                done[i] = over[u] ? 1 : 0;
            }
        }
        if (block_total != 0) {  // uniform per workgroup
            for (uint32_t e = lane; e < wave_total; e += 64u) {  // this wave's finished worlds, one per lane
                const uint32_t entry = s_list[wave][e];
                const float4 fresh = fresh_state(base + lower + (entry >> 16));
                state[first + (entry & 0xffffu)] = fresh;
                s_fresh[wave][e] = fresh;
            }
            uint32_t rounds_before = 0;  // (the slot is worked out again from the ballots, which live in scalar registers)
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const uint32_t slot = rounds_before + (uint32_t)__popcll(votes[u] & ((1ull << lane) - 1ull));
                if (over[u]) s[u] = s_fresh[wave][slot];  // (written by this wave: DS operations of a wave execute in order)
                rounds_before += (uint32_t)__popcll(votes[u]);
            }
        }
        mrl::lds_barrier();  // s_red / s_votes / the lists are rewritten by the next step
    }
    if (last_block && num_steps > 0) {  // counter after the rollout: everybody's count of the last step
        const uint32_t epoch = epoch0 + num_steps - 1u;
        unsigned long long *now = ring + (size_t)(epoch % kRing) * G;
        uint32_t all = 0, unused = 0;
        for (uint32_t f = 0; f < G; f += kBlock * 4u) all += mrl::read_counts<4>(now, f, G, epoch, 0u, &unused, timed_out, threadIdx.x, kBlock);
        for (int off = 32; off > 0; off >>= 1) all += __shfl_down(all, off, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = all;
        mrl::lds_barrier();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (uint32_t w = 0; w < kBlock / 64; w++) total += s_red[w];
            *reset_count = total;
            *next_counter = base + total;
        }
    }
}

// fallback for batches too large for the single-launch step: draw into the ACTION tensor
__global__ void mrl_cartpole_draw_actions(int32_t *action, uint32_t n, uint64_t seed, uint32_t step)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) action[i] = (int32_t)(mrl::policy_hash(seed, step, i, 0) >> 31);
}

__global__ void mrl_cartpole_init(uint32_t n, uint32_t world_offset, float4 *state, int32_t *world_id)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        state[i] = fresh_state(world_offset + i);  // the constructor gives world i episode i (sim.cpp:141)
        world_id[i] = (int32_t)i;
    }
}

// runs f(std::integral_constant<int, V>) for the arithmetic variant picked at creation
template <typename F> void with_variant(int variant, F &&f)
{
    switch (variant) {
    case kRefTyped: f(std::integral_constant<int, kRefTyped>{}); break;
    case kLean: f(std::integral_constant<int, kLean>{}); break;
    case kFloat: f(std::integral_constant<int, kFloat>{}); break;
    default: f(std::integral_constant<int, kLeanBounded>{}); break;
    }
}

struct CartpoleSim final : mrl_sim {
    uint32_t grid = 0;
    int variant = kDefaultVariant;  // arithmetic of the transition (mrl_debug_set cartpole.variant: 1 + Variant; 0 = the default)
    int32_t *action = nullptr, *done = nullptr, *world_id = nullptr;
    float4 *state = nullptr;
    float *reward = nullptr;
    uint32_t *block_counts = nullptr;
    unsigned long long *finished_mask = nullptr;  // one bit per world: the done flags, as each wave's ballot
    uint32_t chunk = 0;  // worlds per workgroup
    uint32_t *counter = nullptr;  // [2]: double-buffered episode counter, [parity] is current
    uint32_t *reset_count = nullptr;
    uint32_t *shard_count = nullptr;  // SHARD_COUNT: finished worlds of the last mrl_step_phase1
    uint32_t parity = 0;
    // single-launch step (see mrl_cartpole_step_fused)
    uint32_t *status = nullptr;
    unsigned long long *group_total = nullptr;  // per 64 workgroups (see mrl_cartpole_step_fused)
    mrl::AlarmOwner alarm;
    mrl::HealTest heal;  // test hook of the healing look-back (mrl_debug_set fused_heal_test)
    uint32_t fused_grid = 0, epoch = 0;
    mrl::LaunchStateOwner launch_state;  // parity / epoch in device memory once a caller wants to capture steps (common.hpp)
    bool capturable() const override { return launch_state.device_mode; }
    void prepare_graph_capture(hipStream_t stream) override { launch_state.to_device(parity, epoch, stream); }
    bool scan_timed_out() const override { return alarm.raised(); }

#ifdef MRL_DIAG
    unsigned long long *stamps = nullptr;
#endif
    bool fused_step = false;  // one launch with the self-healing in-kernel look-back (the default where the grid allows; mrl_debug_set fused_step 2: two launches)

    void step(const int32_t *actions, hipStream_t stream) override
    {
        if (fused_grid == 0 || !fused_step) {
            mrl_sim::step(actions, stream);
            return;
        }
        launch_fused(actions ? actions : action, nullptr, 0, 0, stream);
    }

    void launch_fused(const int32_t *actions, int32_t *action_out, uint64_t seed, uint32_t sample_step, hipStream_t stream,
                      const mrl::FusedExchange &fx = mrl::FusedExchange{})
    {
        epoch += 1;
        if (launch_state.device_mode) launch_state.advance(stream);  // then parity / epoch come from device memory
        with_variant(variant, [&](auto v) {
            hipLaunchKernelGGL(mrl_cartpole_step_fused<decltype(v)::value>, dim3(fused_grid), dim3(kBlock), 0, stream, num_worlds,
                               actions, state, reward, done, status, group_total, epoch, counter + parity, counter + (parity ^ 1u), reset_count,
                               action_out, seed, sample_step, heal, launch_state.counter_args(counter), fx
#ifdef MRL_DIAG
                               , stamps
#endif
            );
        });
        MRL_HIP(hipGetLastError());
        parity ^= 1u;
    }

    // a shard's step with the other ranks' counts taken from the mailboxes inside the single launch (episode_scan.hpp)
    void step_exchanged(const int32_t *actions, hipStream_t stream) override
    {
        if (fused_grid == 0 || !fused_step) {
            mrl_sim::step_exchanged(actions, stream);
            return;
        }
        launch_fused(actions ? actions : action, nullptr, 0, 0, stream, mrl::fused_exchange_of(exchange, alarm.alarm()));
    }

    unsigned long long *ring = nullptr;
    uint32_t ring_epoch = 0;
    bool persistent_ok = false;  // the whole grid of mrl_cartpole_rollout is resident at once

    void rollout_random(uint32_t num_steps, uint64_t seed, uint32_t first_step, hipStream_t stream) override
    {
        if (num_steps == 0) return;
        if (persistent_ok && !launch_state.device_mode) {  // (a cooperative launch cannot be captured; its counters live on the host)
            // cooperative: the runtime checks the grid against what the device can hold at once and
            // refuses it otherwise -- then, and from then on, one launch per step
            uint32_t n = num_worlds, epoch0 = ring_epoch + 1u;
            const uint32_t *base = counter + parity;
            uint32_t *next = counter + (parity ^ 1u);
            mrl::Alarm al = alarm.alarm();
            void *args[] = {&n, &state, &reward, &done, &action, &ring, &epoch0, &num_steps, &first_step, &seed, &base, &next,
                            &reset_count, &al};
            hipError_t err = hipSuccess;
            with_variant(variant, [&](auto v) {
                err = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(&mrl_cartpole_rollout<decltype(v)::value>),
                                                 dim3(fused_grid), dim3(kBlock), args, 0, stream);
            });
            if (err == hipSuccess) {
                ring_epoch += num_steps;
                parity ^= 1u;
                return;
            }
            (void)hipGetLastError();  // clear it; the per-step path below needs no co-residency
            persistent_ok = false;
        }
        for (uint32_t k = 0; k < num_steps; k++) {
            if (fused_grid && fused_step) {
                launch_fused(action, action, seed, first_step + k, stream);
            } else {
                hipLaunchKernelGGL(mrl_cartpole_draw_actions, dim3((num_worlds + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                                   action, num_worlds, seed, first_step + k);
                mrl_sim::step(nullptr, stream);
            }
        }
    }

    void phase1(const int32_t *actions, hipStream_t stream) override
    {
        with_variant(variant, [&](auto v) {
            hipLaunchKernelGGL(mrl_cartpole_step<decltype(v)::value>, dim3(grid), dim3(kBlock), 0, stream, num_worlds, chunk,
                               actions ? actions : action, state, reward, done, block_counts, finished_mask);
        });
        MRL_HIP(hipGetLastError());
    }

    void launch_reset(const uint32_t *base, const mrl::GatheredCounts &gathered, hipStream_t stream, bool external_base = false)
    {
        if (launch_state.device_mode) launch_state.advance(stream);
        hipLaunchKernelGGL(mrl_cartpole_reset, dim3(grid), dim3(kBlock), 0, stream, num_worlds, chunk, state, block_counts,
                           finished_mask, base, counter + (parity ^ 1u), reset_count, gathered, launch_state.counter_args(counter, external_base));
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
        hipLaunchKernelGGL(mrl_cartpole_init, dim3((num_worlds + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, num_worlds,
                           world_offset, state, world_id);
        MRL_HIP(hipGetLastError());
        MRL_HIP(hipMemsetAsync(done, 0, sizeof(int32_t) * num_worlds, stream));
        MRL_HIP(hipMemsetAsync(reward, 0, sizeof(float) * num_worlds, stream));
        set_episode_counter(num_worlds_total, stream);
    }

    bool tensor(int slot, mrl_tensor_desc *out) override
    {
        const int64_t N = num_worlds;
        switch (slot) {
        case MRL_CARTPOLE_RESET: *out = mrl::make_desc(done, MRL_INT32, device, {N, 1}); return true;
        case MRL_CARTPOLE_ACTION: *out = mrl::make_desc(action, MRL_INT32, device, {N, 1}); return true;
        case MRL_CARTPOLE_STATE: *out = mrl::make_desc(state, MRL_FLOAT32, device, {N, 4}); return true;
        case MRL_CARTPOLE_REWARD: *out = mrl::make_desc(reward, MRL_FLOAT32, device, {N, 1}); return true;
        case MRL_CARTPOLE_WORLD_ID: *out = mrl::make_desc(world_id, MRL_INT32, device, {N, 1}); return true;
        case MRL_CARTPOLE_RESET_COUNT: *out = mrl::make_desc(reset_count, MRL_UINT32, device, {1}); return true;
        case MRL_CARTPOLE_SCAN_TIMEOUT: *out = mrl::make_desc(alarm.alarm().dev, MRL_UINT32, device, {1}); return true;
        case MRL_CARTPOLE_SHARD_COUNT: *out = mrl::make_desc(shard_count, MRL_UINT32, device, {1}); return true;
#ifdef MRL_DIAG
        case 14:
            if (!stamps) return false;
            *out = mrl::make_desc(stamps, MRL_UINT8, device, {(int64_t)fused_grid * (kBlock / 64) * 8 * 8});
            return true;
#endif
        default: return false;
        }
    }

    size_t action_elems() const override { return (size_t)num_worlds; }
    const char *kernel_name() const override { return fused_grid && fused_step ? "mrl_cartpole_step_fused" : "mrl_cartpole_step"; }
    const char *rollout_kernel_name() const override { return persistent_ok && !launch_state.device_mode ? "mrl_cartpole_rollout" : kernel_name(); }
    uint64_t bytes_per_world_step() const override { return 44; }
};

}  // namespace

mrl_sim *mrl::create_cartpole(int gpu_id, uint32_t num_worlds)
{
    if (num_worlds == 0) {
        set_error("cartpole: num_worlds must be > 0");
        throw HipError{MRL_ERR_INVALID};
    }
    bind_device(gpu_id);
    auto *sim = new CartpoleSim();
    try {
        sim->game = MRL_GAME_CARTPOLE;
        sim->device = gpu_id;
        sim->num_worlds = num_worlds;
        {
            const int64_t knob = mrl::debug_get("cartpole.variant", 0);  // 0: the default; 1 + Variant otherwise
            if (knob < 0 || knob > kNumVariants) {
                mrl::set_error("cartpole.variant must be 0 (default) or 1..%d", (int)kNumVariants);
                throw mrl::HipError{MRL_ERR_INVALID};
            }
            sim->variant = knob == 0 ? kDefaultVariant : (int)knob - 1;
        }
        {
            const uint32_t groups = (num_worlds + kBlock - 1) / kBlock;
            const uint32_t blocks = groups < mrl::kMaxScanBlocks ? groups : mrl::kMaxScanBlocks;
            sim->chunk = ((groups + blocks - 1) / blocks) * kBlock;
            sim->grid = (num_worlds + sim->chunk - 1) / sim->chunk;
        }
        sim->action = sim->arena.alloc<int32_t>(num_worlds);
        sim->done = sim->arena.alloc<int32_t>(num_worlds);
        sim->world_id = sim->arena.alloc<int32_t>(num_worlds);
        sim->state = sim->arena.alloc<float4>(num_worlds);
        sim->reward = sim->arena.alloc<float>(num_worlds);
        sim->block_counts = sim->arena.alloc<uint32_t>(sim->grid);
        sim->finished_mask = sim->arena.alloc<unsigned long long>(((size_t)sim->grid * sim->chunk + 63) / 64);
        sim->counter = sim->arena.alloc<uint32_t>(2);
        sim->reset_count = sim->arena.alloc<uint32_t>(1);
        sim->shard_count = sim->arena.alloc<uint32_t>(1);
        {
            const uint32_t blocks = (num_worlds + kUnroll * kBlock - 1) / (kUnroll * kBlock);
            if (blocks <= mrl::kMaxFusedBlocks) {
                sim->fused_grid = blocks;
                sim->status = sim->arena.alloc<uint32_t>(blocks);
                sim->group_total = sim->arena.alloc<unsigned long long>((blocks + mrl::kGroup - 1) / mrl::kGroup);
#ifdef MRL_DIAG
                if (mrl::debug_get("stamps", 0)) sim->stamps = sim->arena.alloc<unsigned long long>((size_t)blocks * (kBlock / 64) * 8);
#endif
            }
        }
        sim->alarm.init(sim->arena);
        sim->launch_state.init(sim->arena);
        {
            // mrl_debug_set fused_step: 0 = the library's choice, 1 = one launch, 2 = always two.  Since round 4 (count published
            // before the expensive half, two-level look-back) one launch wins at every size it exists for -- us per step one /
            // two launches, actions from a pool of eight tensors: 4096 worlds 4.3 / 6.1, 100 000 5.1 / 6.6, 262 144 6.2 / 9.7,
            // 524 288 7.7 / 9.6, 1 M 10.7 / 12.3 (tools/cartpole_probe.py, profiles/r04_i_cartpole_probe.txt); round 3's
            // crossover was 4096 worlds.  Two launches remain for the sharded path and above 4 M worlds (kMaxFusedBlocks).
            const int64_t knob = mrl::debug_get("fused_step", 0);
            sim->fused_step = knob != 2;
            sim->heal.mod = (uint32_t)mrl::debug_get("fused_heal_test", 0);
            sim->heal.seen = sim->arena.alloc<uint32_t>(sim->fused_grid ? sim->fused_grid : 1);
        }
        if (sim->fused_grid) {
            int per_cu = 0, cus = 0;
            with_variant(sim->variant, [&](auto v) {
                MRL_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(
                    &per_cu, reinterpret_cast<const void *>(&mrl_cartpole_rollout<decltype(v)::value>), kBlock, 0));
            });
            MRL_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, gpu_id));
            // (the occupancy query can be one workgroup per CU too high, MI355X_MICROARCH.md "Residency and
            // cooperative launch": keep one per CU in hand near the edge; the cooperative launch is the check)
            const int usable = per_cu > 4 ? per_cu - 1 : per_cu;
            // Above kPersistentMaxWorlds one single-launch step per step is faster than the persistent launch (whose per-step
            // hand-off reads every workgroup's count, flat): tools/cartpole_rollout_probe.py; cartpole.persistent_max moves the limit
            const uint64_t persistent_max = (uint64_t)mrl::debug_get("cartpole.persistent_max", kPersistentMaxWorlds);
            sim->persistent_ok = !mrl::debug_get("cartpole.no_persistent", 0) && (uint64_t)sim->fused_grid <= (uint64_t)usable * (uint64_t)cus &&
                                 (uint64_t)num_worlds <= persistent_max;
            sim->ring = sim->arena.alloc<unsigned long long>((size_t)kRing * sim->fused_grid);
        }
        sim->reseed_shard(0, num_worlds, 0);
        if (mrl::debug_get("inject_scan_timeout", 0)) hipLaunchKernelGGL(mrl::raise_alarm_kernel, dim3(1), dim3(1), 0, 0, sim->alarm.alarm());
        MRL_HIP(hipDeviceSynchronize());
    } catch (...) {
        delete sim;
        throw;
    }
    return sim;
}
