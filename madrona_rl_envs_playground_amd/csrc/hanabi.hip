// 2-player Hanabi world step for gfx950.
//
// Semantics: /root/reference/src/hanabi_env/sim.cpp (drawDeck :45-52, encoders
// :54-365, legal moves :381-444, resetWorld :446-532, removeFromHand :567-594,
// actionSystem :596-792, observationSystem :794-810, checkDone :812-850) and
// rng.hpp:5-40, including the behaviours listed in oracle/hanabi_oracle.c
// (plausibility bits test bit <player-loop-index>; only the player to move is
// re-encoded; hint legality scans all five slots; no legality check).  Actions
// outside an agent's legal-move mask are outside the contract (the reference
// then overruns its own encoders); here they are memory-safe but unspecified.
//
// Mapping.  A world's step is serial (one deck, one player to move), but its
// output is 1.5 KB of 0/1 bytes.  So the work is split inside one wave:
//   phase A  lane = world (kWorldsPerWave lanes active): apply the action on the
//            176-byte game record staged in LDS, then build the observation as a
//            783-bit vector (658 obs bits + 125 own-hand bits) and the 20 legal
//            moves as a bit mask, with a handful of shifted ORs per section;
//   phase B  all 64 lanes: expand bits to bytes, 16 bits -> one 16-byte store, so
//            every row is written with full-width coalesced stores.
// Everything one agent receives is one 896-byte block in HBM (state | legal-move mask, see
// kAgentBlock); the reference-shaped tensors are strided views, and the OBSERVATION tensor is the
// first 658 bytes of the state row: the reference fills the state by copying the observation
// (copyObsToState, sim.cpp:333-341), so it holds the same bytes twice.
//
// Episode indices come from one global counter in the reference
// (sim.cpp:449-451).  As for Cartpole the order is fixed to ascending world
// index, which needs a prefix sum over the done flags:
//   mrl_step              one launch, mrl_hanabi_step_fused: transition, in-kernel prefix
//                         (episode_scan.hpp), re-deal of the finished worlds;
//   mrl_step_phase1 / 2   two launches (sharded batches: the episode base comes from the
//                         other ranks in between): mrl_hanabi_step = action + encode +
//                         score/done + per-workgroup done counts, mrl_hanabi_reset = prefix
//                         over the counts, re-deal, encode both agents.
#include "common.hpp"
#include "episode_scan.hpp"
#include "random_policy.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

constexpr int kWave = 64;
#ifndef MRL_HANABI_WPB
#define MRL_HANABI_WPB 8
#endif
#ifndef MRL_HANABI_WPW
#define MRL_HANABI_WPW 32
#endif
constexpr int kWavesPerBlock = MRL_HANABI_WPB;
constexpr int kBlock = kWave * kWavesPerBlock;
constexpr int kWorldsPerWave = MRL_HANABI_WPW;
constexpr int kWorldsPerBlock = kWorldsPerWave * kWavesPerBlock;
#ifndef MRL_HANABI_EU
#define MRL_HANABI_EU 2
#endif

// 1: the full game's phase A steps and encodes on registers (move_world_full); 0: move_world + the encoder's own read of the record.
// One box, 65536 worlds, us per step 1 / 0: single launch 15.98-16.05 / 17.01-17.07, rollout 9.11-9.21 / 9.29-9.30
// (profiles/r04_ag_hanabi_register_transition_ab.txt), the two-launch pair 22.26-22.32 / 22.95-23.07 (r04_am: two waves per SIMD run phase A
// there, so the scalar instructions it saves issue in each other's shadow); the transition's instruction stream went from
// 590 VALU + 300 SALU + 37 branches + 27 waits on LDS to 630 + 95 + 5 + 5.
#ifndef MRL_HANABI_REG_TRANSITION
#define MRL_HANABI_REG_TRANSITION 1
#endif
constexpr int kHand = 5;
constexpr int kRecordBytes = 176;
constexpr int kRecordWords = kRecordBytes / 4;
constexpr int kRecStride = 180;  // LDS stride: 45 words (odd) -> lane-per-record accesses spread over banks
constexpr int kEncWords = 27;    // 25 words of bits + legal-move mask + spare (odd stride)
// Output rows in HBM: everything one agent of one world receives from a step is ONE 896-byte block
// [state 784 | legal-move mask 80 | pad 32], the two agents of a world back to back: whole 128-byte
// lines, written with full-width wave stores; the exported tensors are strided views into the blocks.
// The OBSERVATION tensor is a view of the state row's first 658 bytes.  The reference builds the
// state as "a copy of the observation, then the own hand" (generateObsState, sim.cpp:367-379:
// copyObsToState copies every entry written so far), so observation[i] == state[i] for every i the
// observation has, always -- also while the ninth information token shifts both, and for the agent
// whose buffers stay stale: both are refreshed together or not at all.  Writing those 658 bytes once
// instead of twice takes 42 % off the bytes of a step (rounds 1-2 wrote [state 784 | observation 672 |
// mask 80] = 1536 bytes per agent).  For configurations whose observation is shorter than 658 entries the
// exported view is that much narrower (the reference declares 658 whatever the configuration and its wrappers
// look at [:obs_size], envs/hanabi_env.py:92-104): the row goes on with the own hand, which is not for the observer.
// (Separate obs / state / mask arrays with 672 / 784 / 80-byte rows were measured in round 1: every
// row end shares a line with the next row, the 80-byte mask rows most of all -- dropping the
// mask stores alone, 5 % of the bytes, took 5 us off a 24 us kernel.)
constexpr int kStateRow = 784, kMaskRow = 80, kPadRow = 32;
constexpr int kAgentBlock = kStateRow + kMaskRow + kPadRow;  // 896
constexpr int kWorldBlock = 2 * kAgentBlock;                 // 1792
constexpr int kStateChunks = kStateRow / 16, kMaskChunks = kMaskRow / 16, kAgentChunks = kAgentBlock / 16;  // 49, 5, 56
static_assert(kAgentBlock % 128 == 0, "agent blocks are whole cache lines");
static_assert(MRL_HANABI_OBS_SIZE <= MRL_HANABI_STATE_SIZE && MRL_HANABI_STATE_SIZE <= kStateRow, "the observation is a prefix of the state row");

// record layout (bytes); identical to oracle/hanabi_oracle.c's dump
enum : int {
    R_DECK = 0, R_DECK_SIZE = 50, R_DISCARD = 51, R_FIREWORKS = 76, R_INFO = 81, R_LIFE = 82, R_CUR = 83,
    R_TURNS = 84, R_SCORE = 85, R_NEWREW = 86,
    R_LM_MOVE = 87, R_LM_PLAYER = 88, R_LM_TARGET = 89, R_LM_INDEX = 90, R_LM_SCORED = 91, R_LM_INFOTOK = 92,
    R_LM_COLOR = 93, R_LM_RANK = 94, R_LM_REVEAL = 95, R_LM_NEWLY = 96, R_LM_DEALTO = 97,
    R_HAND = 100, HAND_BYTES = 36, H_CARDS = 0, H_SIZE = 5, H_KCOLOR = 6, H_KRANK = 11, H_PLAUS = 16,
    R_RNG = 172
};
enum : uint32_t { MV_DISCARD = 0, MV_PLAY = 1, MV_REVEAL_COLOR = 2, MV_REVEAL_RANK = 3, MV_INVALID = 4 };

struct HanabiParams {
    uint32_t num_worlds;
    uint32_t colors, ranks, max_info, max_life;
    // bit offsets of the observation sections
    uint32_t bpc, max_deck;
    uint32_t off_flags, off_deck, off_fireworks, off_info, off_life, off_discard, off_last, off_know;
    uint32_t obs_bits, state_bits;
    uint32_t *records;   // N x 44 words
    uint8_t *rows;       // N x 2 x kAgentBlock (state | mask | pad per agent; the observation is the state row's head)
    int32_t *active;     // 2 x N
    float *reward;       // 2 x N
    int32_t *done;       // N
    const int32_t *actions;  // 2 x N
    uint32_t *block_counts;
    uint32_t *shard_count;  // SHARD_COUNT: finished worlds of the last mrl_step_phase1
    uint32_t chunk;  // worlds per workgroup (multiple of kWorldsPerBlock)
    // device-side random policy (mrl_rollout_random): sample != 0 -> the mover's action is drawn in the kernel
    uint32_t sample, sample_step;
    uint64_t sample_seed;
    int32_t *action_out;  // ACTION tensor, receives the drawn action
    uint32_t deck_words[13];  // a fresh game's record bytes 0..51: the ordered deck (sim.cpp:453-468), its size, discard[0]
#ifdef MRL_DIAG
    unsigned long long *stamps;  // diagnostic build only: per-wave s_memtime stamps (first sub-block)
    uint32_t ablate;             // diagnostic build only: 1 = skip action + encode, 2 = skip the row stores
#endif
};

#ifdef MRL_DIAG
#define STAMP(k)                                                                                                  \
    do {                                                                                                          \
        if (p.stamps && lane == 0 && sub == bid * p.chunk)                                                 \
            p.stamps[(size_t)(bid * kWavesPerBlock + wib) * 16 + (k)] = __builtin_amdgcn_s_memtime();      \
    } while (0)
#define STAMP_REALTIME(k)                                                                                         \
    do {                                                                                                          \
        if (p.stamps && lane == 0 && sub == bid * p.chunk)                                                 \
            p.stamps[(size_t)(bid * kWavesPerBlock + wib) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();  \
    } while (0)
#define ABLATED(bit) (p.ablate & (bit))
// the single-launch step: one sub-block per workgroup, stamps of every wave
#define FSTAMP(k)                                                                                                 \
    do {                                                                                                          \
        if (p.stamps && lane == 0) p.stamps[(size_t)(bid * kWavesPerBlock + wib) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
// the scan wave (ninth wave, owns no world) leaves its stamps in the unused slots 8.. of the workgroup's first wave
#define FSTAMP_SCAN(k)                                                                                            \
    do {                                                                                                          \
        if (p.stamps && lane == 0) p.stamps[(size_t)(bid * kWavesPerBlock) * 16 + 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
// the persistent rollout: its last step but one (row = wave 0..7, the scan wave in slots 8.. of wave 0)
#define RSTAMP(row, slot)                                                                                         \
    do {                                                                                                          \
        if (p.stamps && lane == 0 && k + 2u == num_steps) p.stamps[(size_t)(b * kWavesPerBlock + (row)) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define RSTAMP(row, slot) ((void)0)
#define FSTAMP_SCAN(k) ((void)0)
#define FSTAMP(k) ((void)0)
#define STAMP(k) ((void)0)
#define STAMP_REALTIME(k) ((void)0)
#define ABLATED(bit) false
#endif

__device__ __forceinline__ void wave_lds_sync()
{
    // Cross-lane hand-off through LDS inside ONE wave: DS instructions of a wave execute in issue
    // order, so only the compiler has to keep the order.  (A wavefront-scope release/acquire
    // fence is lowered with s_waitcnt vmcnt(0): the wave would sit out every row store of the
    // expansion before storing its records.)
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ uint32_t seed_of(uint32_t episode)
{
    uint32_t v0 = episode, v1 = 0, sum = 0;
#pragma unroll
    for (int round = 0; round < 8; round++) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

__device__ __forceinline__ uint32_t &rng_of(uint8_t *rec) { return *reinterpret_cast<uint32_t *>(rec + R_RNG); }
__device__ __forceinline__ uint32_t *plaus_of(uint8_t *hand) { return reinterpret_cast<uint32_t *>(hand + H_PLAUS); }

// sim.cpp:45-52 (one float multiply, truncation)
__device__ __forceinline__ uint32_t draw(uint8_t *rec)
{
    uint32_t &g = rng_of(rec);
    g = 1664525u * g + 1013904223u;
    const float r = (float)(g & 0x00FFFFFFu) / (float)0x01000000;
    const uint32_t size = rec[R_DECK_SIZE];
    const int32_t at = (int32_t)((float)size * r);
    const uint8_t card = rec[R_DECK + at];
    rec[R_DECK + at] = rec[R_DECK + size - 1];
    rec[R_DECK_SIZE] = (uint8_t)(size - 1);
    return card;
}

// OR `nbits` bits into the bit vector at bit offset `off`.  ds_or_b32 without return: no LDS
// read latency in the dependency chain (a read-modify-write through registers costs one LDS
// round trip per put, ~50 puts per encode).  The second word is OR-ed unconditionally (with
// zero when the field does not cross a word), so there is no branch either.
__device__ __forceinline__ void put(uint32_t *enc, uint32_t off, uint32_t nbits, uint32_t value)
{
    const uint32_t w = off >> 5, s = off & 31;
    const uint32_t lo = value << s, hi = s ? value >> (32u - s) : 0u;
    const uint32_t addr = (uint32_t)reinterpret_cast<uintptr_t>(enc + w);  // low 32 bits of a shared pointer = LDS offset
    // no "memory" clobber: the bit vector is only ever touched by these ORs between the two compiler
    // barriers in encode_agent, so the record reads around a put stay free to be scheduled early
    asm volatile("ds_or_b32 %0, %1\n\tds_or_b32 %0, %2 offset:4" : : "v"(addr), "v"(lo), "v"(hi));
    (void)nbits;
}

__device__ __forceinline__ uint32_t ones(uint32_t n) { return n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u; }

// sim.cpp:367-379 as a bit vector: bits [0, obs_bits) = observation of `agent`,
// bits [obs_bits, state_bits) = its own hand; enc[25] = legal moves (sim.cpp:381-444)
// kR: compile-time rank count (5 in every configuration the reference defines, envs/hanabi_env.py:16-58)
// so that card / R and card % R are multiply-shifts instead of runtime divisions; 0 = use p.ranks.
template <int kR>
__device__ void encode_agent(const HanabiParams &p, uint8_t *rec, uint32_t *enc, uint32_t agent)
{
    const uint32_t K = p.colors, R = kR ? (uint32_t)kR : p.ranks, bpc = p.bpc;
    for (int w = 0; w < kEncWords; w++) enc[w] = 0;
    asm volatile("" ::: "memory");  // zero-fill is issued before the first OR (DS ops of a wave execute in order)
    uint8_t *own = rec + R_HAND + HAND_BYTES * agent;
    uint8_t *other = rec + R_HAND + HAND_BYTES * (agent ^ 1u);
    const uint32_t own_size = own[H_SIZE], other_size = other[H_SIZE];

    // partner hand + "short hand" flags (:54-90)
    for (uint32_t c = 0; c < kHand; c++)
        if (c < other_size) put(enc, c * bpc + other[H_CARDS + c], 1, 1);
    put(enc, p.off_flags, 2, (own_size < kHand ? 1u : 0u) | (other_size < kHand ? 2u : 0u));

    // board (:92-135)
    const uint32_t deck = min((uint32_t)rec[R_DECK_SIZE], p.max_deck);
    put(enc, p.off_deck, min(deck, 32u), ones(min(deck, 32u)));
    if (deck > 32) put(enc, p.off_deck + 32, deck - 32, ones(deck - 32));
    uint32_t fw = 0;
    for (uint32_t c = 0; c < K; c++) {
        const uint32_t f = rec[R_FIREWORKS + c];
        if (f >= 1 && f <= R) fw |= 1u << (c * R + f - 1);
    }
    put(enc, p.off_fireworks, K * R, fw);
    // The reference writes with a running offset, so information tokens beyond the
    // maximum (a rank-5 card played at full tokens adds one unconditionally,
    // sim.cpp:676-678) lengthen this thermometer and move every later section up.
    const uint32_t info_now = min((uint32_t)rec[R_INFO], p.max_info + 5u);
    const uint32_t excess = info_now > p.max_info ? info_now - p.max_info : 0u;
    put(enc, p.off_info, p.max_info + excess, ones(info_now));
    put(enc, p.off_life + excess, p.max_life, ones(min((uint32_t)rec[R_LIFE], p.max_life)));
    enc[26] = excess;

    // discards (:137-156): per colour, thermometers of 3 / 2.. / 1 copies
    for (uint32_t c = 0; c < K; c++) {
        uint32_t bits = 0, at = 0;
        for (uint32_t r = 0; r < R; r++) {
            const uint32_t copies = 2u + (r == 0 ? 1u : 0u) - (r == R - 1 ? 1u : 0u);  // 3, 2.., 1
            bits |= ones(min((uint32_t)rec[R_DISCARD + c * R + r], copies)) << at;
            at += copies;
        }
        put(enc, p.off_discard + excess + c * 2 * R, 2 * R, bits);
    }

    // last action (:158-289)
    {
        const uint32_t move = rec[R_LM_MOVE];
        const int32_t lm_player = (int8_t)rec[R_LM_PLAYER];
        const bool hint = move == MV_REVEAL_COLOR || move == MV_REVEAL_RANK;
        const bool card = move == MV_PLAY || move == MV_DISCARD;
        uint32_t v = 0, at = 0;
        if (lm_player != -1) v |= 1u << (((int32_t)agent - lm_player + 2) & 1);
        at += 2;
        // one-hot over (play, discard, reveal colour, reveal rank); packed table indexed by MoveType
        v |= move < 4 ? (1u << (at + ((0x03020001u >> (8u * move)) & 0xFFu))) : 0u;
        at += 4;
        if (hint) v |= 1u << (at + (((int32_t)agent - (int32_t)(int8_t)rec[R_LM_TARGET] + 2) & 1));
        at += 2;
        if (move == MV_REVEAL_COLOR && (uint32_t)rec[R_LM_COLOR] < K) v |= 1u << (at + rec[R_LM_COLOR]);
        at += K;
        if (move == MV_REVEAL_RANK && (uint32_t)rec[R_LM_RANK] < R) v |= 1u << (at + rec[R_LM_RANK]);
        at += R;
        const uint32_t off_last = p.off_last + excess;
        put(enc, off_last, at, v);
        uint32_t v2 = 0;
        if (hint) v2 |= rec[R_LM_REVEAL] & 31u;
        if (card && (uint32_t)rec[R_LM_INDEX] < kHand) v2 |= 1u << (kHand + rec[R_LM_INDEX]);
        put(enc, off_last + at, 2 * kHand, v2);
        if (card) {
            const uint32_t id = (uint32_t)rec[R_LM_COLOR] * R + rec[R_LM_RANK];
            if (id < bpc) put(enc, off_last + at + 2 * kHand + id, 1, 1);
        }
        if (move == MV_PLAY)
            put(enc, off_last + at + 2 * kHand + bpc, 2, (rec[R_LM_SCORED] ? 1u : 0u) | (rec[R_LM_INFOTOK] ? 2u : 0u));
    }

    // card knowledge (:291-331): own hand first, then the partner's
    for (uint32_t i = 0; i < 2; i++) {
        uint8_t *h = i == 0 ? own : other;
        const uint32_t size = h[H_SIZE];
        const uint32_t *plaus = plaus_of(h);
        for (uint32_t c = 0; c < kHand; c++) {
            if (c >= size) continue;
            const uint32_t base = p.off_know + excess + (i * kHand + c) * (bpc + K + R);
            if ((plaus[c] >> i) & 1u) put(enc, base, bpc, ones(bpc));  // sim.cpp:311: bit <i>, not bit <v>
            uint32_t kr = 0;
            const int32_t kc = (int8_t)h[H_KCOLOR + c], kk = (int8_t)h[H_KRANK + c];
            if (kc >= 0 && (uint32_t)kc < K) kr |= 1u << kc;
            if (kk >= 0 && (uint32_t)kk < R) kr |= 1u << (K + kk);
            put(enc, base + bpc, K + R, kr);
        }
    }

    // state tail: own hand (:343-365)
    for (uint32_t c = 0; c < kHand; c++)
        if (c < own_size) put(enc, p.obs_bits + excess + c * bpc + own[H_CARDS + c], 1, 1);

    // legal moves (:381-444)
    uint32_t legal = 0;
    const uint32_t info = rec[R_INFO];
    for (uint32_t i = 0; i < kHand; i++) {
        if (i < own_size && info < p.max_info) legal |= 1u << i;
        if (i < own_size) legal |= 1u << (kHand + i);
    }
    if (info > 0) {
        for (uint32_t n = 0; n < kHand; n++) {  // all five slots, whatever the hand size (:416-417)
            const uint32_t card = other[H_CARDS + n];
            const uint32_t col = card / R, rk = card % R;
            if (col < K) legal |= 1u << (2 * kHand + col);
            legal |= 1u << (2 * kHand + K + rk);
        }
    }
    asm volatile("" ::: "memory");
    enc[25] = legal & 0xFFFFFu;
}

// ---------------------------------------------------------------------------------------------
// The same encoding for the full game (5 colours, 5 ranks, 8 information and 3 life tokens --
// envs/hanabi_env.py:16-27, the configuration every benchmark uses), with every section offset a
// compile-time constant: the record is read from LDS once (44 words, one wait), the 783 bits are
// assembled in registers with shifts at constant positions, and 27 words go back to LDS.  The
// generic encoder above issues ~50 read-modify-write puts with runtime offsets and ~56 waits on
// LDS reads; measured per wave (16 worlds, four waves per SIMD) 5.7 us for it against 2.0 us for this one.
//
// Sections after the information tokens move up by `excess` (see above); they are assembled
// relative to bit 200 and merged with one funnel shift per word.
// ---------------------------------------------------------------------------------------------
namespace full_game {
constexpr uint32_t kBpc = 25, kK = 5, kRk = 5;
constexpr uint32_t kOffFlags = 125, kOffDeck = 127, kOffFireworks = 167, kOffInfo = 192, kOffLife = 200;
// relative to kOffLife (+ excess)
constexpr uint32_t kRelDiscard = 3, kRelLast = 53, kRelKnow = 108, kRelOwnHand = 458, kRelEnd = 583;
static_assert(kOffLife + kRelKnow + 2 * kHand * (kBpc + kK + kRk) == MRL_HANABI_OBS_SIZE, "observation layout");
static_assert(kOffLife + kRelEnd == MRL_HANABI_STATE_SIZE, "state layout");
constexpr int kLoWords = 7, kHiWords = 19;

// OR the low `width` bits of v into a register-resident bit vector at a position that is a
// constant after unrolling (so w[...] stays in registers)
template <int NW>
__device__ __forceinline__ void orbits(uint32_t (&w)[NW], uint32_t off, uint32_t width, uint32_t v)
{
    const uint32_t i = off >> 5, s = off & 31u;
    w[i] |= v << s;
    if (s + width > 32u) w[i + 1] |= v >> (32u - s);
}
__device__ __forceinline__ uint32_t byte_at(const uint32_t (&r)[kRecordWords], uint32_t byte) { return (r[byte >> 2] >> ((byte & 3u) * 8u)) & 0xFFu; }
}  // namespace full_game

// kFresh: the record is a game that deal_new_game has just written (full hands, 40 cards in the deck, all
// tokens, nothing played, discarded or hinted) -- only the ten dealt cards are read, the rest folds to
// constants.  The re-deal launch is a serial chain on a few lanes per wave, so its length is what counts.
// `early(stage)`: the single-launch step lets another wave store whole 128-byte lines of the row while this one goes on
// encoding.  Stage 1: words 0..3 of the bit vector are in LDS (bits 0..127: the partner's hand, the short-hand flags, the
// first bit of the deck -- line 0).  Stage 2: words 12..19 are (bits 384..639, card knowledge only whatever the shift of
// 0..5 bits: lines 3 and 4); with a hand-off to make, the knowledge section is therefore encoded before the discards and
// the last action, which it does not depend on.
struct NoEarlyLine {
    static constexpr bool kActive = false;
    __device__ __forceinline__ void operator()(uint32_t) const {}
};
struct EarlyLineFlag {  // raises a flag in LDS behind the words (LDS operations of a wave complete in order)
    static constexpr bool kActive = true;
    uint32_t *flag;
    uint32_t lane;
#ifdef MRL_DIAG
    unsigned long long *stamps;  // the wave's 16 stamp slots (diagnostic build): 10 + stage = when the stage was handed over
#endif
    __device__ __forceinline__ void operator()(uint32_t stage) const
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(flag, stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef MRL_DIAG
        if (stamps && lane == 0) stamps[10 + stage] = __builtin_amdgcn_s_memrealtime();
#endif
    }
};
template <bool kFresh, typename EarlyLine = NoEarlyLine>
__device__ __forceinline__ void encode_record_full_t(const uint32_t (&r)[kRecordWords], uint32_t *enc, uint32_t agent, const EarlyLine &early = EarlyLine())
{
    // (r: the record's 44 words; words 0..11 -- the deck's cards -- and 43 -- the generator -- are not looked at)
    using namespace full_game;
    // hands: own = the encoded agent's, other = the partner's (9 words each)
    uint32_t own[9], other[9];
#pragma unroll
    for (int w = 0; w < 9; w++) {
        const uint32_t h0 = r[R_HAND / 4 + w], h1 = r[(R_HAND + HAND_BYTES) / 4 + w];
        own[w] = agent ? h1 : h0;
        other[w] = agent ? h0 : h1;
    }
    // bytes of a hand: cards 0..4, size 5, known colour 6..10, known rank 11..15; plausibility words 4..8
    auto hbyte = [](const uint32_t (&h)[9], uint32_t b) { return (h[b >> 2] >> ((b & 3u) * 8u)) & 0xFFu; };
    const uint32_t own_size = kFresh ? (uint32_t)kHand : hbyte(own, H_SIZE), other_size = kFresh ? (uint32_t)kHand : hbyte(other, H_SIZE);

    uint32_t lo[kLoWords], hi[kHiWords];
#pragma unroll
    for (int w = 0; w < kLoWords; w++) lo[w] = 0;
#pragma unroll
    for (int w = 0; w < kHiWords; w++) hi[w] = 0;

    // partner hand + "short hand" flags (:54-90)
#pragma unroll
    for (uint32_t c = 0; c < kHand; c++) orbits(lo, c * kBpc, kBpc, c < other_size ? 1u << (hbyte(other, H_CARDS + c) & 31u) : 0u);
    orbits(lo, kOffFlags, 2, (own_size < kHand ? 1u : 0u) | (other_size < kHand ? 2u : 0u));
    // board (:92-135)
    const uint32_t deck = kFresh ? 40u : min(byte_at(r, R_DECK_SIZE), 40u);
    orbits(lo, kOffDeck, 32, ones(min(deck, 32u)));
    orbits(lo, kOffDeck + 32, 8, deck > 32u ? ones(deck - 32u) : 0u);
    uint32_t fw = 0;
#pragma unroll
    for (uint32_t c = 0; c < kK; c++) {
        const uint32_t f = kFresh ? 0u : byte_at(r, R_FIREWORKS + c);
        fw |= (f >= 1 && f <= kRk) ? 1u << (c * kRk + f - 1) : 0u;
    }
    orbits(lo, kOffFireworks, 25, fw);
    const uint32_t info = kFresh ? 8u : byte_at(r, R_INFO);
    const uint32_t info_now = min(info, 13u);
    const uint32_t excess = info_now > 8u ? info_now - 8u : 0u;
    orbits(lo, kOffInfo, 13, ones(info_now));
    if constexpr (EarlyLine::kActive) {
#pragma unroll
        for (int w = 0; w < 4; w++) enc[w] = lo[w];
        early(1u);
    }
    const uint32_t sh = 8u + excess;  // 8..13: the relative part goes up by 200 + excess = 6 words + sh bits

    // everything below is placed relative to bit 200 + excess
    orbits(hi, 0, 3, ones(kFresh ? 3u : min(byte_at(r, R_LIFE), 3u)));
    auto discards_and_last_action = [&]() {
    // discards (:137-156)
#pragma unroll
    for (uint32_t c = 0; c < (kFresh ? 0u : kK); c++) {
        uint32_t bits = 0;
        bits |= ones(min(byte_at(r, R_DISCARD + c * kRk + 0), 3u));
        bits |= ones(min(byte_at(r, R_DISCARD + c * kRk + 1), 2u)) << 3;
        bits |= ones(min(byte_at(r, R_DISCARD + c * kRk + 2), 2u)) << 5;
        bits |= ones(min(byte_at(r, R_DISCARD + c * kRk + 3), 2u)) << 7;
        bits |= ones(min(byte_at(r, R_DISCARD + c * kRk + 4), 1u)) << 9;
        orbits(hi, kRelDiscard + c * 2 * kRk, 10, bits);
    }
    // last action (:158-289); a fresh game has none (MV_INVALID, player -1: no bit set)
    if constexpr (!kFresh) {
        const uint32_t move = byte_at(r, R_LM_MOVE);
        const int32_t lm_player = (int8_t)byte_at(r, R_LM_PLAYER);
        const uint32_t lm_color = byte_at(r, R_LM_COLOR), lm_rank = byte_at(r, R_LM_RANK), lm_index = byte_at(r, R_LM_INDEX);
        const bool hint = move == MV_REVEAL_COLOR || move == MV_REVEAL_RANK;
        const bool card = move == MV_PLAY || move == MV_DISCARD;
        uint32_t v = 0;
        v |= lm_player != -1 ? 1u << (((int32_t)agent - lm_player + 2) & 1) : 0u;
        v |= move < 4 ? (1u << (2u + ((0x03020001u >> (8u * move)) & 0xFFu))) : 0u;
        v |= hint ? 1u << (6 + (((int32_t)agent - (int32_t)(int8_t)byte_at(r, R_LM_TARGET) + 2) & 1)) : 0u;
        v |= (move == MV_REVEAL_COLOR && lm_color < kK) ? 1u << (8u + lm_color) : 0u;
        v |= (move == MV_REVEAL_RANK && lm_rank < kRk) ? 1u << (13u + lm_rank) : 0u;
        orbits(hi, kRelLast, 18, v);
        uint32_t v2 = hint ? byte_at(r, R_LM_REVEAL) & 31u : 0u;
        v2 |= (card && lm_index < kHand) ? 1u << (kHand + lm_index) : 0u;
        orbits(hi, kRelLast + 18, 10, v2);
        const uint32_t id = lm_color * kRk + lm_rank;
        orbits(hi, kRelLast + 28, 25, (card && id < kBpc) ? 1u << id : 0u);
        orbits(hi, kRelLast + 53, 2,
               move == MV_PLAY ? (byte_at(r, R_LM_SCORED) ? 1u : 0u) | (byte_at(r, R_LM_INFOTOK) ? 2u : 0u) : 0u);
    }
    };
    if constexpr (!EarlyLine::kActive) discards_and_last_action();
    // card knowledge (:291-331): own hand first, then the partner's
#pragma unroll
    for (uint32_t i = 0; i < 2; i++) {
#pragma unroll
        for (uint32_t c = 0; c < kHand; c++) {
            const uint32_t size = i == 0 ? own_size : other_size;
            const uint32_t plaus = kFresh ? ~0u : (i == 0 ? own[4 + c] : other[4 + c]);
            const uint32_t kcb = kFresh ? 0xFFu : (i == 0 ? hbyte(own, H_KCOLOR + c) : hbyte(other, H_KCOLOR + c));
            const uint32_t kkb = kFresh ? 0xFFu : (i == 0 ? hbyte(own, H_KRANK + c) : hbyte(other, H_KRANK + c));
            const bool have = c < size;
            const uint32_t base = kRelKnow + (i * kHand + c) * (kBpc + kK + kRk);
            orbits(hi, base, kBpc, (have && ((plaus >> i) & 1u)) ? ones(kBpc) : 0u);  // sim.cpp:311: bit <i>, not bit <v>
            uint32_t kr = 0;
            kr |= kcb < kK ? 1u << (kcb & 31u) : 0u;  // 0xFF = unknown
            kr |= kkb < kRk ? 1u << ((kK + kkb) & 31u) : 0u;
            orbits(hi, base + kBpc, kK + kRk, have ? kr : 0u);
        }
    }
    if constexpr (EarlyLine::kActive) {
        static_assert(5u * 32u >= kRelKnow && 14u * 32u <= kRelOwnHand, "hi[5..13], which words 12..19 are made of, hold card knowledge only");
#pragma unroll
        for (int w = 6; w < 14; w++) enc[6 + w] = __builtin_amdgcn_alignbit(hi[w], hi[w - 1], 32u - sh);
        early(2u);
        discards_and_last_action();
    }
    // state tail: own hand (:343-365)
#pragma unroll
    for (uint32_t c = 0; c < kHand; c++) orbits(hi, kRelOwnHand + c * kBpc, kBpc, c < own_size ? 1u << (hbyte(own, H_CARDS + c) & 31u) : 0u);

    // legal moves (:381-444)
    uint32_t legal = 0;
#pragma unroll
    for (uint32_t i = 0; i < kHand; i++) {
        legal |= (i < own_size && info < 8u) ? 1u << i : 0u;
        legal |= i < own_size ? 1u << (kHand + i) : 0u;
    }
    uint32_t hints = 0;
#pragma unroll
    for (uint32_t n = 0; n < kHand; n++) {  // all five slots, whatever the hand size (:416-417)
        const uint32_t cardv = hbyte(other, H_CARDS + n);
        const uint32_t col = (cardv * 205u) >> 10, rk = cardv - col * 5u;  // exact for cardv < 1024
        hints |= col < kK ? 1u << (2 * kHand + col) : 0u;
        hints |= 1u << (2 * kHand + kK + rk);
    }
    legal |= info > 0 ? hints : 0u;

    // merge: words 0..6 from lo, the relative part shifted up by 200 + excess = 6 words + (8 + excess) bits
    uint32_t out[25];
#pragma unroll
    for (int w = 0; w < 6; w++) out[w] = lo[w];
    out[6] = lo[6] | (hi[0] << sh);
#pragma unroll
    for (int w = 1; w < kHiWords; w++) out[6 + w] = __builtin_amdgcn_alignbit(hi[w], hi[w - 1], 32u - sh);
#pragma unroll
    for (int w = 0; w < 25; w++)
        if (!EarlyLine::kActive || (w >= 4 && !(w >= 12 && w < 20))) enc[w] = out[w];
    enc[25] = legal & 0xFFFFFu;
    enc[26] = excess;
}

template <bool kFresh, typename EarlyLine = NoEarlyLine>
__device__ __forceinline__ void encode_agent_full_t(const uint8_t *rec, uint32_t *enc, uint32_t agent, const EarlyLine &early = EarlyLine())
{
    uint32_t r[kRecordWords];
#pragma unroll
    for (int w = 0; w < kRecordWords; w++) r[w] = reinterpret_cast<const uint32_t *>(rec)[w];
    encode_record_full_t<kFresh, EarlyLine>(r, enc, agent, early);
}
__device__ void encode_agent_full(const uint8_t *rec, uint32_t *enc, uint32_t agent) { encode_agent_full_t<false>(rec, enc, agent); }
__device__ void encode_fresh_full(const uint8_t *rec, uint32_t *enc, uint32_t agent) { encode_agent_full_t<true>(rec, enc, agent); }

// kV selects the code variant: 0 = any configuration (runtime rank count), 1 = five ranks,
// 2 = the full game (encode_agent_full)
template <int kV>
__device__ __forceinline__ void encode_variant(const HanabiParams &p, uint8_t *rec, uint32_t *enc, uint32_t agent)
{
    if constexpr (kV == 2)
        encode_agent_full(rec, enc, agent);
    else
        encode_agent<(kV ? 5 : 0)>(p, rec, enc, agent);
}

// encode of a game straight out of deal_new_game
template <int kV>
__device__ __forceinline__ void encode_fresh(const HanabiParams &p, uint8_t *rec, uint32_t *enc, uint32_t agent)
{
    if constexpr (kV == 2)
        encode_fresh_full(rec, enc, agent);
    else
        encode_agent<(kV ? 5 : 0)>(p, rec, enc, agent);
}

// sim.cpp:567-594
__device__ __forceinline__ void take_from_hand(const HanabiParams &p, uint8_t *rec, uint8_t *hand, uint32_t index)
{
    uint32_t *plaus = plaus_of(hand);
    if (rec[R_DECK_SIZE] == 0) {
        const uint32_t size = hand[H_SIZE];
        for (uint32_t i = index + 1; i < size && i < kHand; i++) {
            hand[H_CARDS + i - 1] = hand[H_CARDS + i];
            plaus[i - 1] = plaus[i];
            hand[H_KCOLOR + i - 1] = hand[H_KCOLOR + i];
            hand[H_KRANK + i - 1] = hand[H_KRANK + i];
        }
        hand[H_SIZE] = (uint8_t)(size - 1);
    } else {
        hand[H_CARDS + index] = (uint8_t)draw(rec);
        plaus[index] = ones(p.bpc);
        hand[H_KCOLOR + index] = 0xFF;
        hand[H_KRANK + index] = 0xFF;
    }
}

// generateActionMask (sim.cpp:381-444) for `agent`, as 20 bits
template <int kR>
__device__ __forceinline__ uint32_t legal_moves(const HanabiParams &p, const uint8_t *rec, uint32_t agent)
{
    const uint32_t K = p.colors, R = kR ? (uint32_t)kR : p.ranks;
    const uint8_t *own = rec + R_HAND + HAND_BYTES * agent;
    const uint8_t *other = rec + R_HAND + HAND_BYTES * (agent ^ 1u);
    const uint32_t own_size = own[H_SIZE], info = rec[R_INFO];
    uint32_t legal = 0;
    for (uint32_t i = 0; i < kHand; i++) {
        if (i < own_size && info < p.max_info) legal |= 1u << i;
        if (i < own_size) legal |= 1u << (kHand + i);
    }
    if (info > 0) {
        for (uint32_t n = 0; n < kHand; n++) {  // all five slots, whatever the hand size (:416-417)
            const uint32_t card = other[H_CARDS + n];
            const uint32_t col = card / R, rk = card % R;
            if (col < K) legal |= 1u << (2 * kHand + col);
            legal |= 1u << (2 * kHand + K + rk);
        }
    }
    return legal & 0xFFFFFu;
}

// sim.cpp:596-792
template <int kR>
__device__ void apply_action(const HanabiParams &p, uint8_t *rec, uint32_t uid)
{
    const uint32_t K = p.colors, R = kR ? (uint32_t)kR : p.ranks;
    if (rec[R_DECK_SIZE] == 0) rec[R_TURNS] = (uint8_t)(rec[R_TURNS] - 1);
    const uint32_t actor = rec[R_CUR] & 1u;
    uint8_t *hand = rec + R_HAND + HAND_BYTES * actor;

    rec[R_LM_PLAYER] = (uint8_t)actor;
    rec[R_LM_TARGET] = 0xFF;
    rec[R_LM_INDEX] = 0xFF;
    rec[R_LM_SCORED] = 0;
    rec[R_LM_INFOTOK] = 0;
    rec[R_LM_COLOR] = 0xFF;
    rec[R_LM_RANK] = 0xFF;
    rec[R_LM_REVEAL] = 0;
    rec[R_LM_NEWLY] = 0;
    rec[R_LM_DEALTO] = 0xFF;
    rec[R_CUR] = (uint8_t)(actor ^ 1u);

    if (uid < 2 * kHand) {
        const bool play = uid >= kHand;
        const uint32_t slot = play ? uid - kHand : uid;
        const uint32_t card = min((uint32_t)hand[H_CARDS + slot], 24u);
        const uint32_t col = card / R, rk = card % R;
        rec[R_LM_MOVE] = play ? MV_PLAY : MV_DISCARD;
        rec[R_LM_INDEX] = (uint8_t)slot;
        rec[R_LM_COLOR] = (uint8_t)col;
        rec[R_LM_RANK] = (uint8_t)rk;
        if (!play) {
            rec[R_DISCARD + card]++;
            rec[R_INFO]++;
        } else if (col < 5 && rec[R_FIREWORKS + col] == rk) {
            rec[R_FIREWORKS + col]++;
            if (rec[R_FIREWORKS + col] == R) {
                rec[R_INFO]++;
                rec[R_LM_INFOTOK] = 1;
            }
            rec[R_LM_SCORED] = 1;
        } else {
            rec[R_DISCARD + card]++;
            rec[R_LIFE]--;
        }
        take_from_hand(p, rec, hand, slot);
        return;
    }
    uid -= 2 * kHand;
    uint8_t *ph = rec + R_HAND + HAND_BYTES * (actor ^ 1u);  // one partner: target_offset is always 1
    uint32_t *plaus = plaus_of(ph);
    const uint32_t psize = min((uint32_t)ph[H_SIZE], (uint32_t)kHand);
    rec[R_INFO]--;
    rec[R_LM_TARGET] = (uint8_t)(actor ^ 1u);
    uint32_t reveal = 0, newly = 0, hint = 0;
    if (uid < K) {
        const uint32_t col = uid;
        rec[R_LM_MOVE] = MV_REVEAL_COLOR;
        rec[R_LM_COLOR] = (uint8_t)col;
        for (uint32_t i = 0; i < R; i++) hint |= 1u << (col * R + i);
        for (uint32_t i = 0; i < psize; i++) {
            if (ph[H_CARDS + i] / R == col) {
                reveal |= 1u << i;
                if ((int8_t)ph[H_KCOLOR + i] == -1) newly |= 1u << i;
                ph[H_KCOLOR + i] = (uint8_t)col;
                plaus[i] &= hint;
            } else {
                plaus[i] &= ~hint;
            }
        }
    } else {
        const uint32_t rk = (uid - K) % R;
        rec[R_LM_MOVE] = MV_REVEAL_RANK;
        rec[R_LM_RANK] = (uint8_t)rk;
        for (uint32_t i = 0; i < R; i++) hint |= 1u << (i * R + rk);
        for (uint32_t i = 0; i < psize; i++) {
            if (ph[H_CARDS + i] % R == rk) {
                reveal |= 1u << i;
                if ((int8_t)ph[H_KCOLOR + i] == -1) newly |= 1u << i;  // sim.cpp:776 tests known_color here too
                ph[H_KRANK + i] = (uint8_t)rk;
                plaus[i] &= hint;
            } else {
                plaus[i] &= ~hint;
            }
        }
    }
    rec[R_LM_REVEAL] = (uint8_t)reveal;
    rec[R_LM_NEWLY] = (uint8_t)newly;
}

// actionSystem (sim.cpp:596-792) for the full game on registers: the header words and both hands
// are read from LDS once, every move kind is straight-line selects, and the changed words go back
// once.  Only the discard pile and the deck are touched in place (indexed by a card value / a random
// position).  The branchy version above has shorter per-kind paths but ~50 LDS waits.  Measured: no
// difference at 16 worlds per wave / four waves per SIMD (2.8 us per wave either way), 31.3 against
// 31.7 us per launch at 32 worlds per wave / two waves per SIMD.
__device__ void apply_action_full(uint8_t *rec, uint32_t uid)
{
    constexpr uint32_t kRk = 5, kAll = (1u << 25) - 1u;
    uint32_t *rec32 = reinterpret_cast<uint32_t *>(rec);
    const uint32_t w19 = rec32[19], w20 = rec32[20], w21 = rec32[21];
    const uint32_t deck_size = rec[R_DECK_SIZE];
    const uint32_t actor = (w20 >> 24) & 1u;
    uint32_t *ah = rec32 + (R_HAND + HAND_BYTES * actor) / 4;          // mover's hand
    uint32_t *ph = rec32 + (R_HAND + HAND_BYTES * (actor ^ 1u)) / 4;   // partner's hand
    uint32_t a[9], q[9];
#pragma unroll
    for (int w = 0; w < 9; w++) {
        a[w] = ah[w];
        q[w] = ph[w];
    }
    auto hbyte = [](const uint32_t (&h)[9], uint32_t b) { return (h[b >> 2] >> ((b & 3u) * 8u)) & 0xFFu; };

    uint32_t turns = w21 & 0xFFu;
    if (deck_size == 0) turns = (turns - 1u) & 0xFFu;
    uint32_t info = (w20 >> 8) & 0xFFu, life = (w20 >> 16) & 0xFFu;
    const unsigned long long fw = (unsigned long long)w19 | ((unsigned long long)(w20 & 0xFFu) << 32);

    const bool is_card = uid < 2 * kHand;
    const bool play = is_card && uid >= kHand;
    const uint32_t slot = is_card ? (play ? uid - kHand : uid) : 0u;

    // ---- discard / play (:634-691) ----
    uint32_t card_of[kHand], kc_of[kHand], kk_of[kHand];
#pragma unroll
    for (uint32_t i = 0; i < kHand; i++) {
        card_of[i] = hbyte(a, H_CARDS + i);
        kc_of[i] = hbyte(a, H_KCOLOR + i);
        kk_of[i] = hbyte(a, H_KRANK + i);
    }
    uint32_t chosen = 0;
#pragma unroll
    for (uint32_t i = 0; i < kHand; i++) chosen = slot == i ? card_of[i] : chosen;
    const uint32_t card = min(chosen, 24u);
    const uint32_t col = (card * 205u) >> 10, rk = card - col * kRk;
    const uint32_t top = (uint32_t)(fw >> (8u * col)) & 0xFFu;
    const bool success = play && top == rk;
    const bool completed = success && top + 1u == kRk;
    if (is_card && !success) rec[R_DISCARD + card]++;   // a discard, or a failed play
    info += (is_card && !play) ? 1u : 0u;
    info += completed ? 1u : 0u;
    life -= (play && !success) ? 1u : 0u;
    const unsigned long long fw_new = fw + (success ? 1ull << (8u * col) : 0ull);

    // removeFromHand (:567-594): redraw into the slot, or shift left when the deck is empty
    uint32_t size_a = hbyte(a, H_SIZE);
    uint32_t pl_of[kHand];
#pragma unroll
    for (uint32_t i = 0; i < kHand; i++) pl_of[i] = a[4 + i];
    if (is_card) {
        if (deck_size == 0) {
#pragma unroll
            for (uint32_t i = 0; i + 1 < kHand; i++) {
                const bool take_next = i >= slot && i + 1 < size_a;
                card_of[i] = take_next ? card_of[i + 1] : card_of[i];
                kc_of[i] = take_next ? kc_of[i + 1] : kc_of[i];
                kk_of[i] = take_next ? kk_of[i + 1] : kk_of[i];
                pl_of[i] = take_next ? pl_of[i + 1] : pl_of[i];
            }
            size_a = (size_a - 1u) & 0xFFu;
        } else {
            const uint32_t drawn = draw(rec);
#pragma unroll
            for (uint32_t i = 0; i < kHand; i++) {
                const bool here = i == slot;
                card_of[i] = here ? drawn : card_of[i];
                kc_of[i] = here ? 0xFFu : kc_of[i];
                kk_of[i] = here ? 0xFFu : kk_of[i];
                pl_of[i] = here ? kAll : pl_of[i];
            }
        }
    }

    // ---- hints (:695-788) ----
    const bool hint_move = !is_card;
    const uint32_t u = uid - 2 * kHand;
    const bool by_color = u < 5u;
    const uint32_t u_rank = u - 5u;
    const uint32_t val = by_color ? u : u_rank - ((u_rank * 205u) >> 10) * kRk;  // (uid - K) % R
    const uint32_t hint = by_color ? 0x1Fu << (kRk * val) : 0x108421u << val;     // the 5 cards of a colour / of a rank
    const uint32_t psize = min(hbyte(q, H_SIZE), (uint32_t)kHand);
    uint32_t reveal = 0, newly = 0;
    uint32_t qkc[kHand], qkk[kHand], qpl[kHand];
#pragma unroll
    for (uint32_t i = 0; i < kHand; i++) {
        const uint32_t c = hbyte(q, H_CARDS + i);
        const uint32_t ccol = (c * 205u) >> 10, crk = c - ccol * kRk;
        qkc[i] = hbyte(q, H_KCOLOR + i);
        qkk[i] = hbyte(q, H_KRANK + i);
        qpl[i] = q[4 + i];
        const bool live = hint_move && i < psize;
        const bool match = live && (by_color ? ccol == val : crk == val);
        reveal |= match ? 1u << i : 0u;
        newly |= (match && qkc[i] == 0xFFu) ? 1u << i : 0u;  // sim.cpp:776 tests known_color for rank hints too
        qpl[i] = live ? (match ? qpl[i] & hint : qpl[i] & ~hint) : qpl[i];
        qkc[i] = (match && by_color) ? val : qkc[i];
        qkk[i] = (match && !by_color) ? val : qkk[i];
    }
    info -= hint_move ? 1u : 0u;

    // ---- write back ----
    const uint32_t lm_move = is_card ? (play ? (uint32_t)MV_PLAY : (uint32_t)MV_DISCARD) : (by_color ? (uint32_t)MV_REVEAL_COLOR : (uint32_t)MV_REVEAL_RANK);
    const uint32_t lm_color = is_card ? col : (by_color ? val : 0xFFu);
    const uint32_t lm_rank = is_card ? rk : (by_color ? 0xFFu : val);
    rec32[19] = (uint32_t)fw_new;
    rec32[20] = (uint32_t)(fw_new >> 32) | ((info & 0xFFu) << 8) | ((life & 0xFFu) << 16) | ((actor ^ 1u) << 24);
    rec32[21] = turns | (w21 & 0x00FFFF00u) | (lm_move << 24);
    rec32[22] = actor | ((hint_move ? (actor ^ 1u) : 0xFFu) << 8) | ((is_card ? slot : 0xFFu) << 16) | ((success ? 1u : 0u) << 24);
    rec32[23] = (completed ? 1u : 0u) | (lm_color << 8) | (lm_rank << 16) | (reveal << 24);
    rec32[24] = newly | (0xFFu << 8) | (rec32[24] & 0xFFFF0000u);
    if (is_card) {
        ah[0] = card_of[0] | (card_of[1] << 8) | (card_of[2] << 16) | (card_of[3] << 24);
        ah[1] = card_of[4] | (size_a << 8) | (kc_of[0] << 16) | (kc_of[1] << 24);
        ah[2] = kc_of[2] | (kc_of[3] << 8) | (kc_of[4] << 16) | (kk_of[0] << 24);
        ah[3] = kk_of[1] | (kk_of[2] << 8) | (kk_of[3] << 16) | (kk_of[4] << 24);
#pragma unroll
        for (uint32_t i = 0; i < kHand; i++) ah[4 + i] = pl_of[i];
    } else {
        ph[1] = (q[1] & 0x0000FFFFu) | (qkc[0] << 16) | (qkc[1] << 24);
        ph[2] = qkc[2] | (qkc[3] << 8) | (qkc[4] << 16) | (qkk[0] << 24);
        ph[3] = qkk[1] | (qkk[2] << 8) | (qkk[3] << 16) | (qkk[4] << 24);
#pragma unroll
        for (uint32_t i = 0; i < kHand; i++) ph[4 + i] = qpl[i];
    }
}

template <int kV>
__device__ __forceinline__ void apply_variant(const HanabiParams &p, uint8_t *rec, uint32_t uid)
{
    if constexpr (kV == 2)
        apply_action_full(rec, uid);
    else
        apply_action<(kV ? 5 : 0)>(p, rec, uid);
}

// One world's move on its record in LDS: the mover's action (given, or drawn here under mrl_rollout_random), actionSystem
// (sim.cpp:596-792) and checkDone's score / reward / termination test (sim.cpp:812-850).  Touches nothing but the record
// unless `publish` (then the drawn action goes to the ACTION tensor): the single-launch step's healing look-back runs it
// on a scratch copy of another workgroup's worlds to learn how many of them finish.
struct Moved {
    bool over;
    float reward;
};
template <int kV>
__device__ __forceinline__ Moved move_world(const HanabiParams &p, uint8_t *rec, uint32_t world, int32_t act0, int32_t act1, bool publish)
{
    constexpr int kR = kV ? 5 : 0;
    const uint32_t actor = rec[R_CUR] & 1u;
    uint32_t uid = (uint32_t)(actor ? act1 : act0);
    if (p.sample) {  // uniform over the mover's legal moves (random_policy.hpp)
        const uint32_t legal = legal_moves<kR>(p, rec, actor);
        const uint32_t count = (uint32_t)__popc(legal);
        uid = count ? mrl::nth_set_bit(legal, mrl::scale(mrl::policy_hash(p.sample_seed, p.sample_step, world, actor), count)) : 0u;
        if (publish) p.action_out[(size_t)actor * p.num_worlds + world] = (int32_t)uid;
    }
    apply_variant<kV>(p, rec, uid);
    const int32_t old_score = (int8_t)rec[R_SCORE];
    int32_t score = 0;
    if (rec[R_LIFE] > 0)
        for (uint32_t c = 0; c < p.colors; c++) score += rec[R_FIREWORKS + c];
    rec[R_SCORE] = (uint8_t)score;
    rec[R_NEWREW] = (uint8_t)(score - old_score);
    Moved m;
    m.reward = (float)(int8_t)(score - old_score);
    m.over = rec[R_LIFE] < 1 || (uint32_t)(int8_t)score >= p.colors * (kR ? (uint32_t)kR : p.ranks) || (int8_t)rec[R_TURNS] <= 0;
    return m;
}

// move_world for the full game with the record in REGISTERS (the single-launch step and the persistent rollout, whose phase A is
// one wave per SIMD issuing ~2 400 instructions with nobody to hide a stall behind: there every scalar instruction and every
// LDS round trip counts like a vector instruction -- the transition above compiles to 590 VALU + 300 SALU instructions, 37
// branches around conditional LDS updates, scalar loops over the colours and 27 waits on LDS).  Words 12..43 of the record
// are read once; the mover's legal moves (device-side policy), the action, score and termination are worked out on them
// without a branch; the draw's two deck bytes -- the only accesses by a computed index -- are asked for up front; every
// word a move can change goes back with unconditional stores; and the words stay in r[] for the encoder, which does not
// read the record again.  Same arithmetic, expression by expression, as apply_action_full / draw / move_world: the two-launch
// kernels and the healing recount keep those, and tests/ compare the two paths on every tensor and the whole record.
struct MovedFull {
    bool over;
    float reward;
    uint32_t next;  // the player to move after this step
};
template <bool kKeepWords>
__device__ __forceinline__ MovedFull move_world_full(const HanabiParams &p, uint8_t *rec, uint32_t world, int32_t act0, int32_t act1, bool publish,
                                                     uint32_t (&r)[kRecordWords])
{
    constexpr uint32_t kRk = 5, kAll = (1u << 25) - 1u;
    uint32_t *rec32 = reinterpret_cast<uint32_t *>(rec);
#pragma unroll
    for (int w = 0; w < 12; w++) r[w] = 0u;  // the deck's cards: only ever touched in LDS, by index
    // (kKeepWords false -- the caller's encoder reads the record again, the persistent rollout: the discard piles stay in LDS, or
    // that kernel's thirteen waves at 128 registers spill)
#pragma unroll
    for (int w = 12; w < kRecordWords; w++) r[w] = (kKeepWords || w == 12 || w > 18) ? rec32[w] : 0u;
    const uint32_t w19 = r[19], w20 = r[20], w21 = r[21];
    const uint32_t deck_size = (r[12] >> 16) & 0xFFu;
    const uint32_t actor = (w20 >> 24) & 1u;
    // the draw a play or discard will make (draw(), sim.cpp:45-52): its position depends on the generator and the deck size alone
    const uint32_t g_new = 1664525u * r[43] + 1013904223u;
    const float unit = (float)(g_new & 0x00FFFFFFu) / (float)0x01000000;
    const uint32_t at = (uint32_t)(int32_t)((float)deck_size * unit);  // 0 for an empty deck
    const uint32_t drawn_card = rec[R_DECK + at], last_card = rec[R_DECK + max(deck_size, 1u) - 1u];

    uint32_t a[9], q[9];  // the mover's hand, the partner's
#pragma unroll
    for (int w = 0; w < 9; w++) {
        const uint32_t h0 = r[R_HAND / 4 + w], h1 = r[(R_HAND + HAND_BYTES) / 4 + w];
        a[w] = actor ? h1 : h0;
        q[w] = actor ? h0 : h1;
    }
    auto hbyte = [](const uint32_t (&h)[9], uint32_t b) { return (h[b >> 2] >> ((b & 3u) * 8u)) & 0xFFu; };
    uint32_t info = (w20 >> 8) & 0xFFu, life = (w20 >> 16) & 0xFFu;
    uint32_t size_a = hbyte(a, H_SIZE);

    uint32_t uid = (uint32_t)(actor ? act1 : act0);
    if (p.sample) {  // (wave-uniform) uniform over the mover's legal moves: generateActionMask (sim.cpp:381-444) on the registers
        uint32_t legal = 0;
#pragma unroll
        for (uint32_t i = 0; i < kHand; i++) {
            legal |= ((i < size_a) & (info < 8u)) ? 1u << i : 0u;
            legal |= i < size_a ? 1u << (kHand + i) : 0u;
        }
        uint32_t hints = 0;
#pragma unroll
        for (uint32_t n = 0; n < kHand; n++) {  // all five slots, whatever the hand size (:416-417)
            const uint32_t cardv = hbyte(q, H_CARDS + n);
            const uint32_t col = (cardv * 205u) >> 10, rk = cardv - col * kRk;
            hints |= col < 5u ? 1u << (2 * kHand + col) : 0u;
            hints |= 1u << (2 * kHand + 5u + rk);
        }
        legal |= info > 0 ? hints : 0u;
        legal &= 0xFFFFFu;
        const uint32_t count = (uint32_t)__popc(legal);
        const uint32_t pick = mrl::nth_set_bit(legal, mrl::scale(mrl::policy_hash(p.sample_seed, p.sample_step, world, actor), count));
        uid = count ? pick : 0u;
        if (publish) p.action_out[(size_t)actor * p.num_worlds + world] = (int32_t)uid;
    }

    // ---- apply_action_full, statement by statement, on registers ----
    uint32_t turns = w21 & 0xFFu;
    turns = deck_size == 0 ? (turns - 1u) & 0xFFu : turns;
    const unsigned long long fw = (unsigned long long)w19 | ((unsigned long long)(w20 & 0xFFu) << 32);
    const bool is_card = uid < 2 * kHand;
    // (operands first, then selects, and & | on flags instead of && ||: written as lazy expressions these become branches around
    // a handful of instructions each, and a branch is three or four scalar instructions that a lone wave issues one by one)
    const bool play = is_card & (uid >= kHand);
    const uint32_t slot_play = uid - kHand;
    const uint32_t slot = is_card ? (play ? slot_play : uid) : 0u;
    uint32_t card_of[kHand], kc_of[kHand], kk_of[kHand], pl_of[kHand];
#pragma unroll
    for (uint32_t i = 0; i < kHand; i++) {
        card_of[i] = hbyte(a, H_CARDS + i);
        kc_of[i] = hbyte(a, H_KCOLOR + i);
        kk_of[i] = hbyte(a, H_KRANK + i);
        pl_of[i] = a[4 + i];
    }
    uint32_t chosen = 0;
#pragma unroll
    for (uint32_t i = 0; i < kHand; i++) chosen = slot == i ? card_of[i] : chosen;
    const uint32_t card = min(chosen, 24u);
    const uint32_t col = (card * 205u) >> 10, rk = card - col * kRk;
    const uint32_t top = (uint32_t)(fw >> (8u * col)) & 0xFFu;
    const bool success = play & (top == rk);
    const bool completed = success & (top + 1u == kRk);
    {
        // rec[R_DISCARD + card]++ for a discard or a failed play: a byte of words 12..18
        const uint32_t b = (uint32_t)R_DISCARD + card, wi = b >> 2;
        const uint32_t one = 1u << ((b & 3u) * 8u);
        const uint32_t inc = (is_card & !success) ? one : 0u;
        if constexpr (kKeepWords) {
#pragma unroll
            for (uint32_t w = 12; w <= 18; w++) r[w] += wi == w ? inc : 0u;
        } else {
            __hip_atomic_fetch_add(rec32 + wi, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (no result: ds_add_u32, no round trip)
        }
    }
    info += (is_card & !play) ? 1u : 0u;
    info += completed ? 1u : 0u;
    life -= (play & !success) ? 1u : 0u;
    const unsigned long long fw_step = 1ull << (8u * col);
    const unsigned long long fw_new = fw + (success ? fw_step : 0ull);
    // removeFromHand (:567-594): shift left when the deck is empty, else the drawn card into the slot
    const bool shifts = is_card & (deck_size == 0), draws = is_card & (deck_size != 0);
#pragma unroll
    for (uint32_t i = 0; i + 1 < kHand; i++) {
        const bool take_next = shifts & (i >= slot) & (i + 1 < size_a);
        card_of[i] = take_next ? card_of[i + 1] : card_of[i];
        kc_of[i] = take_next ? kc_of[i + 1] : kc_of[i];
        kk_of[i] = take_next ? kk_of[i + 1] : kk_of[i];
        pl_of[i] = take_next ? pl_of[i + 1] : pl_of[i];
    }
    const uint32_t size_less = (size_a - 1u) & 0xFFu;
    size_a = shifts ? size_less : size_a;
#pragma unroll
    for (uint32_t i = 0; i < kHand; i++) {
        const bool here = draws & (i == slot);
        card_of[i] = here ? drawn_card : card_of[i];
        kc_of[i] = here ? 0xFFu : kc_of[i];
        kk_of[i] = here ? 0xFFu : kk_of[i];
        pl_of[i] = here ? kAll : pl_of[i];
    }
    // hints (:695-788)
    const bool hint_move = !is_card;
    const uint32_t u = uid - 2 * kHand;
    const bool by_color = u < 5u;
    const uint32_t u_rank = u - 5u;
    const uint32_t val_rank = u_rank - ((u_rank * 205u) >> 10) * kRk;  // (uid - K) % R
    const uint32_t val = by_color ? u : val_rank;
    const uint32_t hint_color = 0x1Fu << ((kRk * val) & 31u), hint_rank = 0x108421u << (val & 31u);  // the 5 cards of a colour / of a rank
    const uint32_t hint = by_color ? hint_color : hint_rank;
    const uint32_t psize = min(hbyte(q, H_SIZE), (uint32_t)kHand);
    uint32_t reveal = 0, newly = 0;
    uint32_t qkc[kHand], qkk[kHand], qpl[kHand];
#pragma unroll
    for (uint32_t i = 0; i < kHand; i++) {
        const uint32_t c = hbyte(q, H_CARDS + i);
        const uint32_t ccol = (c * 205u) >> 10, crk = c - ccol * kRk;
        qkc[i] = hbyte(q, H_KCOLOR + i);
        qkk[i] = hbyte(q, H_KRANK + i);
        qpl[i] = q[4 + i];
        const bool live = hint_move & (i < psize);
        const bool same_color = ccol == val, same_rank = crk == val;
        const bool match = live & (by_color ? same_color : same_rank);
        reveal |= match ? 1u << i : 0u;
        newly |= (match & (qkc[i] == 0xFFu)) ? 1u << i : 0u;  // sim.cpp:776 tests known_color for rank hints too
        const uint32_t kept = qpl[i] & hint, struck = qpl[i] & ~hint;
        qpl[i] = live ? (match ? kept : struck) : qpl[i];
        qkc[i] = (match & by_color) ? val : qkc[i];
        qkk[i] = (match & !by_color) ? val : qkk[i];
    }
    info -= hint_move ? 1u : 0u;

    // checkDone's score / reward / termination (sim.cpp:812-850)
    const uint32_t life8 = life & 0xFFu;
    const int32_t old_score = (int8_t)((w21 >> 8) & 0xFFu);
    const int32_t fireworks = (int32_t)(__builtin_amdgcn_sad_u8((uint32_t)fw_new, 0u, 0u) + ((uint32_t)(fw_new >> 32) & 0xFFu));
    const int32_t score = life8 > 0 ? fireworks : 0;
    MovedFull m;
    m.reward = (float)(int8_t)(score - old_score);
    m.over = (life8 < 1) | ((uint32_t)(int8_t)score >= 25u) | ((int8_t)turns <= 0);
    m.next = actor ^ 1u;

    // ---- the record's words after the move ----
    const uint32_t lm_move = is_card ? (play ? (uint32_t)MV_PLAY : (uint32_t)MV_DISCARD) : (by_color ? (uint32_t)MV_REVEAL_COLOR : (uint32_t)MV_REVEAL_RANK);
    const uint32_t lm_color = is_card ? col : (by_color ? val : 0xFFu);
    const uint32_t lm_rank = is_card ? rk : (by_color ? 0xFFu : val);
    const uint32_t deck_less = deck_size - 1u;
    r[12] = (r[12] & 0xFF00FFFFu) | (((draws ? deck_less : deck_size) & 0xFFu) << 16);
    r[19] = (uint32_t)fw_new;
    r[20] = (uint32_t)(fw_new >> 32) | ((info & 0xFFu) << 8) | (life8 << 16) | ((actor ^ 1u) << 24);
    r[21] = turns | (((uint32_t)score & 0xFFu) << 8) | (((uint32_t)(score - old_score) & 0xFFu) << 16) | (lm_move << 24);
    r[22] = actor | ((hint_move ? (actor ^ 1u) : 0xFFu) << 8) | ((is_card ? slot : 0xFFu) << 16) | ((success ? 1u : 0u) << 24);
    r[23] = (completed ? 1u : 0u) | (lm_color << 8) | (lm_rank << 16) | (reveal << 24);
    r[24] = newly | (0xFFu << 8) | (r[24] & 0xFFFF0000u);
    r[43] = draws ? g_new : r[43];
    uint32_t an[9], qn[9];  // the two hands after the move: one of them changed
    an[0] = card_of[0] | (card_of[1] << 8) | (card_of[2] << 16) | (card_of[3] << 24);
    an[1] = card_of[4] | (size_a << 8) | (kc_of[0] << 16) | (kc_of[1] << 24);
    an[2] = kc_of[2] | (kc_of[3] << 8) | (kc_of[4] << 16) | (kk_of[0] << 24);
    an[3] = kk_of[1] | (kk_of[2] << 8) | (kk_of[3] << 16) | (kk_of[4] << 24);
    qn[0] = q[0];
    qn[1] = (q[1] & 0x0000FFFFu) | (qkc[0] << 16) | (qkc[1] << 24);
    qn[2] = qkc[2] | (qkc[3] << 8) | (qkc[4] << 16) | (qkk[0] << 24);
    qn[3] = qkk[1] | (qkk[2] << 8) | (qkk[3] << 16) | (qkk[4] << 24);
#pragma unroll
    for (uint32_t i = 0; i < kHand; i++) {
        an[4 + i] = pl_of[i];
        qn[4 + i] = qpl[i];
    }
#pragma unroll
    for (int w = 0; w < 9; w++) {
        an[w] = is_card ? an[w] : a[w];
        qn[w] = is_card ? q[w] : qn[w];
        r[R_HAND / 4 + w] = actor ? qn[w] : an[w];
        r[(R_HAND + HAND_BYTES) / 4 + w] = actor ? an[w] : qn[w];
    }
    // ---- back to LDS, without a branch: every word a move can change (the hand that changed: one of the two) ----
    if constexpr (kKeepWords) {
        *reinterpret_cast<uint16_t *>(rec + R_DECK_SIZE) = (uint16_t)(r[12] >> 16);  // deck size, discard[0] (the deck's last two cards share the word)
#pragma unroll
        for (int w = 13; w <= 18; w++) rec32[w] = r[w];
    } else {
        rec[R_DECK_SIZE] = (uint8_t)(r[12] >> 16);
    }
#pragma unroll
    for (int w = 19; w <= 24; w++) rec32[w] = r[w];
    {
        uint32_t *changed = rec32 + (R_HAND + HAND_BYTES * (is_card ? actor : actor ^ 1u)) / 4;
#pragma unroll
        for (int w = 0; w < 9; w++) changed[w] = is_card ? an[w] : qn[w];
    }
    rec32[R_RNG / 4] = r[43];
    rec[R_DECK + at] = (uint8_t)(draws ? last_card : drawn_card);  // (no draw: the byte as it was)
    return m;
}

// sim.cpp:446-532, without the encode.  The ten opening draws keep the generator and the deck
// size in registers; the LDS reads of one draw (the drawn card, the deck's last card) do not
// depend on the previous draw's write being waited for (DS ops execute in order).
template <int kR>
__device__ void deal_new_game(const HanabiParams &p, uint8_t *rec, uint32_t episode)
{
    uint32_t *rec32 = reinterpret_cast<uint32_t *>(rec);
#pragma unroll
    for (int w = 0; w < 13; w++) rec32[w] = p.deck_words[w];
#pragma unroll
    for (int w = 13; w < kRecordWords; w++) rec32[w] = 0;
    rec[R_INFO] = (uint8_t)p.max_info;
    rec[R_LIFE] = (uint8_t)p.max_life;
    rec[R_CUR] = 0;
    rec[R_TURNS] = 2;
    rec[R_LM_MOVE] = MV_INVALID;
    rec[R_LM_PLAYER] = 0xFF;
    rec[R_LM_TARGET] = 0xFF;
    rec[R_LM_INDEX] = 0xFF;
    rec[R_LM_COLOR] = 0xFF;
    rec[R_LM_RANK] = 0xFF;
    rec[R_LM_DEALTO] = 0xFF;
    uint32_t g = seed_of(episode);
    uint32_t size = p.deck_words[12] >> 16 & 0xFFu;  // byte 50
    const uint32_t all = ones(p.bpc);
    for (uint32_t a = 0; a < 2; a++) {
        uint8_t *h = rec + R_HAND + HAND_BYTES * a;
#pragma unroll
        for (uint32_t j = 0; j < kHand; j++) {
            // drawDeck (sim.cpp:45-52): one float multiply, truncation
            g = 1664525u * g + 1013904223u;
            const float r = (float)(g & 0x00FFFFFFu) / (float)0x01000000;
            const int32_t at = (int32_t)((float)size * r);
            const uint8_t card = rec[R_DECK + at];
            rec[R_DECK + at] = rec[R_DECK + size - 1];
            size -= 1;
            h[H_CARDS + j] = card;
            plaus_of(h)[j] = all;
            h[H_KCOLOR + j] = 0xFF;
            h[H_KRANK + j] = 0xFF;
        }
        h[H_SIZE] = kHand;
    }
    rec[R_DECK_SIZE] = (uint8_t)size;
    rng_of(rec) = g;
}

// 16-byte write-through (sc1) stores: the observation / state / mask rows are written once per
// step and not read by these kernels; plain stores would pile up dirty in L2 until the
// end-of-kernel write-back (measured on the Overcooked kernel: 14.0 -> 11.8 us per launch).
// Issued as raw buffer stores over ONE ROW: the compiler knows them (an inline-asm store is
// invisible to its s_waitcnt bookkeeping, and a stale `vmcnt(0)` in the expansion loop then
// waits for every store of the previous iteration: 2 300 cycles per iteration measured), and
// lanes whose 16 bytes start past the row end are dropped by the buffer bounds check, so the
// expansion has no per-lane branches.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_resource(void *row_base, uint32_t row_bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(row_base, 0, (int)row_bytes, 0x00020000);
}
__device__ __forceinline__ void row_store(__amdgpu_buffer_rsrc_t row, uint32_t byte_offset, const uint4 &v)
{
    u32x4 r;
    r.x = v.x;
    r.y = v.y;
    r.z = v.z;
    r.w = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(r, row, (int)byte_offset, 0, 16);  // aux bit 4 = sc1
}

__device__ __forceinline__ uint32_t spread4(uint32_t nibble)
{
    // 4 bits -> 4 bytes of 0/1 (the shifted copies do not overlap, so no carries)
    return (nibble * 0x00204081u) & 0x01010101u;
}

// Phase B: chunk `ch` (16 bytes) of one agent block from that agent's bit vector.  Chunks 0..48 are the
// state row (whose first 658 bytes are the observation), 49..53 the legal-move mask as five int32x4,
// 54..55 padding up to whole cache lines.  Straight-line code: a state chunk spreads four nibbles of a 16-bit
// piece of the bit vector into four bytes each, a mask chunk four single bits of the legal-move word into one
// int32 each -- the same  shift, mask, spread  with a different stride (4 or 1) and mask (0xF or 1), chosen per lane
// by selects.  (Written as `is_state ? spread4(..) : bit`, hipcc compiled ten divergent branches per chunk, and
// the three LDS reads of a chunk were waited for before anything else was issued: 0.30 us per 1 KB store and wave,
// whatever HBM was doing -- tools/stamps_hanabi_rollout.py.)
struct ChunkSource {
    uint32_t word, legal, shift;  // the 32 bits holding the chunk's piece, the legal-move word, how far the encoding was shifted
};
__device__ __forceinline__ ChunkSource chunk_source(const uint32_t *enc, uint32_t ch)
{
    return ChunkSource{enc[min(ch >> 1, 24u)], enc[25], enc[26]};
}
__device__ __forceinline__ uint4 chunk_bytes(const HanabiParams &p, const ChunkSource &c, uint32_t ch)
{
    // the row is MRL_HANABI_STATE_SIZE wide whatever the configuration; what the shifted encoding pushes past
    // its end is dropped (the reference writes it out of bounds): keep the bits of [16 ch, 16 ch + 16) below `limit`
    const uint32_t limit = min(p.state_bits + c.shift, (uint32_t)MRL_HANABI_STATE_SIZE);
    const int32_t room = (int32_t)limit - (int32_t)(ch * 16u);
    const uint32_t keep = (1u << (uint32_t)min(max(room, 0), 16)) - 1u;
    const uint32_t piece = (c.word >> ((ch & 1u) * 16u)) & keep;
    const bool is_state = ch < (uint32_t)kStateChunks;
    const uint32_t m = min(ch - (uint32_t)kStateChunks, (uint32_t)kMaskChunks);  // mask chunk 0..4; 5 = padding (and state chunks: unused)
    const uint32_t bits = (c.legal >> (4u * m)) & (m < (uint32_t)kMaskChunks ? 0xFu : 0u);
    const uint32_t src = is_state ? piece : bits, stride = is_state ? 4u : 1u, nib = is_state ? 0xFu : 1u;
    return make_uint4(spread4(src & nib), spread4((src >> stride) & nib), spread4((src >> (2u * stride)) & nib), spread4((src >> (3u * stride)) & nib));
}
__device__ __forceinline__ uint4 agent_chunk(const HanabiParams &p, const uint32_t *enc, uint32_t ch) { return chunk_bytes(p, chunk_source(enc, ch), ch); }

struct WaveLds {
    uint8_t *rec;    // kWorldsPerWave x kRecStride
    uint32_t *enc;   // kWorldsPerWave x 2 x kEncWords
};

constexpr int kWaveLdsBytes = kWorldsPerWave * kRecStride + kWorldsPerWave * 2 * kEncWords * 4;

__device__ __forceinline__ WaveLds wave_lds(uint8_t *smem, uint32_t wib)
{
    uint8_t *base = smem + wib * kWaveLdsBytes;
    return WaveLds{base, reinterpret_cast<uint32_t *>(base + kWorldsPerWave * kRecStride)};
}

// Phase B for a wave's movers.  An agent block is seven 128-byte lines of eight chunks, and a wave store is eight lines:
// lane = (world k = lane / 8 of an octet of worlds, chunk c8 = lane % 8 of a line), so that everything that depends on
// the lane alone -- which half of which bit-vector word, the byte offset inside the octet -- is worked out once, a
// line is an immediate offset of the LDS read and of the store, and a round is  shift, 4 x (bfe, mul, and), store:
// 28 stores for 32 worlds as before, each of eight whole lines, on a quarter of the instructions.  (The run-length form
// it replaces -- 64 consecutive chunks per store, world = f / 56 and chunk = f % 56 by multiply-shift, the kind of chunk
// chosen by selects -- issued 70 VALU instructions per store; two waves of a SIMD then take 0.22 us per pair of stores,
// which was what phase B took.)  Lines 0..5 are state chunks 0..47; line 6 is state chunk 48, the five mask chunks
// and two chunks of padding.  kV == 2: the full game's encoder never sets a bit past state_bits + shift and chunk 48
// reads bits 768..783 only, so the `keep` mask of chunk_bytes folds away; the other variants keep it.
// lines [kFirst, kLast) of the movers' blocks (the single-launch step hands line 0 over early, see there)
template <int kV, uint32_t kFirst, uint32_t kLast>
__device__ __forceinline__ void expand_lines(const HanabiParams &p, const WaveLds &l, uint32_t nw, uint32_t overs, uint32_t movers,
                                             __amdgpu_buffer_rsrc_t out, uint32_t lane)
{
    constexpr uint32_t kLines = kAgentBlock / 128u;  // 7
    static_assert(kStateChunks == 8 * (kLines - 1) + 1 && kMaskChunks == 5, "line 6 = state chunk 48, five mask chunks, padding");
    static_assert(kFirst < kLast && kLast <= kLines, "a range of lines");
    constexpr bool kTail = kLast == kLines;                    // line 6 is among them
    constexpr uint32_t kPlain = kTail ? kLines - 1u : kLast;   // lines [kFirst, kPlain) are eight state chunks each
    constexpr uint32_t kDropped = 0x40000000u;  // past any descriptor, and the line offsets added to it do not wrap
    const uint32_t k = lane >> 3, c8 = lane & 7u;
    const uint32_t half = (c8 & 1u) * 16u;
    const uint32_t lane_at = k * kWorldBlock + c8 * 16u;
    // line 6, by lane: chunk 48 of the state (c8 == 0), mask chunk c8 - 1 (c8 1..5), padding (6, 7)
    const bool tail = c8 == 0u;
    const uint32_t legal_at = 4u * min(c8 - 1u, 5u), legal_keep = (c8 - 1u) < (uint32_t)kMaskChunks ? 0xFu : 0u;
    const uint32_t stride = tail ? 4u : 1u, nib = tail ? 0xFu : 1u;
    const uint32_t octets = (nw + 7u) >> 3;  // wave-uniform
    struct Words {
        uint32_t w[kLines], legal, shift;
    };
    auto request = [&](uint32_t o) {
        const uint32_t *e = l.enc + min(o * 8u + k, (uint32_t)kWorldsPerWave - 1u) * 2 * kEncWords;
        Words q;
#pragma unroll
        for (uint32_t line = kFirst; line < kPlain; line++) q.w[line] = e[line * 4u + (c8 >> 1)];
        if constexpr (kTail) {
            q.w[kLines - 1] = e[24];
            q.legal = e[25];
        }
        if constexpr (kV != 2) q.shift = e[26];
        return q;
    };
    Words next = request(0);
#pragma unroll 1
    for (uint32_t o = 0; o < octets; o++) {
        const Words cur = next;
        next = request(o + 1u);  // requested before this octet's arithmetic
        const uint32_t r = o * 8u + k;
        const bool skip = r >= nw || ((overs >> r) & 1u);  // a finished world's rows come from the re-deal
        const uint32_t agent = (movers >> r) & 1u;
        const uint32_t at = skip ? kDropped : o * (8u * kWorldBlock) + lane_at + agent * kAgentBlock;
        auto piece_of = [&](uint32_t word, uint32_t line) {
            const uint32_t piece = (word >> half) & 0xFFFFu;
            if constexpr (kV == 2) {
                return piece;
            } else {  // what the shifted encoding pushes past the row's end is dropped
                const int32_t limit = (int32_t)min(p.state_bits + cur.shift, (uint32_t)MRL_HANABI_STATE_SIZE) - (int32_t)(c8 * 16u);
                const int32_t room = limit - (int32_t)(line * 128u);
                return piece & ((1u << (uint32_t)min(max(room, 0), 16)) - 1u);
            }
        };
#pragma unroll
        for (uint32_t line = kFirst; line < kPlain; line++) {
            const uint32_t piece = piece_of(cur.w[line], line);
            row_store(out, at + line * 128u,
                      make_uint4(spread4(piece & 0xFu), spread4((piece >> 4) & 0xFu), spread4((piece >> 8) & 0xFu), spread4(piece >> 12)));
        }
        if constexpr (kTail) {
            // (chunk 48 sits in the low half of word 24: `half` is 0 for the lane that takes it)
            const uint32_t src = tail ? piece_of(cur.w[kLines - 1], kLines - 1) : (cur.legal >> legal_at) & legal_keep;
            row_store(out, at + (kLines - 1) * 128u,
                      make_uint4(spread4(src & nib), spread4((src >> stride) & nib), spread4((src >> (2u * stride)) & nib),
                                 spread4((src >> (3u * stride)) & nib)));
        }
    }
}
template <int kV>
__device__ __forceinline__ void expand_movers(const HanabiParams &p, const WaveLds &l, uint32_t nw, uint32_t overs, uint32_t movers,
                                              __amdgpu_buffer_rsrc_t out, uint32_t lane)
{
    expand_lines<kV, 0u, (uint32_t)kAgentBlock / 128u>(p, l, nw, overs, movers, out, lane);
}

// The wave's records, HBM <-> LDS.  A wave's worlds are one contiguous run of 176-byte records (32 worlds = 44 whole
// cache lines), moved 16 bytes per lane and instruction; a record is 11 such chunks, so none straddles two records.
// Loading comes in two halves so that a kernel can request the records with its first instructions and put them into
// LDS when it gets there: all global loads are in flight before the first LDS write.
constexpr int kRecordChunks = kRecordBytes / 16;  // 11
static_assert(kRecordChunks * 16 == kRecordBytes, "records are whole 16-byte chunks");
constexpr int kRecordLoadRounds = (kWorldsPerWave * kRecordChunks + kWave - 1) / kWave;  // 6
struct RecordLoads {
    uint4 v[kRecordLoadRounds];
};
__device__ __forceinline__ void request_records(const uint32_t *records, uint32_t w0, uint32_t nw, uint32_t lane, RecordLoads &r)
{
    const uint32_t total = nw * kRecordChunks;
    const uint4 *src = reinterpret_cast<const uint4 *>(records + (size_t)w0 * kRecordWords);
#pragma unroll
    for (int k = 0; k < kRecordLoadRounds; k++) {
        const uint32_t c = lane + k * kWave;
        r.v[k] = c < total ? src[c] : make_uint4(0u, 0u, 0u, 0u);
    }
}
__device__ __forceinline__ uint32_t *record_chunk_in_lds(const WaveLds &l, uint32_t c)
{
    const uint32_t rec = (c * 745u) >> 13, part = c - rec * kRecordChunks;  // c / 11, exact for c < 2738
    return reinterpret_cast<uint32_t *>(l.rec + rec * kRecStride + part * 16u);  // 4-byte aligned (the stride is 180)
}
__device__ __forceinline__ void place_records(const WaveLds &l, uint32_t nw, uint32_t lane, const RecordLoads &r)
{
    const uint32_t total = nw * kRecordChunks;
#pragma unroll
    for (int k = 0; k < kRecordLoadRounds; k++) {
        const uint32_t c = lane + k * kWave;
        if (c < total) {
            uint32_t *dst = record_chunk_in_lds(l, c);
            dst[0] = r.v[k].x;
            dst[1] = r.v[k].y;
            dst[2] = r.v[k].z;
            dst[3] = r.v[k].w;
        }
    }
}
__device__ __forceinline__ void load_records(const HanabiParams &p, const WaveLds &l, uint32_t w0, uint32_t nw, uint32_t lane)
{
    RecordLoads r;
    request_records(p.records, w0, nw, lane, r);
    place_records(l, nw, lane, r);
}

// all of the wave's records back to HBM: six full-width stores of whole cache lines instead of one 176-byte store per world
__device__ __forceinline__ void store_records(const HanabiParams &p, const WaveLds &l, uint32_t w0, uint32_t nw, uint32_t lane)
{
    const uint32_t total = nw * kRecordChunks;
    uint4 *dst = reinterpret_cast<uint4 *>(p.records + (size_t)w0 * kRecordWords);
#pragma unroll
    for (int k = 0; k < kRecordLoadRounds; k++) {
        const uint32_t c = lane + k * kWave;
        if (c < total) {
            const uint32_t *src = record_chunk_in_lds(l, c);
            dst[c] = make_uint4(src[0], src[1], src[2], src[3]);
        }
    }
}

// Both kernels run on the same grid: workgroup b owns worlds [b*chunk, (b+1)*chunk), p.chunk a
// multiple of kWorldsPerBlock, and walks it kWorldsPerBlock worlds at a time (episode_scan.hpp).
// The transition of the workgroup's worlds; returns how many of them finished (to every thread).
// Rows of finished worlds are not written: the reset that follows writes both agents' rows anew.
template <int kV>
__device__ __forceinline__ uint32_t step_body(const HanabiParams &p, uint8_t *smem, uint32_t *s_counts, const uint32_t bid)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keeps w0, nw and the row descriptors in SGPRs
    const WaveLds l = wave_lds(smem, wib);
    const uint32_t N = p.num_worlds;
    const uint32_t chunk_end = min(N, (bid + 1) * p.chunk);
    uint32_t finished = 0;

    for (uint32_t sub = bid * p.chunk; sub < chunk_end; sub += kWorldsPerBlock) {
        const uint32_t w0 = sub + wib * kWorldsPerWave;
        const uint32_t nw = w0 < chunk_end ? min((uint32_t)kWorldsPerWave, chunk_end - w0) : 0u;
        STAMP(0);
        STAMP_REALTIME(13);
        // both agents' actions are fetched with the record (the actor is only known once the
        // record is in LDS; a dependent global load there costs a full HBM latency in phase A)
        int32_t act0 = 0, act1 = 0;
        if (lane < nw && !p.sample) {
            act0 = p.actions[w0 + lane];
            act1 = p.actions[(size_t)N + w0 + lane];
        }
        load_records(p, l, w0, nw, lane);
        wave_lds_sync();
        STAMP(1);

        bool over = false, next_is_1 = false;
        if (lane < nw) {
            uint8_t *rec = l.rec + lane * kRecStride;
            uint32_t *enc = l.enc + lane * 2 * kEncWords;
            const uint32_t world = w0 + lane;
            Moved m{false, 0.f};
            uint32_t next;
            if constexpr (kV == 2 && MRL_HANABI_REG_TRANSITION) {
                // the full game's transition and encode on registers (move_world_full; the healing recount and the generic
                // variants keep move_world, and tests/ compare variant 2 with 1 and 0 on every tensor and the record)
                uint32_t r[kRecordWords];
                const MovedFull mf = move_world_full<true>(p, rec, world, act0, act1, true, r);
                STAMP(2);
                m = Moved{mf.over, mf.reward};
                next = mf.next;
                encode_record_full_t<false>(r, enc, next);
            } else {
                if (!ABLATED(1)) m = move_world<kV>(p, rec, world, act0, act1, true);
                STAMP(2);
                next = rec[R_CUR] & 1u;
                // observationSystem (:794-810): only the player to move is refreshed
                if (!ABLATED(1)) encode_variant<kV>(p, rec, enc, next);
            }
            next_is_1 = next != 0;
            STAMP(3);
            // (Moving these small per-world stores behind phase B -- hipcc waits on vmcnt before it reuses a
            // store's data register, 0.4..1 us here -- was measured: 36.0 us per step against 34.5.)
            p.active[(size_t)next * N + world] = 1;
            p.active[(size_t)(next ^ 1u) * N + world] = 0;
            p.reward[world] = m.reward;
            p.reward[(size_t)N + world] = m.reward;
            over = m.over;
            p.done[world] = over ? 1 : 0;
        }
        const unsigned long long overs = __ballot(over);
        finished += (uint32_t)__popcll(overs);
        const unsigned long long movers = __ballot(next_is_1);
        wave_lds_sync();
        STAMP(4);

        // phase B: bits -> bytes for every world's player to move; the wave's worlds are one
        // contiguous run of blocks, so a lane's target is a 32-bit offset from a scalar base
        {
            const uint32_t w0s = (uint32_t)__builtin_amdgcn_readfirstlane((int)w0);
            const __amdgpu_buffer_rsrc_t out = row_resource(p.rows + (size_t)w0s * kWorldBlock, nw * kWorldBlock);
            if (!ABLATED(2)) expand_movers<kV>(p, l, nw, (uint32_t)overs, (uint32_t)movers, out, lane);
        }
        STAMP(5);
        store_records(p, l, w0, nw, lane);  // (a finished world's record is dealt anew by the re-deal launch that follows)
        wave_lds_sync();
        STAMP(6);
        STAMP_REALTIME(14);
    }

    if (lane == 0) s_counts[wib] = finished;
    mrl::lds_barrier();
    uint32_t total = 0;
    for (int w = 0; w < kWavesPerBlock; w++) total += s_counts[w];
    mrl::lds_barrier();  // s_counts is reused by the reset
    return total;
}

// Worlds per wave / waves per SIMD, measured at 65536 worlds under the random policy (us per step):
// 16 worlds x 4 waves 36.2, 24 x 3 37.3, 32 x 2 34.7, 48 x 2 39.1, 64 x 1 48.3.  Phase A costs a wave
// the same instructions whether 16 or 32 of its lanes are worlds (the kernel issues ~3 200 VALU
// instructions per wave, PMC), so fewer, fuller waves win until latency is no longer hidden.
// (With 16 worlds per wave the kernel must be held to 128 VGPRs: at the 132 the compiler picks
// unasked, a quarter of the workgroups start only when the first ones finish.)
// The leading scalar arguments -- what a wave needs to find its worlds and request their records and actions -- are
// preloaded into SGPRs by the command processor (-amdgpu-kernarg-preload-count in the Makefile; a by-value struct is not
// eligible): the record loads do not wait for a scalar load of the argument segment.  The struct carries the rest.
template <int kV>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MRL_HANABI_EU)))
mrl_hanabi_step(uint32_t *hot_records, const int32_t *hot_actions, uint32_t hot_num_worlds, uint32_t hot_chunk, const HanabiParams p)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[kWavesPerBlock * kWaveLdsBytes];
    __shared__ uint32_t s_counts[kWavesPerBlock];
    HanabiParams q = p;
    q.records = hot_records;
    q.actions = hot_actions;
    q.num_worlds = hot_num_worlds;
    q.chunk = hot_chunk;
    const uint32_t total = step_body<kV>(q, smem, s_counts, blockIdx.x);
    if (threadIdx.x == 0) q.block_counts[blockIdx.x] = total;
}


// Re-deal the workgroup's finished worlds (kAll: all of them) as episodes base + running,
// base + running + 1, ... in ascending world order, and write both agents' rows.
template <bool kAll, int kV>
__device__ __forceinline__ void reset_body(const HanabiParams &p, uint8_t *smem, uint32_t *s_counts, uint8_t *s_list, uint32_t base,
                                           uint32_t running, const uint32_t bid, bool have_flag = false, bool flag = false)
{
    constexpr int kR = kV ? 5 : 0;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keeps w0, nw and the row descriptors in SGPRs
    const WaveLds l = wave_lds(smem, wib);
    const uint32_t N = p.num_worlds;
    const uint32_t chunk_end = min(N, (bid + 1) * p.chunk);
    // Finished worlds are sparse (a few per 64), so they are compacted over the workgroup first:
    // s_list holds their local indices in ascending order.  Entry e goes to wave e % 4, slot
    // e / 4, so the long serial re-deal runs on all four SIMDs instead of queueing in wave 0;
    // the two agents of a re-dealt world are then encoded by two lanes side by side.
    for (uint32_t sub = bid * p.chunk; sub < chunk_end; sub += kWorldsPerBlock) {  // uniform trip count
        const uint32_t local = wib * kWorldsPerWave + lane;
        // (the single-launch step hands over the flag it has just computed when the workgroup owns one sub-block)
        const bool over = lane < kWorldsPerWave && sub + local < chunk_end && (kAll || (have_flag ? flag : p.done[sub + local] != 0));
        const unsigned long long votes = __ballot(over);
        if (lane == 0) s_counts[wib] = (uint32_t)__popcll(votes);
        mrl::lds_barrier();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < kWavesPerBlock; w++) {
            before += w < wib ? s_counts[w] : 0u;
            total += s_counts[w];
        }
        if (over) s_list[before + (uint32_t)__popcll(votes & ((1ull << lane) - 1ull))] = (uint8_t)local;
        mrl::lds_barrier();
        STAMP(8);

        const uint32_t my_n = (total + kWavesPerBlock - 1 - wib) / kWavesPerBlock;  // entries wib, wib + 4, ...  (<= 16)
        if (lane < my_n) {
            const uint32_t entry = lane * kWavesPerBlock + wib;
            const uint32_t world = sub + s_list[entry];
            // entries are in ascending world order, so entry k of this sub-block is the
            // (running + k)-th finished world of the step
            deal_new_game<kR>(p, l.rec + lane * kRecStride, base + (kAll ? world : running + entry));
            p.active[world] = 1;
            p.active[(size_t)N + world] = 0;
            if (kAll) {
                p.reward[world] = 0.f;
                p.reward[(size_t)N + world] = 0.f;
                p.done[world] = 0;
            }
        }
        wave_lds_sync();
        STAMP(9);
        if (lane < 2 * my_n) encode_fresh<kV>(p, l.rec + (lane >> 1) * kRecStride, l.enc + lane * kEncWords, lane & 1u);
        wave_lds_sync();
        STAMP(10);
        for (uint32_t r = 0; r < my_n; r++) {
            const uint32_t world = sub + s_list[r * kWavesPerBlock + wib];
            const uint32_t ws = (uint32_t)__builtin_amdgcn_readfirstlane((int)world);
            const __amdgpu_buffer_rsrc_t out = row_resource(p.rows + (size_t)ws * kWorldBlock, kWorldBlock);
#pragma unroll
            for (uint32_t f = lane; f < 2 * kAgentChunks; f += kWave) {  // both agents: 112 chunks
                const uint32_t agent = f >= kAgentChunks ? 1u : 0u;
                row_store(out, f * 16u, agent_chunk(p, l.enc + (r * 2 + agent) * kEncWords, f - agent * kAgentChunks));
            }
            if (lane < kRecordWords)
                p.records[(size_t)world * kRecordWords + lane] = reinterpret_cast<const uint32_t *>(l.rec + r * kRecStride)[lane];
        }
        running += total;
        STAMP(11);
        STAMP_REALTIME(12);
        mrl::lds_barrier();  // s_list / s_counts are rewritten by the next sub-block
    }
}

// kAll: (re)initialise every world as episode episode_base + world (construction /
// mrl_reseed_shard); otherwise only the worlds whose done flag is set, numbered in
// ascending world order from *episode_base.
template <bool kAll, int kV>
__global__ void __launch_bounds__(kBlock) mrl_hanabi_reset(const HanabiParams p, const uint32_t *episode_base,
                                                           uint32_t episode_base_value, uint32_t *next_counter,
                                                           uint32_t *reset_count, const mrl::GatheredCounts gathered,
                                                           const mrl::DeviceCounter device_counter)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[kWavesPerBlock * kWaveLdsBytes];
    __shared__ uint32_t s_counts[kWavesPerBlock];
    __shared__ uint32_t s_part[2 * kWavesPerBlock];
    __shared__ uint8_t s_list[kWorldsPerBlock];
    const bool last_block = blockIdx.x == gridDim.x - 1;
    uint32_t running = 0, grand_total = 0;  // finished worlds before this workgroup's
    // everything the re-deal waits for is requested in one round trip: this workgroup's count, the
    // episode base and (when the workgroup owns a single sub-block) its worlds' done flags
    const bool one_sub = !kAll && p.chunk == (uint32_t)kWorldsPerBlock;
#ifdef MRL_DIAG
    if (!kAll && p.stamps && (threadIdx.x & 63u) == 0) {
        p.stamps[(size_t)(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * 16 + 7] = __builtin_amdgcn_s_memtime();
        p.stamps[(size_t)(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * 16 + 15] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    uint32_t unused_epoch = 0;
    if (!kAll) device_counter.apply(episode_base, next_counter, unused_epoch);  // (the launch state may live in device memory: common.hpp)
    uint32_t base = kAll ? episode_base_value : *episode_base, all_ranks = 0;
    const uint32_t counter_now = base;
    if (!kAll && gathered.counts) base += mrl::lower_ranks(gathered, &all_ranks);  // sharded batch: the ranks below come first
    bool flag = false;
    if (one_sub) {
        const uint32_t local = (threadIdx.x >> 6) * kWorldsPerWave + (threadIdx.x & 63u);
        const uint32_t world = blockIdx.x * p.chunk + local;
        flag = (threadIdx.x & 63u) < (uint32_t)kWorldsPerWave && world < p.num_worlds && p.done[world] != 0;
    }
    if (!kAll) {
        if (p.block_counts[blockIdx.x] == 0 && !last_block) return;  // uniform for the workgroup
        running = mrl::scan_prefix(p.block_counts, gridDim.x, blockIdx.x, s_part, last_block, &grand_total);
    }
    if (!kAll && last_block && threadIdx.x == 0) {
        *reset_count = grand_total;
        *next_counter = gathered.counts ? counter_now + all_ranks : base + grand_total;
    }
    reset_body<kAll, kV>(p, smem, s_counts, s_list, base, running, blockIdx.x, one_sub, flag);
}

// How many worlds of workgroup j's chunk finish in this step, worked out by ONE wave of another workgroup from j's
// inputs (records and actions in HBM) on a scratch copy in LDS: what the healing look-back calls for a workgroup whose
// own count has not appeared (episode_scan.hpp).  Inlined (a call would give the kernel a stack in scratch memory: 39.8 against 30.4 us per step measured); it never runs on an idle GPU.
template <int kV>
__device__ __forceinline__ uint32_t recount_chunk(const HanabiParams &p, uint8_t *scratch, uint32_t j)
{
    const uint32_t lane = threadIdx.x & 63;
    const WaveLds l{scratch, nullptr};
    const uint32_t N = p.num_worlds;
    const uint32_t first = j * p.chunk, end = min(N, first + p.chunk);
    uint32_t count = 0;
    for (uint32_t w0 = first; w0 < end; w0 += kWorldsPerWave) {
        const uint32_t nw = min((uint32_t)kWorldsPerWave, end - w0);
        int32_t act0 = 0, act1 = 0;
        if (lane < nw && !p.sample) {
            act0 = p.actions[w0 + lane];
            act1 = p.actions[(size_t)N + w0 + lane];
        }
        load_records(p, l, w0, nw, lane);
        wave_lds_sync();
        bool over = false;
        if (lane < nw) over = move_world<kV>(p, scratch + lane * kRecStride, w0 + lane, act0, act1, false).over;
        count += (uint32_t)__popcll(__ballot(over));
        wave_lds_sync();
    }
    return count;
}

// The scan wave's re-deal: the workgroup's finished worlds dealt anew IN their slots of the stepping waves' LDS (record and
// both agents' bit vectors), 32 per round, in ascending world order: entry e is the (first_episode + e)-th episode
// (entries of wave w: start_of[w] .. start_of[w + 1]; s_fin[w][i] = the wave's i-th finished world; slot_of(w) = where wave
// w's worlds live in LDS).  Their rows are
// written by the wave that owns the slot, at the end of its own stream of stores: from the scan wave every store would be
// a round trip of its own through a saturated fabric (measured: 14 us for five worlds).
template <int kV, typename SlotOf>
__device__ __forceinline__ void deal_finished_worlds(const HanabiParams &p, SlotOf slot_of, const uint8_t (*s_fin)[kWorldsPerWave],
                                                     const uint32_t (&start_of)[kWavesPerBlock + 1], uint32_t first_episode, uint32_t lane)
{
    constexpr int kR = kV ? 5 : 0;
    const uint32_t block_total = start_of[kWavesPerBlock];
    for (uint32_t e0 = 0; e0 < block_total; e0 += kWorldsPerWave) {
        const uint32_t here = min((uint32_t)kWorldsPerWave, block_total - e0);
        uint32_t wv = 0, local = 0;  // entry e0 + (lane & 31): which wave's slot
        {
            const uint32_t e = e0 + (lane & 31u);
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; w++) wv += e >= start_of[w] ? 1u : 0u;
            uint32_t first = 0;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; w++) first = wv == (uint32_t)w ? start_of[w] : first;
            local = (lane & 31u) < here ? s_fin[wv][e - first] : 0u;
        }
        const WaveLds slot = slot_of(wv);
        uint8_t *rec = slot.rec + local * kRecStride;
        if (lane < here) deal_new_game<kR>(p, rec, first_episode + e0 + lane);
        wave_lds_sync();
        // the two agents of a fresh game side by side: lanes 0..31 agent 0, lanes 32..63 agent 1
        if ((lane & 31u) < here) encode_fresh<kV>(p, rec, slot.enc + (local * 2 + (lane >> 5)) * kEncWords, lane >> 5);
        wave_lds_sync();
    }
}

// The rows of a stepping wave's finished worlds (both agents' blocks of each), from the bit vectors the scan wave left in LDS
__device__ __forceinline__ void store_fresh_rows(const HanabiParams &p, const WaveLds &l, const uint8_t *fin, uint32_t mine, uint32_t w0, uint32_t lane)
{
    for (uint32_t j = 0; j < mine; j++) {
        const uint32_t who = (uint32_t)__builtin_amdgcn_readfirstlane((int)fin[j]);
        const __amdgpu_buffer_rsrc_t out = row_resource(p.rows + (size_t)(w0 + who) * kWorldBlock, kWorldBlock);
#pragma unroll
        for (uint32_t f = lane; f < 2 * kAgentChunks; f += kWave) {
            const uint32_t agent = f >= kAgentChunks ? 1u : 0u;
            row_store(out, f * 16u, agent_chunk(p, l.enc + (who * 2 + agent) * kEncWords, f - agent * kAgentChunks));
        }
    }
}

// Hand-offs between the waves of a workgroup through counters in LDS instead of barriers: a counter is raised after the LDS
// traffic it announces.
__device__ __forceinline__ void flag_wait(uint32_t *flag, uint32_t at_least)
{
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < at_least) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void flag_raise(uint32_t *flag, uint32_t lane)  // by one: every wave of a job raises it once per step
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The whole step in ONE launch (mrl_step on one GPU, batches of one sub-block per workgroup: up to 262144 worlds):
// transition, look-back over the lower workgroups' finished counts instead of a kernel boundary, re-deal.  Workgroup
// b owns worlds [256 b, 256 b + 256) and has NINE waves: eight own 32 worlds each (phase A lane = world -- run by four
// LEADERS for their own and a partner's worlds, see below -- phase B bits -> bytes), the ninth -- the scan wave -- owns no
// world.  It waits for the leaders' counts, which they hand over behind the TRANSITION (a counter in LDS, round 4; until
// then: a barrier behind the whole of phase A), publishes the workgroup's count, looks back (recounting what does not
// appear, episode_scan.hpp) while the leaders encode, and deals ALL of the workgroup's finished worlds anew in their slots
// of the other waves' LDS (record + both agents' bit vectors) while the eight are streaming out their rows; each of them
// appends the rows of its own finished worlds to its stream of stores.  The serial part of the re-deal (look-back 2 us,
// ten dependent draws, encode) is in no stepping wave's instruction stream: measured with the re-deal in the stepping
// waves (the last of them looking back first), that wave started its rows 2 us late and ended the workgroup in 208 of
// 256 cases, and every wave with a finished world queued 2.8 us of re-deal behind its rows
// (profiles/r03_e_hanabi_fused_timeline.txt).  No workgroup barrier after the first: every hand-off inside the workgroup is
// a counter in LDS that its writer raises whatever the other waves do, so nobody waits for more than it needs.
// The two-launch pair stays for the RCCL form of the sharded path (the episode base comes from the other ranks in between;
// the mailbox form runs this kernel) and for larger batches.
constexpr int kFusedBlock = kBlock + kWave;
// Measurement variants (same sources, other flags: tools/hanabi_ab.sh).  0: every row store waits for the whole encode; 1: the
// rows' first line is stored by the partner wave while its leader still encodes; 2: lines 3 and 4 (card knowledge) as well.
// One box, 65536 worlds, us per step: 18.05-18.10 / 17.22-17.23 / 17.04-17.15 (profiles/r04_aa_hanabi_early_lines_ab.txt).
// (A waiting wave that sleeps 0.6 us before it starts polling its leader's counter changed nothing: r04_z.)
#ifndef MRL_HANABI_EARLY_LINE
#define MRL_HANABI_EARLY_LINE 2
#endif

// (nine waves = three on one of the four SIMDs: amdgpu_waves_per_eu(3) holds the kernel to 168 VGPRs)
template <int kV>
__global__ void __launch_bounds__(kFusedBlock) __attribute__((amdgpu_waves_per_eu(3)))
mrl_hanabi_step_fused(uint32_t *hot_records, const int32_t *hot_actions, uint32_t hot_num_worlds, uint32_t heal_mod, uint32_t pair_stride,
                      const HanabiParams p0,
                      unsigned long long *status, uint32_t epoch, const uint32_t *episode_base, uint32_t *next_counter,
                      uint32_t *reset_count, uint32_t *heal_seen, const mrl::DeviceCounter device_counter,
                      const mrl::FusedExchange fx)  // sharded batch: the other ranks' counts through the mailboxes (episode_scan.hpp)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[kWavesPerBlock * kWaveLdsBytes];
    __shared__ __attribute__((aligned(16))) uint8_t s_scratch[kWorldsPerWave * kRecStride];  // the scan wave's copy of another workgroup's records (healing)
    __shared__ uint32_t s_counts[kWavesPerBlock];
    __shared__ uint32_t s_ready;  // 1: the workgroup's count is globally visible, records may be overwritten; 2: the fresh games are in LDS as well
    __shared__ uint32_t s_moved, s_encoded;  // leaders past the transition (counts, ballots in LDS) / past the encode (bit vectors in LDS)
    __shared__ uint32_t s_leader_done[kWavesPerBlock];  // per leader: its own and its partner's bit vectors are in LDS
    __shared__ uint32_t s_line0[kWavesPerBlock];        // per leader: ... their first four words are (the rows' first lines can go)
    __shared__ uint8_t s_fin[kWavesPerBlock][kWorldsPerWave];
    __shared__ uint32_t s_overs[kWavesPerBlock], s_movers[kWavesPerBlock];  // per wave of worlds: finished / next mover is agent 1
    // the records are requested with the kernel's first instructions, from the preloaded arguments alone
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (pair_stride bit 8, experiment: workgroups 2k and 2k + 1 swap their worlds -- does a slow workgroup follow its XCD or its addresses?)
    const uint32_t bid = ((pair_stride >> 8) & 1u) && !(gridDim.x & 1u) ? blockIdx.x ^ 1u : blockIdx.x;
    pair_stride &= 0xFFu;
    const bool scan_wave = wib == (uint32_t)kWavesPerBlock;
    const uint32_t w0 = bid * kWorldsPerBlock + wib * kWorldsPerWave;
    const uint32_t nw = (!scan_wave && w0 < hot_num_worlds) ? min((uint32_t)kWorldsPerWave, hot_num_worlds - w0) : 0u;
    RecordLoads pending;
    request_records(hot_records, w0, nw, lane, pending);

    HanabiParams p = p0;
    p.records = hot_records;
    p.actions = hot_actions;
    p.num_worlds = hot_num_worlds;
    const mrl::HealTest heal{heal_mod, heal_seen};
    const uint32_t N = p.num_worlds;
    const bool last_block = bid == gridDim.x - 1;
    if (scan_wave || heal_mod) device_counter.apply(episode_base, next_counter, epoch);  // (the launch state may live in device memory: common.hpp)
    if (threadIdx.x == 0) {
        s_ready = 0u;
        s_moved = 0u;
        s_encoded = 0u;
    }
    if (threadIdx.x < (uint32_t)kWavesPerBlock) {
        s_leader_done[threadIdx.x] = 0u;
        s_line0[threadIdx.x] = 0u;
    }
    if (heal_mod) mrl::heal_test_delay(heal, bid, gridDim.x, epoch);  // test hook only (uniform branch on a preloaded argument)
    const bool paired = pair_stride != 0;
    const uint32_t leaders = paired ? (uint32_t)kWavesPerBlock / 2u : (uint32_t)kWavesPerBlock;

    if (scan_wave) {
        // ================= the scan wave =================
        // The count of finishing worlds is known as soon as the leaders are past the TRANSITION, well before their encode is
        // through: published and looked back for then, the hand-off through memory runs while the fabric is idle -- under the
        // row stores every poll is a round trip of 2 us and a workgroup that publishes late holds up all higher ones
        // (profiles/r04_v_hanabi_fused_timeline.txt: prefix known at 6.2 us in the median, 11.3 at p90, 13.2 at worst).
        const uint32_t base = *episode_base;  // requested now, needed after the look-back
        mrl::lds_barrier();  // the counters above are zero (and the stepping waves' records in LDS)
        flag_wait(&s_moved, leaders);  // the transition is through: s_counts and s_fin are there
        // (look-back and re-deal are a serial chain that two streaming waves share a SIMD with; not while it polls for the
        // leaders, one of which shares its SIMD: a polling wave of higher priority took 0.5 us out of that leader's transition)
        __builtin_amdgcn_s_setprio(3);
        FSTAMP_SCAN(0);
        uint32_t start_of[kWavesPerBlock + 1];
        start_of[0] = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; w++) start_of[w + 1] = start_of[w] + s_counts[w];
        const uint32_t block_total = start_of[kWavesPerBlock];
        if (lane == 0) mrl::publish_count(status, bid, epoch, block_total);
        uint32_t before = 0;
        if (block_total != 0 || last_block)
            before = mrl::wave_prefix_or_recount(status, bid, epoch, heal, [&](uint32_t j) { return recount_chunk<kV>(p, s_scratch, j); });
        // the count is globally visible before any wave of this workgroup overwrites a record (a healing
        // workgroup that does not see the count reads the records as the step's inputs)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&s_ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        FSTAMP_SCAN(1);
        uint32_t lower_ranks = 0, all_counts = before + block_total;
        if (fx.mail.num_ranks) {
            // a shard of a larger batch: the last workgroup tells every rank the shard's total; the ranks below come first in the
            // numbering, and the counter moves on by the sum over all ranks
            if (block_total != 0 || last_block) lower_ranks = mrl::fused_exchange(fx, last_block, before + block_total, fx.mail.rank);
            if (last_block) all_counts = mrl::fused_exchange(fx, false, 0u, fx.mail.num_ranks);
        }
        if (last_block && lane == 0) {
            *reset_count = before + block_total;
            *next_counter = base + all_counts;
        }
        flag_wait(&s_encoded, leaders);  // the leaders have written their last bit vector: the finished worlds' slots are free
        deal_finished_worlds<kV>(p, [&](uint32_t wv) { return wave_lds(smem, wv); }, s_fin, start_of, base + lower_ranks + before, lane);
        if (lane == 0) __hip_atomic_store(&s_ready, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        FSTAMP_SCAN(2);
        return;
    }

    // ================= the eight stepping waves =================
    // Phase A is one instruction stream whether 32 or 64 of a wave's lanes are worlds, and two waves of a SIMD running it
    // side by side take turns at the issue port.  So it is run by FOUR waves ("leaders") with all 64 lanes -- a leader's own
    // 32 worlds on lanes 0..31 and its partner wave's on lanes 32..63 -- while the partners sleep at the barrier: half the
    // instructions per SIMD, and the workgroup's first row store that much earlier.  pair_stride: 4 = waves (w, w + 4), the
    // two waves a SIMD gets when a workgroup's waves are dealt round-robin; 1 = waves (2k, 2k + 1); 0 = every wave for itself.
    const WaveLds l = wave_lds(smem, wib);
    FSTAMP(0);
    const bool leader = !paired || (pair_stride == 4 ? wib < 4 : (wib & 1u) == 0);
    const uint32_t half = paired ? lane >> 5 : 0u, idx = paired ? lane & 31u : lane;
    const uint32_t slot = wib + half * pair_stride;             // whose worlds this lane steps in phase A
    const uint32_t a_w0 = bid * kWorldsPerBlock + slot * kWorldsPerWave;
    const uint32_t a_nw = (leader && a_w0 < N) ? min((uint32_t)kWorldsPerWave, N - a_w0) : 0u;
    int32_t act0 = 0, act1 = 0;
    if (idx < a_nw && !p.sample) {
        act0 = p.actions[a_w0 + idx];
        act1 = p.actions[(size_t)N + a_w0 + idx];
    }
    place_records(l, nw, lane, pending);
    FSTAMP(1);
    mrl::lds_barrier();  // the partner's records are in LDS too, and the hand-off counters are zero

    // ---- phase A: act, encode the next mover ----
    // Two hand-offs, both through counters in LDS (no barrier: nobody waits for more than it needs).  Behind the TRANSITION
    // the finished worlds go to the scan wave, which publishes and looks back while the leaders encode; behind the ENCODE the
    // bit vectors go to the partner (per leader) and the finished worlds' slots to the scan wave's re-deal.
    if (leader) {
        const WaveLds ls = wave_lds(smem, slot);
        bool over = false, next_is_1 = false;
        Moved m{false, 0.f};
        uint8_t *rec = ls.rec + idx * kRecStride;
        const uint32_t world = a_w0 + idx;
        constexpr bool kOnRegisters = kV == 2 && MRL_HANABI_REG_TRANSITION;
        uint32_t r[kOnRegisters ? kRecordWords : 1];  // the record's words after the move, for the encoder
        if (idx < a_nw) {
            if constexpr (kOnRegisters) {
                const MovedFull mf = move_world_full<true>(p, rec, world, act0, act1, true, r);
                m = Moved{mf.over, mf.reward};
                next_is_1 = mf.next != 0;
            } else {
                m = move_world<kV>(p, rec, world, act0, act1, true);
                next_is_1 = (rec[R_CUR] & 1u) != 0;
            }
            over = m.over;
        }
        const unsigned long long all_overs = __ballot(over), all_movers = __ballot(next_is_1);
        // the half's share of the ballots (a wave for itself: the whole ballot is its "low half", idx = lane < 32 worlds)
        const uint32_t my_overs = (uint32_t)(half ? all_overs >> 32 : all_overs), my_movers = (uint32_t)(half ? all_movers >> 32 : all_movers);
        if (over) s_fin[slot][__popc(my_overs & ((1u << idx) - 1u))] = (uint8_t)idx;
        if (idx == 0) {
            s_counts[slot] = (uint32_t)__popc(my_overs);
            s_overs[slot] = my_overs;
            s_movers[slot] = my_movers;
        }
        flag_raise(&s_moved, lane);
        FSTAMP(4);
        if (idx < a_nw) {
            const uint32_t next = next_is_1 ? 1u : 0u;
            if constexpr (kV == 2 && MRL_HANABI_EARLY_LINE) {
                if (paired) {
                    // the rows' first line goes to the partner as soon as its 128 bits are in LDS (lane 0 is a world whenever
                    // the wave has any; LDS operations of a wave complete in order, so the flag follows the words)
                    const EarlyLineFlag hand_over{&s_line0[wib], lane
#ifdef MRL_DIAG
                                                  , p.stamps ? p.stamps + (size_t)(bid * kWavesPerBlock + wib) * 16 : nullptr
#endif
                    };
                    if constexpr (kOnRegisters) encode_record_full_t<false>(r, ls.enc + idx * 2 * kEncWords, next, hand_over);
                    else encode_agent_full_t<false>(rec, ls.enc + idx * 2 * kEncWords, next, hand_over);
                } else {
                    if constexpr (kOnRegisters) encode_record_full_t<false>(r, ls.enc + idx * 2 * kEncWords, next);
                    else encode_variant<kV>(p, rec, ls.enc + idx * 2 * kEncWords, next);
                }
            } else {
                if constexpr (kOnRegisters) encode_record_full_t<false>(r, ls.enc + idx * 2 * kEncWords, next);
                else encode_variant<kV>(p, rec, ls.enc + idx * 2 * kEncWords, next);
            }
            // (a finished world's next episode opens with agent 0 to move, sim.cpp:446-532: written here, so that every
            // ACTIVE word has one writer -- two waves' stores to one address are ordered by nothing short of a wait for the first)
            const uint32_t mover = over ? 0u : next;
            p.active[(size_t)mover * N + world] = 1;
            p.active[(size_t)(mover ^ 1u) * N + world] = 0;
            p.reward[world] = m.reward;
            p.reward[(size_t)N + world] = m.reward;
            p.done[world] = over ? 1 : 0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) {
            __hip_atomic_store(&s_leader_done[wib], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&s_encoded, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        FSTAMP(2);
    } else {
        FSTAMP(2);
        const uint32_t lslot = wib - pair_stride;  // its leader
        if constexpr (kV == 2 && MRL_HANABI_EARLY_LINE) {
            // While the leader encodes on: the first line of the rows of its worlds and of this wave's, 16 stores -- the fabric
            // has something to do a microsecond before phase A is through.
            const uint32_t l_w0 = bid * kWorldsPerBlock + lslot * kWorldsPerWave;
            const uint32_t l_nw = l_w0 < N ? min((uint32_t)kWorldsPerWave, N - l_w0) : 0u;
            if (l_nw != 0) {
                flag_wait(&s_line0[lslot], 1u);
                FSTAMP(11);
                // (the SIMD's arbiter prefers the older wave, and the leader hardly ever stalls: without this the partner's sixteen
                // stores were issued when the leader was through, profiles/r04_x_hanabi_fused_timeline.txt)
                __builtin_amdgcn_s_setprio(2);
                expand_lines<kV, 0u, 1u>(p, wave_lds(smem, lslot), l_nw, s_overs[lslot], s_movers[lslot],
                                         row_resource(p.rows + (size_t)l_w0 * kWorldBlock, l_nw * kWorldBlock), lane);
                expand_lines<kV, 0u, 1u>(p, l, nw, s_overs[wib], s_movers[wib], row_resource(p.rows + (size_t)w0 * kWorldBlock, nw * kWorldBlock), lane);
                __builtin_amdgcn_s_setprio(0);
                FSTAMP(4);
#if MRL_HANABI_EARLY_LINE >= 2
                flag_wait(&s_line0[lslot], 2u);  // lines 3 and 4: card knowledge
                FSTAMP(12);
                __builtin_amdgcn_s_setprio(2);
                expand_lines<kV, 3u, 5u>(p, wave_lds(smem, lslot), l_nw, s_overs[lslot], s_movers[lslot],
                                         row_resource(p.rows + (size_t)l_w0 * kWorldBlock, l_nw * kWorldBlock), lane);
                expand_lines<kV, 3u, 5u>(p, l, nw, s_overs[wib], s_movers[wib], row_resource(p.rows + (size_t)w0 * kWorldBlock, nw * kWorldBlock), lane);
                __builtin_amdgcn_s_setprio(0);
#endif
            }
        }
        flag_wait(&s_leader_done[lslot], 1u);  // its leader's encode of this wave's worlds (and its ballots) is in LDS
    }
    FSTAMP(3);
    const unsigned long long overs = s_overs[wib], movers = s_movers[wib];
    const uint32_t mine = (uint32_t)__popcll(overs);

    // ---- phase B: the movers' rows of the worlds that go on ----
    {
        const __amdgpu_buffer_rsrc_t out = row_resource(p.rows + (size_t)w0 * kWorldBlock, nw * kWorldBlock);
        if (kV == 2 && MRL_HANABI_EARLY_LINE && paired) {
#if MRL_HANABI_EARLY_LINE >= 2
            expand_lines<kV, 1u, 3u>(p, l, nw, (uint32_t)overs, (uint32_t)movers, out, lane);
            expand_lines<kV, 5u, 7u>(p, l, nw, (uint32_t)overs, (uint32_t)movers, out, lane);
#else
            expand_lines<kV, 1u, 7u>(p, l, nw, (uint32_t)overs, (uint32_t)movers, out, lane);
#endif
        } else {
            expand_movers<kV>(p, l, nw, (uint32_t)overs, (uint32_t)movers, out, lane);
        }
    }
    FSTAMP(5);
    // ---- rows of this wave's finished worlds, dealt anew by the scan wave meanwhile; then all 32 records at once ----
    // (s_ready 1: the workgroup's count is public, records may be overwritten; 2: the fresh games are in LDS too)
    const uint32_t need = mine != 0 ? 2u : 1u;
    while (__hip_atomic_load(&s_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    store_fresh_rows(p, l, s_fin[wib], mine, w0, lane);
    store_records(p, l, w0, nw, lane);
    FSTAMP(6);
}

// ---------------------------------------------------------------------------------------------
// mrl_rollout_random in ONE launch: num_steps steps of the
// masked-random policy with the game records resident in LDS.  Every step still writes the
// mover's rows, rewards, dones and the ACTION tensor, and episodes are numbered exactly as by
// the one-launch-per-step path, which takes two grid-wide hand-offs per step:
//   * finished counts of the LOWER workgroups in this step (waited for);
//   * finished counts of ALL workgroups in the previous step (published a whole step earlier),
//     so that every workgroup knows the step's first episode index without a counter in HBM.
// Counts travel through a ring of four status arrays tagged with the step's epoch; a workgroup
// can run at most one step ahead of the slowest one (it needs everybody's previous-step count),
// so a slot is never overwritten while somebody still reads it.  Unlike the single step this
// kernel NEEDS all its workgroups resident at once (nobody exits before the last step): the host
// launches it cooperatively (the runtime refuses a grid the device cannot hold at once) and falls
// back to one launch per step when that fails.  Waits are bounded as everywhere (SCAN_TIMEOUT).
//
// Thirteen waves with fixed jobs, coupled by three counters in LDS instead of barriers, so that phase A of step
// k + 1 runs while the rows of step k are being stored:
//   waves 0..3   phase A for 64 worlds each (slots w and w + 4), into bit-vector buffer k & 1;
//   waves 4..11  phase B, 32 worlds each: the movers' rows, then the finished worlds' new rows.  (Eight of them:
//                a wave expands and stores one 1 KB round per 0.33 us whatever the others do -- four waves with
//                64 worlds each took 17 us over a step's rows, eight take 9.3.);
//   wave  12     the scan wave: both hand-offs, and deals the finished worlds anew in LDS.
// Step k:  A waits for dealt >= k (the records hold step k - 1's new games) and b_done >= 8 (k - 1) (phase B of
// step k - 2 has read buffer k & 1); the scan wave and B wait for a_done >= 4 (k + 1); B's new rows wait for
// dealt >= k + 1.  A counter is raised after the LDS traffic it announces (s_waitcnt lgkmcnt(0)); every raise is
// reached whatever the other waves do (the scan wave's waits on other workgroups are the bounded ones), and the
// dependencies above have no cycle.
// Measured, 65536 worlds, us per step: the re-deal in the stepping waves, two barriers 20.3 (as long as a launch per
// step); the single step's workgroup (leaders, scan wave, barriers between the phases) 15.3, HBM idle during every
// phase A; this one 13.1.  What bounds it now is the chain  phase A 3.4 -> look-back -> deal 1.8 -> next phase A:
// the look-back waits 8 us for the slowest LOWER workgroup's count (a workgroup whose stores are taken late is late
// with everything), twice what the loads themselves cost (tools/stamps_hanabi_rollout.py).
// ---------------------------------------------------------------------------------------------
constexpr int kRing = 4;
constexpr int kSlotRecBytes = kWorldsPerWave * kRecStride, kSlotEncBytes = kWorldsPerWave * 2 * kEncWords * 4;
constexpr int kRolloutLdsBytes = kWavesPerBlock * (kSlotRecBytes + 2 * kSlotEncBytes);
static_assert(kRolloutLdsBytes + 2048 <= 160 * 1024, "records + two bit-vector buffers fit a CU's LDS");
constexpr int kRolloutBlock = 13 * kWave;  // 4 phase-A waves, 8 phase-B waves, the scan wave

__device__ __forceinline__ WaveLds rollout_lds(uint8_t *smem, uint32_t slot, uint32_t buf)
{
    return WaveLds{smem + slot * kSlotRecBytes,
                   reinterpret_cast<uint32_t *>(smem + kWavesPerBlock * kSlotRecBytes + (buf * kWavesPerBlock + slot) * kSlotEncBytes)};
}

template <int kV>
__global__ void __launch_bounds__(kRolloutBlock) __attribute__((amdgpu_waves_per_eu(4)))
mrl_hanabi_rollout(const HanabiParams p0, unsigned long long *ring, uint32_t epoch0, uint32_t num_steps, uint32_t first_step,
                   const uint32_t *episode_base, uint32_t *next_counter, uint32_t *reset_count, const mrl::Alarm timed_out)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[kRolloutLdsBytes];
    __shared__ uint32_t s_counts[2][kWavesPerBlock], s_overs[2][kWavesPerBlock], s_movers[2][kWavesPerBlock];  // by step parity, like the bit vectors
    __shared__ uint8_t s_fin[2][kWavesPerBlock][kWorldsPerWave];
    __shared__ uint32_t s_a_done, s_dealt, s_b_done, s_a_moved;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    HanabiParams p = p0;
    const uint32_t N = p.num_worlds, G = gridDim.x, b = blockIdx.x;
    const bool a_wave = wib < 4, scan_wave = wib == 12;  // the eight in between: phase B, slot wib - 4
    if (threadIdx.x == 0) {
        s_a_done = 0u;
        s_dealt = 0u;
        s_b_done = 0u;
        s_a_moved = 0u;
    }
    const uint32_t b_slot = (wib - 4u) & 7u;
    const uint32_t w0 = b * kWorldsPerBlock + b_slot * kWorldsPerWave;
    const uint32_t nw = (!a_wave && !scan_wave && w0 < N) ? min((uint32_t)kWorldsPerWave, N - w0) : 0u;
    load_records(p, rollout_lds(smem, b_slot, 0), w0, nw, lane);
    mrl::lds_barrier();  // the only barrier: records and counters are in LDS

    if (scan_wave) {
        // ================= the scan wave =================
        __builtin_amdgcn_s_setprio(3);  // (it and phase A are the chain that bounds a step; phase B shares their SIMDs)
        uint32_t base = *episode_base;  // first episode index of the current step
        for (uint32_t k = 0; k < num_steps; k++) {
            const uint32_t epoch = epoch0 + k, buf = k & 1u;
            unsigned long long *now = ring + (size_t)(epoch % kRing) * G;
            const unsigned long long *before_step = ring + (size_t)((epoch - 1u) % kRing) * G;
            uint32_t lower = 0, prev_all = 0, unused = 0;
            if (k > 0)  // everybody's count of the previous step (published long ago)
                for (uint32_t first = 0; first < G; first += kWave * 8u)
                    prev_all += mrl::read_counts<8>(before_step, first, G, epoch - 1u, 0u, &unused, timed_out);
            RSTAMP(0, 8);
#ifdef MRL_HANABI_ROLLOUT_LATE_COUNT  // measurement variant: the counts leave when phase A is through, as until round 4
            flag_wait(&s_a_done, 4u * (k + 1u));
#else
            flag_wait(&s_a_moved, 4u * (k + 1u));  // the transition is through (the encode is not): the counts can travel meanwhile
#endif
            RSTAMP(0, 9);
            uint32_t start_of[kWavesPerBlock + 1];
            start_of[0] = 0;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; w++) start_of[w + 1] = start_of[w] + s_counts[buf][w];
            const uint32_t block_total = start_of[kWavesPerBlock];
            if (lane == 0) mrl::publish_count(now, b, epoch, block_total);
            if (block_total != 0)  // the lower workgroups' counts of this step
                for (uint32_t first = 0; first < b; first += kWave * 8u) lower += mrl::read_counts<8>(now, first, b, epoch, 0u, &unused, timed_out);
            for (int off = 32; off > 0; off >>= 1) {
                lower += __shfl_xor(lower, off, 64);
                prev_all += __shfl_xor(prev_all, off, 64);
            }
            base += prev_all;  // previous step's finished worlds, all workgroups (0 in the first step)
            RSTAMP(0, 10);
            flag_wait(&s_a_done, 4u * (k + 1u));  // the phase-A waves have written their last bit vector: the finished worlds' slots are free
            deal_finished_worlds<kV>(p, [&](uint32_t wv) { return rollout_lds(smem, wv, buf); }, s_fin[buf], start_of, base + lower, lane);
            flag_raise(&s_dealt, lane);
            RSTAMP(0, 11);
        }
        // the counter after the rollout: the last step's counts of everybody (the last workgroup has the
        // highest index, so these are "lower" counts plus its own and the usual wait applies)
        if (b == G - 1 && num_steps > 0) {
            const uint32_t epoch = epoch0 + num_steps - 1u;
            const unsigned long long *now = ring + (size_t)(epoch % kRing) * G;
            uint32_t all = 0, unused = 0;
            for (uint32_t first = 0; first < G; first += kWave * 8u) all += mrl::read_counts<8>(now, first, G, epoch, 0u, &unused, timed_out);
            for (int off = 32; off > 0; off >>= 1) all += __shfl_xor(all, off, 64);
            if (lane == 0) {
                *reset_count = all;
                *next_counter = base + all;
            }
        }
        return;
    }

    if (a_wave) {
        // ================= phase A: draw, act, encode the next mover =================
        __builtin_amdgcn_s_setprio(3);
        const uint32_t half = lane >> 5, idx = lane & 31u, slot = wib + 4u * half;
        const uint32_t a_w0 = b * kWorldsPerBlock + slot * kWorldsPerWave, a_nw = a_w0 < N ? min((uint32_t)kWorldsPerWave, N - a_w0) : 0u;
        for (uint32_t k = 0; k < num_steps; k++) {
            const uint32_t buf = k & 1u;
            RSTAMP(wib, 0);
            if (k >= 1) flag_wait(&s_dealt, k);
            RSTAMP(wib, 1);
            if (k >= 2) flag_wait(&s_b_done, 8u * (k - 1u));
            RSTAMP(wib, 2);
            const WaveLds ls = rollout_lds(smem, slot, buf);
            p.sample_step = first_step + k;
            bool over = false, next_is_1 = false;
            Moved m{false, 0.f};
            uint8_t *rec = ls.rec + idx * kRecStride;
            const uint32_t world = a_w0 + idx;
            constexpr bool kOnRegisters = kV == 2 && MRL_HANABI_REG_TRANSITION;
            uint32_t r[kOnRegisters ? kRecordWords : 1];  // the record's words after the move, for the encoder
            if (idx < a_nw) {
                if constexpr (kOnRegisters) {
                    const MovedFull mf = move_world_full<false>(p, rec, world, 0, 0, true, r);
                    m = Moved{mf.over, mf.reward};
                    next_is_1 = mf.next != 0;
                } else {
                    m = move_world<kV>(p, rec, world, 0, 0, true);
                    next_is_1 = (rec[R_CUR] & 1u) != 0;
                }
                over = m.over;
            }
            const unsigned long long all_overs = __ballot(over), all_movers = __ballot(next_is_1);
            const uint32_t my_overs = (uint32_t)(half ? all_overs >> 32 : all_overs), my_movers = (uint32_t)(half ? all_movers >> 32 : all_movers);
            if (over) s_fin[buf][slot][__popc(my_overs & ((1u << idx) - 1u))] = (uint8_t)idx;
            if (idx == 0) {
                s_counts[buf][slot] = (uint32_t)__popc(my_overs);
                s_overs[buf][slot] = my_overs;
                s_movers[buf][slot] = my_movers;
            }
            flag_raise(&s_a_moved, lane);  // the scan wave publishes and looks back while this wave encodes (as in the single step)
            if (idx < a_nw) {
                const uint32_t next = next_is_1 ? 1u : 0u;
                // (the encoder reads the record again here: 31 words kept across the hand-off push this kernel, thirteen waves at 128
                // registers each, into scratch memory)
                encode_variant<kV>(p, rec, ls.enc + idx * 2 * kEncWords, next);
                const uint32_t mover = over ? 0u : next;  // (a new game opens with agent 0: one writer per ACTIVE word)
                p.active[(size_t)mover * N + world] = 1;
                p.active[(size_t)(mover ^ 1u) * N + world] = 0;
                p.reward[world] = m.reward;
                p.reward[(size_t)N + world] = m.reward;
                p.done[world] = over ? 1 : 0;
            }
            flag_raise(&s_a_done, lane);
            RSTAMP(wib, 3);
        }
        return;
    }

    // ================= phase B: the movers' rows of the worlds that go on; then the finished worlds' new rows =================
    const __amdgpu_buffer_rsrc_t out = row_resource(p.rows + (size_t)w0 * kWorldBlock, nw * kWorldBlock);
    for (uint32_t k = 0; k < num_steps; k++) {
        const uint32_t buf = k & 1u;
        const WaveLds l = rollout_lds(smem, b_slot, buf);
        RSTAMP(b_slot, 4);
        flag_wait(&s_a_done, 4u * (k + 1u));
        RSTAMP(b_slot, 5);
        const uint32_t overs = s_overs[buf][b_slot];
        expand_movers<kV>(p, l, nw, overs, s_movers[buf][b_slot], out, lane);
        RSTAMP(b_slot, 6);
        if (overs != 0) {
            flag_wait(&s_dealt, k + 1u);
            store_fresh_rows(p, l, s_fin[buf][b_slot], (uint32_t)__popc(overs), w0, lane);
        }
        flag_raise(&s_b_done, lane);
        RSTAMP(b_slot, 7);
    }
    flag_wait(&s_dealt, num_steps);  // the last step's new games are in the records
    store_records(p, rollout_lds(smem, b_slot, 0), w0, nw, lane);
}

__global__ void fill_agent_ids(int32_t *world_id, int32_t *agent_id, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * n) {
        world_id[i] = (int32_t)(i % n);
        agent_id[i] = (int32_t)(i / n);
    }
}

struct HanabiSim final : mrl_sim {
    HanabiParams params{};
    uint32_t grid = 0;
    int32_t *action = nullptr, *world_id = nullptr, *agent_id = nullptr;
    uint32_t *counter = nullptr, *reset_count = nullptr;
    uint32_t parity = 0;
    int variant = 0;  // code variant of the kernels (see encode_variant)
    // single-launch step (mrl_hanabi_step_fused)
    unsigned long long *status = nullptr;
    mrl::AlarmOwner alarm;
    mrl::HealTest heal;  // test hook of the healing look-back (mrl_debug_set fused_heal_test)
    uint32_t pair_stride = 4;  // phase A of the single-launch step by four leader waves (see the kernel); mrl_debug_set hanabi.pairing
    uint32_t epoch = 0;
    bool fused = false;
    mrl::LaunchStateOwner launch_state;  // parity / epoch in device memory once a caller wants to capture steps (common.hpp)
    bool scan_timed_out() const override { return alarm.raised(); }
    bool capturable() const override { return launch_state.device_mode; }
    void prepare_graph_capture(hipStream_t stream) override { launch_state.to_device(parity, epoch, stream); }

    void step(const int32_t *actions, hipStream_t stream) override
    {
        if (!fused) {
            mrl_sim::step(actions, stream);
            return;
        }
        launch_fused(actions, stream, mrl::FusedExchange{});
    }
    // a shard's step with the other ranks' counts taken from the mailboxes inside the single launch (episode_scan.hpp)
    void step_exchanged(const int32_t *actions, hipStream_t stream) override
    {
        if (!fused) {
            mrl_sim::step_exchanged(actions, stream);
            return;
        }
        launch_fused(actions, stream, mrl::fused_exchange_of(exchange, alarm.alarm()));
    }
    void launch_fused(const int32_t *actions, hipStream_t stream, const mrl::FusedExchange &fx)
    {
        HanabiParams a = params;
        a.actions = actions ? actions : action;
        epoch += 1;
        const uint32_t *base = counter + parity;
        uint32_t *next = counter + (parity ^ 1u);
        if (launch_state.device_mode) launch_state.advance(stream);  // then parity / epoch come from device memory
        const mrl::DeviceCounter dc = launch_state.counter_args(counter);
        switch (variant) {
        case 2: hipLaunchKernelGGL((mrl_hanabi_step_fused<2>), dim3(grid), dim3(kFusedBlock), 0, stream, a.records, a.actions, a.num_worlds, heal.mod, pair_stride, a, status, epoch, base, next, reset_count, heal.seen, dc, fx); break;
        case 1: hipLaunchKernelGGL((mrl_hanabi_step_fused<1>), dim3(grid), dim3(kFusedBlock), 0, stream, a.records, a.actions, a.num_worlds, heal.mod, pair_stride, a, status, epoch, base, next, reset_count, heal.seen, dc, fx); break;
        default: hipLaunchKernelGGL((mrl_hanabi_step_fused<0>), dim3(grid), dim3(kFusedBlock), 0, stream, a.records, a.actions, a.num_worlds, heal.mod, pair_stride, a, status, epoch, base, next, reset_count, heal.seen, dc, fx); break;
        }
        MRL_HIP(hipGetLastError());
        parity ^= 1u;
    }

    unsigned long long *ring = nullptr;
    uint32_t ring_epoch = 0;
    bool persistent_ok = false;  // the whole grid of mrl_hanabi_rollout is resident at once

    template <int kV> bool launch_rollout(uint32_t num_steps, uint64_t seed, uint32_t first_step, hipStream_t stream)
    {
        HanabiParams a = params;
        a.sample = 1;
        a.sample_seed = seed;
        a.action_out = action;
        // cooperative: the runtime checks the grid against what the device can hold at once
        uint32_t epoch0 = ring_epoch + 1u;
        const uint32_t *base = counter + parity;
        uint32_t *next = counter + (parity ^ 1u);
        mrl::Alarm al = alarm.alarm();
        void *args[] = {&a, &ring, &epoch0, &num_steps, &first_step, &base, &next, &reset_count, &al};
        const hipError_t err = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(&mrl_hanabi_rollout<kV>), dim3(grid), dim3(kRolloutBlock),
                                                          args, 0, stream);
        if (err != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        ring_epoch += num_steps;
        parity ^= 1u;
        return true;
    }

    void rollout_random(uint32_t num_steps, uint64_t seed, uint32_t first_step, hipStream_t stream) override
    {
        if (num_steps == 0) return;
        if (persistent_ok && !launch_state.device_mode) {  // (a cooperative launch cannot be captured; its counters live on the host)
            bool launched;
            switch (variant) {
            case 2: launched = launch_rollout<2>(num_steps, seed, first_step, stream); break;
            case 1: launched = launch_rollout<1>(num_steps, seed, first_step, stream); break;
            default: launched = launch_rollout<0>(num_steps, seed, first_step, stream); break;
            }
            if (launched) return;
            persistent_ok = false;  // refused: one launch per step from now on (needs no co-residency)
        }
        const HanabiParams saved = params;
        params.sample = 1;
        params.sample_seed = seed;
        params.action_out = action;
        for (uint32_t k = 0; k < num_steps; k++) {
            params.sample_step = first_step + k;
            step(nullptr, stream);
        }
        params = saved;
    }

    void phase1(const int32_t *actions, hipStream_t stream) override
    {
        HanabiParams a = params;
        a.actions = actions ? actions : action;
        switch (variant) {
        case 2: hipLaunchKernelGGL((mrl_hanabi_step<2>), dim3(grid), dim3(kBlock), 0, stream, a.records, a.actions, a.num_worlds, a.chunk, a); break;
        case 1: hipLaunchKernelGGL((mrl_hanabi_step<1>), dim3(grid), dim3(kBlock), 0, stream, a.records, a.actions, a.num_worlds, a.chunk, a); break;
        default: hipLaunchKernelGGL((mrl_hanabi_step<0>), dim3(grid), dim3(kBlock), 0, stream, a.records, a.actions, a.num_worlds, a.chunk, a); break;
        }
        MRL_HIP(hipGetLastError());
    }

    void launch_reset(const uint32_t *base, const mrl::GatheredCounts &gathered, hipStream_t stream, bool external_base = false)
    {
        if (launch_state.device_mode) launch_state.advance(stream);
        const mrl::DeviceCounter dc = launch_state.counter_args(counter, external_base);
        switch (variant) {
        case 2:
            hipLaunchKernelGGL((mrl_hanabi_reset<false, 2>), dim3(grid), dim3(kBlock), 0, stream, params, base, 0u,
                               counter + (parity ^ 1u), reset_count, gathered, dc);
            break;
        case 1:
            hipLaunchKernelGGL((mrl_hanabi_reset<false, 1>), dim3(grid), dim3(kBlock), 0, stream, params, base, 0u,
                               counter + (parity ^ 1u), reset_count, gathered, dc);
            break;
        default:
            hipLaunchKernelGGL((mrl_hanabi_reset<false, 0>), dim3(grid), dim3(kBlock), 0, stream, params, base, 0u,
                               counter + (parity ^ 1u), reset_count, gathered, dc);
            break;
        }
        MRL_HIP(hipGetLastError());
        parity ^= 1u;
    }
    void publish_shard_count(hipStream_t stream) override
    {
        hipLaunchKernelGGL(mrl::sum_block_counts, dim3(1), dim3(256), 0, stream, params.block_counts, grid, params.shard_count, mrl::mail_of(exchange));
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
        const uint32_t *none = nullptr;
        uint32_t *no_out = nullptr;
        switch (variant) {
        case 2: hipLaunchKernelGGL((mrl_hanabi_reset<true, 2>), dim3(grid), dim3(kBlock), 0, stream, params, none, world_offset, no_out, no_out, mrl::GatheredCounts{}, mrl::DeviceCounter{}); break;
        case 1: hipLaunchKernelGGL((mrl_hanabi_reset<true, 1>), dim3(grid), dim3(kBlock), 0, stream, params, none, world_offset, no_out, no_out, mrl::GatheredCounts{}, mrl::DeviceCounter{}); break;
        default: hipLaunchKernelGGL((mrl_hanabi_reset<true, 0>), dim3(grid), dim3(kBlock), 0, stream, params, none, world_offset, no_out, no_out, mrl::GatheredCounts{}, mrl::DeviceCounter{}); break;
        }
        MRL_HIP(hipGetLastError());
        set_episode_counter(num_worlds_total, stream);
    }

    bool tensor(int slot, mrl_tensor_desc *out) override
    {
        const int64_t N = num_worlds;
        switch (slot) {
        case MRL_HANABI_DONE: *out = mrl::make_desc(params.done, MRL_INT32, device, {N}); return true;
        case MRL_HANABI_ACTIVE_AGENT: *out = mrl::make_desc(params.active, MRL_INT32, device, {2, N}); return true;
        case MRL_HANABI_ACTION: *out = mrl::make_desc(action, MRL_INT32, device, {2, N, 1}); return true;
        case MRL_HANABI_OBSERVATION:
            // the head of the state row: the reference fills the state by copying the observation (see kAgentBlock).  As
            // wide as THIS configuration's observation (658 for the full game): what follows in the row is the agent's own
            // hand, which an observation must not show (the reference's buffer holds nothing there)
            *out = mrl::make_desc(params.rows, MRL_INT8, device, {2, N, (int64_t)params.obs_bits}, {kAgentBlock, kWorldBlock, 1});
            return true;
        case MRL_HANABI_ACTION_MASK:
            *out = mrl::make_desc(params.rows + kStateRow, MRL_INT32, device, {2, N, 20}, {kAgentBlock / 4, kWorldBlock / 4, 1});
            return true;
        case MRL_HANABI_REWARD: *out = mrl::make_desc(params.reward, MRL_FLOAT32, device, {2, N}); return true;
        case MRL_HANABI_WORLD_ID: *out = mrl::make_desc(world_id, MRL_INT32, device, {2, N}); return true;
        case MRL_HANABI_AGENT_ID: *out = mrl::make_desc(agent_id, MRL_INT32, device, {2, N}); return true;
        case MRL_HANABI_STATE:
            *out = mrl::make_desc(params.rows, MRL_INT8, device, {2, N, MRL_HANABI_STATE_SIZE}, {kAgentBlock, kWorldBlock, 1});
            return true;
        case MRL_HANABI_GAME: *out = mrl::make_desc(params.records, MRL_UINT8, device, {N, kRecordBytes}); return true;
        case MRL_HANABI_RESET_COUNT: *out = mrl::make_desc(reset_count, MRL_UINT32, device, {1}); return true;
        case MRL_HANABI_SCAN_TIMEOUT: *out = mrl::make_desc(alarm.alarm().dev, MRL_UINT32, device, {1}); return true;
        case MRL_HANABI_SHARD_COUNT: *out = mrl::make_desc(params.shard_count, MRL_UINT32, device, {1}); return true;
#ifdef MRL_DIAG
        case 14:
            if (!params.stamps) return false;
            *out = mrl::make_desc(params.stamps, MRL_UINT8, device, {(int64_t)grid * kWavesPerBlock * 16 * 8});
            return true;
#endif
        default: return false;
        }
    }

    size_t action_elems() const override { return (size_t)2 * num_worlds; }
    const char *kernel_name() const override { return fused ? "mrl_hanabi_step_fused" : "mrl_hanabi_step"; }
    const char *rollout_kernel_name() const override { return persistent_ok && !launch_state.device_mode ? "mrl_hanabi_rollout" : kernel_name(); }

    uint64_t bytes_per_world_step() const override
    {
        // SURVEY.md section 8d's list with this engine's sizes: action 8 + record r/w 2*176 + state 783 (its head IS the
        // observation: the reference's second copy of those 658 bytes is a view here) + mask 80 + active 8 + reward 8 +
        // done 4 (non-reset step: one agent re-encoded)
        return 8 + 2 * kRecordBytes + 783 + 80 + 8 + 8 + 4;
    }
};

}  // namespace

mrl_sim *mrl::create_hanabi(const mrl_hanabi_config *cfg, int gpu_id, uint32_t num_worlds)
{
    if (!cfg) {
        set_error("hanabi: null config");
        throw HipError{MRL_ERR_INVALID};
    }
    if (cfg->players != 2) {
        set_error("hanabi: players must be 2 (the reference's tensors are compiled for N_PLAYERS=2, "
                  "src/hanabi_env/sim.hpp:23), got %u",
                  cfg->players);
        throw HipError{MRL_ERR_INVALID};
    }
    if (cfg->colors < 1 || cfg->colors > 5 || cfg->ranks < 2 || cfg->ranks > 5 || cfg->max_information_tokens < 1 ||
        cfg->max_information_tokens > 8 || cfg->max_life_tokens < 1 || cfg->max_life_tokens > 3) {
        set_error("hanabi: need 1..5 colors, 2..5 ranks, 1..8 information tokens, 1..3 life tokens");
        throw HipError{MRL_ERR_INVALID};
    }
    if (num_worlds == 0) {
        set_error("hanabi: num_worlds must be > 0");
        throw HipError{MRL_ERR_INVALID};
    }
    bind_device(gpu_id);
    auto *sim = new HanabiSim();
    try {
        sim->game = MRL_GAME_HANABI;
        sim->device = gpu_id;
        sim->num_worlds = num_worlds;
        {
            const uint32_t groups = (num_worlds + kWorldsPerBlock - 1) / kWorldsPerBlock;
            const uint32_t blocks = groups < mrl::kMaxScanBlocks ? groups : mrl::kMaxScanBlocks;
            sim->params.chunk = ((groups + blocks - 1) / blocks) * kWorldsPerBlock;
            sim->grid = (num_worlds + sim->params.chunk - 1) / sim->params.chunk;
        }
        HanabiParams &a = sim->params;
        const uint32_t K = cfg->colors, R = cfg->ranks, N = num_worlds;
        a.num_worlds = N;
        a.colors = K;
        a.ranks = R;
        a.max_info = cfg->max_information_tokens;
        a.max_life = cfg->max_life_tokens;
        a.bpc = K * R;
        sim->variant = (K == 5 && R == 5 && a.max_info == 8 && a.max_life == 3) ? 2 : (R == 5 ? 1 : 0);
        sim->variant = std::min(sim->variant, (int)mrl::debug_get("hanabi.variant", 2));  // tests: force the generic encoders
        a.max_deck = (4 + (R - 2) * 2) * K - 2 * kHand;
        a.off_flags = kHand * a.bpc;
        a.off_deck = a.off_flags + 2;
        a.off_fireworks = a.off_deck + a.max_deck;
        a.off_info = a.off_fireworks + K * R;
        a.off_life = a.off_info + a.max_info;
        a.off_discard = a.off_life + a.max_life;
        a.off_last = a.off_discard + 2 * R * K;
        a.off_know = a.off_last + (2 + 4 + 2 + K + R + 2 * kHand + a.bpc + 2);
        a.obs_bits = a.off_know + 2 * kHand * (a.bpc + K + R);
        a.state_bits = a.obs_bits + kHand * a.bpc;
        if ((int32_t)a.max_deck < 0 || a.state_bits > 780 + 3) {
            set_error("hanabi: configuration does not leave a deck after dealing two hands");
            throw HipError{MRL_ERR_INVALID};
        }
        a.records = sim->arena.alloc<uint32_t>((size_t)N * kRecordWords);
        a.rows = sim->arena.alloc<uint8_t>((size_t)N * kWorldBlock);
        {
            uint8_t fresh[52] = {0};
            uint32_t k = 0;
            for (uint32_t c = 0; c < K; c++)
                for (uint32_t r = 0; r < R; r++) {
                    const uint32_t copies = 2u + (r == 0 ? 1u : 0u) - (r == R - 1 ? 1u : 0u);  // 3, 2.., 1
                    for (uint32_t i = 0; i < copies; i++) fresh[R_DECK + k++] = (uint8_t)(R * c + r);
                }
            fresh[R_DECK_SIZE] = (uint8_t)k;
            memcpy(a.deck_words, fresh, sizeof(fresh));
        }
        a.active = sim->arena.alloc<int32_t>((size_t)2 * N);
        a.reward = sim->arena.alloc<float>((size_t)2 * N);
        a.done = sim->arena.alloc<int32_t>(N);
        a.block_counts = sim->arena.alloc<uint32_t>(sim->grid);
        a.shard_count = sim->arena.alloc<uint32_t>(1);
#ifdef MRL_DIAG
        a.ablate = (uint32_t)mrl::debug_get("ablate", 0);
        a.stamps = mrl::debug_get("stamps", 0) ? sim->arena.alloc<unsigned long long>((size_t)sim->grid * kWavesPerBlock * 16) : nullptr;
#endif
        sim->action = sim->arena.alloc<int32_t>((size_t)2 * N);
        sim->world_id = sim->arena.alloc<int32_t>((size_t)2 * N, false);
        sim->agent_id = sim->arena.alloc<int32_t>((size_t)2 * N, false);
        sim->counter = sim->arena.alloc<uint32_t>(2);
        sim->reset_count = sim->arena.alloc<uint32_t>(1);
        sim->alarm.init(sim->arena);
        sim->launch_state.init(sim->arena);
        sim->status = sim->arena.alloc<unsigned long long>(sim->grid);
        {
            // mrl_debug_set fused_step: 0 = the library's choice (one launch whenever a workgroup owns one sub-block),
            // 1 = one launch where possible, 2 = always two
            const int64_t knob = mrl::debug_get("fused_step", 0);
            sim->fused = knob != 2 && sim->params.chunk == (uint32_t)kWorldsPerBlock;
            sim->heal.mod = (uint32_t)mrl::debug_get("fused_heal_test", 0);
            sim->heal.seen = sim->arena.alloc<uint32_t>(sim->grid);
            const int64_t pairing = mrl::debug_get("hanabi.pairing", 4);  // 4: waves (w, w + 4), 1: (2k, 2k + 1), 0: no leaders
            sim->pair_stride = (pairing & 0xFF) == 1 ? 1u : (pairing & 0xFF) == 0 ? 0u : 4u;
            sim->pair_stride |= (uint32_t)(pairing & 0x100);  // experiment: see the kernel
        }
        {
            // mrl_hanabi_rollout keeps every workgroup alive for the whole rollout and they wait for each
            // other: only usable when the grid fits the GPU in one go and each workgroup owns one sub-block
            int per_cu = 0, cus = 0;
            const void *fn = sim->variant == 2 ? reinterpret_cast<const void *>(&mrl_hanabi_rollout<2>)
                             : sim->variant == 1 ? reinterpret_cast<const void *>(&mrl_hanabi_rollout<1>)
                                                 : reinterpret_cast<const void *>(&mrl_hanabi_rollout<0>);
            MRL_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kRolloutBlock, 0));
            MRL_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, gpu_id));
            // (the occupancy query can be one workgroup per CU too high, MI355X_MICROARCH.md "Residency and
            // cooperative launch": keep one per CU in hand near the edge; the cooperative launch is the check)
            const int usable = per_cu > 4 ? per_cu - 1 : per_cu;
            sim->persistent_ok = !mrl::debug_get("hanabi.no_persistent", 0) && sim->params.chunk == (uint32_t)kWorldsPerBlock &&
                                 (uint64_t)sim->grid <= (uint64_t)usable * (uint64_t)cus;
            sim->ring = sim->arena.alloc<unsigned long long>((size_t)kRing * sim->grid);
        }
        hipLaunchKernelGGL(fill_agent_ids, dim3((2 * N + 255) / 256), dim3(256), 0, 0, sim->world_id, sim->agent_id, N);
        MRL_HIP(hipGetLastError());
        sim->reseed_shard(0, N, 0);
        if (mrl::debug_get("inject_scan_timeout", 0)) hipLaunchKernelGGL(mrl::raise_alarm_kernel, dim3(1), dim3(1), 0, 0, sim->alarm.alarm());
        MRL_HIP(hipDeviceSynchronize());
    } catch (...) {
        delete sim;
        throw;
    }
    return sim;
}
