/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle, 2-player Hanabi world step.
 *
 * Restates /root/reference/src/hanabi_env/sim.cpp:
 *   :45-52 drawDeck, :54-90 encodeHands, :92-135 encodeBoard, :137-156
 *   encodeDiscards, :158-289 encodeLastAction, :291-331 encodeCardKnowledge,
 *   :333-365 state tail, :381-444 legal-move mask, :446-532 resetWorld,
 *   :567-594 removeFromHand, :596-792 actionSystem, :794-810 observationSystem,
 *   :812-850 checkDone, :852-863 graph order, and rng.hpp:5-40
 * including the behaviours a straight reading would "fix" (DESIGN.md lists them):
 *   - card-knowledge plausibility bits test bit <player-loop-index> of the
 *     mask for all 25 positions (sim.cpp:311),
 *   - only the player to move gets fresh obs/state/mask, the other keeps the
 *     stale buffers (sim.cpp:799-808),
 *   - hint legality scans all five hand slots whatever the hand size, so the
 *     duplicate a left-shift leaves in the last slot counts (sim.cpp:416-417),
 *   - the rank hint's "newly revealed" test looks at known_color (sim.cpp:776),
 *   - no legality check on the action (sim.cpp:604).
 *
 * PARITY UNPINNED for: card-knowledge section, last-action section, the RNG
 * draw sequence and the episode->seed mapping -- the reference holds no second
 * implementation, golden vector or test for them (its checker skips them,
 * envs/hanabi_env.py:296,640-641).  Everything else is pinned by the invariants
 * that checker tests, restated in tests/test_oracle_hanabi.py.
 *
 * Episode order: as in cartpole_oracle.c (ascending world index).
 */
#include "mrl_oracle.h"

#include <stdlib.h>
#include <string.h>

#define HAND 5
#define NPLAYERS 2
#define MAX_CARDS 50

enum { MV_DISCARD = 0, MV_PLAY, MV_REVEAL_COLOR, MV_REVEAL_RANK, MV_INVALID };

typedef struct {
    uint8_t cards[HAND];
    uint8_t size;
    int8_t known_color[HAND];
    int8_t known_rank[HAND];
    uint64_t plausible[HAND];
} hand_t;

typedef struct {
    uint8_t move;
    int8_t player, target_player, card_index;
    uint8_t scored, information_token;
    int8_t color, rank;
    uint8_t reveal_bitmask, newly_revealed_bitmask;
    int8_t deal_to_player;
} lastmove_t;

typedef struct {
    uint8_t cards[MAX_CARDS];
    uint8_t size;
    uint8_t discard_counts[25];
    uint8_t fireworks[5];
    uint8_t information_tokens, life_tokens, cur_player;
    int8_t turns_to_play, score, new_rew;
    lastmove_t last;
    hand_t hands[NPLAYERS];
    uint32_t rng;
} game_t;

/* dump record: fixed little-endian byte layout shared with the GPU parity tests
 *   [0..49] deck cards  [50] deck size  [51..75] discard counts  [76..80] fireworks
 *   [81] info [82] life [83] cur_player [84] turns_to_play [85] score [86] new_rew
 *   [87..97] last move (move, player, target, card_index, scored, info_token,
 *            color, rank, reveal, newly_revealed, deal_to)
 *   [98..99] padding
 *   per hand h (base 100 + 36*h): cards[5], size, known_color[5], known_rank[5],
 *            plausible[5] as uint32 LE (20 bytes, 4-byte aligned)
 *   [172..175] rng state (uint32 LE)
 */
#define RECORD_BYTES 176

struct orc_hanabi {
    uint32_t n;
    uint32_t next_episode;
    uint32_t colors, ranks, players, max_info, max_life, hand_size;
    game_t *games;
    uint8_t *obs;    /* 2 x N x 658 */
    uint8_t *state;  /* 2 x N x 783 */
    int32_t *mask;   /* 2 x N x 20 */
    int32_t *active; /* 2 x N */
    float *reward;   /* 2 x N */
    int32_t *done;   /* N */
};

static inline uint8_t *obs_of(orc_hanabi *s, int agent, uint32_t wi)
{
    return &s->obs[((size_t)agent * s->n + wi) * ORC_HANABI_OBS];
}
static inline uint8_t *state_of(orc_hanabi *s, int agent, uint32_t wi)
{
    return &s->state[((size_t)agent * s->n + wi) * ORC_HANABI_STATE];
}
static inline int32_t *mask_of(orc_hanabi *s, int agent, uint32_t wi)
{
    return &s->mask[((size_t)agent * s->n + wi) * ORC_HANABI_MOVES];
}

static inline int copies_of_rank(const orc_hanabi *s, uint32_t r) { return r == 0 ? 3 : (r == s->ranks - 1 ? 1 : 2); }

/* sim.cpp:45-52 */
static uint8_t draw(game_t *g)
{
    float r = orc_rng_next(&g->rng);
    float scaled = (float)g->size * r; /* int * float -> one float multiply */
    int32_t at = (int32_t)scaled;
    uint8_t card = g->cards[at];
    g->cards[at] = g->cards[g->size - 1];
    g->size--;
    return card;
}

/* sim.cpp:367-379 */
static void encode_agent(orc_hanabi *s, uint32_t wi, int agent)
{
    game_t *g = &s->games[wi];
    /* The reference advances a running offset while it writes (sim.cpp:54-365), so
     * when information_tokens exceeds its maximum (a rank-5 card played at full
     * tokens: sim.cpp:676-678 adds the token unconditionally) the token
     * thermometer is longer and every later section moves up.  For the full
     * configuration the last byte then lands one past the 658-byte row -- an
     * out-of-bounds write in the reference; here the row is encoded in a scratch
     * buffer and only what fits the row is kept. */
    uint8_t scratch[ORC_HANABI_STATE + 64];
    uint8_t *o = scratch;
    const uint32_t colors = s->colors, ranks = s->ranks, np = s->players, hs = s->hand_size;
    const int bpc = (int)(colors * ranks);
    int at = 0;

    /* hands of the others, then "hand is short" flags (:54-90) */
    for (uint32_t i = 1; i < np; i++) {
        const hand_t *h = &g->hands[(agent + i) % np];
        for (uint32_t c = 0; c < h->size; c++)
            for (int b = 0; b < bpc; b++) o[at++] = (b == h->cards[c]);
        for (uint32_t c = h->size; c < hs; c++)
            for (int b = 0; b < bpc; b++) o[at++] = 0;
    }
    for (uint32_t i = 0; i < np; i++) o[at++] = (g->hands[(agent + i) % np].size < hs);

    /* board (:92-135) */
    for (int i = 0; i < g->size; i++) o[at++] = 1;
    int max_deck = (int)((4 + (ranks - 2) * 2) * colors - hs * np);
    for (int i = g->size; i < max_deck; i++) o[at++] = 0;
    for (uint32_t c = 0; c < colors; c++)
        for (uint32_t i = 0; i < ranks; i++) o[at++] = (i + 1 == g->fireworks[c]);
    for (int i = 0; i < g->information_tokens; i++) o[at++] = 1;
    for (uint32_t i = g->information_tokens; i < s->max_info; i++) o[at++] = 0;
    for (int i = 0; i < g->life_tokens; i++) o[at++] = 1;
    for (uint32_t i = g->life_tokens; i < s->max_life; i++) o[at++] = 0;

    /* discards (:137-156) */
    int id = 0;
    for (uint32_t c = 0; c < colors; c++)
        for (uint32_t r = 0; r < ranks; r++) {
            int copies = copies_of_rank(s, r);
            for (int i = 0; i < copies; i++) o[at++] = (g->discard_counts[id] > i);
            id++;
        }

    /* last action (:158-289) */
    const lastmove_t *lm = &g->last;
    int rel = lm->player == -1 ? -1 : (agent - lm->player + (int)np) % (int)np;
    for (int i = 0; i < (int)np; i++) o[at++] = (i == rel);
    for (int i = 0; i < 4; i++) o[at + i] = 0;
    if (lm->move == MV_PLAY) o[at] = 1;
    else if (lm->move == MV_DISCARD) o[at + 1] = 1;
    else if (lm->move == MV_REVEAL_COLOR) o[at + 2] = 1;
    else if (lm->move == MV_REVEAL_RANK) o[at + 3] = 1;
    at += 4;
    int is_hint = lm->move == MV_REVEAL_COLOR || lm->move == MV_REVEAL_RANK;
    int is_card = lm->move == MV_PLAY || lm->move == MV_DISCARD;
    if (is_hint) {
        int8_t rel_target = (int8_t)((agent - lm->target_player + (int)np) % (int)np);
        for (int i = 0; i < (int)np; i++) o[at + i] = (i == rel_target);
    } else {
        for (int i = 0; i < (int)np; i++) o[at + i] = 0;
    }
    at += np;
    for (uint32_t i = 0; i < colors; i++) o[at + i] = (lm->move == MV_REVEAL_COLOR) && (i == (uint8_t)lm->color);
    at += colors;
    for (uint32_t i = 0; i < ranks; i++) o[at + i] = (lm->move == MV_REVEAL_RANK) && (i == (uint8_t)lm->rank);
    at += ranks;
    for (uint32_t i = 0, bit = 1; i < hs; i++, bit <<= 1) o[at + i] = is_hint && ((lm->reveal_bitmask & bit) > 0);
    at += hs;
    for (uint32_t i = 0; i < hs; i++) o[at + i] = is_card && ((int)i == lm->card_index);
    at += hs;
    for (uint32_t i = 0; i < colors * ranks; i++)
        o[at + i] = is_card && (i == (uint32_t)(lm->color * (int)ranks + lm->rank));
    at += colors * ranks;
    o[at] = (lm->move == MV_PLAY) ? lm->scored : 0;
    o[at + 1] = (lm->move == MV_PLAY) ? lm->information_token : 0;
    at += 2;

    /* card knowledge (:291-331); note "1 << i" with i the player loop index */
    for (uint32_t i = 0; i < np; i++) {
        const hand_t *h = &g->hands[(agent + i) % np];
        for (int c = 0; c < h->size; c++) {
            for (int v = 0; v < bpc; v++) o[at++] = ((h->plausible[c] & (uint64_t)(1 << i)) != 0);
            for (uint32_t v = 0; v < colors; v++) o[at++] = (h->known_color[c] == (int)v);
            for (uint32_t v = 0; v < ranks; v++) o[at++] = (h->known_rank[c] == (int)v);
        }
        for (uint32_t c = h->size; c < hs; c++)
            for (uint32_t v = 0; v < (uint32_t)bpc + colors + ranks; v++) o[at++] = 0;
    }

    /* state = obs prefix + own hand (:333-365) */
    const int obs_len = at;
    const hand_t *own = &g->hands[agent];
    for (int c = 0; c < own->size; c++)
        for (int b = 0; b < bpc; b++) o[at++] = (b == own->cards[c]);
    for (uint32_t c = own->size; c < hs; c++)
        for (int b = 0; b < bpc; b++) o[at++] = 0;
    memcpy(obs_of(s, agent, wi), scratch, (size_t)(obs_len < ORC_HANABI_OBS ? obs_len : ORC_HANABI_OBS));
    memcpy(state_of(s, agent, wi), scratch, (size_t)(at < ORC_HANABI_STATE ? at : ORC_HANABI_STATE));
}

/* sim.cpp:381-444 */
static void legal_moves(orc_hanabi *s, uint32_t wi, int agent)
{
    const game_t *g = &s->games[wi];
    int32_t *m = mask_of(s, agent, wi);
    const hand_t *own = &g->hands[agent];
    const uint32_t hs = s->hand_size, np = s->players;
    int at = 0;
    for (uint32_t i = 0; i < hs; i++) m[at++] = (i < own->size && g->information_tokens < s->max_info);
    for (uint32_t i = 0; i < hs; i++) m[at++] = (i < own->size);
    for (uint32_t p = 1; p < np; p++) {
        const hand_t *h = &g->hands[(agent + p) % np];
        for (uint32_t c = 0; c < s->colors; c++) {
            int has = 0;
            for (uint32_t k = 0; k < hs; k++) has |= (h->cards[k] / s->ranks == c);
            m[at++] = (g->information_tokens > 0 && has);
        }
    }
    for (uint32_t p = 1; p < np; p++) {
        const hand_t *h = &g->hands[(agent + p) % np];
        for (uint32_t r = 0; r < s->ranks; r++) {
            int has = 0;
            for (uint32_t k = 0; k < hs; k++) has |= (h->cards[k] % s->ranks == r);
            m[at++] = (g->information_tokens > 0 && has);
        }
    }
    for (; at < ORC_HANABI_MOVES; at++) m[at] = 0;
}

static void clear_last(lastmove_t *lm)
{
    lm->target_player = -1;
    lm->card_index = -1;
    lm->scored = 0;
    lm->information_token = 0;
    lm->color = -1;
    lm->rank = -1;
    lm->reveal_bitmask = 0;
    lm->newly_revealed_bitmask = 0;
    lm->deal_to_player = -1;
}

/* sim.cpp:446-532 */
static void reset_world(orc_hanabi *s, uint32_t wi, uint32_t episode)
{
    game_t *g = &s->games[wi];
    g->rng = orc_rng_seed(episode);
    int k = 0;
    for (uint32_t c = 0; c < s->colors; c++)
        for (uint32_t r = 0; r < s->ranks; r++) {
            int id = (int)(s->ranks * c + r);
            int copies = copies_of_rank(s, r);
            for (int i = 0; i < copies; i++) g->cards[k++] = (uint8_t)id;
            g->discard_counts[id] = 0;
        }
    g->size = (uint8_t)k;
    for (uint32_t c = 0; c < s->colors; c++) g->fireworks[c] = 0;
    g->information_tokens = (uint8_t)s->max_info;
    g->life_tokens = (uint8_t)s->max_life;
    g->cur_player = 0;
    g->turns_to_play = (int8_t)s->players;
    g->score = 0;
    g->new_rew = 0;
    g->last.move = MV_INVALID;
    g->last.player = -1;
    clear_last(&g->last);

    uint64_t all = ((uint64_t)1 << (s->colors * s->ranks)) - 1;
    for (uint32_t i = 0; i < s->players; i++) {
        s->active[(size_t)i * s->n + wi] = (i == 0);
        hand_t *h = &g->hands[i];
        for (uint32_t j = 0; j < s->hand_size; j++) {
            h->cards[j] = draw(g);
            h->plausible[j] = all;
            h->known_color[j] = -1;
            h->known_rank[j] = -1;
        }
        h->size = (uint8_t)s->hand_size;
    }
    for (uint32_t i = 0; i < s->players; i++) {
        encode_agent(s, wi, (int)i);
        legal_moves(s, wi, (int)i);
    }
}

/* sim.cpp:567-594 */
static void take_from_hand(orc_hanabi *s, game_t *g, hand_t *h, int8_t index)
{
    if (g->size == 0) {
        for (int8_t i = (int8_t)(index + 1); i < h->size; i++) {
            h->cards[i - 1] = h->cards[i];
            h->plausible[i - 1] = h->plausible[i];
            h->known_color[i - 1] = h->known_color[i];
            h->known_rank[i - 1] = h->known_rank[i];
        }
        h->size--;
    } else {
        h->cards[index] = draw(g);
        h->plausible[index] = ((uint64_t)1 << (s->colors * s->ranks)) - 1;
        h->known_color[index] = -1;
        h->known_rank[index] = -1;
    }
}

/* sim.cpp:596-792 */
static void apply_action(orc_hanabi *s, uint32_t wi, const int32_t *actions)
{
    game_t *g = &s->games[wi];
    if (g->size == 0) g->turns_to_play--;
    const int actor = g->cur_player;
    hand_t *h = &g->hands[actor];
    lastmove_t *lm = &g->last;
    const uint32_t colors = s->colors, ranks = s->ranks, np = s->players, hs = s->hand_size;
    uint32_t uid = (uint32_t)actions[(size_t)actor * s->n + wi];

    lm->player = (int8_t)g->cur_player;
    clear_last(lm);
    g->cur_player = (uint8_t)((g->cur_player + 1) % np);

    if (uid < hs) { /* discard */
        lm->move = MV_DISCARD;
        lm->card_index = (int8_t)uid;
        uint8_t card = h->cards[uid];
        lm->color = (int8_t)(card / ranks);
        lm->rank = (int8_t)(card % ranks);
        g->discard_counts[card]++;
        g->information_tokens++;
        take_from_hand(s, g, h, (int8_t)uid);
        return;
    }
    uid -= hs;
    if (uid < hs) { /* play */
        lm->move = MV_PLAY;
        lm->card_index = (int8_t)uid;
        uint8_t card = h->cards[uid];
        lm->color = (int8_t)(card / ranks);
        lm->rank = (int8_t)(card % ranks);
        if (g->fireworks[lm->color] == lm->rank) {
            g->fireworks[lm->color]++;
            if (g->fireworks[lm->color] == ranks) {
                g->information_tokens++;
                lm->information_token = 1;
            } else {
                lm->information_token = 0;
            }
            lm->scored = 1;
        } else {
            g->discard_counts[card]++;
            g->life_tokens--;
            lm->scored = 0;
        }
        take_from_hand(s, g, h, (int8_t)uid);
        return;
    }
    uid -= hs;
    if (uid < (np - 1) * colors) { /* reveal colour */
        lm->move = MV_REVEAL_COLOR;
        int target_offset = 1 + (int)(uid / colors);
        g->information_tokens--;
        int partner = (actor + target_offset) % (int)np;
        hand_t *ph = &g->hands[partner];
        lm->target_player = (int8_t)partner;
        lm->color = (int8_t)(uid % colors);
        lm->reveal_bitmask = 0;
        for (int i = 0; i < ph->size; i++)
            if (ph->cards[i] / ranks == (uint32_t)lm->color) lm->reveal_bitmask |= (uint8_t)(1 << i);
        lm->newly_revealed_bitmask = 0;
        uint64_t hint = 0;
        for (uint32_t i = 0; i < ranks; i++) hint |= (uint64_t)(int64_t)((int)1 << (lm->color * (int)ranks + (int)i));
        for (int i = 0; i < ph->size; i++) {
            if (ph->cards[i] / ranks == (uint32_t)lm->color) {
                if (ph->known_color[i] == -1) lm->newly_revealed_bitmask |= (uint8_t)(1 << i);
                ph->known_color[i] = lm->color;
                ph->plausible[i] &= hint;
            } else {
                ph->plausible[i] &= ~hint;
            }
        }
        return;
    }
    uid -= (np - 1) * colors;
    { /* reveal rank */
        lm->move = MV_REVEAL_RANK;
        int target_offset = 1 + (int)(uid / ranks);
        g->information_tokens--;
        int partner = (actor + target_offset) % (int)np;
        hand_t *ph = &g->hands[partner];
        lm->target_player = (int8_t)partner;
        lm->rank = (int8_t)(uid % ranks);
        lm->reveal_bitmask = 0;
        for (int i = 0; i < ph->size; i++)
            if (ph->cards[i] % ranks == (uint32_t)lm->rank) lm->reveal_bitmask |= (uint8_t)(1 << i);
        lm->newly_revealed_bitmask = 0;
        uint64_t hint = 0;
        for (uint32_t i = 0; i < ranks; i++) hint |= (uint64_t)(int64_t)((int)1 << ((int)i * (int)ranks + lm->rank));
        for (int i = 0; i < ph->size; i++) {
            if (ph->cards[i] % ranks == (uint32_t)lm->rank) {
                if (ph->known_color[i] == -1) lm->newly_revealed_bitmask |= (uint8_t)(1 << i);
                ph->known_rank[i] = lm->rank;
                ph->plausible[i] &= hint;
            } else {
                ph->plausible[i] &= ~hint;
            }
        }
    }
}

/* returns 1 when the world finished its episode; the reset itself is applied
 * by the caller in ascending world order (sim.cpp:812-850) */
static int score_and_check(orc_hanabi *s, uint32_t wi)
{
    game_t *g = &s->games[wi];
    int8_t old = g->score;
    g->score = 0;
    if (g->life_tokens > 0)
        for (uint32_t c = 0; c < s->colors; c++) g->score = (int8_t)(g->score + g->fireworks[c]);
    g->new_rew = (int8_t)(g->score - old);
    for (uint32_t i = 0; i < s->players; i++) s->reward[(size_t)i * s->n + wi] = (float)g->new_rew;
    int over = 0;
    if (g->life_tokens < 1) over = 1;
    if ((uint32_t)g->score >= s->colors * s->ranks) over = 1;
    if (g->turns_to_play <= 0) over = 1;
    s->done[wi] = over;
    return over;
}

orc_hanabi *orc_hanabi_create(const orc_hanabi_config *cfg, uint32_t num_worlds)
{
    if (!cfg || cfg->players != NPLAYERS || cfg->colors < 1 || cfg->colors > 5 || cfg->ranks < 2 || cfg->ranks > 5 ||
        cfg->max_information_tokens > 8 || cfg->max_life_tokens > 3)
        return NULL;
    orc_hanabi *s = (orc_hanabi *)calloc(1, sizeof(*s));
    s->n = num_worlds;
    s->colors = cfg->colors;
    s->ranks = cfg->ranks;
    s->players = cfg->players;
    s->max_info = cfg->max_information_tokens;
    s->max_life = cfg->max_life_tokens;
    s->hand_size = cfg->players < 4 ? 5 : 4; /* sim.cpp:875 */
    s->games = (game_t *)calloc(num_worlds, sizeof(game_t));
    s->obs = (uint8_t *)calloc((size_t)NPLAYERS * num_worlds * ORC_HANABI_OBS, 1);
    s->state = (uint8_t *)calloc((size_t)NPLAYERS * num_worlds * ORC_HANABI_STATE, 1);
    s->mask = (int32_t *)calloc((size_t)NPLAYERS * num_worlds * ORC_HANABI_MOVES, sizeof(int32_t));
    s->active = (int32_t *)calloc((size_t)NPLAYERS * num_worlds, sizeof(int32_t));
    s->reward = (float *)calloc((size_t)NPLAYERS * num_worlds, sizeof(float));
    s->done = (int32_t *)calloc(num_worlds, sizeof(int32_t));
    for (uint32_t wi = 0; wi < num_worlds; wi++) reset_world(s, wi, s->next_episode++);
    return s;
}

void orc_hanabi_destroy(orc_hanabi *s)
{
    if (!s) return;
    free(s->games);
    free(s->obs);
    free(s->state);
    free(s->mask);
    free(s->active);
    free(s->reward);
    free(s->done);
    free(s);
}

void orc_hanabi_step(orc_hanabi *s, const int32_t *actions, int num_threads)
{
    const long n = (long)s->n;
    if (num_threads < 1) num_threads = 1;
#pragma omp parallel for schedule(static) num_threads(num_threads)
    for (long wi = 0; wi < n; wi++) {
        apply_action(s, (uint32_t)wi, actions);
        /* sim.cpp:794-810: only the player to move is refreshed */
        game_t *g = &s->games[wi];
        for (uint32_t i = 0; i < s->players; i++) {
            if (i == g->cur_player) {
                s->active[(size_t)i * s->n + wi] = 1;
                encode_agent(s, (uint32_t)wi, (int)i);
                legal_moves(s, (uint32_t)wi, (int)i);
            } else {
                s->active[(size_t)i * s->n + wi] = 0;
            }
        }
        score_and_check(s, (uint32_t)wi);
    }
    for (uint32_t wi = 0; wi < s->n; wi++)
        if (s->done[wi]) reset_world(s, wi, s->next_episode++);
}

const uint8_t *orc_hanabi_obs(const orc_hanabi *s) { return s->obs; }
const uint8_t *orc_hanabi_state(const orc_hanabi *s) { return s->state; }
const int32_t *orc_hanabi_mask(const orc_hanabi *s) { return s->mask; }
const int32_t *orc_hanabi_active(const orc_hanabi *s) { return s->active; }
const float *orc_hanabi_reward(const orc_hanabi *s) { return s->reward; }
const int32_t *orc_hanabi_done(const orc_hanabi *s) { return s->done; }
uint32_t orc_hanabi_episodes(const orc_hanabi *s) { return s->next_episode; }
uint32_t orc_hanabi_record_bytes(void) { return RECORD_BYTES; }

void orc_hanabi_dump(const orc_hanabi *s, uint8_t *records)
{
    for (uint32_t wi = 0; wi < s->n; wi++) {
        const game_t *g = &s->games[wi];
        uint8_t *r = &records[(size_t)wi * RECORD_BYTES];
        memset(r, 0, RECORD_BYTES);
        memcpy(r, g->cards, MAX_CARDS);
        r[50] = g->size;
        memcpy(r + 51, g->discard_counts, 25);
        memcpy(r + 76, g->fireworks, 5);
        r[81] = g->information_tokens;
        r[82] = g->life_tokens;
        r[83] = g->cur_player;
        r[84] = (uint8_t)g->turns_to_play;
        r[85] = (uint8_t)g->score;
        r[86] = (uint8_t)g->new_rew;
        const lastmove_t *lm = &g->last;
        r[87] = lm->move;
        r[88] = (uint8_t)lm->player;
        r[89] = (uint8_t)lm->target_player;
        r[90] = (uint8_t)lm->card_index;
        r[91] = lm->scored;
        r[92] = lm->information_token;
        r[93] = (uint8_t)lm->color;
        r[94] = (uint8_t)lm->rank;
        r[95] = lm->reveal_bitmask;
        r[96] = lm->newly_revealed_bitmask;
        r[97] = (uint8_t)lm->deal_to_player;
        for (int hnd = 0; hnd < NPLAYERS; hnd++) {
            const hand_t *h = &g->hands[hnd];
            uint8_t *b = r + 100 + 36 * hnd;
            memcpy(b, h->cards, HAND);
            b[5] = h->size;
            memcpy(b + 6, h->known_color, HAND);
            memcpy(b + 11, h->known_rank, HAND);
            for (int c = 0; c < HAND; c++) {
                uint32_t m = (uint32_t)h->plausible[c];
                memcpy(b + 16 + 4 * c, &m, 4);
            }
        }
        memcpy(r + 172, &g->rng, 4);
    }
}
