/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle, balance-beam world step.
 *
 * Plain-C restatement of /root/reference/src/balance_beam_env/sim.cpp:
 *   :45-74 resetWorld (episode-seeded generator rng.hpp:5-40, two positions, empty history),
 *   :76-92 actionSystem, :94-97 timeSystem, :99-112 observationSystem, :114-151 checkDone,
 *   graph :155-171, constructor :173-205.
 * Components as in sim.hpp:44-72 (Observation{int32 x[6]; int32 time}, Location{int32 x}).
 * The history shift `for (i = 2*TIME; i > 0; i--) x[i] = x[i-1]` (sim.cpp:104-106) runs one
 * element past x[]: x[6] is the `time` field behind it, assigned right after; kept as written
 * (the row is a 7-int array here, so the same statements do the same thing).
 * Episode indices are handed out in ascending world order within a step.
 *
 * Pinned: transitions against the reference's own one-step checker world
 * (envs/balance_beam_env.py:96-146 PantheonLine, used by validate_step :152-217) through
 * tests/golden/balance_transitions.npz; reset positions only by an independent restatement of
 * rng.hpp in the tests ("parity unpinned" for the episode -> seed order, as for Cartpole).
 */
#include "mrl_oracle.h"

#include <stdlib.h>

#define B_TIME 3
#define B_SPACES 5
#define B_BUFFER 2
#define B_SCALE 0.2
#define B_ROW (2 * B_TIME + 1)

struct orc_balance {
    uint32_t n;
    uint32_t next_episode;
    int32_t *loc;   /* (2, N) Location.x */
    int32_t *time;  /* (N) WorldTime */
    int32_t *obs;   /* (2, N, 7) */
    float *reward;  /* (2, N) */
    int32_t *done;  /* (N) */
};

static void reset_world(orc_balance *s, uint32_t w)
{
    uint32_t g = orc_rng_seed(s->next_episode++);
    s->time[w] = B_TIME - 1;
    for (int i = 0; i < 2; i++) {
        s->loc[(size_t)i * s->n + w] = (int32_t)(B_SPACES * orc_rng_next(&g));
        int32_t *row = &s->obs[((size_t)i * s->n + w) * B_ROW];
        for (int t = 0; t < 2 * B_TIME; t++) row[t] = 0;
        row[2 * B_TIME] = s->time[w];
    }
    for (int i = 0; i < 2; i++) {
        int32_t *row = &s->obs[((size_t)i * s->n + w) * B_ROW];
        row[0] = s->loc[(size_t)i * s->n + w] + B_BUFFER;
        row[B_TIME] = s->loc[(size_t)(1 - i) * s->n + w] + B_BUFFER;
    }
}

orc_balance *orc_balance_create(uint32_t num_worlds)
{
    orc_balance *s = (orc_balance *)calloc(1, sizeof(*s));
    s->n = num_worlds;
    s->loc = (int32_t *)calloc((size_t)2 * num_worlds, sizeof(int32_t));
    s->time = (int32_t *)calloc(num_worlds, sizeof(int32_t));
    s->obs = (int32_t *)calloc((size_t)2 * num_worlds * B_ROW, sizeof(int32_t));
    s->reward = (float *)calloc((size_t)2 * num_worlds, sizeof(float));
    s->done = (int32_t *)calloc(num_worlds, sizeof(int32_t));
    for (uint32_t w = 0; w < num_worlds; w++) reset_world(s, w);
    return s;
}

void orc_balance_destroy(orc_balance *s)
{
    if (!s) return;
    free(s->loc);
    free(s->time);
    free(s->obs);
    free(s->reward);
    free(s->done);
    free(s);
}

void orc_balance_step(orc_balance *s, const int32_t *actions)
{
    /* serial over worlds: the episode counter is handed out in world order */
    for (uint32_t w = 0; w < s->n; w++) {
        for (int i = 0; i < 2; i++) {
            int32_t *x = &s->loc[(size_t)i * s->n + w];
            switch (actions[(size_t)i * s->n + w]) {
            case 0: *x += -2; break;
            case 1: *x += -1; break;
            case 2: *x += 1; break;
            case 3: *x += 2; break;
            default: break;
            }
        }
        s->time[w] -= 1;
        for (int i = 0; i < 2; i++) {
            int32_t *row = &s->obs[((size_t)i * s->n + w) * B_ROW];
            for (int k = B_TIME * 2; k > 0; k--) row[k] = row[k - 1];
            row[B_TIME] = s->loc[(size_t)(1 - i) * s->n + w] + B_BUFFER;
            row[0] = s->loc[(size_t)i * s->n + w] + B_BUFFER;
            row[2 * B_TIME] = s->time[w];
        }
        int reset_now = 0;
        const int32_t l0 = s->loc[w], l1 = s->loc[(size_t)s->n + w];
        float reward = (float)(l0 == l1 ? 1.0 : -abs(l0 - l1) * B_SCALE);
        for (int i = 0; i < 2; i++) {
            const int32_t x = i ? l1 : l0;
            if (x < 0 || x >= B_SPACES) {
                reset_now = 1;
                reward = (float)(-B_SPACES * (s->time[w] + 1) * B_SCALE);
            }
        }
        s->reward[w] = reward;
        s->reward[(size_t)s->n + w] = reward;
        if (s->time[w] == 0) reset_now = 1;
        s->done[w] = reset_now;
        if (reset_now) reset_world(s, w);
    }
}

int32_t *orc_balance_obs(orc_balance *s) { return s->obs; }
int32_t *orc_balance_loc(orc_balance *s) { return s->loc; }
int32_t *orc_balance_time(orc_balance *s) { return s->time; }
const float *orc_balance_reward(const orc_balance *s) { return s->reward; }
const int32_t *orc_balance_done(const orc_balance *s) { return s->done; }
uint32_t orc_balance_episodes(const orc_balance *s) { return s->next_episode; }
