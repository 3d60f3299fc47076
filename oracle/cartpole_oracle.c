/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle, Cartpole world step + the
 * episode-seeded generator shared with Hanabi.
 *
 * Restates /root/reference/src/cartpole_env/sim.cpp:9-21 (constants, double
 * literals), :48-66 (reset), :68-84 (Euler step, float state with double
 * intermediates), :86-96 (termination) and rng.hpp:5-40.  Built with
 * -ffp-contract=off (see Makefile) so every product and sum rounds once, as
 * in the reference's x86-64 CPU executor.
 *
 * Episode indices: the reference draws them from one process-wide atomic
 * (sim.cpp:51-53), so its multi-threaded runs are order-dependent.  The
 * oracle fixes the order a one-thread executor produces: the constructor gives
 * world i episode i, and within a step resetting worlds take the next indices
 * in ascending world order.
 *
 * Pinning: dynamics against the reference's float64 one-step check
 * (envs/cartpole_env.py:177-233,246-288; tolerance 1e-6) restated in
 * tests/test_oracle_cartpole.py; generator against hand-computed known answers.
 */
#include "mrl_oracle.h"

#include <math.h>
#include <stdlib.h>

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

/* rng.hpp:7-26 */
uint32_t orc_rng_seed(uint32_t episode_idx)
{
    uint32_t v0 = episode_idx, v1 = 0, sum = 0;
    for (int round = 0; round < 8; round++) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

/* rng.hpp:28-36 */
float orc_rng_next(uint32_t *state)
{
    *state = 1664525u * (*state) + 1013904223u;
    uint32_t low = *state & 0x00FFFFFFu;
    return (float)low / (float)0x01000000;
}

struct orc_cartpole {
    uint32_t n;
    uint32_t next_episode;
    float *state;  /* N x 4 */
    float *reward; /* N */
    int32_t *done; /* N */
    uint32_t *rng; /* N */
};

/* sim.cpp:48-66 */
static void reset_world(orc_cartpole *s, uint32_t wi, uint32_t episode)
{
    uint32_t g = orc_rng_seed(episode);
    const float lo = -0.05f, hi = 0.05f;
    const float span = hi - lo;
    float *st = &s->state[(size_t)wi * 4];
    for (int k = 0; k < 4; k++) {
        float r = orc_rng_next(&g);
        float scaled = r * span;
        st[k] = lo + scaled;
    }
    s->rng[wi] = g;
}

orc_cartpole *orc_cartpole_create(uint32_t num_worlds)
{
    orc_cartpole *s = (orc_cartpole *)calloc(1, sizeof(*s));
    s->n = num_worlds;
    s->state = (float *)calloc((size_t)num_worlds * 4, sizeof(float));
    s->reward = (float *)calloc(num_worlds, sizeof(float));
    s->done = (int32_t *)calloc(num_worlds, sizeof(int32_t));
    s->rng = (uint32_t *)calloc(num_worlds, sizeof(uint32_t));
    for (uint32_t wi = 0; wi < num_worlds; wi++) reset_world(s, wi, s->next_episode++);
    return s;
}

void orc_cartpole_destroy(orc_cartpole *s)
{
    if (!s) return;
    free(s->state);
    free(s->reward);
    free(s->done);
    free(s->rng);
    free(s);
}

void orc_cartpole_step(orc_cartpole *s, const int32_t *actions, int num_threads)
{
    const long n = (long)s->n;
    if (num_threads < 1) num_threads = 1;
#pragma omp parallel for schedule(static) num_threads(num_threads)
    for (long wi = 0; wi < n; wi++) {
        float *st = &s->state[(size_t)wi * 4];
        float x = st[0], x_dot = st[1], theta = st[2], theta_dot = st[3];
        /* sim.cpp:70-83 */
        float force = (actions[wi] == 1 ? FORCE_MAG : -FORCE_MAG);
        float costheta = cosf(theta);
        float sintheta = sinf(theta);
        float temp = (force + POLEMASS_LENGTH * theta_dot * theta_dot * sintheta) / TOTAL_MASS;
        float thetaacc =
            (GRAVITY * sintheta - costheta * temp) / (LENGTH * (4.0 / 3.0 - MASSPOLE * costheta * costheta / TOTAL_MASS));
        float xacc = temp - POLEMASS_LENGTH * thetaacc * costheta / TOTAL_MASS;
        x = x + TAU * x_dot;
        x_dot = x_dot + TAU * xacc;
        theta = theta + TAU * theta_dot;
        theta_dot = theta_dot + TAU * thetaacc;
        st[0] = x;
        st[1] = x_dot;
        st[2] = theta;
        st[3] = theta_dot;
        s->reward[wi] = 1.f;
        /* sim.cpp:88-91 */
        s->done[wi] = x < -X_THRESHOLD || x > X_THRESHOLD || theta < -THETA_THRESHOLD || theta > THETA_THRESHOLD;
    }
    /* resets in ascending world order (see header comment) */
    for (uint32_t wi = 0; wi < s->n; wi++)
        if (s->done[wi]) reset_world(s, wi, s->next_episode++);
}

float *orc_cartpole_state(orc_cartpole *s) { return s->state; }
const float *orc_cartpole_reward(const orc_cartpole *s) { return s->reward; }
const int32_t *orc_cartpole_done(const orc_cartpole *s) { return s->done; }
uint32_t orc_cartpole_episodes(const orc_cartpole *s) { return s->next_episode; }
