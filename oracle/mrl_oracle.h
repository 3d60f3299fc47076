/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the batched env-step hot path.
 *
 * This header declares a plain-C restatement of the reference's per-world
 * step for the three games on the hot path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the library built from this directory; the product (the HIP kernels behind
 * include/mrl_envs.h) never links, imports or calls it.
 *
 * Reference files restated (read as text, nothing copied):
 *   Overcooked: /root/reference/src/overcooked_env/sim.hpp:39-198, sim.cpp:53-659
 *   Simplecooked: /root/reference/src/overcooked2_env/sim.hpp:12-189, sim.cpp:46-577
 *   Cartpole  : /root/reference/src/cartpole_env/sim.cpp:9-141, rng.hpp:5-40
 *   Hanabi    : /root/reference/src/hanabi_env/sim.hpp:13-140, sim.cpp:45-897, rng.hpp:5-40
 *
 * Pinning (see DESIGN.md "Oracle"):
 *   Overcooked: pinned against the reference's own numpy implementation
 *               (envs/overcooked_reimplement.py) through tests/golden/overcooked_*.npz.
 *   Cartpole  : dynamics pinned against the float64 one-step check the reference
 *               uses (envs/cartpole_env.py:177-233, tolerance 1e-6) restated in
 *               tests; reset/RNG stream pinned by hand-computed known answers.
 *   Hanabi    : the reference holds no second implementation or golden vector
 *               in-tree -> card-knowledge / last-action / RNG sections are
 *               "parity unpinned"; the rest is pinned by the invariants the
 *               reference checker tests (envs/hanabi_env.py:478-657), restated in tests.
 */
#ifndef MRL_ORACLE_H
#define MRL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Overcooked                                                          */
/* ------------------------------------------------------------------ */

#define ORC_MAX_CELLS 255      /* WorldState.size is uint8 (sim.hpp:86)        */
#define ORC_MAX_PLAYERS 64     /* MAX_NUM_PLAYERS (sim.hpp:14)                 */
#define ORC_NUM_RECIPES 16     /* (MAX_NUM_INGREDIENTS+1)^2 (sim.hpp:17)       */

typedef struct orc_overcooked_config {
    int64_t height, width, num_players;
    int64_t placement_in_pot_rew, dish_pickup_rew, soup_pickup_rew;
    int64_t horizon;
    int64_t terrain[ORC_MAX_CELLS];
    int64_t start_player_x[ORC_MAX_PLAYERS];
    int64_t start_player_y[ORC_MAX_PLAYERS];
    int64_t recipe_values[ORC_NUM_RECIPES];
    int64_t recipe_times[ORC_NUM_RECIPES];
} orc_overcooked_config;

typedef struct orc_overcooked orc_overcooked;

orc_overcooked *orc_overcooked_create(const orc_overcooked_config *cfg, uint32_t num_worlds);
void orc_overcooked_destroy(orc_overcooked *s);
/* actions: (P, N) int32, same order as the reference's action tensor.          */
void orc_overcooked_step(orc_overcooked *s, const int32_t *actions, int num_threads);
/* Outputs, all owned by the simulator and valid until destroy:
 *   obs    : (N, P, C, F) uint8   F = 5P + 16, row (p*C + c) = LocationXPlayer id
 *   reward : (P, N) int32
 *   done   : (N) int32
 */
const uint8_t *orc_overcooked_obs(const orc_overcooked *s);
const int32_t *orc_overcooked_reward(const orc_overcooked *s);
const int32_t *orc_overcooked_done(const orc_overcooked *s);
/* Compact dump of the true per-world state, for state-level parity tests:
 *   players: (N, P, 6) uint8 = position, orientation, held{name,onions,tomatoes,tick}
 *   objects: (N, C, 4) uint8 = name, onions, tomatoes, tick
 *   timestep: (N) int32
 */
void orc_overcooked_dump(const orc_overcooked *s, uint8_t *players, uint8_t *objects, int32_t *timestep);

/* ------------------------------------------------------------------ */
/* Simplecooked (overcooked2_env): same config struct, other rules     */
/*   /root/reference/src/overcooked2_env/sim.hpp:12-14,40; sim.cpp     */
/*   pinned against envs/overcooked2_reimplement.py through            */
/*   tests/golden/simplecooked_*.npz                                   */
/* ------------------------------------------------------------------ */

#define ORC_SIMPLE_MAX_CELLS 100   /* MAX_SIZE (overcooked2_env/sim.hpp:12)        */
#define ORC_SIMPLE_MAX_PLAYERS 2   /* MAX_NUM_PLAYERS (overcooked2_env/sim.hpp:13) */

typedef struct orc_simplecooked orc_simplecooked;

/* terrain values follow overcooked2's enum: AIR, POT, COUNTER, ONION_SOURCE, DISH_SOURCE, SERVING, TOMATO_SOURCE */
orc_simplecooked *orc_simplecooked_create(const orc_overcooked_config *cfg, uint32_t num_worlds);
void orc_simplecooked_destroy(orc_simplecooked *s);
void orc_simplecooked_step(orc_simplecooked *s, const int32_t *actions, int num_threads);
/* obs (N, P, C, F) uint8 with F = 5P + 10; reward (P, N) int32; done (N) int32 */
const uint8_t *orc_simplecooked_obs(const orc_simplecooked *s);
const int32_t *orc_simplecooked_reward(const orc_simplecooked *s);
const int32_t *orc_simplecooked_done(const orc_simplecooked *s);
/* players (N, P, 6), objects (N, C, 4) as orc_overcooked_dump; timestep (N); dishes_out (N) = WorldState.num_dishes_out */
void orc_simplecooked_dump(const orc_simplecooked *s, uint8_t *players, uint8_t *objects, int32_t *timestep, int32_t *dishes_out);

/* ------------------------------------------------------------------ */
/* Cartpole                                                            */
/* ------------------------------------------------------------------ */

typedef struct orc_cartpole orc_cartpole;

orc_cartpole *orc_cartpole_create(uint32_t num_worlds);
void orc_cartpole_destroy(orc_cartpole *s);
/* actions: (N, 1) int32.  Episode indices are handed out in ascending world
 * order within a step (what a one-thread executor does). */
void orc_cartpole_step(orc_cartpole *s, const int32_t *actions, int num_threads);
float *orc_cartpole_state(orc_cartpole *s);          /* (N, 4) float32, writable */
const float *orc_cartpole_reward(const orc_cartpole *s);   /* (N, 1) */
const int32_t *orc_cartpole_done(const orc_cartpole *s);   /* (N, 1) */
uint32_t orc_cartpole_episodes(const orc_cartpole *s);

/* ------------------------------------------------------------------ */
/* Balance beam (src/balance_beam_env)                                 */
/* ------------------------------------------------------------------ */

typedef struct orc_balance orc_balance;

orc_balance *orc_balance_create(uint32_t num_worlds);
void orc_balance_destroy(orc_balance *s);
/* actions: (2, N) int32, 0..3 = moves -2, -1, +1, +2.  Worlds are stepped in order (episode indices). */
void orc_balance_step(orc_balance *s, const int32_t *actions);
int32_t *orc_balance_obs(orc_balance *s);   /* (2, N, 7) int32, writable (tests plant states)   */
int32_t *orc_balance_loc(orc_balance *s);   /* (2, N) Location.x, writable                      */
int32_t *orc_balance_time(orc_balance *s);  /* (N) WorldTime, writable                          */
const float *orc_balance_reward(const orc_balance *s);  /* (2, N) */
const int32_t *orc_balance_done(const orc_balance *s);  /* (N)    */
uint32_t orc_balance_episodes(const orc_balance *s);

/* the episode-seeded generator Cartpole, Hanabi and the balance beam use (rng.hpp:5-40) */
uint32_t orc_rng_seed(uint32_t episode_idx);
float orc_rng_next(uint32_t *state);

/* ------------------------------------------------------------------ */
/* Hanabi                                                              */
/* ------------------------------------------------------------------ */

#define ORC_HANABI_OBS 658
#define ORC_HANABI_STATE 783
#define ORC_HANABI_MOVES 20

typedef struct orc_hanabi_config {
    uint32_t colors, ranks, players, max_information_tokens, max_life_tokens;
} orc_hanabi_config;

typedef struct orc_hanabi orc_hanabi;

orc_hanabi *orc_hanabi_create(const orc_hanabi_config *cfg, uint32_t num_worlds);
void orc_hanabi_destroy(orc_hanabi *s);
/* actions: (2, N) int32 */
void orc_hanabi_step(orc_hanabi *s, const int32_t *actions, int num_threads);
const uint8_t *orc_hanabi_obs(const orc_hanabi *s);      /* (2, N, 658) */
const uint8_t *orc_hanabi_state(const orc_hanabi *s);    /* (2, N, 783) */
const int32_t *orc_hanabi_mask(const orc_hanabi *s);     /* (2, N, 20)  */
const int32_t *orc_hanabi_active(const orc_hanabi *s);   /* (2, N)      */
const float *orc_hanabi_reward(const orc_hanabi *s);     /* (2, N)      */
const int32_t *orc_hanabi_done(const orc_hanabi *s);     /* (N)         */
uint32_t orc_hanabi_episodes(const orc_hanabi *s);
/* raw per-world game record, layout documented in hanabi_oracle.c */
uint32_t orc_hanabi_record_bytes(void);
void orc_hanabi_dump(const orc_hanabi *s, uint8_t *records);

#ifdef __cplusplus
}
#endif
#endif
