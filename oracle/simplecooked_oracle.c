/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle, "Simplecooked" (overcooked2_env) world step.
 *
 * Plain-C restatement of the reference task graph
 *   /root/reference/src/overcooked2_env/sim.cpp:422-451 (graph),
 *   :174-287 (get_pot_states, is_dish_pickup_useful, the sequential resolve_interacts),
 *   :289-348 (movement / collisions), :350-360 (pots), :362-420 (reset + bookkeeping),
 *   :62-148 (observation rows), :470-575 (init)
 * with the component widths of sim.hpp:55-184 (uint8 positions and reward parameters, int8
 * tick, uint8 num_dishes_out) kept as they are, and the reference's incremental observation
 * update (player channels are cleared through past_player only; channel 5P+5 is zeroed on
 * every pass, which also wipes the TOMATO_SOURCE terrain bit that shares it, sim.cpp:74 / :557).
 *
 * Differences from overcooked_env that matter (sim.hpp:40, sim.cpp:199-287):
 *   terrain enum AIR, POT, COUNTER, ONION_SOURCE, DISH_SOURCE, SERVING, TOMATO_SOURCE;
 *   rows are 5P + 10 bytes; at most 2 players, 100 cells; interactions strictly in player
 *   order; a pot starts cooking by itself with the third ingredient; no "start cooking"
 *   interaction; picking up a dish pays dish_pickup_rew when no dish lies on a counter and
 *   fewer players hold one than there are pots that could use it.
 * Where the reference has undefined behaviour (is_dish_pickup_useful reads agents[0..1] with one
 * player, sim.cpp:190-194) this follows its numpy twin (envs/overcooked2_reimplement.py:243-245:
 * never useful unless there are exactly two players).
 *
 * Pinned against envs/overcooked2_reimplement.py via tests/golden/simplecooked_*.npz.
 */
#include "mrl_oracle.h"

#include <stdlib.h>
#include <string.h>

enum { A_NORTH = 0, A_SOUTH = 1, A_EAST = 2, A_WEST = 3, A_STAY = 4, A_INTERACT = 5 };
enum { T_AIR = 0, T_POT, T_COUNTER, T_ONION_SRC, T_DISH_SRC, T_SERVING, T_TOMATO_SRC }; /* sim.hpp:40 */
enum { O_NONE = 0, O_TOMATO, O_ONION, O_DISH, O_SOUP };
#define MAX_INGREDIENTS 3

typedef struct {
    uint8_t name, num_onions, num_tomatoes;
    int8_t cooking_tick;
} item_t; /* sim.hpp:59-76 */

static const item_t ITEM_NONE = {O_NONE, 0, 0, -1};

typedef struct {
    uint8_t terrain;
    item_t object;
    int32_t past_player, past_orientation, current_player, future_player;
} cell_t; /* sim.hpp:136-147 */

typedef struct {
    uint8_t position, orientation, proposed_position, proposed_orientation;
    item_t held;
} player_t; /* sim.hpp:152-189 */

typedef struct {
    int32_t timestep;
    int32_t calculated_reward;
    int32_t should_update_pos;
    int32_t reset_now;
    uint8_t num_dishes_out;
} world_t;

struct orc_simplecooked {
    uint32_t n;
    uint8_t size, num_players, height, width;
    uint8_t start_x[ORC_SIMPLE_MAX_PLAYERS], start_y[ORC_SIMPLE_MAX_PLAYERS];
    uint8_t placement_in_pot_rew, dish_pickup_rew, soup_pickup_rew;
    uint8_t recipe_values[ORC_NUM_RECIPES], recipe_times[ORC_NUM_RECIPES];
    int64_t horizon;
    int num_pots;
    int pot_cell[ORC_SIMPLE_MAX_CELLS];
    int row_bytes; /* F = 5P + 10 */

    world_t *worlds;
    cell_t *cells;     /* N * C */
    player_t *players; /* N * P */
    uint8_t *obs;      /* N * P * C * F */
    int32_t *reward;   /* P * N */
    int32_t *done;     /* N */
};

static inline int recipe_of(const item_t *o) { return (MAX_INGREDIENTS + 1) * o->num_onions + o->num_tomatoes; }
static inline int ingredients_of(const item_t *o) { return (uint8_t)(o->num_onions + o->num_tomatoes); }
static inline int cook_time(const orc_simplecooked *s, const item_t *o) { return s->recipe_times[recipe_of(o)]; }
static inline int is_cooking(const orc_simplecooked *s, const item_t *o)
{
    return o->cooking_tick >= 0 && o->cooking_tick < cook_time(s, o);
}
static inline int is_ready(const orc_simplecooked *s, const item_t *o)
{
    return o->cooking_tick >= 0 && o->cooking_tick >= cook_time(s, o);
}

static inline int32_t shifted(int32_t point, int32_t dir, int64_t width)
{
    switch (dir) {
    case A_NORTH: return point - (int32_t)width;
    case A_SOUTH: return point + (int32_t)width;
    case A_EAST: return point + 1;
    case A_WEST: return point - 1;
    default: return point;
    }
}

/* sim.cpp:62-148 -- one (cell, viewer) row, updated in place */
static void observe_row(const orc_simplecooked *s, const cell_t *cells, const player_t *players, uint8_t *row, int row_id)
{
    const int loc = row_id % s->size;
    const int viewer = row_id / s->size;
    const int P = s->num_players;
    const int shift = 5 * P;
    const cell_t *dat = &cells[loc];
    const item_t *obj = &dat->object;

    for (int k = 5; k <= 9; k++) row[shift + k] = 0;

    if (obj->name == O_SOUP) {
        if (dat->terrain == T_POT) {
            row[shift + 5] = obj->num_onions;
            row[shift + 6] = obj->cooking_tick < 0 ? 0 : (uint8_t)obj->cooking_tick;
        } else {
            row[shift + 7] = 1;
        }
    } else if (obj->name == O_DISH) {
        row[shift + 8] = 1;
    } else if (obj->name == O_ONION) {
        row[shift + 9] = 1;
    }

    if (dat->past_player != -1) {
        int rel;
        if (dat->past_player == viewer) rel = 0;
        else if (dat->past_player < viewer) rel = dat->past_player + 1;
        else rel = dat->past_player;
        row[rel] = 0;
        row[P + 4 * rel + dat->past_orientation] = 0;
    }

    if (dat->current_player != -1) {
        const int other = dat->current_player;
        int i;
        if (other == viewer) i = 0;
        else if (other < viewer) i = other + 1;
        else i = other;
        const player_t *ps = &players[other];
        row[i] = 1;
        row[P + 4 * i + ps->orientation] = 1;
        if (ps->held.name == O_SOUP) row[shift + 7] = 1;
        else if (ps->held.name == O_DISH) row[shift + 8] = 1;
        else if (ps->held.name == O_ONION) row[shift + 9] = 1;
    }
}

/* sim.cpp:174-185 */
static int pot_states(const orc_simplecooked *s, const cell_t *cells)
{
    int non_empty = 0;
    for (int k = 0; k < s->num_pots; k++) {
        const item_t *o = &cells[s->pot_cell[k]].object;
        if (o->name != O_NONE && (o->cooking_tick >= 0 || ingredients_of(o) < MAX_INGREDIENTS)) non_empty++;
    }
    return non_empty;
}

/* sim.cpp:187-197 (two players); overcooked2_reimplement.py:243-245 otherwise */
static int dish_pickup_useful(const orc_simplecooked *s, const player_t *pls, int non_empty_pots)
{
    if (s->num_players != 2) return 0;
    int dishes = 0;
    for (int p = 0; p < 2; p++)
        if (pls[p].held.name == O_DISH) dishes++;
    return dishes < non_empty_pots;
}

static void step_world(orc_simplecooked *s, uint32_t wi, const int32_t *actions)
{
    const int P = s->num_players, C = s->size, F = s->row_bytes;
    world_t *w = &s->worlds[wi];
    cell_t *cells = &s->cells[(size_t)wi * C];
    player_t *pls = &s->players[(size_t)wi * P];
    uint8_t *obs = &s->obs[(size_t)wi * P * C * F];
#define ACT(p) (actions[(size_t)(p)*s->n + wi])

    /* resolve_interacts: sim.cpp:199-287, players strictly in ascending id */
    {
        const int pots = pot_states(s, cells);
        int rew = 0;
        for (int i = 0; i < P; i++) {
            player_t *pl = &pls[i];
            if (ACT(i) != A_INTERACT) continue;
            const int32_t i_pos = shifted(pl->position, pl->orientation, s->width);
            cell_t *dat = &cells[i_pos];
            item_t *soup = &dat->object;
            switch (dat->terrain) {
            case T_COUNTER:
                if (pl->held.name != O_NONE && soup->name == O_NONE) {
                    *soup = pl->held;
                    pl->held = ITEM_NONE;
                    if (soup->name == O_DISH) w->num_dishes_out++;
                } else if (pl->held.name == O_NONE && soup->name != O_NONE) {
                    if (soup->name == O_DISH) w->num_dishes_out--;
                    pl->held = *soup;
                    *soup = ITEM_NONE;
                }
                break;
            case T_ONION_SRC:
                if (pl->held.name == O_NONE) {
                    pl->held = ITEM_NONE;
                    pl->held.name = O_ONION;
                }
                break;
            case T_TOMATO_SRC:
                if (pl->held.name == O_NONE) {
                    pl->held = ITEM_NONE;
                    pl->held.name = O_TOMATO;
                }
                break;
            case T_DISH_SRC:
                if (pl->held.name == O_NONE) {
                    if (w->num_dishes_out == 0 && dish_pickup_useful(s, pls, pots)) rew += s->dish_pickup_rew;
                    pl->held = ITEM_NONE;
                    pl->held.name = O_DISH;
                }
                break;
            case T_POT:
                if (pl->held.name == O_DISH && soup->name == O_SOUP && is_ready(s, soup)) {
                    pl->held = *soup;
                    *soup = ITEM_NONE;
                    rew += s->soup_pickup_rew;
                } else if (pl->held.name == O_ONION || pl->held.name == O_TOMATO) {
                    if (soup->name == O_NONE) {
                        *soup = ITEM_NONE;
                        soup->name = O_SOUP;
                    }
                    if (!(soup->cooking_tick >= 0 || ingredients_of(soup) == MAX_INGREDIENTS)) {
                        const item_t obj = pl->held;
                        pl->held = ITEM_NONE;
                        if (obj.name == O_ONION) soup->num_onions++;
                        else soup->num_tomatoes++;
                        rew += s->placement_in_pot_rew;
                    }
                    /* soup_to_be_cooked_at_location && full: cooks by itself (sim.cpp:268-270) */
                    if (soup->name == O_SOUP && !is_cooking(s, soup) && !is_ready(s, soup) && ingredients_of(soup) > 0 &&
                        ingredients_of(soup) == MAX_INGREDIENTS)
                        soup->cooking_tick = 0;
                }
                break;
            case T_SERVING:
                if (pl->held.name == O_SOUP) {
                    const item_t obj = pl->held;
                    pl->held = ITEM_NONE;
                    rew += s->recipe_values[recipe_of(&obj)];
                }
                break;
            default: break;
            }
        }
        w->calculated_reward = rew;
    }

    /* movement: sim.cpp:289-348 */
    for (int p = 0; p < P; p++) {
        player_t *pl = &pls[p];
        const int32_t a = ACT(p);
        if (a == A_INTERACT) {
            pl->proposed_position = pl->position;
            pl->proposed_orientation = pl->orientation;
        } else {
            const int32_t np = shifted(pl->position, a, s->width);
            const int32_t no = (a == A_STAY) ? pl->orientation : a;
            pl->proposed_position = (uint8_t)(cells[np].terrain != T_AIR ? pl->position : np);
            pl->proposed_orientation = (uint8_t)no;
        }
        cells[pl->proposed_position].future_player = p;
    }
    for (int p = 0; p < P; p++) {
        player_t *pl = &pls[p];
        cell_t *orig = &cells[pl->position], *prop = &cells[pl->proposed_position];
        const int comp = prop->current_player;
        if (prop->future_player != p || (comp != -1 && comp != p && orig->future_player == comp)) w->should_update_pos = 0;
    }
    for (int p = 0; p < P; p++) {
        player_t *pl = &pls[p];
        cells[pl->position].current_player = -1;
        cells[pl->proposed_position].future_player = -1;
        cells[pl->position].past_player = p;
        cells[pl->position].past_orientation = pl->orientation;
    }
    for (int p = 0; p < P; p++) {
        player_t *pl = &pls[p];
        if (w->should_update_pos) pl->position = pl->proposed_position;
        pl->orientation = pl->proposed_orientation;
        cells[pl->position].current_player = p;
    }

    /* pots: sim.cpp:350-360 */
    for (int k = 0; k < s->num_pots; k++) {
        item_t *o = &cells[s->pot_cell[k]].object;
        if (o->name == O_SOUP && is_cooking(s, o)) o->cooking_tick++;
    }

    /* horizon: sim.cpp:415-420 */
    w->timestep += 1;
    w->reset_now = (w->timestep >= s->horizon);

    /* reset systems: sim.cpp:362-413 */
    w->should_update_pos = 1;
    if (w->reset_now) {
        w->timestep = 0;
        w->num_dishes_out = 0;
    }
    if (w->reset_now)
        for (int c = 0; c < C; c++) cells[c].object = ITEM_NONE;
    if (w->reset_now)
        for (int p = 0; p < P; p++) cells[pls[p].position].current_player = -1;
    for (int p = 0; p < P; p++) {
        s->reward[(size_t)p * s->n + wi] = w->calculated_reward;
        if (w->reset_now) {
            player_t *pl = &pls[p];
            pl->position = (uint8_t)(s->start_y[p] * s->width + s->start_x[p]);
            cells[pl->position].current_player = p;
            pl->orientation = A_NORTH;
            pl->proposed_position = pl->position;
            pl->proposed_orientation = pl->orientation;
            pl->held = ITEM_NONE;
        }
    }
    s->done[wi] = w->reset_now;

    /* observation rows: sim.cpp:445-449 */
    for (int r = 0; r < P * C; r++) observe_row(s, cells, pls, &obs[(size_t)r * F], r);
    for (int c = 0; c < C; c++) {
        cells[c].past_player = -1;
        cells[c].past_orientation = -1;
    }
#undef ACT
}

orc_simplecooked *orc_simplecooked_create(const orc_overcooked_config *cfg, uint32_t num_worlds)
{
    if (!cfg || cfg->height * cfg->width > ORC_SIMPLE_MAX_CELLS || cfg->height * cfg->width <= 0 || cfg->num_players <= 0 ||
        cfg->num_players > ORC_SIMPLE_MAX_PLAYERS)
        return NULL;
    orc_simplecooked *s = (orc_simplecooked *)calloc(1, sizeof(*s));
    s->n = num_worlds;
    s->height = (uint8_t)cfg->height;
    s->width = (uint8_t)cfg->width;
    s->size = (uint8_t)(cfg->height * cfg->width);
    s->num_players = (uint8_t)cfg->num_players;
    s->placement_in_pot_rew = (uint8_t)cfg->placement_in_pot_rew;
    s->dish_pickup_rew = (uint8_t)cfg->dish_pickup_rew;
    s->soup_pickup_rew = (uint8_t)cfg->soup_pickup_rew;
    s->horizon = cfg->horizon;
    for (int r = 0; r < ORC_NUM_RECIPES; r++) {
        s->recipe_values[r] = (uint8_t)cfg->recipe_values[r];
        s->recipe_times[r] = (uint8_t)cfg->recipe_times[r];
    }
    const int P = s->num_players, C = s->size;
    for (int p = 0; p < P; p++) {
        s->start_x[p] = (uint8_t)cfg->start_player_x[p];
        s->start_y[p] = (uint8_t)cfg->start_player_y[p];
    }
    for (int c = 0; c < C; c++)
        if (cfg->terrain[c] == T_POT) s->pot_cell[s->num_pots++] = c;
    s->row_bytes = 5 * P + 10;
    const int F = s->row_bytes;

    s->worlds = (world_t *)calloc(num_worlds, sizeof(world_t));
    s->cells = (cell_t *)calloc((size_t)num_worlds * C, sizeof(cell_t));
    s->players = (player_t *)calloc((size_t)num_worlds * P, sizeof(player_t));
    s->obs = (uint8_t *)calloc((size_t)num_worlds * P * C * F, 1);
    s->reward = (int32_t *)calloc((size_t)num_worlds * P, sizeof(int32_t));
    s->done = (int32_t *)calloc(num_worlds, sizeof(int32_t));

    /* Sim::Sim, sim.cpp:470-575 */
    for (uint32_t wi = 0; wi < num_worlds; wi++) {
        world_t *w = &s->worlds[wi];
        cell_t *cells = &s->cells[(size_t)wi * C];
        player_t *pls = &s->players[(size_t)wi * P];
        uint8_t *obs = &s->obs[(size_t)wi * P * C * F];
        for (int c = 0; c < C; c++) {
            cells[c].terrain = (uint8_t)cfg->terrain[c];
            cells[c].object = ITEM_NONE;
            cells[c].past_player = cells[c].past_orientation = -1;
            cells[c].current_player = cells[c].future_player = -1;
            for (int v = 0; v < P; v++) {
                uint8_t *row = &obs[(size_t)(v * C + c) * F];
                if (cells[c].terrain) row[cells[c].terrain - 1 + 5 * P] = 1;
            }
        }
        w->should_update_pos = 1;
        w->timestep = 0;
        w->num_dishes_out = 0;
        for (int p = 0; p < P; p++) {
            player_t *pl = &pls[p];
            pl->position = (uint8_t)(s->start_y[p] * s->width + s->start_x[p]);
            cells[pl->position].current_player = p;
            pl->orientation = A_NORTH;
            pl->proposed_position = pl->position;
            pl->proposed_orientation = pl->orientation;
            pl->held = ITEM_NONE;
        }
        w->reset_now = 0;
        for (int r = 0; r < P * C; r++) observe_row(s, cells, pls, &obs[(size_t)r * F], r);
    }
    return s;
}

void orc_simplecooked_destroy(orc_simplecooked *s)
{
    if (!s) return;
    free(s->worlds);
    free(s->cells);
    free(s->players);
    free(s->obs);
    free(s->reward);
    free(s->done);
    free(s);
}

void orc_simplecooked_step(orc_simplecooked *s, const int32_t *actions, int num_threads)
{
    const long n = (long)s->n;
    if (num_threads < 1) num_threads = 1;
#pragma omp parallel for schedule(static) num_threads(num_threads)
    for (long wi = 0; wi < n; wi++) step_world(s, (uint32_t)wi, actions);
}

const uint8_t *orc_simplecooked_obs(const orc_simplecooked *s) { return s->obs; }
const int32_t *orc_simplecooked_reward(const orc_simplecooked *s) { return s->reward; }
const int32_t *orc_simplecooked_done(const orc_simplecooked *s) { return s->done; }

void orc_simplecooked_dump(const orc_simplecooked *s, uint8_t *players, uint8_t *objects, int32_t *timestep, int32_t *dishes_out)
{
    const int P = s->num_players, C = s->size;
    for (uint32_t wi = 0; wi < s->n; wi++) {
        for (int p = 0; p < P; p++) {
            const player_t *pl = &s->players[(size_t)wi * P + p];
            uint8_t *o = &players[((size_t)wi * P + p) * 6];
            o[0] = pl->position;
            o[1] = pl->orientation;
            o[2] = pl->held.name;
            o[3] = pl->held.num_onions;
            o[4] = pl->held.num_tomatoes;
            o[5] = (uint8_t)pl->held.cooking_tick;
        }
        for (int c = 0; c < C; c++) {
            const item_t *it = &s->cells[(size_t)wi * C + c].object;
            uint8_t *o = &objects[((size_t)wi * C + c) * 4];
            o[0] = it->name;
            o[1] = it->num_onions;
            o[2] = it->num_tomatoes;
            o[3] = (uint8_t)it->cooking_tick;
        }
        timestep[wi] = s->worlds[wi].timestep;
        dishes_out[wi] = s->worlds[wi].num_dishes_out;
    }
}
