/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle, Overcooked world step.
 *
 * Plain-C restatement of the reference task graph
 *   /root/reference/src/overcooked_env/sim.cpp:498-537 (graph),
 *   :199-358 (interactions), :361-426 (movement/collisions), :428-438 (pots),
 *   :440-495 (reset + bookkeeping), :68-167 (observation rows), :556-659 (init)
 * with the component widths of sim.hpp:59-184 (uint8 positions, int8 tick,
 * uint8 reward parameters) kept as they are, and the reference's *incremental*
 * observation update (player channels are cleared through past_player only).
 * The HIP kernel recomputes rows from scratch instead; agreement between the
 * two is what the GPU parity tests establish.
 *
 * Systems run per world in the dependency order of the graph, players in
 * ascending id (what one worker of the reference's CPU executor does).
 *
 * Pinned against envs/overcooked_reimplement.py via tests/golden/overcooked_*.npz.
 */
#include "mrl_oracle.h"

#include <stdlib.h>
#include <string.h>

enum { A_NORTH = 0, A_SOUTH = 1, A_EAST = 2, A_WEST = 3, A_STAY = 4, A_INTERACT = 5 };
enum { T_AIR = 0, T_POT, T_COUNTER, T_ONION_SRC, T_TOMATO_SRC, T_DISH_SRC, T_SERVING };
enum { O_NONE = 0, O_TOMATO, O_ONION, O_DISH, O_SOUP };
#define MAX_INGREDIENTS 3

typedef struct {
    uint8_t name, num_onions, num_tomatoes;
    int8_t cooking_tick;
} item_t; /* sim.hpp:59-74 */

static const item_t ITEM_NONE = {O_NONE, 0, 0, -1};

typedef struct {
    uint8_t terrain;
    item_t object;
    int32_t past_player, past_orientation, current_player, future_player;
    int32_t interacting_players[4];
    int32_t num_interacting_players;
} cell_t; /* sim.hpp:133-144 */

typedef struct {
    uint8_t position, orientation, proposed_position, proposed_orientation;
    item_t held;
    int8_t interaction_index;
} player_t; /* sim.hpp:149-184 */

typedef struct {
    int32_t timestep;
    int32_t calculated_reward;
    int32_t should_update_pos;
    int32_t reset_now;
} world_t;

struct orc_overcooked {
    uint32_t n;
    /* the per-world constant copy the reference keeps in WorldState (sim.hpp:83-105) */
    uint8_t size, num_players, height, width;
    uint8_t start_x[ORC_MAX_PLAYERS], start_y[ORC_MAX_PLAYERS];
    uint8_t placement_in_pot_rew, dish_pickup_rew, soup_pickup_rew;
    uint8_t recipe_values[ORC_NUM_RECIPES], recipe_times[ORC_NUM_RECIPES];
    int64_t horizon;
    int num_pots;
    int pot_cell[ORC_MAX_CELLS];
    int row_bytes; /* F = 5P + 16 */

    world_t *worlds;
    cell_t *cells;     /* N * C */
    player_t *players; /* N * P */
    uint8_t *obs;      /* N * P * C * F */
    int32_t *reward;   /* P * N */
    int32_t *done;     /* N */
};

static inline int recipe_of(const item_t *o) { return (MAX_INGREDIENTS + 1) * o->num_onions + o->num_tomatoes; }
static inline int ingredients_of(const item_t *o) { return (uint8_t)(o->num_onions + o->num_tomatoes); }
static inline int cook_time(const orc_overcooked *s, const item_t *o) { return s->recipe_times[recipe_of(o)]; }
static inline int is_cooking(const orc_overcooked *s, const item_t *o)
{
    return o->cooking_tick >= 0 && o->cooking_tick < cook_time(s, o);
}
static inline int is_ready(const orc_overcooked *s, const item_t *o)
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

/* sim.cpp:68-167 -- one (cell, viewer) row, updated in place */
static void observe_row(const orc_overcooked *s, const world_t *w, const cell_t *cells, const player_t *players,
                        uint8_t *row, int row_id)
{
    const int loc = row_id % s->size;
    const int viewer = row_id / s->size;
    const int P = s->num_players;
    const int shift = 5 * P;
    const cell_t *dat = &cells[loc];
    const item_t *obj = &dat->object;

    row[shift + 15] = (s->horizon - w->timestep < 40) ? 1 : 0;
    for (int k = 6; k <= 14; k++) row[shift + k] = 0;

    if (obj->name == O_SOUP) {
        if (dat->terrain == T_POT) {
            if (obj->cooking_tick < 0) {
                row[shift + 6] = obj->num_onions;
                row[shift + 7] = obj->num_tomatoes;
            } else {
                row[shift + 8] = obj->num_onions;
                row[shift + 9] = obj->num_tomatoes;
                row[shift + 10] = (uint8_t)(cook_time(s, obj) - obj->cooking_tick);
                if (is_ready(s, obj)) row[shift + 11] = 1;
            }
        } else {
            row[shift + 8] = obj->num_onions;
            row[shift + 9] = obj->num_tomatoes;
            row[shift + 10] = 0;
            row[shift + 11] = 1;
        }
    } else if (obj->name == O_DISH) {
        row[shift + 12] = 1;
    } else if (obj->name == O_ONION) {
        row[shift + 13] = 1;
    } else if (obj->name == O_TOMATO) {
        row[shift + 14] = 1;
    }

    if (dat->past_player != -1) {
        int rel = dat->past_player == viewer ? 0 : (dat->past_player < viewer ? dat->past_player + 1 : dat->past_player);
        row[rel] = 0;
        row[P + 4 * rel + dat->past_orientation] = 0;
    }

    if (dat->current_player != -1) {
        int other = dat->current_player;
        int rel = other == viewer ? 0 : (other < viewer ? other + 1 : other);
        const player_t *ps = &players[other];
        row[rel] = 1;
        row[P + 4 * rel + ps->orientation] = 1;
        if (ps->held.name != O_NONE) {
            const item_t *h = &ps->held;
            if (h->name == O_SOUP) {
                row[shift + 8] = h->num_onions;
                row[shift + 9] = h->num_tomatoes;
                row[shift + 10] = 0;
                row[shift + 11] = 1;
            } else if (h->name == O_DISH) {
                row[shift + 12] = 1;
            } else if (h->name == O_ONION) {
                row[shift + 13] = 1;
            } else if (h->name == O_TOMATO) {
                row[shift + 14] = 1;
            }
        }
    }
}

/* sim.cpp:208-255 */
static void resolve_interact(const orc_overcooked *s, world_t *w, cell_t *cells, player_t *pl, int id, int32_t action)
{
    pl->interaction_index = -1;
    if (action != A_INTERACT) return;
    int32_t tgt = shifted(pl->position, pl->orientation, s->width);
    cell_t *dat = &cells[tgt];
    switch (dat->terrain) {
    case T_COUNTER:
    case T_POT: {
        int slot = dat->num_interacting_players++;
        dat->interacting_players[slot] = id;
        break;
    }
    case T_ONION_SRC:
        if (pl->held.name == O_NONE) pl->held = (item_t){O_ONION, 0, 0, -1};
        break;
    case T_TOMATO_SRC:
        if (pl->held.name == O_NONE) pl->held = (item_t){O_TOMATO, 0, 0, -1};
        break;
    case T_DISH_SRC:
        if (pl->held.name == O_NONE) pl->held = (item_t){O_DISH, 0, 0, -1};
        break;
    case T_SERVING:
        if (pl->held.name == O_SOUP) {
            item_t soup = pl->held;
            pl->held = ITEM_NONE;
            w->calculated_reward += s->recipe_values[recipe_of(&soup)];
        }
        break;
    default: break;
    }
}

/* sim.cpp:259-282 */
static void rank_interact(const orc_overcooked *s, cell_t *cells, player_t *pl, int id, int32_t action)
{
    if (action != A_INTERACT) return;
    int32_t tgt = shifted(pl->position, pl->orientation, s->width);
    cell_t *dat = &cells[tgt];
    if (dat->terrain == T_COUNTER || dat->terrain == T_POT) {
        pl->interaction_index = 0;
        for (int k = 0; k < dat->num_interacting_players; k++)
            if (dat->interacting_players[k] < id) pl->interaction_index++;
    }
}

/* sim.cpp:286-338 */
static void counter_pot_phase(const orc_overcooked *s, world_t *w, cell_t *cells, player_t *pl, int phase)
{
    if (pl->interaction_index != phase) return;
    int32_t tgt = shifted(pl->position, pl->orientation, s->width);
    uint8_t terrain = cells[tgt].terrain;
    item_t *there = &cells[tgt].object;
    int holding = pl->held.name != O_NONE;

    if (terrain == T_COUNTER) {
        if (holding && there->name == O_NONE) {
            *there = pl->held;
            pl->held = ITEM_NONE;
        } else if (!holding && there->name != O_NONE) {
            pl->held = *there;
            *there = ITEM_NONE;
        }
    } else if (terrain == T_POT) {
        if (!holding) {
            if (there->name == O_SOUP && !is_cooking(s, there) && !is_ready(s, there) && ingredients_of(there) > 0)
                there->cooking_tick = 0;
        } else if (pl->held.name == O_DISH && there->name == O_SOUP && is_ready(s, there)) {
            pl->held = *there;
            *there = ITEM_NONE;
            w->calculated_reward += s->soup_pickup_rew;
        } else if (pl->held.name == O_ONION || pl->held.name == O_TOMATO) {
            if (there->name == O_NONE) *there = (item_t){O_SOUP, 0, 0, -1};
            if (!(there->cooking_tick >= 0 || ingredients_of(there) == MAX_INGREDIENTS)) {
                item_t put = pl->held;
                pl->held = ITEM_NONE;
                if (put.name == O_ONION)
                    there->num_onions++;
                else
                    there->num_tomatoes++;
                w->calculated_reward += s->placement_in_pot_rew;
            }
        }
    }
}

static void step_world(orc_overcooked *s, uint32_t wi, const int32_t *actions)
{
    const int P = s->num_players, C = s->size, F = s->row_bytes;
    world_t *w = &s->worlds[wi];
    cell_t *cells = &s->cells[(size_t)wi * C];
    player_t *pls = &s->players[(size_t)wi * P];
    uint8_t *obs = &s->obs[(size_t)wi * P * C * F];
#define ACT(p) actions[(size_t)(p) * s->n + wi]

    /* interaction chain: sim.cpp:501-507 */
    w->calculated_reward = 0;
    for (int p = 0; p < P; p++) resolve_interact(s, w, cells, &pls[p], p, ACT(p));
    for (int p = 0; p < P; p++) rank_interact(s, cells, &pls[p], p, ACT(p));
    for (int phase = 0; phase < 4; phase++)
        for (int p = 0; p < P; p++) counter_pot_phase(s, w, cells, &pls[p], phase);

    /* movement chain: sim.cpp:510-515 */
    for (int p = 0; p < P; p++) {
        player_t *pl = &pls[p];
        int32_t a = ACT(p);
        if (a == A_INTERACT) {
            pl->proposed_position = pl->position;
            pl->proposed_orientation = pl->orientation;
        } else {
            int32_t np = shifted(pl->position, a, s->width);
            int32_t no = (a == A_STAY) ? pl->orientation : a;
            pl->proposed_position = (uint8_t)(cells[np].terrain != T_AIR ? pl->position : np);
            pl->proposed_orientation = (uint8_t)no;
        }
        cells[pl->proposed_position].future_player = p;
    }
    for (int p = 0; p < P; p++) {
        player_t *pl = &pls[p];
        cell_t *orig = &cells[pl->position], *prop = &cells[pl->proposed_position];
        int comp = prop->current_player;
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

    /* pots: sim.cpp:430-438 */
    for (int k = 0; k < s->num_pots; k++) {
        item_t *o = &cells[s->pot_cell[k]].object;
        if (o->name == O_SOUP && is_cooking(s, o)) o->cooking_tick++;
    }

    /* horizon: sim.cpp:485-489 */
    w->timestep += 1;
    w->reset_now = (w->timestep >= s->horizon);

    /* reset systems: sim.cpp:441-482 */
    w->should_update_pos = 1;
    if (w->reset_now) w->timestep = 0;
    for (int c = 0; c < C; c++) {
        cells[c].num_interacting_players = 0;
        if (w->reset_now) cells[c].object = ITEM_NONE;
    }
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

    /* observation rows: sim.cpp:532-534 */
    for (int r = 0; r < P * C; r++) observe_row(s, w, cells, pls, &obs[(size_t)r * F], r);
    for (int c = 0; c < C; c++) {
        cells[c].past_player = -1;
        cells[c].past_orientation = -1;
    }
#undef ACT
}

orc_overcooked *orc_overcooked_create(const orc_overcooked_config *cfg, uint32_t num_worlds)
{
    if (!cfg || cfg->height * cfg->width > ORC_MAX_CELLS || cfg->height * cfg->width <= 0 || cfg->num_players <= 0 ||
        cfg->num_players > ORC_MAX_PLAYERS)
        return NULL;
    orc_overcooked *s = (orc_overcooked *)calloc(1, sizeof(*s));
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
    s->row_bytes = 5 * P + 16;
    const int F = s->row_bytes;

    s->worlds = (world_t *)calloc(num_worlds, sizeof(world_t));
    s->cells = (cell_t *)calloc((size_t)num_worlds * C, sizeof(cell_t));
    s->players = (player_t *)calloc((size_t)num_worlds * P, sizeof(player_t));
    s->obs = (uint8_t *)calloc((size_t)num_worlds * P * C * F, 1);
    s->reward = (int32_t *)calloc((size_t)num_worlds * P, sizeof(int32_t));
    s->done = (int32_t *)calloc(num_worlds, sizeof(int32_t));

    /* Sim::Sim, sim.cpp:556-659 */
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
        /* initial reset with resetNow = true (sim.cpp:649-651) */
        w->should_update_pos = 1;
        w->timestep = 0;
        for (int p = 0; p < P; p++) {
            player_t *pl = &pls[p];
            pl->position = (uint8_t)(s->start_y[p] * s->width + s->start_x[p]);
            cells[pl->position].current_player = p;
            pl->orientation = A_NORTH;
            pl->proposed_position = pl->position;
            pl->proposed_orientation = pl->orientation;
            pl->held = ITEM_NONE;
            pl->interaction_index = -1;
        }
        w->reset_now = 0;
        for (int r = 0; r < P * C; r++) observe_row(s, w, cells, pls, &obs[(size_t)r * F], r);
    }
    return s;
}

void orc_overcooked_destroy(orc_overcooked *s)
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

void orc_overcooked_step(orc_overcooked *s, const int32_t *actions, int num_threads)
{
    const long n = (long)s->n;
    if (num_threads < 1) num_threads = 1;
#pragma omp parallel for schedule(static) num_threads(num_threads)
    for (long wi = 0; wi < n; wi++) step_world(s, (uint32_t)wi, actions);
}

const uint8_t *orc_overcooked_obs(const orc_overcooked *s) { return s->obs; }
const int32_t *orc_overcooked_reward(const orc_overcooked *s) { return s->reward; }
const int32_t *orc_overcooked_done(const orc_overcooked *s) { return s->done; }

void orc_overcooked_dump(const orc_overcooked *s, uint8_t *players, uint8_t *objects, int32_t *timestep)
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
    }
}
