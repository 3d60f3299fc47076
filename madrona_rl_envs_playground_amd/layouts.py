"""Overcooked layout data and the layout -> simulator-config transform.

Mirrors ``get_base_layout_params`` of the reference
(/root/reference/envs/overcooked_env.py:226-371): a layout dict (``grid`` plus
optional order / timing / value / shaping keys) becomes the keyword arguments of
``OvercookedSimulator`` (terrain ints, start positions, 16-entry recipe tables
indexed ``4*onions + tomatoes``, shaping rewards, horizon).

The reference reads layouts through ``overcooked_ai_py`` (not installed here) or
``eval``s a ``.layout`` file.  The five standard grids below are the ones the
reference maps those names to (/root/reference/overcooked_demo/overcooked_utils.py:7-14:
cramped_room=simple, asymmetric_advantages=unident_s, coordination_ring=random1,
forced_coordination=random0, counter_circuit=random3), written in the new-style
format (``start_all_orders``) the transform expects; ``.layout`` files are parsed
with ``ast.literal_eval`` instead of ``eval``.
"""
import ast
import os

MAX_NUM_INGREDIENTS = 3
NUM_RECIPES = (MAX_NUM_INGREDIENTS + 1) ** 2

BASE_REW_SHAPING_PARAMS = {
    "PLACEMENT_IN_POT_REW": 3,
    "DISH_PICKUP_REWARD": 0,
    "SOUP_PICKUP_REWARD": 5,
}

# glyph -> TerrainT value (src/overcooked_env/sim.hpp:40)
TERRAIN_GLYPHS = [" ", "P", "X", "O", "T", "D", "S"]
# glyph -> player index (envs/overcooked_env.py:243-246)
PLAYER_GLYPHS = list("1234567890") + list("!@#$%^&*()") + list("abcdefghij") + list("klmnopqrst")

_THREE_ONIONS = [{"ingredients": ["onion", "onion", "onion"]}]

LAYOUTS = {
    "cramped_room": {
        "grid": """XXPXX
                   O  2O
                   X1  X
                   XDXSX""",
        "start_all_orders": _THREE_ONIONS,
        "rew_shaping_params": None,
    },
    "asymmetric_advantages": {
        "grid": """XXXXXXXXX
                   O XSXOX S
                   X   P 1 X
                   X2  P   X
                   XXXDXDXXX""",
        "start_all_orders": _THREE_ONIONS,
        "rew_shaping_params": None,
    },
    "coordination_ring": {
        "grid": """XXXPX
                   X 1 P
                   D2X X
                   O   X
                   XOSXX""",
        "start_all_orders": _THREE_ONIONS,
        "rew_shaping_params": None,
    },
    "forced_coordination": {
        "grid": """XXXPX
                   O X1P
                   O2X X
                   D X X
                   XXXSX""",
        "start_all_orders": _THREE_ONIONS,
        "rew_shaping_params": None,
    },
    "counter_circuit": {
        "grid": """XXXPPXXX
                   X  2   X
                   D XXXX S
                   X  1   X
                   XXXOOXXX""",
        "start_all_orders": _THREE_ONIONS,
        "rew_shaping_params": None,
    },
    # 4 players, mixed soups; exercises P > 2 and tomato recipes
    "multiplayer_schelling": {
        "grid": """XXSPDXX
                   X  1  X
                   X  X  X
                   O3   4O
                   X  X  X
                   X  2  X
                   XXDPSXX""",
        "start_all_orders": _THREE_ONIONS,
        "rew_shaping_params": None,
    },
    # tomato + onion sources, two recipes with different values/times and a bonus order
    "asymmetric_advantages_tomato": {
        "grid": """XXXXXXXXX
                   O XSXOX T
                   X   P 1 X
                   X2  P   X
                   XXXDXDXXX""",
        "start_all_orders": [
            {"ingredients": ["onion", "onion", "onion"]},
            {"ingredients": ["tomato", "tomato", "tomato"]},
            {"ingredients": ["onion", "tomato"]},
        ],
        "start_bonus_orders": [{"ingredients": ["tomato", "tomato", "tomato"]}],
        "onion_value": 7,
        "tomato_value": 5,
        "onion_time": 4,
        "tomato_time": 3,
        "rew_shaping_params": None,
    },
}

STANDARD_LAYOUTS = ["cramped_room", "asymmetric_advantages", "coordination_ring", "forced_coordination",
                    "counter_circuit"]



def _many_player_layout():
    """The 40-player 15x17 stress layout of the reference's player-scaling table
    (/root/reference/src/overcooked_env/README.org:115-121): 8 corridors of 5
    players, each above a row of tomato/pot/onion/dish stations, serving windows
    on both sides.  Built programmatically (it is a regular pattern)."""
    rows = ["X" * 15]
    for band in range(8):
        glyphs = PLAYER_GLYPHS[5 * band:5 * band + 5]
        rows.append("X" + "  ".join(glyphs) + "X")
        if band < 7:
            rows.append("S TX PX OX DX S")
    rows.append("X" * 15)
    return {
        "grid": "\n".join(rows),
        "start_all_orders": [
            {"ingredients": ["onion", "onion", "onion"]},
            {"ingredients": ["onion", "onion", "tomato"]},
            {"ingredients": ["tomato", "tomato", "tomato"]},
            {"ingredients": ["tomato"]},
        ],
        "start_bonus_orders": [
            {"ingredients": ["tomato", "tomato", "tomato"]},
            {"ingredients": ["onion", "onion", "tomato"]},
        ],
        "onion_value": 21,
        "tomato_value": 13,
        "onion_time": 15,
        "tomato_time": 7,
    }


LAYOUTS["many_player_layout"] = _many_player_layout()


def load_layout_file(path):
    """Parse a ``.layout`` file: one Python-literal dict (triple-quoted grid)."""
    with open(path, "r") as f:
        return ast.literal_eval(f.read())


def read_layout_dict(layout_name):
    if layout_name not in LAYOUTS:
        raise KeyError(f"unknown layout {layout_name!r}; known: {sorted(LAYOUTS)} or a path ending in .layout")
    d = dict(LAYOUTS[layout_name])
    return d


def _recipe_index(order):
    onions = sum(1 for x in order["ingredients"] if x == "onion")
    tomatoes = sum(1 for x in order["ingredients"] if x == "tomato")
    return (MAX_NUM_INGREDIENTS + 1) * onions + tomatoes


def _order_flags(orders):
    flags = [0] * NUM_RECIPES
    for order in orders:
        flags[_recipe_index(order)] = 1
    return flags


def _per_ingredient_table(onion_unit, tomato_unit):
    return [o * onion_unit + t * tomato_unit
            for o in range(MAX_NUM_INGREDIENTS + 1) for t in range(MAX_NUM_INGREDIENTS + 1)]


def get_base_layout_params(layout_name, horizon, max_num_players=None):
    """Same contract as the reference function of this name
    (envs/overcooked_env.py:261-371); keys consumed by the transform are removed,
    unknown keys are passed through untouched like the reference does."""
    if isinstance(layout_name, dict):
        params = dict(layout_name)
    elif layout_name.endswith(".layout"):
        params = load_layout_file(layout_name)
    else:
        params = read_layout_dict(layout_name)

    rows = [row.strip() for row in params.pop("grid").split("\n")]
    cells = [list(row) for row in rows]

    starts = [None] * 64
    for y, row in enumerate(cells):
        for x, glyph in enumerate(row):
            if glyph in PLAYER_GLYPHS:
                row[x] = " "
                idx = PLAYER_GLYPHS.index(glyph)
                if max_num_players is None or idx < max_num_players:
                    starts[idx] = (x, y)
    num_players = sum(1 for s in starts if s is not None)
    starts = starts[:num_players]

    params["height"] = len(cells)
    params["width"] = len(cells[0])
    params["terrain"] = [TERRAIN_GLYPHS.index(g) for row in cells for g in row]
    params["num_players"] = len(starts)
    params["start_player_x"] = [s[0] for s in starts]
    params["start_player_y"] = [s[1] for s in starts]

    shaping = params.pop("rew_shaping_params", None) or BASE_REW_SHAPING_PARAMS
    params["placement_in_pot_rew"] = shaping["PLACEMENT_IN_POT_REW"]
    params["dish_pickup_rew"] = shaping["DISH_PICKUP_REWARD"]
    params["soup_pickup_rew"] = shaping["SOUP_PICKUP_REWARD"]

    all_orders = params.pop("start_all_orders", None) or []
    bonus_orders = params.pop("start_bonus_orders", None) or []
    wanted = _order_flags(all_orders)
    bonus = _order_flags(bonus_orders)
    order_bonus = params.pop("order_bonus", 2)

    times = [20] * NUM_RECIPES
    if "onion_time" in params and "tomato_time" in params:
        times = _per_ingredient_table(params.pop("onion_time"), params.pop("tomato_time"))
    if "recipe_times" in params:
        for order, t in zip(all_orders, params["recipe_times"]):
            times[_recipe_index(order)] = t
    if "cook_time" in params:
        times = [params.pop("cook_time")] * NUM_RECIPES
    params["recipe_times"] = times

    values = [20] * NUM_RECIPES
    if "onion_value" in params and "tomato_value" in params:
        values = _per_ingredient_table(params.pop("onion_value"), params.pop("tomato_value"))
    if "recipe_values" in params:
        for order, v in zip(all_orders, params["recipe_values"]):
            values[_recipe_index(order)] = v
    if "delivery_reward" in params:
        values = [params.pop("delivery_reward")] * NUM_RECIPES
    for i in range(NUM_RECIPES):
        if bonus[i]:
            values[i] *= order_bonus
        if not wanted[i]:
            values[i] = 0
    params["recipe_values"] = values

    params["horizon"] = horizon
    return params


# ---------------------------------------------------------------------------------------------
# "Simplecooked" (overcooked2_env): the old-style layouts and their transform
# (/root/reference/envs/overcooked2_env.py:119-258; layout files
# /root/reference/oldercooked_ai/oldercooked_ai_py/data/layouts/*.layout).  Differences from the
# transform above: the terrain enum puts the tomato source last (src/overcooked2_env/sim.hpp:40),
# dish pickup is shaped (3), recipe values are NOT zeroed for unordered soups, old keys
# (`start_order_list`, `num_items_for_soup`) are dropped.
# ---------------------------------------------------------------------------------------------
SIMPLECOOKED_REW_SHAPING_PARAMS = {
    "PLACEMENT_IN_POT_REW": 3,
    "DISH_PICKUP_REWARD": 3,
    "SOUP_PICKUP_REWARD": 5,
}

# glyph -> TerrainT value (src/overcooked2_env/sim.hpp:40)
SIMPLECOOKED_TERRAIN_GLYPHS = [" ", "P", "X", "O", "D", "S", "T"]


def _old_style(grid):
    return {"grid": grid, "start_order_list": None, "cook_time": 20, "num_items_for_soup": 3,
            "delivery_reward": 20, "rew_shaping_params": None}


# the reference's names for the five standard grids (overcooked_demo/overcooked_utils.py:7-14) and a tomato variant
SIMPLECOOKED_LAYOUTS = {
    "simple": _old_style(LAYOUTS["cramped_room"]["grid"]),
    "unident_s": _old_style(LAYOUTS["asymmetric_advantages"]["grid"]),
    "random1": _old_style(LAYOUTS["coordination_ring"]["grid"]),
    "random0": _old_style(LAYOUTS["forced_coordination"]["grid"]),
    "random3": _old_style(LAYOUTS["counter_circuit"]["grid"]),
    "simple_tomato": _old_style("""XXPXX
                                   T  2T
                                   X1  O
                                   XXDSX"""),
}
SIMPLECOOKED_STANDARD_LAYOUTS = ["simple", "unident_s", "random1", "random0", "random3"]


def get_simplecooked_layout_params(layout_name, horizon, max_num_players=None):
    """``get_base_layout_params`` of the reference's envs/overcooked2_env.py (:155-258): the keyword
    arguments of ``SimplecookedSimulator``."""
    if isinstance(layout_name, dict):
        params = dict(layout_name)
    elif layout_name.endswith(".layout"):
        params = load_layout_file(layout_name)
    else:
        if layout_name not in SIMPLECOOKED_LAYOUTS:
            raise KeyError(f"unknown layout {layout_name!r}; known: {sorted(SIMPLECOOKED_LAYOUTS)} or a path ending in .layout")
        params = dict(SIMPLECOOKED_LAYOUTS[layout_name])
    grid = params.pop("grid")
    params.pop("start_order_list", None)
    params.pop("num_items_for_soup", None)

    cells = [list(row.strip()) for row in grid.split("\n")]
    starts = [None] * 64
    for y, row in enumerate(cells):
        for x, glyph in enumerate(row):
            if glyph in PLAYER_GLYPHS:
                row[x] = " "
                idx = PLAYER_GLYPHS.index(glyph)
                if max_num_players is None or idx < max_num_players:
                    starts[idx] = (x, y)
    num_players = sum(1 for s in starts if s is not None)
    starts = starts[:num_players]

    params["height"] = len(cells)
    params["width"] = len(cells[0])
    params["terrain"] = [SIMPLECOOKED_TERRAIN_GLYPHS.index(g) for row in cells for g in row]
    params["num_players"] = len(starts)
    params["start_player_x"] = [s[0] for s in starts]
    params["start_player_y"] = [s[1] for s in starts]

    shaping = params.pop("rew_shaping_params", None) or SIMPLECOOKED_REW_SHAPING_PARAMS
    params["placement_in_pot_rew"] = shaping["PLACEMENT_IN_POT_REW"]
    params["dish_pickup_rew"] = shaping["DISH_PICKUP_REWARD"]
    params["soup_pickup_rew"] = shaping["SOUP_PICKUP_REWARD"]

    all_orders = params.pop("start_all_orders", None) or []
    params.pop("start_bonus_orders", None)
    params.pop("order_bonus", None)

    times = [20] * NUM_RECIPES
    if "onion_time" in params and "tomato_time" in params:
        times = _per_ingredient_table(params.pop("onion_time"), params.pop("tomato_time"))
    if "recipe_times" in params:
        for order, t in zip(all_orders, params["recipe_times"]):
            times[_recipe_index(order)] = t
    if "cook_time" in params:
        times = [params.pop("cook_time")] * NUM_RECIPES
    params["recipe_times"] = times

    values = [20] * NUM_RECIPES
    if "onion_value" in params and "tomato_value" in params:
        values = _per_ingredient_table(params.pop("onion_value"), params.pop("tomato_value"))
    if "recipe_values" in params:
        for order, v in zip(all_orders, params["recipe_values"]):
            values[_recipe_index(order)] = v
    if "delivery_reward" in params:
        values = [params.pop("delivery_reward")] * NUM_RECIPES
    params["recipe_values"] = values

    params["horizon"] = horizon
    return params
