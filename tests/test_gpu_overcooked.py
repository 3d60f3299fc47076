"""GPU parity of the Overcooked HIP step (through the C ABI) against
(a) the golden vectors produced by the reference's numpy implementation and
(b) the CPU oracle on the same seeded action streams -- bit-exact
(integer/byte state)."""
import glob
import zlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

from madrona_rl_envs_playground_amd import layouts  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import ExecMode, OvercookedSimulator  # noqa: E402


def make_sim(params, n):
    return OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)


def world_major(sim):
    return sim.observation_world_major_tensor().to_torch()


def unpack_players(t):
    """(N,P,8) uint8 -> (N,P,6) in the oracle's dump order."""
    t = t.cpu().numpy()
    return np.stack([t[..., 0], t[..., 1], t[..., 4], t[..., 5], t[..., 6], t[..., 7]], axis=-1)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "overcooked_*.npz"))),
                         ids=lambda p: os.path.basename(p)[11:-4])
def test_golden_vectors(path, hip_lib):
    """Every world of a small batch replays the fixture's action stream and must
    reproduce the reference's observations, rewards and dones exactly."""
    z = np.load(path)
    params = json.loads(str(z["params"]))
    acts, obs, rew, done = z["actions"], z["obs"], z["reward"], z["done"]
    P, C, F = params["num_players"], params["height"] * params["width"], 5 * params["num_players"] + 16
    n = 5  # odd, so the last workgroup is partly empty and blocks straddle 16-byte alignment
    sim = make_sim(params, n)
    o = world_major(sim).view(n, P, C, F)
    a = sim.action_tensor().to_torch()
    r = sim.reward_tensor().to_torch()
    d = sim.done_tensor().to_torch()
    assert np.array_equal(o.cpu().numpy().astype(np.uint8), np.broadcast_to(obs[0], (n, P, C, F)))
    for t in range(len(acts)):
        a.copy_(torch.from_numpy(acts[t].astype(np.int32)).cuda()[:, None, None].expand(P, n, 1))
        sim.step()
        got = o.cpu().numpy().astype(np.uint8)
        assert np.array_equal(got, np.broadcast_to(obs[t + 1], (n, P, C, F))), f"obs differ at step {t}"
        assert (r.cpu().numpy() == rew[t]).all(), f"reward differs at step {t}"
        assert (d.cpu().numpy() == done[t]).all(), f"done differs at step {t}"
    sim.close()


@pytest.mark.parametrize("layout,horizon,cap,n,steps", [
    ("cramped_room", 400, None, 1000, 450),
    ("cramped_room", 37, None, 4099, 120),
    ("asymmetric_advantages", 60, None, 513, 150),
    ("coordination_ring", 50, None, 777, 150),
    ("forced_coordination", 50, None, 256, 120),
    ("counter_circuit", 50, None, 300, 120),
    ("multiplayer_schelling", 45, None, 301, 130),
    ("asymmetric_advantages_tomato", 80, None, 640, 200),
    ("many_player_layout", 30, 8, 33, 70),
    ("many_player_layout", 25, None, 9, 60),
    ("multiplayer_schelling", 40, 3, 130, 120),      # odd player count: byte-wise row tails
    ("cramped_room", 30, 1, 77, 100),                # a single player
    ("many_player_layout", 20, 5, 40, 60),
    ("many_player_layout", 30, 2, 50, 70),           # few worlds of a large layout: four waves share a world
    ("many_player_layout", 22, 3, 21, 50),           # the same with an odd player count
    ("many_player_layout", 26, 4, 35, 60),
    ("cramped_room", 400, None, 40000, 12),          # 8 worlds per wave, ragged last group
    ("many_player_layout", 30, 2, 1100, 70),         # a 13 KB world as one wave's single-pass tile (from 384 worlds on), slabs off 16-byte boundaries
    ("many_player_layout", 30, 2, 8200, 8),
])
def test_against_oracle(layout, horizon, cap, n, steps, hip_lib, oracle_lib):
    """Independent random actions per world; obs, reward, done and the full
    internal state must match the CPU oracle after every step."""
    params = layouts.get_base_layout_params(layout, horizon, max_num_players=cap)
    P, C = params["num_players"], params["height"] * params["width"]
    F = 5 * P + 16
    orc = oracle_lib.OvercookedOracle(params, n, num_threads=8)
    sim = make_sim(params, n)
    o = world_major(sim).view(n, P, C, F)
    a = sim.action_tensor().to_torch()
    rng = np.random.default_rng(zlib.crc32(f"{layout}-{n}".encode()))
    assert np.array_equal(o.cpu().numpy().astype(np.uint8), orc.obs)
    for t in range(steps):
        # 35% interact so pots fill, cook and get served within short horizons
        acts = rng.integers(0, 5, size=(P, n)).astype(np.int32)
        acts[rng.random((P, n)) < 0.35] = 5
        orc.step(acts)
        a.copy_(torch.from_numpy(acts).cuda().view(P, n, 1))
        sim.step()
        assert np.array_equal(o.cpu().numpy().astype(np.uint8), orc.obs), f"obs differ at step {t}"
        assert np.array_equal(sim.reward_tensor().to_torch().cpu().numpy(), orc.reward), f"reward, step {t}"
        assert np.array_equal(sim.done_tensor().to_torch().cpu().numpy(), orc.done), f"done, step {t}"
        if t % 10 == 0 or t == steps - 1:
            pl, ob, ts = orc.dump()
            assert np.array_equal(unpack_players(sim.state_players_tensor().to_torch()), pl), f"players, step {t}"
            assert np.array_equal(sim.state_objects_tensor().to_torch().cpu().numpy(), ob), f"objects, step {t}"
            assert np.array_equal(sim.state_timestep_tensor().to_torch().cpu().numpy(), ts), f"timestep, step {t}"
    sim.close()


@pytest.mark.parametrize("layout,n,wpw", [("cramped_room", 4099, 8), ("asymmetric_advantages", 1000, 4), ("coordination_ring", 513, 4),
                                          ("counter_circuit", 1200, 4)])
def test_generic_kernel_equals_specialised(layout, n, wpw, hip_lib):
    """The five standard layouts run kernels specialised at compile time for their size; `overcooked.no_fixed` sends
    the same simulator through the generic kernel.  Same actions (int32 array, int64 tensor, device-side rollout,
    action sequence): same tensors."""
    from madrona_rl_envs_playground_amd._lib import debug_knobs
    params = layouts.get_base_layout_params(layout, 60)
    P = params["num_players"]
    with debug_knobs({"overcooked.wpw": wpw}):  # the group size the library picks for these layouts from 32768 worlds on
        fixed = make_sim(params, n)
    with debug_knobs({"overcooked.wpw": wpw, "overcooked.no_fixed": 1}):
        generic = make_sim(params, n)
    assert "_fixed<" in fixed.kernel_name and generic.kernel_name == "mrl_overcooked_step<false, 2>"
    gen = torch.Generator(device="cuda").manual_seed(3)
    for t in range(80):
        if t % 4 == 0:
            a = torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
            fixed.step_with_actions(a)
            generic.step_with_actions(a)
        elif t % 4 == 1:
            a = torch.randint(0, 6, (P, n, 1), dtype=torch.int64, device="cuda", generator=gen)
            fixed.step_with_actions_i64(a)
            generic.step_with_actions_i64(a)
        elif t % 4 == 2:
            fixed.rollout_random(3, seed=9, first_step=3 * t)
            generic.rollout_random(3, seed=9, first_step=3 * t)
        else:
            seq = torch.randint(0, 6, (4, P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
            fixed.step_sequence(seq)
            generic.step_sequence(seq)
        for get in ("observation_world_major_tensor", "reward_tensor", "done_tensor", "action_tensor", "state_objects_tensor",
                    "state_players_tensor", "state_timestep_tensor"):
            assert torch.equal(getattr(fixed, get)().to_torch(), getattr(generic, get)().to_torch()), f"{get}, step {t}"
    fixed.close()
    generic.close()


@pytest.mark.parametrize("layout,n,knobs", [("coordination_ring", 4099, {"overcooked.wpw": 4}), ("asymmetric_advantages", 16391, {"overcooked.wpw": 4}),
                                            ("coordination_ring", 515, {"overcooked.no_fixed": 1}), ("multiplayer_schelling", 130, {})])
def test_plain_stream_out_equals_write_through(layout, n, knobs, hip_lib):
    """The single-pass stream-out uses write-through stores, or ordinary ones where the slab is larger than the Infinity
    Cache and a group's slab is not whole 128-byte lines (chosen at construction; `overcooked.whole_store` forces either).
    Both instantiations of every kernel family (specialised, two groups per wave, generic, multi-step): same tensors."""
    from madrona_rl_envs_playground_amd._lib import debug_knobs
    params = layouts.get_base_layout_params(layout, 45)
    P = params["num_players"]
    with debug_knobs({**knobs, "overcooked.whole_store": 1}):
        through = make_sim(params, n)
    with debug_knobs({**knobs, "overcooked.whole_store": 2}):
        plain = make_sim(params, n)
    gen = torch.Generator(device="cuda").manual_seed(29)
    for t in range(90):
        if t % 3 == 0:
            a = torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
            through.step_with_actions(a)
            plain.step_with_actions(a)
        elif t % 3 == 1:
            through.rollout_random(3, seed=5, first_step=3 * t)
            plain.rollout_random(3, seed=5, first_step=3 * t)
        else:
            seq = torch.randint(0, 6, (4, P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
            through.step_sequence(seq)
            plain.step_sequence(seq)
        for get in ("observation_world_major_tensor", "reward_tensor", "done_tensor", "state_objects_tensor", "state_players_tensor",
                    "state_timestep_tensor"):
            assert torch.equal(getattr(through, get)().to_torch(), getattr(plain, get)().to_torch()), f"{get}, step {t}"
    through.close()
    plain.close()


@pytest.mark.parametrize("players,n,horizon", [(2, 50, 30), (3, 21, 22), (4, 35, 26), (8, 33, 30), (13, 17, 25), (30, 9, 20)])
def test_shared_world_one_state_copy_equals_private_copies(players, n, horizon, hip_lib, oracle_lib):
    """Few worlds of a large layout: the four waves of a workgroup share a world.  By default the workgroup keeps ONE copy of
    the world's state in LDS (wave 0 steps it, all threads compute the cells' tails: `mrl_overcooked_step_team`);
    `overcooked.share_private` = 1 gives every wave its own copy and a redundant transition.  Both equal the oracle."""
    from madrona_rl_envs_playground_amd._lib import debug_knobs
    params = layouts.get_base_layout_params("many_player_layout", horizon, max_num_players=players)
    P, C = params["num_players"], params["height"] * params["width"]
    team = make_sim(params, n)
    with debug_knobs({"overcooked.share_private": 1}):
        private = make_sim(params, n)
    assert "step_team" in team.kernel_name and "step_team" not in private.kernel_name
    orc = oracle_lib.OvercookedOracle(params, n, num_threads=8)
    rng = np.random.default_rng(players)
    assert np.array_equal(world_major(team).view(n, P, C, 5 * P + 16).cpu().numpy().astype(np.uint8), orc.obs)
    for t in range(3 * horizon + 5):
        acts = rng.integers(0, 5, size=(P, n)).astype(np.int32)
        acts[rng.random((P, n)) < 0.35] = 5
        orc.step(acts)
        a = torch.from_numpy(acts).cuda().view(P, n, 1)
        if t % 2 == 0:
            team.step_with_actions(a)
            private.step_with_actions(a)
        else:
            team.step_with_actions_i64(a.to(torch.int64))
            private.step_with_actions_i64(a.to(torch.int64))
        assert np.array_equal(world_major(team).view(n, P, C, 5 * P + 16).cpu().numpy().astype(np.uint8), orc.obs), f"obs differ at step {t}"
        for get in ("observation_world_major_tensor", "reward_tensor", "done_tensor", "action_tensor", "state_objects_tensor", "state_players_tensor",
                    "state_timestep_tensor"):
            assert torch.equal(getattr(team, get)().to_torch(), getattr(private, get)().to_torch()), f"{get}, step {t}"
    team.close()
    private.close()


def test_steps_captured_in_a_hip_graph_equal_eager_steps(hip_lib):
    """The C ABI only enqueues on the caller's stream (no host synchronisation, no allocation): step calls can be
    captured with torch.cuda.graph like any other stream work and replayed (bench.py's `graph_replay` leg).
    A replayed graph of 12 steps leaves the same tensors as the same 12 calls issued one by one."""
    params = layouts.get_base_layout_params("cramped_room", 37)
    n, P = 33000, 2
    eager, captured = make_sim(params, n), make_sim(params, n)
    gen = torch.Generator(device="cuda").manual_seed(23)
    acts = [torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen) for _ in range(12)]
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for a in acts:
                captured.step_with_actions(a)
    for rep in range(4):  # 48 steps: across a horizon reset
        graph.replay()
        for a in acts:
            eager.step_with_actions(a)
        torch.cuda.synchronize()
        for get in ("observation_world_major_tensor", "reward_tensor", "done_tensor", "state_objects_tensor", "state_players_tensor",
                    "state_timestep_tensor"):
            assert torch.equal(getattr(eager, get)().to_torch(), getattr(captured, get)().to_torch()), f"{get}, replay {rep}"
    eager.close()
    captured.close()


@pytest.mark.parametrize("layout,cap,n", [("cramped_room", None, 4099), ("counter_circuit", None, 1001), ("asymmetric_advantages", None, 515),
                                          ("multiplayer_schelling", None, 130), ("cramped_room", None, 7),
                                          ("many_player_layout", 2, 1003)])  # one 13260-byte world per wave: slabs off 16-byte boundaries
def test_direct_patch_equals_searched(layout, cap, n, hip_lib):
    """The single-pass encode takes its dynamic rows from the player lanes and a table of holder cells (counters and pots a
    player can face); `overcooked.no_direct` makes it look for them through the cell -> player map instead.  Same tensors,
    through the generic kernels, the multi-step launches included."""
    from madrona_rl_envs_playground_amd._lib import debug_knobs
    params = layouts.get_base_layout_params(layout, 45, max_num_players=cap)
    P = params["num_players"]
    with debug_knobs({"overcooked.no_fixed": 1}):
        direct = make_sim(params, n)
    with debug_knobs({"overcooked.no_fixed": 1, "overcooked.no_direct": 1}):
        searched = make_sim(params, n)
    gen = torch.Generator(device="cuda").manual_seed(11)
    for t in range(120):
        if t % 3 == 0:
            a = torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
            a[torch.rand((P, n, 1), device="cuda", generator=gen) < 0.3] = 5
            direct.step_with_actions(a)
            searched.step_with_actions(a)
        elif t % 3 == 1:
            direct.rollout_random(3, seed=5, first_step=3 * t)
            searched.rollout_random(3, seed=5, first_step=3 * t)
        else:
            seq = torch.randint(0, 6, (4, P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
            direct.step_sequence(seq)
            searched.step_sequence(seq)
        for get in ("observation_world_major_tensor", "reward_tensor", "done_tensor", "state_objects_tensor", "state_players_tensor",
                    "state_timestep_tensor"):
            assert torch.equal(getattr(direct, get)().to_torch(), getattr(searched, get)().to_torch()), f"{get}, step {t}"
    direct.close()
    searched.close()


@pytest.mark.parametrize("layout,n,wpw", [("cramped_room", 4099, 8), ("cramped_room", 40003, 0), ("asymmetric_advantages", 16391, 0),
                                          ("counter_circuit", 1001, 4), ("coordination_ring", 24581, 0)])
def test_two_groups_per_wave_equals_one(layout, n, wpw, hip_lib):
    """Mid-sized batches of the standard layouts step two consecutive groups per wave (the second group's state is
    requested with the first's, the tile is reused): `overcooked.groups` forces either shape.  Batches that end inside
    a wave's first group, int32 and int64 actions: same tensors as one wave per group."""
    from madrona_rl_envs_playground_amd._lib import debug_knobs
    params = layouts.get_base_layout_params(layout, 60)
    P = params["num_players"]
    with debug_knobs({"overcooked.wpw": wpw, "overcooked.groups": 1}):
        one = make_sim(params, n)
    with debug_knobs({"overcooked.wpw": wpw, "overcooked.groups": 2}):
        two = make_sim(params, n)
    assert "_fixed<" in one.kernel_name and "groups" not in one.kernel_name and "step_groups_fixed<" in two.kernel_name
    assert two.launch_shape[0] < one.launch_shape[0]
    gen = torch.Generator(device="cuda").manual_seed(3)
    for t in range(130):
        if t % 2 == 0:
            a = torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
            one.step_with_actions(a)
            two.step_with_actions(a)
        else:
            a = torch.randint(0, 6, (P, n, 1), dtype=torch.int64, device="cuda", generator=gen)
            one.step_with_actions_i64(a)
            two.step_with_actions_i64(a)
        for get in ("observation_world_major_tensor", "reward_tensor", "done_tensor", "action_tensor", "state_objects_tensor",
                    "state_players_tensor", "state_timestep_tensor"):
            assert torch.equal(getattr(one, get)().to_torch(), getattr(two, get)().to_torch()), f"{get}, step {t}"
    one.close()
    two.close()


@pytest.mark.parametrize("policy", [1, 2, 3])
def test_multi_pass_store_policies(policy, hip_lib, oracle_lib):
    """Large layouts stream their rows out with write-through stores when the slab fits the Infinity Cache and with
    plain stores above that (chosen by size at construction); the knob forces each flavour on a small batch."""
    from madrona_rl_envs_playground_amd._lib import debug_knobs
    params = layouts.get_base_layout_params("many_player_layout", 40, max_num_players=12)
    n, P, C = 37, 12, params["height"] * params["width"]
    orc = oracle_lib.OvercookedOracle(params, n, num_threads=8)
    with debug_knobs({"overcooked.store_policy": policy}):
        sim = make_sim(params, n)
    o = world_major(sim).view(n, P, C, 5 * P + 16)
    rng = np.random.default_rng(policy)
    for t in range(60):
        acts = rng.integers(0, 6, size=(P, n)).astype(np.int32)
        orc.step(acts)
        sim.step_with_actions(torch.from_numpy(acts).cuda().view(P, n, 1))
        assert np.array_equal(o.cpu().numpy().astype(np.uint8), orc.obs), f"obs differ at step {t}"
        assert np.array_equal(sim.reward_tensor().to_torch().cpu().numpy(), orc.reward), f"reward, step {t}"
    sim.close()


@pytest.mark.parametrize("layout,cap,n,variant", [("cramped_room", None, 4099, 1), ("counter_circuit", None, 300, 1),
                                                   ("multiplayer_schelling", None, 257, 0), ("many_player_layout", 16, 40, 0)])
def test_same_cell_interactions_in_player_order(layout, cap, n, variant, hip_lib, oracle_lib):
    """The transition runs one lane per (world, player); players facing the same counter or pot must still
    act in ascending id (the reference's rank phases, sim.cpp:259-358).  70% INTERACT and orientation
    changes only now and then, so several players stay turned to one cell for many steps.  variant 1: the
    two-player layouts through the any-player-count code path (LDS exchange instead of DPP)."""
    from madrona_rl_envs_playground_amd._lib import debug_knobs
    params = layouts.get_base_layout_params(layout, 60, max_num_players=cap)
    P, C = params["num_players"], params["height"] * params["width"]
    F = 5 * P + 16
    orc = oracle_lib.OvercookedOracle(params, n, num_threads=8)
    with debug_knobs({"overcooked.variant": variant}):
        sim = make_sim(params, n)
    assert "step_team" in sim.kernel_name or sim.kernel_name.endswith("0>") == (variant == 1 or P != 2)  # (few worlds of a large layout: team kernel)
    o = world_major(sim).view(n, P, C, F)
    rng = np.random.default_rng(99)
    for t in range(150):
        acts = rng.integers(0, 5, size=(P, n)).astype(np.int32)
        acts[rng.random((P, n)) < 0.7] = 5
        orc.step(acts)
        sim.step_with_actions(torch.from_numpy(acts).cuda().view(P, n, 1))
        assert np.array_equal(o.cpu().numpy().astype(np.uint8), orc.obs), f"obs differ at step {t}"
        assert np.array_equal(sim.reward_tensor().to_torch().cpu().numpy(), orc.reward), f"reward, step {t}"
        if t % 10 == 0:
            pl, ob, ts = orc.dump()
            assert np.array_equal(unpack_players(sim.state_players_tensor().to_torch()), pl), f"players, step {t}"
            assert np.array_equal(sim.state_objects_tensor().to_torch().cpu().numpy(), ob), f"objects, step {t}"
    sim.close()


def test_reference_view_layout(hip_lib):
    """The (P*C, N, F) view + id tensors behave like the reference's exports:
    the reference wrapper's scatter (envs/overcooked_env.py:94-96) must give the
    same (P, N, W, H, F) tensor as the world-major fast path."""
    params = layouts.get_base_layout_params("coordination_ring", 400)
    P, H, W = params["num_players"], params["height"], params["width"]
    F, n = 5 * P + 16, 37
    sim = make_sim(params, n)
    a = sim.action_tensor().to_torch()
    for _ in range(20):
        a.copy_(torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda"))
        sim.step()
    static_obs = sim.observation_tensor().to_torch()
    assert static_obs.shape == (P * H * W, n, F) and static_obs.dtype == torch.int8
    loc_id = sim.location_id_tensor().to_torch().long()
    loc_world = sim.location_world_id_tensor().to_torch().long()
    scattered = torch.empty((P * H * W, n, F), dtype=torch.int8, device="cuda")
    scattered[loc_id, loc_world, :] = static_obs[:, :, :F]
    ref_style = scattered.reshape(P, H, W, n, F).transpose(1, 3)            # (P, N, W, H, F)
    fast = world_major(sim).permute(1, 0, 3, 2, 4)                           # (N,P,H,W,F) -> (P,N,W,H,F)
    assert torch.equal(ref_style, fast)
    wid = sim.world_id_tensor().to_torch()
    aid = sim.agent_id_tensor().to_torch()
    assert torch.equal(wid, torch.arange(n, device="cuda", dtype=torch.int32).expand(P, n))
    assert torch.equal(aid, torch.arange(P, device="cuda", dtype=torch.int32)[:, None].expand(P, n))
    assert (sim.active_agent_tensor().to_torch() == 1).all() and (sim.action_mask_tensor().to_torch() == 1).all()
    assert sim.action_mask_tensor().to_torch().shape == (P, n, 6)
    sim.close()


def test_step_with_actions_matches_in_place(hip_lib):
    params = layouts.get_base_layout_params("cramped_room", 30)
    n = 200
    s1, s2 = make_sim(params, n), make_sim(params, n)
    for _ in range(70):
        acts = torch.randint(0, 6, (2, n, 1), dtype=torch.int32, device="cuda")
        s1.action_tensor().to_torch().copy_(acts)
        s1.step()
        s2.step_with_actions(acts)
        assert torch.equal(world_major(s1), world_major(s2))
        assert torch.equal(s1.reward_tensor().to_torch(), s2.reward_tensor().to_torch())
    s1.close()
    s2.close()


def test_rejects_bad_configs(hip_lib):
    params = layouts.get_base_layout_params("cramped_room", 400)
    bad = dict(params)
    bad["terrain"] = list(params["terrain"])
    bad["terrain"][0] = 0  # walkable cell on the border
    with pytest.raises(RuntimeError):
        make_sim(bad, 4)
    with pytest.raises(NotImplementedError):
        OvercookedSimulator(exec_mode=ExecMode.CPU, gpu_id=0, num_worlds=4, **params)


@pytest.mark.parametrize("layout,horizon,cap,n,chunks", [
    ("cramped_room", 23, None, 4099, [1, 7, 30, 2, 25]),   # crosses several horizon resets, ragged last group
    ("cramped_room", 400, None, 40000, [16]),              # 8 worlds per wave
    ("multiplayer_schelling", 31, 3, 130, [9, 40]),        # odd player count: generic transition
    ("many_player_layout", 15, None, 9, [5, 20]),          # no single-pass encode: draw kernel + ordinary step
])
def test_device_random_rollout(layout, horizon, cap, n, chunks, hip_lib, oracle_lib):
    """mrl_rollout_random == the oracle fed the documented action stream: obs, reward,
    done, last actions and the full state after every chunk of fused steps."""
    from madrona_rl_envs_playground_amd.simulators import random_action
    params = layouts.get_base_layout_params(layout, horizon, max_num_players=cap)
    P, C = params["num_players"], params["height"] * params["width"]
    F = 5 * P + 16
    seed = 0x1234_5678_9ABC_DEF1
    orc = oracle_lib.OvercookedOracle(params, n, num_threads=8)
    sim = make_sim(params, n)
    o = world_major(sim).view(n, P, C, F)
    world, player = np.meshgrid(np.arange(n), np.arange(P))
    k = 1000  # first_step offsets the stream
    for chunk in chunks:
        sim.rollout_random(chunk, seed=seed, first_step=k)
        for s in range(chunk):
            acts = random_action(seed, k + s, world, player)
            assert acts.min() >= 0 and acts.max() <= 5
            orc.step(acts)
        k += chunk
        assert np.array_equal(sim.action_tensor().to_torch().cpu().numpy()[..., 0], acts)
        assert np.array_equal(o.cpu().numpy().astype(np.uint8), orc.obs), f"obs differ after step {k}"
        assert np.array_equal(sim.reward_tensor().to_torch().cpu().numpy(), orc.reward)
        assert np.array_equal(sim.done_tensor().to_torch().cpu().numpy(), orc.done)
        pl, ob, ts = orc.dump()
        assert np.array_equal(unpack_players(sim.state_players_tensor().to_torch()), pl)
        assert np.array_equal(sim.state_objects_tensor().to_torch().cpu().numpy(), ob)
        assert np.array_equal(sim.state_timestep_tensor().to_torch().cpu().numpy(), ts)
    # an ordinary step continues from the rollout's state
    acts = random_action(seed, k, world, player)
    sim.action_tensor().to_torch().copy_(torch.from_numpy(acts).cuda().view(P, n, 1))
    sim.step()
    orc.step(acts)
    assert np.array_equal(o.cpu().numpy().astype(np.uint8), orc.obs)
    sim.close()


@pytest.mark.parametrize("layout,cap,n,steps", [("cramped_room", None, 5000, 90), ("coordination_ring", None, 777, 70),
                                                 ("many_player_layout", 8, 33, 25)])
def test_step_sequence_equals_single_steps(layout, cap, n, steps, hip_lib):
    """mrl_step_sequence (one launch for the whole action array where the layout allows) leaves the
    tensors K calls of step_with_actions leave, with the final step's observation, reward and done."""
    params = layouts.get_base_layout_params(layout, 31, max_num_players=cap)
    P = params["num_players"]
    a, b = make_sim(params, n), make_sim(params, n)
    gen = torch.Generator(device="cuda").manual_seed(8)
    acts = torch.randint(0, 6, (steps, P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
    done_any = 0
    for lo, hi in ((0, 1), (1, steps // 2), (steps // 2, steps)):
        a.step_sequence(acts[lo:hi].contiguous())
        for k in range(lo, hi):
            b.step_with_actions(acts[k])
        for name in ("observation_world_major_tensor", "reward_tensor", "done_tensor", "state_players_tensor",
                     "state_objects_tensor", "state_timestep_tensor"):
            assert torch.equal(getattr(a, name)().to_torch(), getattr(b, name)().to_torch()), f"{name} after step {hi}"
        done_any += int(b.done_tensor().to_torch().sum().item())
    a.close()
    b.close()


def _random_layout(rng):
    """A random rectangular kitchen in the reference's Config format (sim.hpp:44-57): non-AIR border, sparse
    interior walls, a few pots / sources / serving cells on border or interior wall cells, 1..6 players on AIR."""
    H, W = int(rng.integers(3, 10)), int(rng.integers(3, 12))
    while H * W > 255:
        W -= 1
    terr = np.full((H, W), 2, np.int64)                     # COUNTER
    terr[1:-1, 1:-1] = 0                                    # AIR
    interior = [(y, x) for y in range(1, H - 1) for x in range(1, W - 1)]
    for (y, x) in interior:
        if rng.random() < 0.12:
            terr[y, x] = 2
    walls = [(y, x) for y in range(H) for x in range(W) if terr[y, x] == 2]
    rng.shuffle(walls)
    kinds = [1] * int(rng.integers(1, 7)) + [3] * int(rng.integers(1, 3)) + [4] * int(rng.integers(0, 2)) + \
            [5] * int(rng.integers(1, 3)) + [6] * int(rng.integers(1, 3))   # POT, ONION, TOMATO, DISH, SERVING
    for k, (y, x) in zip(kinds, walls):
        terr[y, x] = k
    air = [(y, x) for y in range(H) for x in range(W) if terr[y, x] == 0]
    if not air:
        terr[1, 1] = 0
        air = [(1, 1)]
    P = int(min(len(air), rng.integers(1, 7)))
    rng.shuffle(air)
    starts = air[:P]
    times = [int(rng.integers(1, 25)) for _ in range(16)]
    values = [int(rng.integers(0, 40)) if rng.random() < 0.5 else 0 for _ in range(16)]
    return dict(height=H, width=W, terrain=[int(v) for v in terr.reshape(-1)], num_players=P,
                start_player_x=[x for (_, x) in starts], start_player_y=[y for (y, _) in starts],
                placement_in_pot_rew=int(rng.integers(0, 6)), dish_pickup_rew=0, soup_pickup_rew=int(rng.integers(0, 8)),
                recipe_times=times, recipe_values=values, horizon=int(rng.integers(15, 70)))


@pytest.mark.parametrize("seed", range(24))
def test_random_layouts_against_oracle(seed, hip_lib, oracle_lib):
    """Kitchens nobody tuned a launch shape for: random sizes, player counts (odd ones too), pot counts beyond the
    kernel-argument table, recipe tables and horizons; a few world counts so that every group size gets picked."""
    rng = np.random.default_rng(1000 + seed)
    params = _random_layout(rng)
    n = int(rng.choice([1, 7, 64, 300, 1111, 5000, 40000 if params["num_players"] <= 2 else 9000]))
    P, C = params["num_players"], params["height"] * params["width"]
    F = 5 * P + 16
    orc = oracle_lib.OvercookedOracle(params, n, num_threads=8)
    sim = make_sim(params, n)
    o = world_major(sim).view(n, P, C, F)
    assert np.array_equal(o.cpu().numpy().astype(np.uint8), orc.obs), f"initial obs, {params} x {n}"
    steps = 40 if n <= 5000 else 14
    for t in range(steps):
        acts = rng.integers(0, 5, size=(P, n)).astype(np.int32)
        acts[rng.random((P, n)) < 0.4] = 5
        orc.step(acts)
        sim.step_with_actions(torch.from_numpy(acts).cuda().view(P, n, 1))
        assert np.array_equal(o.cpu().numpy().astype(np.uint8), orc.obs), f"obs differ at step {t}: {params} x {n} ({sim.kernel_name}, {sim.launch_shape})"
        assert np.array_equal(sim.reward_tensor().to_torch().cpu().numpy(), orc.reward), f"reward, step {t}"
        assert np.array_equal(sim.done_tensor().to_torch().cpu().numpy(), orc.done), f"done, step {t}"
    pl, ob, ts = orc.dump()
    assert np.array_equal(unpack_players(sim.state_players_tensor().to_torch()), pl)
    assert np.array_equal(sim.state_objects_tensor().to_torch().cpu().numpy(), ob)
    assert np.array_equal(sim.state_timestep_tensor().to_torch().cpu().numpy(), ts)
    sim.close()


@pytest.mark.parametrize("layout,n", [("cramped_room", 4099), ("coordination_ring", 1001), ("counter_circuit", 33000),
                                      ("forced_coordination", 33001)])  # 1001 / 33001 worlds x 1300 bytes: slots 1..4 of the dense ring start off 16-byte boundaries (staged)
def test_steps_into_a_ring_of_caller_slots_equal_steps_plus_clones(layout, n, hip_lib, oracle_lib):
    """mrl_set_observation_output (the rollout-buffer side): K steps whose observations the
    kernel writes into slot k % T of a caller's ring equal K ordinary steps each followed by a clone (what the reference's
    trainer does, train/MAPPO/main_player.py:245-247), and equal the oracle; the simulator's own tensor is untouched
    meanwhile and takes the output back afterwards; the multi-step launches follow the redirect as well."""
    params = layouts.get_base_layout_params(layout, 25)
    P = params["num_players"]
    into, plain = make_sim(params, n), make_sim(params, n)
    orc = oracle_lib.OvercookedOracle(params, n, num_threads=8)
    own = into.observation_world_major_tensor().to_torch()
    before = own.clone()
    T, K = 5, 23
    ring = torch.zeros((T,) + tuple(own.shape), dtype=torch.int8, device="cuda")
    clones = []
    gen = torch.Generator(device="cuda").manual_seed(n)
    for k in range(K):
        a = torch.randint(0, 8, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen).clamp_(max=5)
        into.set_observation_output(ring[k % T])
        into.step_with_actions(a)
        plain.step_with_actions(a)
        clones.append(plain.observation_world_major_tensor().to_torch().clone())
        orc.step(a[:, :, 0].cpu().numpy())
        assert np.array_equal(ring[k % T].cpu().numpy().astype(np.uint8).reshape(orc.obs.shape), orc.obs), f"slot differs from the oracle at step {k}"
        assert torch.equal(into.reward_tensor().to_torch(), plain.reward_tensor().to_torch())
        assert torch.equal(into.state_objects_tensor().to_torch(), plain.state_objects_tensor().to_torch())
    for k in range(K - T, K):
        assert torch.equal(ring[k % T], clones[k]), f"slot {k % T} != clone of step {k}"
    assert torch.equal(own, before)  # nothing was written to the simulator's own tensor
    # multi-step launches follow the redirect: the slot holds the last step's observations
    into.set_observation_output(ring[0])
    into.rollout_random(7, seed=5, first_step=0)
    plain.rollout_random(7, seed=5, first_step=0)
    assert torch.equal(ring[0], plain.observation_world_major_tensor().to_torch()) and torch.equal(own, before)
    # ... and NULL hands the output back
    into.set_observation_output(None)
    a = torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
    into.step_with_actions(a)
    plain.step_with_actions(a)
    assert torch.equal(own, plain.observation_world_major_tensor().to_torch())
    # validation: wrong size / dtype / device are refused
    with pytest.raises(Exception):
        into.set_observation_output(ring[0].flatten()[:-16])
    with pytest.raises(ValueError):
        into.set_observation_output(ring[0].to(torch.int32))
    with pytest.raises(ValueError):
        into.set_observation_output(ring[0].cpu())
    # a slot at ANY byte offset is taken (off 16-byte boundaries it is staged: a slab of the simulator's + one device copy)
    for off in (1, 3, 8, 13):
        odd = torch.zeros(own.numel() + 16, dtype=torch.int8, device="cuda")
        slot = odd[off:off + own.numel()].view(own.shape)
        assert slot.data_ptr() % 16 == (odd.data_ptr() + off) % 16 != 0
        into.set_observation_output(slot)
        a = torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
        into.step_with_actions(a)
        plain.step_with_actions(a)
        assert torch.equal(slot, plain.observation_world_major_tensor().to_torch()), f"slot at offset {off}"
        assert not odd[:off].any() and not odd[off + own.numel():].any(), "bytes around the slot were written"
    into.close()
    plain.close()


@pytest.mark.parametrize("layout,n,dense", [("cramped_room", 4100, False), ("counter_circuit", 1000, False), ("multiplayer_schelling", 128, False),
                                            ("coordination_ring", 33000, False), ("coordination_ring", 1001, True), ("asymmetric_advantages", 4099, True)])
def test_multi_step_launches_fill_a_ring_of_slots(layout, n, dense, hip_lib):
    """mrl_set_observation_ring: step k from the call on writes its observations to slot k % T of the caller's buffer -- inside
    ONE multi-step launch (mrl_rollout_random, mrl_step_sequence: the kernel moves from slot to slot itself), across launches,
    and for single steps.  Every slot must hold exactly what a twin simulator, stepped one launch at a time, shows after that step."""
    params = layouts.get_base_layout_params(layout, 30)
    P = params["num_players"]
    ringed, twin = make_sim(params, n), make_sim(params, n)
    own = ringed.observation_world_major_tensor().to_torch()
    before = own.clone()
    T = 6
    # padded: slots start on 16-byte boundaries whatever N, written in place; dense: a plain (T, N, P, H, W, F) buffer, whose
    # slots then start wherever N x block bytes puts them (staged: one launch + one device copy per step)
    slot_bytes = own.numel() if dense else (own.numel() + 15) // 16 * 16
    assert not dense or slot_bytes % 16 != 0
    store = torch.zeros(T * slot_bytes, dtype=torch.int8, device="cuda")
    ring = torch.as_strided(store, (T,) + tuple(own.shape), (slot_bytes,) + tuple(own.stride()))
    ringed.set_observation_ring(ring)
    history = []  # observations after every step, from the twin

    def twin_random(steps, seed, first):
        for k in range(steps):
            twin.rollout_random(1, seed=seed, first_step=first + k)
            history.append(twin.observation_world_major_tensor().to_torch().clone())

    def check(tag):
        k_total = len(history)
        for k in range(max(0, k_total - T), k_total):
            assert torch.equal(ring[k % T], history[k]), f"{tag}: slot {k % T} is not the observation after step {k}"
        assert torch.equal(own, before), f"{tag}: the simulator's own tensor was written"

    ringed.rollout_random(4, seed=11, first_step=0)          # slots 0..3
    twin_random(4, 11, 0)
    check("first launch")
    ringed.rollout_random(9, seed=11, first_step=4)          # wraps around the ring one and a half times
    twin_random(9, 11, 4)
    check("wrapping launch")
    gen = torch.Generator(device="cuda").manual_seed(n)
    seq = torch.randint(0, 6, (5, P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
    ringed.step_sequence(seq)                                # an open-loop action sequence, one launch
    for k in range(5):
        twin.step_with_actions(seq[k])
        history.append(twin.observation_world_major_tensor().to_torch().clone())
    check("action sequence")
    for k in range(4):                                       # single steps take their slots from the same count
        a = torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
        ringed.step_with_actions(a)
        twin.step_with_actions(a)
        history.append(twin.observation_world_major_tensor().to_torch().clone())
    check("single steps")
    assert torch.equal(ringed.state_objects_tensor().to_torch(), twin.state_objects_tensor().to_torch())
    assert torch.equal(ringed.reward_tensor().to_torch(), twin.reward_tensor().to_torch())
    ringed.set_observation_ring(None)
    ringed.rollout_random(3, seed=2, first_step=0)
    twin.rollout_random(3, seed=2, first_step=0)
    assert torch.equal(own, twin.observation_world_major_tensor().to_torch())
    with pytest.raises(Exception, match="stride"):
        ringed.set_observation_ring(torch.zeros((2, own.numel() - 16), dtype=torch.int8, device="cuda"))
    ringed.close()
    twin.close()
