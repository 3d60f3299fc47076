"""GPU parity of the Simplecooked (overcooked2_env) HIP step (through the C ABI) against
(a) the golden vectors produced by the reference's numpy twin of that world and
(b) the CPU oracle on seeded action streams -- bit-exact, including the internal state."""
import glob
import json
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

from madrona_rl_envs_playground_amd import layouts  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import ExecMode, SimplecookedSimulator  # noqa: E402


def make_sim(params, n):
    return SimplecookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)


def unpack_players(t):
    t = t.cpu().numpy()
    return np.stack([t[..., 0], t[..., 1], t[..., 4], t[..., 5], t[..., 6], t[..., 7]], axis=-1)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "simplecooked_*.npz"))),
                         ids=lambda p: os.path.basename(p)[13:-4])
def test_golden_vectors(path, hip_lib):
    """Reference numpy observations / rewards / dones; the one byte per tomato-source cell where the
    reference's C++ and numpy differ (`differs`) must be 0 as in the C++ (sim.cpp:74)."""
    z = np.load(path)
    params = json.loads(str(z["params"]))
    acts, obs, rew, done, keep = z["actions"], z["obs"], z["reward"], z["done"], ~z["differs"]
    P, C, F = params["num_players"], params["height"] * params["width"], 5 * params["num_players"] + 10
    n = 5
    sim = make_sim(params, n)
    o = sim.observation_world_major_tensor().to_torch().view(n, P, C, F)
    a = sim.action_tensor().to_torch()
    r, d = sim.reward_tensor().to_torch(), sim.done_tensor().to_torch()
    got = o.cpu().numpy().astype(np.uint8)
    assert np.array_equal(got[:, :, keep], np.broadcast_to(obs[0], (n, P, C, F))[:, :, keep]) and (got[:, :, ~keep] == 0).all()
    for t in range(len(acts)):
        a.copy_(torch.from_numpy(acts[t].astype(np.int32)).cuda()[:, None, None].expand(P, n, 1))
        sim.step()
        got = o.cpu().numpy().astype(np.uint8)
        assert np.array_equal(got[:, :, keep], np.broadcast_to(obs[t + 1], (n, P, C, F))[:, :, keep]), f"obs differ at step {t}"
        assert (got[:, :, ~keep] == 0).all()
        assert (r.cpu().numpy() == rew[t]).all(), f"reward differs at step {t}"
        assert (d.cpu().numpy() == done[t]).all(), f"done differs at step {t}"
    sim.close()


@pytest.mark.parametrize("layout,horizon,cap,n,steps,p_interact", [
    ("simple", 200, None, 1000, 450, 0.35),
    ("simple", 37, None, 4099, 120, 0.5),
    ("unident_s", 60, None, 513, 150, 0.35),
    ("random1", 50, None, 777, 150, 0.45),
    ("random0", 50, None, 256, 120, 0.35),
    ("random3", 50, None, 300, 120, 0.6),
    ("simple_tomato", 80, None, 640, 200, 0.45),
    ("simple", 30, 1, 77, 100, 0.4),                 # a single player: byte-wise row tails, no dish shaping
    ("unident_s", 45, 1, 130, 100, 0.4),
    ("simple", 200, None, 40000, 12, 0.35),          # 8 worlds per wave, ragged last group
])
def test_against_oracle(layout, horizon, cap, n, steps, p_interact, hip_lib, oracle_lib):
    params = layouts.get_simplecooked_layout_params(layout, horizon, max_num_players=cap)
    P, C = params["num_players"], params["height"] * params["width"]
    F = 5 * P + 10
    orc = oracle_lib.SimplecookedOracle(params, n, num_threads=8)
    sim = make_sim(params, n)
    o = sim.observation_world_major_tensor().to_torch().view(n, P, C, F)
    rng = np.random.default_rng(zlib.crc32(f"{layout}-{n}".encode()))
    assert np.array_equal(o.cpu().numpy().astype(np.uint8), orc.obs)
    rewards = 0
    for t in range(steps):
        acts = rng.integers(0, 5, size=(P, n)).astype(np.int32)
        acts[rng.random((P, n)) < p_interact] = 5
        orc.step(acts)
        sim.step_with_actions(torch.from_numpy(acts).cuda().view(P, n, 1))
        assert np.array_equal(o.cpu().numpy().astype(np.uint8), orc.obs), f"obs differ at step {t}"
        assert np.array_equal(sim.reward_tensor().to_torch().cpu().numpy(), orc.reward), f"reward, step {t}"
        assert np.array_equal(sim.done_tensor().to_torch().cpu().numpy(), orc.done), f"done, step {t}"
        rewards += int(orc.reward[0].sum())
        if t % 10 == 0 or t == steps - 1:
            pl, ob, ts, dishes = orc.dump()
            assert np.array_equal(unpack_players(sim.state_players_tensor().to_torch()), pl), f"players, step {t}"
            assert np.array_equal(sim.state_objects_tensor().to_torch().cpu().numpy(), ob), f"objects, step {t}"
            assert np.array_equal(sim.state_timestep_tensor().to_torch().cpu().numpy(), ts), f"timestep, step {t}"
            assert np.array_equal(sim.dishes_out_tensor().to_torch().cpu().numpy(), dishes), f"dishes out, step {t}"
    assert rewards > 0 or steps < 50
    sim.close()


@pytest.mark.parametrize("layout,n,wpw", [("simple", 4099, 8), ("random3", 1200, 4)])
def test_generic_kernel_equals_specialised(layout, n, wpw, hip_lib):
    """The standard layout sizes run kernels specialised at compile time; `overcooked.no_fixed` sends the same
    simulator through the generic kernel.  Same actions (int32 array, int64 tensor, device-side draw): same tensors."""
    from madrona_rl_envs_playground_amd._lib import debug_knobs
    params = layouts.get_simplecooked_layout_params(layout, 60)
    P = params["num_players"]
    with debug_knobs({"overcooked.wpw": wpw}):  # the group size the library picks for these layouts on large batches
        fixed = make_sim(params, n)
    with debug_knobs({"overcooked.wpw": wpw, "overcooked.no_fixed": 1}):
        generic = make_sim(params, n)
    assert "step_fixed<" in fixed.kernel_name and "step_fixed<" not in generic.kernel_name
    gen = torch.Generator(device="cuda").manual_seed(3)
    for t in range(90):
        if t % 3 == 0:
            a = torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
            fixed.step_with_actions(a)
            generic.step_with_actions(a)
        elif t % 3 == 1:
            a = torch.randint(0, 6, (P, n, 1), dtype=torch.int64, device="cuda", generator=gen)
            fixed.step_with_actions_i64(a)
            generic.step_with_actions_i64(a)
        elif t % 6 == 2:
            fixed.rollout_random(5, seed=9, first_step=5 * t)   # all five steps in one launch, tile reused
            generic.rollout_random(5, seed=9, first_step=5 * t)
        else:
            seq = torch.randint(0, 6, (4, P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
            fixed.step_sequence(seq)
            generic.step_sequence(seq)
        for get in ("observation_world_major_tensor", "reward_tensor", "done_tensor", "action_tensor", "state_objects_tensor",
                    "state_players_tensor", "state_timestep_tensor"):
            assert torch.equal(getattr(fixed, get)().to_torch(), getattr(generic, get)().to_torch()), f"{get}, step {t}"
    fixed.close()
    generic.close()


@pytest.mark.parametrize("layout,n", [("simple", 4099), ("random0", 1030), ("unident_s", 7)])
def test_direct_patch_equals_searched(layout, n, hip_lib):
    """The single step takes its dynamic rows from the player lanes and a table of holder cells; `overcooked.no_direct`
    makes it look for them through the cell -> player map.  Same tensors (generic kernels, interact-heavy actions)."""
    from madrona_rl_envs_playground_amd._lib import debug_knobs
    params = layouts.get_simplecooked_layout_params(layout, 37)
    P = params["num_players"]
    with debug_knobs({"overcooked.no_fixed": 1}):
        direct = make_sim(params, n)
    with debug_knobs({"overcooked.no_fixed": 1, "overcooked.no_direct": 1}):
        searched = make_sim(params, n)
    gen = torch.Generator(device="cuda").manual_seed(5)
    for t in range(120):
        a = torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
        a[torch.rand((P, n, 1), device="cuda", generator=gen) < 0.3] = 5
        direct.step_with_actions(a)
        searched.step_with_actions(a)
        for get in ("observation_world_major_tensor", "reward_tensor", "done_tensor", "state_objects_tensor", "state_players_tensor",
                    "state_timestep_tensor"):
            assert torch.equal(getattr(direct, get)().to_torch(), getattr(searched, get)().to_torch()), f"{get}, step {t}"
    direct.close()
    searched.close()


@pytest.mark.parametrize("layout,n,knobs", [("random0", 4099, {"overcooked.wpw": 8}), ("unident_s", 515, {"overcooked.no_fixed": 1})])
def test_plain_stream_out_equals_write_through(layout, n, knobs, hip_lib):
    """Write-through or ordinary stores for the stream-out (chosen at construction by group alignment, slab size and kind of
    launch; `overcooked.whole_store` forces either): both instantiations of the specialised, generic and multi-step kernels
    leave the same tensors."""
    from madrona_rl_envs_playground_amd._lib import debug_knobs
    params = layouts.get_simplecooked_layout_params(layout, 45)
    P = params["num_players"]
    with debug_knobs({**knobs, "overcooked.whole_store": 1}):
        through = make_sim(params, n)
    with debug_knobs({**knobs, "overcooked.whole_store": 2}):
        plain = make_sim(params, n)
    gen = torch.Generator(device="cuda").manual_seed(31)
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


def test_device_random_policy_and_sequence(hip_lib, oracle_lib):
    """mrl_rollout_random draws the documented stream (all steps of a call in one launch, state and tile resident);
    mrl_step_sequence runs an action array the same way; both equal the oracle and one launch per step."""
    from madrona_rl_envs_playground_amd.simulators import random_action
    params = layouts.get_simplecooked_layout_params("random1", 23)
    n, P = 2051, 2
    seed = 0xABCDEF0123
    orc = oracle_lib.SimplecookedOracle(params, n, num_threads=8)
    sim, twin, single = make_sim(params, n), make_sim(params, n), make_sim(params, n)  # one launch / one launch / one launch per step
    o = sim.observation_world_major_tensor().to_torch().view(orc.obs.shape)
    world, player = np.meshgrid(np.arange(n), np.arange(P))
    k = 100
    for chunk in (1, 6, 30, 2):
        sim.rollout_random(chunk, seed=seed, first_step=k)
        seq = []
        for s in range(chunk):
            acts = random_action(seed, k + s, world, player)
            orc.step(acts)
            seq.append(torch.from_numpy(acts).view(P, n, 1))
        twin.step_sequence(torch.stack(seq).cuda().contiguous())
        for a in seq:
            single.step_with_actions(a.cuda().contiguous())
        k += chunk
        assert np.array_equal(sim.action_tensor().to_torch().cpu().numpy()[..., 0], acts)
        assert np.array_equal(o.cpu().numpy().astype(np.uint8), orc.obs), f"obs differ after step {k}"
        assert np.array_equal(sim.reward_tensor().to_torch().cpu().numpy(), orc.reward)
        assert np.array_equal(sim.done_tensor().to_torch().cpu().numpy(), orc.done)
        for name in ("observation_world_major_tensor", "reward_tensor", "done_tensor", "state_players_tensor", "state_objects_tensor",
                     "state_timestep_tensor", "dishes_out_tensor"):
            assert torch.equal(getattr(sim, name)().to_torch(), getattr(twin, name)().to_torch()), name
            assert torch.equal(getattr(sim, name)().to_torch(), getattr(single, name)().to_torch()), name
    sim.close()
    twin.close()
    single.close()


def test_env_wrapper_and_reference_views(hip_lib):
    """envs/overcooked2_env.OvercookedMadrona: the reference's attribute names, shapes and action mask; the
    (P*C, N, F) strided export re-indexed through the id tensors like the reference's get_obs
    (envs/overcooked2_env.py:91-101) equals the world-major fast path."""
    from madrona_rl_envs_playground_amd.envs.overcooked2_env import OvercookedMadrona
    from madrona_rl_envs_playground_amd.pantheonrl_extension import RandomVectorAgent
    n = 96
    env = OvercookedMadrona("simple", n, 0, horizon=25)
    P, H, W, F = 2, env.height, env.width, 20
    assert env.observation_space.shape == (W, H, F) and env.action_space.n == 6 and env.obs_size == W * H * F
    assert env.static_actions.shape == (P, n, 1) and env.static_observations.shape == (P * H * W, n, F)
    env.add_partner_agent(RandomVectorAgent(lambda: torch.randint(0, 6, (n, 1), device=env.device)))
    ob = env.reset()
    assert ob.obs.shape == (n, W, H, F) and ob.obs.dtype == torch.int8 and ob.action_mask.shape == (n, 6) and ob.action_mask.all()
    finished = 0
    for _ in range(60):
        ob, rew, done, _ = env.step(torch.randint(0, 6, (n, 1), device=env.device))
        assert rew.shape == (n,) and done.shape == (n,)
        finished += int(done.sum())
    assert finished == 2 * n
    sim = env.sim
    loc_id, loc_world = sim.location_id_tensor().to_torch().long(), sim.location_world_id_tensor().to_torch().long()
    scattered = torch.empty((P * H * W, n, F), dtype=torch.int8, device="cuda")
    scattered[loc_id, loc_world, :] = env.static_observations[:, :, :F]
    ref_style = scattered.reshape(P, H, W, n, F).transpose(1, 3)            # (P, N, W, H, F)
    assert torch.equal(ref_style[0], ob.obs) and torch.equal(ref_style, env.static_world_major_observations.permute(1, 0, 3, 2, 4))
    env.close()


def test_rejects_what_the_reference_cannot_hold(hip_lib):
    big = layouts.get_base_layout_params("many_player_layout", 100, max_num_players=2)  # 15 x 17 > MAX_SIZE
    with pytest.raises(RuntimeError, match="100"):
        make_sim(big, 4)
    four = layouts.get_base_layout_params("multiplayer_schelling", 100)
    with pytest.raises(RuntimeError, match="1..2"):
        make_sim(four, 4)


def test_full_shard_properties(hip_lib, oracle_lib):
    """32768 worlds of `simple` (what the reference's trainer runs, at its per-GPU benchmark size)."""
    horizon, steps = 90, 110
    params = layouts.get_simplecooked_layout_params("simple", horizon)
    n, P, C, F = 32768, 2, 20, 20
    sim = make_sim(params, n)
    assert sim.launch_shape[2] <= 40960  # four workgroups per CU
    obs = sim.observation_world_major_tensor().to_torch().view(n, P, C, F)
    sample = 64
    orc = oracle_lib.SimplecookedOracle(params, 2 * sample, num_threads=4)
    torch.manual_seed(2)
    total = 0
    for t in range(steps):
        a = torch.randint(0, 8, (P, n, 1), dtype=torch.int32, device="cuda").clamp_(max=5)
        sim.step_with_actions(a)
        orc.step(torch.cat([a[:, :sample, 0], a[:, n - sample:, 0]], dim=1).cpu().numpy())
        rew, done = sim.reward_tensor().to_torch(), sim.done_tensor().to_torch()
        assert torch.equal(rew[0], rew[1]) and (rew >= 0).all()
        total += int(rew[0].sum())
        assert bool(done.all()) == (t == horizon - 1) and bool(done.any()) == (t == horizon - 1)
        if t % 20 == 0:
            assert (obs[:, :, :, 0:P].sum(dim=2) == 1).all() and (obs[:, :, :, P:5 * P].sum(dim=(2, 3)) == P).all()
            assert torch.equal(obs[:, 0, :, 5 * P:], obs[:, 1, :, 5 * P:])
            assert (sim.dishes_out_tensor().to_torch() >= 0).all()
        got = torch.cat([obs[:sample], obs[n - sample:]]).cpu().numpy().astype(np.uint8)
        assert np.array_equal(got, orc.obs), f"sampled worlds differ from the oracle at step {t}"
    assert total > 0
    sim.close()


def _random_kitchen(rng):
    """A random old-style kitchen in the reference's Config format (src/overcooked2_env/sim.hpp:40-53): at most 100
    cells and two players; terrain codes AIR 0, POT 1, COUNTER 2, ONION 3, (TOMATO 4 is wiped by the C++), DISH 4 -> see layouts."""
    ref = layouts.get_simplecooked_layout_params("simple", 50)
    codes = {"pot": 1, "counter": 2, "onion": 3, "dish": ref["terrain"][16], "serve": ref["terrain"][18]}
    H, W = int(rng.integers(3, 10)), int(rng.integers(3, 11))
    while H * W > 100:
        W -= 1
    terr = np.full((H, W), codes["counter"], np.int64)
    terr[1:-1, 1:-1] = 0
    for y in range(1, H - 1):
        for x in range(1, W - 1):
            if rng.random() < 0.1:
                terr[y, x] = codes["counter"]
    walls = [(y, x) for y in range(H) for x in range(W) if terr[y, x] == codes["counter"]]
    rng.shuffle(walls)
    kinds = ["pot"] * int(rng.integers(1, 5)) + ["onion"] * int(rng.integers(1, 3)) + ["dish"] * int(rng.integers(1, 3)) + \
            ["serve"] * int(rng.integers(1, 3))
    for k, (y, x) in zip(kinds, walls):
        terr[y, x] = codes[k]
    air = [(y, x) for y in range(H) for x in range(W) if terr[y, x] == 0]
    if not air:
        terr[1, 1] = 0
        air = [(1, 1)]
    P = int(min(len(air), rng.integers(1, 3)))
    rng.shuffle(air)
    starts = air[:P]
    return dict(ref, height=H, width=W, terrain=[int(v) for v in terr.reshape(-1)], num_players=P,
                start_player_x=[x for (_, x) in starts], start_player_y=[y for (y, _) in starts],
                placement_in_pot_rew=int(rng.integers(0, 6)), dish_pickup_rew=int(rng.integers(0, 6)),
                soup_pickup_rew=int(rng.integers(0, 8)), horizon=int(rng.integers(15, 70)))


@pytest.mark.parametrize("seed", range(16))
def test_random_kitchens_against_oracle(seed, hip_lib, oracle_lib):
    """Kitchens nobody tuned a launch shape for: random sizes, one or two players, several pots, reward shaping values
    and horizons; a few world counts so that every group size gets picked."""
    rng = np.random.default_rng(2000 + seed)
    params = _random_kitchen(rng)
    n = int(rng.choice([1, 7, 64, 300, 1111, 5000, 40000]))
    P, C = params["num_players"], params["height"] * params["width"]
    F = 5 * P + 10
    orc = oracle_lib.SimplecookedOracle(params, n, num_threads=8)
    sim = make_sim(params, n)
    o = sim.observation_world_major_tensor().to_torch().view(n, P, C, F)
    assert np.array_equal(o.cpu().numpy().astype(np.uint8), orc.obs), f"initial obs, {params} x {n}"
    steps = 50 if n <= 5000 else 14
    for t in range(steps):
        acts = rng.integers(0, 5, size=(P, n)).astype(np.int32)
        acts[rng.random((P, n)) < 0.45] = 5
        orc.step(acts)
        sim.step_with_actions(torch.from_numpy(acts).cuda().view(P, n, 1))
        assert np.array_equal(o.cpu().numpy().astype(np.uint8), orc.obs), f"obs differ at step {t}: {params} x {n} ({sim.kernel_name}, {sim.launch_shape})"
        assert np.array_equal(sim.reward_tensor().to_torch().cpu().numpy(), orc.reward), f"reward, step {t}"
        assert np.array_equal(sim.done_tensor().to_torch().cpu().numpy(), orc.done), f"done, step {t}"
    pl, ob, ts, dishes = orc.dump()
    assert np.array_equal(unpack_players(sim.state_players_tensor().to_torch()), pl)
    assert np.array_equal(sim.state_objects_tensor().to_torch().cpu().numpy(), ob)
    assert np.array_equal(sim.state_timestep_tensor().to_torch().cpu().numpy(), ts)
    assert np.array_equal(sim.dishes_out_tensor().to_torch().cpu().numpy(), dishes)
    sim.close()


def test_env_steps_into_a_callers_slot(hip_lib, oracle_lib):
    """`env.n_step(actions, out=slot)` (mrl_set_observation_output through the drop-in wrapper): the step kernel writes the
    observations into the caller's (N, P, H, W, F) slot, the observations handed out are views of it and equal the oracle;
    without `out` the next step goes back to the simulator's own tensor."""
    from madrona_rl_envs_playground_amd.envs.overcooked2_env import OvercookedMadrona
    n = 3001  # N x 1800 bytes is not a multiple of 16: slots 1 and 2 of the dense ring are staged (a slab of the simulator's + one copy)
    env = OvercookedMadrona("unident_s", n, 0, horizon=30)
    params = layouts.get_simplecooked_layout_params("unident_s", 30)
    orc = oracle_lib.SimplecookedOracle(params, n, num_threads=8)
    P, H, W = 2, env.height, env.width
    ring = torch.zeros((3,) + tuple(env.static_world_major_observations.shape), dtype=torch.int8, device="cuda")
    own = env.static_world_major_observations.clone()
    gen = torch.Generator(device="cuda").manual_seed(9)
    for t in range(40):
        a = torch.randint(0, 6, (P, n, 1), device="cuda", generator=gen)  # int64, the harness's dtype
        slot = ring[t % 3]
        obs, rew, done, _ = env.n_step(a, out=slot)
        orc.step(a[:, :, 0].to(torch.int32).cpu().numpy())
        assert obs[1].obs.data_ptr() == slot[:, 1].data_ptr() and obs[0].obs.shape == (n, W, H, 20) and obs[0].action_mask.all()
        got = slot.cpu().numpy().astype(np.uint8).reshape(orc.obs.shape)
        assert np.array_equal(got, orc.obs), f"step {t}"
        assert np.array_equal(rew.cpu().numpy(), orc.reward) and np.array_equal(done.cpu().numpy(), orc.done)
        assert torch.equal(env.static_world_major_observations, own)
    with pytest.raises(ValueError):
        env.n_step(a, out=ring[0].reshape(-1))
    obs, _, _, _ = env.n_step(a)  # back to the simulator's own tensor
    orc.step(a[:, :, 0].to(torch.int32).cpu().numpy())
    assert obs[0].obs.data_ptr() == env.static_world_major_observations.data_ptr()
    assert np.array_equal(env.static_world_major_observations.cpu().numpy().astype(np.uint8).reshape(orc.obs.shape), orc.obs)
    env.close()


def test_rollout_fills_a_ring_of_slots(hip_lib):
    """mrl_set_observation_ring for Simplecooked: one mrl_rollout_random launch of K steps leaves step k's observations in slot
    k % T; single steps continue the count."""
    params = layouts.get_simplecooked_layout_params("simple", 40)
    n, T = 4096, 4
    ringed, twin = make_sim(params, n), make_sim(params, n)
    own = ringed.observation_world_major_tensor().to_torch()
    before = own.clone()
    ring = torch.zeros((T,) + tuple(own.shape), dtype=torch.int8, device="cuda")
    ringed.set_observation_ring(ring)
    ringed.rollout_random(10, seed=3, first_step=0)
    history = []
    for k in range(10):
        twin.rollout_random(1, seed=3, first_step=k)
        history.append(twin.observation_world_major_tensor().to_torch().clone())
    a = torch.randint(0, 6, (2, n, 1), dtype=torch.int32, device="cuda")
    ringed.step_with_actions(a)
    twin.step_with_actions(a)
    history.append(twin.observation_world_major_tensor().to_torch().clone())
    for k in range(len(history) - T, len(history)):
        assert torch.equal(ring[k % T], history[k]), f"slot {k % T} is not the observation after step {k}"
    assert torch.equal(own, before)
    ringed.close()
    twin.close()
