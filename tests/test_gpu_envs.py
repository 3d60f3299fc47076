"""GPU: the drop-in env classes (reference names and return shapes) over the HIP
simulators, plus golden-fixture replays for Hanabi and Cartpole."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

from madrona_rl_envs_playground_amd import hanabi_spec, layouts  # noqa: E402
from madrona_rl_envs_playground_amd.distributed import ShardedSimulator  # noqa: E402
from madrona_rl_envs_playground_amd.envs import (CartpoleMadronaNumpy, CartpoleMadronaTorch, FULL_CONFIG,  # noqa: E402
                                                 HanabiMadrona, OvercookedMadrona)
from madrona_rl_envs_playground_amd.pantheonrl_extension import RandomVectorAgent  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import (CartpoleSimulator, ExecMode, HanabiSimulator,  # noqa: E402
                                                       OvercookedSimulator)


def test_overcooked_env_matches_reference_numpy_fixture(hip_lib):
    """OvercookedMadrona.n_step returns, per player, an (N, W, H, F) int8 view with
    the values the reference wrapper would return (its obs is the DummyMDP row
    block reshaped (H, W, F) and transposed to (W, H, F), envs/overcooked_env.py:404)."""
    z = np.load(os.path.join(GOLDEN, "overcooked_cramped_room_cook.npz"))
    params = json.loads(str(z["params"]))
    n, P, H, W, F = 6, 2, params["height"], params["width"], 26
    env = OvercookedMadrona("cramped_room", n, 0, horizon=params["horizon"])
    assert env.observation_space.shape == (W, H, F) and env.action_space.n == 6
    assert env.static_actions.shape == (P, n, 1) and env.static_observations.shape == (P * H * W, n, F)
    obs = env.n_reset()
    expect0 = z["obs"][0].reshape(P, H, W, F).transpose(0, 2, 1, 3)
    for p in range(P):
        assert obs[p].obs.shape == (n, W, H, F) and obs[p].obs.dtype == torch.int8 and obs[p].active.all()
        assert np.array_equal(obs[p].obs[3].cpu().numpy().astype(np.uint8), expect0[p])
    for t in range(300):
        a = torch.from_numpy(z["actions"][t].astype(np.int64))[:, None, None].expand(P, n, 1)  # int64, CPU: as the harness
        obs, rew, done, info = env.n_step(a)
        want = z["obs"][t + 1].reshape(P, H, W, F).transpose(0, 2, 1, 3)
        for p in range(P):
            assert np.array_equal(obs[p].obs[n - 1].cpu().numpy().astype(np.uint8), want[p]), f"step {t}"
        assert rew.shape == (P, n) and (rew.cpu().numpy() == z["reward"][t]).all()
        assert done.shape == (n,) and (done.cpu().numpy() == z["done"][t]).all() and len(info) == n
    env.close()


@pytest.mark.parametrize("which", ["overcooked", "simplecooked"])
def test_wrapped_step_reads_int64_device_actions_directly(which, hip_lib, oracle_lib):
    """The reference's harness hands `env.n_step` an int64 tensor on the device (scripts/overcooked_example.py:99-106).
    The wrappers pass it straight to the step kernel (mrl_step_with_actions_i64): same results as the oracle, and
    `static_actions` mirrors the actions like the reference's copy would."""
    n, horizon = 3000, 45
    if which == "overcooked":
        env = OvercookedMadrona("coordination_ring", n, 0, horizon=horizon)
        orc = oracle_lib.OvercookedOracle(env.base_layout_params, n, num_threads=4)
    else:
        from madrona_rl_envs_playground_amd.envs.overcooked2_env import OvercookedMadrona as Simplecooked
        env = Simplecooked("random1", n, 0, horizon=horizon)
        orc = oracle_lib.SimplecookedOracle(env.base_layout_params, n, num_threads=4)
    P, H, W = 2, env.height, env.width
    F = orc.F
    gen = torch.Generator(device="cuda").manual_seed(4)
    for t in range(100):
        a = torch.randint(0, 6, (P, n, 1), device="cuda", generator=gen)   # int64, like randint_like of a long tensor
        assert a.dtype == torch.int64
        obs, rew, done, _ = env.n_step(a)
        orc.step(a[..., 0].cpu().numpy())
        want = orc.obs.reshape(n, P, H, W, F).transpose(0, 1, 3, 2, 4)
        for p in range(P):
            assert np.array_equal(obs[p].obs.cpu().numpy().astype(np.uint8), want[:, p]), f"step {t}"
        assert np.array_equal(rew.cpu().numpy(), orc.reward) and np.array_equal(done.cpu().numpy(), orc.done)
        assert torch.equal(env.static_actions, a.to(torch.int32))
    env.close()


def test_overcooked_ego_step_with_random_partner(hip_lib):
    n = 64
    env = OvercookedMadrona("coordination_ring", n, 0, horizon=30)
    env.add_partner_agent(RandomVectorAgent(lambda: torch.randint(0, 6, (n, 1), device=env.device)))
    ob = env.reset()
    assert ob.obs.shape == (n, 5, 5, 26)
    finished = 0
    for _ in range(70):
        ob, rew, done, _ = env.step(torch.randint(0, 6, (n, 1), device=env.device))
        assert rew.shape == (n,) and done.shape == (n,)
        finished += int(done.sum())
    assert finished == 2 * n  # horizon 30 -> every world finished twice in 70 steps
    env.close()


@pytest.mark.parametrize("name", ["full", "small", "very_small"])
def test_hanabi_checked_fixture_on_gpu(name, hip_lib):
    """The sequences accepted by the reference's checker (make_hanabi_golden.py)."""
    z = np.load(os.path.join(GOLDEN, f"hanabi_{name}.npz"))
    cfg = dict(FULL_CONFIG) if name == "full" else (
        dict(colors=2, ranks=5, players=2, max_information_tokens=3, max_life_tokens=1) if name == "small"
        else dict(colors=1, ranks=5, players=2, max_information_tokens=3, max_life_tokens=1))
    n = z["actions"].shape[2]
    no, ns = hanabi_spec.observation_size(cfg), hanabi_spec.state_size(cfg)
    sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=cfg["colors"], ranks=cfg["ranks"],
                          players=2, max_information_tokens=cfg["max_information_tokens"],
                          max_life_tokens=cfg["max_life_tokens"])
    get = lambda t: t.to_torch().cpu().numpy()
    assert np.array_equal(get(sim.observation_tensor()).astype(np.uint8)[..., :no], z["first_obs"][..., :no])
    assert np.array_equal(get(sim.agent_state_tensor()).astype(np.uint8)[..., :ns], z["first_state"][..., :ns])
    for t in range(z["actions"].shape[0]):
        sim.action_tensor().to_torch().copy_(torch.from_numpy(z["actions"][t].astype(np.int32)).cuda().view(2, n, 1))
        sim.step()
        assert np.array_equal(get(sim.observation_tensor()).astype(np.uint8)[..., :no], z["obs"][t][..., :no]), t
        assert np.array_equal(get(sim.agent_state_tensor()).astype(np.uint8)[..., :ns], z["state"][t][..., :ns]), t
        assert np.array_equal(get(sim.action_mask_tensor()), z["mask"][t]), t
        assert np.array_equal(get(sim.active_agent_tensor()), z["active"][t]), t
        assert np.array_equal(get(sim.reward_tensor()), z["reward"][t]), t
        assert np.array_equal(get(sim.done_tensor()), z["done"][t]), t
    sim.close()


def test_hanabi_env_wrapper(hip_lib):
    n = 128
    env = HanabiMadrona(n, 0, config=FULL_CONFIG)
    assert env.observation_space.shape == (658,) and env.share_observation_space.shape == (783,)
    assert env.action_space.n == 20 and env.n_players == 2
    obs = env.n_reset()
    assert obs[0].obs.shape == (n, 658) and obs[0].state.shape == (n, 783) and obs[0].action_mask.shape == (n, 20)
    assert obs[0].active.all() and not obs[1].active.any() and obs[0].action_mask.dtype == torch.bool
    for _ in range(40):
        masks = torch.stack([o.action_mask for o in obs])
        acts = (torch.rand(masks.shape, device=masks.device) * masks).argmax(-1, keepdim=True)  # harness sampling
        obs, rew, done, _ = env.n_step(acts)
        assert (obs[0].active ^ obs[1].active).all() and rew.shape == (2, n) and done.shape == (n,)
        assert rew.dtype == torch.float32
    env.close()


def test_cartpole_golden_transitions_on_gpu(hip_lib):
    """One step from the reference-generated start states: within the reference's own bound of 1e-6 against its float64
    numpy twin (envs/cartpole_env.py:277; the north star asks 1e-5)."""
    z = np.load(os.path.join(GOLDEN, "cartpole_transitions.npz"))
    states, actions, next64, done = z["states"], z["actions"], z["next64"], z["done"]
    m = len(states)
    sim = CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=m)
    st = sim.observation_tensor().to_torch()
    st.copy_(torch.from_numpy(states).cuda())
    sim.action_tensor().to_torch().copy_(torch.from_numpy(actions).cuda().view(m, 1))
    sim.step_phase1(None)  # dynamics + termination only, so terminal worlds keep their last state
    got, got_done = st.cpu().numpy().astype(np.float64), sim.reset_tensor().to_torch().cpu().numpy()[:, 0]
    near = (np.abs(np.abs(next64[:, 0]) - 2.4) < 1e-5) | (np.abs(np.abs(next64[:, 2]) - 12 * 2 * np.pi / 360) < 1e-5)
    assert ((got_done == done) | near).all()
    assert np.abs(got - next64).max() < 1e-6
    sim.close()


def test_cartpole_env_wrappers(hip_lib):
    n = 256
    env = CartpoleMadronaTorch(n, 0)
    assert env.action_space.n == 2 and env.observation_space.shape == (4,)
    ob = env.reset()
    assert ob.shape == (n, 4) and ob.abs().max() <= 0.05
    for _ in range(30):
        ob, rew, done, info = env.step(torch.randint(0, 2, (n,), device=ob.device, dtype=torch.int32))
        assert ob.shape == (n, 4) and rew.shape == (n, 1) and done.shape == (n,) and len(info) == n
    env.close()
    env = CartpoleMadronaNumpy(n, 0)
    ob, rew, done, _ = env.step(np.ones(n, np.int32))
    assert isinstance(ob, np.ndarray) and ob.shape == (n, 4) and done.shape == (n,)
    env.close()


def test_sharded_wrapper_single_rank_equals_plain(hip_lib):
    """ShardedSimulator with one rank (no process group) == the plain simulator."""
    n = 3000
    plain = CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n)
    sh = ShardedSimulator(lambda k: CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=k), n)
    torch.manual_seed(3)
    for _ in range(100):
        a = torch.randint(0, 2, (n, 1), dtype=torch.int32, device="cuda")
        plain.step_with_actions(a)
        sh.step(a)
        assert torch.equal(plain.observation_tensor().to_torch(), sh.sim.observation_tensor().to_torch())
    ov = ShardedSimulator(lambda k: OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=k,
                                                        **layouts.get_base_layout_params("cramped_room", 400)),
                          512, needs_episode_exchange=False)
    ov.step(torch.randint(0, 6, (2, 512, 1), dtype=torch.int32, device="cuda"))
    local = ov.sim.observation_world_major_tensor().to_torch()
    assert ov.gather(local, 0) is local
    plain.close()


def test_reference_style_scatter_on_hanabi_exports(hip_lib):
    """What the reference's generic wrapper does with a simulator's exports
    (pantheonrl_extension/vectorenv.py:283-293,306-329: clone, then index_put
    through the agent/world id tensors, then slice) must work on this engine's
    strided views and give the same tensors as reading them directly."""
    n = 97
    sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=5, ranks=5, players=2,
                          max_information_tokens=8, max_life_tokens=3)
    static_obs = sim.observation_tensor().to_torch()
    static_state = sim.agent_state_tensor().to_torch()
    static_mask = sim.action_mask_tensor().to_torch()
    wid = sim.world_id_tensor().to_torch().to(torch.long)
    aid = sim.agent_id_tensor().to_torch().to(torch.long)
    scattered_obs = static_obs.detach().clone()
    scattered_state = static_state.detach().clone()
    assert scattered_obs.shape == (2, n, 658)  # clone() keeps the export's stride order; the wrapper never needs contiguity
    act = sim.action_tensor().to_torch()
    for _ in range(25):
        actions = (torch.rand(static_mask.shape, device="cuda") * static_mask).argmax(-1, keepdim=True)
        act.copy_(actions[aid, wid, :])          # the reference's action gather
        sim.step()
        scattered_obs[aid, wid, :] = static_obs
        scattered_state[aid, wid, :] = static_state
        assert torch.equal(scattered_obs, static_obs) and torch.equal(scattered_state, static_state)
        assert torch.equal(scattered_obs[1, :, :658].to("cpu"), static_obs[1].cpu())
    sim.close()
