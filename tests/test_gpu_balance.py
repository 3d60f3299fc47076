"""GPU parity of the balance-beam HIP step: the reference checker world's transitions (golden fixture), a
lock-step against the CPU oracle including resets and episode numbering, the device-side random policy,
and the env wrapper."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

from madrona_rl_envs_playground_amd.simulators import BalanceBeamSimulator, ExecMode, random_balance_action  # noqa: E402


def make(n):
    return BalanceBeamSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n)


def test_reference_transitions_on_gpu(hip_lib):
    z = np.load(os.path.join(GOLDEN, "balance_transitions.npz"))
    before, acts, after, rew, done = z["before"], z["actions"], z["after"], z["reward"], z["done"]
    n = len(before)
    sim = make(n)
    obs = sim.observation_tensor().to_torch()
    assert obs.shape == (2, n, 7) and obs.dtype == torch.int32
    obs.copy_(torch.from_numpy(np.ascontiguousarray(before.transpose(1, 0, 2))).cuda())   # the observation IS the state
    sim.step_with_actions(torch.from_numpy(np.ascontiguousarray(acts.T)).cuda().view(2, n, 1))
    assert np.array_equal(sim.done_tensor().to_torch().cpu().numpy(), done)
    r = sim.reward_tensor().to_torch().cpu().numpy()
    assert np.allclose(r[0], rew, rtol=0, atol=1e-7) and np.array_equal(r[0], r[1])
    alive = done == 0
    assert np.array_equal(obs.cpu().numpy()[:, alive].transpose(1, 0, 2), after[alive])
    assert int(sim.reset_count_tensor().to_torch().item()) == int(done.sum())
    sim.close()


@pytest.mark.parametrize("n", [1000, 70001])
def test_lockstep_vs_oracle(n, hip_lib, oracle_lib):
    sim, orc = make(n), oracle_lib.BalanceOracle(n)
    obs = sim.observation_tensor().to_torch()
    assert np.array_equal(obs.cpu().numpy(), orc.obs)
    assert sim.action_mask_tensor().to_torch().shape == (2, n, 4) and (sim.active_agent_tensor().to_torch() == 1).all()
    rng = np.random.default_rng(n)
    for t in range(40):
        acts = rng.integers(0, 4, size=(2, n)).astype(np.int32)
        orc.step(acts)
        sim.action_tensor().to_torch().copy_(torch.from_numpy(acts).cuda().view(2, n, 1))
        sim.step()
        assert np.array_equal(obs.cpu().numpy(), orc.obs), f"obs differ at step {t}"
        assert np.array_equal(sim.reward_tensor().to_torch().cpu().numpy(), orc.reward), f"reward, step {t}"
        assert np.array_equal(sim.done_tensor().to_torch().cpu().numpy(), orc.done), f"done, step {t}"
    assert orc.episodes > 10 * n
    sim.close()


def test_device_random_policy_and_sharding(hip_lib, oracle_lib):
    """mrl_rollout_random == the oracle fed the documented stream; two shards with the episode exchange == one simulator."""
    from madrona_rl_envs_playground_amd.distributed import ShardedSimulator
    n, seed = 5000, 0x5EED
    sim, orc = make(n), oracle_lib.BalanceOracle(n)
    world, player = np.meshgrid(np.arange(n), np.arange(2))
    k = 0
    for chunk in (1, 5, 20):
        sim.rollout_random(chunk, seed=seed, first_step=k)
        for s in range(chunk):
            acts = random_balance_action(seed, k + s, world, player)
            orc.step(acts)
        k += chunk
        assert np.array_equal(sim.action_tensor().to_torch().cpu().numpy()[..., 0], acts)
        assert np.array_equal(sim.observation_tensor().to_torch().cpu().numpy(), orc.obs)
    sim.close()
    whole = make(n)
    lo = make(2000)
    hi = make(3000)
    lo.reseed_shard(0, n)
    hi.reseed_shard(2000, n)
    counter = n
    gen = torch.Generator(device="cuda").manual_seed(3)
    for _ in range(15):
        a = torch.randint(0, 4, (2, n, 1), dtype=torch.int32, device="cuda", generator=gen)
        whole.step_with_actions(a)
        lo.step_phase1(a[:, :2000].contiguous())
        hi.step_phase1(a[:, 2000:].contiguous())
        c_lo = int(lo.reset_count_tensor().to_torch().item()) if False else int(lo.done_tensor().to_torch().sum())
        c_hi = int(hi.done_tensor().to_torch().sum())
        lo.step_phase2(torch.tensor([counter], dtype=torch.int32, device="cuda"))
        hi.step_phase2(torch.tensor([counter + c_lo], dtype=torch.int32, device="cuda"))
        counter += c_lo + c_hi
        both = torch.cat([lo.observation_tensor().to_torch(), hi.observation_tensor().to_torch()], dim=1)
        assert torch.equal(both, whole.observation_tensor().to_torch())
    for s in (whole, lo, hi):
        s.close()


def test_env_wrapper(hip_lib):
    from madrona_rl_envs_playground_amd.envs import BalanceMadronaTorch
    from madrona_rl_envs_playground_amd.pantheonrl_extension import RandomVectorAgent
    n = 128
    env = BalanceMadronaTorch(n, 0)
    assert env.action_space.n == 4 and tuple(env.observation_space.nvec) == (9,) * 6 + (3,)
    env.add_partner_agent(RandomVectorAgent(lambda: torch.randint(0, 4, (n, 1), device=env.device)))
    ob = env.reset()
    assert ob.obs.shape == (n, 7) and ob.action_mask.shape == (n, 4)
    ends = 0
    for _ in range(12):
        ob, rew, done, _ = env.step(torch.randint(0, 4, (n, 1), device=env.device))
        assert rew.shape == (n,) and done.shape == (n,) and (rew <= 1).all()
        ends += int(done.sum())
    assert ends >= 4 * n  # at most three steps per episode
    env.close()


def test_steps_captured_in_a_hip_graph_after_prepare(hip_lib):
    """A third of the worlds re-seed every step: a captured sequence of three steps replayed 30 times on a simulator prepared
    with mrl_prepare_graph_capture equals 90 steps issued one by one on an ordinary one."""
    from madrona_rl_envs_playground_amd.simulators import BalanceBeamSimulator, ExecMode
    n = 50000
    eager, graphed = BalanceBeamSimulator(ExecMode.CUDA, 0, n), BalanceBeamSimulator(ExecMode.CUDA, 0, n)
    graphed.prepare_graph_capture()
    gen = torch.Generator(device="cuda").manual_seed(4)
    acts = [torch.randint(0, 4, (2, n, 1), dtype=torch.int32, device="cuda", generator=gen) for _ in range(3)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for a in acts:
                graphed.step_with_actions(a)
    torch.cuda.current_stream().wait_stream(side)
    for rep in range(30):
        graph.replay()
        for a in acts:
            eager.step_with_actions(a)
        for name in ("observation_tensor", "done_tensor", "reward_tensor", "reset_count_tensor"):
            assert torch.equal(getattr(eager, name)().to_torch(), getattr(graphed, name)().to_torch()), f"{name} differs after replay {rep}"
    eager.close()
    graphed.close()


@pytest.mark.parametrize("n,heal", [(1000, 0), (300001, 0), (300001, 3), (70000, 1), (1 << 20, 0)],
                         ids=["1000", "300001", "300001_late_workgroups", "70000_all_late", "1M"])
def test_single_launch_step_equals_two_launches(n, heal, hip_lib):
    """mrl_step as ONE launch (every row written exactly once: a finished world's rows are replaced by its next episode's before
    they are stored; episode indices from the self-healing look-back) against the two-launch pair -- also with workgroups
    made to arrive late, so that higher ones recount them from their rows and actions."""
    from madrona_rl_envs_playground_amd._lib import debug_knobs
    with debug_knobs({"fused_step": 1, "fused_heal_test": heal}):
        one = make(n)
    with debug_knobs({"fused_step": 2}):
        two = make(n)
    assert one.kernel_name == "mrl_balance_step_fused" and two.kernel_name == "mrl_balance_step"
    gen = torch.Generator(device="cuda").manual_seed(n)
    for t in range(25):
        a = torch.randint(0, 4, (2, n, 1), dtype=torch.int32, device="cuda", generator=gen)
        one.step_with_actions(a)
        two.step_with_actions(a)
        for name in ("observation_tensor", "done_tensor", "reward_tensor", "reset_count_tensor"):
            assert torch.equal(getattr(one, name)().to_torch(), getattr(two, name)().to_torch()), f"{name} differs at step {t}"
    one.rollout_random(9, seed=4, first_step=0)
    two.rollout_random(9, seed=4, first_step=0)
    assert torch.equal(one.observation_tensor().to_torch(), two.observation_tensor().to_torch())
    assert torch.equal(one.action_tensor().to_torch(), two.action_tensor().to_torch())
    one.close()
    two.close()
