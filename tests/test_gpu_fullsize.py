"""GPU, BASELINE.json's full sizes.  Overcooked config 2: size-independent properties over all 32768 worlds plus a
lock-step against the oracle on a slice of the batch.  Hanabi config 3 (65536 worlds) and the 1 M-world Cartpole batch:
the WHOLE batch in lock-step against the multi-threaded C oracle (a few seconds of CPU per test), through the kernels
mrl_step and mrl_rollout_random actually run at those sizes -- the single-launch step, whose look-back across all 256
workgroups decides every episode number, and the persistent rollout -- and, beside them, the size-independent
properties."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from madrona_rl_envs_playground_amd import hanabi_spec, layouts  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import (CartpoleSimulator, ExecMode, HanabiSimulator,  # noqa: E402
                                                       OvercookedSimulator)


def test_overcooked_32768_worlds_properties(hip_lib, oracle_lib):
    """configs[1]: cramped_room, 32768 worlds, uniform random actions, seed 0."""
    params = layouts.get_base_layout_params("cramped_room", 400)
    n, P, H, W, F = 32768, 2, 4, 5, 26
    C = H * W
    sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
    assert sim.launch_shape == (1024, 256, sim.launch_shape[2], 8) and sim.launch_shape[2] <= 40960  # 4 workgroups per CU
    obs = sim.observation_world_major_tensor().to_torch().view(n, P, C, F)
    terrain = torch.tensor(params["terrain"], device="cuda")
    onehot = torch.zeros(C, 6, dtype=torch.int8, device="cuda")
    nz = terrain > 0
    onehot[nz, (terrain[nz] - 1)] = 1
    sample = 64  # worlds [0, sample) and the last `sample` worlds are also checked against the oracle
    orc = oracle_lib.OvercookedOracle(params, 2 * sample, num_threads=4)
    torch.manual_seed(0)
    total_reward = 0
    for t in range(420):  # crosses the horizon once
        a = torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda")
        sim.step_with_actions(a)
        sel = torch.cat([a[:, :sample, 0], a[:, n - sample:, 0]], dim=1).cpu().numpy()
        orc.step(sel)
        if t % 30 == 0 or t >= 398:
            o = obs
            # terrain channels are the static layout, for every world and viewer
            assert torch.equal(o[:, :, :, 10:16], onehot.expand(n, P, C, 6))
            # every viewer sees itself exactly once and every other player exactly once, on walkable cells
            assert (o[:, :, :, 0:2].sum(dim=2) == 1).all()
            assert (o[:, :, :, 2:10].sum(dim=(2, 3)) == 2).all()          # one orientation bit per player
            assert (o[:, :, :, 0:2].sum(dim=3) * (terrain != 0).to(torch.int8)).sum() == 0
            # viewer 1's view is viewer 0's with the two player blocks swapped
            assert torch.equal(o[:, 0, :, 0], o[:, 1, :, 1]) and torch.equal(o[:, 0, :, 2:6], o[:, 1, :, 6:10])
            assert torch.equal(o[:, 0, :, 10:], o[:, 1, :, 10:])          # the 16-byte tails are viewer-independent
            # urgency flag == (horizon - timestep < 40), uniform over the rows of a world
            ts = sim.state_timestep_tensor().to_torch()
            assert torch.equal(o[:, 0, 0, 25].to(torch.int32), ((400 - ts) < 40).to(torch.int32))
            assert (ts == (t + 1) % 400).all()
        rew = sim.reward_tensor().to_torch()
        assert torch.equal(rew[0], rew[1]) and (rew >= 0).all()
        total_reward += int(rew[0].sum())
        done = sim.done_tensor().to_torch()
        assert bool(done.all()) == (t == 399) and bool(done.any()) == (t == 399)
        got = torch.cat([obs[:sample], obs[n - sample:]]).cpu().numpy().astype(np.uint8)
        assert np.array_equal(got, orc.obs), f"sampled worlds differ from the oracle at step {t}"
    assert total_reward > 0
    sim.close()


def test_overcooked_shard_invariance(hip_lib):
    """Worlds are independent: one simulator of 2k worlds == two simulators of k worlds."""
    params = layouts.get_base_layout_params("asymmetric_advantages", 70)
    k = 4099
    whole = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=2 * k, **params)
    lo = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=k, **params)
    hi = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=k, **params)
    torch.manual_seed(4)
    for _ in range(150):
        a = torch.randint(0, 6, (2, 2 * k, 1), dtype=torch.int32, device="cuda")
        whole.step_with_actions(a)
        lo.step_with_actions(a[:, :k].contiguous())
        hi.step_with_actions(a[:, k:].contiguous())
        both = torch.cat([lo.observation_world_major_tensor().to_torch(), hi.observation_world_major_tensor().to_torch()])
        assert torch.equal(both, whole.observation_world_major_tensor().to_torch())
        assert torch.equal(torch.cat([lo.reward_tensor().to_torch(), hi.reward_tensor().to_torch()], dim=1),
                           whole.reward_tensor().to_torch())
    for s in (whole, lo, hi):
        s.close()


def test_hanabi_65536_worlds_properties(hip_lib):
    """configs[2]: full 2-player Hanabi, 65536 worlds, masked-random legal actions."""
    n = 65536
    sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=5, ranks=5, players=2,
                          max_information_tokens=8, max_life_tokens=3)
    obs, state = sim.observation_tensor().to_torch(), sim.agent_state_tensor().to_torch()
    mask, active = sim.action_mask_tensor().to_torch(), sim.active_agent_tensor().to_torch()
    act = sim.action_tensor().to_torch()
    torch.manual_seed(0)
    idx = torch.arange(n, device="cuda")
    finished = 0
    prev_active = active.clone()
    for t in range(60):
        act.copy_((torch.rand(mask.shape, device="cuda") * mask).argmax(-1, keepdim=True))
        sim.step()
        done = sim.done_tensor().to_torch()
        finished += int(done.sum())
        assert (active.sum(0) == 1).all()
        assert ((active[0] != prev_active[0]) | (done == 1)).all()           # the mover alternates unless the game ended
        assert (active[0][done == 1] == 1).all()
        cur = active[1].long()                                                # index of the active agent
        o, s = obs[cur, idx], state[cur, idx]
        info = o[:, 192:200].sum(1)
        ok = info <= 8                                                        # skip worlds in the token-overflow encoding
        assert torch.equal(s[ok][:, :658], o[ok])                             # state prefix == obs
        deck, life = o[:, 127:167].sum(1), o[:, 200:203].sum(1)
        ranks = torch.arange(1, 6, device="cuda")
        fireworks = (o[:, 167:192].view(n, 5, 5).long() * ranks).sum(dim=(1, 2))       # one-hot of the top rank per colour
        discards = o[:, 203:253].sum(1)
        hands = o[:, :125].sum(1) + s[:, 658:783].sum(1)
        assert ((deck + fireworks + discards + hands)[ok] == 50).all()        # card conservation
        assert (life[ok] >= 1).all()                                          # finished games were re-dealt
        m = mask[cur, idx]
        assert (m[ok][:, 5:10].sum(1) == s[ok][:, 658:783].sum(1)).all()      # one play move per card in hand
        assert ((m[:, 10:].sum(1) > 0) == (info > 0))[ok].all()               # hints need a token
        assert ((m[:, :5].sum(1) > 0) == (info < 8))[ok].all()                # discards need room for a token
        rew = sim.reward_tensor().to_torch()
        assert torch.equal(rew[0], rew[1])
        prev_active = active.clone()
    assert finished > n  # more than one finished game per world on average
    assert int(sim.reset_count_tensor().to_torch().item()) == int(sim.done_tensor().to_torch().sum())
    sim.close()


def test_cartpole_one_million_worlds(hip_lib):
    n = 1 << 20
    sim = CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n)
    st = sim.observation_tensor().to_torch()
    assert st.abs().max() <= 0.05
    torch.manual_seed(0)
    resets = 0
    for _ in range(40):
        sim.step_with_actions(torch.randint(0, 2, (n, 1), dtype=torch.int32, device="cuda"))
        done = sim.reset_tensor().to_torch()[:, 0]
        resets += int(done.sum())
        assert (st[done == 1].abs() <= 0.05).all()                            # re-seeded worlds
        assert (st[:, 0].abs() <= 2.4 + 0.2).all() and torch.isfinite(st).all()
        assert int(sim.reset_count_tensor().to_torch().item()) == int(done.sum())
    assert resets > n  # random policy: ~20-step episodes
    sim.close()


HANABI_FULL = dict(colors=5, ranks=5, players=2, max_information_tokens=8, max_life_tokens=3)


def _hanabi_equal(sim, orc, tag):
    """Every exported tensor and the 176-byte game record of every world against the oracle (compared on the device: the
    oracle's arrays go up, 190 MB per step, instead of the simulator's coming down)."""
    no, ns = hanabi_spec.observation_size(HANABI_FULL), hanabi_spec.state_size(HANABI_FULL)

    def same(name, got, want):
        want = torch.from_numpy(np.ascontiguousarray(want)).cuda()
        assert torch.equal(got.view(want.dtype) if got.dtype != want.dtype else got, want), f"{name} differs from the oracle, {tag}"

    same("observation", sim.observation_tensor().to_torch()[..., :no], orc.obs[..., :no].view(np.int8))
    same("state", sim.agent_state_tensor().to_torch()[..., :ns], orc.state[..., :ns].view(np.int8))
    same("legal moves", sim.action_mask_tensor().to_torch(), orc.mask)
    same("active", sim.active_agent_tensor().to_torch(), orc.active)
    same("reward", sim.reward_tensor().to_torch(), orc.reward)
    same("done", sim.done_tensor().to_torch(), orc.done)
    same("game record", sim.game_tensor().to_torch(), orc.dump())


def test_hanabi_65536_worlds_lockstep_vs_oracle(hip_lib, oracle_lib):
    """configs[2] at its own size against the ORACLE: 40 masked-random steps (the reference harness's policy,
    scripts/hanabi_example.py:64-67) through sim.step() -- one launch per step, episode numbers from the look-back over
    all 256 workgroups (reference: one global atomic, src/hanabi_env/sim.cpp:446-451; done + inline reset :812-850)."""
    import os
    n = 65536
    sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **HANABI_FULL)
    assert sim.kernel_name == "mrl_hanabi_step_fused"
    orc = oracle_lib.HanabiOracle(HANABI_FULL, n, num_threads=min(16, os.cpu_count() or 1))
    _hanabi_equal(sim, orc, "initial")
    rng = np.random.default_rng(65536)
    act = sim.action_tensor().to_torch()
    finished = 0
    for t in range(40):
        logits = rng.random(orc.mask.shape, dtype=np.float32) * (orc.mask != 0)
        a = logits.argmax(-1).astype(np.int32)
        orc.step(a)
        act.copy_(torch.from_numpy(a).cuda().view(2, n, 1))
        sim.step()
        _hanabi_equal(sim, orc, f"step {t}")
        finished += int(orc.done.sum())
        assert int(sim.reset_count_tensor().to_torch().item()) == int(orc.done.sum())
    assert finished > n // 2, "too few games ended for the episode numbering to matter"
    assert int(sim.scan_timeout_tensor().to_torch().item()) == 0
    sim.close()


def test_hanabi_65536_worlds_persistent_rollout_vs_oracle(hip_lib, oracle_lib):
    """One mrl_rollout_random(K) -- the persistent cooperative launch, all K steps with the records in LDS -- replayed
    through the oracle with the documented action stream (simulators.random_hanabi_action); then a second call continuing
    the step count, and an ordinary step after it."""
    import os
    from madrona_rl_envs_playground_amd.simulators import random_hanabi_action
    n, seed = 65536, 0x5EED0004
    sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **HANABI_FULL)
    orc = oracle_lib.HanabiOracle(HANABI_FULL, n, num_threads=min(16, os.cpu_count() or 1))
    world = np.arange(n)

    def replay(first, count):
        last = None
        for t in range(first, first + count):
            mover = (orc.active[1] != 0).astype(np.int64)
            want = random_hanabi_action(seed, t, world, mover, orc.mask[mover, world])
            acts = np.zeros((2, n), np.int32)
            acts[mover, world] = want
            orc.step(acts)
            last = (mover, want)
        return last

    assert sim.rollout_kernel_name == "mrl_hanabi_rollout"
    sim.rollout_random(33, seed=seed, first_step=7)
    mover, want = replay(7, 33)
    _hanabi_equal(sim, orc, "after a 33-step persistent launch")
    assert np.array_equal(sim.action_tensor().to_torch().cpu().numpy()[mover, world, 0], want)  # the last step's draws
    sim.rollout_random(9, seed=seed, first_step=40)
    replay(40, 9)
    _hanabi_equal(sim, orc, "after a second launch of 9 steps")
    assert sim.rollout_kernel_name == "mrl_hanabi_rollout", "the runtime refused the cooperative launch: these were launches per step"
    rng = np.random.default_rng(3)
    a = (rng.random(orc.mask.shape, dtype=np.float32) * (orc.mask != 0)).argmax(-1).astype(np.int32)
    orc.step(a)
    sim.action_tensor().to_torch().copy_(torch.from_numpy(a).cuda().view(2, n, 1))
    sim.step()
    _hanabi_equal(sim, orc, "an ordinary step after the rollouts")
    assert int(sim.scan_timeout_tensor().to_torch().item()) == 0
    sim.close()


@pytest.mark.parametrize("fused", [0, 2], ids=["library_choice_single_launch", "two_launches"])
def test_cartpole_one_million_worlds_lockstep_vs_oracle(fused, hip_lib, oracle_lib):
    """1 048 576 worlds x 30 random-action steps against the oracle with the resync protocol of
    tests/test_gpu_cartpole.py::test_lockstep_vs_oracle: one-step differential at 1e-5 (float state; the reference's own check
    is one step too, envs/cartpole_env.py:246-288), fresh states of re-seeded worlds bit for bit -- 50 000 of them per step,
    numbered across all 1024 workgroups (reference: src/cartpole_env/sim.cpp:48-66)."""
    import os
    from madrona_rl_envs_playground_amd._lib import debug_knobs
    n = 1 << 20
    with debug_knobs({"fused_step": fused}):
        sim = CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n)
    assert sim.kernel_name == ("mrl_cartpole_step" if fused == 2 else "mrl_cartpole_step_fused")
    orc = oracle_lib.CartpoleOracle(n, num_threads=min(16, os.cpu_count() or 1))
    st, act = sim.observation_tensor().to_torch(), sim.action_tensor().to_torch()
    assert np.array_equal(st.cpu().numpy().view(np.uint32), orc.state.view(np.uint32)), "initial states"
    torch.manual_seed(0)
    worst, flips, resets, exact_steps = 0.0, 0, 0, 0
    for t in range(30):
        a = torch.randint(0, 2, (n, 1), dtype=torch.int32)
        orc.step(a.numpy())
        act.copy_(a.cuda())
        sim.step()
        got = st.cpu().numpy()
        done_gpu, done_cpu = sim.reset_tensor().to_torch().cpu().numpy()[:, 0], orc.done[:, 0]
        agree = done_gpu == done_cpu
        flips += int((~agree).sum())
        keep = agree & (done_cpu == 0)
        worst = max(worst, float(np.abs(got[keep] - orc.state[keep]).max()))
        if agree.all():  # then every finished world must have taken the oracle's episode number
            r = done_cpu == 1
            resets += int(r.sum())
            exact_steps += 1
            assert np.array_equal(got[r].view(np.uint32), orc.state[r].view(np.uint32)), f"fresh states, step {t}"
            assert int(sim.reset_count_tensor().to_torch().item()) == int(r.sum())
        st.copy_(torch.from_numpy(orc.state).cuda())
        if not agree.all():
            sim.set_episode_counter(orc.episodes)
    assert worst <= 1e-5, f"max |gpu - oracle| over non-terminal worlds = {worst}"
    assert flips <= 30 * n // 200000, f"{flips} done flags disagree"
    assert exact_steps >= 15 and resets > n // 2, (exact_steps, resets)
    assert int(sim.scan_timeout_tensor().to_torch().item()) == 0
    sim.close()
