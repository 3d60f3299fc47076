"""GPU parity of the Hanabi HIP step against the CPU oracle: observation,
state, legal-move mask, active flags, reward, done and the raw game record,
bit-exact after every step, on masked-random legal actions (the reference
harness samples argmax(rand * mask), scripts/hanabi_example.py:64-67)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from madrona_rl_envs_playground_amd import hanabi_spec  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import ExecMode, HanabiSimulator  # noqa: E402

FULL = dict(colors=5, ranks=5, players=2, max_information_tokens=8, max_life_tokens=3)
SMALL = dict(colors=2, ranks=5, players=2, max_information_tokens=3, max_life_tokens=1)
VERY_SMALL = dict(colors=1, ranks=5, players=2, max_information_tokens=3, max_life_tokens=1)


def make(cfg, n):
    return HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **cfg)


def legal_random(rng, mask):
    """mask (2,N,20) int -> (2,N) int32: a uniformly random legal move per agent."""
    logits = rng.random(mask.shape) * (mask != 0)
    return logits.argmax(-1).astype(np.int32)


def compare(sim, orc, tag, cfg):
    # compared on the windows the reference wrapper exposes (envs/hanabi_env.py:92-104:
    # [:obs_size], [:state_size]); bytes past them are leftovers of longer encodings in the
    # reference (see oracle/hanabi_oracle.c) and zero here
    no, ns = hanabi_spec.observation_size(cfg), hanabi_spec.state_size(cfg)
    got_o = sim.observation_tensor().to_torch().cpu().numpy().astype(np.uint8)
    got_s = sim.agent_state_tensor().to_torch().cpu().numpy().astype(np.uint8)
    assert np.array_equal(got_o[..., :no], orc.obs[..., :no]), f"obs {tag}"
    assert np.array_equal(got_s[..., :ns], orc.state[..., :ns]), f"state {tag}"
    assert np.array_equal(sim.action_mask_tensor().to_torch().cpu().numpy(), orc.mask), f"mask {tag}"
    assert np.array_equal(sim.active_agent_tensor().to_torch().cpu().numpy(), orc.active), f"active {tag}"
    assert np.array_equal(sim.reward_tensor().to_torch().cpu().numpy(), orc.reward), f"reward {tag}"
    assert np.array_equal(sim.done_tensor().to_torch().cpu().numpy(), orc.done), f"done {tag}"
    assert np.array_equal(sim.game_tensor().to_torch().cpu().numpy(), orc.dump()), f"game record {tag}"


@pytest.mark.parametrize("cfg,n,steps", [(FULL, 1000, 260), (FULL, 4133, 90), (SMALL, 700, 150),
                                         (VERY_SMALL, 333, 120)], ids=["full", "full_4133", "small", "very_small"])
def test_lockstep_vs_oracle(cfg, n, steps, hip_lib, oracle_lib):
    sim, orc = make(cfg, n), oracle_lib.HanabiOracle(cfg, n, num_threads=8)
    obs = sim.observation_tensor().to_torch()
    assert obs.shape == (2, n, 658) and obs.dtype == torch.int8
    assert sim.agent_state_tensor().to_torch().shape == (2, n, 783)
    assert sim.action_mask_tensor().to_torch().shape == (2, n, 20)
    assert sim.reward_tensor().to_torch().dtype == torch.float32
    compare(sim, orc, "initial", cfg)
    rng = np.random.default_rng(n)
    act = sim.action_tensor().to_torch()
    finished = 0
    for t in range(steps):
        a = legal_random(rng, orc.mask)
        orc.step(a)
        act.copy_(torch.from_numpy(a).cuda().view(2, n, 1))
        sim.step()
        compare(sim, orc, f"step {t}", cfg)
        finished += int(orc.done.sum())
        assert int(sim.reset_count_tensor().to_torch().item()) == int(orc.done.sum())
    assert finished > 0, "no episode finished; the reset path was not exercised"
    sim.close()


def test_sharded_episode_numbering(hip_lib, oracle_lib):
    """Two shards + exchanged reset counts == one simulator of the whole batch."""
    n, half = 2048, 1024
    whole, lo, hi = make(FULL, n), make(FULL, half), make(FULL, half)
    lo.reseed_shard(0, n)
    hi.reseed_shard(half, n)
    counter = torch.tensor([n], dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(5)
    for _ in range(130):
        mask = whole.action_mask_tensor().to_torch().cpu().numpy()
        a = torch.from_numpy(legal_random(rng, mask)).cuda()
        whole.step_with_actions(a.view(2, n, 1).contiguous())
        lo.step_phase1(a[:, :half].contiguous())
        hi.step_phase1(a[:, half:].contiguous())
        c_lo = lo.done_tensor().to_torch().sum().to(torch.int32).reshape(1)
        c_hi = hi.done_tensor().to_torch().sum().to(torch.int32).reshape(1)
        lo.step_phase2(counter)
        hi.step_phase2(counter + c_lo)
        counter = counter + c_lo + c_hi
        for name in ("observation_tensor", "agent_state_tensor", "action_mask_tensor", "reward_tensor",
                     "active_agent_tensor"):
            both = torch.cat([getattr(lo, name)().to_torch(), getattr(hi, name)().to_torch()], dim=1)
            assert torch.equal(both, getattr(whole, name)().to_torch()), name
    for s in (whole, lo, hi):
        s.close()


def test_rejects_bad_config(hip_lib):
    with pytest.raises(RuntimeError):
        make(dict(FULL, players=3), 8)
