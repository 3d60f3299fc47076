"""GPU parity of the Hanabi HIP step against the CPU oracle: observation,
state, legal-move mask, active flags, reward, done and the raw game record,
bit-exact after every step, on masked-random legal actions (the reference
harness samples argmax(rand * mask), scripts/hanabi_example.py:64-67)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from madrona_rl_envs_playground_amd import hanabi_spec  # noqa: E402
from madrona_rl_envs_playground_amd._lib import debug_knobs  # noqa: E402
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
    # reference (see oracle/hanabi_oracle.c).  The OBSERVATION tensor is exactly obs_size wide: it is a view of
    # the state row, which goes on with the agent's own hand -- not for the observer
    no, ns = hanabi_spec.observation_size(cfg), hanabi_spec.state_size(cfg)
    assert sim.observation_tensor().to_torch().shape[-1] == no
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
    assert obs.shape == (2, n, hanabi_spec.observation_size(cfg)) and obs.dtype == torch.int8
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


def test_full_game_encoder_matches_generic_encoder(hip_lib):
    """The full game runs a specialised encoder (constant section offsets, assembled in
    registers); the generic one is what the oracle lock-step covers on the small configurations,
    including the shifted encoding when information tokens exceed their maximum (sim.cpp:676-678).
    Same states, same actions -> identical tensors, also with token counts 9..13 injected."""
    n = 3000
    fast = make(FULL, n)
    with debug_knobs({"hanabi.variant": 0}):
        slow = make(FULL, n)
    gen = torch.Generator(device="cuda").manual_seed(5)

    def tensors(sim):
        return [sim.observation_tensor().to_torch(), sim.agent_state_tensor().to_torch(), sim.action_mask_tensor().to_torch(),
                sim.active_agent_tensor().to_torch(), sim.reward_tensor().to_torch(), sim.done_tensor().to_torch(),
                sim.game_tensor().to_torch()]

    def same(tag):
        for k, (a, b) in enumerate(zip(tensors(fast), tensors(slow))):
            assert torch.equal(a, b), f"tensor {k} differs {tag}"

    same("initially")
    mask = fast.action_mask_tensor().to_torch()
    for t in range(120):
        a = (torch.rand(mask.shape, device="cuda", generator=gen) * mask).argmax(-1, keepdim=True).to(torch.int32)
        fast.action_tensor().to_torch().copy_(a)
        slow.action_tensor().to_torch().copy_(a)
        fast.step()
        slow.step()
        same(f"after step {t}")
        if t % 10 == 5:  # push a third of the worlds over the token maximum (byte 81 of the record = information tokens)
            extra = torch.randint(8, 14, (n,), device="cuda", generator=gen, dtype=torch.int32).to(torch.uint8)
            pick = torch.rand(n, device="cuda", generator=gen) < 0.33
            for sim in (fast, slow):
                rec = sim.game_tensor().to_torch()
                rec[:, 81] = torch.where(pick, extra, rec[:, 81])
    fast.close()
    slow.close()


def test_persistent_rollout_in_all_three_code_variants(hip_lib):
    """The full game through mrl_hanabi_rollout<2> (specialised encoders), <1> (five ranks) and <0> (any configuration): the same
    seed leaves the same tensors after 150 steps in one launch each (the pipeline of phase-A / phase-B / scan waves is one
    template, its phase A differs)."""
    n, sims = 20000, []
    for variant in (2, 1, 0):
        with debug_knobs({"hanabi.variant": variant}):
            sims.append(make(FULL, n))
    assert all(s.rollout_kernel_name == "mrl_hanabi_rollout" for s in sims)
    for s in sims:
        s.rollout_random(150, seed=77, first_step=0)
    names = ["observation_tensor", "agent_state_tensor", "action_mask_tensor", "active_agent_tensor", "reward_tensor", "done_tensor",
             "game_tensor", "reset_count_tensor", "action_tensor"]
    for other in sims[1:]:
        for name in names:
            assert torch.equal(getattr(sims[0], name)().to_torch(), getattr(other, name)().to_torch()), name
    assert int(sims[0].reset_count_tensor().to_torch().item()) >= 0 and all(int(s.scan_timeout_tensor().to_torch().item()) == 0 for s in sims)
    for s in sims:
        s.close()


@pytest.mark.parametrize("cfg,n,steps,heal", [(FULL, 5000, 80, 0), (FULL, 70001, 40, 0), (SMALL, 3000, 60, 0), (FULL, 70001, 60, 3),
                                              (FULL, 9000, 80, 1), (VERY_SMALL, 3000, 60, 2)],
                         ids=["full", "full_70001", "small", "full_70001_late_workgroups", "full_9000_all_late", "very_small_late"])
def test_single_launch_step_equals_two_phase(cfg, n, steps, heal, hip_lib):
    """mrl_step as ONE launch (every workgroup publishes its finished count, its last wave looks back at the lower ones, a
    count that does not appear is recounted from that workgroup's records and actions: csrc/episode_scan.hpp); the
    two-phase calls are two launches.  Same tensors either way -- also when workgroups arrive late (`fused_heal_test`
    = m: workgroups whose index is a multiple of m do nothing until a higher one has recounted them, so the recount
    path runs against records nobody has touched yet)."""
    with debug_knobs({"fused_step": 1, "fused_heal_test": heal}):
        s1 = make(cfg, n)
    with debug_knobs({"fused_step": 2}):
        s2 = make(cfg, n)
    assert s1.kernel_name == "mrl_hanabi_step_fused" and s2.kernel_name == "mrl_hanabi_step"
    gen = torch.Generator(device="cuda").manual_seed(11)
    mask = s1.action_mask_tensor().to_torch()
    names = ["observation_tensor", "agent_state_tensor", "action_mask_tensor", "active_agent_tensor", "reward_tensor",
             "done_tensor", "game_tensor", "reset_count_tensor"]
    total = 0
    for t in range(steps):
        a = (torch.rand(mask.shape, device="cuda", generator=gen) * mask).argmax(-1, keepdim=True).to(torch.int32)
        s1.action_tensor().to_torch().copy_(a)
        s1.step()
        s2.step_phase1(a)
        s2.step_phase2(None)
        for name in names:
            assert torch.equal(getattr(s1, name)().to_torch(), getattr(s2, name)().to_torch()), f"{name} differs at step {t}"
        total += int(s1.reset_count_tensor().to_torch().item())
    assert total > 0
    assert int(s1.scan_timeout_tensor().to_torch().item()) == 0
    s1.close()
    s2.close()


@pytest.mark.parametrize("fused", [True, False], ids=["single_launch", "two_launch"])
def test_device_random_policy(fused, hip_lib, oracle_lib):
    """mrl_rollout_random == the oracle fed the documented stream (uniform over the mover's legal
    moves, include/mrl_envs.h): checked step by step, then a multi-step call against a replay."""
    from madrona_rl_envs_playground_amd.simulators import random_hanabi_action
    n, seed = 2500, 0xC0FFEE1234
    with debug_knobs({"fused_step": 1 if fused else 2}):
        sim, twin = make(FULL, n), make(FULL, n)
    orc = oracle_lib.HanabiOracle(FULL, n, num_threads=8)
    world = np.arange(n)
    hist = np.zeros(20, np.int64)
    for t in range(90):
        mover = (orc.active[1] != 0).astype(np.int64)            # agent whose turn it is
        legal = orc.mask[mover, world]                            # (n, 20)
        want = random_hanabi_action(seed, 500 + t, world, mover, legal)
        assert (legal[world, want] != 0).all()
        sim.rollout_random(1, seed=seed, first_step=500 + t)
        got = sim.action_tensor().to_torch().cpu().numpy()[mover, world, 0]
        assert np.array_equal(got, want), f"drawn actions differ at step {t}"
        acts = np.zeros((2, n), np.int32)
        acts[mover, world] = want
        orc.step(acts)
        compare(sim, orc, f"step {t}", FULL)
        hist += np.bincount(want, minlength=20)
    assert (hist > 0).all(), "some move kind was never drawn"
    twin.rollout_random(90, seed=seed, first_step=500)           # one call, 90 steps
    assert torch.equal(twin.game_tensor().to_torch(), sim.game_tensor().to_torch())
    assert torch.equal(twin.observation_tensor().to_torch(), sim.observation_tensor().to_torch())
    sim.close()
    twin.close()


def test_long_games_reach_the_empty_deck(hip_lib, oracle_lib):
    """A policy that rarely plays keeps its life tokens, so games run until the deck is empty:
    the shift-left removal from a hand, short hands and the turn countdown (sim.cpp:567-600) are
    all exercised, in lock-step with the oracle."""
    n, steps = 600, 230
    sim, orc = make(FULL, n), oracle_lib.HanabiOracle(FULL, n, num_threads=8)
    rng = np.random.default_rng(99)
    act = sim.action_tensor().to_torch()
    short_hands = empty_decks = finished = 0
    for t in range(steps):
        mask = orc.mask.copy()
        keep_plays = rng.random((2, n)) < 0.03
        plays = mask[..., 5:10].copy()
        mask[..., 5:10] = np.where(keep_plays[..., None], plays, 0)
        none_left = mask.sum(-1) == 0                      # only plays were legal
        mask[..., 5:10] = np.where(none_left[..., None], plays, mask[..., 5:10])
        a = legal_random(rng, mask)
        orc.step(a)
        act.copy_(torch.from_numpy(a).cuda().view(2, n, 1))
        sim.step()
        compare(sim, orc, f"step {t}", FULL)
        rec = orc.dump()
        empty_decks += int((rec[:, 50] == 0).sum())
        short_hands += int(((rec[:, 105] < 5) | (rec[:, 141] < 5)).sum())
        finished += int(orc.done.sum())
    assert empty_decks > 0 and short_hands > 0 and finished > 0
    sim.close()


@pytest.mark.parametrize("cfg,n", [(FULL, 65536), (FULL, 10000 + 37), (SMALL, 4096)], ids=["full_65536", "full_ragged", "small"])
def test_persistent_rollout_equals_stepwise(cfg, n, hip_lib):
    """mrl_rollout_random keeps the records in LDS for all steps of a call (one launch) when the
    grid fits the GPU; forced back to one launch per step it must leave identical tensors."""
    one = make(cfg, n)
    with debug_knobs({"hanabi.no_persistent": 1}):
        many = make(cfg, n)
    assert one.rollout_kernel_name == "mrl_hanabi_rollout" and many.rollout_kernel_name == many.kernel_name
    names = ["observation_tensor", "agent_state_tensor", "action_mask_tensor", "active_agent_tensor", "reward_tensor",
             "done_tensor", "game_tensor", "reset_count_tensor", "action_tensor"]
    step = 0
    for chunk in (1, 2, 7, 40, 1, 64):
        one.rollout_random(chunk, seed=2024, first_step=step)
        many.rollout_random(chunk, seed=2024, first_step=step)
        step += chunk
        for name in names:
            assert torch.equal(getattr(one, name)().to_torch(), getattr(many, name)().to_torch()), f"{name} differs after {step} steps"
    # and an ordinary step continues from either
    mask = one.action_mask_tensor().to_torch()
    a = (torch.rand(mask.shape, device="cuda") * mask).argmax(-1, keepdim=True).to(torch.int32)
    for sim in (one, many):
        sim.action_tensor().to_torch().copy_(a)
        sim.step()
    assert torch.equal(one.game_tensor().to_torch(), many.game_tensor().to_torch())
    assert int(one.scan_timeout_tensor().to_torch().item()) == 0
    assert one.rollout_kernel_name == "mrl_hanabi_rollout", "the runtime refused the cooperative launch: these were launches per step"
    one.close()
    many.close()


@pytest.mark.parametrize("script,n", [("script_card_moves", 200), ("script_empty_deck", 96), ("script_complete_a_firework", 300)])
def test_known_answers_by_hand_on_gpu(script, n, hip_lib):
    """tests/hanabi_by_hand.py -- scripted games worked out from the reference text, NOT from the oracle: a successful and a
    failed play, a discard, the knowledge reset of a redrawn slot, the shift-left of a hand on an empty deck, a completed
    firework with the ninth information token and its shifted encoding -- replayed
    through the HIP step: every entry of the mover's observation, the state's own-hand tail and the legal moves, for the
    single-launch step and for the two-launch pair."""
    import hanabi_by_hand as by_hand
    scripts = [getattr(by_hand, script)(w) for w in range(n)]
    for knob in (1, 2):
        with debug_knobs({"fused_step": knob}):
            sim = make(FULL, n)

        def step(acts):
            sim.step_with_actions(torch.from_numpy(acts).cuda().view(2, n, 1).contiguous())

        def read():
            return (sim.observation_tensor().to_torch().cpu().numpy(), sim.agent_state_tensor().to_torch().cpu().numpy(),
                    sim.action_mask_tensor().to_torch().cpu().numpy(), sim.active_agent_tensor().to_torch().cpu().numpy(),
                    sim.done_tensor().to_torch().cpu().numpy())
        kinds = by_hand.run_scripts(step, read, scripts)
        assert len(kinds) >= 3
        sim.close()


@pytest.mark.parametrize("policy,n,steps,reason", [("policy_score_then_lose", 600, 40, "life"), ("policy_run_out_the_deck", 300, 90, "turns")])
def test_endings_and_next_episodes_by_hand_on_gpu(policy, n, steps, reason, hip_lib):
    """Whole games worked out by hand (tests/hanabi_by_hand.py, not the oracle), endings included: every step's reward and
    done -- the move that burns the last life token is paid minus the score --, and after an ending BOTH agents' rows of the
    world's next game, dealt from the episode index it gets when the step's finished worlds (several workgroups' worth) take
    the indices in ascending world order; through the single-launch step and through the two-launch pair."""
    import hanabi_by_hand as by_hand
    for knob in (1, 2):
        with debug_knobs({"fused_step": knob}):
            sim = make(FULL, n)

        def step(acts):
            sim.step_with_actions(torch.from_numpy(acts).cuda().view(2, n, 1).contiguous())

        def read():
            return tuple(getattr(sim, name)().to_torch().cpu().numpy() for name in
                         ("observation_tensor", "agent_state_tensor", "action_mask_tensor", "active_agent_tensor", "done_tensor", "reward_tensor"))
        started, reasons = by_hand.run_policy_games(step, read, n, getattr(by_hand, policy), steps)
        assert reasons == {reason} and started >= n
        assert int(sim.reset_count_tensor().to_torch().item()) >= 0
        sim.close()


@pytest.mark.parametrize("n", [70001, 300])
def test_phase_a_pairings_agree(n, hip_lib):
    """The single-launch step runs phase A in four "leader" waves on all 64 lanes (their own worlds and a partner wave's);
    `hanabi.pairing` picks the partner (4: wave w + 4, 1: wave 2k + 1, 0: every wave for itself).  Same tensors whichever."""
    sims = []
    for pairing in (4, 1, 0):
        with debug_knobs({"fused_step": 1, "hanabi.pairing": pairing}):
            sims.append(make(FULL, n))
    assert all(s.kernel_name == "mrl_hanabi_step_fused" for s in sims)
    gen = torch.Generator(device="cuda").manual_seed(3)
    mask = sims[0].action_mask_tensor().to_torch()
    names = ["observation_tensor", "agent_state_tensor", "action_mask_tensor", "active_agent_tensor", "reward_tensor", "done_tensor",
             "game_tensor", "reset_count_tensor"]
    finished = 0
    for t in range(70):
        a = (torch.rand(mask.shape, device="cuda", generator=gen) * mask).argmax(-1, keepdim=True).to(torch.int32)
        for s in sims:
            s.step_with_actions(a)
        for name in names:
            ref = getattr(sims[0], name)().to_torch()
            for k, s in enumerate(sims[1:]):
                assert torch.equal(ref, getattr(s, name)().to_torch()), f"{name}: pairing variant {k + 1} differs at step {t}"
        finished += int(sims[0].reset_count_tensor().to_torch().item())
    assert finished > 0
    for s in sims:
        s.close()


@pytest.mark.parametrize("n,fused", [(9000, 1), (70001, 1), (70001, 2)], ids=["one_launch_9000", "one_launch_70001", "two_launches_70001"])
def test_steps_captured_in_a_hip_graph_after_prepare(n, fused, hip_lib):
    """mrl_prepare_graph_capture: the episode counter's parity and the look-back's epoch live in device memory from then on, so
    a captured sequence of three steps (legal moves drawn by the step kernel from a fixed stream) replayed 25 times -- with
    eager steps in between -- equals the same calls issued one by one on an ordinary simulator."""
    with debug_knobs({"fused_step": fused, "hanabi.no_persistent": 1}):
        eager, graphed = make(FULL, n), make(FULL, n)
    graphed.prepare_graph_capture()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            graphed.rollout_random(3, seed=21, first_step=100)
    torch.cuda.current_stream().wait_stream(side)
    names = ["observation_tensor", "agent_state_tensor", "action_mask_tensor", "active_agent_tensor", "reward_tensor", "done_tensor",
             "game_tensor", "reset_count_tensor", "action_tensor"]
    finished = 0
    for rep in range(25):
        graph.replay()
        eager.rollout_random(3, seed=21, first_step=100)
        if rep % 4 == 0:
            graphed.rollout_random(1, seed=5, first_step=rep)
            eager.rollout_random(1, seed=5, first_step=rep)
        for name in names:
            assert torch.equal(getattr(eager, name)().to_torch(), getattr(graphed, name)().to_torch()), f"{name} differs after replay {rep}"
        finished += int(eager.reset_count_tensor().to_torch().item())
    assert finished > 0
    eager.close()
    graphed.close()
