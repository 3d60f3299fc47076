"""GPU: BASELINE.json configs 4 and 5 and the multi-rank protocol.

* config 4: every standard layout at the per-GPU shard of the 8-GPU run (32768 worlds): size-independent
  properties on the whole batch plus a sampled lock-step against the oracle, like test_gpu_fullsize.py does
  for cramped_room; `ShardedSimulator.gather` on a real simulator's tensor, single rank and as a two-rank
  rehearsal on this one GPU (gloo; the driver's multi-GPU runs use one rank per GPU over RCCL);
* config 5: the MAPPO rollout loop (tools/mappo_rollout_loop.py, reference train/MAPPO/main_player.py:211-261):
  the observations the policy receives equal the oracle's for the actions the policies chose;
* `python bench.py --gpus 2` starts its own ranks (rehearsal on one GPU) and prints one JSON line.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu

from madrona_rl_envs_playground_amd import layouts  # noqa: E402
from madrona_rl_envs_playground_amd.distributed import ShardedSimulator, shard_range  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import ExecMode, OvercookedSimulator  # noqa: E402

STANDARD = ["cramped_room", "asymmetric_advantages", "coordination_ring", "forced_coordination", "counter_circuit"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("layout", STANDARD[1:])
def test_standard_layout_full_shard(layout, hip_lib, oracle_lib):
    """configs[3]'s per-GPU shard: 32768 worlds of each standard layout, uniform random actions."""
    horizon, steps = 120, 150
    params = layouts.get_base_layout_params(layout, horizon)
    n, P, H, W = 32768, params["num_players"], params["height"], params["width"]
    C, F = H * W, 5 * P + 16
    sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
    grid, block, lds, wpw = sim.launch_shape
    assert lds <= 46 * 1024 and grid * (block // 64) * wpw >= n  # at least three workgroups per CU's 160 KB
    obs = sim.observation_world_major_tensor().to_torch().view(n, P, C, F)
    terrain = torch.tensor(params["terrain"][:C], device="cuda")
    onehot = torch.zeros(C, 6, dtype=torch.int8, device="cuda")
    nz = terrain > 0
    onehot[nz, (terrain[nz] - 1)] = 1
    sample = 48  # the first and the last `sample` worlds are also stepped by the oracle
    orc = oracle_lib.OvercookedOracle(params, 2 * sample, num_threads=4)
    torch.manual_seed(11)
    total_reward = 0
    for t in range(steps):
        # interact-heavy stream so that pots fill and soups get served at this size too
        a = torch.randint(0, 8, (P, n, 1), dtype=torch.int32, device="cuda").clamp_(max=5)
        sim.step_with_actions(a)
        orc.step(torch.cat([a[:, :sample, 0], a[:, n - sample:, 0]], dim=1).cpu().numpy())
        if t % 25 == 0 or horizon - 2 <= t <= horizon:
            o = obs
            assert torch.equal(o[:, :, :, 5 * P:5 * P + 6], onehot.expand(n, P, C, 6))       # static terrain channels
            assert (o[:, :, :, 0:P].sum(dim=2) == 1).all()                                  # each player on exactly one cell
            assert (o[:, :, :, P:5 * P].sum(dim=(2, 3)) == P).all()                         # one orientation bit per player
            assert (o[:, :, :, 0:P].sum(dim=3) * (terrain != 0).to(torch.int8)).sum() == 0  # players stand on floor
            assert torch.equal(o[:, 0, :, 5 * P:], o[:, 1, :, 5 * P:])                      # tails are viewer-independent
            assert torch.equal(o[:, 0, :, 0], o[:, 1, :, 1]) and torch.equal(o[:, 0, :, P:P + 4], o[:, 1, :, P + 4:P + 8])
            ts = sim.state_timestep_tensor().to_torch()
            assert (ts == (t + 1) % horizon).all()
            assert torch.equal(o[:, 0, 0, F - 1].to(torch.int32), ((horizon - ts) < 40).to(torch.int32))
        rew = sim.reward_tensor().to_torch()
        assert torch.equal(rew[0], rew[1]) and (rew >= 0).all()
        total_reward += int(rew[0].sum())
        done = sim.done_tensor().to_torch()
        assert bool(done.all()) == (t == horizon - 1) and bool(done.any()) == (t == horizon - 1)
        got = torch.cat([obs[:sample], obs[n - sample:]]).cpu().numpy().astype(np.uint8)
        assert np.array_equal(got.reshape(orc.obs.shape), orc.obs), f"sampled worlds differ from the oracle at step {t}"
        assert np.array_equal(torch.cat([rew[:, :sample], rew[:, n - sample:]], dim=1).cpu().numpy(), orc.reward)
    assert total_reward > 0
    sim.close()


def test_gather_on_a_real_simulator_single_rank(hip_lib):
    """ShardedSimulator over an Overcooked simulator without a process group: the gather is the identity
    on the world-major slab and the shard is the whole batch."""
    params = layouts.get_base_layout_params("coordination_ring", 60)
    total = 1000
    sh = ShardedSimulator(lambda k: OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=k, **params), total,
                          needs_episode_exchange=False)
    assert (sh.lo, sh.n) == (0, total)
    torch.manual_seed(3)
    for _ in range(20):
        sh.step(torch.randint(0, 6, (2, total, 1), dtype=torch.int32, device="cuda"))
    local = sh.sim.observation_world_major_tensor().to_torch()
    out = torch.empty_like(local)
    got = sh.gather(local, world_dim=0, out=out)
    assert got.data_ptr() == out.data_ptr() and torch.equal(out, local)
    assert sh.gather(sh.sim.reward_tensor().to_torch(), world_dim=1).shape == (2, total)
    sh.close()


def _gather_rank(rank, ws, port, layout, total, steps, out_dir):
    """One rank of the two-rank rehearsal: its shard of a `total`-world batch on GPU 0, the global action
    stream sliced like the worlds, observations/rewards/dones gathered with ShardedSimulator.gather."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, REPO)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        params = layouts.get_base_layout_params(layout, 40)
        sh = ShardedSimulator(lambda k: OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=k, **params), total,
                              needs_episode_exchange=False)
        assert (sh.lo, sh.n) == shard_range(total, rank, ws)
        gen = torch.Generator().manual_seed(77)  # same global stream on every rank
        obs = sh.sim.observation_world_major_tensor().to_torch()
        snaps = []
        for t in range(steps):
            a = torch.randint(0, 6, (2, total, 1), dtype=torch.int32, generator=gen)
            sh.step(a[:, sh.lo:sh.lo + sh.n].contiguous().cuda())
            if t % 9 == 0 or t == steps - 1:
                full = sh.gather(obs, world_dim=0)
                rew = sh.gather(sh.sim.reward_tensor().to_torch(), world_dim=1)
                done = sh.gather(sh.sim.done_tensor().to_torch(), world_dim=0)
                assert full.shape[0] == total and rew.shape == (2, total) and done.shape == (total,)
                snaps.append((full.cpu(), rew.cpu(), done.cpu()))
        if rank == 0:
            torch.save(snaps, os.path.join(out_dir, "gathered.pt"))
        sh.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [4096, 1001], ids=["equal_shards", "ragged_shards"])
def test_two_rank_gather_equals_one_simulator(total, hip_lib, tmp_path):
    layout, steps, ws = "asymmetric_advantages", 50, 2
    mp.start_processes(_gather_rank, args=(ws, _free_port(), layout, total, steps, str(tmp_path)), nprocs=ws, join=True,
                       start_method="spawn")
    snaps = torch.load(os.path.join(str(tmp_path), "gathered.pt"))
    params = layouts.get_base_layout_params(layout, 40)
    whole = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=total, **params)
    gen = torch.Generator().manual_seed(77)
    k = 0
    for t in range(steps):
        whole.step_with_actions(torch.randint(0, 6, (2, total, 1), dtype=torch.int32, generator=gen).cuda())
        if t % 9 == 0 or t == steps - 1:
            full, rew, done = snaps[k]
            k += 1
            assert torch.equal(full, whole.observation_world_major_tensor().to_torch().cpu()), f"step {t}"
            assert torch.equal(rew, whole.reward_tensor().to_torch().cpu())
            assert torch.equal(done, whole.done_tensor().to_torch().cpu())
    assert k == len(snaps) and k > 3
    whole.close()


def test_bench_starts_its_own_ranks(hip_lib):
    """`python bench.py --gpus 2 ...` without a launcher: rc 0 and ONE JSON line from rank 0 with the
    no-communication `value` and the obs_gather leg (one-GPU rehearsal: both ranks on this card, gloo)."""
    env = dict(os.environ, MRL_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                          env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), proc.stdout  # stdout is the JSON line and nothing else
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["scaling"] == "weak"
    assert out["ranks"] == {"world_size": 2, "backend": "gloo", "rehearsal_on_one_gpu": True}
    assert out["value"] > 0 and out["config"]["obs_gather"] is False
    g = out["obs_gather"]
    assert g["value"] > 0 and g["bytes_per_rank_per_step"] == 32768 * 1040 and len(g["per_rank_ms_per_step"]) == 2
    assert len(out["timing"]["per_rank_ms_per_step"]) == 2 and out["timing"]["blocks"] >= 3
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["launches_timed"] >= 300


def test_bench_line_carries_every_single_gpu_config(hip_lib):
    """`python bench.py` (N = 1): besides the headline, `other_configs` holds the other BASELINE.json configurations one
    GPU can run -- each with kernel, launch duration over >= 300 launches, bytes per world-step and roofline fraction --
    and every `traffic` is either null or comes from PMC passes taken on exactly this build of csrc/."""
    from madrona_rl_envs_playground_amd import _lib
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MRL_BENCH_REHEARSE", "MRL_BENCH_FORCE_DIST"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-seconds", "2",
                           "--large-worlds", "0", "--fused-steps", "100"], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    out = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["roofline"]["bound"] == "hbm" and out["cpu_baseline"]["kind"] == "port"
    assert out["build_hash"] == _lib.build_hash() == _lib.source_hash()
    legs = out["other_configs"]
    want = ([f"overcooked_{name}_32768" for name in STANDARD[1:]] + ["cartpole_1024", "cartpole_1048576", "hanabi_65536"] +
            ["simplecooked_simple_32768", "balance_beam_1048576"])  # (the two sibling worlds of SURVEY section 8(f)-4)
    mappo = legs.pop("mappo_rollout_loop_32768")  # configs[4] on one GPU: the policy's number, and this engine's piece of it
    assert sorted(legs) == sorted(want)
    assert mappo["steps"] == 50 and 0.5 < mappo["policy_share_of_loop"] < 1.0
    assert set(mappo["loop_ms_per_step"]) == set(mappo["env_step_plus_buffer_insert_us"]) == {"step_into_slot", "step_then_clone_insert"}
    # (host-clocked figures from a shared box: sanity, not a race -- the into-slot step is 11 us against 33 us in profiles/)
    assert 0 < mappo["env_step_plus_buffer_insert_us"]["step_into_slot"] < 2 * mappo["env_step_plus_buffer_insert_us"]["step_then_clone_insert"]
    for name in ("hanabi_65536", "cartpole_1048576"):  # the reference quotes its CPU figures beside the GPU's
        cpu = legs[name]["cpu_baseline"]
        assert cpu["kind"] == "port" and cpu["cores"] >= 1 and 0 < cpu["value"] < legs[name]["value"]
    assert legs["cartpole_1048576"]["kernel"] == "mrl_cartpole_step_fused"
    for name, leg in legs.items():
        assert leg["launches_timed"] >= 300 and leg["kernel_us_avg"] > 0 and leg["value"] > 0 and leg["kernel"].startswith("mrl_"), name
        assert abs(leg["frac"] - leg["bytes_per_world_step"] * leg["worlds"] / (leg["kernel_us_avg"] * 1e-6) / 8e12) < 1e-9
    assert legs["hanabi_65536"]["kernel"] == "mrl_hanabi_step_fused" and legs["hanabi_65536"]["bytes_per_world_step"] == 1243
    assert legs["hanabi_65536"]["harness_loop_us_per_step"] > legs["hanabi_65536"]["kernel_us_avg"]
    assert legs["hanabi_65536"]["rollout_kernel"] == "mrl_hanabi_rollout" and legs["hanabi_65536"]["persistent_rollout_us_per_step"] < legs["hanabi_65536"]["kernel_us_avg"]
    # traffic: only ever from this build's PMC passes
    try:
        pmc = json.load(open(os.path.join(REPO, "profiles", "step_traffic.json")))
    except OSError:
        pmc = None
    same_build = pmc is not None and pmc["csrc_sha16"] == _lib.build_hash()
    quoted = [out["roofline"]["traffic"]] + [leg["traffic"] for leg in legs.values()]
    if not same_build:
        assert all(t is None for t in quoted)
    else:
        assert out["roofline"]["traffic"] is not None and 0.9 < out["roofline"]["traffic"] / out["roofline"]["bytes_per_launch"] < 1.3


@pytest.mark.parametrize("layout,in_place", [("cramped_room", True), ("counter_circuit", True), ("cramped_room", False)],
                         ids=["cramped_room_into_slot", "counter_circuit_into_slot", "cramped_room_clone_insert"])
def test_mappo_rollout_loop_policy_sees_oracle_observations(layout, in_place, hip_lib, oracle_lib):
    """configs[4]: the loop of train/MAPPO/main_player.py:211-261 over the drop-in env.  Both players act
    through CNN policies on the int8 observations; replaying the actions they chose through the oracle
    must reproduce exactly the observations each policy was shown, the rewards and the dones -- with the
    step kernel writing every step's observations straight into the rollout buffer's slot (in_place, section 8f item 3:
    no clone, no insert) and with the reference's clone + insert."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import mappo_rollout_loop as loop
    n, horizon, steps = 512, 30, 75
    env, ego, buffers = loop.build(layout, n, horizon=horizon, steps_in_buffer=16, seed=5, in_place=in_place)
    partner = env.partners[0][0]
    ego.keep_inputs = partner.keep_inputs = True
    params = layouts.get_base_layout_params(layout, horizon)
    P, H, W = 2, params["height"], params["width"]
    F = 5 * P + 16
    orc = oracle_lib.OvercookedOracle(params, n, num_threads=4)
    own = env.static_world_major_observations.clone()  # the simulator's own tensor: untouched while steps go into the slots

    def as_env_view(o):  # oracle rows (n, P, C, F) -> per player (n, W, H, F), the wrapper's view
        return o.reshape(n, P, H, W, F).transpose(0, 1, 3, 2, 4)

    seen = {"steps": 0, "dones": 0}

    def check(t, ob_in, ego_action, ob_out, rew, done):
        want = as_env_view(orc.obs)
        assert np.array_equal(ego.last_obs.cpu().numpy().astype(np.uint8), want[:, 0]), f"ego obs differs at step {t}"
        assert np.array_equal(partner.last_obs.cpu().numpy().astype(np.uint8), want[:, 1]), f"partner obs differs at step {t}"
        acts = np.stack([ego.last_action[:, 0].cpu().numpy(), partner.last_action[:, 0].cpu().numpy()]).astype(np.int32)
        orc.step(acts)
        assert np.array_equal(rew.cpu().numpy(), orc.reward[0]) and np.array_equal(done.cpu().numpy(), orc.done)
        assert np.array_equal(ob_out.obs.cpu().numpy().astype(np.uint8), as_env_view(orc.obs)[:, 0])
        slot = buffers["obs"][t % 16]
        if in_place:  # the slot IS what the kernel wrote: (n, P, H, W, F), the observation handed out is a view of it
            assert ob_out.obs.data_ptr() == slot[:, 0].data_ptr()
            assert np.array_equal(slot.cpu().numpy().astype(np.uint8).reshape(orc.obs.shape), orc.obs)
            assert torch.equal(env.static_world_major_observations, own)
        else:
            assert torch.equal(slot, ob_out.obs)
        seen["steps"] += 1
        seen["dones"] += int(done.sum())

    ob = env.reset()
    loop.rollout(env, ego, buffers, ob, steps, on_step=check)
    assert seen["steps"] == steps and seen["dones"] == 2 * n  # two horizons crossed
    env.close()


@pytest.mark.parametrize("mode", ["gloo_two_ranks_one_gpu", "nccl_one_rank"])
def test_mappo_rollout_loop_starts_its_own_ranks(mode, hip_lib):
    """configs[4] names 8 GPUs: `tools/mappo_rollout_loop.py --gpus N` starts one rank per GPU (env shard + policy
    replica, no collective in the loop) and rank 0 prints per-rank and summed env-steps/s.  Rehearsed on this one GPU
    with two gloo ranks, and over nccl (= RCCL) at world_size 1."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MRL_BENCH_REHEARSE", "MRL_BENCH_FORCE_DIST"):
        env.pop(k, None)
    gpus = 2 if mode.startswith("gloo") else 1
    env["MRL_BENCH_REHEARSE" if gpus == 2 else "MRL_BENCH_FORCE_DIST"] = "1"
    proc = subprocess.run([sys.executable, os.path.join(REPO, "tools", "mappo_rollout_loop.py"), "--gpus", str(gpus), "--worlds", "4096",
                           "--steps", "30"], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), proc.stdout  # the JSON line and nothing else
    out = json.loads(lines[0])
    assert out["n_gpus"] == gpus and out["steps"] == 30 and len(out["per_rank_loop_env_steps_per_s"]) == gpus
    assert out["ranks"] == {"world_size": gpus, "backend": "gloo" if gpus == 2 else "nccl", "rehearsal_on_one_gpu": gpus == 2}
    assert 0 < out["loop_env_steps_per_s"] <= out["sum_of_ranks_env_steps_per_s"] * 1.0001
    assert out["observations"].startswith("written into the buffer slot")


@pytest.mark.parametrize("mode", ["forked_streams", "graph", "one_launch", "sequential", "auto"])
def test_several_layouts_as_one_batch(mode, hip_lib):
    """OvercookedMultiLayout: the five standard layouts stepped side by side give, layout by layout, what five separate envs
    give -- on forked streams call by call, as ONE captured HIP graph replayed per step, and as ONE kernel launch
    (mrl_step_many: the grid is the concatenation of the five simulators' grids)."""
    from madrona_rl_envs_playground_amd.envs import OvercookedMadrona
    from madrona_rl_envs_playground_amd.envs.multi_layout import OvercookedMultiLayout
    counts = [300, 77, 512, 64, 129]
    multi = OvercookedMultiLayout(STANDARD, counts, 0, horizon=35, mode=mode)
    assert multi.mode == ("one_launch" if mode == "auto" else mode)
    singles = [OvercookedMadrona(name, n, 0, horizon=35) for name, n in zip(STANDARD, counts)]
    gen = torch.Generator(device="cuda").manual_seed(21)
    for _ in range(80):
        acts = [torch.randint(0, 6, (2, n, 1), device="cuda", generator=gen) for n in counts]
        got = multi.n_step(acts)
        for k, env in enumerate(singles):
            obs, rew, done, _ = env.n_step(acts[k])
            mobs, mrew, mdone, _ = got[k]
            assert all(torch.equal(a.obs, b.obs) for a, b in zip(obs, mobs)), STANDARD[k]
            assert torch.equal(rew, mrew) and torch.equal(done, mdone)
    if multi.mode == "one_launch":  # int32 actions are read where they are
        for _ in range(20):
            acts32 = [torch.randint(0, 6, (2, n, 1), dtype=torch.int32, device="cuda", generator=gen) for n in counts]
            got = multi.n_step(acts32)
            for k, env in enumerate(singles):
                obs, rew, done, _ = env.n_step(acts32[k])
                assert all(torch.equal(a.obs, b.obs) for a, b in zip(obs, got[k][0])) and torch.equal(rew, got[k][1])
    multi.close()
    for env in singles:
        env.close()


@pytest.mark.parametrize("mode", ["auto", "one_launch"])
def test_several_layouts_with_one_that_steps_alone(mode, hip_lib):
    """A large layout at a small world count runs the kernel whose workgroups share one copy of a world, which mrl_step_many
    does not take: the default mode must still step such a batch -- the others in one launch, that one by its own
    call -- and ten small layouts are two launches of at most eight."""
    from madrona_rl_envs_playground_amd.envs import OvercookedMadrona
    from madrona_rl_envs_playground_amd.envs.multi_layout import OvercookedMultiLayout
    from madrona_rl_envs_playground_amd.simulators import can_step_with_others
    names = ["cramped_room", "many_player_layout", "coordination_ring"] + ["cramped_room", "forced_coordination"] * 4
    counts = [200, 500, 130] + [33, 65] * 4
    multi = OvercookedMultiLayout(names, counts, 0, horizon=30, num_players=4, mode=mode)
    players = [env.num_players for env in multi.envs]
    assert players[1] == 4 and players[0] == 2
    assert multi.mode == "one_launch" and multi._alone == [1] and [len(g) for g in multi._shared] == [8, 2]
    assert not can_step_with_others(multi.envs[1].sim) and multi.envs[1].sim.kernel_name.startswith("mrl_overcooked_step_team")
    singles = [OvercookedMadrona(name, n, 0, horizon=30, num_players=4) for name, n in zip(names, counts)]
    gen = torch.Generator(device="cuda").manual_seed(5)
    for t in range(45):
        dtype = torch.int32 if t % 2 else torch.int64
        acts = [torch.randint(0, 6, (p, n, 1), device="cuda", generator=gen, dtype=dtype) for p, n in zip(players, counts)]
        got = multi.n_step(acts)
        for k, env in enumerate(singles):
            obs, rew, done, _ = env.n_step(acts[k])
            assert all(torch.equal(a.obs, b.obs) for a, b in zip(obs, got[k][0])), names[k]
            assert torch.equal(rew, got[k][1]) and torch.equal(done, got[k][2])
    multi.close()
    for env in singles:
        env.close()


def test_one_launch_for_several_simulators_of_any_shape(hip_lib, oracle_lib):
    """mrl_step_many: eight simulators -- different layouts, player counts (2, 3, 4), world counts from 5 to 4099, horizons --
    stepped by one launch per step, each against its own oracle; bad lists are refused."""
    from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, step_many
    specs = [("cramped_room", None, 4099, 30), ("counter_circuit", None, 1000, 50), ("multiplayer_schelling", None, 130, 25), ("cramped_room", None, 5, 400),
             ("asymmetric_advantages", None, 777, 40), ("forced_coordination", None, 64, 33), ("counter_circuit", None, 33, 20), ("coordination_ring", None, 3001, 45)]
    sims, orcs, ps = [], [], []
    for name, cap, n, horizon in specs:
        params = layouts.get_base_layout_params(name, horizon, max_num_players=cap)
        sims.append(OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params))
        orcs.append(oracle_lib.OvercookedOracle(params, n, num_threads=8))
        ps.append(params["num_players"])
    assert len({p for p in ps}) >= 2
    gen = torch.Generator(device="cuda").manual_seed(17)
    for t in range(70):
        acts = [torch.randint(0, 8, (P, spec[2], 1), dtype=torch.int32, device="cuda", generator=gen).clamp_(max=5) for P, spec in zip(ps, specs)]
        step_many(sims, acts)
        for k, (sim, orc) in enumerate(zip(sims, orcs)):
            orc.step(acts[k][:, :, 0].cpu().numpy())
            got = sim.observation_world_major_tensor().to_torch().cpu().numpy().astype(np.uint8)
            assert np.array_equal(got.reshape(orc.obs.shape), orc.obs), f"{specs[k][0]}: observations differ at step {t}"
            assert np.array_equal(sim.reward_tensor().to_torch().cpu().numpy(), orc.reward) and np.array_equal(sim.done_tensor().to_torch().cpu().numpy(), orc.done)
    with pytest.raises(Exception, match="twice"):
        step_many([sims[0], sims[0]])
    # few worlds of a large layout run one state copy per workgroup (another kernel): refused, with a message that says so
    team = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=500, **layouts.get_base_layout_params("many_player_layout", 60, max_num_players=4))
    with pytest.raises(Exception, match="share one state copy"):
        step_many([sims[0], team])
    team.close()
    with pytest.raises(ValueError):
        step_many([sims[0], CartpoleSimulator(ExecMode.CUDA, 0, 8)])
    with pytest.raises(Exception, match="at most"):
        extra = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=4, **layouts.get_base_layout_params("cramped_room", 10))
        step_many(sims + [extra])
    for sim in sims:
        sim.close()
