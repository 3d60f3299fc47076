#!/bin/bash
# One profiling session on the GPU box: everything DESIGN.md quotes.  Usage: tools/profile_round.sh <tag> [a|b]   (e.g. r04_zz)
# Part a: PMC traffic table, bench lines, rank-protocol legs, kernel traces, PMC instruction counters; part b: the per-game tools,
# soaks and the diagnostic build's timelines (a gpurun call is limited to 20 minutes; no part = both).
# Writes under gpurun_out/<tag>_*; the summaries worth keeping are copied into profiles/ by hand afterwards
# (gpurun_out/<tag>_traffic/step_traffic.json -> profiles/step_traffic.json: it carries the hash of csrc/ it was taken on).
set -e
tag=$1
part=${2:-ab}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p $out
cd $root
if [[ $part == *a* ]]; then
# the PMC traffic table first: bench.py quotes profiles/step_traffic.json when its csrc hash is the running library's
timeout -k 10 600 python3 tools/pmc_traffic.py $out/${tag}_traffic > $out/${tag}_traffic.log 2>&1
cp $out/${tag}_traffic/step_traffic.json profiles/step_traffic.json
echo "pmc traffic done"
( time python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err ) 2> $out/${tag}_bench_time.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_steps20.json 2>> $out/${tag}_bench.err ) 2>> $out/${tag}_bench_time.txt
echo "bench done"; head -c 400 $out/${tag}_bench.json; echo; grep real $out/${tag}_bench_time.txt
MRL_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 500 --warmup 20 > $out/${tag}_bench_nccl_world_size_1.json 2>> $out/${tag}_bench.err
MRL_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 200 --warmup 20 > $out/${tag}_bench_rehearse2_one_gpu.json 2>> $out/${tag}_bench.err
echo "rank protocol legs done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt -o kt -- python3 $root/bench.py --steps 2000 --warmup 50 --no-extras --no-cpu-baseline > $out/${tag}_kt.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_hanabi_kt -o kt -- python3 $root/tools/prof_step.py --game hanabi --worlds 65536 --steps 300 > $out/${tag}_hanabi_kt.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_cartpole_kt -o kt -- python3 $root/tools/prof_step.py --game cartpole --worlds 1048576 --steps 300 > $out/${tag}_cartpole_kt.log 2>&1
echo "kernel traces done"
cd $root
bash tools/pmc_passes.sh hanabi 65536 $out/${tag}_hanabi_pmc
python tools/pmc_summary.py $out/${tag}_hanabi_pmc --match mrl_hanabi > $out/${tag}_hanabi_pmc.txt
bash tools/pmc_passes.sh overcooked 32768 $out/${tag}_pmc
python tools/pmc_summary.py $out/${tag}_pmc --match mrl_overcooked_step > $out/${tag}_overcooked_pmc.txt
echo "pmc done"
fi
if [[ $part == *b* ]]; then
python tools/bench_games.py > $out/${tag}_games.json
python tools/bench_games.py --knob fused_step=2 > $out/${tag}_games_two_launch.json
python tools/sharded_step_cost.py > $out/${tag}_sharded_step_cost.json
python tools/sharded_step_cost.py --nccl > $out/${tag}_sharded_step_cost_nccl.json 2>> $out/${tag}_bench.err
python tools/fused_crossover.py > $out/${tag}_fused_crossover.txt 2>&1
python tools/small_batch_probe.py > $out/${tag}_small_batch_probe.json
python tools/mappo_rollout_loop.py > $out/${tag}_mappo_rollout_loop.json
python tools/mappo_rollout_loop.py --copy-insert > $out/${tag}_mappo_rollout_loop_clone_insert.json
MRL_BENCH_FORCE_DIST=1 python tools/mappo_rollout_loop.py --gpus 1 > $out/${tag}_mappo_rollout_loop_nccl_world_size_1.json 2>> $out/${tag}_bench.err
python tools/scaling_tables.py > $out/${tag}_scaling_tables.json 2>/dev/null
python tools/hanabi_probe.py --worlds 32 1000 65536 > $out/${tag}_hanabi_probe.json 2>> $out/${tag}_bench.err
python tools/hanabi_traffic_account.py > $out/${tag}_hanabi_traffic_account.json 2>> $out/${tag}_bench.err
python tools/cartpole_probe.py > $out/${tag}_cartpole_probe.txt 2>&1
python tools/cartpole_rollout_probe.py > $out/${tag}_cartpole_rollout_probe.txt 2>&1
python tools/cartpole_variants.py > $out/${tag}_cartpole_variants.json 2>> $out/${tag}_bench.err
echo "games / loops done"
python tools/soak_single_launch.py 10000 > $out/${tag}_soak_single_launch.txt 2>&1
python tools/soak_rollouts.py > $out/${tag}_soak_rollouts.txt 2>&1
echo "soaks done"
make -C madrona_rl_envs_playground_amd/csrc -j16 diag > $out/${tag}_diag_build.log 2>&1
python tools/stamps_hanabi_fused.py > $out/${tag}_hanabi_fused_timeline.txt 2>&1
python tools/stamps_hanabi_rollout.py > $out/${tag}_hanabi_rollout_timeline.txt 2>&1
python tools/stamps_cartpole.py > $out/${tag}_cartpole_fused_timeline.txt 2>&1
python tools/stamps.py > $out/${tag}_overcooked_wave_timeline.txt 2>&1
python tools/graph_probe.py > $out/${tag}_graph_probe.txt 2>&1
echo "all done"
fi
