#!/bin/bash
# One profiling session on the GPU box: everything DESIGN.md quotes.  Usage: tools/profile_round.sh <tag>   (e.g. r02_a)
# Writes under gpurun_out/<tag>_*; the summaries worth keeping are copied into profiles/ by hand afterwards.
set -e
tag=$1
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p $out
cd $root
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "bench done"; tail -c 600 $out/${tag}_bench.json; echo
for l in asymmetric_advantages coordination_ring forced_coordination counter_circuit; do
  python bench.py --layout $l --no-extras --no-cpu-baseline --steps 1000 > $out/${tag}_bench_$l.json 2>> $out/${tag}_bench.err
done
echo "layout benches done"
MRL_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 200 --warmup 20 > $out/${tag}_bench_rehearse2.json 2>> $out/${tag}_bench.err
echo "rehearsal done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt -o kt -- python3 $root/bench.py --steps 2000 --warmup 50 --no-extras --no-cpu-baseline > $out/${tag}_kt.log 2>&1
echo "kernel trace done"
cd $root
bash tools/pmc_passes.sh overcooked 32768 $out/${tag}_pmc
python tools/pmc_summary.py $out/${tag}_pmc --match mrl_overcooked_step > $out/${tag}_overcooked_pmc.txt
python tools/stamps.py > $out/${tag}_overcooked_wave_timeline.txt 2>&1
python tools/stamps_hanabi.py > $out/${tag}_hanabi_wave_timeline.txt 2>&1
python tools/bench_games.py > $out/${tag}_games.json
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_hanabi_kt -o kt -- python3 $root/tools/prof_step.py --game hanabi --worlds 65536 --steps 300 > $out/${tag}_hanabi_kt.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_cartpole_kt -o kt -- python3 $root/tools/prof_step.py --game cartpole --worlds 1048576 --steps 300 > $out/${tag}_cartpole_kt.log 2>&1
cd $root
echo "game traces done"
python tools/scaling_tables.py > $out/${tag}_scaling_tables.json 2>/dev/null
python tools/mappo_rollout_loop.py > $out/${tag}_mappo_rollout_loop.json
# host side of a step call, one launch or two for small batches, GPU time per launch without the host (graph replay) and by phase
python tools/host_overhead.py 32 > $out/${tag}_host_overhead.txt 2>&1
python tools/fused_crossover.py > $out/${tag}_fused_crossover.txt 2>&1
python tools/graph_probe.py > $out/${tag}_graph_probe.txt 2>&1
MRL_ENVS_LIB=$root/madrona_rl_envs_playground_amd/diag/libmrl_envs_diag.so python tools/graph_probe.py --ablate 16,32,8,4,2,0 >> $out/${tag}_graph_probe.txt 2>&1
for l in simple unident_s random0 random3; do python tools/quick_perf_simple.py $l 32768; done > $out/${tag}_simplecooked.txt 2>&1
timeout -k 10 600 python tools/soak_overcooked.py 6000 > $out/${tag}_soak_overcooked.txt 2>&1
echo "all done"
