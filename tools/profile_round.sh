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
echo "all done"
