#!/bin/bash
# Same-box A/B of builds of the SAME sources with other flags (see tools/hanabi_ab.sh): the five standard layouts' single step,
# rollout and action-sequence launches at a world count.  Usage: tools/layouts_ab.sh <worlds> <name> ...   ("default" = the shipped library)
V=$(cd "$(dirname "$0")/.." && pwd)/madrona_rl_envs_playground_amd/variants
n=$1; shift
for round in 1 2; do
for lay in cramped_room coordination_ring forced_coordination counter_circuit asymmetric_advantages; do
for lib in "$@"; do
  if [ $lib = default ]; then unset MRL_ENVS_LIB; else export MRL_ENVS_LIB=$V/libmrl_$lib.so; fi
  echo -n "$lib: "; python tools/quick_perf.py $lay $n 2>/dev/null
done; done; done
