#!/bin/bash
# Same-box A/B of builds of the SAME sources with other flags (madrona_rl_envs_playground_amd/variants/libmrl_<name>.so, built with
# `make -C csrc OBJDIR=build_<name> OUT=../variants/libmrl_<name>.so CXXFLAGS="... -D..."`): us per step, one launch / two launches / rollout.
# Usage: tools/hanabi_ab.sh <name> ...   ("default" = the shipped library)
V=$(cd "$(dirname "$0")/.." && pwd)/madrona_rl_envs_playground_amd/variants
for round in 1 2; do
for lib in "$@"; do
  if [ $lib = default ]; then unset MRL_ENVS_LIB; else export MRL_ENVS_LIB=$V/libmrl_$lib.so; fi
  python tools/hanabi_probe.py --worlds 65536 --repeat 5 2>> gpurun_out/hanabi_ab.err | python -c "import sys,json; d=json.load(sys.stdin); print('$lib', d['65536']['one_launch_us'], d['65536']['two_launches_us'], d['65536']['rollout_us'])"
done; done
