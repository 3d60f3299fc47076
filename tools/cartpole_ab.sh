#!/bin/bash
# Same-box A/B of builds of the SAME sources with other flags (see tools/hanabi_ab.sh) on the Cartpole step: tools/cartpole_probe.py's lines.
# Usage: tools/cartpole_ab.sh <name> ...   ("default" = the shipped library)
V=$(cd "$(dirname "$0")/.." && pwd)/madrona_rl_envs_playground_amd/variants
for round in 1 2; do
for lib in "$@"; do
  if [ $lib = default ]; then unset MRL_ENVS_LIB; else export MRL_ENVS_LIB=$V/libmrl_$lib.so; fi
  echo "== $lib"; python tools/cartpole_probe.py 1048576 262144 4096 2>/dev/null | grep "step_fused"
done; done
