#!/bin/bash
# rocprofv3 PMC passes for one game's step kernel, one counter group per pass (FETCH_SIZE and
# WRITE_SIZE cannot share a pass on gfx950).  Usage: tools/pmc_passes.sh <game> <worlds> <outdir>
# Run from anywhere; every pass is bounded by `timeout` and the chain stops at the first failure.
set -e
game=$1; worlds=$2; out=$3
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i + 1))
    timeout -k 10 150 rocprofv3 --pmc $group --output-format csv -d "$out/pass$i" -o p -- python3 "$root/tools/prof_step.py" --game "$game" --worlds "$worlds" --steps 30 > "$out.pass$i.log" 2>&1
    echo "pass $i ($group) done"
done
