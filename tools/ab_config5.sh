#!/bin/bash
# A/B of the config-5 job (bounded): seeds and shared exact passes on/off, two rounds, one box.  usage: bash tools/ab_config5.sh [lib] [scenarios]
export MP_HIP_LIB=$PWD/${1:-motion_planning_amd/libmp_hip.so}
n=${2:-4000}
out=gpurun_out/r03_ab_config5.txt
: > $out
for round in 1 2; do
  for combo in "1 1" "0 1" "1 0" "0 0"; do
    set -- $combo
    MP_SEEDS=$1 MP_DEFER=$2 timeout -k 10 200 python tools/run_config5.py --scenarios $n --restarts 32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('seeds=$1 defer=$2 round=$round: %.0f scenarios/s, %.4g evals/s, solved %d, cost sum %.9f' % (d['value'], d['evals_per_s'], d['solved'], d['best_cost_sum_of_solved']))" >> $out || exit 1
  done
done
cat $out
