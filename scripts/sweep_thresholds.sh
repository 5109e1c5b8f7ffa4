#!/bin/bash
# Sweeps the three schedule thresholds of the wave-scheduled render kernel on configuration c2 (tuning aid;
# the library reads RT_LEAVE_THRESHOLD / RT_INNER_THRESHOLD / RT_TRAV_THRESHOLD when it builds launch parameters).
#   scripts/sweep_thresholds.sh "12 20 28" "4 8 16" "8 12 16"
for lt in ${1:-20}; do for it in ${2:-8}; do for tt in ${3:-12}; do
  RT_LEAVE_THRESHOLD=$lt RT_INNER_THRESHOLD=$it RT_TRAV_THRESHOLD=$tt timeout -k 10 60 python3 scripts/run_config.py c2 --steps 6 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('leave',$lt,'inner',$it,'trav',$tt,d['kernel_ms'])" || exit 1
done; done; done
