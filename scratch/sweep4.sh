#!/bin/bash
timeout -k 10 60 python scripts/run_config.py c2 --stats 2>/dev/null | tail -n 1
for lt in 12 20 28 40; do for it in 4 8 16; do for tt in 8 16; do
  RT_LEAVE_THRESHOLD=$lt RT_INNER_THRESHOLD=$it RT_TRAV_THRESHOLD=$tt timeout -k 10 60 python scripts/run_config.py c2 --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('leave',$lt,'inner',$it,'trav',$tt,d['kernel_ms'])" || exit 1
done; done; done
