#!/bin/bash
for lt in 12 16 20 24 28; do for tt in 8 12 20 32; do
  RT_LEAVE_THRESHOLD=$lt RT_TRAV_THRESHOLD=$tt timeout -k 10 60 python scripts/run_config.py c2 --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('leave',$lt,'trav',$tt,d['kernel_ms'])" || exit 1
done; done
