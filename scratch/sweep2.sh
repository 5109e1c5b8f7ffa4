#!/bin/bash
for lt in 64 40 32 24 16 8; do for it in 4 8 16; do
  RT_LEAVE_THRESHOLD=$lt RT_INNER_THRESHOLD=$it timeout -k 10 60 python scripts/run_config.py c2 --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('leave',$lt,'inner',$it,d['kernel_ms'])" || exit 1
done; done
