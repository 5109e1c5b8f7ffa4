#!/bin/bash
# sweep thresholds on c2; one line per combination
for tt in 8 12 16 24; do for it in 4 8 12 16 24; do
  RT_TRAV_THRESHOLD=$tt RT_INNER_THRESHOLD=$it timeout -k 10 60 python scripts/run_config.py c2 --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print($tt,$it,d['kernel_ms'])" || exit 1
done; done
