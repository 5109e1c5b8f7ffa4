#!/bin/bash
# A/B of library builds in raytracer_amd/lib/variants/librt_<name>.so (RT_MI355X_LIB), three rounds over the configurations:
#   scripts/ab_variants.sh out.log "c2 c5" base other
out=$1; cfgs=$2; shift 2
: > $out
for rep in 1 2 3; do for c in $cfgs; do for v in "$@"; do
  echo "== $c $v" >> $out; RT_MI355X_LIB=$GRAFT_REPO_ROOT/raytracer_amd/lib/variants/librt_$v.so timeout -k 10 200 python3 scripts/run_config.py $c --steps 6 2>&1 | tail -n 1 | grep -o '"kernel_ms": [0-9.]*' >> $out
done; done; done
