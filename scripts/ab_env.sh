#!/bin/bash
# Same-box A/B of environment switches: scripts/ab_env.sh OUT ROUNDS "NAME=VAL ..." "NAME2=VAL ..." ...   ("-" = no switch)
# extra bench args through AB_ARGS.  Each line of OUT: <switches> <value> <ms_per_step>
out=$1; rounds=$2; shift 2
: > $out
for i in $(seq 1 $rounds); do
  for sw in "$@"; do
    if [ "$sw" = "-" ]; then e=""; else e="$sw"; fi
    env $e timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 --step-stats 0 $AB_ARGS 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$sw', d['value'], d['ms_per_step'])" >> $out || exit 1
  done
done
cat $out
