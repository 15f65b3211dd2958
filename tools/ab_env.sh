#!/bin/bash
# A/B of one environment switch on the QP-active legs, alternating on one box: tools/ab_env.sh lib.so VAR valA valB [legs...]
cd "$(dirname "$0")/.."
LIB=$1; VAR=$2; VA=$3; VB=$4; shift 4
for rep in 1 2 3; do
  for val in $VA $VB; do
    env $VAR=$val LBMPC_LIB=$PWD/$LIB timeout -k 10 200 python -u tools/qp_legs.py "$@" --reps 10 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$VAR=$val', d['leg'], round(d['ms_per_launch'], 3))
" || exit 1
  done
done
