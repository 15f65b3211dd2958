#!/bin/bash
# A/B of two library builds on the QP-active legs, alternating on one box: tools/ab_legs.sh libA.so libB.so [legs...]
cd "$(dirname "$0")/.."
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for lib in $A $B; do
    LBMPC_LIB=$PWD/$lib timeout -k 10 200 python -u tools/qp_legs.py "$@" --reps 10 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib'.split('/')[-1], d['leg'], round(d['ms_per_launch'], 3))
" || exit 1
  done
done
