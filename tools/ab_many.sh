#!/bin/bash
# same-box comparison of several library builds on the QP-active legs, alternating, two rounds:
#   tools/ab_many.sh "C1 C3" libA.so libB.so libC.so ...
cd "$(dirname "$0")/.."
LEGS=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    LBMPC_LIB=$PWD/$lib timeout -k 10 200 python -u tools/qp_legs.py $LEGS --reps 10 2>gpurun_out/ab_err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib'.split('/')[-1], d['leg'], round(d['ms_per_launch'], 3))
" || exit 1
  done
done
