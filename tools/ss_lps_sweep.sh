#!/bin/bash
# launch time of the QP-active legs against the lanes per information set of the I0 pre-pass (LBMPC_SS_LPS; 0 = heuristic)
cd "$(dirname "$0")/.."
for lps in 0 4 8 16 32; do
  LBMPC_SS_LPS=$lps timeout -k 10 200 python -u tools/qp_legs.py C1 C4t --reps 10 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('LBMPC_SS_LPS=$lps', d['leg'], round(d['ms_per_launch'], 3))
" || exit 1
done
