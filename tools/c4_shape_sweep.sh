IFS=";" read -ra CFGS <<< "${C4_CFGS:-1 64 8;2 64 8;2 32 8;1 64 4}"
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  LBMPC_MC_T=$1 LBMPC_MC_TPW=$2 LBMPC_MC_WAVES=$3 timeout -k 10 100 python -u tools/qp_legs.py C4t --reps 10 --check 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('T $1 tpw $2 waves $3', d['leg'], round(d['ms_per_launch'], 3), d.get('parity'), d['counters']['lane_scans'], d['counters']['lane_trips'])
" || exit 1
done
