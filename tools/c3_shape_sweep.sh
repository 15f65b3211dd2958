export LBMPC_LIB=${LBMPC_LIB:-$PWD/learning-based-mpc_amd/csrc/liblbmpc_hip.so}
for cfg in "4 10" "8 5" "8 6" "8 8" "8 10" "8 4"; do
  set -- $cfg
  LBMPC_MC_WAVES=$1 LBMPC_MC_TPW=$2 timeout -k 10 100 python -u tools/qp_legs.py C3 --reps 10 --check 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('waves $1 tpw $2', d['leg'], round(d['ms_per_launch'], 3), d.get('parity'))
" || exit 1
done
