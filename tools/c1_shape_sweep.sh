export LBMPC_LIB=$PWD/learning-based-mpc_amd/csrc/dev_lbmpc.so
for cfg in "8 32 32" "4 64 32" "4 64 64" "8 64 32" "6 43 32" "8 32 16"; do
  set -- $cfg
  LBMPC_MC_WAVES=$1 LBMPC_MC_TPW=$2 LBMPC_GEN_LANES=$3 timeout -k 10 100 python -u tools/qp_legs.py C1 --reps 10 --check 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('waves $1 tpw $2 gen lanes $3', d['leg'], round(d['ms_per_launch'], 3), d.get('parity'))
" || exit 1
done
