#!/bin/bash
export LBMPC_MC_KERNEL=lds
LBMPC_LIB=$PWD/learning-based-mpc_amd/csrc/ablate_SOLVE.so timeout -k 10 200 python -u bench.py --steps 1 --warmup 1 --no-cpu-baseline --sl-sets 64 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('LDS-kernel no SOLVE', d['ms_per_step'])"
timeout -k 10 200 python -u bench.py --steps 1 --warmup 1 --no-cpu-baseline --sl-sets 64 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('LDS-kernel FULL', d['ms_per_step'])"
