#!/bin/bash
# diagnostic: time the MC kernel with one phase stubbed out (builds made with -DLBMPC_ABLATE_<PHASE>)
for v in SAMPLER SOLVE FIRSTPASS; do LBMPC_LIB=$PWD/learning-based-mpc_amd/csrc/ablate_$v.so timeout -k 10 200 python -u bench.py --steps 1 --warmup 1 --no-cpu-baseline --sl-sets 64 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"; done
timeout -k 10 200 python -u bench.py --steps 1 --warmup 1 --no-cpu-baseline --sl-sets 64 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('FULL', d['ms_per_step'])"
