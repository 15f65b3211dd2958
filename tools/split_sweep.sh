#!/bin/bash
# C4 launch time against the transient/steady split step (LBMPC_MC_SPLIT)
cd "$(dirname "$0")/.."
for s in 9 12 16 24 32 9 12 16 24 32; do
  LBMPC_MC_SPLIT=$s timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-extra-legs --no-cpu-baseline > gpurun_out/split_$s.log 2>&1 || { tail -3 gpurun_out/split_$s.log; exit 1; }
  tail -1 gpurun_out/split_$s.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('split $s', round(d['ms_per_step'],3), 'ms')"
done
