#!/bin/bash
# kernel times of a diagnostic library build ($1 = path of the .so) on the bench workload
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
LBMPC_LIB=$PWD/$1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --sl-sets 64 > gpurun_out/abl.log 2>&1
f=$(find gpurun_out/abl -name "*kernel_stats.csv" | head -1)
grep -E "mc_" $f | cut -d, -f1-4 | cut -c1-60,100-
