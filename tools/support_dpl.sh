#!/bin/bash
# support_kernel time at 1 / 2 / 4 directions per lane (LBMPC_SUPPORT_DPL) for two batch sizes -- informational
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for d in 1 2 4; do for n in 1000000 250000; do
  rm -rf gpurun_out/sp
  LBMPC_SUPPORT_DPL=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --sl-sets 64 --trials 4096 --horizon 10 --sweep-dirs $n > gpurun_out/sp.log 2>&1 || exit 1
  python - "$(find gpurun_out/sp -name '*kernel_stats.csv' | head -1)" $d $n <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "support_kernel" in r["Name"]:
        print("dpl %s n %s: calls %s avg %.2f us min %.2f us" % (sys.argv[2], sys.argv[3], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done; done
rm -rf gpurun_out/sp
