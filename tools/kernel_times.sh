#!/bin/bash
# kernel times of a diagnostic library build ($1 = path of the .so) on the bench workload ($BENCH_ARGS: extra bench.py flags)
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/abl; LBMPC_LIB=$PWD/$1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --sl-sets 64 $BENCH_ARGS > gpurun_out/abl.log 2>&1
f=$(find gpurun_out/abl -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mc_" in r["Name"]:
        print("%-28s calls %3s  avg %9.3f ms  total %9.3f ms" % (r["Name"].split("(")[0].replace("void lbmpc::", ""), r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
