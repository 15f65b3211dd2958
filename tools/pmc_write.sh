#!/bin/bash
# WRITE_SIZE of the MC kernels at the default bench size (scratch-spill traffic shows up here)
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/pw
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pw -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --sl-sets 64 > gpurun_out/pw.log 2>&1 || { tail gpurun_out/pw.log; exit 1; }
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pw/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]) * 1024 / 1e9)
for k, v in agg.items():
    if "mc_" in k: print(k, "write GB/launch", sum(v) / len(v))
PY
