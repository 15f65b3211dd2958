#!/bin/bash
# SQ / instruction-cache counters of the MC kernels at the default bench size (two separate passes)
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
PASSES=("$@"); [ ${#PASSES[@]} -eq 0 ] && PASSES=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_IFETCH")
for pass in "${PASSES[@]}"; do
  rm -rf gpurun_out/psq
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/psq -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --sl-sets 64 > gpurun_out/psq.log 2>&1 || { tail -5 gpurun_out/psq.log; continue; }
  python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/psq/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "mc_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items(): print("mc_kernel", k, "%.4g" % (sum(v) / len(v)))
PY
done
