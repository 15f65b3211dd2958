cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export LBMPC_LIB=$PWD/learning-based-mpc_amd/csrc/dev_lbmpc.so
rm -rf gpurun_out/ic; mkdir -p gpurun_out/ic
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/ic -- python tools/qp_legs.py C1 C3 --reps 2 > gpurun_out/ic.log 2>&1 || { tail -5 gpurun_out/ic.log; exit 1; }
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/ic/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "lbmpc" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items(): print(k, {c: "%.4g" % max(v) for c, v in d.items()})
PY
