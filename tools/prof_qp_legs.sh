#!/bin/bash
# kernel times + SQ counter passes of the QP-active legs (C1 -> mc_kernel<2,1>, C3 -> mc_kernel<3,1>); $1 = output tag
cd "$(dirname "$0")/.."
tag=${1:-qp}; out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 200 python -u tools/qp_legs.py C1 C3 --check > $out/legs.log 2>&1 || { tail $out/legs.log; exit 1; }
cat $out/legs.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python tools/qp_legs.py C1 C3 > $out/kt.log 2>&1 || { tail $out/kt.log; exit 1; }
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $out/sq$i -- python tools/qp_legs.py C1 C3 --reps 2 > $out/sq$i.log 2>&1 || { tail -5 $out/sq$i.log; exit 1; }
  cp $(find $out/sq$i -name "*counter_collection.csv" | head -1) $out/sq$i.csv
done
python - "$out" <<'PY'
import csv, collections, glob, json, sys
out = sys.argv[1]
sq = {}
for f in sorted(glob.glob(f"{out}/sq*.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "lbmpc::mc_" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for fld in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
                if fld in r: sq.setdefault(k, {})[fld] = r[fld]
    for k, d in agg.items():
        for cn, v in d.items(): sq.setdefault(k, {})[cn + "_per_launch_max"] = max(v)
for k, d in sq.items():
    wc = d.get("SQ_WAVE_CYCLES_per_launch_max")
    if wc:
        for cn in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT"):
            if cn + "_per_launch_max" in d: d[cn + "_frac_of_wave_cycles"] = d[cn + "_per_launch_max"] / wc
json.dump(sq, open(f"{out}/sq_summary.json", "w"), indent=1)
for k, v in sq.items(): print(k, {a: (("%.3g" % v[a]) if isinstance(v[a], float) else v[a]) for a in v})
PY
grep -E "lbmpc" $out/kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,140-
