#!/bin/bash
# SQ counters of the kernels outside the closed loop (offline LPs, scenario MPC, support sweep, set learning) on the default
# bench.py legs: two passes of <= 8 SQ counters -> gpurun_out/other/sq_summary.json (per kernel: per-launch maxima, fractions of wave cycles)
cd "$(dirname "$0")/.."
out=gpurun_out/other; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $out/sq$i -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/sq$i.log 2>&1 || { tail -5 $out/sq$i.log; exit 1; }
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
        if "lbmpc::" in k and "lbmpc::mc_" not in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for fld in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size"):
                if fld in r: sq.setdefault(k, {})[fld] = r[fld]
    for k, d in agg.items():
        for cn, v in d.items():
            sq.setdefault(k, {})[cn + "_per_launch_max"] = max(v)
            sq[k]["launches"] = len(v)
for k, d in sq.items():
    wc = d.get("SQ_WAVE_CYCLES_per_launch_max")
    if wc:
        for cn in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT"):
            if cn + "_per_launch_max" in d: d[cn + "_frac_of_wave_cycles"] = d[cn + "_per_launch_max"] / wc
json.dump({"note": "rocprofv3 VGPR_Count is in granules of 2 registers; per-launch maxima over the launches of the default bench.py legs", "kernels": sq},
          open(f"{out}/sq_summary.json", "w"), indent=1)
for k, v in sq.items(): print(k, {a: round(v[a], 3) for a in v if "frac" in a}, v.get("VGPR_Count"), v.get("launches"))
PY
