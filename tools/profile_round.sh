#!/bin/bash
# Round profile: (1) default bench.py under rocprofv3 --kernel-trace --stats, (2) separate --pmc passes (FETCH_SIZE,
# WRITE_SIZE) for the HBM traffic, (3) SQ passes (<= 8 SQ counters each) for occupancy / VALU utilisation of the MC
# kernels.  Outputs under gpurun_out/round/; copy what is to be judged into profiles/ (named per round).
cd "$(dirname "$0")/.."
out=gpurun_out/round; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python bench.py > $out/bench.log 2>&1 || { tail $out/bench.log; exit 1; }
grep '^{"metric"' $out/bench.log > $out/bench.json
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
LEAN="--steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python bench.py $LEAN > $out/pmc_$c.log 2>&1 || { tail $out/pmc_$c.log; exit 1; }
  cp $(find $out/pmc_$c -name "*counter_collection.csv" | head -1) $out/pmc_$c.csv
done
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $out/sq$i -- python bench.py $LEAN > $out/sq$i.log 2>&1 || { tail -5 $out/sq$i.log; continue; }
  cp $(find $out/sq$i -name "*counter_collection.csv" | head -1) $out/sq$i.csv
done
python - <<'PY'
import csv, collections, glob, hashlib, json, os
out = "gpurun_out/round"
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B: doubled (MI355X_MICROARCH.md, HBM)
summ = {}
for name, scale, key in (("pmc_FETCH_SIZE", 2.0 * 1024.0, "hbm_read_bytes"), ("pmc_WRITE_SIZE", 1024.0, "hbm_write_bytes")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{out}/{name}.csv")):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "lbmpc" in k:
            summ.setdefault(k, {})[key] = sum(v) / len(v) * scale
            summ[k][key.replace("bytes", "total_bytes")] = sum(v) * scale
            summ[k]["launches"] = len(v)
# the lean run does warmup 1 + steps 1 = 2 bench steps of the MC leg
mc = sum(v.get("hbm_read_total_bytes", 0) + v.get("hbm_write_total_bytes", 0) for k, v in summ.items() if "mc_" in k) / 2.0
lib = "learning-based-mpc_amd/csrc/liblbmpc_hip.so"
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
        for cn, v in d.items():
            sq.setdefault(k, {})[cn + "_per_launch_max"] = max(v)   # the full-size launch (resume launches are tiny)
for k, d in sq.items():
    wc = d.get("SQ_WAVE_CYCLES_per_launch_max")
    if wc:
        for cn in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if cn + "_per_launch_max" in d: d[cn + "_frac_of_wave_cycles"] = d[cn + "_per_launch_max"] / wc
json.dump({"note": "per-launch means; FETCH_SIZE x2 x1024 (gfx950 correction), WRITE_SIZE x1024; default bench.py sizes; "
                   "mc_bytes_per_bench_step = read + write of all mc_* launches of one bench step",
           "library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(), "mc_bytes_per_bench_step": mc,
           "kernels": summ, "sq": sq}, open(f"{out}/pmc_summary.json", "w"), indent=1)
for k, v in summ.items(): print(k, {a: "%.4g" % b for a, b in v.items()})
for k, v in sq.items(): print(k, {a: v[a] for a in v if "frac" in a or "Count" in a or "Size" in a})
PY
grep -E "lbmpc" $out/kernel_stats.csv | cut -d, -f1-4 | cut -c1-70,150-
