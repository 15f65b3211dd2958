#!/bin/bash
# Round profile: (1) default bench.py under rocprofv3 --kernel-trace --stats, (2) a separate --pmc pass (FETCH_SIZE) on
# a set-learning-only run for the HBM traffic of halfspace_max_kernel.  Outputs under gpurun_out/round/.
cd "$(dirname "$0")/.."
out=gpurun_out/round; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python bench.py > $out/bench.log 2>&1 || { tail $out/bench.log; exit 1; }
grep '^{"metric"' $out/bench.log > $out/bench.json
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1 || { tail $out/pmc_fetch.log; exit 1; }
cp $(find $out/pmc_fetch -name "*counter_collection.csv" | head -1) $out/pmc_fetch.csv
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/pmc_write.log 2>&1 || { tail $out/pmc_write.log; exit 1; }
cp $(find $out/pmc_write -name "*counter_collection.csv" | head -1) $out/pmc_write.csv
python - <<'PY'
import csv, collections, json
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B: doubled (MI355X_MICROARCH.md, HBM)
summ = {}
for name, scale in (("pmc_fetch", 2.0 * 1024.0), ("pmc_write", 1024.0)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"gpurun_out/round/{name}.csv")):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "lbmpc" in k:
            summ.setdefault(k, {})["hbm_read_bytes" if name == "pmc_fetch" else "hbm_write_bytes"] = sum(v) / len(v) * scale
            summ[k]["launches"] = len(v)
json.dump({"note": "per-launch means; FETCH_SIZE x2 x1024 (gfx950 correction), WRITE_SIZE x1024; default bench.py sizes", "kernels": summ},
          open("gpurun_out/round/pmc_summary.json", "w"), indent=1)
for k, v in summ.items(): print(k, v)
PY
grep -E "lbmpc" $out/kernel_stats.csv | cut -d, -f1-4 | cut -c1-70,150-
