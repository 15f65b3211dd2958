#!/bin/bash
# copy the summaries of the last tools/profile_round.sh run (gpurun_out/round) into profiles/ under the round's prefix
# ($1, e.g. r02); the per-dispatch counter csvs are cut down to the library's kernels
cd "$(dirname "$0")/.."
p=${1:-r03}; src=gpurun_out/round
cp $src/bench.json profiles/${p}_bench_under_rocprof.json
cp $src/kernel_stats.csv profiles/${p}_kernel_stats.csv
cp $src/pmc_summary.json profiles/${p}_pmc_summary.json
for pair in "pmc_FETCH_SIZE:pmc_fetch_size" "pmc_WRITE_SIZE:pmc_write_size" "sq1:sq_pass1" "sq2:sq_pass2"; do
  a=${pair%%:*}; b=${pair##*:}
  pat=lbmpc; case $a in sq*) pat="lbmpc::mc_";; esac   # SQ passes: the closed-loop kernels only
  [ -f $src/$a.csv ] && { head -1 $src/$a.csv; grep "$pat" $src/$a.csv; } > profiles/${p}_$b.csv
done
ls -la profiles/${p}_*
