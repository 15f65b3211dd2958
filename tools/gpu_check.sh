#!/bin/bash
# GPU parity tests, then phase profile, then a short bench (each under its own time limit; stops at the first failure)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
if [ -f learning-based-mpc_amd/csrc/prof_lbmpc.so ]; then tools/prof.sh > gpurun_out/prof.log 2>&1 || { tail gpurun_out/prof.log; exit 1; }; cat gpurun_out/prof.log; fi
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b.log 2>&1 || { tail gpurun_out/b.log; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b.log").read().strip().splitlines()[-1])
print("bench: %.4g steps/s  %.2f ms/launch  set-learning %.4g samples/s" % (d["value"], d["ms_per_step"], d["set_learning"]["value"]))
PY
