cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/prof.sh > gpurun_out/prof.log 2>&1 || { tail gpurun_out/prof.log; exit 1; }
cat gpurun_out/prof.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/b.log 2>&1 || { tail gpurun_out/b.log; exit 1; }
tail -1 gpurun_out/b.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench: %.4g steps/s  %.2f ms/launch kern_ms %.2f set-learning %.4g samples/s sweep %.1f us cpu %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['set_learning']['value'], d['support_sweep']['ms_per_step']*1e3, d['cpu_baseline']['value'] if d['cpu_baseline'] else None))
print(d['roofline']['executed']['counters'])"
