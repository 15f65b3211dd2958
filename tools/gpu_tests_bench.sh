cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -8 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 3 --warmup 1 $BENCH_ARGS > gpurun_out/b.log 2>&1 || { tail gpurun_out/b.log; exit 1; }
tail -1 gpurun_out/b.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench: %.4g steps/s  %.2f ms/launch kern_ms %.2f set-learning %.4g samples/s sweep %.1f us cpu %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['set_learning']['value'], d['support_sweep']['ms_per_step']*1e3, d['cpu_baseline']['value'] if d['cpu_baseline'] else None))
print(d['roofline']['executed']['counters'])"
