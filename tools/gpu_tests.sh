# full GPU suite as the driver runs it (one process, per-test timeout)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -15 gpurun_out/gpu_tests.log
exit $rc
