# GPU tests, bench, then per-kernel times (stops at the first failure)
cd $GRAFT_REPO_ROOT
bash tools/gpu_tests_bench.sh || exit 1
bash tools/kernel_times.sh learning-based-mpc_amd/csrc/liblbmpc_hip.so
