# round-3 GPU call 1: micro-benchmark of the split-f16 step, GPU tests, bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 120 tools/micro/split_f16 > gpurun_out/r3a/split_f16.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3a/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3a/gpu_tests.log
timeout 600 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc $?"
tail -5 gpurun_out/r3a/gpu_tests.log
cat gpurun_out/r3a/split_f16.log
