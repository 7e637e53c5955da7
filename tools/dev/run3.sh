# round-3 GPU call 3: full GPU suite, bench, rocprof evidence of the split-f16 build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/r3c/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3c/gpu_tests.log
tail -4 gpurun_out/r3c/gpu_tests.log
timeout 600 python bench.py > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err; echo "bench rc $?"
timeout 900 bash tools/prof_round.sh v1 > gpurun_out/r3c/prof.log 2>&1; echo "prof rc $?"
tail -30 gpurun_out/r3c/prof.log
