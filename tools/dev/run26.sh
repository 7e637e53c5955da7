# round-3 GPU call: final validation of HEAD — full GPU suite, smoke, bench, rocprof / PMC evidence (v5)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r3m/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3m/gpu_tests.log
tail -3 gpurun_out/r3m/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r3m/bench.json 2> gpurun_out/r3m/bench.err; echo "bench rc $?"
timeout 900 bash tools/prof_round.sh v5 > gpurun_out/r3m/prof.log 2>&1; echo "prof rc $?"
tail -5 gpurun_out/r3m/prof.log
