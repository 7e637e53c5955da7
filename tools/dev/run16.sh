# round-3 GPU call: HEAD validation — full GPU suite, smoke, bench, rocprof evidence (v3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/r3i/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3i/gpu_tests.log
tail -3 gpurun_out/r3i/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err; echo "bench rc $?"
timeout 900 bash tools/prof_round.sh v3 > gpurun_out/r3i/prof.log 2>&1; echo "prof rc $?"
python tools/online_probe.py 1000 2>&1 | grep "graph="
