# round-3 GPU call 6: DMA overlap on / off, full GPU suite, bench, rocprof evidence (v2)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
T="python tools/stage_times.py --obs-batch 512 --iters 20 --enc bf16 --search-kernel 4"
$T 2>&1 | grep "B=" | sed "s/^/HEAD (overlap) /"
RIP_EXTRA_HIPCC_FLAGS="-DRIP_SPLIT_OVERLAP=0" python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i " error" | head -3
RIP_EXTRA_HIPCC_FLAGS="-DRIP_SPLIT_OVERLAP=0" $T 2>&1 | grep "B=" | sed "s/^/-DRIP_SPLIT_OVERLAP=0 /"
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i " error" | head -3
for b in 1 8 32 128 256; do python tools/stage_times.py --obs-batch $b --iters 20 --enc bf16 2>&1 | grep "B=" | sed "s/^/auto /"; done
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/r3f/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3f/gpu_tests.log
tail -4 gpurun_out/r3f/gpu_tests.log
timeout 900 python bench.py > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err; echo "bench rc $?"
timeout 900 bash tools/prof_round.sh v2 > gpurun_out/r3f/prof.log 2>&1; echo "prof rc $?"
