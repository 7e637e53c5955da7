cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 600 python -m pytest tests/test_gpu_parity.py -q -s -k "regrouping or (search_candidates_vs_oracle and split) or bench_configuration or config4 or mfma_kernel_matches" > gpurun_out/r3h/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3h/tests.log
grep -E "adjoints per|passed|failed|rc |Error|assert" gpurun_out/r3h/tests.log | tail -12
T="python tools/stage_times.py --obs-batch 512 --iters 20 --enc bf16 --search-kernel 4"
$T 2>&1 | grep "B=" | sed "s/^/regroup on  /"
RIP_EXTRA_HIPCC_FLAGS="-DRIP_SPLIT_SORT=0" python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i " error" | head -3
RIP_EXTRA_HIPCC_FLAGS="-DRIP_SPLIT_SORT=0" $T 2>&1 | grep "B=" | sed "s/^/regroup off /"
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i " error" | head -3
for b in 128 256; do python tools/stage_times.py --obs-batch $b --iters 20 --enc bf16 2>&1 | grep "B=" | sed "s/^/regroup on  /"; done
