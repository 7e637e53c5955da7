# round-3 GPU call 2: the split-f16 search kernel — parity tests, then launch time per workgroup shape
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "g6_search_traces or teacher_forced or search_candidates_vs_oracle or mfma_kernel_matches or r11 or graph_equals_eager" > gpurun_out/r3b/split_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3b/split_tests.log
grep -v "^$" gpurun_out/r3b/split_tests.log | grep "split\|passed\|failed\|rc " | tail -40
for k in 3 4; do python tools/stage_times.py --obs-batch 512 --iters 20 --enc bf16 --search-kernel $k 2>&1 | grep "B=" | sed "s/^/kernel=$k /"; done
for w in 8 4 2; do RIP_SPLIT_WPB=$w python tools/stage_times.py --obs-batch 512 --iters 20 --enc bf16 --search-kernel 4 2>&1 | grep "B=" | sed "s/^/split wpb=$w /"; done
for b in 256 128 64; do python tools/stage_times.py --obs-batch $b --iters 20 --enc bf16 --search-kernel 4 2>&1 | grep "B=" | sed "s/^/split auto /"; done
