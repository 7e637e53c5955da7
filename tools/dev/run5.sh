# round-3 GPU call 5: the 8-wave split build after the register work (late tape loads, parked Adam state)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
T="python tools/stage_times.py --obs-batch 512 --iters 20 --enc bf16 --search-kernel 4"
for w in 4 8; do RIP_SPLIT_WPB=$w $T 2>&1 | grep "B=" | sed "s/^/HEAD wpb=$w /"; done
RIP_SPLIT_WPB=8 timeout 600 python -m pytest tests/test_gpu_parity.py -q -s -k "teacher_forced and split or search_candidates_vs_oracle and split or g6_search_traces and split" > gpurun_out/r3e/tests8.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3e/tests8.log
grep -E "teacher-forced|passed|failed|rc " gpurun_out/r3e/tests8.log | tail -8
timeout 600 python -m pytest tests/test_gpu_parity.py -q -s -k "packed_cache" > gpurun_out/r3e/cache.log 2>&1; tail -15 gpurun_out/r3e/cache.log
