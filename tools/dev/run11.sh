cd $GRAFT_REPO_ROOT
echo "== merged lo accumulators"; timeout 120 tools/micro/split_f16 2>&1 | grep -E "numerics|split-f16 step, [12] wave"
echo "== separate"; timeout 120 tools/micro/split_f16_nomerge 2>&1 | grep -E "split-f16 step, [12] wave"
T="python tools/stage_times.py --obs-batch 512 --iters 20 --enc bf16 --search-kernel 4"
$T 2>&1 | grep "B=" | sed "s/^/merged /"
$T 2>&1 | grep "B=" | sed "s/^/merged (again) /"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -s -k "teacher_forced and split" 2>&1 | grep -E "teacher-forced|passed|failed"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
