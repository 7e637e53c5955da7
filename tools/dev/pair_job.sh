# GPU box, round 6: the paired split-f16 search kernel — parity gates (guarded by a timeout: a broken exchange spins
# forever), then the launch time beside the one-wave shape.   bash tools/dev/pair_job.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pair
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "pair" > gpurun_out/pair/tests.log 2>&1; echo "tests rc=$?"
grep -n "passed\|failed\|Error\|error\|teacher-forced\|candidates outside\|adjoints per" gpurun_out/pair/tests.log | cut -c1-230 | tail -40
for k in 4 5; do
  timeout 300 python tools/stage_times.py --obs-batch 512 --iters 20 --enc bf16 --search-kernel $k 2>&1 | tail -1
done
