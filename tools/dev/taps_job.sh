# GPU box, round 6: bf16-valued depthwise taps in the bf16 encoder — the bf16 gates, then the encoder time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/taps
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bf16 or bench_configuration or cil or replay or four_channel or smoke" > gpurun_out/taps/tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/taps/tests.log | cut -c1-300
for b in 512 2048; do timeout 300 python tools/stage_times.py --obs-batch $b --iters 10 --enc bf16 --blocks 2>&1 | grep -v amdgpu | grep "blk \|B=" | cut -c1-110; done
timeout 300 python tools/stage_times.py --obs-batch 1 --iters 200 --enc bf16 2>&1 | tail -1
