# GPU box: the bf16 encoder gates + the one-observation and small-batch paths after a change of the kernel selection
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bf16 or bench_configuration or cil or replay or four_channel or online or graph or batched or abi" 2>&1 | tail -3 | cut -c1-300
for b in 1 4 16 64 512; do timeout 300 python tools/stage_times.py --obs-batch $b --iters 30 --enc bf16 2>&1 | tail -1; done
