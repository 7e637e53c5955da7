# GPU box: the fp32 encoder's split-f16 tile blocks: parity test, per-phase cycles, stage times
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "split_tile or layerwise_encoder or g5_params" 2>&1 | grep -v "^$" | tail -12
bash tools/dev/split_sweep.sh "$1"
