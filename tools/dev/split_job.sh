# GPU box: the fp32 encoder's split-f16 blocks: parity test, kernel timeline, stage time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "split_tile or layerwise_encoder or g5_params" 2>&1 | grep -v "^$" | tail -${2:-14}
bash tools/dev/ktrace32.sh 2>&1 | grep -v "irb_split_tile" | tail -22
python tools/stage_times.py --obs-batch 512 --iters 8 --enc fp32 2>&1 | tail -1
