# GPU box: the fp32 encoder's split-f16 blocks: parity test, tile ticks, kernel timeline, stage time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "split_tile or layerwise_encoder or g5_params" 2>&1 | grep -v "^$" | tail -${2:-14}
bash tools/dev/split_ticks.sh 2>&1 | awk '!seen[$2]++' | cut -c1-330
unset RIP_SOURCE_FLAGS
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bash tools/dev/ktrace32.sh 2>&1 | grep "irb_split_tile\|head_split" | tail -11
python tools/stage_times.py --obs-batch 512 --iters 8 --enc fp32 2>&1 | tail -1
