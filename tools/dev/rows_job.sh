# GPU box: row-streaming blocks / front: parity test, ticks, stage time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "split_tile" 2>&1 | grep -v "^$" | tail -8
bash tools/dev/rows_ticks.sh "$1" | cut -c1-300
unset RIP_SOURCE_FLAGS
export RIP_SOURCE_FLAGS="encoder_split_rows.hip=$1"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
python tools/stage_times.py --obs-batch 512 --iters 8 --enc fp32 2>&1 | tail -1
