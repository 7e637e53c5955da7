#!/bin/bash
# GPU box: per-phase shader cycles of the fp32 encoder's split-f16 tile blocks (encoder_split_tile.hip with -DRIP_SPLIT_TICKS)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export RIP_SOURCE_FLAGS="encoder_split_tile.hip=-DRIP_SPLIT_TICKS $RIP_SPLIT_EXTRA"
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/split_ticks_build.log 2>&1 || { tail -5 gpurun_out/split_ticks_build.log; exit 1; }
python tools/stage_times.py --obs-batch 512 --iters 2 --enc fp32 2>&1 | grep "^split tile<" | tail -10 | tee gpurun_out/split_ticks.log
