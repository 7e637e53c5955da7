#!/bin/bash
# GPU box: per-phase shader cycles of the fp32 encoder's row-streaming blocks and front (encoder_split_rows.hip with -DRIP_ROWS_TICKS)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export RIP_SOURCE_FLAGS="encoder_split_rows.hip=-DRIP_ROWS_TICKS $1"
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/rows_ticks_build.log 2>&1 || { tail -5 gpurun_out/rows_ticks_build.log; exit 1; }
python tools/stage_times.py --obs-batch 512 --iters 2 --enc fp32 2>&1 | grep "^split rows<\|^split front<" | tail -7 | tee gpurun_out/rows_ticks.log
