#!/bin/bash
# GPU box: fp32 encoder time against the launch size: layer-wise true-fp32 kernels (variant 16), split-f16 kernels everywhere,
# and the row-streaming blocks / front alone (tiles and head layer-wise)
cd $GRAFT_REPO_ROOT
for b in 1 2 4 8 16 32 40 48 56 64; do
  e16=$(python tools/stage_times.py --obs-batch $b --iters 20 --enc fp32 --variant 16 2>&1 | tail -1 | sed 's/.*encode \([0-9.]*\) us.*/\1/')
  eall=$(RIP_SPLIT_TILE_MIN=1 RIP_SPLIT_ROWS_MIN=1 python tools/stage_times.py --obs-batch $b --iters 20 --enc fp32 2>&1 | tail -1 | sed 's/.*encode \([0-9.]*\) us.*/\1/')
  erows=$(RIP_SPLIT_TILE_MIN=100000 RIP_SPLIT_ROWS_MIN=1 python tools/stage_times.py --obs-batch $b --iters 20 --enc fp32 2>&1 | tail -1 | sed 's/.*encode \([0-9.]*\) us.*/\1/')
  echo "B=$b (x 4 models): layer-wise $e16 us | split kernels everywhere $eall us | row-streaming blocks + front only $erows us"
done
