#!/bin/bash
# GPU box: build variants of encoder_split_tile.hip (-D... sets in $@, separated by ';') with tick counters, print per-phase cycles and encoder time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
IFS=';' read -ra SETS <<< "$1"
for set in "${SETS[@]}"; do
  export RIP_SOURCE_FLAGS="encoder_split_tile.hip=-DRIP_SPLIT_TICKS $set"
  python -c "import __graft_entry__ as g; g.build()" > gpurun_out/split_sweep_build.log 2>&1 || { echo "build failed [$set]"; tail -5 gpurun_out/split_sweep_build.log; continue; }
  echo "=== [$set]"
  python tools/stage_times.py --obs-batch 512 --iters 2 --enc fp32 2>&1 | grep "^split tile<" | tail -10 | awk '!seen[$2]++' | cut -c1-300
  export RIP_SOURCE_FLAGS="encoder_split_tile.hip=$set"
  python -c "import __graft_entry__ as g; g.build()" > gpurun_out/split_sweep_build.log 2>&1
  python tools/stage_times.py --obs-batch 512 --iters 8 --enc fp32 2>&1 | tail -1
done
