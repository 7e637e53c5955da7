#!/bin/bash
# GPU box: per-phase shader cycles of the tile blocks (build_abl/ticks.so = encoder_bf16_tile.hip with -DRIP_TILE_TICKS)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp oatomobile_amd/librip_hip.so build_abl/base.so
cp build_abl/${1:-ticks}.so oatomobile_amd/librip_hip.so
python tools/stage_times.py --obs-batch 512 --iters 2 --enc bf16 2>&1 | grep "^tile<" | tail -10 | tee gpurun_out/tile_ticks.log
cp build_abl/base.so oatomobile_amd/librip_hip.so
