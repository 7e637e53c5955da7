#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp oatomobile_amd/librip_hip.so build_abl/base.so
for v in base abl1 abl3 abl4; do
  cp build_abl/$v.so oatomobile_amd/librip_hip.so
  for w in 8 4; do
    RIP_SPLIT_WPB=$w python tools/stage_times.py --obs-batch 512 --iters 20 --enc bf16 2>&1 | grep "B=" | sed "s/^/$v wpb=$w /" | tee -a gpurun_out/wpb8_abl.log
  done
done
cp build_abl/base.so oatomobile_amd/librip_hip.so
