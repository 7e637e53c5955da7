#!/bin/bash
# GPU box: encoder time kernel by kernel for prebuilt variants ("base" = HEAD's library); $GREP selects the lines kept
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp oatomobile_amd/librip_hip.so build_abl/base.so
for v in "$@"; do
  cp build_abl/$v.so oatomobile_amd/librip_hip.so
  python tools/stage_times.py --obs-batch 512 --iters 40 --enc bf16 --variant ${VAR:-0} --blocks 2>&1 | grep -E "B=|${GREP:-blk}" | sed "s/^/$v /" | tee -a gpurun_out/enc_blocks.log
done
cp build_abl/base.so oatomobile_amd/librip_hip.so
