#!/bin/bash
# GPU box: encoder time of prebuilt variants under an encoder-variant mask ($VAR)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp oatomobile_amd/librip_hip.so build_abl/base.so
for v in "$@"; do
  cp build_abl/$v.so oatomobile_amd/librip_hip.so
  python tools/stage_times.py --obs-batch 512 --iters 40 --enc bf16 --variant ${VAR:-0} 2>&1 | grep "B=" | sed "s/^/$v /" | tee -a gpurun_out/enc_sweep.log
done
cp build_abl/base.so oatomobile_amd/librip_hip.so
