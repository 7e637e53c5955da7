#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp oatomobile_amd/librip_hip.so build_abl/base.so
for v in "$@"; do
  cp build_abl/$v.so oatomobile_amd/librip_hip.so
  python tools/online_probe.py 600 2>&1 | grep graph=True | sed "s/^/$v /" | tee -a gpurun_out/online_sweep.log
done
cp build_abl/base.so oatomobile_amd/librip_hip.so
