#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp oatomobile_amd/librip_hip.so build_abl/_cur.so
for v in "$@"; do
  cp build_abl/$v.so oatomobile_amd/librip_hip.so
  echo "== $v" | tee -a gpurun_out/range_seeds.log
  python tools/dev/range_seeds.py 10.0 1.0 11 12 13 14 15 16 2>&1 | grep -E "^z x|seed" | tee -a gpurun_out/range_seeds.log
done
cp build_abl/_cur.so oatomobile_amd/librip_hip.so
