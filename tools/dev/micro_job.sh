#!/bin/bash
# GPU box: run the prebuilt micro-benchmark variants (build_abl/micro/*), log to gpurun_out/micro.log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for b in "$@"; do
  echo "== $b" | tee -a gpurun_out/micro.log
  timeout 120 build_abl/micro/$b 2>&1 | grep -E "numerics|cycles/step" | tee -a gpurun_out/micro.log
done
