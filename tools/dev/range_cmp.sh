#!/bin/bash
# GPU box: test_split_kernel_operand_ranges numbers under several prebuilt libraries (build_abl/<name>.so)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp oatomobile_amd/librip_hip.so build_abl/_cur.so
for v in "$@"; do
  cp build_abl/$v.so oatomobile_amd/librip_hip.so
  echo "== $v" | tee -a gpurun_out/range_cmp.log
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -rA -k "operand_ranges" 2>&1 | grep -E "^z x|passed|failed" | sort -u | tee -a gpurun_out/range_cmp.log
done
cp build_abl/_cur.so oatomobile_amd/librip_hip.so
