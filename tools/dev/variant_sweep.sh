#!/bin/bash
# GPU box: time the search (and encoder) of prebuilt build_abl/<name>.so variants; parity-check the ones named after "--".
#   tools/dev/variant_sweep.sh base p0 p0pipe -- p0pipe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp oatomobile_amd/librip_hip.so build_abl/base.so
out=gpurun_out/variant_sweep.log
check=0
for v in "$@"; do
  if [ "$v" == "--" ]; then check=1; continue; fi
  cp build_abl/$v.so oatomobile_amd/librip_hip.so
  if [ $check == 0 ]; then
    for rep in 1 2; do
      python tools/stage_times.py --obs-batch ${SWEEP_B:-512} --iters 30 --enc bf16 2>&1 | grep "B=" | sed "s/^/$v /" | tee -a $out
    done
  else
    echo "parity $v" | tee -a $out
    timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "${SWEEP_K:-g6 or teacher or bench_configuration or operand_ranges}" 2>&1 | tail -3 | tee -a $out
  fi
done
cp build_abl/base.so oatomobile_amd/librip_hip.so
