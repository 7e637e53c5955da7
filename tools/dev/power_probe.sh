#!/bin/bash
# GPU box: sample shader clock and socket power while the act() loop runs (is the search launch power-capped?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/stage_times.py --obs-batch 512 --iters 1500 --enc bf16 > gpurun_out/power_stage.log 2>&1 &
pid=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi -d 0 --showpower --showclocks --showperflevel 2>&1 | grep -E "sclk|Power|mclk|fclk|Perf" | tr '\n' ' ' | tee -a gpurun_out/power.log; echo | tee -a gpurun_out/power.log
  sleep 0.7
done
wait $pid
cat gpurun_out/power_stage.log | grep "B=" | tee -a gpurun_out/power.log
rocm-smi -d 0 --showpower --showclocks 2>&1 | grep -E "sclk|Power" | tr '\n' ' ' | tee -a gpurun_out/power.log; echo
