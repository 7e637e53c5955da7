# GPU box: per-source compiler scheduling options (only the named object is rebuilt per variant).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sched
run() {
  RIP_SOURCE_FLAGS="$1" python __graft_entry__.py > gpurun_out/sched/build.log 2>&1 || { echo "build failed: $1"; tail -3 gpurun_out/sched/build.log; return; }
  echo "== [$1] $(RIP_SOURCE_FLAGS="$1" python tools/dev/misc_times.py 2>&1 | tail -1) $(RIP_SOURCE_FLAGS="$1" python tools/stage_times.py --obs-batch 512 --iters 10 --enc fp32 2>&1 | tail -1)"
}
S="-mllvm -amdgpu-sched-strategy"
run ""
for st in max-memory-clause iterative-maxocc max-ilp; do
  run "train.hip=$S=$st"
  run "encoder.hip=$S=$st;encoder_fused.hip=$S=$st"
  run "flow_phase.hip=$S=$st;flow_split.hip=$S=$st;flow.hip=$S=$st"
done
run ""
