# GPU box: compiler scheduling strategy of flow_split.hip (re-swept in round 6 after the forward step changed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sched
S="-mllvm -amdgpu-sched-strategy"
for st in "$S=iterative-maxocc" "$S=max-memory-clause" "$S=max-ilp" "$S=iterative-ilp" "$S=iterative-minreg" ""; do
  RIP_SOURCE_FLAGS="flow_split.hip=$st" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/sched/build.log 2>&1 || { echo "build failed [$st]"; continue; }
  echo "== [$st] $(RIP_SOURCE_FLAGS="flow_split.hip=$st" timeout 300 python tools/stage_times.py --obs-batch 2048 --iters 8 --enc bf16 2>&1 | tail -1)"
done
