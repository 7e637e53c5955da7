# GPU box: compiler scheduling strategy of the row-streaming / front encoder kernels (re-swept in round 6 after the depthwise changed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sched
S="-mllvm -amdgpu-sched-strategy"
for f in encoder_bf16_irb2.hip encoder_bf16_front2.hip; do
for st in "" "$S=iterative-maxocc" "$S=max-memory-clause" "$S=max-ilp" "$S=iterative-ilp"; do
  RIP_SOURCE_FLAGS="$f=$st" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/sched/build.log 2>&1 || { echo "build failed [$f $st]"; continue; }
  echo "== [$f $st] $(RIP_SOURCE_FLAGS="$f=$st" timeout 300 python tools/stage_times.py --obs-batch 2048 --iters 8 --enc bf16 2>&1 | tail -1)"
done; done
