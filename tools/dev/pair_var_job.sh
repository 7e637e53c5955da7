# GPU box, round 6: build variants of the paired search kernel (only flow_pair.hip is recompiled per variant) and time
# one 512-observation launch each; variant "ticks" prints per-phase cycle counters.   bash tools/dev/pair_var_job.sh "v1;v2;..."
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pair
IFS=';' read -ra VARS <<< "${1:-base}"
for v in "${VARS[@]}"; do
  f="${v#*=}"; n="${v%%=*}"
  if [ "$n" = "$v" ]; then f=""; fi
  RIP_SOURCE_FLAGS="flow_pair.hip=$f" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/pair/build_$n.log 2>&1 || { echo "build failed: $n"; tail -5 gpurun_out/pair/build_$n.log; continue; }
  echo "== $n [$f]"
  RIP_SOURCE_FLAGS="flow_pair.hip=$f" timeout 300 python tools/stage_times.py --obs-batch 512 --iters ${ITERS:-10} --enc bf16 --search-kernel 5 2>&1 | grep -v "^$" | tail -${TAILN:-1}
done
