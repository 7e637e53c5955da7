# GPU box: encoder timeline under the environment given as arguments ("VAR=val ..." per run, separated by '--').
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timing; rm -rf $O; mkdir -p $O; cd $R
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs timeout 300 rocprofv3 --kernel-trace -d $O/t$i --output-format csv -- python tools/stage_times.py --obs-batch 512 --iters 8 --enc bf16 > $O/log$i.txt 2>&1
  python tools/trace_timeline.py $O/t$i > $O/timeline$i.txt 2>&1
  echo "== $envs"; tail -1 $O/log$i.txt; grep "${FILTER:-front\|irb2\|sum of}" $O/timeline$i.txt
done
