cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tile; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace -d $O/t --output-format csv -- python tools/stage_times.py --obs-batch ${1:-512} --iters 5 --enc bf16 --fused 17 > $O/log.txt 2>&1
python tools/trace_timeline.py $O/t > $O/timeline.txt 2>&1
grep -v "^pw_bf16\|^dw_bf16" $O/timeline.txt | tail -40
